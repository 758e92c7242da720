#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) as a per-kernel table:
calls, total/avg/min/max duration, share.  Usage: rocpd_summary.py results.db [skip_first_n_dispatch_fraction]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\bvoid\s+", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    rows = cur.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d
                          join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
    agg = {}
    for name, st, en in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        dur = en - st
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    span = rows[-1][2] - rows[0][1]
    print(f"# {path}: {len(rows)} dispatches, kernel time {total / 1e6:.2f} ms, span {span / 1e6:.2f} ms")
    print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:72s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100 * a[1] / total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
