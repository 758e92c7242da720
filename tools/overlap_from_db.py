"""Kernel concurrency from a rocprofv3 rocpd database: fraction of the busy time with >= 2 kernels in flight."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
rows = rows[len(rows) // 2:]            # steady state: second half
ev = []
for st, en, _ in rows:
    ev.append((st, 1)); ev.append((en, -1))
ev.sort()
busy = over = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} dispatches, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, >=2 kernels in flight {over/1e6:.2f} ms ({100*over/max(busy,1):.1f} % of busy)")
