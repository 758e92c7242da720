"""Per-position durations of one kernel family inside the train step, from a rocprofv3 kernel-trace database of tools/step_loop.py
(4 cycling batches): position = index of the launch among ALL launches of its step; the kernel before it is named too.
python tools/step_positions.py results.db <kernel-name-substring> [launches_per_step]"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1]); want = sys.argv[2]
rows = con.execute("""select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d
                      join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
names = [re.sub(r"\(anonymous namespace\)::", "", r[0]) for r in rows]
# a step ends with the Adam kernel
ends = [i for i, n in enumerate(names) if "adam_kernel" in n]
steps = [(ends[k] + 1, ends[k + 1] + 1) for k in range(len(ends) - 1)]
steps = steps[3:]                                     # warm-up
agg = {}
for a, b in steps:
    for pos, i in enumerate(range(a, b)):
        if want in names[i]:
            key = (pos, rows[i][3] // max(rows[i][4], 1), names[i - 1][:38])
            agg.setdefault(key, []).append((rows[i][2] - rows[i][1]) / 1e3)
print(f"{len(steps)} steps; {want}: position in step, blocks, launches, avg / min / max us, kernel before")
tot = 0.0
for (pos, blocks, prev), v in sorted(agg.items()):
    print(f"  {pos:4d} {blocks:6d} {len(v):4d}   {sum(v) / len(v):7.2f} {min(v):7.2f} {max(v):7.2f}   {prev}")
    tot += sum(v)
print(f"total {tot / len(steps) / 1e3:.3f} ms per step")
