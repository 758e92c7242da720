import sqlite3,re,sys
c=sqlite3.connect(sys.argv[1])
rows=c.execute("select name, grid_x, workgroup_x, count(*), avg(end-start), min(end-start) from kernels group by name, grid_x order by name, grid_x").fetchall()
for n,g,w,cnt,avg,mn in rows:
    n=re.sub(r'\(anonymous namespace\)::','',n)
    if any(k in n for k in ('layernorm','segment_tail','embed','group_rows','pack_rows','transpose','mixture','partial_finish','attn','split','gemm','adam')):
        print(f"{cnt:6d} blocks {g//w:6d} avg {avg/1e3:8.2f} min {mn/1e3:8.2f}  {n[:90]}")
