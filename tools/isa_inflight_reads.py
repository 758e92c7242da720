"""Instructions that READ a register whose LDS read may still be in flight (no GPU needed; heuristic, straight-line scan).

The ring kernels issue their fragment reads as inline-assembly `ds_read_b128` and wait for them with hand-written `s_waitcnt lgkmcnt`:
the compiler does not know the destination registers are not valid yet, and where two code paths define the fragments differently
it may reconcile them with `v_mov` copies placed IN FRONT of the wait (round 6: the unit-end path of gemm3_kernel did exactly that -
benign in practice, the copies came four MFMAs behind the reads, but a race).  The scan follows the text of each kernel linearly
(branches ignored: hits in a different basic block than the read can be false positives - look at the listing).

    python tools/isa_inflight_reads.py /tmp/gemm.hip.s [kernel-name-substring]      # assembly from tools/isa_loop_waits.py"""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
want=sys.argv[2] if len(sys.argv)>2 else ''
def regs(tok):
    m=re.match(r'v\[(\d+):(\d+)\]',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'v(\d+)$',tok)
    if m: return {int(m.group(1))}
    return set()
name=None; pending=[]; inasm=False; hits={}
for i,l in enumerate(lines):
    m=re.match(r'^(_Z\w+):\s*; @',l)
    if m: name=m.group(1); pending=[]; continue
    if name is None or want not in name: continue
    t=l.strip()
    if 'ASMSTART' in t: inasm=True; continue
    if 'ASMEND' in t: inasm=False; continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    t=t.split(';')[0].strip()
    op=t.split()[0]
    args=[a.strip() for a in t[len(op):].split(',')]
    if op=='s_waitcnt':
        m=re.search(r'lgkmcnt\((\d+)\)',t)
        if m:
            n=int(m.group(1)); pending=pending[len(pending)-n:] if n else []
        continue
    if op.startswith('ds_read') or op.startswith('ds_load'):
        pending.append(regs(args[0])); continue
    if op.startswith('ds_write') or op.startswith('ds_store'):
        pending.append(set()); continue
    if op.startswith('s_') : continue
    # reads: all operands except first for most VALU; for stores/mfma all
    rd=set()
    for a in (args[1:] if not op.startswith(('global_store','buffer_store','scratch_store','global_load_lds')) else args):
        rd|=regs(a)
    pend=set().union(*pending) if pending else set()
    if rd & pend:
        hits.setdefault(name,[]).append((i+1,t[:70]))
for k,v in hits.items():
    print(k[:80], len(v)); 
    for x in v[:5]: print('   ',x)
