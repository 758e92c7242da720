"""Absorbed cross-attention launch (pa_dec_cross_mq) next to the K/V-cache launch it replaces, at the decode benchmark's shape.
    python tools/dec_mq_bench.py [B S]        env: PLANK_DECODE_MQ_NT, PLANK_DECODE_MQ_SWAP"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plankassembly_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
torch.manual_seed(0)
mem = torch.randn(B, S, 512, device="cuda").bfloat16()
qt = (torch.randn(B, 8, 512, device="cuda") * 0.1).bfloat16()
def t(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
# several memories so that consecutive launches do not find their rows in the Infinity Cache
DT = torch.float32 if os.environ.get("DTYPE") == "f32" else torch.bfloat16
mems = [torch.randn(B, S, 512, device="cuda").to(DT) for _ in range(6)]
qt = qt.to(DT)
i = [0]
def run():
    i[0] = (i[0] + 1) % len(mems)
    ops.dec_cross_mq(qt, mems[i[0]])
us = t(run)
gb = B * S * 512 * mems[0].element_size() / 1e9
print(f"dec_cross_mq B {B} S {S} NT {os.environ.get('PLANK_DECODE_MQ_NT', '0')} SWAP {os.environ.get('PLANK_DECODE_MQ_SWAP', '1')}: {us:.1f} us  {gb / us * 1e6 / 1e3:.2f} TB/s of memory rows")
lens = torch.randint(300, S + 1, (B,))
cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0)
pm = torch.randn(int(cu[-1]), 512, device="cuda").to(DT); cud = cu.cuda()
us2 = t(lambda: ops.dec_cross_mq(qt, pm, cu=cud, S=S))
print(f"   packed rows, lengths 300..{S} ({int(cu[-1])} rows): {us2:.1f} us  {int(cu[-1]) * 512 * pm.element_size() / us2 / 1e6:.2f} TB/s")
