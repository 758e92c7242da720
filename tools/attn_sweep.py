"""Forward / backward encoder self-attention at S = 1024, dh = 64, H = 8 over batch sizes and mask / dropout settings: separates the
kernel's per-chunk rate from launch-shape effects (rounds of blocks over the CUs, masked tails).  HIP events, 20 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plankassembly_amd import ops
S, D, H = 1024, 512, 8
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


bwd = os.environ.get("BWD", "0") == "1"
for B in [int(x) for x in os.environ.get("BS", "12,16,24,48").split(",")]:
    qkv = torch.randn(B, S, 3 * D, device=dev, generator=g).to(torch.bfloat16)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    do = torch.randn(B, S, D, device=dev, generator=g).to(torch.bfloat16)
    valid = torch.randint(S // 2, S + 1, (B,), device=dev, generator=g)
    for masked in (False, True):
        kpm = (torch.arange(S, device=dev)[None] >= valid[:, None]) if masked else None
        for drop in (0.0, 0.2):
            kw = dict(drop_p=drop, drop_seed=1)
            if masked and os.environ.get("ORDER", "0") == "1":
                kw["order"] = ops.mask_order(kpm)
            fl = 4.0 * S * S * D * B
            flx = fl if not masked else float(sum(4.0 * S * int(x) * D for x in valid))
            t = timeit(lambda: ops.attn_fwd(q, k, v, H, kpm=kpm, **kw))
            line = f"B {B:3d} mask {int(masked)} drop {drop:.1f}  fwd {t:7.1f} us  {fl / t / 1e6:6.1f} TF dense  {flx / t / 1e6:6.1f} TF executed"
            if bwd:
                o, lse = ops.attn_fwd(q, k, v, H, kpm=kpm, **kw)
                tb = timeit(lambda: ops.attn_bwd(do, q, k, v, o, lse, H, kpm=kpm, **kw), iters=10)
                line += f"   bwd {tb:7.1f} us  {2.5 * fl / tb / 1e6:6.1f} TF dense"
            print(line, flush=True)
