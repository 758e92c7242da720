"""HIP-event timing of the attention kernels at the benchmark shapes: (a) the padded encoder self-attention with a
key-padding mask (what bench.py's kernels.attn_*_enc_self measures), (b) the same rows PACKED (variable length, as the
training step runs them: lengths of the synthetic headline batch), (c) decoder cross-attention (T = 128 queries against
the packed memory), (d) decoder causal self-attention.  Dropout 0.2 as in the benchmarked step; PA_ATTN_OCC selects the
register budget of the backward kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from plankassembly_amd import ops
B, S, T, D, H = 16, 1024, 128, 512, 8
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)


DT = torch.float32 if os.environ.get("DT", "bf16") == "f32" else torch.bfloat16
if os.environ.get("X3", "0") == "1":          # f32 inputs, bf16x3 products (csrc/attention_x3.h)
    from plankassembly_amd import _lib as L
    L.check(L.lib().pa_attn_split_config(1), "pa_attn_split_config")


def rnd(*s):
    return torch.randn(*s, device=dev, generator=g).to(DT)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


drop = float(os.environ.get("DROP", "0.2"))
kw = dict(drop_p=drop, drop_seed=1)
# (a) padded
qkv = rnd(B, S, 3 * D)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
valid = torch.randint(S // 2, S + 1, (B,), device=dev, generator=g)
kpm = torch.arange(S, device=dev)[None] >= valid[:, None]
do = rnd(B, S, D)
o, lse = ops.attn_fwd(q, k, v, H, kpm=kpm, **kw)
fl = 4.0 * S * S * D * B
t = timeit(lambda: ops.attn_fwd(q, k, v, H, kpm=kpm, **kw))
print(f"padded  fwd        {t:8.1f} us  {fl / t / 1e6:7.1f} TF (dense-equivalent)")
t = timeit(lambda: ops.attn_bwd(do, q, k, v, o, lse, H, kpm=kpm, **kw), iters=10)
print(f"padded  bwd dq+dkv {t:8.1f} us  {2.5 * fl / t / 1e6:7.1f} TF")
# (b) packed, lengths of the synthetic headline batch (4 tokens per line + END)
rng = np.random.default_rng(2022)
lens = [4 * int(rng.integers(8, 256)) + 1 for _ in range(B)]
cu, order = ops.pack_lengths(lens, dev)
n = int(cu[-1])
qkvp = rnd(n, 3 * D)
qp, kp, vp = qkvp[:, :D], qkvp[:, D:2 * D], qkvp[:, 2 * D:]
dop = rnd(n, D)
flp = sum(4.0 * l * l * D for l in lens)
op, lsep = ops.attn_varlen_fwd(qp, kp, vp, H, cu, cu, B, S, S, order=order, **kw)
t = timeit(lambda: ops.attn_varlen_fwd(qp, kp, vp, H, cu, cu, B, S, S, order=order, **kw))
print(f"packed  fwd        {t:8.1f} us  {flp / t / 1e6:7.1f} TF  ({n} rows, lengths {min(lens)}..{max(lens)})")
t = timeit(lambda: ops.attn_varlen_bwd(dop, qp, kp, vp, op, lsep, H, cu, cu, B, S, S, order=order, **kw), iters=10)
print(f"packed  bwd dq+dkv {t:8.1f} us  {2.5 * flp / t / 1e6:7.1f} TF")
# (c) cross attention: T queries per sample (dense) against the packed memory
qc = rnd(B * T, D)
kvc = rnd(n, 2 * D)
kc, vc = kvc[:, :D], kvc[:, D:]
doc = rnd(B * T, D)
flc = sum(4.0 * T * l * D for l in lens)
oc, lsec = ops.attn_varlen_fwd(qc, kc, vc, H, None, cu, B, T, S, **kw)
t = timeit(lambda: ops.attn_varlen_fwd(qc, kc, vc, H, None, cu, B, T, S, **kw))
print(f"cross   fwd        {t:8.1f} us  {flc / t / 1e6:7.1f} TF")
t = timeit(lambda: ops.attn_varlen_bwd(doc, qc, kc, vc, oc, lsec, H, None, cu, B, T, S, **kw), iters=10)
print(f"cross   bwd dq+dkv {t:8.1f} us  {2.5 * flc / t / 1e6:7.1f} TF")
# (d) decoder causal self-attention
qd = rnd(B, T, 3 * D)
od, lsed = ops.attn_fwd(qd[..., :D], qd[..., D:2 * D], qd[..., 2 * D:], H, causal=True, **kw)
t = timeit(lambda: ops.attn_fwd(qd[..., :D], qd[..., D:2 * D], qd[..., 2 * D:], H, causal=True, **kw))
print(f"causal  fwd        {t:8.1f} us")
dod = rnd(B, T, D)
t = timeit(lambda: ops.attn_bwd(dod, qd[..., :D], qd[..., D:2 * D], qd[..., 2 * D:], od, lsed, H, causal=True, **kw), iters=10)
print(f"causal  bwd dq+dkv {t:8.1f} us")
