"""Calibration only (never on the product path): what the vendor libraries reach on this box at the step's shapes -
torch.matmul (hipBLASLt / rocBLAS) for the Linears and F.scaled_dot_product_attention (whatever backend torch picks)
for the padded encoder self-attention - next to pa_gemm / pa_attn at the same shapes.  A target, not a dependency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from plankassembly_amd import ops


def t(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def rnd(*s):
    return torch.randn(*s, device="cuda").to(torch.bfloat16)


print(f"{'shape':16s} {'M':>6s} {'N':>5s} {'K':>5s} | {'pa fwd':>7s} {'lib':>7s} | {'pa dX':>7s} {'lib':>7s} | {'pa dW':>7s} {'lib':>7s}   (us)")
for (M, N, K, tag) in [(8704, 1536, 512, "enc in_proj"), (8704, 512, 512, "enc out_proj"), (8704, 1024, 512, "ffn1"),
                       (8704, 512, 1024, "ffn2"), (2048, 1536, 512, "dec in_proj"), (2048, 512, 512, "dec out_proj"),
                       (2048, 1024, 512, "dec ffn1"), (2048, 512, 1024, "dec ffn2"), (8704, 1024, 512, "cross kv"),
                       (16384, 1536, 512, "padded in_proj")]:
    x, w, dy = rnd(M, K), rnd(N, K), rnd(M, N)
    bias = torch.zeros(N, device="cuda")
    bb = bias.to(torch.bfloat16)
    a = t(lambda: ops.gemm(x, w, bias=bias))
    la = t(lambda: F.linear(x, w, bb))
    b = t(lambda: ops.gemm(dy, w, b_kcontig=False))
    lb = t(lambda: torch.matmul(dy, w))
    sk = max(1, min(16, 512 // ((N // 128) * (K // 128))))
    c = t(lambda: ops.gemm(dy, x, a_kcontig=False, b_kcontig=False, out_dtype=torch.float32, splitk=sk))
    lc = t(lambda: torch.matmul(dy.t(), x))
    print(f"{tag:16s} {M:6d} {N:5d} {K:5d} | {a*1e6:7.1f} {la*1e6:7.1f} | {b*1e6:7.1f} {lb*1e6:7.1f} | {c*1e6:7.1f} {lc*1e6:7.1f}")

B, S, D, H = 16, 1024, 512, 8
qkv = rnd(B, S, 3 * D)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
fl = 4.0 * S * S * D * B
for drop in (0.0, 0.2):
    ta = t(lambda: ops.attn_fwd(q, k, v, H, drop_p=drop, drop_seed=1), iters=20)
    print(f"pa   attn fwd drop {drop}: {ta*1e6:7.1f} us {fl/ta/1e12:6.0f} TF")
qh = q.reshape(B, S, H, 64).transpose(1, 2).contiguous().requires_grad_(True)
kh = k.reshape(B, S, H, 64).transpose(1, 2).contiguous().requires_grad_(True)
vh = v.reshape(B, S, H, 64).transpose(1, 2).contiguous().requires_grad_(True)
for drop in (0.0, 0.2):
    try:
        with torch.no_grad():
            tf = t(lambda: F.scaled_dot_product_attention(qh, kh, vh, dropout_p=drop), iters=20)
        o = F.scaled_dot_product_attention(qh, kh, vh, dropout_p=drop)
        do = torch.randn_like(o)
        tb = t(lambda: torch.autograd.grad(o, (qh, kh, vh), do, retain_graph=True), iters=10)
        print(f"sdpa attn drop {drop}: fwd {tf*1e6:7.1f} us {fl/tf/1e12:6.0f} TF   bwd {tb*1e6:7.1f} us {2.5*fl/tb/1e12:6.0f} TF")
    except Exception as e:                                      # noqa: BLE001
        print("sdpa failed:", repr(e)[:200])
o, lse = ops.attn_fwd(q, k, v, H, drop_p=0.2, drop_seed=1)
do = rnd(B, S, D)
tb = t(lambda: ops.attn_bwd(do, q, k, v, o, lse, H, drop_p=0.2, drop_seed=1), iters=10)
print(f"pa   attn bwd drop 0.2: {tb*1e6:7.1f} us {2.5*fl/tb/1e12:6.0f} TF")
