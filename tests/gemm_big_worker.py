"""Worker of tests/test_kernels_gpu.py::test_gemm_eight_wave_big_tiles: runs in its own process with PA_GEMM_BIG=1 (pa_gemm reads the
switch once per process) and checks the opt-in eight-wave kernel (plankassembly_amd/csrc/gemm8.h) against torch fp32, including that
the launches really went to it (pa_gemm_record / pa_gemm_recorded_kinds)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from plankassembly_amd import _lib as L, ops

KIND_BIG = 5
assert os.environ.get("PA_GEMM_BIG") == "1"


def rnd(*shape, seed=0, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def rel_err(got, ref):
    got, ref = got.detach().float().cpu().double(), ref.detach().float().cpu().double()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


lib = L.lib()
lib.pa_gemm_record(1)
n_launch = 0
# (M, N, K, epilogue): 256 x 256 tiles (N >= 1024) and 256 x 128 tiles; ragged last row / column tiles; several units per block
# (4100 x 2100: 17 x 9 tiles on 256 blocks is one round, 9000 x 3072: 36 x 12 = 432 tiles are two); every epilogue stage
for i, (M, N, K, epi) in enumerate([(8704, 1536, 512, "plain"), (7940, 512, 512, "res_drop"), (4100, 2100, 640, "relu_drop"),
                                    (9000, 3072, 128, "plain"), (4099, 520, 1024, "gate"), (8704, 512, 1536, "f32out")]):
    a, b = rnd(M, K, seed=100 + i, scale=0.3), rnd(N, K, seed=200 + i, scale=0.3)
    bias = rnd(N, seed=300 + i, dtype=torch.float32)
    acc = a.float() @ b.float().t() + bias
    kw = dict(bias=bias.cuda())
    if epi == "res_drop":
        res = rnd(M, N, seed=400 + i)
        out = ops.gemm(a.cuda(), b.cuda(), residual=res.cuda(), drop_p=0.2, drop_seed=7, **kw)
        keep = (out.float().cpu() - res.float()).abs() > 0                       # dropped entries are exactly the residual
        ref = torch.where(keep, acc / 0.8, torch.zeros_like(acc)) + res.float()
        assert 0.78 < float(keep.float().mean()) < 0.82, float(keep.float().mean())
    elif epi == "relu_drop":
        out = ops.gemm(a.cuda(), b.cuda(), relu=True, drop_p=0.2, drop_seed=9, **kw)
        pos = acc > 0
        keep = out.float().cpu() != 0
        ref = torch.where(keep, acc.clamp_min(0) / 0.8, torch.zeros_like(acc))
        assert 0.77 < float(keep[pos].float().mean()) < 0.83
    elif epi == "gate":
        gate = rnd(M, N, seed=500 + i)
        out = ops.gemm(a.cuda(), b.cuda(), aux=gate.cuda(), aux_scale=1.25, **kw)
        ref = torch.where(gate.float() > 0, acc * 1.25, torch.zeros_like(acc))
    elif epi == "f32out":
        out = ops.gemm(a.cuda(), b.cuda(), out_dtype=torch.float32, **kw)
        ref = acc
    else:
        out = ops.gemm(a.cuda(), b.cuda(), **kw)
        ref = acc
    e = rel_err(out, ref)
    assert e < 2.5e-2, (M, N, K, epi, e)
    n_launch += 1
# batched (the cross-attention K | V launch: members side by side in C, one bias per member)
Bm, M, N, K = 3, 4200, 1024, 512
a, w = rnd(M, K, seed=31, scale=0.3), rnd(Bm, N, K, seed=32, scale=0.3)
bias = rnd(Bm, N, seed=33, dtype=torch.float32)
out = torch.empty(M, Bm * N, dtype=torch.bfloat16, device="cuda")
o3 = out.view(M, Bm, N).permute(1, 0, 2)                                           # [Bm, M, N] view: member stride N, row stride Bm * N
ops.gemm(a.cuda()[None].expand(Bm, M, K), w.cuda(), bias=bias.cuda(), out=o3)
for j in range(Bm):
    ref = a.float() @ w[j].float().t() + bias[j]
    assert rel_err(out[:, j * N:(j + 1) * N], ref) < 2.5e-2
n_launch += 1
n = int(lib.pa_gemm_record(0))
kinds = (C.c_int32 * max(n, 1))()
nk = int(lib.pa_gemm_recorded_kinds(C.cast(kinds, C.c_void_p), n))
assert nk == n_launch and all(kinds[i] == KIND_BIG for i in range(nk)), [kinds[i] for i in range(nk)]
print(f"gemm8 ok: {nk} launches on the eight-wave kernel")
