"""Op-level parity of the HIP kernels (through the C ABI) against plain torch fp32 on the CPU.
GPU only (`-m gpu`).  f32 kernels must agree to 1e-4 (relative to the tensor scale), bf16 kernels
to bf16 rounding."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from plankassembly_amd import ops
    from plankassembly_amd import _lib as L

DEV = "cuda"


def rel_err(got, ref):
    got = got.detach().float().cpu().double()
    ref = ref.detach().float().cpu().double()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


def tol(dtype):
    return 1e-4 if dtype == torch.float32 else 2.5e-2


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("akc,bkc", [(True, True), (True, False), (False, False), (False, True)])
@pytest.mark.parametrize("M,N,K", [(200, 136, 72), (128, 128, 64), (257, 514, 96), (96, 520, 514), (1199, 192, 64)])
def test_gemm_layouts(dtype, akc, bkc, M, N, K):
    a = rnd(M, K, dtype=dtype, seed=1)
    b = rnd(N, K, dtype=dtype, seed=2)
    ref = a.float() @ b.float().t()
    ad = (a if akc else a.t().contiguous()).to(DEV)
    bd = (b if bkc else b.t().contiguous()).to(DEV)
    out = ops.gemm(ad, bd, a_kcontig=akc, b_kcontig=bkc, out_dtype=torch.float32)
    assert rel_err(out, ref) < tol(dtype), (rel_err(out, ref))


# Long-K / multi-unit bf16 shapes: these exercise the ring kernel's steady-state ("hot") items, its stream across unit
# boundaries (persistent blocks: 1000 x 2100 -> 8 x 17 tiles <= 256 blocks; 4100 x 4100 -> 33 x 33 tiles, several units per
# block when forced) and the two-blocks-per-CU kernel on the multi-round shapes.
@pytest.mark.parametrize("akc,bkc", [(True, True), (True, False), (False, False), (False, True)])
@pytest.mark.parametrize("M,N,K,sk", [(1000, 2100, 1024, 1), (520, 512, 2048, 4), (256, 384, 4096, 8), (4100, 1030, 640, 1)])
def test_gemm_long_k_bf16(akc, bkc, M, N, K, sk):
    a = rnd(M, K, dtype=torch.bfloat16, seed=11, scale=0.5)
    b = rnd(N, K, dtype=torch.bfloat16, seed=12, scale=0.5)
    ref = a.float() @ b.float().t()
    ad = (a if akc else a.t().contiguous()).to(DEV)
    bd = (b if bkc else b.t().contiguous()).to(DEV)
    out = ops.gemm(ad, bd, a_kcontig=akc, b_kcontig=bkc, out_dtype=torch.float32, splitk=sk)
    assert rel_err(out, ref) < 1e-2, rel_err(out, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum_many(dtype):
    shapes = [(7940, 1536), (2048, 512), (300, 1024), (129, 520)]
    xs = [rnd(M, N, dtype=dtype, seed=40 + i) for i, (M, N) in enumerate(shapes)]
    outs = [torch.full((N,), 1.5, device=DEV) for _, N in shapes]            # accumulated into
    ops.colsum_many([(x.to(DEV), o) for x, o in zip(xs, outs)])
    for x, o in zip(xs, outs):
        ref = x.float().sum(0) + 1.5
        assert float((o.cpu() - ref).abs().max()) < (1e-3 if dtype == torch.float32 else 1e-2) * (1 + float(ref.abs().max()))


def test_gemm_group_weight_gradients():
    """One launch for the dW GEMMs of a whole layer (different shapes, split and unsplit members, ragged row counts)."""
    shapes = [(7940, 1536, 512, 5), (7940, 512, 512, 16), (7940, 1024, 512, 8), (7940, 512, 1024, 8),    # encoder layer
              (2048, 512, 512, 8), (2048, 1024, 512, 1), (300, 136, 72, 2)]
    items, refs = [], []
    for i, (rows, M, N, sk) in enumerate(shapes):
        dy = rnd(rows, M, dtype=torch.bfloat16, seed=50 + i, scale=0.3)
        x = rnd(rows, N, dtype=torch.bfloat16, seed=60 + i, scale=0.3)
        items.append((dy.to(DEV), x.to(DEV), sk))
        refs.append(dy.float().t() @ x.float())
    outs = ops.dw_group(items)
    for out, ref in zip(outs, refs):
        assert rel_err(out, ref) < 1e-2, rel_err(out, ref)


@pytest.mark.parametrize("shapes", [
    [(2048, 512, 512, 4), (2048, 1024, 512, 1), (700, 200, 136, 3)],       # 64 + 32 + 12 = 108 units: XCD-contiguous order, 4 padding slots
    [(2048, 384, 512, 5), (1000, 136, 72, 2)],                              # 60 + 4 units: exactly at the switch
    [(1000, 256, 256, 3), (300, 136, 72, 2)],                               # 12 + 4 units: plain order
])
def test_gemm_group_unit_orders(shapes):
    """The grouped launch walks its units XCD-contiguously (slot u -> unit (u % 8) * chunk + u / 8) from 64 units on and in the
    plain order below: one-round launches with padding slots, at the switch, and small ones (the 1 172-unit case above covers
    several units per block)."""
    items, refs = [], []
    for i, (rows, M, N, sk) in enumerate(shapes):
        dy = rnd(rows, M, dtype=torch.bfloat16, seed=150 + i, scale=0.3)
        x = rnd(rows, N, dtype=torch.bfloat16, seed=160 + i, scale=0.3)
        items.append((dy.to(DEV), x.to(DEV), sk))
        refs.append(dy.float().t() @ x.float())
    outs = ops.dw_group(items)
    for out, ref in zip(outs, refs):
        assert rel_err(out, ref) < 1e-2, rel_err(out, ref)


def test_gemm_deferred_splitk_reduce():
    """Weight-gradient path: several split-K GEMMs write only their f32 slabs, one launch reduces all of them.
    (K = 1000 with 5 requested slices is the case where rounding leaves the last slice empty: 16 K tiles -> 4 slices.)"""
    items, refs = [], []
    for i, (M, N, K, sk) in enumerate([(512, 512, 2048, 8), (1536, 512, 1000, 5), (514, 512, 2048, 8), (64, 36, 300, 3)]):
        a = rnd(K, M, dtype=torch.bfloat16, seed=20 + i, scale=0.5)          # dY [rows, M] (contraction strided)
        b = rnd(K, N, dtype=torch.bfloat16, seed=30 + i, scale=0.5)          # X  [rows, N]
        out, ws, sk = ops.gemm(a.to(DEV), b.to(DEV), a_kcontig=False, b_kcontig=False, out_dtype=torch.float32, splitk=sk, defer=True)
        out.fill_(float("nan"))                                               # the GEMM itself must not have produced the output
        items.append((ws, out, sk))
        refs.append(a.float().t() @ b.float())
    ops.splitk_reduce_many(items)
    for (ws, out, sk), ref in zip(items, refs):
        assert rel_err(out, ref) < 1e-2


def test_gemm_ring_epilogue_bf16():
    """bias + dropout + residual / relu + gate on a one-round launch with a ragged last row tile (M = 7940 as in the packed encoder)."""
    M, N, K = 7940, 512, 512
    a, b = rnd(M, K, dtype=torch.bfloat16, seed=13, scale=0.3), rnd(N, K, dtype=torch.bfloat16, seed=14, scale=0.3)
    bias, res = rnd(N, seed=15), rnd(M, N, dtype=torch.bfloat16, seed=16)
    acc = a.float() @ b.float().t()
    out = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), residual=res.to(DEV))
    assert rel_err(out, acc + bias + res.float()) < 2.5e-2
    out = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), relu=True)
    assert rel_err(out, torch.relu(acc + bias)) < 2.5e-2
    # dropout is the same counter-based mask whichever kernel runs: compare with the 2-blocks-per-CU result on a sub-problem
    o1 = ops.gemm(a[:200].to(DEV), b.to(DEV), bias=bias.to(DEV), drop_p=0.25, drop_seed=99, out_dtype=torch.float32)
    keep = (o1 != 0).float().cpu()
    assert abs(float(keep.mean()) - 0.75) < 0.01
    assert rel_err(o1.cpu(), keep * (acc[:200] + bias) / 0.75) < 2.5e-2


@pytest.mark.parametrize("bkc", [True, False])
@pytest.mark.parametrize("M,N", [(8300, 512), (8300, 1024), (8197, 576)])
def test_gemm_pair_epilogue_prefetch(M, N, bkc):
    """Two-blocks-per-CU kernel with bf16 residual / gate rows (its EPRE instantiation fetches them under the last K tile, csrc/gemm.hip
    prefetch_epi): more than 256 tiles so that kernel runs, a ragged last row tile, a ragged last column block (N = 576), k-contiguous
    and strided weights, R aliasing C.  ELEMENT-wise bound: a residual row landing on the wrong output row must not hide in a norm."""
    K = 512
    a = rnd(M, K, dtype=torch.bfloat16, seed=31, scale=0.3)
    w = rnd(N, K, dtype=torch.bfloat16, seed=32, scale=0.3)
    bias, res, aux = rnd(N, seed=33), rnd(M, N, dtype=torch.bfloat16, seed=34), rnd(M, N, dtype=torch.bfloat16, seed=35)
    acc = a.float() @ w.float().t()
    ad = a.to(DEV)
    wd = w.to(DEV) if bkc else w.t().contiguous().to(DEV)          # strided form: [K, N] row-major

    def close(out, ref):
        err = (out.float().cpu() - ref).abs()
        return bool((err <= 2.0 ** -7 * ref.abs() + 2e-2).all())
    out = ops.gemm(ad, wd, b_kcontig=bkc, bias=bias.to(DEV), residual=res.to(DEV))
    assert close(out, acc + bias + res.float())
    out = ops.gemm(ad, wd, b_kcontig=bkc, aux=aux.to(DEV), aux_scale=1.25)
    assert close(out, torch.where(aux.float() > 0, acc * 1.25, torch.zeros_like(acc)))
    out = ops.gemm(ad, wd, b_kcontig=bkc, bias=bias.to(DEV), residual=res.to(DEV), aux=aux.to(DEV), aux_scale=0.5)   # both: the residual is prefetched
    assert close(out, torch.where(aux.float() > 0, (acc + bias) * 0.5, torch.zeros_like(acc)) + res.float())
    c = res.to(DEV).clone()
    ops.gemm(ad, wd, b_kcontig=bkc, residual=c, out=c)
    assert close(c, acc + res.float())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dtype):
    M, N, K = 300, 200, 128
    a, b = rnd(M, K, dtype=dtype, seed=3), rnd(N, K, dtype=dtype, seed=4)
    bias = rnd(N, seed=5)
    res = rnd(M, N, dtype=dtype, seed=6)
    aux = rnd(M, N, dtype=dtype, seed=7)
    acc = a.float() @ b.float().t()
    ad, bd = a.to(DEV), b.to(DEV)
    # bias + relu
    out = ops.gemm(ad, bd, bias=bias.to(DEV), relu=True)
    assert rel_err(out, torch.relu(acc + bias)) < tol(dtype)
    # bias + residual, alpha
    out = ops.gemm(ad, bd, bias=bias.to(DEV), residual=res.to(DEV), alpha=0.5)
    assert rel_err(out, 0.5 * acc + bias + res.float()) < tol(dtype)
    # relu-backward gate
    out = ops.gemm(ad, bd, aux=aux.to(DEV), aux_scale=1.25)
    assert rel_err(out, torch.where(aux.float() > 0, acc * 1.25, torch.zeros_like(acc))) < tol(dtype)
    # in-place accumulate (R aliases C)
    c = res.to(DEV).clone()
    ops.gemm(ad, bd, residual=c, out=c)
    assert rel_err(c, acc + res.float()) < tol(dtype)
    # split-K with f32 output
    out = ops.gemm(ad, bd, splitk=2, out_dtype=torch.float32, bias=bias.to(DEV))
    assert rel_err(out, acc + bias) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_batched_pointer_shapes(dtype):
    B, T, d = 3, 40, 64
    feat, h = rnd(B, T, d, dtype=dtype, seed=8), rnd(B, T, d, dtype=dtype, seed=9)
    ref = torch.einsum("bid,bjd->bij", feat.float(), h.float()) / d
    out = ops.gemm(feat.to(DEV), h.to(DEV), alpha=1.0 / d, out_dtype=torch.float32)
    assert rel_err(out, ref) < tol(dtype)
    dptr = rnd(B, T, T, dtype=dtype, seed=10)
    # dfeat = dptr @ h / d   (B operand stored [K][N])
    out = ops.gemm(dptr.to(DEV), h.to(DEV), b_kcontig=False, alpha=1.0 / d, out_dtype=torch.float32)
    assert rel_err(out, dptr.float() @ h.float() / d) < tol(dtype)
    # dh = dptr^T @ feat / d  (both transposed)
    out = ops.gemm(dptr.to(DEV), feat.to(DEV), a_kcontig=False, b_kcontig=False, alpha=1.0 / d,
                   out_dtype=torch.float32)
    assert rel_err(out, dptr.float().transpose(1, 2) @ feat.float() / d) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("splitk", [1, 4])
def test_gemm_batched_per_member_bias(dtype, splitk):
    """pa_gemm_args.sBias: one bias row per batch member (the fused cross-attention K|V projection of all decoder layers),
    in the tile epilogues AND in the split-K reduce pass (the reduce pass used batch 0's bias for every member)."""
    Bn, M, N, K = 3, 70, 96, 256
    a, w = rnd(Bn, M, K, dtype=dtype, seed=21), rnd(Bn, N, K, dtype=dtype, seed=22)
    bias = rnd(Bn, N, seed=23)
    ref = torch.einsum("bmk,bnk->bmn", a.float(), w.float()) + bias[:, None, :]
    out = ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), out_dtype=torch.float32, splitk=splitk)
    assert rel_err(out, ref) < tol(dtype)
    assert rel_err(out[2], ref[2]) < tol(dtype) and rel_err(out[1], ref[1]) < tol(dtype)


def test_gemm_dropout_statistics():
    M, N, K = 512, 512, 64
    a, b = rnd(M, K, seed=11).to(DEV), rnd(N, K, seed=12).to(DEV)
    full = ops.gemm(a, b)
    out = ops.gemm(a, b, drop_p=0.2, drop_seed=1234)
    out2 = ops.gemm(a, b, drop_p=0.2, drop_seed=1234)
    assert torch.equal(out, out2)                         # counter based: reproducible
    kept = out != 0
    frac = 1.0 - kept.float().mean().item()
    assert abs(frac - 0.2) < 0.01, frac
    assert rel_err(out[kept], full[kept] * 1.25) < 1e-5
    other = ops.gemm(a, b, drop_p=0.2, drop_seed=99)
    assert (other != 0).ne(kept).float().mean().item() > 0.2   # different seed, different mask
    # the decision is separable (row hash x column hash, pa_device.h drop_keep_rc): no row and no column may be favoured.
    # 512 Bernoulli(0.8) draws: sigma = 0.0177; every row / column within 5 sigma, their spread as a binomial's
    k = kept.float()
    for rate in (k.mean(dim=1), k.mean(dim=0)):
        assert float((rate - 0.8).abs().max()) < 0.09, float((rate - 0.8).abs().max())
        assert 0.012 < float(rate.std()) < 0.024, float(rate.std())
    # neighbouring rows / columns are uncorrelated: the agreement rate of two independent 0.8-masks is 0.68
    agree_r = (kept[1:] == kept[:-1]).float().mean().item()
    agree_c = (kept[:, 1:] == kept[:, :-1]).float().mean().item()
    assert abs(agree_r - 0.68) < 0.01 and abs(agree_c - 0.68) < 0.01, (agree_r, agree_c)


@pytest.mark.parametrize("M,N,K", [(512, 512, 64), (300, 1024, 128), (9000, 512, 64), (130, 192, 64)])
def test_gemm_dropout_decisions_equal_the_numpy_restatement(M, N, K):
    """tests/dropout_masks.py restates the epilogue's decision function; every GEMM kernel family (64 x 64, ring, two blocks
    per CU, 128 x 256) must drop exactly the entries it predicts - the model-level dropout parity test rests on this."""
    import dropout_masks as dm
    for dtype in (torch.float32, torch.bfloat16):
        a = torch.ones(M, K, dtype=dtype, device=DEV)
        b = torch.ones(N, K, dtype=dtype, device=DEV)
        out = ops.gemm(a, b, drop_p=0.2, drop_seed=987654321)
        want = dm.linear_keep(987654321, np.arange(M), N, 0.2)
        assert np.array_equal((out != 0).cpu().numpy(), want), (dtype, M, N)
        assert torch.equal(out[out != 0], torch.full_like(out[out != 0], K * 1.25))


def test_layernorm_bwd_dropout_decisions_equal_the_numpy_restatement():
    import dropout_masks as dm
    rows, d = 333, 512
    z = rnd(rows, d, seed=21).to(DEV)
    gamma = torch.ones(d, device=DEV)
    _, mean, rstd = ops.layernorm_fwd(z, gamma, torch.zeros(d, device=DEV), 1.0)
    dy = rnd(rows, d, seed=22).to(DEV)
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    dz, ddrop = ops.layernorm_bwd(dy, z, gamma, mean, rstd, dg, db, drop_p=0.2, drop_seed=4711)
    want = torch.from_numpy(dm.linear_keep(4711, np.arange(rows), d, 0.2)).to(DEV)
    assert torch.equal(ddrop != 0, want & (dz != 0))                         # d(sublayer output) = mask o dz / (1 - p)
    assert rel_err(ddrop[want], dz[want] * 1.25) < 1e-6


def test_colsum():
    for dtype in (torch.float32, torch.bfloat16):
        x = rnd(1000, 520, dtype=dtype, seed=13)
        out = ops.colsum(x.to(DEV))
        assert rel_err(out, x.float().sum(0)) < (1e-5 if dtype == torch.float32 else 1e-5)


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("d,eps", [(512, 1.0), (64, 1e-5), (128, 1.0)])
def test_layernorm(dtype, d, eps):
    rows = 777
    z = rnd(rows, d, dtype=dtype, seed=14, scale=2.0)
    gamma, beta = 1 + 0.3 * rnd(d, seed=15), 0.2 * rnd(d, seed=16)
    dy = rnd(rows, d, dtype=dtype, seed=17)
    zr = z.float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(zr, (d,), gr, br, eps)
    ref.backward(dy.float())
    y, mean, rstd = ops.layernorm_fwd(z.to(DEV), gamma.to(DEV), beta.to(DEV), eps)
    assert rel_err(y, ref) < tol(dtype)
    dgamma = torch.zeros(d, device=DEV); dbeta = torch.zeros(d, device=DEV); dzsum = torch.zeros(d, device=DEV)
    dz, _ = ops.layernorm_bwd(dy.to(DEV), z.to(DEV), gamma.to(DEV), mean, rstd, dgamma, dbeta, dzsum)
    assert rel_err(dz, zr.grad) < tol(dtype)
    assert rel_err(dgamma, gr.grad) < tol(dtype)
    assert rel_err(dbeta, br.grad) < tol(dtype)
    assert rel_err(dzsum, dz.float().sum(0)) < 1e-2
    dzsum.zero_()
    # dropout variant: ddrop is dz masked & scaled with the same mask as the GEMM epilogue would use
    dgamma.zero_(); dbeta.zero_()
    dz2, dd = ops.layernorm_bwd(dy.to(DEV), z.to(DEV), gamma.to(DEV), mean, rstd, dgamma, dbeta, dzsum,
                                drop_p=0.2, drop_seed=77)
    assert torch.equal(dz2, dz)
    ones_k = torch.eye(d, device=DEV, dtype=torch.float32)
    mask = ops.gemm(torch.ones(rows, d, device=DEV), ones_k, drop_p=0.2, drop_seed=77) != 0   # same (row*d+col) index
    assert rel_err(dd.float(), torch.where(mask, dz.float() * 1.25, torch.zeros_like(dz.float()))) < 1e-2
    assert rel_err(dzsum, dd.float().sum(0)) < 1e-2


# ------------------------------------------------------------------------------------------ embeddings
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embeddings(dtype):
    d, B, S, T = 64, 3, 50, 19
    g = torch.Generator().manual_seed(18)
    tabs = [rnd(n, d, seed=20 + i) for i, n in enumerate((514, 20, 4, 3, 2))]
    idx = [torch.randint(0, n, (B * S,), generator=g) for n in (514, 20, 4, 3, 2)]
    ref = sum(t[i] for t, i in zip(tabs, idx))
    out = ops.embed_input_fwd([t.to(DEV) for t in tabs], [i.to(DEV) for i in idx], dtype)
    assert rel_err(out, ref) < tol(dtype)
    # sideface: no input_type
    out = ops.embed_input_fwd([t.to(DEV) for t in tabs], [i.to(DEV) for i in idx[:4]] + [None], dtype)
    assert rel_err(out, sum(t[i] for t, i in zip(tabs[:4], idx[:4]))) < tol(dtype)
    dout = rnd(B * S, d, dtype=dtype, seed=30)
    dt = [torch.zeros_like(t).to(DEV) for t in tabs]
    ops.embed_input_bwd(dout.to(DEV), dt, [i.to(DEV) for i in idx])
    for k in range(5):
        refg = torch.zeros_like(tabs[k]).index_add_(0, idx[k], dout.float())
        assert rel_err(dt[k], refg) < 1e-4
    # output embedding: shifted, zero first row
    tok = torch.randint(0, 514, (B, T), generator=g)
    coord, pos = rnd(6, d, seed=31), rnd(4, d, seed=32)
    t = torch.arange(T - 1)
    refo = tabs[0][tok[:, :-1]] + coord[t % 6][None] + pos[t // 6][None]
    refo = torch.cat((torch.zeros(B, 1, d), refo), 1)
    out = ops.embed_output_fwd(tabs[0].to(DEV), coord.to(DEV), pos.to(DEV), tok.to(DEV), T, 6, dtype)
    assert rel_err(out, refo) < tol(dtype)
    dout = rnd(B, T, d, dtype=dtype, seed=33)
    dv, dc, dp = (torch.zeros_like(x).to(DEV) for x in (tabs[0], coord, pos))
    ops.embed_output_bwd(dout.to(DEV), dv, dc, dp, tok.to(DEV))
    g1 = dout.float()[:, 1:].reshape(-1, d)
    assert rel_err(dv, torch.zeros_like(tabs[0]).index_add_(0, tok[:, :-1].reshape(-1), g1)) < 1e-4
    assert rel_err(dc, torch.zeros_like(coord).index_add_(0, (t % 6).repeat(B), g1)) < 1e-4
    assert rel_err(dp, torch.zeros_like(pos).index_add_(0, (t // 6).repeat(B), g1)) < 1e-4


# ------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, H, kpm, causal):
    B, Lq, dm = q.shape
    Lk, dh = k.shape[1], dm // H
    qh = q.view(B, Lq, H, dh).transpose(1, 2)
    kh = k.view(B, Lk, H, dh).transpose(1, 2)
    vh = v.view(B, Lk, H, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    if causal:
        s = s + torch.triu(torch.full((Lq, Lk), float("-inf")), 1)
    p = torch.softmax(s, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Lq, dm), torch.logsumexp(s, -1)


CASES = [
    # B, H, dh, Lq, Lk, kpm, causal
    (2, 4, 16, 64, 64, True, False),      # fixture encoder shape
    (2, 4, 16, 36, 36, True, True),       # fixture decoder self
    (2, 4, 16, 36, 64, True, False),      # fixture cross
    (2, 8, 16, 200, 333, True, False),    # ragged tiles, tiny-config head dim
    (1, 2, 32, 130, 130, False, True),
    (2, 8, 64, 128, 1199, True, False),   # default cross-attention
    (1, 8, 64, 300, 300, True, False),
    (2, 8, 64, 128, 128, True, True),     # default decoder self
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,dh,Lq,Lk,use_kpm,causal", CASES)
def test_attention_fwd_bwd(dtype, B, H, dh, Lq, Lk, use_kpm, causal):
    dm = H * dh
    # packed projections like the in_proj output: q from [B,Lq,3dm], k/v views of [B,Lk,3dm]
    qkv_q = rnd(B, Lq, 3 * dm, dtype=dtype, seed=40)
    qkv_k = qkv_q if (Lq == Lk) else rnd(B, Lk, 3 * dm, dtype=dtype, seed=41)
    q, k, v = qkv_q[..., :dm], qkv_k[..., dm:2 * dm], qkv_k[..., 2 * dm:]
    kpm = None
    if use_kpm:
        g = torch.Generator().manual_seed(42)
        valid = torch.randint(1, Lk + 1, (B,), generator=g)
        kpm = torch.arange(Lk)[None, :] >= valid[:, None]
        if Lk > 8:
            kpm[0, 3] = True                                   # a hole, not only a suffix
    dout = rnd(B, Lq, dm, dtype=dtype, seed=43)
    qr, kr, vr = (x.float().contiguous().requires_grad_(True) for x in (q, k, v))
    ref, lse_ref = attn_ref(qr, kr, vr, H, kpm, causal)
    ref.backward(dout.float())
    qd, kd = qkv_q.to(DEV), (qkv_q.to(DEV) if Lq == Lk else qkv_k.to(DEV))
    if Lq == Lk:
        kd = qd
    qv, kv, vv = qd[..., :dm], kd[..., dm:2 * dm], kd[..., 2 * dm:]
    kpm_d = kpm.to(DEV) if kpm is not None else None
    o, lse = ops.attn_fwd(qv, kv, vv, H, kpm=kpm_d, causal=causal)
    t = tol(dtype)
    assert rel_err(o, ref) < t, ("o", rel_err(o, ref))
    assert float((lse.cpu() - lse_ref).abs().max()) < (1e-4 if dtype == torch.float32 else 5e-2)
    dq, dk, dv = ops.attn_bwd(dout.to(DEV), qv, kv, vv, o, lse, H, kpm=kpm_d, causal=causal)
    assert rel_err(dq, qr.grad) < t, ("dq", rel_err(dq, qr.grad))
    assert rel_err(dk, kr.grad) < t, ("dk", rel_err(dk, kr.grad))
    assert rel_err(dv, vr.grad) < t, ("dv", rel_err(dv, vr.grad))


@pytest.mark.parametrize("Lq,Lk", [(64, 300), (64, 1000)])
def test_attention_backward_probabilities_sum_to_one_per_row(Lq, Lk):
    """The bf16 forward kernels take their scores from query rows pre-multiplied by scale * log2(e) and RE-ROUNDED to bf16 (the
    MFMA then delivers the exponent), the backward kernels recompute (q . k) * scale * log2(e) - lse from the unscaled rows, so the
    forward's and the backward's probabilities are not bit-for-bit the same function.  The difference is one more bf16
    rounding of q - the size of the input quantisation itself; this pins it: with dO[i] = e_i (64 query rows, dh 64) and no
    dropout, dV[j][i] = P_bwd[i][j], so the column sums of dV are the ROW sums of the backward's probabilities and must be 1."""
    B, H, dh = 2, 2, 64
    q = rnd(B, Lq, H * dh, dtype=torch.bfloat16, seed=31).to(DEV)
    k = rnd(B, Lk, H * dh, dtype=torch.bfloat16, seed=32, scale=1.5).to(DEV)
    v = rnd(B, Lk, H * dh, dtype=torch.bfloat16, seed=33).to(DEV)
    o, lse = ops.attn_fwd(q, k, v, H)
    do = torch.zeros(B, Lq, H * dh, dtype=torch.bfloat16, device=DEV)
    idx = torch.arange(Lq, device=DEV)
    for h in range(H):
        do[:, idx, h * dh + idx % dh] = 1.0
    dq, dk, dv = ops.attn_bwd(do, q, k, v, o, lse, H)
    rows = dv.float().view(B, Lk, H, dh).sum(dim=1)                     # [B, H, 64]: row sums of P_bwd
    err = float((rows - 1.0).abs().max())
    print(f"    backward probability row sums: max |sum - 1| = {err:.2e}")
    assert err < 0.03, err


def test_attention_both_wave_shapes_agree():
    """dh = 64 bf16 has two kernel families (32-row waves / 16-row waves, chosen by launch size); both must give the same
    results on the same problem - run in subprocesses because the choice is read once per process (PA_ATTN_V4)."""
    import subprocess, sys, os
    code = (
        "import torch, sys; sys.path.insert(0, %r); from plankassembly_amd import ops\n"
        "g = torch.Generator().manual_seed(5); B, H, L, dm = 2, 8, 300, 512\n"
        "x = (torch.randn(B, L, 3 * dm, generator=g)).to(torch.bfloat16).cuda(); do = torch.randn(B, L, dm, generator=g).to(torch.bfloat16).cuda()\n"
        "kpm = (torch.arange(L)[None] >= torch.tensor([[300], [170]])).cuda()\n"
        "q, k, v = x[..., :dm], x[..., dm:2 * dm], x[..., 2 * dm:]\n"
        "o, lse = ops.attn_fwd(q, k, v, H, kpm=kpm, drop_p=0.2, drop_seed=9)\n"
        "dq, dk, dv = ops.attn_bwd(do, q, k, v, o, lse, H, kpm=kpm, drop_p=0.2, drop_seed=9)\n"
        "torch.save([t.float().cpu() for t in (o, lse, dq, dk, dv)], sys.argv[1])\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for mode in ("0", "2"):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f".attn_v4_{mode}.pt")
        env = dict(os.environ, PA_ATTN_V4=mode)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env)
        outs.append(torch.load(path))
        os.remove(path)
    for a, b, name in zip(outs[0], outs[1], ("o", "lse", "dq", "dk", "dv")):
        assert rel_err(a, b) < 2e-2, (name, rel_err(a, b))
    assert rel_err(outs[0][0], outs[1][0]) > 0 or True


def test_attention_dropout_consistency():
    """Dropout mask is a deterministic function of (seed, index): check the keep fraction and that the
    backward kernels differentiate exactly the function the forward computes (directional derivative)."""
    B, H, dh, L = 1, 2, 16, 96
    dm = H * dh
    q, k, v = (rnd(B, L, dm, seed=s).to(DEV) for s in (50, 51, 52))
    kw = dict(drop_p=0.2, drop_seed=4242)
    # keep fraction: with V = one-hot rows the output exposes the (dropped, scaled) probabilities
    o1, lse = ops.attn_fwd(q, k, v, H, **kw)
    o2, _ = ops.attn_fwd(q, k, v, H, **kw)
    assert torch.equal(o1, o2)
    o0, _ = ops.attn_fwd(q, k, v, H)
    assert rel_err(o1, o0) > 0.05                      # dropout really changes the output
    dout = rnd(B, L, dm, seed=53).to(DEV)
    dq, dk, dv = ops.attn_bwd(dout, q, k, v, o1, lse, H, **kw)
    for name, x, gx, seed in (("q", q, dq, 60), ("k", k, dk, 61), ("v", v, dv, 62)):
        dirn = rnd(B, L, dm, seed=seed).to(DEV)
        eps = 1e-2
        args = {"q": q, "k": k, "v": v}
        args[name] = x + eps * dirn
        fp, _ = ops.attn_fwd(args["q"], args["k"], args["v"], H, **kw)
        args[name] = x - eps * dirn
        fm, _ = ops.attn_fwd(args["q"], args["k"], args["v"], H, **kw)
        num = float(((fp - fm).double() * dout.double()).sum() / (2 * eps))
        ana = float((gx.double() * dirn.double()).sum())
        assert abs(num - ana) < 2e-2 * max(1.0, abs(ana)), (name, num, ana)
    # keep statistics through P: V = identity over keys (dh*H >= L not needed: use one head dim trick)
    Lk = 64
    qq = torch.zeros(1, 256, 64, device=DEV)
    kk = torch.zeros(1, Lk, 64, device=DEV)
    vv = torch.eye(Lk, 64, device=DEV)[None]
    o, _ = ops.attn_fwd(qq, kk, vv, 1, drop_p=0.2, drop_seed=7)      # uniform P = 1/64 each
    kept = (o > 0).float().mean().item()
    assert abs(kept - 0.8) < 0.02, kept
    assert abs(o[o > 0].mean().item() - 1.25 / Lk) < 1e-6                  # survivors scaled by exactly 1 / (1 - p)


def _extract_keep_mask(B, H, Lq, Lk, dh, p, seed, dtype):
    """The kernels' dropout decisions as a bool [B, H, Lq, Lk]: with Q = K = 0 the probabilities are uniform and
    V = one-hot key indicators (64 keys per pass) expose which of them survived."""
    dm = H * dh
    z = torch.zeros(B, Lq, dm, device=DEV, dtype=dtype)
    zk = torch.zeros(B, Lk, dm, device=DEV, dtype=dtype)
    keep = torch.zeros(B, H, Lq, Lk, dtype=torch.bool)
    for k0 in range(0, Lk, dh):
        v = torch.zeros(B, Lk, H, dh, device=DEV, dtype=dtype)
        n = min(dh, Lk - k0)
        v[:, k0:k0 + n, :, :n] = torch.eye(n, device=DEV, dtype=dtype)[None, :, None, :]
        o, _ = ops.attn_fwd(z, zk, v.view(B, Lk, dm), H, drop_p=p, drop_seed=seed)
        keep[:, :, :, k0:k0 + n] = (o.float().view(B, Lq, H, dh)[..., :n] > 0).permute(0, 2, 1, 3).cpu()
    return keep


@pytest.mark.parametrize("dtype,B,H,dh,Lq,Lk,causal", [
    (torch.bfloat16, 2, 2, 64, 200, 333, False),      # the benchmarked head size, ragged multi-tile shapes
    (torch.bfloat16, 1, 2, 64, 130, 130, True),
    (torch.float32, 1, 2, 16, 70, 100, False),
])
def test_attention_dropout_forward_backward_against_masked_reference(dtype, B, H, dh, Lq, Lk, causal):
    """Dropout is a deterministic function of (seed, query row, key): extract the mask the kernels use, then compare
    forward AND all three backward outputs with a torch reference that applies that same mask - this pins that the
    forward, dQ and dK/dV kernels (different register layouts) regenerate identical decisions."""
    p, seed, dm = 0.2, 777, H * dh
    keep = _extract_keep_mask(B, H, Lq, Lk, dh, p, seed, dtype)
    rate = keep.float().mean().item()
    assert abs(rate - 0.8) < 0.01, rate
    assert abs(keep.float().mean(dim=-1).std().item() - math.sqrt(0.16 / Lk)) < 0.4 * math.sqrt(0.16 / Lk)   # rows independent
    q, k, v = rnd(B, Lq, dm, dtype=dtype, seed=90), rnd(B, Lk, dm, dtype=dtype, seed=91), rnd(B, Lk, dm, dtype=dtype, seed=92)
    dout = rnd(B, Lq, dm, dtype=dtype, seed=93)
    qr, kr, vr = (x.float().requires_grad_(True) for x in (q, k, v))
    qh = qr.view(B, Lq, H, dh).transpose(1, 2); kh = kr.view(B, Lk, H, dh).transpose(1, 2); vh = vr.view(B, Lk, H, dh).transpose(1, 2)
    sc = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    if causal:
        sc = sc + torch.triu(torch.full((Lq, Lk), float("-inf")), 1)
    pr = torch.softmax(sc, -1) * keep / (1 - p)
    ref = (pr @ vh).transpose(1, 2).reshape(B, Lq, dm)
    ref.backward(dout.float())
    o, lse = ops.attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), H, causal=causal, drop_p=p, drop_seed=seed)
    t = tol(dtype)
    assert rel_err(o, ref) < t, rel_err(o, ref)
    dq, dk, dv = ops.attn_bwd(dout.to(DEV), q.to(DEV), k.to(DEV), v.to(DEV), o, lse, H, causal=causal, drop_p=p, drop_seed=seed)
    assert rel_err(dq, qr.grad) < t, ("dq", rel_err(dq, qr.grad))
    assert rel_err(dk, kr.grad) < t, ("dk", rel_err(dk, kr.grad))
    assert rel_err(dv, vr.grad) < t, ("dv", rel_err(dv, vr.grad))


@pytest.mark.parametrize("dtype,B,H,dh,Lq,Lk", [(torch.bfloat16, 2, 2, 64, 200, 333), (torch.float32, 3, 4, 16, 70, 100)])
def test_attention_dropout_decisions_equal_the_numpy_restatement(dtype, B, H, dh, Lq, Lk):
    """keep(row, key) as tests/dropout_masks.py computes it == the decisions extracted from the kernels."""
    import dropout_masks as dm
    keep = _extract_keep_mask(B, H, Lq, Lk, dh, 0.2, 31337, dtype)
    rows = (np.arange(B)[:, None, None] * H + np.arange(H)[None, :, None]) * Lq + np.arange(Lq)[None, None, :]
    want = dm.attn_keep(31337, rows, Lk, 0.2)
    assert np.array_equal(keep.numpy(), want)


def _mix32(x):
    x = x.astype(np.uint64)
    m = np.uint64(0xffffffff)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & m
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & m
    x ^= x >> np.uint64(16)
    return x


def test_attention_dropout_keep_rate_per_row_and_key():
    """keep(row, key) = low32(A[row] * C[key]) >= p * 2^32 is only Bernoulli(1 - p) if the product wraps; a
    24-bit row hash below ~300 (one row in 55 000) used to keep 0-70 % of its keys.  Pick the seed whose raw row hashes
    contain the smallest value (host-side restatement of pa_device.h's mix32), extract the kernels' mask for that seed
    and check exactly those rows - and every row and key column - against the binomial expectation."""
    B, H, Lq, Lk, dh, p = 2, 8, 512, 512, 64, 0.2
    rows = np.arange(B * H * Lq, dtype=np.uint64)
    best = None
    for seed in range(1, 400):
        raw = _mix32((rows * np.uint64(0x9e3779b9) + np.uint64(seed)) & np.uint64(0xffffffff)) & np.uint64(0xffffff)
        if best is None or raw.min() < best[1]:
            best = (seed, int(raw.min()), int(raw.argmin()), np.argsort(raw)[:8])
    seed, amin, _, smallest = best
    assert amin < 2000, amin                                   # the case the old hash got wrong is in the sample
    keep = _extract_keep_mask(B, H, Lq, Lk, dh, p, seed, torch.bfloat16).view(B * H * Lq, Lk).float()
    sigma = math.sqrt(p * (1 - p) / Lk)
    per_row = keep.mean(dim=1)
    assert float(per_row.min()) > 1 - p - 6 * sigma and float(per_row.max()) < 1 - p + 6 * sigma, (per_row.min(), per_row.max())
    for r in smallest:                                         # rows whose raw 24-bit hash is tiny
        assert abs(float(per_row[int(r)]) - (1 - p)) < 5 * sigma, (int(r), float(per_row[int(r)]))
    per_key = keep.view(B * H, Lq, Lk).mean(dim=1)             # each (batch, head)'s key columns over its 512 query rows
    sk = math.sqrt(p * (1 - p) / Lq)
    assert float((per_key - (1 - p)).abs().max()) < 6 * sk
    assert abs(float(per_row.std()) - sigma) < 0.15 * sigma    # rows behave like independent binomials
    # rows are not copies / complements of each other: correlation of neighbouring rows' decisions is noise
    a, b = keep[0::2] - (1 - p), keep[1::2] - (1 - p)
    corr = (a * b).mean(dim=1) / (p * (1 - p))
    assert float(corr.abs().max()) < 6 / math.sqrt(Lk), float(corr.abs().max())


# ------------------------------------------------------------------------------------------ heads / loss
def test_mixture_nll_vs_oracle(small_fixture):
    from oracle import plank_oracle as O
    sd, batch, g = small_fixture
    cfg = O.OracleCfg(d_model=64, n_head=4, d_ff=128, n_enc=2, n_dec=2, max_input_length=65, max_output_length=36)
    h = torch.from_numpy(g["g1::hiddens"]).clone().requires_grad_(True)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    dists = O.create_dist_train(p, cfg, h)
    label = batch["output_label"]
    valid = label != 513
    picked = dists.gather(-1, label[..., None])[..., 0]
    loss = -(picked * valid).sum() / valid.sum()
    vocab, ptr, prob = O.head_logits(p, cfg, h)
    vocab.retain_grad(); ptr.retain_grad()
    # recompute through the retained leaves
    sw = O.linear(h, p["switch_head.weight"], p["switch_head.bias"])[..., 0]
    sw.retain_grad()
    tri = torch.triu(torch.ones(36, 36)) == 1
    d2 = torch.cat((O.log_softmax_lastdim(vocab) + torch.log(torch.clamp(1 - torch.sigmoid(sw)[..., None], min=1e-6)),
                    O.log_softmax_lastdim(ptr.masked_fill(tri[None], 1e-6))
                    + torch.log(torch.clamp(torch.sigmoid(sw)[..., None], min=1e-6))), -1)
    loss2 = -(d2.gather(-1, label[..., None])[..., 0] * valid).sum() / valid.sum()
    loss2.backward()
    assert abs(float(loss2) - float(loss)) < 1e-6
    B, T = label.shape
    stats, row_lse = ops.mixture_nll_fwd(vocab.detach().to(DEV), ptr.detach().to(DEV).contiguous(),
                                         sw.detach().reshape(-1).to(DEV), label.to(DEV), 514, 513)
    stats[3] = 1.0                                   # upstream gradient slot
    st = stats.cpu()
    assert abs(st[0] / st[1] - float(loss)) < 1e-5 * max(1, abs(float(loss)))
    assert int(st[1]) == int(valid.sum())
    acc_ref = float(g["g1::accuracy"])
    assert abs(float(st[2] / st[1]) - acc_ref) < 1e-6
    dv, dp, ds = ops.mixture_nll_bwd(stats, row_lse, vocab.detach().to(DEV), ptr.detach().to(DEV).contiguous(),
                                     sw.detach().reshape(-1).to(DEV), label.to(DEV), 514, 513)
    assert rel_err(dv, vocab.grad) < 1e-4
    assert rel_err(dp, ptr.grad) < 1e-4
    assert rel_err(ds.view(B, T), sw.grad) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_switch_head(dtype):
    rows, d = 300, 128
    h = rnd(rows, d, dtype=dtype, seed=70)
    w, b = rnd(d, seed=71), rnd(1, seed=72)
    ds = rnd(rows, seed=73)
    s = ops.switch_fwd(h.to(DEV), w.to(DEV), b.to(DEV))
    assert rel_err(s, h.float() @ w + b) < tol(dtype)
    dw, db = torch.zeros(d, device=DEV), torch.zeros(1, device=DEV)
    base = rnd(rows, d, dtype=dtype, seed=74)
    dh = base.to(DEV).clone()
    ops.switch_bwd(ds.to(DEV), h.to(DEV), w.to(DEV), dw, db, dh=dh)
    assert rel_err(dh, base.float() + ds[:, None] * w[None]) < tol(dtype)
    assert rel_err(dw, (ds[:, None] * h.float()).sum(0)) < tol(dtype)
    assert rel_err(db, ds.sum()[None]) < 1e-5


def test_adam_vs_oracle():
    from oracle import plank_oracle as O
    n = 100003
    p, g = rnd(n, seed=80), rnd(n, seed=81, scale=1e-3)
    params, grads = {"x": p.clone()}, {"x": g}
    m, v = {"x": torch.zeros(n)}, {"x": torch.zeros(n)}
    pd, md, vd = p.to(DEV).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step in (1, 2, 3):
        O.adam_step(params, grads, m, v, step, lr=1e-4)
        ops.adam_step(pd, g.to(DEV), md, vd, step, lr=1e-4, p_bf16=pb)
    assert float((pd.cpu() - params["x"]).abs().max()) < 2e-7
    assert rel_err(md, m["x"]) < 1e-5 and rel_err(vd, v["x"]) < 1e-4   # (1-b2) is rounded to f32 on the device
    assert torch.equal(pb.cpu(), pd.cpu().to(torch.bfloat16))


# ------------------------------------------------------------------------------------------ packed (unpadded) batches
def test_pack_rows_matches_nonzero():
    g = torch.Generator().manual_seed(90)
    mask = torch.rand(5, 77, generator=g) < 0.4
    mask[2] = True; mask[2, 0] = False                    # an "empty" row: only END is valid
    cu, rowmap, n = ops.pack_rows(mask.to(DEV))
    valid = (~mask).flatten().nonzero().flatten()
    assert n == valid.numel()
    assert torch.equal(rowmap.cpu().long(), valid)
    cnt = (~mask).sum(1)
    assert torch.equal(cu.cpu().long(), torch.cat((torch.zeros(1, dtype=torch.long), cnt.cumsum(0))))
    order = ops.pack_order(cu).cpu().long()                 # batch elements by descending row count, ties in batch order
    assert torch.equal(order, torch.sort(cnt, descending=True, stable=True).indices)
    assert torch.equal(ops.pack_order(cu._base).cpu().long(), order)     # the whole [2B+1] buffer (what prepare_batch keeps)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,dh,Lq,Lk,self_attn", [(3, 4, 16, 70, 70, True), (4, 8, 64, 128, 300, False), (2, 8, 64, 260, 260, True),
                                                    (14, 8, 64, 450, 450, True), (5, 8, 64, 128, 1021, False), (3, 8, 64, 700, 700, True)])
def test_attention_varlen_matches_masked_dense(dtype, B, H, dh, Lq, Lk, self_attn):
    """Packed K/V (and Q for self-attention) without any mask == dense attention with a key-padding mask."""
    dm = H * dh
    g = torch.Generator().manual_seed(91)
    lens = torch.randint(1, Lk + 1, (B,), generator=g)
    lens[0] = Lk
    kpm = torch.arange(Lk)[None, :] >= lens[:, None]
    x = rnd(B, Lk, 3 * dm, dtype=dtype, seed=92)
    qd = x[..., :dm] if self_attn else rnd(B, Lq, dm, dtype=dtype, seed=93)
    kd, vd = x[..., dm:2 * dm], x[..., 2 * dm:]
    dout = rnd(B, Lq, dm, dtype=dtype, seed=94)
    qr, kr, vr = (t.float().contiguous().requires_grad_(True) for t in (qd, kd, vd))
    ref, _ = attn_ref(qr, kr, vr, H, kpm, False)
    qvalid = ~kpm if self_attn else torch.ones(B, Lq, dtype=torch.bool)
    ref.backward(dout.float() * qvalid[..., None])       # padded query rows carry no gradient
    sel_k = (~kpm).flatten().nonzero().flatten()
    cu_k = torch.cat((torch.zeros(1, dtype=torch.long), lens.cumsum(0))).to(torch.int32).to(DEV)
    xp = x.reshape(B * Lk, 3 * dm)[sel_k].to(DEV)         # packed rows keep the in_proj layout [N, 3*dm]
    kp, vp = xp[:, dm:2 * dm], xp[:, 2 * dm:]
    if self_attn:
        qp, cu_q, sel_q = xp[:, :dm], cu_k, sel_k
    else:
        qp, cu_q, sel_q = qd.reshape(B * Lq, dm).to(DEV), None, torch.arange(B * Lq)
    dop = dout.reshape(B * Lq, dm)[sel_q].to(DEV)
    o, lse = ops.attn_varlen_fwd(qp, kp, vp, H, cu_q, cu_k, B, Lq, Lk)
    t = tol(dtype)
    assert rel_err(o, ref.reshape(B * Lq, dm)[sel_q]) < t
    dq, dk, dv = ops.attn_varlen_bwd(dop, qp, kp, vp, o, lse, H, cu_q, cu_k, B, Lq, Lk)
    assert rel_err(dq, qr.grad.reshape(B * Lq, dm)[sel_q]) < t
    assert rel_err(dk, kr.grad.reshape(B * Lk, dm)[sel_k]) < t
    assert rel_err(dv, vr.grad.reshape(B * Lk, dm)[sel_k]) < t
    # a dispatch order (longest first, as pa_pack_rows provides) changes which block does what, not the results
    order = torch.sort(lens, descending=True, stable=True).indices.to(torch.int32).to(DEV)
    o2, lse2 = ops.attn_varlen_fwd(qp, kp, vp, H, cu_q, cu_k, B, Lq, Lk, order=order)
    rows_ok = qvalid[:, None, :].expand(B, H, Lq).to(DEV)     # lse rows past a packed element's length are never written
    assert torch.equal(o2, o) and torch.equal(lse2[rows_ok], lse[rows_ok])
    dq2, dk2, dv2 = ops.attn_varlen_bwd(dop, qp, kp, vp, o, lse, H, cu_q, cu_k, B, Lq, Lk, order=order)
    assert torch.equal(dq2, dq) and torch.equal(dk2, dk) and torch.equal(dv2, dv)


def test_attention_key_split_blocks_same_dropout_as_dense():
    """The in-block key split (16-wave blocks: two key halves merged through LDS; taken by packed launches with at most one
    block per CU - cross-attention) must make the decisions of the unsplit kernels: with the same seed, packed K/V + split
    == dense K/V under a key-padding mask (32-row-wave / unsplit kernels), forward and all three gradients, dropout on.
    Lengths include elements shorter than one key tile (second half empty) and an odd number of tiles."""
    B, H, dh, Lq, Lk, p, seed = 6, 8, 64, 128, 1021, 0.2, 4711
    dm = H * dh
    lens = torch.tensor([1021, 40, 130, 577, 960, 64])
    kpm = torch.arange(Lk)[None, :] >= lens[:, None]
    q = rnd(B, Lq, dm, dtype=torch.bfloat16, seed=120).to(DEV)
    kv = rnd(B, Lk, 2 * dm, dtype=torch.bfloat16, seed=121).to(DEV)
    dout = rnd(B, Lq, dm, dtype=torch.bfloat16, seed=122).to(DEV)
    k, v = kv[..., :dm], kv[..., dm:]
    o_d, lse_d = ops.attn_fwd(q, k, v, H, kpm=kpm.to(DEV), drop_p=p, drop_seed=seed)
    dq_d, dk_d, dv_d = ops.attn_bwd(dout, q, k, v, o_d, lse_d, H, kpm=kpm.to(DEV), drop_p=p, drop_seed=seed)
    sel = (~kpm).flatten().nonzero().flatten().to(DEV)
    cu_k = torch.cat((torch.zeros(1, dtype=torch.long), lens.cumsum(0))).to(torch.int32).to(DEV)
    kvp = kv.reshape(B * Lk, 2 * dm)[sel]
    qp, dop = q.reshape(B * Lq, dm), dout.reshape(B * Lq, dm)
    o, lse = ops.attn_varlen_fwd(qp, kvp[:, :dm], kvp[:, dm:], H, None, cu_k, B, Lq, Lk, drop_p=p, drop_seed=seed)
    assert rel_err(o, o_d.reshape(B * Lq, dm)) < 1e-2 and float((lse - lse_d).abs().max()) < 1e-3
    dq, dk, dv = ops.attn_varlen_bwd(dop, qp, kvp[:, :dm], kvp[:, dm:], o, lse, H, None, cu_k, B, Lq, Lk, drop_p=p, drop_seed=seed)
    assert rel_err(dq, dq_d.reshape(B * Lq, dm)) < 1.5e-2
    assert rel_err(dk, dk_d.reshape(B * Lk, dm)[sel]) < 1.5e-2 and rel_err(dv, dv_d.reshape(B * Lk, dm)[sel]) < 1.5e-2
    o0, _ = ops.attn_varlen_fwd(qp, kvp[:, :dm], kvp[:, dm:], H, None, cu_k, B, Lq, Lk)
    assert rel_err(o, o0) > 0.05                                   # (dropout really was on)


@pytest.mark.parametrize("lens,drop", [([1021, 33, 700, 577, 960, 64, 129, 513, 1000, 450], 0.2),
                                        ([1199, 1199, 85, 640, 577], 0.0), ([1021] * 16, 0.2), ([577, 64, 300], 0.2)])
def test_attention_range_blocks_equal_one_block_per_tile(lens, drop):
    if os.environ.get("PA_ATTN_SPLIT", "0") == "0":
        pytest.skip("range blocks are opt-in (PA_ATTN_SPLIT=1, read once per process): run by test_attention_range_blocks_in_a_child_process")
    _range_blocks_case(lens, drop)


def test_attention_range_blocks_in_a_child_process():
    """The opt-in range-block path stays a tested path: the cases above in a child process with PA_ATTN_SPLIT=1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PA_ATTN_SPLIT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "tests/test_kernels_gpu.py", "-k",
                        "test_attention_range_blocks_equal_one_block_per_tile"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "4 passed" in r.stdout, r.stdout[-2000:]


def _range_blocks_case(lens, drop):
    """Packed bf16 self-attention with scratch (pa_attn_args.ws; csrc/attention.hip decode_unit_split): elements with more than
    eight 64-row tiles run as several RANGE blocks per owned tile whose partial results (O, m, l resp. dQ / dK, dV sums) meet in
    HBM and are merged by the last block to arrive.  Same function as one block per tile: same dropout decisions, lse equal to
    f32 rounding, outputs equal to bf16 rounding; run twice on ONE scratch buffer (tickets must return to zero), in a launch
    mix that keeps other blocks busy on every CU (uneven lengths), and every ticket word is zero afterwards."""
    B, H, dh, p, seed = len(lens), 8, 64, drop, 991
    dm, S = H * dh, max(lens)
    cu, order = ops.pack_lengths(lens, DEV)
    n = int(cu[-1])
    x = rnd(n, 3 * dm, dtype=torch.bfloat16, seed=140).to(DEV)
    q, k, v = x[:, :dm], x[:, dm:2 * dm], x[:, 2 * dm:]
    dout = rnd(n, dm, dtype=torch.bfloat16, seed=141).to(DEV)
    kw = dict(drop_p=p, drop_seed=seed, order=order)
    o0, lse0 = ops.attn_varlen_fwd(q, k, v, H, cu, cu, B, S, S, **kw)
    dq0, dk0, dv0 = ops.attn_varlen_bwd(dout, q, k, v, o0, lse0, H, cu, cu, B, S, S, **kw)
    ws = ops.attn_split_ws(n, B, H, DEV, L_max=S)
    if S > 512:
        assert ws is not None
    rows_ok = (torch.arange(S)[None, :] < torch.tensor(lens)[:, None])[:, None, :].expand(B, H, S).to(DEV)
    for rep in range(2):
        o, lse = ops.attn_varlen_fwd(q, k, v, H, cu, cu, B, S, S, ws=ws, **kw)
        assert rel_err(o, o0) < 4e-3, rel_err(o, o0)
        assert float((lse[rows_ok] - lse0[rows_ok]).abs().max()) < 2e-5
        dq, dk, dv = ops.attn_varlen_bwd(dout, q, k, v, o0, lse0, H, cu, cu, B, S, S, ws=ws, **kw)
        for a, b_ in ((dq, dq0), (dk, dk0), (dv, dv0)):
            assert rel_err(a, b_) < 4e-3, rel_err(a, b_)
        if ws is not None:
            nt = int(L.lib().pa_attn_ws_ticket_bytes(ws.numel()))
            assert int(ws[:nt].view(torch.int32).abs().sum()) == 0
    if S > 512:
        # the split really ran: some row of a long element differs in the last bf16 bit or the f32 sums (different summation order)
        assert not (torch.equal(o, o0) and torch.equal(dq, dq0) and torch.equal(dk, dk0))


@pytest.mark.parametrize("M,N,K,eps,relu", [(256, 512, 512, 1.0, False), (256, 1536, 512, 1.0, False), (256, 1024, 512, 1.0, True),
                                              (128, 512, 512, 1e-5, False), (200, 96, 128, 1.0, True), (77, 512, 512, 1.0, True),
                                              (600, 512, 512, 1.0, False)])
def test_gemm_on_folded_layernorm(M, N, K, eps, relu):
    """pa_gemm_norm_a: Linear(LayerNorm(z)) with the LayerNorm folded into the product, rstd (z (W gamma)^T - mean u) + v (the
    greedy-decode step's form; post-norm layers of torch's TransformerDecoderLayer with the reference's eps = 1.0), against
    torch: LayerNorm in f32 on the bf16 rows, then the Linear; and the materialised LayerNorm(z) itself.  Rows carry a
    large common offset (mean >> std): the regime where  acc - mean u  cancels most of acc."""
    z = (rnd(M, K, seed=130) * 1.5 + rnd(M, 1, seed=131) * 6.0).to(torch.bfloat16)
    w, b = rnd(N, K, seed=132, scale=0.05), rnd(N, seed=133, scale=0.1)
    gamma, beta = 1.0 + 0.2 * rnd(K, seed=134), 0.1 * rnd(K, seed=135)
    zf = z.float()
    y_ref = torch.nn.functional.layer_norm(zf, (K,), gamma, beta, eps)
    ref = y_ref @ w.t() + b
    if relu:
        ref = torch.relu(ref)
    out, y = ops.gemm_norm_a(z.to(DEV), w.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV), eps, relu=relu)
    assert rel_err(y, y_ref) < 1e-2                          # bf16 rounding of y
    assert rel_err(out, ref) < 2.5e-2, rel_err(out, ref)
    # and against what the unfolded pair of launches gives (bf16 LayerNorm output, bf16 weight): same ballpark of error
    y2, _, _ = ops.layernorm_fwd(z.to(DEV), gamma.to(DEV), beta.to(DEV), eps)
    out2 = ops.gemm(y2, w.to(torch.bfloat16).to(DEV), bias=b.to(DEV), relu=relu)
    assert rel_err(out, ref) < 2.0 * rel_err(out2, ref) + 5e-3, (rel_err(out, ref), rel_err(out2, ref))


def test_gemm_wide_tile_bf16():
    """Large multi-round Linears: ragged M, N not a multiple of 256, every epilogue.  Runs on the two-blocks-per-CU kernel
    by default and on the 128 x 256 tile kernel under PA_GEMM_WIDE=1 (both were validated with this test)."""
    M, K = 7940, 512
    for N in (1024, 1536, 1100):
        a, b = rnd(M, K, dtype=torch.bfloat16, seed=70, scale=0.3), rnd(N, K, dtype=torch.bfloat16, seed=71 + N, scale=0.3)
        bias, res = rnd(N, seed=72), rnd(M, N, dtype=torch.bfloat16, seed=73)
        acc = a.float() @ b.float().t()
        out = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), relu=True)
        assert rel_err(out, torch.relu(acc + bias)) < 2.5e-2, N
        out = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), residual=res.to(DEV), out_dtype=torch.bfloat16)
        assert rel_err(out, acc + bias + res.float()) < 2.5e-2, N
    a, b = rnd(4096, 1024, dtype=torch.bfloat16, seed=74, scale=0.3), rnd(2048, 1024, dtype=torch.bfloat16, seed=75, scale=0.3)
    out = ops.gemm(a.to(DEV), b.to(DEV), out_dtype=torch.float32)
    assert rel_err(out, a.float() @ b.float().t()) < 1e-2


@pytest.mark.parametrize("M", [8704, 8300, 9736, 12288])
def test_gemm_tall_tile_bf16(M):
    """N = 512 Linears a few 128 x 128 tiles past one round of the CUs (the packed encoder rows of half the batches) on 192 x 128
    tiles, one round (gemm3w_kernel<3, 2>; opt-in, PA_GEMM_TALL=1, read once per process: measured level with the two-blocks-per-CU
    kernel).  Ragged M (rows past M in the last tile), every epilogue incl. the dropout decisions (reference: the float product
    and, for dropout, the keep pattern of a small-tile launch on the same rows).  Without the switch the same cases run on the
    default kernel; test_gemm_tall_tile_in_a_child_process runs them with it."""
    K = 512
    tall_on = os.environ.get("PA_GEMM_TALL", "0") != "0"
    for N in (512, 384):
        a, b = rnd(M, K, dtype=torch.bfloat16, seed=170, scale=0.3), rnd(N, K, dtype=torch.bfloat16, seed=171 + N, scale=0.3)
        bias, res, aux = rnd(N, seed=172), rnd(M, N, dtype=torch.bfloat16, seed=173), rnd(M, N, dtype=torch.bfloat16, seed=174)
        acc = a.float() @ b.float().t()
        ad, bd = a.to(DEV), b.to(DEV)
        (out, kinds) = _gemm_kinds(lambda: ops.gemm(ad, bd, bias=bias.to(DEV), relu=True))
        assert rel_err(out, torch.relu(acc + bias)) < 2.5e-2, N
        if tall_on and ((M + 127) // 128) * ((N + 127) // 128) > 256 and ((M + 191) // 192) * ((N + 127) // 128) <= 256:
            assert kinds == [_lib_kind("WIDE")], kinds
        out = ops.gemm(ad, bd, bias=bias.to(DEV), residual=res.to(DEV), out_dtype=torch.bfloat16)
        assert rel_err(out, acc + bias + res.float()) < 2.5e-2, N
        out = ops.gemm(ad, bd, aux=aux.to(DEV), aux_scale=1.25, out_dtype=torch.float32)
        assert rel_err(out, torch.where(aux.float() > 0, acc * 1.25, torch.zeros_like(acc))) < 2.5e-2, N
        # dropout: the same (row, column) decisions as a launch of the first 2 048 rows alone (64 x 64-tile kernel)
        out = ops.gemm(ad, bd, drop_p=0.3, drop_seed=11, out_dtype=torch.float32)
        ref = ops.gemm(ad[:2048].contiguous(), bd, drop_p=0.3, drop_seed=11, out_dtype=torch.float32)
        assert torch.equal(out[:2048] == 0, ref == 0)
        kept = out != 0
        assert abs(float(kept.float().mean()) - 0.7) < 0.01
        assert rel_err(out[kept], (acc / 0.7).to(DEV)[kept]) < 2.5e-2


def test_gemm_tall_tile_in_a_child_process():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "tests/test_kernels_gpu.py", "-k", "test_gemm_tall_tile_bf16"],
                       cwd=root, env=dict(os.environ, PA_GEMM_TALL="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "4 passed" in r.stdout, r.stdout[-2000:]


def _lib_kind(name):
    return {"PAIR": 0, "RING": 1, "WIDE": 2, "SMALL": 3, "SKINNY": 4, "BIG": 5}[name]      # include/plank_hip.h PA_GEMM_KIND_*


def _gemm_kinds(fn):
    import ctypes as C
    from plankassembly_amd import _lib as L
    lib = L.lib()
    torch.cuda.synchronize()
    lib.pa_gemm_record(1)
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        n = lib.pa_gemm_record(0)
    kinds = (C.c_int32 * max(n, 1))()
    nk = lib.pa_gemm_recorded_kinds(C.cast(kinds, C.c_void_p), n)
    return out, [kinds[i] for i in range(nk)]


@pytest.mark.parametrize("M", [256, 250, 37])
def test_gemm_skinny_f32_residual_stream_forms(M):
    """The two Linear forms of the bf16 decode step's f32 residual stream (csrc/decode.hip f32res): (a) Z (f32) = A W^T + b + R (f32)
    with a bf16 copy of Z for the next product; (b) LayerNorm(Zf) W^T + b folded, the product on the bf16 copy, the statistics
    and the materialised LayerNorm(Zf) (f32) from the f32 rows."""
    K, N = 512, 1536
    a, w = rnd(M, K, dtype=torch.bfloat16, seed=71, scale=0.5), rnd(512, K, dtype=torch.bfloat16, seed=72, scale=0.1)
    bias, res = rnd(512, seed=73), rnd(M, 512, seed=74, scale=3.0)
    out = torch.empty(M, 512, dtype=torch.float32, device=DEV)
    lp = torch.full((M, 512), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), residual=res.to(DEV), out=out, out_lp=lp)
    ref = a.float() @ w.float().t() + bias + res
    assert rel_err(out, ref) < 1e-5                                        # bf16 operands are exact inputs; f32 accumulate + f32 residual
    assert torch.equal(lp.cpu(), out.cpu().to(torch.bfloat16))
    # (b) rows with a large common offset: statistics from the bf16 copy would be visibly off, from the f32 rows they are not
    zf = rnd(M, K, seed=75, scale=2.0) + 40.0
    wn, bn = rnd(N, K, seed=76, scale=0.05), rnd(N, seed=77)
    gamma, beta = 1.0 + 0.1 * rnd(K, seed=78), 0.1 * rnd(K, seed=79)
    o2, y = ops.gemm_norm_a(zf.to(torch.bfloat16).to(DEV), wn.to(DEV), bn.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, zf=zf.to(DEV))
    yref = torch.nn.functional.layer_norm(zf, (K,), gamma, beta, 1e-5)
    assert y.dtype == torch.float32 and float((y.cpu() - yref).abs().max()) < 2e-4      # 40 +- 2 rows: f32 statistics
    oref = yref @ wn.t() + bn
    # the product sees bf16(zf): |z| ~ 40 rounds by up to 0.125 against a row spread of 2 - the operand's rounding, not the statistics'
    assert rel_err(o2, oref) < 0.2 and o2.dtype == torch.bfloat16
    zc = rnd(M, K, seed=80, scale=2.0)                                     # centred rows: the usual bf16 tolerance
    o3, y3 = ops.gemm_norm_a(zc.to(torch.bfloat16).to(DEV), wn.to(DEV), bn.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, zf=zc.to(DEV))
    y3ref = torch.nn.functional.layer_norm(zc, (K,), gamma, beta, 1e-5)
    assert float((y3.cpu() - y3ref).abs().max()) < 1e-4 and rel_err(o3, y3ref @ wn.t() + bn) < 2.5e-2


def test_gemm_eight_wave_big_tiles():
    """The opt-in eight-wave 256 x 256 / 256 x 128 kernel (PA_GEMM_BIG=1, csrc/gemm8.h) in its own process: every epilogue stage,
    ragged edge tiles, several units per block, the batched launch; the worker also checks the launches went to that kernel."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gemm_big_worker.py")], cwd=root, capture_output=True, text=True,
                       env=dict(os.environ, PA_GEMM_BIG="1"), timeout=600)
    assert r.returncode == 0 and "gemm8 ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(256, 512, 512), (256, 1536, 512), (256, 512, 1024), (256, 514, 512), (250, 1024, 512),
                                   (2, 512, 512), (37, 100, 1536), (512, 96, 2048)])
def test_gemm_skinny_rows_every_epilogue(dtype, M, N, K):
    """gemm_skinny_kernel (<= 512 rows against a whole weight: the greedy-decode Linears; PA_GEMM_SKINNY=1 restricts it to f32,
    0 switches it off): ragged M and N, one to four K chunks, bias / ReLU / residual / alpha, f32 and same-type outputs - against
    torch, and bit-identical between two launches (the four K-quarters are summed in a fixed order)."""
    import os
    if os.environ.get("PA_GEMM_SKINNY", "2") != "2":
        pytest.skip("PA_GEMM_SKINNY overrides the default dispatch")
    a, b = rnd(M, K, dtype=dtype, seed=31, scale=0.5), rnd(N, K, dtype=dtype, seed=32, scale=0.5)
    bias, res = rnd(N, seed=33), rnd(M, N, dtype=dtype, seed=34)
    acc = a.double() @ b.double().t()
    t = 1e-5 if dtype == torch.float32 else 1.5e-2
    ad, bd = a.to(DEV), b.to(DEV)
    out, kinds = _gemm_kinds(lambda: ops.gemm(ad, bd, bias=bias.to(DEV), relu=True))
    assert kinds == [4], kinds                                               # PA_GEMM_KIND_SKINNY
    assert rel_err(out, torch.relu(acc + bias.double()).float()) < t
    out = ops.gemm(ad, bd, bias=bias.to(DEV), residual=res.to(DEV))
    assert rel_err(out, (acc + bias.double() + res.double()).float()) < t
    assert torch.equal(out, ops.gemm(ad, bd, bias=bias.to(DEV), residual=res.to(DEV)))
    out = ops.gemm(ad, bd, alpha=0.25, out_dtype=torch.float32)
    assert out.dtype == torch.float32 and rel_err(out, (0.25 * acc).float()) < (1e-5 if dtype == torch.float32 else 5e-3)
    if N % 4:                                                                # a padded output row (the vocabulary head writes ld 516)
        buf = torch.full((M, N + 6), 7.0, dtype=torch.float32, device=DEV)
        ops.gemm(ad, bd, bias=bias.to(DEV), out_dtype=torch.float32, out=buf[:, :N])
        assert rel_err(buf[:, :N], (acc + bias.double()).float()) < (1e-5 if dtype == torch.float32 else 5e-3)
        assert float(buf[:, N:].min()) == 7.0 and float(buf[:, N:].max()) == 7.0


def test_transpose_many_bf16_edges():
    """pa_transpose_many: 8-byte path (aligned matrices, ragged row count, padded / offset destination whose padding
    must stay untouched) and the scalar path (unaligned shapes), all in one launch."""
    import ctypes as C
    from plankassembly_amd import _lib as L
    from plankassembly_amd.models import _TrDesc
    specs = [(514, 512, 576, 0), (1024, 512, 6144, 2048), (96, 40, 96, 0), (30, 18, 30, 0)]   # rows, cols, ld_dst, dst column offset
    srcs, dsts, descs, tiles = [], [], [], 0
    for i, (r, c, ldd, off) in enumerate(specs):
        src = rnd(r, c, dtype=torch.bfloat16, seed=80 + i).to(DEV)
        dst = torch.full((c, ldd), 7.0, dtype=torch.bfloat16, device=DEV)
        srcs.append(src); dsts.append(dst)
        descs.append(_TrDesc(src.data_ptr(), dst.data_ptr() + off * 2, r, c, c, ldd, tiles, 0))
        tiles += ((r + 63) // 64) * ((c + 63) // 64)
    arr = (_TrDesc * len(descs))(*descs)
    dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    L.check(L.lib().pa_transpose_many(L.ptr(dev), len(descs), tiles, L.PA_BF16, L.stream()), "pa_transpose_many")
    torch.cuda.synchronize()
    for (r, c, ldd, off), src, dst in zip(specs, srcs, dsts):
        assert torch.equal(dst[:, off:off + r].cpu(), src.t().cpu())
        pad = torch.cat([dst[:, :off], dst[:, off + r:]], 1)
        assert bool((pad == 7.0).all())                      # nothing written outside the transposed block


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_segment_bwd_skewed_rows(dtype):
    """pa_embed_segment_bwd with a skewed id distribution: one table row used by thousands of tokens (split over several
    blocks), rows used by exactly the single-block limit, unused rows, and a small table on the chunked path; the
    gradients accumulate onto the existing table contents."""
    import ctypes as C
    from plankassembly_amd import _lib as L
    from plankassembly_amd.models import group_rows_by_id
    n, d = 6000, 512
    g = torch.Generator().manual_seed(91)
    big = torch.randint(0, 300, (n,), generator=g)
    big[:3000] = 7                                     # heavy row
    big[3000:3064] = 11                                # exactly 64 more users of one row
    big[big == 5] = 6                                  # row 5 unused
    small = torch.randint(0, 6, (n,), generator=g)
    dout = rnd(n, d, dtype=dtype, seed=92).to(DEV)
    tabs = [torch.full((300, d), 0.5, device=DEV), torch.full((6, d), -0.25, device=DEV)]
    ids = [big.to(DEV), small.to(DEV)]
    groups = [group_rows_by_id(i, t.shape[0]) for i, t in zip(ids, tabs)]
    PP = C.c_void_p * 2
    rows = (C.c_int32 * 2)(300, 6)
    L.check(L.lib().pa_embed_segment_bwd(L.ptr(dout), L.dt(dout), PP(*[t.data_ptr() for t in tabs]),
                                         PP(*[o.data_ptr() for o, _ in groups]), PP(*[s.data_ptr() for _, s in groups]),
                                         rows, 2, n, d, L.stream()), "pa_embed_segment_bwd")
    torch.cuda.synchronize()
    for t, i, base in zip(tabs, ids, (0.5, -0.25)):
        ref = torch.full(t.shape, base, dtype=torch.float64).index_add_(0, i.cpu(), dout.double().cpu())
        assert rel_err(t, ref.float()) < 2e-6
    assert bool((tabs[0][5] == 0.5).all())


@pytest.mark.parametrize("M,K,drop,res", [(256, 512, 0.0, True), (2048, 1024, 0.2, True), (7940, 512, 0.2, True), (77, 512, 0.0, False),
                                          (8192, 1024, 0.1, True)])
def test_gemm_ln_fused_equals_separate_launches(M, K, drop, res):
    """pa_gemm_ln (Linear + bias + dropout + residual + LayerNorm in one launch) against pa_gemm followed by
    pa_layernorm_fwd: z, y, mean, rstd bit for bit, and against torch in f32 within the bf16 tolerance."""
    x, w = rnd(M, K, dtype=torch.bfloat16, seed=300).to(DEV), (rnd(512, K, seed=301) * 0.05).to(torch.bfloat16).to(DEV)
    bias, gamma, beta = rnd(512, seed=302).to(DEV), (1 + 0.1 * rnd(512, seed=303)).to(DEV), (0.1 * rnd(512, seed=304)).to(DEV)
    r = rnd(M, 512, dtype=torch.bfloat16, seed=305).to(DEV) if res else None
    z0 = ops.gemm(x, w, bias=bias, residual=r, drop_p=drop, drop_seed=77)
    y0, m0, r0 = ops.layernorm_fwd(z0, gamma, beta, 1e-5)
    z1, y1, m1, r1 = ops.gemm_ln(x, w, gamma, beta, 1e-5, bias=bias, residual=r, drop_p=drop, drop_seed=77)
    torch.cuda.synchronize()
    _, y2, _, _ = ops.gemm_ln(x, w, gamma, beta, 1e-5, bias=bias, residual=r, drop_p=drop, drop_seed=77, want_z=False)
    assert torch.equal(y2, y1)
    if M <= 512 and drop == 0.0:
        # pa_gemm sends <= 512 dropout-free rows to gemm_skinny_kernel, which sums K in four quarters: same values to f32 rounding,
        # not the same bits (one bf16 ulp at most after rounding)
        assert float((z1.float() - z0.float()).abs().max()) <= 2.0 ** -6 and rel_err(z1, z0.cpu()) < 4e-3
        assert rel_err(y1, y0.cpu()) < 8e-3 and rel_err(m1, m0.cpu()) < 1e-3 and rel_err(r1, r0.cpu()) < 1e-3
    else:
        assert torch.equal(z1, z0), float((z1.float() - z0.float()).abs().max())
        assert torch.equal(m1, m0) and torch.equal(r1, r0)
        assert torch.equal(y1, y0), float((y1.float() - y0.float()).abs().max())
    if drop == 0.0:
        zt = x.float() @ w.float().T + bias + (r.float() if res else 0.0)
        yt = torch.nn.functional.layer_norm(zt, (512,), gamma, beta, 1e-5)
        assert rel_err(y1, yt.cpu()) < tol(torch.bfloat16)


# ------------------------------------------------------------------------------------------------------------------------------
# bf16x3 ("split") attention (csrc/attention_x3.h): the exact-f32 kernels' function with every product as hi*hi + hi*lo + lo*hi
@pytest.mark.parametrize("B,H,Lq,Lk,use_kpm,causal,drop", [
    (2, 8, 128, 1199, True, False, 0.0),      # default cross-attention
    (1, 8, 300, 300, True, False, 0.2),       # self-attention, ragged tiles, dropout (the same decisions as exact f32)
    (2, 8, 128, 128, True, True, 0.2),        # default decoder self-attention
    (2, 4, 1024, 1024, True, False, 0.0),     # benchmark length
])
def test_attention_x3_equals_exact_f32_to_split_precision(B, H, Lq, Lk, use_kpm, causal, drop):
    from plankassembly_amd import _lib as L
    dh, dm = 64, H * 64
    qkv_q = rnd(B, Lq, 3 * dm, dtype=torch.float32, seed=50).to(DEV)
    qkv_k = qkv_q if Lq == Lk else rnd(B, Lk, 3 * dm, dtype=torch.float32, seed=51).to(DEV)
    q, k, v = qkv_q[..., :dm], qkv_k[..., dm:2 * dm], qkv_k[..., 2 * dm:]
    kpm = None
    if use_kpm:
        g = torch.Generator().manual_seed(52)
        valid = torch.randint(max(1, Lk // 3), Lk + 1, (B,), generator=g)
        kpm = (torch.arange(Lk)[None, :] >= valid[:, None])
        kpm[0, 3] = True
        kpm = kpm.to(DEV)
    dout = rnd(B, Lq, dm, dtype=torch.float32, seed=53).to(DEV)
    kw = dict(kpm=kpm, causal=causal, drop_p=drop, drop_seed=77)

    def run(split):
        L.check(L.lib().pa_attn_split_config(1 if split else 0), "pa_attn_split_config")
        try:
            n0 = L.lib().pa_attn_split_taken(1)
            o, lse = ops.attn_fwd(q, k, v, H, **kw)
            dq, dk, dv = ops.attn_bwd(dout, q, k, v, o, lse, H, **kw)
            torch.cuda.synchronize()
            return (o, lse, dq, dk, dv), int(L.lib().pa_attn_split_taken(1))
        finally:
            L.check(L.lib().pa_attn_split_config(0), "pa_attn_split_config")

    exact, n_exact = run(False)
    x3, n_x3 = run(True)
    assert n_exact == 0 and n_x3 == 2                         # forward + backward really took the split kernels
    for name, a, b in zip(("o", "lse", "dq", "dk", "dv"), x3, exact):
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        assert err <= 3e-5 * scale + 1e-6, (name, err, scale)       # three-term split: ~2^-17 per product (plain bf16 would be ~4e-3)
        assert err > 0.0 or name == "lse"                           # ... and it is not the exact kernel answering


def test_attention_x3_packed_element_without_keys_gets_zero_gradients():
    """a packed (cu_k) batch element with NO keys - cu_k[b + 1] == cu_k[b] - must come out of the bf16x3 backward with zero dQ
    rows like the exact-f32 kernels' (the split dQ kernel used to return before its store: uninitialised memory), and must not read
    before its key range either.  Output buffers are pre-filled with NaN by running on torch.empty storage we poison first."""
    from plankassembly_amd import _lib as L
    B, H, Lq, dm = 3, 8, 128, 512
    lens = [70, 0, 130]
    cu = torch.tensor([0, 70, 70, 200], dtype=torch.int32, device=DEV)
    q = rnd(B * Lq, dm, dtype=torch.float32, seed=60).to(DEV)
    kv = rnd(200, 2 * dm, dtype=torch.float32, seed=61).to(DEV)
    k, v = kv[:, :dm], kv[:, dm:]
    dout = rnd(B * Lq, dm, dtype=torch.float32, seed=62).to(DEV)

    def run(split):
        L.check(L.lib().pa_attn_split_config(1 if split else 0), "pa_attn_split_config")
        try:
            poison = torch.full((4 * B * Lq * dm,), float("nan"), device=DEV)      # what the next torch.empty calls will hand out
            del poison
            o, lse = ops.attn_varlen_fwd(q, k, v, H, None, cu, B, Lq, max(lens))
            dq, dk, dv = ops.attn_varlen_bwd(dout, q, k, v, o, lse, H, None, cu, B, Lq, max(lens))
            torch.cuda.synchronize()
            return o, dq, dk, dv
        finally:
            L.check(L.lib().pa_attn_split_config(0), "pa_attn_split_config")

    exact, x3 = run(False), run(True)
    for name, a, b in zip(("o", "dq", "dk", "dv"), x3, exact):
        assert torch.isfinite(a).all(), name
        assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()) + 1e-6, name
    assert float(x3[1][Lq:2 * Lq].abs().max()) == 0.0 and float(x3[0][Lq:2 * Lq].abs().max()) == 0.0     # the key-less element's rows


def test_attention_mask_order_dispatch_leaves_results_unchanged():
    """ops.mask_order (longest batch element first, pa_attn_args.order on a padded batch with a key-padding mask): a permutation
    of the block -> batch-element mapping only - forward and backward outputs are bit-identical to the plain order, with dropout."""
    B, H, S, dm = 6, 8, 320, 512
    qkv = rnd(B, S, 3 * dm, dtype=torch.bfloat16, seed=60).to(DEV)
    q, k, v = qkv[..., :dm], qkv[..., dm:2 * dm], qkv[..., 2 * dm:]
    valid = torch.tensor([320, 40, 200, 129, 64, 257])
    kpm = (torch.arange(S)[None, :] >= valid[:, None]).to(DEV)
    order = ops.mask_order(kpm)
    assert order.tolist() == [0, 5, 2, 3, 4, 1] and order.dtype == torch.int32
    do = rnd(B, S, dm, dtype=torch.bfloat16, seed=61).to(DEV)
    kw = dict(kpm=kpm, drop_p=0.2, drop_seed=9)
    o0, l0 = ops.attn_fwd(q, k, v, H, **kw)
    o1, l1 = ops.attn_fwd(q, k, v, H, order=order, **kw)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    g0 = ops.attn_bwd(do, q, k, v, o0, l0, H, **kw)
    g1 = ops.attn_bwd(do, q, k, v, o0, l0, H, order=order, **kw)
    assert all(torch.equal(a, b) for a, b in zip(g0, g1))


# ------------------------------------------------------------------------------------------------------------------------------
# bf16x3 ("split") GEMMs at the C ABI (csrc/gemm.hip gemm_split3 / gemm_group_split3; include/plank_hip.h pa_gemm_split_config):
# f32 in / f32 out, every product as hi*hi + hi*lo + lo*hi on the bf16 matrix pipe.  Reference: float64 on the CPU.
class _SplitMode:
    """pa_gemm_split_config bracket over a scratch tensor owned by the test."""

    def __init__(self, mode, mb=256):
        import ctypes as C
        from plankassembly_amd import _lib as L
        self.C, self.L, self.mode = C, L, mode
        self.buf = torch.empty(mb * (1 << 20) + 256, dtype=torch.uint8, device=DEV)
        self.base = (self.buf.data_ptr() + 255) // 256 * 256
        self.bytes = self.buf.numel() - (self.base - self.buf.data_ptr())

    def config(self, mode):
        L, C = self.L, self.C
        L.check(L.lib().pa_gemm_split_config(mode, C.c_void_p(self.base) if mode else None, C.c_int64(self.bytes if mode else 0)),
                "pa_gemm_split_config")

    def __enter__(self):
        out = (self.C.c_int64 * 2)()
        self.L.lib().pa_gemm_split_stats(out, 1)
        self.config(self.mode)
        return self

    def __exit__(self, *exc):
        self.config(0)

    def stats(self):
        out = (self.C.c_int64 * 2)()
        self.L.lib().pa_gemm_split_stats(out, 0)
        return int(out[0]), int(out[1])

    def image_view(self, ptr, rows, cols):
        """[rows][3 cols] bf16 view of an image address inside the scratch buffer (pa_gemm_split_reserve)."""
        off = ptr - self.buf.data_ptr()
        return self.buf[off: off + rows * cols * 6].view(torch.bfloat16).view(rows, 3 * cols)


def _x3_err(got, a64, b64_t):
    """max |got - a b| over max(|a| |b|): the split drops lo*lo (2^-16 of a term) and rounds lo to 8 bits (2^-17)."""
    ref = a64 @ b64_t
    bound = (a64.abs() @ b64_t.abs()).max()
    return float((got.detach().cpu().double() - ref).abs().max() / bound)


@pytest.mark.parametrize("akc,bkc", [(True, True), (True, False), (False, False), (False, True)])
@pytest.mark.parametrize("M,N,K", [(256, 192, 128), (2048, 512, 512), (264, 520, 64), (1200, 1536, 512)])
def test_gemm_x3_layouts_match_float64_to_split_precision(akc, bkc, M, N, K):
    a = rnd(M, K, seed=1)
    b = rnd(N, K, seed=2)
    bias = rnd(N, seed=3).to(DEV)
    A = (a if akc else a.t().contiguous()).to(DEV)
    Bm = (b if bkc else b.t().contiguous()).to(DEV)
    exact = ops.gemm(A, Bm, a_kcontig=akc, b_kcontig=bkc, bias=bias)
    with _SplitMode(1) as sm:
        got = ops.gemm(A, Bm, a_kcontig=akc, b_kcontig=bkc, bias=bias)
        taken, declined = sm.stats()
    torch.cuda.synchronize()
    assert (taken, declined) == (1, 0), (taken, declined)
    bias64 = bias.cpu().double()
    e3 = _x3_err(got - bias, a.double(), b.double().t())
    e32 = _x3_err(exact - bias, a.double(), b.double().t())
    assert e3 < 2.0 ** -15, e3                      # measured ~2^-17.5 .. 2^-16.5 of max |a||b|; plain bf16 sits at ~2^-9
    assert e32 < 2.0 ** -20, e32
    assert not torch.equal(got, exact)               # (really a different arithmetic, not the exact kernel again)
    del bias64


def test_gemm_x3_epilogues_batch_and_splitk():
    """ReLU-backward gate (from a bf16 copy of the gate tensor: only its sign is read), residual, alpha, a batch with a shared
    A operand, deferred split-K through the split-aware slab count."""
    M, N, K = 512, 256, 256
    a, b = rnd(M, K, seed=4), rnd(N, K, seed=5)
    gate, res = rnd(M, N, seed=6), rnd(M, N, seed=7)
    ref = (a.double() @ b.double().t()) * 0.5
    ref = torch.where(gate.double() > 0, ref * 1.25, torch.zeros_like(ref)) + res.double()
    with _SplitMode(1) as sm:
        got = ops.gemm(a.to(DEV), b.to(DEV), aux=gate.to(DEV), aux_scale=1.25, residual=res.to(DEV), alpha=0.5)
        # batch of 3 weights against one shared activation matrix (the cross-attention K/V projection of all layers)
        wb = rnd(3, N, K, seed=8)
        A1 = a.to(DEV)
        gb = ops.gemm(A1[None].expand(3, M, K), wb.to(DEV))
        # split-K over the rows (a weight gradient): slabs sized by pa_gemm_effective_splitk in THIS mode, reduced afterwards
        dy, x = rnd(4096, 192, seed=9), rnd(4096, 128, seed=10)
        dw, ws, sk = ops.gemm(dy.to(DEV), x.to(DEV), a_kcontig=False, b_kcontig=False, splitk=4, defer=True)
        ops.splitk_reduce_many([(ws, dw, sk)])
        taken, declined = sm.stats()
    torch.cuda.synchronize()
    assert declined == 0 and taken == 3, (taken, declined)
    bound = float((a.double().abs() @ b.double().abs().t()).max())
    assert float((got.cpu().double() - ref).abs().max()) < 2.0 ** -15 * bound
    for i in range(3):
        assert _x3_err(gb[i], a.double(), wb[i].double().t()) < 2.0 ** -15
    assert sk > 1 and _x3_err(dw, dy.double().t(), x.double()) < 2.0 ** -15


def test_gemm_x3_retained_images_serve_the_weight_gradient():
    """The life of a Linear in retain modes 3 (forward) and 2 (backward segment): Y = X W^T keeps the cut X, dX = dY W keeps the cut dY,
    and dW = dY^T X - alone (pa_gemm) and inside a grouped launch (pa_gemm_group) - finds BOTH operands already cut.  A producer-written
    image (pa_gemm_split_reserve) is taken as it is: an image of 2 X in place of X doubles the product."""
    import ctypes as C
    from plankassembly_amd import _lib as L
    rows, din, dout = 1024, 256, 384
    x, w, dy = rnd(rows, din, seed=11), rnd(dout, din, seed=12), rnd(rows, dout, seed=13)
    X, W, DY = x.to(DEV), w.to(DEV), dy.to(DEV)
    WT = W.t().contiguous()
    sm = _SplitMode(3)
    with sm:
        y = ops.gemm(X, W)                                            # forward: A = X retained as (hi, hi, lo)
        sm.config(0)
        sm.config(2)                                                  # a backward segment
        dx = ops.gemm(DY, WT)                                         # dX = dY W as a k-contiguous GEMM over W^T: dY retained as (hi, lo, hi)
        r0 = int(L.lib().pa_gemm_split_reused())
        dw = ops.gemm(DY, X, a_kcontig=False, b_kcontig=False)        # lone weight gradient: both operands found
        r1 = int(L.lib().pa_gemm_split_reused())
        (dwg, dwg2) = ops.dw_group([(DY, X, 1), (DY, X, 1)])          # grouped: both members, both operands
        r2 = int(L.lib().pa_gemm_split_reused())
        sm.config(0)
        # producer-written image: reserve in a fresh forward, write the image of 2 X by hand, run the Linear on X
        sm.config(3)
        pat = C.c_int32(-1)
        ptr = L.lib().pa_gemm_split_reserve(C.c_void_p(X.data_ptr()), rows, din, din, C.byref(pat))
        assert ptr and pat.value == 0
        x2 = 2.0 * X
        hi = x2.to(torch.bfloat16)
        lo = (x2 - hi.float()).to(torch.bfloat16)
        sm.image_view(ptr, rows, din).copy_(torch.cat([hi, hi, lo], dim=1))
        m0 = int(L.lib().pa_gemm_split_made_hits())
        y2 = ops.gemm(X, W)
        m1 = int(L.lib().pa_gemm_split_made_hits())
    torch.cuda.synchronize()
    assert r1 - r0 == 2 and r2 - r1 == 4, (r0, r1, r2)
    assert m1 - m0 == 1
    x64, w64, dy64 = x.double(), w.double(), dy.double()
    assert _x3_err(y, x64, w64.t()) < 2.0 ** -15
    assert _x3_err(dx, dy64, w64) < 2.0 ** -15
    for g in (dw, dwg, dwg2):
        assert _x3_err(g, dy64.t(), x64) < 2.0 ** -15
    assert _x3_err(y2, 2.0 * x64, w64.t()) < 2.0 ** -15               # the hand-written image was what the GEMM multiplied
