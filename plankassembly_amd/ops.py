"""Thin torch-tensor wrappers over the op-level C ABI (include/plank_hip.h).

Used by the kernel parity tests and by host code that needs a single op.  The model-level
path (plankassembly_amd.models) drives the same kernels through the C++ runtime entry points.
All tensors must live on the GPU; everything enqueues on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib as L


def _f32(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


def gemm(a, b, *, a_kcontig=True, b_kcontig=True, bias=None, residual=None, aux=None, aux_scale=1.0,
         relu=False, alpha=1.0, drop_p=0.0, drop_seed=0, out_dtype=None, splitk=1, out=None, defer=False, out_lp=None):
    """C[b] = epi(alpha * A[b] @ B[b]).  a: [batch?, M, K] (or [K, M] if not a_kcontig);
    b: [batch?, N, K] if b_kcontig (Linear weight layout) else [K, N]."""
    batched = a.dim() == 3
    A3 = a if batched else a[None]
    B3 = b if b.dim() == 3 else b[None]
    batch = A3.shape[0]
    M, K = (A3.shape[1], A3.shape[2]) if a_kcontig else (A3.shape[2], A3.shape[1])
    N = B3.shape[1] if b_kcontig else B3.shape[2]
    assert (B3.shape[2] if b_kcontig else B3.shape[1]) == K
    assert A3.stride(2) == 1 and B3.stride(2) == 1
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = torch.empty((batch, M, N) if batched else (M, N), dtype=out_dtype, device=a.device)
    O3 = out if out.dim() == 3 else out[None]
    g = L.GemmArgs()
    g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    g.bias = bias.data_ptr() if bias is not None else None
    g.R = residual.data_ptr() if residual is not None else None
    g.aux = aux.data_ptr() if aux is not None else None
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = A3.stride(1), B3.stride(1), O3.stride(1)
    g.ldr = (residual if residual is not None else O3).stride(-2)
    g.ldaux = aux.stride(-2) if aux is not None else 0
    g.sA = A3.stride(0) if batch > 1 else 0
    g.sB = B3.stride(0) if (b.dim() == 3 and B3.shape[0] > 1) else 0
    g.sC = O3.stride(0) if batch > 1 else 0
    g.sR = residual.stride(0) if (residual is not None and residual.dim() == 3 and batch > 1) else 0
    g.sAux = aux.stride(0) if (aux is not None and aux.dim() == 3 and batch > 1) else 0
    g.sBias = bias.stride(0) if (bias is not None and bias.dim() == 2 and batch > 1) else 0     # [batch, N]: one bias per member
    g.batch = batch
    g.a_kcontig, g.b_kcontig = int(a_kcontig), int(b_kcontig)
    g.in_dtype, g.out_dtype = L.dt(a), L.dt(out)
    g.alpha, g.relu, g.aux_scale = alpha, int(relu), aux_scale
    g.drop_p, g.drop_seed = drop_p, drop_seed
    if out_lp is not None:                    # bf16 copy of an f32 output (skinny kernel: the f32-residual decode step's form)
        g.C_lp, g.ldc_lp = out_lp.data_ptr(), out_lp.stride(-2)
    splitk = int(L.lib().pa_gemm_effective_splitk(K, L.dt(a), splitk))       # slabs actually written
    g.splitk = splitk
    ws = None
    if splitk > 1:
        ws = _f32(splitk * batch * M * N, device=a.device)
        g.ws = ws.data_ptr()
        g.splitk_defer = int(defer)
    L.check(L.lib().pa_gemm(C.byref(g), L.stream()), "pa_gemm")
    if defer and splitk > 1:
        return out, ws, splitk  # slabs only: reduce with splitk_reduce_many([(ws, out, splitk), ...])
    return out


def dw_group(items):
    """Weight gradients dW_i = dY_i^T @ X_i for [(dY[rows, M], X[rows, N], splitk)], bf16 in / f32 out: ONE ring-kernel
    launch for all products (pa_gemm_group) plus one reduction launch for the split ones."""
    n = len(items)
    args = (L.GemmArgs * n)()
    outs, keep, red = [], [], []
    for i, (dy, x, sk) in enumerate(items):
        rows, M = dy.shape
        N = x.shape[1]
        sk = int(L.lib().pa_gemm_effective_splitk(rows, L.dt(dy), sk))
        out = torch.empty(M, N, dtype=torch.float32, device=dy.device)
        g = args[i]
        g.A, g.B, g.C = dy.data_ptr(), x.data_ptr(), out.data_ptr()
        g.M, g.N, g.K = M, N, rows
        g.lda, g.ldb, g.ldc = dy.stride(0), x.stride(0), out.stride(0)
        g.batch, g.a_kcontig, g.b_kcontig = 1, 0, 0
        g.in_dtype, g.out_dtype = L.dt(dy), L.dt(out)
        g.alpha, g.aux_scale, g.splitk = 1.0, 1.0, sk
        if sk > 1:
            ws = _f32(sk * M * N, device=dy.device)
            g.ws, g.splitk_defer = ws.data_ptr(), 1
            keep.append(ws)
            red.append((ws, out, sk))
        outs.append(out)
    L.check(L.lib().pa_gemm_group(C.cast(args, C.c_void_p), n, L.stream()), "pa_gemm_group")
    if red:
        splitk_reduce_many(red)
    return outs


class ColsumDesc(C.Structure):          # mirrors pa_colsum_desc
    _fields_ = [("X", C.c_void_p), ("out", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("ldx", C.c_int32),
                ("pad_", C.c_int32)]


def colsum_many(items):
    """items: [(x[M, N], out[N] f32)]: out += column sums of x, one launch for all items."""
    n = len(items)
    descs = (ColsumDesc * n)()
    for i, (x, out) in enumerate(items):
        descs[i].X, descs[i].out = x.data_ptr(), out.data_ptr()
        descs[i].M, descs[i].N, descs[i].ldx = x.shape[0], x.shape[1], x.stride(0)
    L.check(L.lib().pa_colsum_many(C.cast(descs, C.c_void_p), n, L.dt(items[0][0]), L.stream()), "pa_colsum_many")


class ReduceDesc(C.Structure):          # mirrors pa_reduce_desc
    _fields_ = [("ws", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32),
                ("ld_out", C.c_int32), ("splitk", C.c_int32)]


def splitk_reduce_many(items):
    """items: [(ws, out[rows, cols] f32, splitk)]: out = sum of the splitk slabs in ws, one launch for all items."""
    n = len(items)
    descs = (ReduceDesc * n)()
    for i, (ws, out, sk) in enumerate(items):
        descs[i].ws, descs[i].out = ws.data_ptr(), out.data_ptr()
        descs[i].rows, descs[i].cols, descs[i].ld_out, descs[i].splitk = out.shape[0], out.shape[1], out.stride(0), sk
    L.check(L.lib().pa_splitk_reduce_many(C.cast(descs, C.c_void_p), n, L.stream()), "pa_splitk_reduce_many")


def colsum(x, out=None, accumulate=False):
    M, N = x.shape
    if out is None:
        out = torch.zeros(N, dtype=torch.float32, device=x.device)
    part = _f32(int(L.lib().pa_colsum_ws_floats(M, N)), device=x.device)
    L.check(L.lib().pa_colsum(L.ptr(x), L.dt(x), M, N, x.stride(0), L.ptr(out), int(accumulate), L.ptr(part),
                              L.stream()), "pa_colsum")
    return out


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def embed_input_fwd(tables, idx, dtype=torch.float32, rowmap=None, n_rows=None):
    n_tok = n_rows if n_rows is not None else next(i for i in idx if i is not None).numel()
    d = tables[0].shape[1]
    out = torch.empty(n_tok, d, dtype=dtype, device=tables[0].device)
    L.check(L.lib().pa_embed_input_fwd(L.ptr(out), L.dt(out), _ptr_array(tables), _ptr_array(idx), L.ptr(rowmap), len(tables),
                                       C.c_int64(n_tok), d, L.stream()), "pa_embed_input_fwd")
    return out


def pack_rows(mask):
    """mask: bool/uint8 [B, S] (True = PAD).  Returns (cu int32 [B+1], rowmap int32 [n_valid], n_valid); cu is a view of
    the [2B+1] buffer whose tail holds the batch elements by descending row count (``pack_order(cu)``)."""
    B, S = mask.shape
    m8 = mask.to(torch.uint8).contiguous()
    cu = torch.empty(2 * B + 1, dtype=torch.int32, device=mask.device)
    rowmap = torch.empty(B * S, dtype=torch.int32, device=mask.device)
    L.check(L.lib().pa_pack_rows(L.ptr(m8), B, S, L.ptr(cu), L.ptr(rowmap), L.stream()), "pa_pack_rows")
    n = int(cu[B])
    return cu[: B + 1], rowmap[:n], n


def pack_order(cu):
    """The dispatch order pa_pack_rows left behind cu (int32 [B]: batch elements, longest first)."""
    base = cu._base if cu._base is not None else cu           # the [B+1] view pack_rows returns, or the whole buffer
    if base.numel() % 2 == 0:
        raise ValueError("pack_order needs the [2B+1] buffer of pa_pack_rows (or a view of it)")
    B = (base.numel() - 1) // 2
    return base[B + 1: 2 * B + 1]


def embed_input_bwd(dout, dtables, idx, rowmap=None):
    n_tok, d = dout.shape
    rows = (C.c_int32 * len(dtables))(*[t.shape[0] for t in dtables])
    L.check(L.lib().pa_embed_input_bwd(L.ptr(dout), L.dt(dout), _ptr_array(dtables), _ptr_array(idx), L.ptr(rowmap), rows,
                                       len(dtables), C.c_int64(n_tok), d, L.stream()), "pa_embed_input_bwd")


def embed_output_fwd(value, coord, pos, tok, T, dof=6, dtype=torch.float32):
    B = tok.shape[0]
    d = value.shape[1]
    out = torch.empty(B, T, d, dtype=dtype, device=value.device)
    L.check(L.lib().pa_embed_output_fwd(L.ptr(out), L.dt(out), L.ptr(value), L.ptr(coord), L.ptr(pos), L.ptr(tok),
                                        tok.stride(0), B, T, d, dof, L.stream()), "pa_embed_output_fwd")
    return out


def embed_output_bwd(dout, dvalue, dcoord, dpos, tok, dof=6):
    B, T, d = dout.shape
    L.check(L.lib().pa_embed_output_bwd(L.ptr(dout), L.dt(dout), L.ptr(dvalue), L.ptr(dcoord), L.ptr(dpos),
                                        L.ptr(tok), tok.stride(0), B, T, d, dof, L.stream()), "pa_embed_output_bwd")


def gemm_ln(x, w, gamma, beta, eps, *, bias=None, residual=None, drop_p=0.0, drop_seed=0, want_z=True):
    """z = residual + drop(x @ w.T + bias), y = LayerNorm(z) in one launch (bf16, w: [512, K]).  Returns (z | None, y, mean, rstd)
    - bit-identical to gemm(..., bias, residual, drop_p) followed by layernorm_fwd."""
    M, K = x.shape
    assert w.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1
    y = torch.empty(M, w.shape[0], dtype=x.dtype, device=x.device)
    z = torch.empty_like(y) if want_z else None
    mean, rstd = _f32(M, device=x.device), _f32(M, device=x.device)
    g = L.GemmLnArgs()
    g.A, g.W, g.Y, g.Z = x.data_ptr(), w.data_ptr(), y.data_ptr(), (z.data_ptr() if want_z else None)
    g.bias = bias.data_ptr() if bias is not None else None
    g.R = residual.data_ptr() if residual is not None else None
    g.gamma, g.beta, g.mean, g.rstd = gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    g.M, g.N, g.K = M, w.shape[0], K
    g.lda, g.ldw, g.ldy = x.stride(0), w.stride(0), y.stride(0)
    g.ldz = z.stride(0) if want_z else 0
    g.ldr = residual.stride(0) if residual is not None else 0
    g.eps, g.drop_p, g.drop_seed = eps, drop_p, drop_seed
    L.check(L.lib().pa_gemm_ln(C.byref(g), L.stream()), "pa_gemm_ln")
    return z, y, mean, rstd


def gemm_norm_a(z, w, bias, gamma, beta, eps, *, relu=False, want_y=True, out_dtype=None, zf=None):
    """epi(LayerNorm(z) @ w.T + bias) with the LayerNorm folded into the product (pa_ln_fold_weights + pa_gemm_norm_a; the
    greedy-decode step's form).  z: bf16 [M, K]; w, bias, gamma, beta: f32 master parameters.  Returns (out, y | None).
    zf: the f32 rows of which z is the bf16 copy - statistics and y (then f32) come from them (f32 residual stream)."""
    M, K = z.shape
    N = w.shape[0]
    dev = z.device
    wf = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    u, v = _f32(N, device=dev), _f32(N, device=dev)
    L.check(L.lib().pa_ln_fold_weights(L.ptr(wf), L.ptr(u), L.ptr(v), L.ptr(w), L.ptr(bias), L.ptr(gamma), L.ptr(beta), N, K,
                                       L.stream()), "pa_ln_fold_weights")
    out = torch.empty(M, N, dtype=out_dtype or z.dtype, device=dev)
    y = torch.empty(M, K, dtype=(torch.float32 if zf is not None else z.dtype), device=dev) if want_y else None
    g = L.GemmArgs()
    g.A, g.B, g.C, g.bias = z.data_ptr(), wf.data_ptr(), out.data_ptr(), v.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = z.stride(0), K, N
    g.batch, g.a_kcontig, g.b_kcontig = 1, 1, 1
    g.in_dtype, g.out_dtype = L.dt(z), L.dt(out)
    g.alpha, g.relu, g.aux_scale, g.splitk = 1.0, int(relu), 1.0, 1
    x = L.GemmNormExt()
    x.u, x.gamma, x.beta = u.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    x.y, x.ldy, x.eps = (y.data_ptr() if want_y else None), K, eps
    if zf is not None:
        x.zf, x.ldzf, x.y_f32 = zf.data_ptr(), zf.stride(0), 1
    L.check(L.lib().pa_gemm_norm_a(C.byref(g), C.byref(x), L.stream()), "pa_gemm_norm_a")
    return out, y


def layernorm_fwd(z, gamma, beta, eps):
    rows, d = z.numel() // z.shape[-1], z.shape[-1]
    y = torch.empty_like(z)
    mean, rstd = _f32(rows, device=z.device), _f32(rows, device=z.device)
    L.check(L.lib().pa_layernorm_fwd(L.ptr(y), L.ptr(z), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd),
                                     C.c_int64(rows), d, C.c_float(eps), L.dt(z), L.stream()), "pa_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, z, gamma, mean, rstd, dgamma, dbeta, dzsum=None, drop_p=0.0, drop_seed=0):
    rows, d = z.numel() // z.shape[-1], z.shape[-1]
    dz = torch.empty_like(z)
    ddrop = torch.empty_like(z) if drop_p > 0 else None
    part = _f32(int(L.lib().pa_layernorm_ws_floats(rows, d)), device=z.device)
    L.check(L.lib().pa_layernorm_bwd(L.ptr(dz), L.ptr(ddrop), L.ptr(dy), L.ptr(z), L.ptr(gamma), L.ptr(mean),
                                     L.ptr(rstd), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dzsum), L.ptr(part),
                                     C.c_int64(rows), d, L.dt(z), C.c_float(drop_p), C.c_uint32(drop_seed),
                                     L.stream()), "pa_layernorm_bwd")
    return dz, ddrop


def _attn_args(q, k, v, o, lse, kpm, causal, scale, drop_p, drop_seed, H, cu_q=None, cu_k=None, B=None, Lq=None, Lk=None, order=None):
    if B is None:
        B, Lq = q.shape[0], q.shape[1]
        Lk = k.shape[1]
    dh = q.shape[-1] // H
    a = L.AttnArgs()
    a.q, a.k, a.v, a.o, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
    a.kpm = kpm.data_ptr() if kpm is not None else None
    a.B, a.H, a.Lq, a.Lk, a.dh = B, H, Lq, Lk, dh
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(-2), k.stride(-2), v.stride(-2), o.stride(-2)
    a.cu_q = cu_q.data_ptr() if cu_q is not None else None
    a.cu_k = cu_k.data_ptr() if cu_k is not None else None
    a.order = order.data_ptr() if order is not None else None        # int32 [B] dispatch order (pa_pack_rows)
    a.causal = int(causal)
    a.scale = scale if scale is not None else 1.0 / math.sqrt(dh)
    a.drop_p, a.drop_seed = drop_p, drop_seed
    a.dtype = L.dt(q)
    return a


def mask_order(kpm):
    """Dispatch order for a padded batch with a key-padding mask: int32 [B], batch elements by descending number of unmasked
    keys (what pa_pack_rows leaves behind `cu` for packed batches).  A block's duration is proportional to its element's key
    count (tiles past the last unmasked key are skipped) and the hardware hands out blocks in index order as slots free up, so
    longest-first is list scheduling's LPT rule: the launch no longer ends with a long element that started late.  Computed
    once per batch (the mask is the same for every layer, forward and backward); results do not depend on it."""
    return torch.argsort((kpm == 0).sum(dim=1), descending=True, stable=True).to(torch.int32)


def attn_fwd(q, k, v, H, kpm=None, causal=False, scale=None, drop_p=0.0, drop_seed=0, order=None):
    """q: [B, Lq, H*dh] (may be a strided view of a packed projection), k/v: [B, Lk, H*dh];
    kpm: uint8/bool [B, Lk] (1 = PAD); order: optional int32 [B] dispatch order (mask_order).
    Returns (o [B, Lq, H*dh], lse [B, H, Lq])."""
    B, Lq, dm = q.shape
    o = torch.empty(B, Lq, dm, dtype=q.dtype, device=q.device)
    lse = _f32(B, H, Lq, device=q.device)
    if kpm is not None:
        kpm = kpm.to(torch.uint8).contiguous()
    a = _attn_args(q, k, v, o, lse, kpm, causal, scale, drop_p, drop_seed, H, order=order)
    L.check(L.lib().pa_attn_fwd(C.byref(a), L.stream()), "pa_attn_fwd")
    return o, lse


def attn_bwd(dout, q, k, v, o, lse, H, kpm=None, causal=False, scale=None, drop_p=0.0, drop_seed=0, order=None):
    if kpm is not None:
        kpm = kpm.to(torch.uint8).contiguous()
    a = _attn_args(q, k, v, o, lse, kpm, causal, scale, drop_p, drop_seed, H, order=order)
    dq = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    dk = torch.empty(k.shape, dtype=k.dtype, device=k.device)
    dv = torch.empty(v.shape, dtype=v.dtype, device=v.device)
    delta = _f32(lse.shape, device=q.device)
    a.dout, a.dq, a.dk, a.dv, a.delta = dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr()
    a.lddo, a.lddq, a.lddk, a.lddv = dout.stride(1), dq.stride(1), dk.stride(1), dv.stride(1)
    L.check(L.lib().pa_attn_bwd(C.byref(a), L.stream()), "pa_attn_bwd")
    return dq, dk, dv


def pack_lengths(lengths, device):
    """int lengths [B] -> (cu int32 [B+1] on `device`, order int32 [B] = batch elements by descending length) - the
    layout pa_pack_rows leaves behind (cu[0..B], cu[B+1..2B])."""
    ln = torch.as_tensor(lengths, dtype=torch.int64)
    cu = torch.zeros(len(ln) + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(ln, 0).to(torch.int32)
    order = torch.argsort(ln, descending=True, stable=True).to(torch.int32)
    return cu.to(device), order.to(device)


def switch_fwd(h, w, b):
    rows, d = h.numel() // h.shape[-1], h.shape[-1]
    s = _f32(rows, device=h.device)
    L.check(L.lib().pa_switch_fwd(L.ptr(s), L.ptr(h), L.dt(h), L.ptr(w), L.ptr(b), C.c_int64(rows), d, L.stream()),
            "pa_switch_fwd")
    return s


def switch_bwd(ds, h, w, dw, db, dh=None):
    rows, d = h.numel() // h.shape[-1], h.shape[-1]
    acc = dh is not None
    if dh is None:
        dh = torch.empty_like(h)
    # one [2][d] partial per backward block; the block height is the library's (it changed from 32 to 8 rows once
    # and a hard-coded 32 here wrote past this buffer), so ask it
    part = _f32(int(L.lib().pa_layernorm_bwd_nparts(C.c_int64(rows))) * 2 * d, device=h.device)
    L.check(L.lib().pa_switch_bwd(L.ptr(dh), int(acc), L.ptr(dw), L.ptr(db), L.ptr(ds), L.ptr(h), L.dt(h), L.ptr(w),
                                  L.ptr(part), C.c_int64(rows), d, L.stream()), "pa_switch_bwd")
    return dh


def mixture_nll_fwd(vocab, ptr, sw, label, V, pad):
    """vocab [B,T,ldv] f32, ptr [B,T,T] f32 (scaled), sw [B*T] f32, label int64 [B,T].
    Returns stats (sum_nll, n_valid, n_correct) and per-row lse."""
    B, T = label.shape
    stats = torch.zeros(4, dtype=torch.float32, device=vocab.device)
    row_lse = _f32(B * T, 2, device=vocab.device)
    L.check(L.lib().pa_mixture_nll_fwd(L.ptr(stats), L.ptr(row_lse), L.ptr(vocab), vocab.stride(-2), L.ptr(ptr),
                                       L.ptr(sw), L.ptr(label), B, T, V, pad, L.stream()), "pa_mixture_nll_fwd")
    return stats, row_lse


def mixture_nll_bwd(stats, row_lse, vocab, ptr, sw, label, V, pad, gscale=1.0, out_dtype=torch.float32):
    B, T = label.shape
    dvocab = torch.empty(vocab.shape, dtype=out_dtype, device=vocab.device)
    dptr = torch.empty(ptr.shape, dtype=out_dtype, device=vocab.device)
    dsw = torch.empty_like(sw)
    L.check(L.lib().pa_mixture_nll_bwd(L.ptr(dvocab), L.ptr(dptr), L.dt(dvocab), L.ptr(dsw), L.ptr(stats), L.ptr(row_lse),
                                       L.ptr(vocab), vocab.stride(-2), L.ptr(ptr), L.ptr(sw), L.ptr(label), B, T, V,
                                       pad, C.c_float(gscale), L.stream()), "pa_mixture_nll_bwd")
    return dvocab, dptr, dsw


def adam_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, gscale=1.0, p_bf16=None):
    L.check(L.lib().pa_adam_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(p_bf16), C.c_int64(p.numel()),
                                 C.c_float(lr), C.c_float(b1), C.c_float(b2), C.c_float(eps), int(step),
                                 C.c_float(gscale), L.stream()), "pa_adam_step")


def cast(src, dtype):
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    L.check(L.lib().pa_cast(L.ptr(dst), L.dt(dst), L.ptr(src), L.dt(src), C.c_int64(src.numel()), L.stream()),
            "pa_cast")
    return dst


def attn_split_ws(rows_total, B, H, device, L_max=1 << 20):
    """Scratch for the range blocks of packed self-attention launches (include/plank_hip.h pa_attn_args.ws), zeroed as its
    contract asks, or None when the library would not use one."""
    n = int(L.lib().pa_attn_ws_bytes(int(rows_total), int(B), int(H), int(L_max)))
    return torch.zeros(n, dtype=torch.uint8, device=device) if n > 0 else None


def attn_varlen_fwd(q, k, v, H, cu_q, cu_k, B, Lq_max, Lk_max, causal=False, scale=None, drop_p=0.0, drop_seed=0, order=None, ws=None):
    """Packed ("unpadded") attention: q [Nq, H*dh], k/v [Nk, H*dh] with int32 row offsets cu_q / cu_k [B+1]
    (either may be None = dense [B*L] rows).  Returns (o [Nq, H*dh], lse [B, H, Lq_max]).  ``ws``: attn_split_ws()."""
    o = torch.empty(q.shape[0], q.shape[1], dtype=q.dtype, device=q.device)
    lse = _f32(B, H, Lq_max, device=q.device)
    a = _attn_args(q, k, v, o, lse, None, causal, scale, drop_p, drop_seed, H, cu_q, cu_k, B, Lq_max, Lk_max, order)
    if ws is not None:
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    L.check(L.lib().pa_attn_fwd(C.byref(a), L.stream()), "pa_attn_fwd")
    return o, lse


def attn_varlen_bwd(dout, q, k, v, o, lse, H, cu_q, cu_k, B, Lq_max, Lk_max, causal=False, scale=None, drop_p=0.0, drop_seed=0, order=None, ws=None):
    a = _attn_args(q, k, v, o, lse, None, causal, scale, drop_p, drop_seed, H, cu_q, cu_k, B, Lq_max, Lk_max, order)
    if ws is not None:
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    dq = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    dk = torch.empty(k.shape, dtype=k.dtype, device=k.device)
    dv = torch.empty(v.shape, dtype=v.dtype, device=v.device)
    delta = _f32(lse.shape, device=q.device)
    a.dout, a.dq, a.dk, a.dv, a.delta = dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr()
    a.lddo, a.lddq, a.lddk, a.lddv = dout.stride(-2), dq.stride(-2), dk.stride(-2), dv.stride(-2)
    L.check(L.lib().pa_attn_bwd(C.byref(a), L.stream()), "pa_attn_bwd")
    return dq, dk, dv


def dec_cross_mq(qt, mem, *, kpm=None, cu=None, S=None, ws=None):
    """Absorbed ("multi-query") cross-attention of one decode step (csrc/decode_mq.h, pa_dec_cross_mq): qt [B, H, 512] bf16 (already
    carrying scale * log2 e), mem dense [B, S, 512] (optional kpm [B, S] uint8, 1 = PAD) or packed [rows, 512] with cu int32 [B + 1].
    Returns ctx [B, H, 512] bf16: per head the softmax-weighted sum of the element's memory rows."""
    B, H, d = qt.shape
    if cu is None:
        S = mem.shape[1]
    if qt.dtype == torch.float32:                      # the exact-f32 form (pa_dec_cross_mq32)
        ctx = torch.empty(B, H, d, dtype=torch.float32, device=qt.device)
        if ws is not None:
            L.check(L.lib().pa_dec_cross_mq32_ws(L.ptr(ctx), L.ptr(qt), L.ptr(mem), L.ptr(kpm), L.ptr(cu), B, int(S), H, d, L.ptr(ws),
                                                 C.c_int64(ws.numel()), L.stream()), "pa_dec_cross_mq32_ws")
            return ctx
        L.check(L.lib().pa_dec_cross_mq32(L.ptr(ctx), L.ptr(qt), L.ptr(mem), L.ptr(kpm), L.ptr(cu), B, int(S), H, d, L.stream()),
                "pa_dec_cross_mq32")
        return ctx
    ctx = torch.empty(B, H, d, dtype=torch.bfloat16, device=qt.device)
    if ws is not None:                                 # range blocks (pa_dec_cross_mq_ws); ws = dec_cross_mq_ws(B, S, device)
        L.check(L.lib().pa_dec_cross_mq_ws(L.ptr(ctx), L.ptr(qt), L.ptr(mem), L.ptr(kpm), L.ptr(cu), B, int(S), H, d, L.ptr(ws),
                                           C.c_int64(ws.numel()), L.stream()), "pa_dec_cross_mq_ws")
        return ctx
    L.check(L.lib().pa_dec_cross_mq(L.ptr(ctx), L.ptr(qt), L.ptr(mem), L.ptr(kpm), L.ptr(cu), B, int(S), H, d, L.stream()),
            "pa_dec_cross_mq")
    return ctx


def dec_cross_mq_ws(B, S, device):
    """Zeroed scratch for dec_cross_mq(..., ws=): None when the library would launch one block per element anyway."""
    n = int(L.lib().pa_dec_cross_mq_ws_bytes(int(B), int(S)))
    return torch.zeros(n, dtype=torch.uint8, device=device) if n > 0 else None


def dec_self_mq32(qt, xcache, t_dev):
    """Self-attention form of the absorbed decode attention (pa_dec_self_mq32): qt [B, H, 512] f32, xcache [B, Tmax, 512] f32, t_dev int32
    device scalar (element b attends over rows 0 .. t).  Returns ctx [B, H, 512] f32."""
    B, H, d = qt.shape
    ctx = torch.empty(B, H, d, dtype=torch.float32, device=qt.device)
    L.check(L.lib().pa_dec_self_mq32(L.ptr(ctx), L.ptr(qt), L.ptr(xcache), L.ptr(t_dev), B, xcache.shape[1], H, d, L.stream()),
            "pa_dec_self_mq32")
    return ctx
