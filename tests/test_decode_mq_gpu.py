"""Absorbed ("multi-query") cross-attention of the decode step (csrc/decode_mq.h) at the C ABI (pa_dec_cross_mq) against float64:
dense rows with and without a key-padding mask, packed rows, ragged / tiny / tile-edge lengths, fewer than 8 heads, and the
algebra it rests on - attention over the memory rows with q~ = W_k^T q equals the reference's attention over K = W_k m + b_k,
V = W_v m + b_v (plankassembly/models.py:284-307 -> nn.MultiheadAttention)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

LN2 = math.log(2.0)


def _ref(qt, mem_rows, valid):
    """qt [H, d] (log2 domain), mem_rows [L, d], valid [L] bool -> ctx [H, d] (float64)."""
    x = qt.double() @ mem_rows.double().T * LN2
    x = x.masked_fill(~valid[None, :], float("-inf"))
    p = torch.softmax(x, dim=-1)
    return p @ mem_rows.double()


def _check(ctx, ref, tol=2e-2):
    err = (ctx.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * max(scale, 1e-3), (err, scale)


@pytest.mark.parametrize("B,S,H", [(3, 16, 8), (2, 48, 8), (4, 200, 8), (2, 1024, 8), (3, 130, 4), (2, 17, 1), (5, 7, 8)])
def test_dense_rows_no_mask(B, S, H):
    from plankassembly_amd import ops
    torch.manual_seed(B * 1000 + S + H)
    mem = torch.randn(B, S, 512, device="cuda").bfloat16()
    qt = (torch.randn(B, H, 512, device="cuda") * 0.15).bfloat16()
    ctx = ops.dec_cross_mq(qt, mem)
    torch.cuda.synchronize()
    for b in range(B):
        _check(ctx[b], _ref(qt[b], mem[b], torch.ones(S, dtype=torch.bool, device="cuda")))


@pytest.mark.parametrize("B,S", [(4, 64), (3, 299), (2, 1024), (3, 33)])
def test_dense_rows_with_key_padding_mask(B, S):
    from plankassembly_amd import ops
    torch.manual_seed(S)
    mem = torch.randn(B, S, 512, device="cuda").bfloat16()
    qt = (torch.randn(B, 8, 512, device="cuda") * 0.15).bfloat16()
    lens = torch.randint(1, S + 1, (B,))
    lens[0] = S
    kpm = (torch.arange(S)[None, :] >= lens[:, None]).to(torch.uint8).cuda()
    kpm[1, 0] = 1                                            # a hole in front of a valid key: the mask is per key, not a length
    ctx = ops.dec_cross_mq(qt, mem, kpm=kpm)
    torch.cuda.synchronize()
    for b in range(B):
        _check(ctx[b], _ref(qt[b], mem[b], kpm[b] == 0))


def test_all_keys_masked_gives_zeros():
    from plankassembly_amd import ops
    mem = torch.randn(2, 40, 512, device="cuda").bfloat16()
    qt = torch.randn(2, 8, 512, device="cuda").bfloat16()
    kpm = torch.zeros(2, 40, dtype=torch.uint8, device="cuda")
    kpm[1] = 1
    ctx = ops.dec_cross_mq(qt, mem, kpm=kpm)
    assert torch.isfinite(ctx.float()).all() and float(ctx[1].float().abs().max()) == 0.0


@pytest.mark.parametrize("lens", [[16, 1, 1024, 333], [5, 0, 129], [1024] * 3, [112, 113, 127, 128, 96]])
def test_packed_rows(lens):
    from plankassembly_amd import ops
    torch.manual_seed(sum(lens))
    B = len(lens)
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.tensor(lens).cumsum(0)
    rows = int(cu[-1])
    mem = torch.randn(max(rows, 1), 512, device="cuda").bfloat16()
    qt = (torch.randn(B, 8, 512, device="cuda") * 0.15).bfloat16()
    ctx = ops.dec_cross_mq(qt, mem, cu=cu.cuda(), S=max(lens))
    torch.cuda.synchronize()
    for b in range(B):
        if lens[b] == 0:
            assert float(ctx[b].float().abs().max()) == 0.0
            continue
        m = mem[int(cu[b]):int(cu[b + 1])]
        _check(ctx[b], _ref(qt[b], m, torch.ones(lens[b], dtype=torch.bool, device="cuda")))


@pytest.mark.parametrize("B,S,mode", [(16, 1024, "dense"), (16, 1024, "mask"), (3, 1199, "mask"), (1, 1024, "dense"), (40, 640, "dense"),
                                      (5, 1024, "packed"), (16, 299, "packed"), (2, 130, "dense")])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_range_blocks_equal_one_block_per_element(B, S, mode, dtype):
    """pa_dec_cross_mq_ws (csrc/decode_mq.h range blocks: several blocks per batch element, partial (O, m, l) merged by the last to
    arrive): against float64 and against the one-block launch; twice on one scratch buffer; every ticket word zero afterwards.  Packed
    lengths include elements shorter than one range (empty range blocks) and the ramp case makes the ranges' reference points differ
    by tens of log2 units."""
    from plankassembly_amd import ops
    from plankassembly_amd import _lib as L
    torch.manual_seed(B * 7 + S)
    kpm = cu = None
    if mode == "packed":
        lens = [S, 1, 17, S // 2, 333][:B] if B <= 5 else [max(1, (S * (i + 1)) // B) for i in range(B)]
        cu_h = torch.zeros(B + 1, dtype=torch.int32); cu_h[1:] = torch.tensor(lens).cumsum(0)
        mem = torch.randn(int(cu_h[-1]), 512, device="cuda")
        cu = cu_h.cuda()
    else:
        lens = [S] * B
        mem = torch.randn(B, S, 512, device="cuda")
        if mode == "mask":
            ln = torch.randint(1, S + 1, (B,)); ln[0] = S
            kpm = (torch.arange(S)[None, :] >= ln[:, None]).to(torch.uint8).cuda()
            kpm[1 % B, 0] = 1
    q = torch.randn(B, 8, 512, device="cuda")
    q = q / q.norm(dim=-1, keepdim=True)
    if mode == "dense":                                      # scores of element 0 grow along the keys: later ranges dominate
        ramp = torch.linspace(-40, 40, S, device="cuda")
        mem[0] = mem[0] * 0.05 + ramp[:, None] * q[0, 0][None, :]
    mem = mem.to(dtype)
    qt = (q * (1.0 if mode == "dense" else 3.0)).to(dtype)
    ws = ops.dec_cross_mq_ws(B, S, "cuda")
    if B <= 64 and S >= 256:
        assert ws is not None
    one = ops.dec_cross_mq(qt, mem, kpm=kpm, cu=cu, S=S)
    for rep in range(2):
        ctx = ops.dec_cross_mq(qt, mem, kpm=kpm, cu=cu, S=S, ws=ws)
        torch.cuda.synchronize()
        for b in range(B):
            if mode == "packed":
                m = mem[int(cu[b]):int(cu[b + 1])]; valid = torch.ones(lens[b], dtype=torch.bool, device="cuda")
            else:
                m = mem[b]; valid = (kpm[b] == 0) if kpm is not None else torch.ones(S, dtype=torch.bool, device="cuda")
            _check(ctx[b], _ref(qt[b], m, valid), tol=2e-2 if dtype == torch.bfloat16 else 2e-5)
        assert float((ctx.float() - one.float()).abs().max()) <= (2e-2 if dtype == torch.bfloat16 else 2e-5) * float(one.float().abs().max())
        if rep == 1:
            assert torch.equal(ctx, first)                     # merged in range order: run-to-run identical
        first = ctx
        if ws is not None:
            assert int(ws[:(B * 4 + 255) // 256 * 256].view(torch.int32).abs().sum()) == 0


def test_large_scores_and_moving_reference_point():
    """Scores spanning +-60 in the log2 domain with the largest key LAST: every tile moves the running reference point (the
    accumulator rescale path) and the early tiles' probabilities underflow - the result must still be the exact softmax."""
    from plankassembly_amd import ops
    torch.manual_seed(3)
    S = 256
    mem = torch.randn(1, S, 512, device="cuda").bfloat16()
    q = torch.randn(8, 512, device="cuda")
    q = q / q.norm(dim=-1, keepdim=True)
    # the score of key s grows with s: add a multiple of q to the rows
    ramp = torch.linspace(-60, 60, S, device="cuda")
    mem = (mem.float() * 0.05 + ramp[None, :, None] * q[0][None, None, :]).bfloat16()
    qt = q[None].bfloat16()
    ctx = ops.dec_cross_mq(qt, mem)
    torch.cuda.synchronize()
    _check(ctx[0], _ref(qt[0], mem[0], torch.ones(S, dtype=torch.bool, device="cuda")))


def test_absorbed_form_equals_attention_over_projected_keys_and_values():
    """float64 algebra + the kernel: softmax(q_h . (W_k m + b_k)_h / sqrt(dh)) (W_v m + b_v)_h == W_v,h ctx_h + b_v,h with
    ctx from pa_dec_cross_mq on q~_h = scale log2e W_k,h^T q_h."""
    from plankassembly_amd import ops
    torch.manual_seed(11)
    B, S, H, d = 3, 150, 8, 512
    dh = d // H
    mem = torch.randn(B, S, d, device="cuda").bfloat16()
    q = torch.randn(B, d, device="cuda", dtype=torch.float64) * 0.5
    Wk = torch.randn(d, d, device="cuda", dtype=torch.float64) / math.sqrt(d)
    Wv = torch.randn(d, d, device="cuda", dtype=torch.float64) / math.sqrt(d)
    bk = torch.randn(d, device="cuda", dtype=torch.float64)
    bv = torch.randn(d, device="cuda", dtype=torch.float64)
    m64 = mem.double()
    K = m64 @ Wk.T + bk
    V = m64 @ Wv.T + bv
    qh = q.view(B, H, dh)
    Kh = K.view(B, S, H, dh).permute(0, 2, 1, 3)
    Vh = V.view(B, S, H, dh).permute(0, 2, 1, 3)
    att = torch.softmax(torch.einsum("bhc,bhsc->bhs", qh, Kh) / math.sqrt(dh), dim=-1)
    want = torch.einsum("bhs,bhsc->bhc", att, Vh).reshape(B, d)
    sl = 1.0 / math.sqrt(dh) / LN2
    qt = torch.einsum("bhc,bhcj->bhj", qh, Wk.view(H, dh, d)[None].expand(B, H, dh, d)) * sl
    # (i) the algebra, in float64: attention over the memory rows with q~, W_v applied behind the softmax
    p64 = torch.softmax(torch.einsum("bhj,bsj->bhs", qt, m64) * LN2, dim=-1)
    ctx64 = torch.einsum("bhs,bsj->bhj", p64, m64)
    alg = torch.einsum("bhj,hcj->bhc", ctx64, Wv.view(H, dh, d)).reshape(B, d) + bv
    assert (alg - want).abs().max().item() <= 1e-9 * max(1.0, want.abs().max().item())
    # (ii) the kernel on the bf16-rounded q~ against float64 on the same rounded q~
    qtb = qt.bfloat16()
    ctx = ops.dec_cross_mq(qtb, mem).double()
    pr = torch.softmax(torch.einsum("bhj,bsj->bhs", qtb.double(), m64) * LN2, dim=-1)
    ctxr = torch.einsum("bhs,bsj->bhj", pr, m64)
    assert (ctx - ctxr).abs().max().item() <= 2e-2 * ctxr.abs().max().item()


# ---- exact-f32 form (pa_dec_cross_mq32): the cross-attention of the token-exact decode ------------------------------------------------
def _check32(ctx, ref):
    err = (ctx.double() - ref).abs().max().item()
    assert err <= 2e-5 * max(ref.abs().max().item(), 1e-3), err


@pytest.mark.parametrize("B,S,H", [(3, 16, 8), (2, 48, 8), (4, 200, 8), (2, 1024, 8), (3, 130, 4), (2, 17, 1), (5, 7, 8), (2, 33, 8)])
def test_f32_dense_rows_no_mask(B, S, H):
    from plankassembly_amd import ops
    torch.manual_seed(B * 1000 + S + H)
    mem = torch.randn(B, S, 512, device="cuda")
    qt = torch.randn(B, H, 512, device="cuda") * 0.15
    ctx = ops.dec_cross_mq(qt, mem)
    torch.cuda.synchronize()
    for b in range(B):
        _check32(ctx[b], _ref(qt[b], mem[b], torch.ones(S, dtype=torch.bool, device="cuda")))


@pytest.mark.parametrize("B,S", [(4, 64), (3, 299), (2, 1024), (3, 33)])
def test_f32_dense_rows_with_key_padding_mask(B, S):
    from plankassembly_amd import ops
    torch.manual_seed(S)
    mem = torch.randn(B, S, 512, device="cuda")
    qt = torch.randn(B, 8, 512, device="cuda") * 0.15
    lens = torch.randint(1, S + 1, (B,))
    lens[0] = S
    kpm = (torch.arange(S)[None, :] >= lens[:, None]).to(torch.uint8).cuda()
    kpm[1, 0] = 1
    ctx = ops.dec_cross_mq(qt, mem, kpm=kpm)
    torch.cuda.synchronize()
    for b in range(B):
        _check32(ctx[b], _ref(qt[b], mem[b], kpm[b] == 0))


@pytest.mark.parametrize("lens", [[16, 1, 1024, 333], [5, 0, 129], [112, 113, 127, 128, 96], [31, 32, 33, 47, 48, 49, 63, 64, 65]])
def test_f32_packed_rows(lens):
    from plankassembly_amd import ops
    torch.manual_seed(sum(lens))
    B = len(lens)
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.tensor(lens).cumsum(0)
    mem = torch.randn(max(int(cu[-1]), 1), 512, device="cuda")
    qt = torch.randn(B, 8, 512, device="cuda") * 0.15
    ctx = ops.dec_cross_mq(qt, mem, cu=cu.cuda(), S=max(lens))
    torch.cuda.synchronize()
    for b in range(B):
        if lens[b] == 0:
            assert float(ctx[b].abs().max()) == 0.0
            continue
        m = mem[int(cu[b]):int(cu[b + 1])]
        _check32(ctx[b], _ref(qt[b], m, torch.ones(lens[b], dtype=torch.bool, device="cuda")))


def test_f32_moving_reference_point():
    from plankassembly_amd import ops
    torch.manual_seed(3)
    S = 256
    q = torch.randn(8, 512, device="cuda")
    q = q / q.norm(dim=-1, keepdim=True)
    ramp = torch.linspace(-60, 60, S, device="cuda")
    mem = (torch.randn(1, S, 512, device="cuda") * 0.05 + ramp[None, :, None] * q[0][None, None, :]).contiguous()
    qt = q[None].contiguous()
    ctx = ops.dec_cross_mq(qt, mem)
    torch.cuda.synchronize()
    _check32(ctx[0], _ref(qt[0], mem[0], torch.ones(S, dtype=torch.bool, device="cuda")))


@pytest.mark.parametrize("B,Tmax,t", [(3, 128, 0), (2, 128, 15), (4, 256, 16), (2, 1024, 1023), (5, 64, 40)])
def test_f32_self_attention_form_reads_its_length_from_the_device(B, Tmax, t):
    """pa_dec_self_mq32: rows 0 .. t of a [B][Tmax] cache, t in device memory (rows behind t hold NaN: they must not be touched)."""
    from plankassembly_amd import ops
    torch.manual_seed(Tmax + t)
    x = torch.randn(B, Tmax, 512, device="cuda")
    x[:, t + 1:] = float("nan")                                    # (rows past the end are never fetched: the last tile re-reads row t)
    qt = torch.randn(B, 8, 512, device="cuda") * 0.15
    td = torch.tensor([t, 0], dtype=torch.int32, device="cuda")
    ctx = ops.dec_self_mq32(qt, x, td)
    torch.cuda.synchronize()
    assert torch.isfinite(ctx).all()
    for b in range(B):
        _check32(ctx[b], _ref(qt[b], x[b, :t + 1], torch.ones(t + 1, dtype=torch.bool, device="cuda")))


def test_kv_cache_form_of_the_decode_step_still_passes_the_token_exact_gate():
    """The per-layer K / V-cache form of the cross-attention (PLANK_DECODE_MQ=0 / PLANK_DECODE_MQ_F32=0, read once per process) stays a
    tested path: the headline-shape f32 decode test in a child process with both switches off (in this process the same test runs on
    the absorbed form - both are token-exact against the reference, i.e. they produce identical tokens)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PLANK_DECODE_MQ="0", PLANK_DECODE_MQ_F32="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "tests/test_headline_gpu.py", "-k",
                        "test_f32_greedy_decode_token_exact_at_headline_shape or test_f32_greedy_decode_batch8_vs_oracle_and_bf16_agreement"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
