#!/usr/bin/env python
"""Benchmark of the PlankAssembly hot path on MI355X (contract: see the task description).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full training step (forward + backward + gradient all-reduce + Adam) of the
reference's headline configuration -- train_complete.yaml model (d_model 512, 8 heads, dff 1024, 6+6
post-norm layers, dropout 0.2) at seq = 1024 (MAX_INPUT_LENGTH 1025), decoder length 128, batch 16 per
GPU, bf16 compute / f32 master weights -- on synthetic tokenised-drawing batches that are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line: `value` = train samples/s over all GPUs;
the greedy-decode throughput (batch 256, max_len 1024, hipGraph-replayed step), the attention/GEMM
roofline fractions (HIP-event timed) and the CPU baseline (the oracle on the host cores) ride in
the same line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
import types

# the host driver of this pool only supports dmabuf IPC: RCCL's peer buffers need this before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TOKEN = types.SimpleNamespace(END=512, PAD=513)
T_START = time.perf_counter()
D, H, FF, NE, ND, V = 512, 8, 1024, 6, 6, 514
S_IN, T_OUT, B_TRAIN = 1024, 128, 16
B_DEC, T_DEC = 256, 1024
# Workloads (BASELINE.json `configs`).  The default - what the driver records - is the headline line.
BIG = dict(d=512, h=8, ff=1024, ne=6, nd=6)
CONFIGS = {
    "headline": dict(BIG, max_in=1025, max_out=128, batch=16, spec="headline",
                     name="train_complete.yaml model d_model=512 H=8 dff=1024 6+6 post-norm layers, dropout 0.2"),
    "complete": dict(BIG, max_in=1200, max_out=128, batch=16, spec="complete", name="train_complete.yaml as shipped (C2)"),
    "visible": dict(BIG, max_in=1000, max_out=128, batch=16, spec="visible", name="train_visible.yaml lengths (C4)"),
    "sideface": dict(BIG, max_in=300, max_out=128, batch=64, spec="sideface",
                     name="train_sideface.yaml lengths, no input_type, empty rows (C4)"),
    "t1024": dict(BIG, max_in=1025, max_out=1024, batch=16, spec="headline", planks=(2, 170),
                  name="headline encoder with decoder length 1024 (SURVEY 8d T=1024 variant)"),
    "tiny": dict(d=128, h=8, ff=256, ne=2, nd=2, max_in=1200, max_out=128, batch=4, spec="complete",
                 name="BASELINE config 0: d_model=128, 2+2 layers, batch 4 (the reference's CPU-runnable case, C1)"),
}
PEAK_BF16_TFLOPS = 2500.0        # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def cfg_spec(c):
    from plankassembly_amd.data import spec_for
    sp = spec_for(c["spec"], max_input_length=c["max_in"], max_output_length=c["max_out"])
    if "planks" in c:
        sp.n_planks = c["planks"]
    return sp


def fwd_flops_per_sample(S, T, d=D, ff=FF, ne=NE, nd=ND, v=V):
    """2*MAC, forward."""
    enc = 8 * S * d * d + 4 * S * S * d + 4 * S * d * ff
    dec = 12 * T * d * d + 4 * S * d * d + 4 * T * T * d + 4 * T * S * d + 4 * T * d * ff
    heads = 2 * T * d * v + 2 * T * d * d + 2 * T * T * d + 2 * T * d
    return ne * enc + nd * dec + heads


# Random-init weights collapse under greedy decoding (every row emits one repeated token: attention averages wash the
# token-dependent part of the residual stream out).  The decode benchmark therefore rescales the random
# init per parameter group - larger embeddings / heads, damped attention / FFN output projections - exactly as the parity
# fixtures do (tests/large_cases.py): the timed work is identical, but the 256 x 1024 tokens are diverse, pointers fire at
# late steps, and comparing a slice of them with the CPU oracle (`decode.*.token_exact`) means something.
DECODE_GAINS = {"input_embeddings.": 8.0, "query_": 16.0, "vocab_head.weight": 4.0, "pointer_head.weight": 120.0,
                "switch_head.weight": 1.0, "out_proj.weight": 0.15, "multihead_attn.out_proj.weight": 3.0, "linear2.weight": 0.5}


def apply_gains(model, gains):
    with torch.no_grad():
        for k, p in model.named_parameters():
            g = 1.0
            for pat, f in gains.items():
                if pat in k:
                    g *= f
            if g != 1.0:
                p.mul_(g)
    model.invalidate_shadow()
    return model


def build(compute_dtype, max_in, max_out, dropout, c=BIG):
    from plankassembly_amd.models import PlankModel
    torch.manual_seed(2022)
    m = PlankModel(c["d"], c["h"], c["ff"], dropout, "relu", True, c["ne"], c["nd"], 3, 2, 4, 6, max_in, max_out, V, TOKEN,
                   compute_dtype=compute_dtype)
    return m.cuda()


def time_kernel(fn, iters=20, warm=3):
    """Average duration (s) of `fn` (which enqueues on torch's current stream) using HIP events on that stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def decode_cross_roofline(ddtype):
    """The decode step's dominant launch since round 5 - the absorbed cross-attention (csrc/decode_mq.h, pa_dec_cross_mq / _mq32) - timed
    under HIP events at the benchmark's shape over rotating memories (so a launch does not find its rows in the Infinity Cache).
    HBM-bound: algorithmic bytes = the memory rows once + query / context rows."""
    from plankassembly_amd import ops
    dt = torch.bfloat16 if ddtype == "bf16" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(3)
    mems = [torch.randn(B_DEC, S_IN, D, device="cuda", generator=g).to(dt) for _ in range(4)]
    qt = (torch.randn(B_DEC, H, D, device="cuda", generator=g) * 0.1).to(dt)
    i = [0]

    def run():
        i[0] = (i[0] + 1) % len(mems)
        ops.dec_cross_mq(qt, mems[i[0]])
    t = time_kernel(run, iters=24, warm=4)
    esz = mems[0].element_size()
    nbytes = (B_DEC * S_IN * D + 2 * B_DEC * H * D) * esz
    del mems
    torch.cuda.empty_cache()
    return {"kernel": "dec_cross_mq_kernel" if ddtype == "bf16" else "dec_cross_mq32_kernel", "bound": "hbm", "avg_launch_us": t * 1e6,
            "algorithmic_bytes_per_launch": nbytes, "achieved": nbytes / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": nbytes / t / 1e9 / PEAK_HBM_GBS, "launches_per_step": ND,
            "replaces": "dec_attn_kernel over the per-layer K / V caches: %.0f MB per launch" % (2 * B_DEC * S_IN * D * esz / 1e6)}


def kernel_rooflines(B):
    """HIP-event timing of the hot kernels at the workload's shapes (random bf16 data)."""
    from plankassembly_amd import ops
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)

    def rnd(*s):
        return torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)

    out = {}
    qkv = rnd(B, S_IN, 3 * D)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    valid = torch.randint(S_IN // 2, S_IN + 1, (B,), device=dev, generator=g)
    kpm = torch.arange(S_IN, device=dev)[None] >= valid[:, None]
    o, lse = ops.attn_fwd(q, k, v, H, kpm=kpm, drop_p=0.2, drop_seed=1)
    t = time_kernel(lambda: ops.attn_fwd(q, k, v, H, kpm=kpm, drop_p=0.2, drop_seed=1))
    fl = 4.0 * S_IN * S_IN * D * B
    out["attn_fwd_enc_self"] = dict(ms=t * 1e3, tflops=fl / t / 1e12, flops=fl)
    do = rnd(B, S_IN, D)
    t = time_kernel(lambda: ops.attn_bwd(do, q, k, v, o, lse, H, kpm=kpm, drop_p=0.2, drop_seed=1), iters=10)
    out["attn_bwd_enc_self"] = dict(ms=t * 1e3, tflops=2.5 * fl / t / 1e12, flops=2.5 * fl)
    # the north_star kernel's other operating points (roofline_attention): no dropout, and a batch of 3 x B - four rounds of its
    # 1 024-block launch instead of one and a third, i.e. the kernel's steady-state rate without the launch / first-tile / tail share
    t = time_kernel(lambda: ops.attn_fwd(q, k, v, H, kpm=kpm, drop_p=0.0, drop_seed=1))
    out["attn_fwd_enc_self_nodrop"] = dict(ms=t * 1e3, tflops=fl / t / 1e12, flops=fl)
    t = time_kernel(lambda: ops.attn_bwd(do, q, k, v, o, lse, H, kpm=kpm, drop_p=0.0, drop_seed=1), iters=10)
    out["attn_bwd_enc_self_nodrop"] = dict(ms=t * 1e3, tflops=2.5 * fl / t / 1e12, flops=2.5 * fl)
    B3 = 3 * B
    qkv3 = rnd(B3, S_IN, 3 * D)
    q3, k3, v3 = qkv3[..., :D], qkv3[..., D:2 * D], qkv3[..., 2 * D:]
    fl3 = 4.0 * S_IN * S_IN * D * B3
    for tag, dp in (("", 0.2), ("_nodrop", 0.0)):
        t = time_kernel(lambda: ops.attn_fwd(q3, k3, v3, H, drop_p=dp, drop_seed=1), iters=10)
        out["attn_fwd_enc_self_steady" + tag] = dict(ms=t * 1e3, tflops=fl3 / t / 1e12, flops=fl3)
    del qkv3, q3, k3, v3
    # the three GEMM layouts of one encoder Linear (in_proj: [B*S,512] x [1536,512]^T)
    M = B * S_IN
    x, w, dy = rnd(M, D), rnd(3 * D, D), rnd(M, 3 * D)
    bias = torch.zeros(3 * D, device=dev)
    gf = 2.0 * M * D * 3 * D
    t = time_kernel(lambda: ops.gemm(x, w, bias=bias))
    out["gemm_fwd_qkv"] = dict(ms=t * 1e3, tflops=gf / t / 1e12, flops=gf)
    t = time_kernel(lambda: ops.gemm(dy, w, b_kcontig=False))
    out["gemm_dx_qkv"] = dict(ms=t * 1e3, tflops=gf / t / 1e12, flops=gf)
    t = time_kernel(lambda: ops.gemm(dy, x, a_kcontig=False, b_kcontig=False, out_dtype=torch.float32, splitk=8))
    out["gemm_dw_qkv"] = dict(ms=t * 1e3, tflops=gf / t / 1e12, flops=gf)
    return out


def merge_census(parts):
    """Average per-step census over several batches: {family: {launches, flops, seconds}} summed, then divided by the
    number of batches (launches becomes the average number of launches per step; a family that only some batches use -
    e.g. the two-blocks-per-CU kernel past 8 192 packed rows - is weighted by how often the pool uses it)."""
    out = {}
    for fam in parts:
        for k, v in fam.items():
            o = out.setdefault(k, dict(launches=0.0, flops=0.0, seconds=0.0))
            for f in o:
                o[f] += v[f] / len(parts)
    return out


def gemm_census(model, batch, train_step):
    """Per-launch census of the GEMM kernels over ONE training step: the library records every GEMM argument block of
    the step, the kernel it dispatched to and - for the weight gradients - the grouped launch it was a member of
    (pa_gemm_record); then every launch is replayed on the same buffers under HIP events, grouped launches as groups
    (pa_gemm_group).  Returns {family: {launches, flops, seconds}}; family = kernel + operand layout:
    'ring' gemm3_kernel (128x128 tiles, one block per CU, 4-stage LDS ring), 'small' gemm3s_kernel (64x64 tiles),
    'pair' gemm_kernel (two blocks per CU), 'wide' gemm3w_kernel (128x256, opt-in), 'group' one pa_gemm_group launch
    of several weight-gradient GEMMs (flops = sum of its members); 'tt' both operands k-contiguous (forward and dX
    Linears), 'nn' neither (dW; the split-K slab reductions run as separate batched launches and are not included)."""
    import ctypes as C
    from plankassembly_amd import _lib as L
    lib = L.lib()
    torch.cuda.synchronize()
    lib.pa_gemm_record(1)
    train_step(0)
    torch.cuda.synchronize()
    n = lib.pa_gemm_record(0)
    rec = (L.GemmArgs * n)()
    kinds = (C.c_int32 * n)()
    groups = (C.c_int32 * n)()
    nk = lib.pa_gemm_recorded_kinds(C.cast(kinds, C.c_void_p), n)
    ng = lib.pa_gemm_recorded_groups(C.cast(groups, C.c_void_p), n)
    n = lib.pa_gemm_recorded(C.cast(rec, C.c_void_p), n)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fam = {}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 5
    names = {0: "pair", 1: "ring", 2: "wide", 3: "small", 4: "skinny"}
    i = 0
    while i < n:
        gid = groups[i] if i < ng else -1
        j = i + 1
        if gid >= 0:
            while j < n and j < ng and groups[j] == gid:
                j += 1
        a = rec[i]
        first = C.cast(C.byref(rec, i * C.sizeof(L.GemmArgs)), C.c_void_p)
        launch = (lambda: lib.pa_gemm_group(first, j - i, st)) if gid >= 0 else (lambda: lib.pa_gemm(first, st))
        launch()
        ev[0].record()
        for _ in range(reps):
            launch()
        ev[1].record()
        ev[1].synchronize()
        t = ev[0].elapsed_time(ev[1]) * 1e-3 / reps
        lay = "tt" if (a.a_kcontig and a.b_kcontig) else ("nn" if not (a.a_kcontig or a.b_kcontig) else "mixed")
        key = ("group" if gid >= 0 else names.get(kinds[i] if i < nk else 0, "pair")) + "_" + lay
        f = fam.setdefault(key, dict(launches=0, flops=0.0, seconds=0.0))
        f["launches"] += 1
        f["flops"] += sum(2.0 * rec[q].M * rec[q].N * rec[q].K * rec[q].batch for q in range(i, j))
        f["seconds"] += t
        i = j
    return fam


def pmc_traffic(kernel_key):
    """HBM bytes per launch of one kernel from the committed rocprofv3 PMC passes (profiles/r0N_pmc_traffic.json,
    made by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command)."""
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(here, "profiles", name)) as f:
                return json.load(f)["kernels"][kernel_key]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def in_step_trace(key):
    """Average duration (us) and launch count of a GEMM family INSIDE the profiled training step, from the committed
    rocprofv3 --kernel-trace summary of this same command (profiles/r0N*_train_kernel_trace_summary.txt, newest round first):
    the independent clock next to bench.py's own HIP-event census.  None for families the kernel name cannot tell apart
    (the attention kernels serve self-, cross- and causal attention under one name)."""
    import re
    pats = {"ring_tt": r"gemm3_kernelILb1ELb1ENS_5GemmPE", "pair_tt": r"gemm_kernelIDF16bLi64ELi2ELb1ELb1", "small_tt": r"gemm3s_kernel",
            "wide_tt": r"gemm3w_kernel", "group_nn": r"GemmGroup"}
    if key not in pats:
        return None
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        names = sorted((f for f in os.listdir(here) if re.match(r"r\d+\w*_train_kernel_trace_summary\.txt$", f) and "x3" not in f), reverse=True)
    except OSError:
        return None
    for name in names:
        calls, total_us = 0, 0.0          # every instantiation of the family (e.g. the pair kernel with and without the epilogue prefetch)
        with open(os.path.join(here, name)) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 4 and re.search(pats[key], parts[0]):
                    try:
                        calls += int(parts[1]); total_us += int(parts[1]) * float(parts[3])
                    except ValueError:
                        continue
        if calls:
            return {"file": "profiles/" + name, "calls": calls, "avg_launch_us": total_us / calls}
    return None


def usable_cores():
    """Host cores this process may really use: affinity mask, clipped by the cgroup CPU quota, capped at 64
    (torch's CPU kernels stop scaling - and oversubscription is catastrophic - well before that)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def attention_trace():
    """Average in-step durations (us) of the encoder self-attention kernels from the newest committed rocprofv3 summary: the
    forward kernel serves nothing else; the backward pair also serves no other launch (cross / decoder self run merged)."""
    import re
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        names = sorted((f for f in os.listdir(here) if re.match(r"r\d+\w*_train_kernel_trace_summary\.txt$", f) and "x3" not in f), reverse=True)
    except OSError:
        return None
    pats = {"fwd_us": r"attn5_fwd_kernelILb1E", "dq_us": r"attn4_bwd_dq_kernelILb1E", "dkv_us": r"attn4_bwd_dkv_kernelILb1E"}
    for name in names:
        got = {}
        with open(os.path.join(here, name)) as f:
            for line in f:
                parts = line.split()
                for k, pat in pats.items():
                    if len(parts) >= 4 and re.search(pat, parts[0]):
                        try:
                            got[k] = float(parts[3])
                        except ValueError:
                            pass
        if len(got) == 3:
            return dict(got, file="profiles/" + name)
    return None


def attention_census(cfgd, batch, drop_p):
    """The attention launches of ONE training step at the packed shapes of `batch`, timed under HIP events at the
    step's own arguments (variable-length rows, longest-first dispatch order, dropout): {family: {launches, flops,
    seconds}} with launches = per train step (one per layer).  flops = algorithmic 4*Lq*Lk*d per sample and layer for
    the forward, 2.5x that for the backward (dQ + dK/dV kernels together)."""
    from plankassembly_amd import ops
    d, h, ne, nd = cfgd["d"], cfgd["h"], cfgd["ne"], cfgd["nd"]
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)

    def rnd(*s_):
        return torch.randn(*s_, device=dev, generator=g).to(torch.bfloat16)

    cu, rowmap, n = batch["_pack"]
    B = batch["input_value"].shape[0]
    S = batch["input_value"].shape[1]
    T = batch["output_value"].shape[1]
    order = ops.pack_order(cu)
    lens = (cu[1:B + 1] - cu[:B]).tolist()
    kw = dict(drop_p=drop_p, drop_seed=5)
    fam = {}
    qkv = rnd(n, 3 * d)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    do = rnd(n, d)
    fl = sum(4.0 * l * l * d for l in lens)
    o, lse = ops.attn_varlen_fwd(q, k, v, h, cu, cu, B, S, S, order=order, **kw)
    t = time_kernel(lambda: ops.attn_varlen_fwd(q, k, v, h, cu, cu, B, S, S, order=order, **kw))
    fam["attn_enc_self_fwd"] = dict(launches=ne, flops=fl * ne, seconds=t * ne)
    t = time_kernel(lambda: ops.attn_varlen_bwd(do, q, k, v, o, lse, h, cu, cu, B, S, S, order=order, **kw), iters=10)
    fam["attn_enc_self_bwd"] = dict(launches=2 * ne, flops=2.5 * fl * ne, seconds=t * ne)
    qc, kvc, doc = rnd(B * T, d), rnd(n, 2 * d), rnd(B * T, d)
    kc, vc = kvc[:, :d], kvc[:, d:]
    flc = sum(4.0 * T * l * d for l in lens)
    oc, lsec = ops.attn_varlen_fwd(qc, kc, vc, h, None, cu, B, T, S, **kw)
    t = time_kernel(lambda: ops.attn_varlen_fwd(qc, kc, vc, h, None, cu, B, T, S, **kw))
    fam["attn_cross_fwd"] = dict(launches=nd, flops=flc * nd, seconds=t * nd)
    t = time_kernel(lambda: ops.attn_varlen_bwd(doc, qc, kc, vc, oc, lsec, h, None, cu, B, T, S, **kw), iters=10)
    # (T <= 128 queries per (sample, head): dQ and dK/dV blocks run in ONE launch, attn4_bwd_merged_kernel)
    merged = T <= 128
    fam["attn_cross_bwd"] = dict(launches=nd if merged else 2 * nd, flops=2.5 * flc * nd, seconds=t * nd)
    qd = rnd(B, T, 3 * d)
    q1, k1, v1 = qd[..., :d], qd[..., d:2 * d], qd[..., 2 * d:]
    dod = rnd(B, T, d)
    fld = 4.0 * T * T * d * B
    od, lsed = ops.attn_fwd(q1, k1, v1, h, causal=True, **kw)
    t = time_kernel(lambda: ops.attn_fwd(q1, k1, v1, h, causal=True, **kw))
    fam["attn_dec_self_fwd"] = dict(launches=nd, flops=fld * nd, seconds=t * nd)
    t = time_kernel(lambda: ops.attn_bwd(dod, q1, k1, v1, od, lsed, h, causal=True, **kw), iters=10)
    fam["attn_dec_self_bwd"] = dict(launches=nd if merged else 2 * nd, flops=2.5 * fld * nd, seconds=t * nd)
    return fam


def cpu_decode_check(dec_keep):
    """Part of the CPU-baseline leg: the oracle's KV-cached greedy decode of the first two rows of the benchmarked decode
    batch, all 1024 steps - timed (the reference-style CPU figure for the decode metric) and compared token for token with
    what the timed HIP runs produced for those rows (`token_exact`, per dtype: measured, not assumed)."""
    from oracle import plank_oracle as O
    torch.set_num_threads(usable_cores())
    cfg = O.OracleCfg(d_model=D, n_head=H, d_ff=FF, n_enc=NE, n_dec=ND, max_input_length=S_IN + 1, max_output_length=T_DEC)
    t0 = time.perf_counter()
    with torch.no_grad():
        s_ref, a_ref, marg = O.greedy_decode_cached(dec_keep["sd"], cfg, dec_keep["batch"], early_stop=False, return_margins=True)
    dt = time.perf_counter() - t0
    res = {"cpu": dict(value=s_ref.numel() / dt, unit="tokens/s", cores=usable_cores(), kind="port",
                       sample=f"oracle KV-cached greedy decode, f32, B={s_ref.shape[0]}, S={S_IN}, {s_ref.shape[1]} steps, {dt:.1f}s")}
    for ddtype, (s, a) in dec_keep["got"].items():
        neq = ((s != s_ref) | (a != a_ref)).nonzero()
        info = {"token_exact": len(neq) == 0, "rows_checked": int(s_ref.shape[0]), "steps_checked": int(s_ref.shape[1]),
                "pointer_copies_in_checked_rows": int((a_ref >= 0).sum()), "min_oracle_margin": float(marg.min())}
        if len(neq):
            r, t = int(neq[0][0]), int(neq[0][1])
            info.update(first_mismatch_step=t, first_mismatch_row=r, oracle_margin_there=float(marg[r, t]))
        res[ddtype] = info
    return res


def cpu_baseline(cfgd, sample_b=2, budget_s=20.0, fit_style=False):
    """The oracle (CPU restatement, fixture-pinned to the reference) timed on the host cores: full
    train step (fwd + bwd + Adam) on a bounded sample of the same workload (time-boxed).  fit_style (the tiny
    BASELINE config 0): the whole batch, warm-up 2, time 10 steps."""
    from oracle import plank_oracle as O
    from plankassembly_amd.data import synth_batch
    from plankassembly_amd.models import PlankModel
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(2022)
    c = cfgd
    m = PlankModel(c["d"], c["h"], c["ff"], 0.0, "relu", True, c["ne"], c["nd"], 3, 2, 4, 6, c["max_in"], c["max_out"], V, TOKEN)
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cfg = O.OracleCfg(d_model=c["d"], n_head=c["h"], d_ff=c["ff"], n_enc=c["ne"], n_dec=c["nd"], max_input_length=c["max_in"],
                      max_output_length=c["max_out"])
    if fit_style:
        sample_b = c["batch"]
    batch = synth_batch(sample_b, cfg_spec(c), seed=2022)
    batch.pop("name")
    mom = {k: torch.zeros_like(v) for k, v in params.items()}
    var = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(i):
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        O.train_forward(p, cfg, batch)["loss"].backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
        O.adam_step(params, grads, mom, var, i, lr=1e-4)

    warm = 2 if fit_style else 1
    tw = time.perf_counter()
    for i in range(warm):
        step(i + 1)
    tw = time.perf_counter() - tw
    steps = 0
    t0 = time.perf_counter()
    while True:
        step(steps + warm + 1)
        steps += 1
        dt = time.perf_counter() - t0
        if steps >= 10 or (not fit_style and dt + tw > budget_s):
            break
    return dict(value=sample_b * steps / dt, unit="samples/s", cores=cores, kind="port",
                sample=f"oracle train step (fwd+bwd+Adam, f32, torch CPU ops, {cores} threads), B={sample_b}, "
                       f"S={c['max_in'] - 1}, T={c['max_out']}, d_model={c['d']}, {c['ne']}+{c['nd']} layers, {steps} timed "
                       f"step(s) after {warm} warm-up, {dt:.1f}s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS))
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-kernels", action="store_true")
    ap.add_argument("--no-f32", action="store_true", help="skip the exact-f32 training leg (probes only)")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--long-steps", type=int, default=200, help="length of the extra steady-state region (noise estimate)")
    args = ap.parse_args()
    cfgd = CONFIGS[args.config]
    headline = args.config == "headline"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PLANK_BENCH_BACKEND=gloo is for TESTS only (tests/test_bench_launch_gpu.py): RCCL refuses two ranks on one device, gloo
    # carries device tensors through the host, so the N > 1 code path of this file can run on a one-GPU box.
    backend = os.environ.get("PLANK_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from plankassembly_amd.data import DevicePrefetcher, synth_batch
    from plankassembly_amd.distributed import GradSync
    from plankassembly_amd.optim import FusedAdam

    # ------------------------------------------------------------------ training
    B = args.batch or cfgd["batch"]
    S_in, T_out = cfgd["max_in"] - 1, cfgd["max_out"]
    model = build(args.dtype, cfgd["max_in"], cfgd["max_out"], 0.2, cfgd).train()
    opt = FusedAdam(model, lr=1e-4, grad_scale=1.0 / world)
    sync = None
    if dist.is_initialized():
        sync = GradSync(model)                      # (world size 1 under torchrun: the RCCL path still runs)
        sync.broadcast_parameters(0)
    # Raw collated batches (what a dataloader hands over), resident in HBM before the timed region.  Every rank draws the
    # SAME length distribution (seeds do not depend on the rank): weak scaling then measures the machine, not which rank
    # happened to draw the longest drawings.
    n_pool = 16
    raw = []
    for i in range(n_pool):
        b = synth_batch(B, cfg_spec(cfgd), seed=2022 + 1000 * i, device="cuda")
        b.pop("name")
        # what a CPU collate function knows for free (the reference's builds the padding mask on the host): the number of valid
        # encoder rows, as host metadata beside the device tensors - the step then needs no device -> host read and the two batch
        # preparation launches run on the main stream (DevicePrefetcher; tools/prep_probe.py: +0.04 instead of +0.10 ms per step)
        b["_n_valid"] = int((~b["input_mask"]).sum())
        raw.append(b)
    prepared = [model.prepare_batch(b) for b in raw]        # the same batches, prepared ahead (for `resident_prepared`)

    def step_on(batch):
        opt.zero_grad()
        out = model(batch)
        out["loss"].backward()
        opt.step()
        return out

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def cycle(pool, n):
        for i in range(n):
            yield pool[i % len(pool)]

    rank_ms = {}

    def timed(pool, fresh, steps=None, warmup=None, tag=None, mdl=None, stepper=None):
        """`warmup` untimed + `steps` timed steps (default: args.warmup / args.steps).  fresh: every step gets a batch that
        has not been prepared - pa_pack_rows / pa_group_rows (and the host -> device copies when the pool lives on the host)
        run inside the timed region, one step ahead on a side stream (DevicePrefetcher), as in trainer.run().  Returns the
        MAX over ranks of the bracketed wall time; `tag` additionally records every rank's own time (rank_ms)."""
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        mdl = model if mdl is None else mdl
        stepper = step_on if stepper is None else stepper
        total = warmup + steps
        it = DevicePrefetcher(mdl, cycle(pool, total)) if fresh else cycle(pool, total)
        out = None
        for _ in range(warmup):
            out = stepper(next(it))
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = stepper(next(it))
        fence()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        if dist.is_initialized() and world > 1:
            if tag:
                every = [torch.zeros_like(tt) for _ in range(world)]
                dist.all_gather(every, tt)
                rank_ms[tag] = [float(x.item()) / steps * 1e3 for x in every]
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elif tag:
            rank_ms[tag] = [dt / steps * 1e3]
        return float(tt.item()), out

    if dist.is_initialized() and backend == "nccl":
        # Set-up, not warm-up: RCCL builds its communicator state lazily on the first collectives of every size, and the stream
        # hand-over in front of Adam settles over the first few dozen steps (profiles/r04_gradsync_coalesce_sweep.txt: the first
        # 100 steps under a process group run 0.2 - 0.4 ms slower than the next 100).  The contract's W warm-up steps follow.
        for i in range(30):
            step_on(prepared[i % len(prepared)])
        fence()
    log(f"model + batches ready (rank {rank}/{world}, config {args.config})")
    dt, out = timed(raw, fresh=True, tag="train")          # the headline figure: a fresh batch every step
    loss = float(out["loss"].detach())
    assert math.isfinite(loss), "training diverged"
    samples_s = args.steps * B * world / dt
    log(f"train: {samples_s:.1f} samples/s, {dt / args.steps * 1e3:.2f} ms/step, loss {loss:.4f} (fresh batch every step)")
    # A longer region of the same loop (--long-steps, default 200 steps ~ 1 s): `value` is bound to exactly --steps steps by the driver's contract
    # (20 steps = 0.1 s there), so the run-to-run noise of `value` is read off this figure.
    long_steps = max(args.long_steps, args.steps)
    dt_long, _ = timed(raw, fresh=True, steps=long_steps, warmup=0)
    steady = dict(value=long_steps * B * world / dt_long, unit="samples/s", steps=long_steps, ms_per_step=dt_long / long_steps * 1e3,
                  note="the same fresh-batch loop over a longer timed region (barrier + synchronize on both sides, max over ranks)")
    log(f"       {steady['value']:.1f} samples/s, {steady['ms_per_step']:.3f} ms/step over {long_steps} steps")
    dt_res, _ = timed(prepared, fresh=False)               # round-1 figure: four prepared batches cycled
    resident = dict(value=args.steps * B * world / dt_res, unit="samples/s", ms_per_step=dt_res / args.steps * 1e3,
                    note="the same batches prepared before the timed region and cycled (what round 1 reported)")
    log(f"       {resident['value']:.1f} samples/s, {resident['ms_per_step']:.2f} ms/step with prepared batches")
    host_pool = [{k: (v.cpu().pin_memory() if torch.is_tensor(v) else v) for k, v in b.items() if k != "_n_valid"} for b in raw]   # (CPU mask: the prefetcher counts it itself)
    dt_h, _ = timed(host_pool, fresh=True)
    fresh_host = dict(value=args.steps * B * world / dt_h, unit="samples/s", ms_per_step=dt_h / args.steps * 1e3,
                      note="batches start in pinned host memory: PCIe copy + preparation + step (never `value`)")
    log(f"       {fresh_host['value']:.1f} samples/s, {fresh_host['ms_per_step']:.2f} ms/step from host batches")
    del host_pool

    def train_step(i):
        return step_on(prepared[i % len(prepared)])

    # ------------------------------------------------------------------ the same step on the parity path (exact-f32 MFMA)
    # north_star's tolerance (1e-4 against the f32 reference) is met by the f32 path only; bf16 is the throughput path
    # (BASELINE configs[1] names bf16 for the 1-GPU line).  Both figures go into the line; `value` stays the bf16 one.
    train_f32, train_x3 = None, None
    PARITY_NOTES = {
        "f32": "exact-f32 MFMA (v_mfma_f32_32x32x2_f32), f32 activations: the checker path of the 1e-4 parity tests",
        "x3": "f32 activations / parameters / LayerNorm / softmax / loss; every GEMM on the bf16 matrix pipe as hi*hi + hi*lo + lo*hi of "
              "the operands' bf16 hi / lo parts, f32 accumulation (pa_gemm_split_config), the attention products likewise "
              "(csrc/attention_x3.h); passes the f32 gate of tests/test_headline_gpu.py unchanged (test_x3_*)",
    }

    def parity_leg(name):
        m32 = build(name, cfgd["max_in"], cfgd["max_out"], 0.2, cfgd).train()
        opt32 = FusedAdam(m32, lr=1e-4, grad_scale=1.0 / world)
        sync32 = None
        if dist.is_initialized():
            sync32 = GradSync(m32)
            sync32.broadcast_parameters(0)

        def step32(batch):
            opt32.zero_grad()
            o = m32(batch)
            o["loss"].backward()
            opt32.step()
            return o

        # (x3: 30 steps of ~12 ms behind 8 warm-up steps - the first steps allocate the split scratch; exact f32: 10 of ~26 ms)
        n32 = max(5, min(args.steps, 30 if name == "x3" else 10))
        w32 = 8 if name == "x3" else 4
        dt32, o32 = timed(raw, fresh=True, steps=n32, warmup=w32, tag="train_" + name, mdl=m32, stepper=step32)
        assert math.isfinite(float(o32["loss"].detach())), name + " training diverged"
        res = dict(value=n32 * B * world / dt32, unit="samples/s", steps=n32, warmup=w32, ms_per_step=dt32 / n32 * 1e3,
                   dtype=name, final_loss=float(o32["loss"].detach()), note=PARITY_NOTES[name])
        log(f"train {name} (parity path): {res['value']:.1f} samples/s, {res['ms_per_step']:.2f} ms/step")
        if sync32 is not None:
            fence()
            sync32.detach()
        m32.register_grad_ready_hook(None)
        del m32, opt32, sync32, o32, step32
        torch.cuda.empty_cache()
        return res

    if args.dtype == "bf16" and not args.no_f32:
        train_f32 = parity_leg("f32")
        train_x3 = parity_leg("x3")

    # Everything below steps the model on rank 0 ONLY (kernel census, padded-encoder variant): the gradient exchange must be
    # off by then, or rank 0's backward would enqueue collectives the other ranks never join (found by
    # tests/test_bench_launch_gpu.py: the two-rank run hung here).
    if sync is not None:
        fence()
        sync.detach()
        sync = None

    census = None
    if rank == 0 and not args.no_kernels:
        # every batch of the pool (the batches the timed region cycled through): which GEMM kernel an encoder Linear goes to
        # depends on the batch's number of valid rows, so one batch cannot stand for the step
        parts = []
        for bi in range(len(prepared)):
            fam = gemm_census(model, prepared[bi], lambda i, bi=bi: step_on(prepared[bi]))
            if args.dtype == "bf16":
                fam.update(attention_census(cfgd, prepared[bi], 0.2))
            parts.append(fam)
        census = merge_census(parts)
        log("kernel census: " + ", ".join(f"{k}: {v['launches']:.1f} launches, {v['seconds'] / v['launches'] * 1e6:.1f} us avg, "
                                           f"{v['flops'] / v['seconds'] / 1e12:.0f} TF" for k, v in census.items()))
    fw = fwd_flops_per_sample(S_in, T_out, cfgd["d"], cfgd["ff"], cfgd["ne"], cfgd["nd"])
    train_flops = 3.0 * fw * B                                 # per GPU step, counted at the padded length (dense-equivalent)
    step_tflops = train_flops * args.steps / dt / 1e12
    valid_rows = sum(bt["_pack"][2] for bt in prepared) / len(prepared) if model.unpad else B * S_in
    # executed FLOPs: the encoder runs on the packed rows, its attention on the true lengths
    exec_flops = 0.0
    for bt in prepared:
        cu = bt["_pack"][0]
        lens = (cu[1:B + 1] - cu[:B]).tolist()
        exec_flops += 3.0 * sum(fwd_flops_per_sample(l, T_out, cfgd["d"], cfgd["ff"], cfgd["ne"], cfgd["nd"]) for l in lens)
    exec_tflops = exec_flops / len(prepared) * args.steps / dt / 1e12

    # secondary: the same step with the encoder left padded (what the reference computes row for row), rank 0, N=1 only
    dense = None
    if rank == 0 and world == 1 and not args.no_kernels and model.unpad and headline:
        model.unpad = False
        dbatches = [{k: v for k, v in bt.items() if not k.startswith("_")} for bt in prepared]
        for i in range(3):
            step_on(dbatches[i % len(dbatches)])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(10):
            step_on(dbatches[i % len(dbatches)])
        torch.cuda.synchronize()
        ddt = (time.perf_counter() - t1) / 10
        dense = dict(value=B / ddt, unit="samples/s", ms_per_step=ddt * 1e3,
                     note="encoder not packed: all B*S rows computed under the key-padding mask")
        model.unpad = True
        log(f"dense (padded encoder): {dense['value']:.1f} samples/s, {dense['ms_per_step']:.2f} ms/step")

    # ------------------------------------------------------------------ greedy decode
    decode, dec_keep = None, None
    if not args.no_decode and headline:
        from plankassembly_amd.decode import GreedyDecoder
        sync = None
        model.register_grad_ready_hook(None)
        del opt, model, prepared, raw, out
        torch.cuda.empty_cache()
        decode = {}
        for ddtype in ([args.dtype] if args.dtype == "f32" else ["bf16", "f32"]):
            dm = apply_gains(build(ddtype, S_IN + 1, T_DEC, 0.0), DECODE_GAINS).eval()
            dm._ensure_handle(); dm._refresh_shadow()
            dec = GreedyDecoder(dm, use_graph=True, strict_graph=True)     # a failed capture is an error here, not a silent eager run
            from plankassembly_amd.data import spec_for
            db = synth_batch(B_DEC, spec_for("decode"), seed=7, device="cuda")
            db.pop("name")
            if dec_keep is None:
                dec_keep = {"sd": {k: v.detach().float().cpu().clone() for k, v in dm.state_dict().items()},
                            "batch": {k: v[:2].cpu() for k, v in db.items()}, "got": {}}
            db = dm.prepare_batch(db)
            with torch.no_grad():
                dec.run(db, max_len=T_DEC, early_stop=False)              # warm-up (captures the step graph)
                fence()
                t0 = time.perf_counter()
                toks, atts = dec.run(db, max_len=T_DEC, early_stop=False)
                fence()
                ddt = time.perf_counter() - t0
            dec_keep["got"][ddtype] = (toks[:2].cpu(), atts[:2].cpu())
            tt = torch.tensor([ddt], device="cuda", dtype=torch.float64)
            if dist.is_initialized() and world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ddt = float(tt.item())
            esz = 2 if ddtype == "bf16" else 4
            # algorithmic HBM bytes per step averaged over t: weights + cross K/V + self K/V
            w_bytes = (ND * (4 * D * D + 4 * D * D + 2 * D * FF) + V * D + D * D) * esz
            kv_bytes = 2 * ND * B_DEC * (S_IN + T_DEC / 2) * D * esz
            kv_cache_form_bytes = w_bytes + kv_bytes
            # bf16 default since round 5: cross-attention in absorbed form (csrc/decode_mq.h) - every head attends over the encoder
            # output rows themselves: ONE [S][d] stream per layer and step instead of K and V; W_o,h W_v,h is one [d][H d] matrix
            mq = (ddtype == "bf16" and D == 512 and H <= 8 and B_DEC <= 512 and os.environ.get("PLANK_DECODE_MQ", "1") != "0"
                  and os.environ.get("PLANK_DECODE_FOLD_LN", "1") != "0")
            mqf = (ddtype == "f32" and D == 512 and H <= 8 and B_DEC <= 512 and os.environ.get("PLANK_DECODE_MQ_F32", "1") != "0"
                   and os.environ.get("PLANK_DECODE_FOLD_LN", "1") != "0")
            if mq:
                # round 6: from 200 batch elements on the bf16 step absorbs its SELF-attention too (csrc/decode.hip mq_self_bf): the layer-input
                # rows are cached instead of K and V (one [t][d] stream), q~ from the expand launch, W_o,h W_v,h on the context rows
                sbf = os.environ.get("PLANK_DECODE_MQ_SELF_BF16")
                self_abs = (sbf != "0") if sbf is not None else B_DEC >= int(os.environ.get("PLANK_DECODE_MQ_SELF_MINB", "200"))
                self_abs = self_abs and os.environ.get("PLANK_DECODE_F32_RESID", "1") != "0" and D // H == 64
                w_bytes = (ND * (((4 + 2 * H) if self_abs else (6 + H)) * D * D + 2 * D * FF) + V * D + D * D) * esz
                kv_bytes = ND * B_DEC * (S_IN + (1 if self_abs else 2) * T_DEC / 2) * D * esz
            if mqf:        # (f32: W_v as its own launch, the weights are the reference's; the SELF-attention is absorbed too - layer-input rows cached)
                self_rows = 1 if os.environ.get("PLANK_DECODE_MQ_SELF", "1") != "0" else 2
                kv_bytes = ND * B_DEC * (S_IN + self_rows * T_DEC / 2) * D * esz
            decode[ddtype] = dict(value=B_DEC * T_DEC * world / ddt, unit="tokens/s", batch=B_DEC, max_len=T_DEC, seq_in=S_IN,
                                  ms_per_step=ddt / T_DEC * 1e3, graph=bool(dec.use_graph),
                                  hbm_gbs=(w_bytes + kv_bytes) * T_DEC / ddt / 1e9,
                                  hbm_frac=(w_bytes + kv_bytes) * T_DEC / ddt / 1e9 / PEAK_HBM_GBS,
                                  token_exact=None,      # filled in by the CPU leg (cpu_decode_check); None = not checked
                                  includes="encoder + cross-K/V projection + 1024 decode steps")
            if mq or mqf:
                decode[ddtype]["cross_attention"] = ("absorbed: q~_h = W_k,h^T q_h attends over the memory rows, W_v behind the softmax "
                                                     "(no cross-K/V projection, one [S][d] stream per layer and step)")
                decode[ddtype]["includes"] = "encoder + 1024 decode steps"
                if mqf and os.environ.get("PLANK_DECODE_MQ_SELF", "1") != "0":
                    decode[ddtype]["self_attention"] = "absorbed too: the step caches the layer-input rows instead of K and V"
                if mq and self_abs:
                    decode[ddtype]["self_attention"] = "absorbed too (from 200 batch elements on): the step caches the layer-input rows instead of K and V"
                # the same time against the bytes the K/V-cache form moves (what rounds 1-4 quoted hbm_frac on)
                decode[ddtype]["kv_cache_form_equiv_gbs"] = kv_cache_form_bytes * T_DEC / ddt / 1e9
            log(f"decode {ddtype}: {decode[ddtype]['value']:.0f} tokens/s, {decode[ddtype]['ms_per_step']:.3f} ms/step")
            # the reference's own evaluation batch (configs/train_complete.yaml BATCH_SIZE 16 -> trainer_complete.py validation_step):
            # one lane, the same model, max_len steps.  Below ~256 elements the absorbed attention launches run as RANGE BLOCKS
            # (several blocks per element, csrc/decode_mq.h) - one block per element left 240 of 256 CUs idle.
            if B_DEC > 16:
                del dec
                dec16 = GreedyDecoder(dm, use_graph=True, strict_graph=True, lanes=1)
                db16 = synth_batch(16, spec_for("decode"), seed=7, device="cuda"); db16.pop("name")
                db16 = dm.prepare_batch(db16)
                with torch.no_grad():
                    dec16.run(db16, max_len=T_DEC, early_stop=False)
                    fence()
                    t0 = time.perf_counter()
                    dec16.run(db16, max_len=T_DEC, early_stop=False)
                    fence()
                    d16 = time.perf_counter() - t0
                decode[ddtype]["b16"] = dict(value=16 * T_DEC / d16, unit="tokens/s", batch=16, ms_per_step=d16 / T_DEC * 1e3,
                                             us_per_token_row=d16 / T_DEC / 16 * 1e6,
                                             vs_b256_per_token_row=(d16 / 16) / (ddt / B_DEC),
                                             note="latency-bound: ~57 dependent launches per step whatever the batch")
                log(f"decode {ddtype} at batch 16: {decode[ddtype]['b16']['value']:.0f} tokens/s, {decode[ddtype]['b16']['ms_per_step']:.3f} ms/step")
                dec = dec16
                del db16
            del dec, dm, db
            torch.cuda.empty_cache()
            mq32 = ddtype == "f32" and D == 512 and H <= 8 and B_DEC <= 512 and os.environ.get("PLANK_DECODE_MQ_F32", "1") != "0" \
                and os.environ.get("PLANK_DECODE_FOLD_LN", "1") != "0"
            if (mq or mq32) and rank == 0 and not args.no_kernels:
                decode[ddtype]["roofline_cross_attention"] = decode_cross_roofline(ddtype)

    kern = None
    if rank == 0 and not args.no_kernels and headline:
        kern = kernel_rooflines(B)
        log("kernel rooflines: " + ", ".join(f"{k} {v['tflops']:.0f} TF" for k, v in kern.items()))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(cfgd, fit_style=(args.config == "tiny"))
        log(f"cpu baseline: {cpu['value']:.3f} samples/s on {cpu['cores']} threads")
        if dec_keep is not None and dec_keep["got"]:
            chk = cpu_decode_check(dec_keep)
            cpu["decode"] = chk.pop("cpu")
            for ddtype, info in chk.items():
                decode[ddtype].update(info)
            log("decode check vs oracle: " + ", ".join(f"{k}: token_exact={v['token_exact']}"
                                                       + (f" (first mismatch at step {v['first_mismatch_step']}, oracle margin {v['oracle_margin_there']:.2e})"
                                                          if not v["token_exact"] else "") for k, v in chk.items()))
            # the parity-meeting decode figure (north star: token indices bit-exact) is the f32 one
            if "f32" in decode:
                decode["parity_meeting"] = {"dtype": "f32", "value": decode["f32"]["value"], "unit": "tokens/s",
                                            "token_exact": decode["f32"]["token_exact"]}
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        if train_x3 and train_f32:
            # the fastest path that passes the 1e-4 / 1e-5 + 1e-4 * scale gates (tests/test_headline_gpu.py: test_f32_* and test_x3_*)
            best = max((train_x3, train_f32), key=lambda t: t["value"])
            parity = {"dtype": best["dtype"], "value": best["value"], "unit": "samples/s", "ms_per_step": best["ms_per_step"]}
        elif train_f32:
            parity = {"dtype": "f32", "value": train_f32["value"], "unit": "samples/s", "ms_per_step": train_f32["ms_per_step"]}
        elif args.dtype == "f32":
            parity = {"dtype": "f32", "value": samples_s, "unit": "samples/s", "ms_per_step": dt / args.steps * 1e3}
        else:
            parity = None
        prec = (" [value: bf16 MFMA, f32 accumulate / master weights; train.f32 / train.x3: the same step in exact f32 / f32 with bf16x3 products = the parity paths]"
                if args.dtype == "bf16" else " [exact-f32 MFMA: the parity path]")
        line = {
            "metric": ("train samples/sec (fwd+bwd+all-reduce+Adam), d_model=512 seq=1024" if headline else
                       f"train samples/sec (fwd+bwd+all-reduce+Adam), config {args.config}") + prec,
            "value": samples_s, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic tokenised-drawing batches (SURVEY 8d), random-init weights; collated "
                                        "batches resident in HBM (+ the collate function's host-side count of valid rows), a fresh one "
                                        "every step (row packing + embedding-row grouping kernels inside the timed region)",
            "config": {"workload": f"{cfgd['name']}, S={S_in} (MAX_INPUT_LENGTH {cfgd['max_in']}), T={T_out}, batch {B}/GPU",
                       "global_batch": B * world, "seq_len": S_in, "parallelism": f"dp{world}"},
            "final_loss": loss,
            "train": {args.dtype: dict(value=samples_s, unit="samples/s", ms_per_step=dt / args.steps * 1e3, steps=args.steps),
                      **({"f32": train_f32} if train_f32 else {}), **({"x3": train_x3} if train_x3 else {}),
                      "parity_meeting": parity},
            "steady_state": steady,
            "rccl_ranks": {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
                           "backend": (dist.get_backend() if dist.is_initialized() else None),
                           "process_group": bool(dist.is_initialized()),
                           "ms_per_step_by_rank": {k: {"min": min(v), "max": max(v), "ranks": len(v)} for k, v in rank_ms.items()}},
            "resident_prepared": resident, "fresh_host": fresh_host,
            "encoder_rows": {"padded": B * S_in, "valid_avg": valid_rows,
                             "note": "padded encoder rows are packed away before the first layer (results identical: "
                                     "they never reach the loss); `dense_padded` is the same step without packing"},
            "dense_padded": dense,
            "train_dense_equiv_tflops_per_gpu": step_tflops, "train_dense_equiv_mfma_frac": step_tflops / PEAK_BF16_TFLOPS,
            "train_executed_tflops_per_gpu": exec_tflops, "train_executed_mfma_frac": exec_tflops / PEAK_BF16_TFLOPS,
            "decode": decode,
        }
        if census:
            # dominant kernel family of the step by time, over the GEMM families AND the attention launches
            key = max(census, key=lambda k: census[k]["seconds"])
            c = census[key]
            names = {"ring_tt": "gemm3_kernel<true,true,GemmP> (bf16, 128x128x64 tiles, one block per CU, 4-stage LDS ring)",
                     "ring_nn": "gemm3_kernel<false,false,GemmP> (bf16 dW, split-K slabs)",
                     "group_nn": "gemm3_kernel<false,false,GemmGroup> (all weight-gradient GEMMs of a backward segment in one "
                                 "launch; bf16, 128x128x64 tiles, split-K slabs)",
                     "small_tt": "gemm3s_kernel (bf16, 64x64x64 tiles, 4-stage LDS ring)",
                     "pair_tt": "gemm_kernel<bf16,64,2,true,true,...> (two blocks per CU)",
                     "wide_tt": "gemm3w_kernel<2,4> (bf16, 128x256x64 tiles, one block per CU, 3-stage LDS ring)",
                     "attn_enc_self_fwd": "attn5_fwd_kernel<true,3,3> on the packed encoder rows (self-attention forward)",
                     "attn_enc_self_bwd": "attn4_bwd_dq_kernel + attn4_bwd_dkv_kernel on the packed encoder rows",
                     "attn_cross_fwd": "attn4_fwd_kernel<true,2> (in-block key split), decoder cross-attention forward",
                     "attn_cross_bwd": "attn4_bwd_merged_kernel (dQ and dK/dV blocks in one launch), decoder cross-attention backward",
                     "attn_dec_self_fwd": "attn4_fwd_kernel<true,1>, decoder causal self-attention forward",
                     "attn_dec_self_bwd": "attn4_bwd_merged_kernel (dQ and dK/dV blocks in one launch), decoder causal self-attention backward"}
            ach = c["flops"] / c["seconds"] / 1e12
            line["roofline"] = {"kernel": names.get(key, key) + " - every launch of it in a train step, timed under HIP events at the "
                                          f"step's own arguments, averaged over the {n_pool} batches the timed region cycles through",
                                "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                "frac": ach / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(key if key.startswith("attn") else "gemm_" + key),
                                "launches_per_step": c["launches"],
                                "algorithmic_flops_per_launch": c["flops"] / c["launches"],
                                "avg_launch_us": c["seconds"] / c["launches"] * 1e6}
            tr = in_step_trace(key)
            if tr:
                # `achieved` / `frac` are the figure that FOLLOWS FROM profiles/ - this run's algorithmic FLOPs per
                # launch over the kernel's average duration INSIDE the profiled step (the committed rocprofv3 --kernel-trace summary
                # of this same command).  bench.py's own HIP-event replay of the recorded launches (back to back, operands warm) runs
                # ~10 % faster than the same launches inside the step; it stays on the line as `live_replay`.
                tf = c["flops"] / c["launches"] / (tr["avg_launch_us"] * 1e-6) / 1e12
                line["roofline"]["live_replay"] = {"achieved": ach, "frac": ach / PEAK_BF16_TFLOPS,
                                                   "avg_launch_us": c["seconds"] / c["launches"] * 1e6,
                                                   "note": "HIP events over the step's recorded launches replayed back to back (this run)"}
                line["roofline"].update(achieved=tf, frac=tf / PEAK_BF16_TFLOPS, avg_launch_us=tr["avg_launch_us"],
                                        duration_source=dict(tr, archived=True,
                                                             note="average in-step duration parsed from the committed rocprofv3 summary named in "
                                                                  "`file`; FLOPs per launch measured by this run"))
                line["roofline"]["in_step_rocprof"] = dict(tr, achieved=tf, frac=tf / PEAK_BF16_TFLOPS, archived=True)
            # context, ARCHIVED (measured once in round 4, not by this run): what the vendor's GEMM reaches at these shapes
            line["roofline"]["vendor_same_shapes"] = {
                "archived": True, "file": "profiles/r04_step_floor_probes.txt",
                "note": "kernel durations under rocprofv3, torch.matmul (hipBLASLt) vs pa_gemm at 8704 rows: 8704x1536x512 22.0 vs 28.8 us "
                        "(0.25 of the bf16 MFMA peak), x1024x512 17.3 vs 23.8, x512x1024 16.2 vs 20.7, x512x512 13.2 vs 14.4; weight "
                        "gradients 54-58 vs 39; 2048-row Linears 7.2-12.5 vs 7.0-17.6 (section 12 of the file: tile-count quantisation)"}
            line["kernel_census"] = {k: {"launches": round(v["launches"], 2), "avg_launch_us": round(v["seconds"] / v["launches"] * 1e6, 2),
                                         "ms_per_step": round(v["seconds"] * 1e3, 3),
                                         "tflops": round(v["flops"] / v["seconds"] / 1e12, 1),
                                         "mfma_frac": round(v["flops"] / v["seconds"] / 1e12 / PEAK_BF16_TFLOPS, 4)}
                                     for k, v in census.items()}
        if kern:
            # The kernel north_star sets a target for (>= 0.70 of the MFMA roofline on the attention at d_model 512, seq 1024): the
            # encoder self-attention.  Forward with / without dropout, backward, at the padded benchmark shape (dense-equivalent
            # FLOPs 4 S^2 d B; ~25 % of the key tiles are masked and skipped), the steady-state rate (3 x the batch, no mask), the
            # packed launches of the training step (HIP events, `kernel_census`) and the archived in-step rocprofv3 durations.
            def frac(v):
                return {"us": round(v["ms"] * 1e3, 2), "tflops": round(v["tflops"], 1), "frac": round(v["tflops"] / PEAK_BF16_TFLOPS, 4)}
            ra = {"kernel": "attn5_fwd_kernel<DROP,3,3> (forward; csrc/attention5.h), attn4_bwd_dq_kernel + attn4_bwd_dkv_kernel (backward); dh 64, bf16",
                  "bound": "mfma", "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "target_frac": 0.70,
                  "padded_b16_s1024": {"fwd_dropout": frac(kern["attn_fwd_enc_self"]), "fwd_no_dropout": frac(kern["attn_fwd_enc_self_nodrop"]),
                                       "bwd_dropout": frac(kern["attn_bwd_enc_self"]), "bwd_no_dropout": frac(kern["attn_bwd_enc_self_nodrop"])},
                  "steady_state_b48_s1024_unmasked": {"fwd_dropout": frac(kern["attn_fwd_enc_self_steady"]),
                                                      "fwd_no_dropout": frac(kern["attn_fwd_enc_self_steady_nodrop"])},
                  "note": "issue-bound at dh 64: per 32 x 32 score chunk 8 MFMAs (256 cycles of the matrix pipe) stand against 16 v_exp_f32, "
                          "8 v_cvt_pk, 8 packed adds, 12-16 LDS reads and - with dropout - 48 hash / compare / select instructions; a SIMD "
                          "issues ~one instruction per 4.7 cycles whatever its type (profiles/r04_attn_issue_bound.txt, r05_attention_shape_sweep.txt)"}
            if census and "attn_enc_self_fwd" in census:
                ra["in_step_packed"] = {k2: {"launches": round(census[k]["launches"], 2),
                                             "us": round(census[k]["seconds"] / census[k]["launches"] * 1e6, 2),
                                             "tflops": round(census[k]["flops"] / census[k]["seconds"] / 1e12, 1),
                                             "frac": round(census[k]["flops"] / census[k]["seconds"] / 1e12 / PEAK_BF16_TFLOPS, 4)}
                                        for k, k2 in (("attn_enc_self_fwd", "fwd"), ("attn_enc_self_bwd", "bwd_dq_plus_dkv")) if k in census}
                ra["all_attention_ms_per_step"] = round(sum(v["seconds"] for k, v in census.items() if k.startswith("attn")) * 1e3, 3)
                tr5 = attention_trace()
                if tr5:
                    flf = census["attn_enc_self_fwd"]["flops"] / census["attn_enc_self_fwd"]["launches"]
                    ra["in_step_rocprof"] = dict(tr5, archived=True,
                                                 fwd_frac=round(flf / (tr5["fwd_us"] * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                                 bwd_frac=round(2.5 * flf / ((tr5["dq_us"] + tr5["dkv_us"]) * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4))
            line["roofline_attention"] = ra
            line["kernels"] = {k: {"ms": round(v["ms"], 4), "tflops": round(v["tflops"], 1),
                                   "mfma_frac": round(v["tflops"] / PEAK_BF16_TFLOPS, 4)} for k, v in kern.items()}
        if cpu:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
