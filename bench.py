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

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TOKEN = types.SimpleNamespace(END=512, PAD=513)
T_START = time.perf_counter()
D, H, FF, NE, ND, V = 512, 8, 1024, 6, 6, 514
S_IN, T_OUT, B_TRAIN = 1024, 128, 16
B_DEC, T_DEC = 256, 1024
PEAK_BF16_TFLOPS = 2500.0        # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def fwd_flops_per_sample(S, T, d=D, ff=FF, ne=NE, nd=ND, v=V):
    """2*MAC, forward."""
    enc = 8 * S * d * d + 4 * S * S * d + 4 * S * d * ff
    dec = 12 * T * d * d + 4 * S * d * d + 4 * T * T * d + 4 * T * S * d + 4 * T * d * ff
    heads = 2 * T * d * v + 2 * T * d * d + 2 * T * T * d + 2 * T * d
    return ne * enc + nd * dec + heads


def build(compute_dtype, max_in, max_out, dropout):
    from plankassembly_amd.models import PlankModel
    torch.manual_seed(2022)
    m = PlankModel(D, H, FF, dropout, "relu", True, NE, ND, 3, 2, 4, 6, max_in, max_out, V, TOKEN,
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
    names = {0: "pair", 1: "ring", 2: "wide", 3: "small"}
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
    """HBM bytes per launch of one kernel from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json,
    made by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["kernels"][kernel_key]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
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


def cpu_baseline(sample_b=2, budget_s=20.0):
    """The oracle (CPU restatement, fixture-pinned to the reference) timed on the host cores: full
    train step (fwd + bwd + Adam) on a bounded sample of the same workload (time-boxed)."""
    from oracle import plank_oracle as O
    from plankassembly_amd.data import spec_for, synth_batch
    from plankassembly_amd.models import PlankModel
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(2022)
    m = PlankModel(D, H, FF, 0.0, "relu", True, NE, ND, 3, 2, 4, 6, S_IN + 1, T_OUT, V, TOKEN)
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cfg = O.OracleCfg(d_model=D, n_head=H, d_ff=FF, n_enc=NE, n_dec=ND, max_input_length=S_IN + 1,
                      max_output_length=T_OUT)
    batch = synth_batch(sample_b, spec_for("headline"), seed=2022)
    batch.pop("name")
    mom = {k: torch.zeros_like(v) for k, v in params.items()}
    var = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(i):
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        O.train_forward(p, cfg, batch)["loss"].backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
        O.adam_step(params, grads, mom, var, i, lr=1e-4)

    tw = time.perf_counter()
    step(1)
    tw = time.perf_counter() - tw
    steps = 0
    t0 = time.perf_counter()
    while True:
        step(steps + 2)
        steps += 1
        dt = time.perf_counter() - t0
        if dt + tw > budget_s or steps >= 10:
            break
    return dict(value=sample_b * steps / dt, unit="samples/s", cores=cores, kind="port",
                sample=f"oracle train step (fwd+bwd+Adam, f32, torch CPU ops, {cores} threads), B={sample_b}, S={S_IN}, "
                       f"T={T_OUT}, {steps} timed step(s) after 1 warm-up, {dt:.1f}s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-kernels", action="store_true")
    ap.add_argument("--batch", type=int, default=B_TRAIN)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from plankassembly_amd.data import spec_for, synth_batch
    from plankassembly_amd.distributed import GradSync
    from plankassembly_amd.optim import FusedAdam

    # ------------------------------------------------------------------ training
    B = args.batch
    model = build(args.dtype, S_IN + 1, T_OUT, 0.2).train()
    opt = FusedAdam(model, lr=1e-4, grad_scale=1.0 / world)
    if world > 1:
        sync = GradSync(model)
        sync.broadcast_parameters(0)
    batches = []
    for i in range(4):
        b = synth_batch(B, spec_for("headline"), seed=2022 + 1000 * i + rank, device="cuda")
        b.pop("name")
        batches.append(model.prepare_batch(b))      # resident in HBM, encoder rows packed, before the timed region

    def train_step(i):
        opt.zero_grad()
        out = model(batches[i % len(batches)])
        out["loss"].backward()
        opt.step()
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"model + batches ready (rank {rank}/{world})")
    for i in range(args.warmup):
        out = train_step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = train_step(i)
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    loss = float(out["loss"].detach())
    assert math.isfinite(loss), "training diverged"
    census = None
    if rank == 0 and not args.no_kernels:
        census = gemm_census(model, batches[0], train_step)
        log("gemm census: " + ", ".join(f"{k}: {v['launches']} launches, {v['seconds'] / v['launches'] * 1e6:.1f} us avg, "
                                         f"{v['flops'] / v['seconds'] / 1e12:.0f} TF" for k, v in census.items()))
    samples_s = args.steps * B * world / dt
    log(f"train: {samples_s:.1f} samples/s, {dt / args.steps * 1e3:.2f} ms/step, loss {loss:.4f}")
    train_flops = 3.0 * fwd_flops_per_sample(S_IN, T_OUT) * B      # per GPU step, counted at the padded length (dense-equivalent)
    step_tflops = train_flops * args.steps / dt / 1e12
    valid_rows = sum(bt["_pack"][2] for bt in batches) / len(batches) if model.unpad else B * S_IN

    # secondary: the same step with the encoder left padded (what the reference computes row for row), rank 0, N=1 only
    dense = None
    if rank == 0 and world == 1 and not args.no_kernels and model.unpad:
        model.unpad = False
        dbatches = [{k: v for k, v in bt.items() if k != "_pack"} for bt in batches]
        def dense_step(i):
            opt.zero_grad()
            o = model(dbatches[i % len(dbatches)])
            o["loss"].backward()
            opt.step()
        for i in range(3):
            dense_step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(10):
            dense_step(i)
        torch.cuda.synchronize()
        ddt = (time.perf_counter() - t1) / 10
        dense = dict(value=B / ddt, unit="samples/s", ms_per_step=ddt * 1e3,
                     note="encoder not packed: all B*S rows computed under the key-padding mask")
        model.unpad = True
        log(f"dense (padded encoder): {dense['value']:.1f} samples/s, {dense['ms_per_step']:.2f} ms/step")

    # ------------------------------------------------------------------ greedy decode
    decode = None
    if not args.no_decode:
        from plankassembly_amd.decode import GreedyDecoder
        if world > 1:
            del sync
        del opt, model, batches, out
        torch.cuda.empty_cache()
        dm = build(args.dtype, S_IN + 1, T_DEC, 0.0).eval()
        dm._ensure_handle(); dm._refresh_shadow()
        dec = GreedyDecoder(dm)
        db = synth_batch(B_DEC, spec_for("decode"), seed=7 + rank, device="cuda")
        db.pop("name")
        db = dm.prepare_batch(db)
        log("decode model + batch ready")
        with torch.no_grad():
            dec.run(db, max_len=T_DEC, early_stop=False)              # warm-up (captures the step graph)
            log("decode warm-up done")
            fence()
            t0 = time.perf_counter()
            toks, _ = dec.run(db, max_len=T_DEC, early_stop=False)
            fence()
            ddt = time.perf_counter() - t0
        tt = torch.tensor([ddt], device="cuda", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ddt = float(tt.item())
        esz = 2 if args.dtype == "bf16" else 4
        # algorithmic HBM bytes per step averaged over t: weights + cross K/V + self K/V
        w_bytes = (ND * (4 * D * D + 4 * D * D + 2 * D * FF) + V * D + D * D) * esz
        kv_bytes = 2 * ND * B_DEC * (S_IN + T_DEC / 2) * D * esz
        decode = dict(value=B_DEC * T_DEC * world / ddt, unit="tokens/s", batch=B_DEC, max_len=T_DEC, seq_in=S_IN,
                      ms_per_step=ddt / T_DEC * 1e3, graph=bool(dec.use_graph),
                      hbm_gbs=(w_bytes + kv_bytes) * T_DEC / ddt / 1e9,
                      hbm_frac=(w_bytes + kv_bytes) * T_DEC / ddt / 1e9 / PEAK_HBM_GBS,
                      includes="encoder + cross-K/V projection + 1024 decode steps")
        log(f"decode: {decode['value']:.0f} tokens/s, {decode['ms_per_step']:.3f} ms/step")
        del dec, dm
        torch.cuda.empty_cache()

    kern = None
    if rank == 0 and not args.no_kernels:
        kern = kernel_rooflines(B)
        log("kernel rooflines: " + ", ".join(f"{k} {v['tflops']:.0f} TF" for k, v in kern.items()))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline()
        log(f"cpu baseline: {cpu['value']:.3f} samples/s on {cpu['cores']} threads")
    if world > 1:
        dist.barrier()
    if rank == 0:
        line = {
            "metric": "train samples/sec (fwd+bwd+all-reduce+Adam), d_model=512 seq=1024",
            "value": samples_s, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic tokenised-drawing batches (SURVEY 8d), random-init weights",
            "config": {"workload": "train_complete.yaml model d_model=512 H=8 dff=1024 6+6 post-norm layers, dropout 0.2, "
                                   f"S={S_IN} (MAX_INPUT_LENGTH {S_IN + 1}), T={T_OUT}, batch {B}/GPU",
                       "global_batch": B * world, "seq_len": S_IN, "parallelism": f"dp{world}"},
            "final_loss": loss,
            "encoder_rows": {"padded": B * S_IN, "valid_avg": valid_rows,
                             "note": "padded encoder rows are packed away before the first layer (results identical: "
                                     "they never reach the loss); `dense_padded` is the same step without packing"},
            "dense_padded": dense,
            "train_dense_equiv_tflops_per_gpu": step_tflops, "train_dense_equiv_mfma_frac": step_tflops / PEAK_BF16_TFLOPS,
            "decode": decode,
        }
        if census:
            # dominant kernel of the step by time: gemm3_kernel<A_KC, B_KC> (k-contiguous operands) = the forward and dX
            # Linears whose tile count fits one round of the 256 CUs
            key = max(census, key=lambda k: census[k]["seconds"])
            c = census[key]
            names = {"ring_tt": "gemm3_kernel<true,true,GemmP> (bf16, 128x128x64 tiles, one block per CU, 4-stage LDS ring)",
                     "ring_nn": "gemm3_kernel<false,false,GemmP> (bf16 dW, split-K slabs)",
                     "group_nn": "gemm3_kernel<false,false,GemmGroup> (all weight-gradient GEMMs of a backward segment in one "
                                 "launch; bf16, 128x128x64 tiles, split-K slabs)",
                     "small_tt": "gemm3s_kernel (bf16, 64x64x64 tiles, 4-stage LDS ring)",
                     "pair_tt": "gemm_kernel<bf16,64,2,true,true,...> (two blocks per CU)",
                     "wide_tt": "gemm3w_kernel<2,4> (bf16, 128x256x64 tiles, one block per CU, 3-stage LDS ring)"}
            ach = c["flops"] / c["seconds"] / 1e12
            line["roofline"] = {"kernel": names.get(key, key) + " - every launch of it in one train step, replayed under HIP events",
                                "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                "frac": ach / PEAK_BF16_TFLOPS, "traffic": pmc_traffic("gemm_" + key),
                                "launches_per_step": c["launches"],
                                "algorithmic_flops_per_launch": c["flops"] / c["launches"],
                                "avg_launch_us": c["seconds"] / c["launches"] * 1e6}
            line["gemm_census"] = {k: {"launches": v["launches"], "avg_launch_us": round(v["seconds"] / v["launches"] * 1e6, 2),
                                       "tflops": round(v["flops"] / v["seconds"] / 1e12, 1)} for k, v in census.items()}
        if kern:
            line["kernels"] = {k: {"ms": round(v["ms"], 4), "tflops": round(v["tflops"], 1),
                                   "mfma_frac": round(v["tflops"] / PEAK_BF16_TFLOPS, 4)} for k, v in kern.items()}
        if cpu:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
