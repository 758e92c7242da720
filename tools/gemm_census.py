#!/usr/bin/env python
"""Per-shape table of the GEMM launches of one training step at the bench workload (replayed under HIP events)."""
import collections
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from plankassembly_amd import _lib as L
from plankassembly_amd.data import spec_for, synth_batch
from plankassembly_amd.optim import FusedAdam

B = 16
model = bench.build("bf16", bench.S_IN + 1, bench.T_OUT, 0.2).train()
opt = FusedAdam(model, lr=1e-4)
b = synth_batch(B, spec_for("headline"), seed=2022, device="cuda"); b.pop("name")
b = model.prepare_batch(b)
def step(i):
    opt.zero_grad(); out = model(b); out["loss"].backward(); opt.step()
for _ in range(3): step(0)
lib = L.lib()
torch.cuda.synchronize(); lib.pa_gemm_record(1); step(0); torch.cuda.synchronize()
n = lib.pa_gemm_record(0); rec = (L.GemmArgs * n)(); n = lib.pa_gemm_recorded(C.cast(rec, C.c_void_p), n)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tab = collections.OrderedDict()
for i in range(n):
    a = rec[i]; ref = C.cast(C.byref(a), C.c_void_p)
    lib.pa_gemm(ref, st); e0.record()
    for _ in range(10): lib.pa_gemm(ref, st)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 10
    key = (a.M, a.N, a.K, a.batch, a.a_kcontig, a.b_kcontig, a.splitk, a.out_dtype, int(bool(a.R)), int(bool(a.aux)), a.relu, a.drop_p > 0)
    r = tab.setdefault(key, [0, 0.0]); r[0] += 1; r[1] += t
print(f"{'M':>6} {'N':>5} {'K':>6} {'b':>3} kc sk od R aux relu drop {'cnt':>4} {'us':>7} {'TF':>6} {'tot_us':>8}")
tot = 0
for k, (c, t) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * k[0] * k[1] * k[2] * k[3]
    print(f"{k[0]:6d} {k[1]:5d} {k[2]:6d} {k[3]:3d} {k[4]}{k[5]} {k[6]:2d} {k[7]:2d} {k[8]} {k[9]:3d} {k[10]:4d} {int(k[11]):4d} {c:4d} {t / c * 1e6:7.1f} {fl * c / t / 1e12:6.0f} {t * 1e6:8.1f}")
    tot += t
print(f"total {tot * 1e3:.3f} ms over {n} launches")
