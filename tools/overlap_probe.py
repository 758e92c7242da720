#!/usr/bin/env python
"""Does independent GEMM work on a second stream overlap with the training step?  Times (a) the step alone, (b) a replay
of the step's weight-gradient GEMMs alone, (c) both at once on two streams."""
import ctypes as C
import os
import sys
import time
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
def step():
    opt.zero_grad(); out = model(b); out["loss"].backward(); opt.step()
for _ in range(3): step()
lib = L.lib()
torch.cuda.synchronize(); lib.pa_gemm_record(1); step(); torch.cuda.synchronize()
n = lib.pa_gemm_record(0); rec = (L.GemmArgs * n)(); n = lib.pa_gemm_recorded(C.cast(rec, C.c_void_p), n)
# private operands for the replayed dW GEMMs (so the replay cannot race with the step): same shapes
dws = []
keep = []
for i in range(n):
    a = rec[i]
    if a.a_kcontig or a.b_kcontig or a.batch != 1: continue
    A = torch.randn(a.K, a.lda, device="cuda").to(torch.bfloat16); Bm = torch.randn(a.K, a.ldb, device="cuda").to(torch.bfloat16)
    Cc = torch.empty(a.M, a.N, device="cuda"); ws = torch.empty(max(1, a.splitk) * a.M * a.N, device="cuda")
    g = L.GemmArgs(); C.memmove(C.byref(g), C.byref(a), C.sizeof(g))
    g.A, g.B, g.C, g.ws, g.ldc = A.data_ptr(), Bm.data_ptr(), Cc.data_ptr(), ws.data_ptr(), a.N
    dws.append(g); keep += [A, Bm, Cc, ws]
print(len(dws), "dW gemms")
side = torch.cuda.Stream()
def replay(stream):
    st = C.c_void_p(stream.cuda_stream)
    for g in dws: lib.pa_gemm(C.cast(C.byref(g), C.c_void_p), st)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
t_step = timeit(step)
t_dw = timeit(lambda: replay(torch.cuda.current_stream()))
def both():
    replay(side); step()
t_both = timeit(both)
print(f"step alone {t_step:.2f} ms, dW replay alone {t_dw:.2f} ms, both on two streams {t_both:.2f} ms (sum {t_step + t_dw:.2f})")
