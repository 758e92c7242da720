"""Host-side cost of one training step (time to ENQUEUE it) vs its GPU time, and of prepare_batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from plankassembly_amd.data import synth_batch
from plankassembly_amd.optim import FusedAdam
c = bench.CONFIGS["headline"]
model = bench.build("bf16", c["max_in"], c["max_out"], 0.2, c).train()
opt = FusedAdam(model, lr=1e-4)
raw = []
for i in range(8):
    b = synth_batch(16, bench.cfg_spec(c), seed=2022 + 1000 * i, device="cuda"); b.pop("name"); raw.append(b)
pb = [model.prepare_batch(b) for b in raw]
def step(b):
    opt.zero_grad(); out = model(b); out["loss"].backward(); opt.step()
for i in range(5): step(pb[i % 8])
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for i in range(N): step(pb[i % 8])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/N:.2f} ms/step, total {1e3*(t2-t0)/N:.2f} ms/step")
# host cost of prepare_batch alone (device idle)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N): model.prepare_batch(raw[i % 8])
torch.cuda.synchronize()
print(f"prepare_batch alone {1e3*(time.perf_counter()-t0)/N:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(20): model.prepare_batch(raw[i % 8])
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(12)
