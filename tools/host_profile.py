import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
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
for i in range(10): step(pb[i % 8])
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for i in range(N): step(pb[i % 8])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/N:.2f} ms/step, total {1e3*(t2-t0)/N:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for i in range(N): step(pb[i % 8])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
