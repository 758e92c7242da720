import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from torch.profiler import profile, ProfilerActivity
from plankassembly_amd.data import spec_for, synth_batch
from plankassembly_amd.optim import FusedAdam
model = bench.build("bf16", bench.S_IN + 1, bench.T_OUT, 0.2).train()
opt = FusedAdam(model, lr=1e-4)
b = synth_batch(16, spec_for("headline"), seed=2022, device="cuda"); b.pop("name")
b = model.prepare_batch(b)
def step():
    opt.zero_grad(); out = model(b); out["loss"].backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
for e in prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=40).splitlines():
    if "copy" in e.lower() or "Memcpy" in e or "Name" in e or "fill" in e.lower() or "zero" in e.lower():
        print(e[:230])
