"""N training steps of the headline model in one compute dtype (DTYPE = bf16 | f32 | x3, STEPS, BATCH) on cycling prepared
batches - the thing to put under `rocprofv3 --kernel-trace --stats` when one dtype's kernel mix is wanted alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from plankassembly_amd.data import spec_for, synth_batch
from plankassembly_amd.optim import FusedAdam
dtype, steps, B = os.environ.get("DTYPE", "x3"), int(os.environ.get("STEPS", "12")), int(os.environ.get("BATCH", "16"))
model = bench.build(dtype, bench.S_IN + 1, bench.T_OUT, 0.2).train()
opt = FusedAdam(model, lr=1e-4)
pool = []
for i in range(4):
    b = synth_batch(B, spec_for("headline"), seed=2022 + i, device="cuda"); b.pop("name")
    pool.append(model.prepare_batch(b))


def step(i):
    opt.zero_grad(); out = model(pool[i % len(pool)]); out["loss"].backward(); opt.step()
    return out


for i in range(3):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    out = step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{dtype}: {dt * 1e3:.3f} ms/step, {B / dt:.1f} samples/s, loss {float(out['loss']):.4f}")
