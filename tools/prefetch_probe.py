import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from plankassembly_amd.data import synth_batch, DevicePrefetcher, _tensors_of
import plankassembly_amd.data as D
from plankassembly_amd.optim import FusedAdam
c = bench.CONFIGS["headline"]
model = bench.build("bf16", c["max_in"], c["max_out"], 0.2, c).train()
opt = FusedAdam(model, lr=1e-4)
raw = []
for i in range(16):
    b = synth_batch(16, bench.cfg_spec(c), seed=2022 + 1000 * i, device="cuda"); b.pop("name"); raw.append(b)
pb = [model.prepare_batch(b) for b in raw]
def step(b):
    opt.zero_grad(); out = model(b); out["loss"].backward(); opt.step()
def run(name, it, n=30, warm=5):
    for _ in range(warm): step(next(it))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(next(it))
    torch.cuda.synchronize(); print(f"{name:40s} {1e3*(time.perf_counter()-t0)/n:.3f} ms/step")
def cyc(pool):
    i = 0
    while True:
        yield pool[i % len(pool)]; i += 1
which = sys.argv[1] if len(sys.argv) > 1 else "all"
run("prepared", cyc(pb))
if which in ("all", "side"): run("prefetcher (side stream)", DevicePrefetcher(model, cyc(raw)))
def inline():
    for b in cyc(raw): yield model.prepare_batch(b)
if which in ("all", "inline"): run("prepare inline on the main stream", inline())
# (a variant without record_stream faults - the allocator hands a block still read by the main stream to the next
# side-stream prepare: 'Write access to a read-only page' - which is why DevicePrefetcher records the consuming stream)
run("prepared again", cyc(pb))
