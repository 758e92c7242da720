"""Where the fresh-batch overhead of the train step comes from: the same 200 steps (a) on batches prepared ahead, (b) prepared one
step ahead on a side stream (DevicePrefetcher, the product loop), (c) as (b) with the valid-row count known on the host (no
device -> host read), (d) prepared on the MAIN stream right before the step with the host-known count (no side stream at all)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from plankassembly_amd.data import DevicePrefetcher, spec_for, synth_batch
from plankassembly_amd.optim import FusedAdam
steps = int(os.environ.get("STEPS", "200"))
model = bench.build("bf16", bench.S_IN + 1, bench.T_OUT, 0.2).train()
opt = FusedAdam(model, lr=1e-4)
raw = []
for i in range(16):
    b = synth_batch(16, spec_for("headline"), seed=2022 + 1000 * i, device="cuda"); b.pop("name")
    raw.append(b)
prepared = [model.prepare_batch(b) for b in raw]
hinted = [dict(b, _n_valid=int((~b["input_mask"]).sum())) for b in raw]


def step(batch):
    opt.zero_grad(); out = model(batch); out["loss"].backward(); opt.step()
    return out


def cyc(pool, n):
    for i in range(n):
        yield pool[i % len(pool)]


def run(name, it_fn):
    for rep in range(2):
        it = it_fn(steps + 10)
        for _ in range(10):
            step(next(it))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            step(next(it))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        print(f"{name:46s} {dt * 1e3:.3f} ms/step", flush=True)


run("(a) prepared ahead", lambda n: cyc(prepared, n))
run("(b) side stream, device->host row count", lambda n: DevicePrefetcher(model, cyc(raw, n)))
run("(c) side stream, host-known row count", lambda n: DevicePrefetcher(model, cyc(hinted, n)))
run("(d) main stream, host-known row count", lambda n: (model.prepare_batch(b) for b in cyc(hinted, n)))
run("(a) prepared ahead", lambda n: cyc(prepared, n))
