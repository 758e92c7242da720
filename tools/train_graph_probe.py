"""Probe: what would replaying the WHOLE training step from a hipGraph buy?  One prepared batch, the step (zero_grad +
forward + backward + Adam) captured once with torch.cuda.CUDAGraph and replayed, against the same step enqueued eagerly.
Also prints the host's enqueue time per eager step (how far ahead of the GPU the host runs).  Timing probe only: the
replayed step re-uses the captured dropout seed and Adam bias correction."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from plankassembly_amd.data import synth_batch
from plankassembly_amd.optim import FusedAdam

c = bench.CONFIGS["headline"]
model = bench.build("bf16", c["max_in"], c["max_out"], 0.2, c).train()
opt = FusedAdam(model, lr=1e-4)
N = int(os.environ.get("N", "100"))
for seed in (2022, 3022, 4022):
    b = synth_batch(16, bench.cfg_spec(c), seed=seed, device="cuda"); b.pop("name")
    pb = model.prepare_batch(b)

    def step():
        opt.zero_grad()
        out = model(pb)
        out["loss"].backward()
        opt.step()
        return out

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    eager = (t2 - t0) / N * 1e3
    print(f"batch seed {seed}: rows {pb['_pack'][2]}  eager {eager:.3f} ms/step (host enqueue {(t1 - t0) / N * 1e3:.3f} ms/step)", flush=True)
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = step()
        torch.cuda.synchronize()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            g.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / N * 1e3
        print(f"                 graph replay {rep:.3f} ms/step  ({(eager - rep) * 1e3:.0f} us saved, loss {float(out['loss']):.4f})", flush=True)
        del g
    except Exception as e:                                          # noqa: BLE001
        print("graph capture failed:", repr(e)[:400], flush=True)
        torch.cuda.synchronize()
