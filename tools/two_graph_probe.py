"""Probe (GPU-side only, no host in the loop): the whole training step of a batch-8 replica captured as a hipGraph, two such
graphs replayed on two streams at once, against one batch-16 graph.  Tells whether two independent kernel streams overlap
their per-launch latencies on the chip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from plankassembly_amd.data import synth_batch
from plankassembly_amd.optim import FusedAdam

c = bench.CONFIGS["headline"]
N = int(os.environ.get("N", "100"))


def capture(bsz, seed):
    m = bench.build("bf16", c["max_in"], c["max_out"], 0.2, c).train()
    o = FusedAdam(m, lr=1e-4)
    b = synth_batch(bsz, bench.cfg_spec(c), seed=seed, device="cuda"); b.pop("name")
    pb = m.prepare_batch(b)

    def step():
        o.zero_grad(); out = m(pb); out["loss"].backward(); o.step()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(5):
            step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    torch.cuda.synchronize()
    return g, s, (m, o, pb)


def timed(graphs, n):
    for g, s, _ in graphs:
        with torch.cuda.stream(s):
            for _ in range(3):
                g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for g, s, _ in graphs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


g16 = capture(16, 2022)
print(f"one graph, batch 16: {timed([g16], N):.3f} ms/step", flush=True)
ga = capture(8, 3022)
print(f"one graph, batch 8: {timed([ga], N):.3f} ms/step", flush=True)
gb = capture(8, 4022)
print(f"two graphs on two streams, batch 8 + 8: {timed([ga, gb], N):.3f} ms per pair", flush=True)
