"""Probe: two HALF-batch training steps side by side on two HIP streams of ONE process (two model replicas, two host
threads) against one full-batch step.  Per-launch serial latencies (kernel ramp, first K tile, epilogue, drain) are ~55 % of
the batch-16 step (step(B) = 2.7 ms + 0.14 ms x B, profiles/r04_step_floor_probes.txt); if the blocks of two independent
kernel streams share the CUs, those latencies overlap.  Timing probe only (the replicas do not exchange gradients)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from plankassembly_amd.data import synth_batch
from plankassembly_amd.optim import FusedAdam

c = bench.CONFIGS["headline"]
N = int(os.environ.get("N", "150"))


def make(bsz, n_pool=8):
    m = bench.build("bf16", c["max_in"], c["max_out"], 0.2, c).train()
    o = FusedAdam(m, lr=1e-4)
    pool = []
    for i in range(n_pool):
        b = synth_batch(bsz, bench.cfg_spec(c), seed=2022 + 1000 * i + bsz, device="cuda"); b.pop("name")
        pool.append(m.prepare_batch(b))
    return m, o, pool


def run(m, o, pool, n, stream):
    with torch.cuda.stream(stream):
        for i in range(n):
            o.zero_grad()
            out = m(pool[i % len(pool)])
            out["loss"].backward()
            o.step()


def timed(workers, n):
    for w in workers:                       # warm-up, one after the other
        run(*w, 10, w[3]) if False else run(w[0], w[1], w[2], 10, w[3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(w[0], w[1], w[2], n, w[3])) for w in workers]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


full = make(16) + (torch.cuda.Stream(),)
dt = timed([full], N)
print(f"one stream,  batch 16: {dt / N * 1e3:.3f} ms/step  {16 * N / dt:.0f} samples/s", flush=True)
h1 = make(8) + (torch.cuda.Stream(),)
dt1 = timed([h1], N)
print(f"one stream,  batch  8: {dt1 / N * 1e3:.3f} ms/step  {8 * N / dt1:.0f} samples/s", flush=True)
h2 = make(8) + (torch.cuda.Stream(),)
dt2 = timed([h1, h2], N)
print(f"two streams, batch 8 + 8: {dt2 / N * 1e3:.3f} ms per pair of steps  {16 * N / dt2:.0f} samples/s", flush=True)
q = [make(4) + (torch.cuda.Stream(),) for _ in range(4)]
dt4 = timed(q, N)
print(f"four streams, batch 4 x 4: {dt4 / N * 1e3:.3f} ms per four steps  {16 * N / dt4:.0f} samples/s", flush=True)
