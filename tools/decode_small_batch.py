"""Greedy-decode step time against the batch size (S 1024, graph replay, one lane): the reference evaluates with BATCH_SIZE 16
(configs/train_complete.yaml), the benchmark line is quoted at 256.  STEPS decode steps timed after a warm-up run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from plankassembly_amd.data import spec_for, synth_batch
from plankassembly_amd.decode import GreedyDecoder
dtype = os.environ.get("DTYPE", "bf16")
STEPS = int(os.environ.get("STEPS", "256"))
for B in [int(x) for x in os.environ.get("BATCHES", "16,64,256").split(",")]:
    dm = bench.apply_gains(bench.build(dtype, STEPS + 1, 1024, 0.0), bench.DECODE_GAINS).eval()
    dm._ensure_handle(); dm._refresh_shadow()
    dec = GreedyDecoder(dm, use_graph=True, strict_graph=True, lanes=1)
    db = synth_batch(B, spec_for("decode"), seed=7, device="cuda"); db.pop("name")
    db = dm.prepare_batch(db)
    with torch.no_grad():
        dec.run(db, max_len=STEPS, early_stop=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        toks, _ = dec.run(db, max_len=STEPS, early_stop=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{dtype} B {B:4d}: {dt / STEPS * 1e3:.3f} ms/step, {B * STEPS / dt:.0f} tok/s, {dt / STEPS / B * 1e6:.2f} us per token-row", flush=True)
    del dec, dm
    torch.cuda.empty_cache()
