"""Greedy-decode timing at the benchmarked size (B 256, S 1024, 1024 steps, graph replay) for 1 and 2 lanes; environment
toggles (PLANK_DECODE_FOLD_LN, PLANK_HIP_LIB, ...) are read by the library, so A/B runs are separate processes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from plankassembly_amd.data import spec_for, synth_batch
from plankassembly_amd.decode import GreedyDecoder
dtype = os.environ.get("DTYPE", "bf16")
for lanes in (2, 1):
    dm = bench.apply_gains(bench.build(dtype, 1025, 1024, 0.0), bench.DECODE_GAINS).eval()
    dm._ensure_handle(); dm._refresh_shadow()
    dec = GreedyDecoder(dm, use_graph=True, strict_graph=True, lanes=lanes)
    db = synth_batch(256, spec_for("decode"), seed=7, device="cuda"); db.pop("name")
    db = dm.prepare_batch(db)
    with torch.no_grad():
        dec.run(db, max_len=1024, early_stop=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        toks, _ = dec.run(db, max_len=1024, early_stop=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{dtype} lanes {lanes}: {dt / 1024 * 1e3:.3f} ms/step, {256 * 1024 / dt:.0f} tok/s, distinct tokens {len(torch.unique(toks))}", flush=True)
    del dec, dm
    torch.cuda.empty_cache()
