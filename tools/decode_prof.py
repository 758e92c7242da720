"""Short decode run for profiling: B=256, S=1024, graph replay of the step (LANES=2: two half-batch lanes); GRAPH=0 runs it eagerly."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from plankassembly_amd.data import spec_for, synth_batch
from plankassembly_amd.decode import GreedyDecoder
dm = bench.build(os.environ.get("DTYPE", "bf16"), 1025, 1024, 0.0).eval()
dm._ensure_handle(); dm._refresh_shadow()
dec = GreedyDecoder(dm, use_graph=os.environ.get("GRAPH", "1") != "0", strict_graph=True, lanes=int(os.environ.get("LANES", "1")))
db = synth_batch(int(os.environ.get("BATCH", "256")), spec_for("decode"), seed=7, device="cuda"); db.pop("name")
with torch.no_grad():
    B, T = dec.begin(db, 1024)
    dec.steps(int(os.environ.get("STEPS", "48")))
torch.cuda.synchronize()
print("done, lanes", dec._active)
