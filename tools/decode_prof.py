"""Short decode run for profiling: B=256, S=1024, 64 steps eager (no graph) so rocprof sees every kernel."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from plankassembly_amd.data import spec_for, synth_batch
from plankassembly_amd.decode import GreedyDecoder
dm = bench.build("bf16", 1025, 1024, 0.0).eval()
dm._ensure_handle(); dm._refresh_shadow()
dec = GreedyDecoder(dm, use_graph=False)
db = synth_batch(256, spec_for("decode"), seed=7, device="cuda"); db.pop("name")
with torch.no_grad():
    B, T = dec.begin(db, 1024)
    # jump the step counter to t = 512 so the profiled steps see a half-full self K/V cache
    tokens, attach, first_end, t_dev = dec._buffers(B, T)
    dec.steps(8)
    t_dev.fill_(512)
    dec.steps(32)
torch.cuda.synchronize()
print("done")
