"""Batch preparation on an otherwise idle GPU: pa_pack_rows (+ the host's read of the row count) and pa_group_rows timed with HIP
events - what the launches cost by themselves (inside the step group_rows shows ~100 us in the trace).  python tools/prep_alone.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from plankassembly_amd.data import synth_batch

c = B.CONFIGS["headline"]
model = B.build("bf16", c["max_in"], c["max_out"], 0.2, c).train()
dev = [synth_batch(c["batch"], B.cfg_spec(c), seed=2022 + 1000 * i, device="cuda") for i in range(4)]
for b in dev:
    model.prepare_batch(b)
torch.cuda.synchronize()
for groups in (False, True):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(40):
        model.prepare_batch(dev[i % 4], groups=groups)
    e1.record(); e1.synchronize()
    print(f"prepare_batch(groups={groups}) on resident tensors, idle GPU: {e0.elapsed_time(e1) / 40 * 1e3:.1f} us per batch")
# the grouping launch alone, back to back (no host read between the launches)
p = [model.prepare_batch(b, groups=False) for b in dev]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(40):
    cu, rowmap, n = p[i % 4]["_pack"]
    model._group_rows(p[i % 4], rowmap, n)
e1.record(); e1.synchronize()
print(f"pa_group_rows alone, back to back: {e0.elapsed_time(e1) / 40 * 1e3:.1f} us per launch")
