"""Does running Adam per backward segment on a side stream (as soon as the segment's gradient slice is final) hide the
optimizer's 160 us behind the remaining backward kernels?  ms/step, plain vs per-segment."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from plankassembly_amd import _lib as L
from plankassembly_amd.data import spec_for, synth_batch
from plankassembly_amd.optim import FusedAdam

model = bench.build("bf16", bench.S_IN + 1, bench.T_OUT, 0.2).train()
opt = FusedAdam(model, lr=1e-4)
batches = []
for s in range(4):
    b = synth_batch(16, spec_for("headline"), seed=2022 + s, device="cuda"); b.pop("name")
    batches.append(model.prepare_batch(b))
side = torch.cuda.Stream()
mode = {"overlap": False}
pending = []

def adam_slice(lo, hi, stream):
    flat, g = model.flat_params, model.flat_grads
    grp = opt.param_groups[0]
    sh = model._shadow
    L.check(L.lib().pa_adam_step(L.ptr(flat[lo:hi]), L.ptr(g[lo:hi]), L.ptr(opt._m[lo:hi]), L.ptr(opt._v[lo:hi]), L.ptr(sh[lo:hi]),
                                 C.c_int64(hi - lo), C.c_float(grp["lr"]), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8),
                                 opt._step, C.c_float(1.0), C.c_void_p(stream.cuda_stream)), "pa_adam_step")

def hook(seg, lo, hi):
    if not mode["overlap"]:
        return
    assert lo % 8 == 0 and (hi % 8 == 0 or hi == model.flat_params.numel()), (seg, lo, hi)
    side.wait_stream(torch.cuda.current_stream())
    adam_slice(lo, hi, side)

model.register_grad_ready_hook(hook)

def step(i):
    opt.zero_grad()
    if mode["overlap"]:
        opt._step += 1
    out = model(batches[i % 4]); out["loss"].backward()
    if mode["overlap"]:
        torch.cuda.current_stream().wait_stream(side)
        model.mark_shadow_fresh()
    else:
        opt.step()

def run(n):
    for i in range(10): step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): step(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

print("segments:", model.segment_slices())
step(0)   # allocates the moments
for rep in range(2):
    mode["overlap"] = False
    t0 = run(100)
    mode["overlap"] = True
    t1 = run(100)
    print(f"plain {t0:.3f} ms/step   per-segment Adam on a side stream {t1:.3f} ms/step")
