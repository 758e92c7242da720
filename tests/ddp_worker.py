"""Worker of tests/test_distributed_gpu.py::test_two_ranks_on_one_gpu_*: one of TWO processes that share cuda:0.

RCCL refuses two ranks on one device; gloo does not, and ProcessGroupGloo all-reduces / broadcasts device tensors (staged
through the host).  So on a 1-GPU box this is the real thing except for the transport: the real PlankModel on the GPU, the
real GradSync (segment hooks, async all-reduce per flat-buffer slice), the real FusedAdam with grad_scale = 1 / world.
Rank r trains on rows [r*B/2, (r+1)*B/2) of the batch.
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch
import torch.distributed as dist

TOKEN = types.SimpleNamespace(END=512, PAD=513)


def build(c, dtype):
    import large_cases as LC
    from plankassembly_amd.models import PlankModel
    m = PlankModel(c["d"], c["h"], c["ff"], 0.0, "relu", True, c["ne"], c["nd"], 3, 2, 4, 6, c["max_in"], c["max_out"], 514,
                   TOKEN, compute_dtype=dtype)
    m.load_state_dict(LC.case_state_dict(c))
    return m.cuda().train()


def half(batch, r, world):
    B = batch["input_value"].shape[0]
    n = B // world
    return {k: v[r * n:(r + 1) * n] for k, v in batch.items()}


def main(rank, world, port, case, dtype, grad_dtype, out_path):
    import large_cases as LC
    from plankassembly_amd import _lib as L
    from plankassembly_amd.distributed import GradSync, allreduce_metric_sums
    from plankassembly_amd.optim import FusedAdam
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    # PLANK_DDP_BACKEND=nccl (tests/test_distributed_gpu.py::test_rccl_two_devices_*, boxes with >= 2 GPUs): one device per
    # rank and the exchange over RCCL - the transport the product uses; default: both ranks on cuda:0 over gloo
    backend = os.environ.get("PLANK_DDP_BACKEND", "gloo")
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = LC.CASES[case]
        m = build(c, dtype)
        if rank != 0:                               # DDP constructor semantics: rank 0's parameters win
            with torch.no_grad():
                m.flat_params.add_(0.5)
            m.invalidate_shadow()
        sync = GradSync(m, grad_dtype=grad_dtype)
        sync.broadcast_parameters(0)
        p_start = m.flat_params.detach().cpu().clone()
        opt = FusedAdam(m, lr=1e-3, grad_scale=1.0 / world)
        mine = half(LC.case_batch(c, batch_size=4), rank, world)
        opt.zero_grad()
        out = m(m.prepare_batch(mine))
        out["loss"].backward()
        sync.wait()                                 # idempotent: the last segment's hook has already waited
        torch.cuda.synchronize()
        g = m.flat_grads.detach().cpu().clone()     # the SUM over ranks (Adam divides by the world size)
        opt.step()
        torch.cuda.synchronize()
        sums = allreduce_metric_sums(torch.tensor([1.0 + rank, 2.0, 3.0, 1.0], dtype=torch.float64))
        torch.save({"loss": float(out["loss"]), "grads": g, "p_start": p_start, "params": m.flat_params.detach().cpu().clone(),
                    "launched": list(sync.launched), "fired": list(sync.fired), "sums": sums,
                    "reserve_log": list(sync.reserve_log), "reserved_after": int(L.lib().pa_get_reserved_cus()),
                    "backend": dist.get_backend(), "device": torch.cuda.current_device()}, f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]), int(a[1]), int(a[2]), a[3], a[4], a[5], a[6])
