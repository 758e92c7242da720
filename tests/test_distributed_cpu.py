"""world_size-2 gloo test of the gradient-exchange logic (no GPU): the same GradSync object that
drives RCCL on the GPU is fed the backward-segment hooks of a stand-in model holding CPU tensors."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plankassembly_amd.distributed import GradSync, allreduce_metric_sums
from plankassembly_amd.models import PlankModel
import types

TOKEN = types.SimpleNamespace(END=512, PAD=513)


class FakeModel:
    """Exposes the three things GradSync uses; slices come from a real PlankModel layout."""

    def __init__(self, rank):
        real = PlankModel(64, 4, 128, 0.0, "relu", True, 2, 2, 3, 2, 4, 6, 65, 36, 514, TOKEN)
        self._slices = real.segment_slices()
        n = real.flat_params.numel()
        g = torch.Generator().manual_seed(100 + rank)
        self.flat_grads = torch.randn(n, generator=g)
        self.flat_params = torch.full((n,), float(rank))
        self._hook = None

    def segment_slices(self):
        return self._slices

    def register_grad_ready_hook(self, fn):
        self._hook = fn

    def backward(self):
        for s, (lo, hi) in enumerate(self._slices):
            if self._hook is not None:
                self._hook(s, lo, hi)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, coalesce, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = FakeModel(rank)
        expect = sum(torch.randn(m.flat_grads.numel(), generator=torch.Generator().manual_seed(100 + r))
                     for r in range(world))
        sync = GradSync(m, coalesce_below=coalesce)
        sync.broadcast_parameters(0)
        assert float(m.flat_params.abs().max()) == 0.0            # rank 0's parameters everywhere
        m.backward()
        assert torch.allclose(m.flat_grads, expect, atol=1e-6)
        # launched slices are disjoint and cover the whole buffer exactly once
        cover = sorted(sync.launched)
        assert cover[0][0] == 0 and cover[-1][1] == m.flat_grads.numel()
        assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
        # second step reuses the object
        m.flat_grads = torch.ones_like(m.flat_grads)
        m.backward()
        assert torch.allclose(m.flat_grads, torch.full_like(m.flat_grads, float(world)))
        v = allreduce_metric_sums(torch.tensor([1.0 + rank, 2.0, 3.0, 1.0]))
        assert v.tolist() == [3.0, 4.0, 6.0, 2.0]
        # detach(): rank 0 goes on stepping ALONE (bench.py's rank-0 kernel census) - nothing may be exchanged any more,
        # or this would wait for a rank 1 that never comes
        sync.detach()
        if rank == 0:
            m.flat_grads = torch.full_like(m.flat_grads, 7.0)
            m.backward()
            assert float(m.flat_grads.min()) == 7.0 and float(m.flat_grads.max()) == 7.0
        dist.barrier()
        if rank == 0:
            out.put(len(sync.launched))
    finally:
        dist.destroy_process_group()


def _run(coalesce):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, coalesce, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return q.get(timeout=5)


def test_gradsync_gloo_world2_one_bucket_per_segment():
    assert _run(0) == 8            # 2+2 layers: heads, 2 dec, out-emb, enc.norm, 2 enc, in-emb


def test_gradsync_gloo_world2_coalesced():
    n = _run(60_000)
    assert 1 <= n < 8
