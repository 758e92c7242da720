"""The RCCL path on ONE GPU: a world-size-1 `nccl` process group runs the real PlankModel + GradSync + FusedAdam step.
No scaling can be measured on one device; what this pins is that the data-parallel machinery executes on the GPU
(segment hooks in backward order, async all-reduces of flat-buffer slices through RCCL, Adam behind the last
collective) and leaves gradients and parameters exactly as the plain step does."""
import os

import pytest
import torch
import torch.distributed as dist

from conftest import load_fixture
from test_model_gpu import make, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    if dist.is_initialized():
        yield
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,grad_dtype", [("f32", "f32"), ("bf16", "f32"), ("bf16", "bf16")])
def test_rccl_world_size_one_step_equals_plain_step(nccl_world1, dtype, grad_dtype):
    from plankassembly_amd.distributed import GradSync
    from plankassembly_amd.optim import FusedAdam
    sd, batch, _ = load_fixture("fixture_small.npz")
    gb = to_dev(batch)

    def run(with_sync):
        m = make(sd, dtype=dtype).train()
        opt = FusedAdam(m, lr=1e-3, grad_scale=1.0)
        sync = GradSync(m, grad_dtype=grad_dtype) if with_sync else None
        if sync is not None:
            sync.broadcast_parameters(0)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            out = m(m.prepare_batch(batch))
            out["loss"].backward()
            opt.step()
            losses.append(out["loss"].item())
        torch.cuda.synchronize()
        return m, sync, losses, m.flat_grads.clone(), m.flat_params.clone()

    m0, _, l0, g0, p0 = run(False)
    m1, sync, l1, g1, p1 = run(True)
    nseg = m1.num_encoder_layers + m1.num_decoder_layers + 4
    assert sync.fired == list(range(nseg)), sync.fired                      # hooks fire once per segment, in backward order
    covered = sorted(sync.launched)
    assert covered[0][0] == 0 and covered[-1][1] == m1._numel               # the slices tile the whole flat buffer
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert not sync._works                                                  # the last segment's hook waited for every collective
    if grad_dtype == "f32":
        # sum over one rank = identity; two runs of the step itself differ in the last bits (f32 atomics in the loss and
        # small-table reductions), so the comparison is to 1e-5 of the largest entry, not bitwise
        assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-5 * max(1.0, abs(l0[0]))
        assert float((g1 - g0).abs().max()) <= 1e-5 * float(g0.abs().max()) + 1e-9
        # (Adam normalises by |g|: on this memorised fixture's noise-level gradients a last-bit difference can move an entry
        # by a full step, so parameters are only bounded by the three steps of lr 1e-3 they can have taken)
        assert float((p1 - p0).abs().max()) <= 3.1e-3
    else:
        assert max(abs(a - b) for a, b in zip(l0, l1)) < 5e-3 * max(1.0, abs(l0[0]))
        rel = float((g1 - g0).norm() / (g0.norm() + 1e-30))
        assert rel < 5e-2, rel                                              # gradients rounded to bf16 for the exchange


def test_metric_sums_travel_through_the_device_under_an_rccl_only_group(nccl_world1):
    """Criterion sums are CPU float64; an RCCL-only group has no CPU backend."""
    from plankassembly_amd.distributed import allreduce_metric_sums
    from plankassembly_amd.metric import Criterion
    c = Criterion()
    c.update(1.0, 0.5, 2 / 3)
    p, r, f = c.compute(sync=True)
    assert abs(float(p) - 1.0) < 1e-12 and abs(float(r) - 0.5) < 1e-12
    v = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64)
    assert torch.equal(allreduce_metric_sums(v.clone()), v)


def test_reserved_cus_knob_keeps_results():
    import ctypes as C
    from plankassembly_amd import _lib as L, ops
    a = torch.randn(2048, 512, device="cuda").to(torch.bfloat16)
    b = torch.randn(512, 512, device="cuda").to(torch.bfloat16)
    ref = ops.gemm(a, b)
    assert L.lib().pa_get_reserved_cus() == 0
    L.check(L.lib().pa_set_reserved_cus(32), "pa_set_reserved_cus")
    try:
        assert L.lib().pa_get_reserved_cus() == 32
        out = ops.gemm(a, b)
    finally:
        L.check(L.lib().pa_set_reserved_cus(0), "pa_set_reserved_cus")
    assert torch.equal(out, ref)
