"""Data-parallel gradient exchange for PlankModel: one process per GPU, RCCL over xGMI.

The reference gets data parallelism implicitly from Lightning's ``strategy: ddp``
(configs/train_complete.yaml:18 -> torch DistributedDataParallel: 25 MB buckets of per-parameter
gradients, NCCL all-reduce).  Here the gradients already live in ONE flat f32 buffer ordered
[embeddings | encoder layers | encoder.norm | decoder layers | decoder.norm | heads], and the HIP
backward runs in segments that finish contiguous slices of it (heads first).  After each segment
the slice is handed to RCCL as a single large all-reduce that overlaps with the remaining
backward kernels - a few contiguous 8-13 MB collectives per step instead of hundreds of
per-parameter ones, sized for xGMI's per-link bound rather than for NVSwitch.

The sum is left un-averaged; FusedAdam applies 1/world_size inside its kernel (``grad_scale``).
Works on CPU tensors with the gloo backend too (that is how the CPU test-suite covers it).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model, process_group=None, coalesce_below=None, grad_dtype=None, reserve_cus=None):
        """``coalesce_below``: finished segments are merged until a run has at least this many elements, then sent as one
        all-reduce (encoder.norm / output-embedding slices are a few KB; a layer is 2.1 - 3.2 M elements).  None reads
        PLANK_SYNC_COALESCE, default 2 M elements (8 MB).

        ``grad_dtype``: 'f32' (default; PLANK_GRAD_DTYPE overrides) exchanges the f32 gradients as they are - 130 MB per
        step for the 32.5 M-parameter model; 'bf16' casts each finished slice to bf16, all-reduces 65 MB and widens the sums
        back before Adam (the reference's DDP exchanges f32; opt-in because the sum of bf16-rounded gradients differs in the
        last bits).

        ``reserve_cus``: CUs every persistent GEMM launch leaves free WHILE COLLECTIVES ARE IN FLIGHT (pa_set_reserved_cus from
        the first slice that is sent until wait() has seen the last one), when the process group is larger than one rank.  The
        persistent GEMM grids are one block per CU with the whole register file of their SIMDs: a CU that hosts one of RCCL's
        long-running blocks cannot take a GEMM block.  Whether leaving room helps depends on the launch: a multi-unit grid (the
        grouped weight gradients, the two-blocks-per-CU kernel) re-balances over the CUs it is given, but most of the model's
        Linears are ONE round of ~250 tiles, and a grid of 224 blocks turns those into two rounds whether or not a collective
        happens to be resident.  None reads PA_RESERVE_CUS; default 0 (off).  UNTUNED: no multi-GPU node was available; measure
        0 / 16 / 32 on the target box."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        import os
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.slices = model.segment_slices()
        self.nseg = len(self.slices)
        if coalesce_below is None:
            coalesce_below = int(float(os.environ.get("PLANK_SYNC_COALESCE", 2 * 1024 * 1024)))
        self.coalesce_below = coalesce_below
        self.grad_dtype = grad_dtype or os.environ.get("PLANK_GRAD_DTYPE", "f32")
        if self.grad_dtype not in ("f32", "bf16"):
            raise ValueError("grad_dtype must be 'f32' or 'bf16'")
        self._lp = None               # bf16 staging buffer of the flat gradients (grad_dtype == 'bf16')
        self._works = []
        self._to_widen = []           # [(target buffer, lo, hi)] bf16 slices in flight that wait() casts back exactly once
        self._pending = None          # (lo, hi) run of contiguous finished-but-unsent slices
        self.launched = []            # [(lo, hi)] of the last backward, for tests / introspection
        self.fired = []               # segment indices in the order the hook saw them (tests)
        # One-GPU rehearsal (PLANK_FAKE_COLLECTIVE="<blocks>:<GB/s>[:<ranks>]", e.g. "32:200:8"; only with a one-rank group on the
        # GPU): every slice that would be all-reduced instead launches pa_fake_collective on a side stream - `blocks` CUs held for
        # 2 (ranks - 1) / ranks * bytes / (GB/s) + 20 us, the slice streamed through HBM twice - so the interplay of the
        # persistent GEMM grids with a resident collective (and PA_RESERVE_CUS) can be measured without an 8-GPU node.
        self.fake = None
        spec = os.environ.get("PLANK_FAKE_COLLECTIVE", "")
        if spec and spec != "0" and self.world == 1 and model.flat_params.is_cuda:
            parts = (spec.split(":") + ["", "", ""])[:3] if ":" in spec else ["32", "200", "8"]
            self.fake = (int(parts[0] or 32), float(parts[1] or 200.0), int(parts[2] or 8))
            self._fake_stream = torch.cuda.Stream()
            self.fake_launched = 0
        if reserve_cus is None:
            reserve_cus = int(os.environ.get("PA_RESERVE_CUS", "0"))
        self.reserve_cus = int(reserve_cus) if ((self.world > 1 or self.fake) and model.flat_params.is_cuda) else 0
        self._reserved = False
        self.reserve_log = []         # ('on' | 'off') transitions, for tests
        model.register_grad_ready_hook(self._on_segment)

    def _reserve(self, on):
        if self.reserve_cus <= 0 or on == self._reserved:
            return
        from . import _lib as L
        L.check(L.lib().pa_set_reserved_cus(self.reserve_cus if on else 0), "pa_set_reserved_cus")
        self._reserved = on
        self.reserve_log.append("on" if on else "off")

    # the flat-buffer order is the reverse of the backward order, so finished slices extend DOWNWARDS
    def _on_segment(self, seg, lo, hi):
        if seg == 0:
            # A backward that was abandoned half way leaves nothing behind: its collectives are drained, but its bf16 staging
            # sums are DROPPED, not widened - this backward has already written segment 0's fresh gradient into the buffer
            # those stale sums would land in.
            self._to_widen = []
            self.wait()
            self._works, self.launched, self._pending, self.fired = [], [], None, []
        self.fired.append(seg)
        if self._pending is None:
            self._pending = (lo, hi)
        elif hi == self._pending[0]:
            self._pending = (lo, self._pending[1])
        elif lo == self._pending[1]:
            self._pending = (self._pending[0], hi)
        else:                               # not adjacent: flush what we have, start a new run
            self._flush()
            self._pending = (lo, hi)
        last = seg == self.nseg - 1
        if last or (self._pending[1] - self._pending[0]) >= self.coalesce_below:
            self._flush()
        if last:
            self.wait()

    def _flush(self):
        if self._pending is None:
            return
        lo, hi = self._pending
        self._pending = None
        if hi <= lo:
            return
        g = getattr(self.model, "grad_sync_buffer", None)
        if g is None:
            g = self.model.flat_grads
        # ProcessGroupNCCL (= RCCL on ROCm) orders the collective after the work already enqueued on
        # the current stream and runs it on its own stream: the remaining backward kernels overlap.
        buf = g[lo:hi]
        if self.grad_dtype == "bf16" and g.is_cuda:
            if self._lp is None or self._lp.numel() != g.numel() or self._lp.device != g.device:
                self._lp = torch.empty(g.numel(), dtype=torch.bfloat16, device=g.device)
            buf = self._lp[lo:hi]
            self._cast(buf, g[lo:hi])
            self._to_widen.append((g, lo, hi))      # the buffer this slice came from: it may differ by the time of wait()
        self._reserve(True)                 # from here until wait(): the GEMM grids leave room for the collective's blocks
        if self.fake is not None:
            self._fake_flush(buf)
        else:
            self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.launched.append((lo, hi))

    def _fake_flush(self, buf):
        import ctypes as C
        from . import _lib as L
        blocks, gbps, ranks = self.fake
        nbytes = buf.numel() * buf.element_size() // 16 * 16
        us = 2.0 * (ranks - 1) / ranks * nbytes / (gbps * 1e3) + 20.0
        side = self._fake_stream
        side.wait_stream(torch.cuda.current_stream())          # like ProcessGroupNCCL: after the work already enqueued, on its own stream
        L.check(L.lib().pa_fake_collective(L.ptr(buf), C.c_int64(nbytes), blocks, 2, C.c_float(us), C.c_void_p(side.cuda_stream)),
                "pa_fake_collective")
        self.fake_launched += 1

    @staticmethod
    def _cast(dst, src):
        import ctypes as C
        from . import _lib as L
        L.check(L.lib().pa_cast(L.ptr(dst), L.dt(dst), L.ptr(src), L.dt(src), C.c_int64(src.numel()), L.stream()), "pa_cast")

    def wait(self):
        """Block (stream-level on the GPU) until every collective launched so far has finished; bf16 exchanges are widened
        back into the buffer they were taken from.  Idempotent: a second call finds nothing to do, so it can neither
        re-cast stale staging data nor - with gradient accumulation, where the slices come from the fresh micro-batch
        buffer and not from the accumulated one - write into a different buffer than the one that was reduced."""
        for w in self._works:
            w.wait()                        # stream-level wait on GPU, blocking on gloo
        self._works = []
        if self.fake is not None:
            torch.cuda.current_stream().wait_stream(self._fake_stream)
        self._reserve(False)
        widen, self._to_widen = self._to_widen, []
        for g, lo, hi in widen:             # the reduced sums, after the collectives and before Adam
            self._cast(g[lo:hi], self._lp[lo:hi])

    def detach(self):
        """Finish what is in flight and stop exchanging: later backward passes of this model are rank-local (a rank that
        goes on stepping alone - bench.py's rank-0 kernel census - must not enqueue collectives nobody else joins)."""
        self.wait()
        self._pending = None
        self.model.register_grad_ready_hook(None)

    def broadcast_parameters(self, src=0):
        """DDP constructor semantics: every rank starts from rank ``src``'s parameters."""
        dist.broadcast(self.model.flat_params, src=src, group=self.group)
        if hasattr(self.model, "invalidate_shadow"):
            self.model.invalidate_shadow()


def allreduce_metric_sums(values, group=None):
    """Sum a small vector of metric accumulators over ranks (reference plankassembly/metric.py:13-16,
    ``dist_reduce_fx='sum'``; trainer_complete.py:87-89 ``sync_dist=True``)."""
    if dist.is_available() and dist.is_initialized():
        # (world size 1 included: `torchrun --nproc-per-node 1` must exercise the same path as 8 ranks do)
        backend = str(dist.get_backend(group)).lower()
        if values.device.type == "cpu" and "nccl" in backend and "gloo" not in backend:
            # the trainer / bench initialise an RCCL-only group: it has no CPU backend, so the (CPU, float64) metric
            # accumulators travel through the current device
            dev = values.to(torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(dev, op=dist.ReduceOp.SUM, group=group)
            values.copy_(dev.cpu())
        else:
            dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
    return values
