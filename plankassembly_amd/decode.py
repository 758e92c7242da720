"""Host side of the KV-cached greedy decode (reference models.py:267-323, eval_step).

One decode step is a fixed kernel sequence that reads its step index from device memory
(csrc/decode.hip), so the step is captured ONCE in a hipGraph and replayed; the reference's
early-stop test (models.py:306, a device->host sync per token) becomes a check of a device flag
every ``check_every`` replays.

Lanes.  A decode step alternates between HBM-bound kernels (single-query attention over the K/V
caches: ~60 % of the step) and 38 small Linears that are pure launch latency at M = batch rows
(~9 us each whatever they compute).  Samples are independent, so the batch is split into two
halves ("lanes") that run the same step on two streams inside one captured graph (fork / join):
while one lane sits in its latency-bound Linears the other streams its caches.  Each lane has its
own runtime handle (same parameter buffers), encoder workspace and decode arena; results are
concatenated.  ``lanes=1`` (or a batch too small to split) is the plain single-stream step.
Measured on MI355X (rocprofv3 kernel trace of the replayed graph, tools/overlap_from_db.py): kernels
of the two lanes are in flight together for only 6.6 % of the busy time - the dispatcher rarely
co-schedules them - so the gain is 1 % (187 k vs 185 k tokens/s); the fix for the latency-bound
Linears is fewer launches per step, not a second stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L


class _Lane:
    def __init__(self, model, own_handle):
        self.model = model
        self.own = own_handle
        self.handle = model.new_bound_handle() if own_handle else None
        self.enc_ws = None
        self.ws = None
        self.keep = None
        self.key = None
        self.stream = None

    def h(self):
        return self.handle if self.own else self.model._handle

    def close(self):
        if self.own and self.handle is not None:
            L.lib().pa_model_destroy(self.handle)
            self.handle = None

    def begin(self, batch, Tmax):
        """Encoder + cross-K/V projection + state reset for this lane's samples.  Returns B."""
        m, lib = self.model, L.lib()
        b, keep = m._make_batch(batch, with_output=False)
        b.T = 1
        need = int(lib.pa_model_train_ws_bytes(self.h(), b.B, b.S, 1))
        if need < 0:
            L.check(need, "pa_model_train_ws_bytes")
        dev = m.flat_params.device
        if self.enc_ws is None or self.enc_ws.numel() < need + 256:
            self.enc_ws = torch.empty(need + 512, dtype=torch.uint8, device=dev)
        base = (self.enc_ws.data_ptr() + 255) // 256 * 256
        stats = torch.empty(L.lib().pa_model_stats_floats(), dtype=torch.float32, device=dev)   # include/plank_hip.h: f32[8]
        L.check(lib.pa_model_train_fwd(self.h(), C.byref(b), C.c_void_p(base),
                                       C.c_int64(self.enc_ws.numel() - (base - self.enc_ws.data_ptr())), C.c_uint32(0), 0,
                                       L.ptr(stats), L.stream()), "pa_model_train_fwd(encoder)")
        need = int(lib.pa_decode_ws_bytes(self.h(), b.B, b.S, Tmax))
        if need < 0:
            L.check(need, "pa_decode_ws_bytes")
        fresh = False
        if self.ws is None or self.ws.numel() < need + 256:
            self.ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
            fresh = True
        dbase = (self.ws.data_ptr() + 255) // 256 * 256
        L.check(lib.pa_decode_begin(self.h(), C.c_void_p(dbase), C.c_int64(self.ws.numel() - (dbase - self.ws.data_ptr())),
                                    Tmax, L.stream()), "pa_decode_begin")
        self.keep = (b, keep, stats)
        shadow = m._shadow.data_ptr() if m._shadow is not None else 0
        key = (b.B, b.S, Tmax, m._flat.data_ptr(), shadow, self.ws.data_ptr())
        changed = fresh or key != self.key
        self.key = key
        return b.B, changed

    def step(self):
        L.check(L.lib().pa_decode_step(self.h(), L.stream()), "pa_decode_step")

    def buffers(self, B, Tmax):
        ptrs = [C.c_void_p() for _ in range(4)]
        L.check(L.lib().pa_decode_buffers(self.h(), *[C.byref(p) for p in ptrs]), "pa_decode_buffers")
        base = self.ws.data_ptr()

        def view(p, nbytes, dtype, shape):
            off = p.value - base
            return self.ws[off: off + nbytes].view(dtype).view(shape)

        return (view(ptrs[0], B * Tmax * 8, torch.int64, (B, Tmax)), view(ptrs[1], B * Tmax * 8, torch.int64, (B, Tmax)),
                view(ptrs[2], B * 4, torch.int32, (B,)))


def _split_batch(batch, lo, hi):
    out = {}
    for k, v in batch.items():
        if k.startswith("_"):
            continue                                   # packing / groupings are per sub-batch: recomputed by the lane
        out[k] = v[lo:hi] if (torch.is_tensor(v) or isinstance(v, list)) else v
    return out


class GreedyDecoder:
    def __init__(self, model, use_graph=None, check_every=16, strict_graph=False, lanes=None):
        """``strict_graph``: a failed hipGraph capture raises instead of falling back to eager launches (benchmarks must
        not silently measure the slow path; PLANK_DECODE_GRAPH=1 has the same effect).  ``lanes``: 1 or 2 (default 1,
        PLANK_DECODE_LANES overrides); batches of fewer than 32 samples always run as one lane."""
        self.model = model
        self.check_every = check_every
        self.strict_graph = strict_graph
        if use_graph is None:
            use_graph = os.environ.get("PLANK_DECODE_GRAPH", "1") != "0"
        self.use_graph = use_graph
        # one lane by default since round 3: with the K/V append folded into the attention kernel the single-stream step
        # (1.210 ms at B 256) is ahead of the two half-batch lanes (1.239 ms)
        self.max_lanes = int(lanes if lanes is not None else os.environ.get("PLANK_DECODE_LANES", "1"))
        # PLANK_DECODE_ALTERNATE=1: the lanes' attention launches strictly alternate (pa_decode_step_pair).  OFF: measured
        # 2.15 ms / step against 1.23 - 24 cross-queue event edges per step cost more than the overlap they arrange.
        self.alternate = os.environ.get("PLANK_DECODE_ALTERNATE", "0") == "1"
        self._lanes = []
        self._graph = None
        self._side = None
        self._active = 0
        self.last_steps = 0

    def __del__(self):
        try:
            for ln in self._lanes:
                ln.close()
        except Exception:
            pass

    def _lane(self, i):
        while len(self._lanes) <= i:
            self._lanes.append(_Lane(self.model, own_handle=len(self._lanes) > 0))
        return self._lanes[i]

    def begin(self, batch, max_len=None):
        """Encoder + cross-K/V projection + state reset.  Returns (B, Tmax)."""
        m = self.model
        Tmax = int(max_len or m.max_output_length)
        B = batch["input_value"].shape[0]
        n = 2 if (self.max_lanes >= 2 and B >= 32) else 1
        self._bounds = [(0, B)] if n == 1 else [(0, B // 2), (B // 2, B)]
        changed = n != self._active
        self._active = n
        if n == 2 and self._side is None:
            self._side = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        for i, (lo, hi) in enumerate(self._bounds):
            ln = self._lane(i)
            sub = batch if n == 1 else m.prepare_batch(_split_batch(batch, lo, hi), groups=False) if m.unpad else _split_batch(batch, lo, hi)
            if i == 0:
                _, ch = ln.begin(sub, Tmax)
            else:
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    _, ch = ln.begin(sub, Tmax)
                main.wait_stream(self._side)
            changed = changed or ch
        if changed:
            self._graph = None
        return B, Tmax

    def _step_eager(self):
        main = torch.cuda.current_stream()
        if self._active == 2:
            self._side.wait_stream(main)               # fork
            if self.alternate:
                # attention launches of the two lanes strictly alternate (pa_decode_step_pair): one lane streams its K/V
                # caches while the other runs the latency-bound launches in between
                L.check(L.lib().pa_decode_step_pair(self._lanes[0].h(), self._lanes[1].h(), C.c_void_p(main.cuda_stream),
                                                    C.c_void_p(self._side.cuda_stream)), "pa_decode_step_pair")
            else:
                with torch.cuda.stream(self._side):
                    self._lanes[1].step()
                self._lanes[0].step()
            main.wait_stream(self._side)               # join
        else:
            self._lanes[0].step()

    def _capture(self):
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap):
                self._step_eager()
        torch.cuda.current_stream().wait_stream(cap)
        return g

    def steps(self, n):
        """Enqueue n decode steps (graph replay when enabled)."""
        if self.use_graph and self._graph is None:
            try:
                self._graph = self._capture()
            except Exception as exc:                                  # pragma: no cover
                if self.strict_graph or os.environ.get("PLANK_DECODE_GRAPH") == "1":
                    raise
                print(f"[plankassembly_amd] hipGraph capture of the decode step failed ({exc}); running eagerly")
                self.use_graph = False
        for _ in range(n):
            if self.use_graph:
                self._graph.replay()
            else:
                self._step_eager()

    def run(self, batch, max_len=None, early_stop=True):
        """Full greedy decode.  Returns (samples int64 [B, n], attach int64 [B, n]) with the
        reference's early-stop length n."""
        B, Tmax = self.begin(batch, max_len)
        bufs = [self._lanes[i].buffers(hi - lo, Tmax) for i, (lo, hi) in enumerate(self._bounds)]
        done = 0
        n = Tmax
        while done < Tmax:
            k = min(self.check_every, Tmax - done) if early_stop else Tmax - done
            self.steps(k)
            done += k
            if early_stop:
                fe = torch.cat([b[2] for b in bufs]).cpu()
                if bool((fe >= 0).all()):
                    n = int(fe.max()) + 1
                    break
        self.last_steps = done
        tokens = torch.cat([b[0][:, :n] for b in bufs]) if len(bufs) > 1 else bufs[0][0][:, :n].clone()
        attach = torch.cat([b[1][:, :n] for b in bufs]) if len(bufs) > 1 else bufs[0][1][:, :n].clone()
        return tokens, attach
