"""Host side of the KV-cached greedy decode (reference models.py:267-323, eval_step).

One decode step is a fixed kernel sequence that reads its step index from device memory
(csrc/decode.hip), so the step is captured ONCE in a hipGraph and replayed; the reference's
early-stop test (models.py:306, a device->host sync per token) becomes a check of a device flag
every ``check_every`` replays.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L


class GreedyDecoder:
    def __init__(self, model, use_graph=None, check_every=16, strict_graph=False):
        """``strict_graph``: a failed hipGraph capture raises instead of falling back to eager launches (benchmarks must
        not silently measure the slow path; PLANK_DECODE_GRAPH=1 has the same effect)."""
        self.model = model
        self.check_every = check_every
        self.strict_graph = strict_graph
        if use_graph is None:
            use_graph = os.environ.get("PLANK_DECODE_GRAPH", "1") != "0"
        self.use_graph = use_graph
        self._ws = None
        self._graph = None
        self._graph_key = None
        self.last_steps = 0

    def _buffers(self, B, Tmax):
        m = self.model
        ptrs = [C.c_void_p() for _ in range(4)]
        L.check(L.lib().pa_decode_buffers(m._handle, *[C.byref(p) for p in ptrs]), "pa_decode_buffers")
        base = self._ws.data_ptr()

        def view(p, nbytes, dtype, shape):
            off = p.value - base
            return self._ws[off: off + nbytes].view(dtype).view(shape)

        return (view(ptrs[0], B * Tmax * 8, torch.int64, (B, Tmax)), view(ptrs[1], B * Tmax * 8, torch.int64, (B, Tmax)),
                view(ptrs[2], B * 4, torch.int32, (B,)), view(ptrs[3], 4, torch.int32, (1,)))

    def begin(self, batch, max_len=None):
        """Encoder + cross-K/V projection + state reset.  Returns (B, Tmax)."""
        m = self.model
        lib = L.lib()
        b, keep = m._make_batch(batch, with_output=False)
        b.T = 1
        Tmax = int(max_len or m.max_output_length)
        ws = m._workspace(b.B, b.S, 1)
        base = (ws.data_ptr() + 255) // 256 * 256
        stats = torch.empty(4, dtype=torch.float32, device=ws.device)
        L.check(lib.pa_model_train_fwd(m._handle, C.byref(b), C.c_void_p(base),
                                       C.c_int64(ws.numel() - (base - ws.data_ptr())), C.c_uint32(0), 0, L.ptr(stats),
                                       L.stream()), "pa_model_train_fwd(encoder)")
        need = int(lib.pa_decode_ws_bytes(m._handle, b.B, b.S, Tmax))
        if need < 0:
            L.check(need, "pa_decode_ws_bytes")
        if self._ws is None or self._ws.numel() < need + 256 or self._ws.device != ws.device:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=ws.device)
            self._graph = None
        dbase = (self._ws.data_ptr() + 255) // 256 * 256
        L.check(lib.pa_decode_begin(m._handle, C.c_void_p(dbase), C.c_int64(self._ws.numel() - (dbase - self._ws.data_ptr())),
                                    Tmax, L.stream()), "pa_decode_begin")
        self._keep = (b, keep, stats)
        shadow = m._shadow.data_ptr() if m._shadow is not None else 0
        key = (b.B, b.S, Tmax, m._flat.data_ptr(), shadow, self._ws.data_ptr())
        if key != self._graph_key:
            self._graph = None
            self._graph_key = key
        return b.B, Tmax

    def _step_eager(self):
        L.check(L.lib().pa_decode_step(self.model._handle, L.stream()), "pa_decode_step")

    def _capture(self):
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                self._step_eager()
        torch.cuda.current_stream().wait_stream(side)
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
        tokens, attach, first_end, t_dev = self._buffers(B, Tmax)
        done = 0
        n = Tmax
        while done < Tmax:
            k = min(self.check_every, Tmax - done) if early_stop else Tmax - done
            self.steps(k)
            done += k
            if early_stop:
                fe = first_end.cpu()
                if bool((fe >= 0).all()):
                    n = int(fe.max()) + 1
                    break
        self.last_steps = done
        return tokens[:, :n].clone(), attach[:, :n].clone()
