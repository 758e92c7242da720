"""Fused Adam over the model's flat parameter buffer (one HIP kernel per step).

Same update rule and defaults as ``torch.optim.Adam(model.parameters(), lr=cfg.LR)`` used by the
reference (trainer_complete.py:127-129): betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad.
Also refreshes the model's bf16 GEMM-operand shadow in the same pass.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.model = model
        self.grad_scale = grad_scale
        self._step = 0
        self._m = None
        self._v = None

    def zero_grad(self, set_to_none: bool = True):
        for p in self.model._params.values():
            p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        m = self.model
        flat, g = m.flat_params, m.flat_grads
        if self._m is None or self._m.device != flat.device:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
        self._step += 1
        grp = self.param_groups[0]
        shadow = m._shadow if m.compute_dtype == "bf16" else None
        m.wait_transposed()                      # (the side-stream W^T refresh of the step before reads what this kernel rewrites)
        L.check(L.lib().pa_adam_step(L.ptr(flat), L.ptr(g), L.ptr(self._m), L.ptr(self._v), L.ptr(shadow),
                                     C.c_int64(flat.numel()), C.c_float(grp["lr"]), C.c_float(grp["betas"][0]),
                                     C.c_float(grp["betas"][1]), C.c_float(grp["eps"]), self._step,
                                     C.c_float(self.grad_scale), L.stream()), "pa_adam_step")
        # (the in-place update through the C ABI does not bump torch's version counters)
        if shadow is not None:
            m.mark_shadow_fresh()
        else:
            m.invalidate_shadow()
        return loss

    def state_dict(self):
        """Flat moments + step (CPU tensors: checkpoint payload)."""
        cpu = lambda t: None if t is None else t.detach().cpu()
        return {"step": self._step, "m": cpu(self._m), "v": cpu(self._v),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    # ---- torch.optim.Adam <-> flat layout (Lightning checkpoints store `optimizer_states` in torch's format)
    def _trainable(self):
        """(name, parameter) in ``model.parameters()`` order = the index order of torch.optim.Adam's state."""
        return [(k, p) for k, p in self.model.named_parameters() if p.requires_grad]

    def torch_state_dict(self):
        """The state as ``torch.optim.Adam(model.parameters()).state_dict()`` would hold it (CPU tensors)."""
        state = {}
        if self._m is not None and self._step > 0:
            m, v = self._m.detach().cpu(), self._v.detach().cpu()
            for i, (k, p) in enumerate(self._trainable()):
                off, n = self.model._offsets[k], p.numel()
                state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": m[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": v[off:off + n].view(p.shape).clone()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "params": list(range(len(self._trainable())))}
        return {"state": state, "param_groups": [group]}

    def load_torch_state_dict(self, sd):
        """Inverse of :meth:`torch_state_dict`: accepts the ``optimizer_states[0]`` entry of a Lightning checkpoint written
        with the reference's torch.optim.Adam (trainer_complete.py:127-129)."""
        flat = self.model.flat_params
        self._m = torch.zeros_like(flat)
        self._v = torch.zeros_like(flat)
        steps = set()
        names = self._trainable()
        for i, st in sd.get("state", {}).items():
            k, p = names[int(i)]
            off, n = self.model._offsets[k], p.numel()
            self._m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(flat.device, torch.float32))
            self._v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(flat.device, torch.float32))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a plain Adam state")
        self._step = steps.pop() if steps else 0
        for g, sg in zip(self.param_groups, sd.get("param_groups", [])):
            g["lr"], g["betas"], g["eps"] = sg.get("lr", g["lr"]), tuple(sg.get("betas", g["betas"])), sg.get("eps", g["eps"])

    def load_state_dict(self, sd):
        if "state" in sd:                                   # torch.optim.Adam layout
            return self.load_torch_state_dict(sd)
        dev = self.model.flat_params.device
        self._step = int(sd["step"])
        self._m = None if sd.get("m") is None else sd["m"].to(dev, torch.float32).clone()
        self._v = None if sd.get("v") is None else sd["v"].to(dev, torch.float32).clone()
        for g, sg in zip(self.param_groups, sd.get("param_groups", [])):
            for k in ("lr", "betas", "eps"):
                if k in sg:
                    g[k] = tuple(sg[k]) if k == "betas" else sg[k]
