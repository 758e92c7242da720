"""MI355X-native drop-in for the reference's ``plankassembly/models.py``.

Same surface as the reference (reference models.py:11-76, 325-343):

* ``build_model(cfg)`` -> ``nn.Module``; ``cfg.MODEL.* / cfg.DATA.* / cfg.TOKEN`` as in the reference;
* ``module.forward(batch)``: training -> ``{'loss', 'accuracy'}``, eval -> ``{'samples', 'attach',
  'predicts', 'groundtruths'}``;
* ``state_dict()`` keys and shapes identical to the reference's torch modules (SURVEY.md appendix
  A), so reference / published checkpoints load unchanged; parameters are real ``nn.Parameter`` s
  whose ``.grad`` is populated by ``loss.backward()``.

Behind that surface nothing of torch.nn is used: the whole train step (embeddings, 6+6 post-norm
transformer layers, heads, mixture NLL) and its backward run as hand-written HIP kernels sequenced
by the C++ runtime in ``csrc/runtime.hip`` through the C ABI in ``include/plank_hip.h``; greedy
decoding runs the KV-cached decode kernels.  There is no CPU or eager fallback: the module only
works on a ROCm device with ``libplank_hip.so`` built.

MI355X-first layout: all parameters are views into ONE flat f32 buffer (and gradients into one
flat f32 buffer, bf16 GEMM operands into one flat shadow buffer) so that Adam is a single fused
kernel and the data-parallel gradient exchange is a handful of large contiguous RCCL all-reduces
issued per backward segment (see ``plankassembly_amd/distributed.py``).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L


# ------------------------------------------------------------------------------------------------
class _ModelCfg(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("n_head", C.c_int32), ("d_ff", C.c_int32), ("n_enc", C.c_int32),
                ("n_dec", C.c_int32), ("vocab", C.c_int32), ("out_dof", C.c_int32), ("in_table_rows", C.c_int32 * 5),
                ("eps_layer", C.c_float), ("eps_final", C.c_float), ("has_enc_norm", C.c_int32),
                ("dropout", C.c_float), ("pad", C.c_int32), ("end", C.c_int32), ("dtype", C.c_int32),
                ("activation", C.c_int32)]


class _TrDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32),
                ("ld_src", C.c_int32), ("ld_dst", C.c_int32), ("tile_begin", C.c_int32), ("pad_", C.c_int32)]


class _Batch(C.Structure):
    _fields_ = [("input_idx", C.c_void_p * 5), ("input_mask", C.c_void_p), ("output_value", C.c_void_p),
                ("output_label", C.c_void_p), ("output_mask", C.c_void_p),
                ("B", C.c_int32), ("S", C.c_int32), ("T", C.c_int32),
                ("cu_in", C.c_void_p), ("rowmap", C.c_void_p), ("n_valid", C.c_int32),
                ("in_order", C.c_void_p * 5), ("in_seg", C.c_void_p * 5),
                ("out_order", C.c_void_p * 3), ("out_seg", C.c_void_p * 3)]


INPUT_KEYS = ("input_value", "input_pos", "input_coord", "input_view", "input_type")


def group_rows_by_id(ids, rows, row_index=None):
    """torch statement of what pa_group_rows computes (kept as the checker of tests/test_model_gpu.py; the product path
    calls the HIP kernel).  Group row indices by embedding-table row (pa_embed_segment_bwd's input).  ids[i] = table row used by the i-th
    entry; row_index[i] = the gradient row that entry reads (default i).  Returns (order int32 [len(ids)] = gradient rows
    sorted by id, seg int32 [rows + 1] with seg[r] .. seg[r+1] = the slice of `order` that uses table row r)."""
    order = torch.argsort(ids, stable=True)
    if row_index is not None:
        order = row_index[order]
    seg = torch.zeros(rows + 1, dtype=torch.int32, device=ids.device)
    seg[1:] = torch.cumsum(torch.bincount(ids, minlength=rows)[:rows], 0).to(torch.int32)
    return order.to(torch.int32).contiguous(), seg


def param_order(n_enc: int, n_dec: int):
    """Canonical parameter order = the reference's state_dict order (csrc/runtime.hip enums)."""
    keys = [f"input_embeddings.{k}.weight" for k in INPUT_KEYS]
    keys += ["query_coord_embedding.weight", "query_pos_embedding.weight"]
    for i in range(n_enc):
        p = f"encoder.layers.{i}."
        keys += [p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias", p + "self_attn.out_proj.weight",
                 p + "self_attn.out_proj.bias", p + "linear1.weight", p + "linear1.bias", p + "linear2.weight",
                 p + "linear2.bias", p + "norm1.weight", p + "norm1.bias", p + "norm2.weight", p + "norm2.bias"]
    keys += ["encoder.norm.weight", "encoder.norm.bias"]
    for i in range(n_dec):
        p = f"decoder.layers.{i}."
        keys += [p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias", p + "self_attn.out_proj.weight",
                 p + "self_attn.out_proj.bias", p + "multihead_attn.in_proj_weight", p + "multihead_attn.in_proj_bias",
                 p + "multihead_attn.out_proj.weight", p + "multihead_attn.out_proj.bias", p + "linear1.weight",
                 p + "linear1.bias", p + "linear2.weight", p + "linear2.bias", p + "norm1.weight", p + "norm1.bias",
                 p + "norm2.weight", p + "norm2.bias", p + "norm3.weight", p + "norm3.bias"]
    keys += ["decoder.norm.weight", "decoder.norm.bias"]
    keys += ["vocab_head.weight", "vocab_head.bias", "pointer_head.weight", "pointer_head.bias",
             "switch_head.weight", "switch_head.bias"]
    return keys


class _Holder(nn.Module):
    """Pure parameter container (gives the reference's dotted state_dict names)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container; the fused HIP path does the arithmetic")


def _xavier_(t):
    if t.dim() > 1:
        nn.init.xavier_uniform_(t)


class _TrainStepFn(torch.autograd.Function):
    """loss = stats[0] / stats[1]; backward drives the HIP backward segments and fills the flat
    gradient buffer (parameter .grad tensors are views of it)."""

    @staticmethod
    def forward(ctx, hook, model, stats):
        ctx.model = model
        # stats[4] = loss, written by the forward's last kernel (pa_mixture_nll_fwd_fin): a view, no element-wise launch.  `hook`
        # (a leaf that requires grad) is what makes this node part of the autograd graph.
        return stats[4].detach()

    @staticmethod
    def backward(ctx, gloss):
        ctx.model._run_backward(gloss)
        return None, None, None


class PlankModel(nn.Module):
    """See module docstring.  Constructor signature = reference models.py:13-29 plus
    ``compute_dtype`` ('f32' parity path / 'bf16' throughput path / 'x3': the f32 path with every matrix product of the
    training step on the bf16 matrix pipe as a three-term hi / lo split - f32-accurate to ~2^-17 per product, see
    include/plank_hip.h pa_gemm_split_config; parameters, activations, LayerNorm, softmax statistics, the loss and the
    greedy decode are exactly the f32 path's)."""
    _x3_scratch = {}            # (rounds 4-5: one scratch per device shared by every 'x3' model; now each model owns its context and scratch)

    def __init__(self, num_model=512, num_head=8, num_feedforward=1024, dropout=0.1, activation="relu",
                 normalize_before=True, num_encoder_layers=6, num_decoder_layers=6, num_view=3, num_type=2,
                 num_input_dof=4, num_output_dof=6, max_input_length=400, max_output_length=128, vocab_size=514,
                 token=None, compute_dtype=None):
        super().__init__()
        # the reference hands the string to torch's Transformer layers (models.py:60-61,66-67), which take "relu" or "gelu"
        if activation not in ("relu", "gelu"):
            raise ValueError(f"ACTIVATION must be 'relu' or 'gelu' (torch's _get_activation_fn), got {activation!r}")
        self.activation = activation
        # run the encoder on the valid (non-PAD) rows only; PLANK_UNPAD=0 keeps the dense layout
        self.unpad = os.environ.get("PLANK_UNPAD", "1") != "0"
        compute_dtype = compute_dtype or os.environ.get("PLANK_COMPUTE_DTYPE", "f32")
        if compute_dtype not in ("f32", "bf16", "x3"):
            raise ValueError("compute_dtype must be 'f32', 'bf16' or 'x3'")
        self.compute_mode = compute_dtype                      # what the caller asked for
        self.split3 = compute_dtype == "x3"
        self.compute_dtype = "f32" if self.split3 else compute_dtype     # storage / kernel dtype: 'x3' is f32 with split products
        self.num_model, self.num_head, self.num_feedforward = num_model, num_head, num_feedforward
        self.dropout = float(dropout)
        # the reference passes normalize_before in torch's layer_norm_eps slot (models.py:60-61,66-67)
        self.eps_layer = float(normalize_before)
        self.has_enc_norm = bool(normalize_before)
        self.num_encoder_layers, self.num_decoder_layers = num_encoder_layers, num_decoder_layers
        self.max_num_input = math.ceil(max_input_length / num_input_dof)
        self.max_num_output = math.ceil(max_output_length / num_output_dof)
        self.max_input_length, self.max_output_length = max_input_length, max_output_length
        self.num_input_dof, self.num_output_dof = num_input_dof, num_output_dof
        self.vocab_size = vocab_size
        self.token = token

        d, ff = num_model, num_feedforward
        shapes = OrderedDict()
        tab_rows = dict(input_value=vocab_size, input_pos=self.max_num_input, input_coord=num_input_dof,
                        input_view=num_view, input_type=num_type)
        for k in INPUT_KEYS:
            shapes[f"input_embeddings.{k}.weight"] = (tab_rows[k], d)
        shapes["query_coord_embedding.weight"] = (num_output_dof, d)
        shapes["query_pos_embedding.weight"] = (self.max_num_output, d)

        def attn(p):
            shapes[p + "in_proj_weight"] = (3 * d, d); shapes[p + "in_proj_bias"] = (3 * d,)
            shapes[p + "out_proj.weight"] = (d, d); shapes[p + "out_proj.bias"] = (d,)

        def ffn(p):
            shapes[p + "linear1.weight"] = (ff, d); shapes[p + "linear1.bias"] = (ff,)
            shapes[p + "linear2.weight"] = (d, ff); shapes[p + "linear2.bias"] = (d,)

        def norm(p):
            shapes[p + "weight"] = (d,); shapes[p + "bias"] = (d,)

        for i in range(num_encoder_layers):
            p = f"encoder.layers.{i}."
            attn(p + "self_attn."); ffn(p); norm(p + "norm1."); norm(p + "norm2.")
        norm("encoder.norm.")
        for i in range(num_decoder_layers):
            p = f"decoder.layers.{i}."
            attn(p + "self_attn."); attn(p + "multihead_attn."); ffn(p)
            norm(p + "norm1."); norm(p + "norm2."); norm(p + "norm3.")
        norm("decoder.norm.")
        shapes["vocab_head.weight"] = (vocab_size, d); shapes["vocab_head.bias"] = (vocab_size,)
        shapes["pointer_head.weight"] = (d, d); shapes["pointer_head.bias"] = (d,)
        shapes["switch_head.weight"] = (1, d); shapes["switch_head.bias"] = (1,)

        self._order = param_order(num_encoder_layers, num_decoder_layers)
        assert list(shapes) == self._order
        self._shapes = shapes
        self._offsets, off = {}, 0
        for k, s in shapes.items():
            self._offsets[k] = off
            off += (math.prod(s) + 63) // 64 * 64            # 256-byte aligned sub-buffers
        self._numel = off

        # module tree with the reference's names; parameters are created as views of one flat buffer
        flat = torch.zeros(self._numel, dtype=torch.float32)
        self._params = OrderedDict()
        for k, s in shapes.items():
            parts = k.split(".")
            mod = self
            for name in parts[:-1]:
                if not hasattr(mod, name):
                    mod.add_module(name, _Holder())
                mod = getattr(mod, name)
            n = math.prod(s)
            view = flat[self._offsets[k]: self._offsets[k] + n].view(s)
            prm = nn.Parameter(view)
            mod.register_parameter(parts[-1], prm)
            self._params[k] = prm
        if not self.has_enc_norm:
            # the reference builds no encoder.norm in this case (models.py:62); keep the slots out of state_dict
            del self.encoder._modules["norm"]
            for k in ("encoder.norm.weight", "encoder.norm.bias"):
                self._params[k].requires_grad_(False)
        self._flat = flat
        self._gflat = None
        self._gtmp = None
        self._shadow = None
        self._shadowT = self._vocabT = self._kvT = None
        self._tr_descs = None
        self._tr_stream = self._tr_event = None
        self._tr_pending = False
        self._shadow_version = -1
        self._handle = None
        self._ws = None
        self._stats = None
        self._step_seed = 0
        self._grad_hook = None
        self._hook_leaf = None
        self._decoder = None
        self._reset_parameters()

    # ---------------------------------------------------------------------------------- init / layout
    def _reset_parameters(self):
        """reference models.py:78-83 (xavier on dim>1) + torch defaults for 1-D parameters."""
        with torch.no_grad():
            for k, p in self._params.items():
                if p.dim() > 1:
                    _xavier_(p)
                elif "norm" in k and k.endswith("weight"):
                    p.fill_(1.0)
                elif k.endswith("linear1.bias") or k.endswith("linear2.bias") or k in (
                        "vocab_head.bias", "pointer_head.bias", "switch_head.bias"):
                    fan_in = self._shapes[k.replace("bias", "weight")][1]
                    bound = 1.0 / math.sqrt(fan_in)
                    p.uniform_(-bound, bound)
                else:
                    p.zero_()
            # torch's nn.TransformerEncoder / nn.TransformerDecoder deep-copy ONE layer (reference models.py:60-69): the 1-D parameters
            # that _reset_parameters leaves alone - linear1.bias / linear2.bias - start IDENTICAL in every layer of a stack (SURVEY
            # appendix C).  Same here: layer 0's draw is the stack's.
            for stack in ("encoder.layers.", "decoder.layers."):
                for name in ("linear1.bias", "linear2.bias"):
                    first = self._params.get(stack + "0." + name)
                    i = 1
                    while first is not None and (stack + f"{i}." + name) in self._params:
                        self._params[stack + f"{i}." + name].copy_(first)
                        i += 1

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._reflatten()
        return out

    def _reflatten(self):
        """Re-establish the single flat buffer after .to()/.cuda()/load_state_dict re-created tensors."""
        first = next(iter(self._params.values()))
        dev = first.device
        need = any(p.data.untyped_storage().data_ptr() != self._flat.untyped_storage().data_ptr()
                   or p.device != self._flat.device or p.dtype != torch.float32 for p in self._params.values())
        if not need:
            return
        flat = torch.zeros(self._numel, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for k, p in self._params.items():
                n = p.numel()
                view = flat[self._offsets[k]: self._offsets[k] + n].view(self._shapes[k])
                view.copy_(p.data.to(torch.float32))
                p.data = view
                p.grad = None
        self._flat = flat
        self._gflat = self._gtmp = self._shadow = self._shadowT = self._vocabT = self._kvT = None
        self._tr_descs = None
        self._shadow_version = -1
        self._tr_stream = self._tr_event = None
        self._tr_pending = False
        self._drop_handle()

    def _drop_handle(self):
        if self._handle is not None:
            L.lib().pa_model_destroy(self._handle)
            self._handle = None
        self._ws = None
        self._decoder = None
        self._drop_x3_cache()

    def _drop_x3_cache(self):
        cache = getattr(self, "_x3_cache", None)
        if cache is not None:
            L.lib().pa_gemm_split_cache_destroy(cache[0])
        self._x3_cache = None
        ctx = getattr(self, "_x3_ctx", None)
        if ctx is not None:                         # this model's bf16x3 context and scratch (include/plank_hip.h pa_split_ctx_*)
            L.lib().pa_split_ctx_destroy(ctx)
        self._x3_ctx = None
        self._x3_ws = None

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:
            pass

    # state_dict without encoder.norm when the reference would not have it
    def state_dict(self, *args, **kwargs):
        return super().state_dict(*args, **kwargs)

    # ---------------------------------------------------------------------------------- runtime binding
    @property
    def flat_params(self):
        return self._flat

    @property
    def flat_grads(self):
        self._ensure_grads()
        return self._gflat

    @property
    def grad_sync_buffer(self):
        """The flat buffer the running backward writes (what a grad-ready hook must exchange): the accumulation
        scratch while gradients are being accumulated, else ``flat_grads``."""
        t = getattr(self, "_sync_target", None)
        return t if t is not None else self.flat_grads

    def segment_slices(self):
        """[(lo, hi)] element ranges of the flat gradient buffer that become final after each backward
        segment, in execution order (heads, decoder layers last->first, output embedding, encoder.norm,
        encoder layers last->first, input embedding).  The shared value table lives in the last range."""
        o = self._offsets
        ne, nd = self.num_encoder_layers, self.num_decoder_layers
        end = self._numel
        out = [(o["decoder.norm.weight"], end)]
        for i in range(nd - 1, -1, -1):
            lo = o[f"decoder.layers.{i}.self_attn.in_proj_weight"]
            hi = o[f"decoder.layers.{i + 1}.self_attn.in_proj_weight"] if i + 1 < nd else o["decoder.norm.weight"]
            out.append((lo, hi))
        first_dec = o["decoder.layers.0.self_attn.in_proj_weight"] if nd else o["decoder.norm.weight"]
        out.append((o["query_coord_embedding.weight"], o["encoder.layers.0.self_attn.in_proj_weight"] if ne
                    else o["encoder.norm.weight"]))
        out.append((o["encoder.norm.weight"], first_dec))
        for i in range(ne - 1, -1, -1):
            lo = o[f"encoder.layers.{i}.self_attn.in_proj_weight"]
            hi = o[f"encoder.layers.{i + 1}.self_attn.in_proj_weight"] if i + 1 < ne else o["encoder.norm.weight"]
            out.append((lo, hi))
        out.append((0, o["query_coord_embedding.weight"]))
        return out

    def register_grad_ready_hook(self, fn):
        """fn(seg_index, lo, hi) is called right after backward segment ``seg_index`` was enqueued."""
        self._grad_hook = fn

    def _require_gpu(self):
        if self._flat.device.type != "cuda":
            raise L.PlankHipError(
                "plankassembly_amd.PlankModel runs only on a ROCm GPU (module is on "
                f"'{self._flat.device}'): move it with .cuda(); there is no CPU fallback path.")
        L.lib()

    def _pa_dtype(self):
        return L.PA_BF16 if self.compute_dtype == "bf16" else L.PA_F32

    def _split(self, on, retain=False):
        """'x3': this host thread enters / leaves the model's OWN bf16x3 context around its library calls (no-op for the other
        dtypes): mode, scratch, retained images and counters belong to the model, not to the process (pa_split_ctx_*)."""
        if not self.split3:
            return
        attn = os.environ.get("PLANK_X3_ATTN", "1") != "0"         # (0: attention stays on the exact-f32 kernels)
        if not on:
            if getattr(self, "_x3_ctx", None) is not None:
                L.check(L.lib().pa_split_ctx_enter(self._x3_ctx), "pa_split_ctx_enter")
                L.check(L.lib().pa_gemm_split_config(0, None, 0), "pa_gemm_split_config")
                L.check(L.lib().pa_attn_split_config(0), "pa_attn_split_config")
                L.check(L.lib().pa_gemm_split_cache_use(None), "pa_gemm_split_cache_use")
            L.check(L.lib().pa_split_ctx_enter(None), "pa_split_ctx_enter")
            return
        try:
            self._split_on(attn, retain)
        except BaseException:
            #  a failure half way - scratch allocation, cache creation - must not leave the process-global mode on
            # for every other f32 model of the process
            try:
                self._split(False)
            except Exception:
                pass
            raise

    def _split_on(self, attn, retain):
        if getattr(self, "_x3_ctx", None) is None:
            h = C.c_void_p()
            L.check(L.lib().pa_split_ctx_create(C.byref(h)), "pa_split_ctx_create")
            self._x3_ctx = h
        L.check(L.lib().pa_split_ctx_enter(self._x3_ctx), "pa_split_ctx_enter")
        if attn:
            L.check(L.lib().pa_attn_split_config(1), "pa_attn_split_config")
        ws = getattr(self, "_x3_ws", None)
        if ws is None or ws.device != self._flat.device:
            # the cut operands of one GEMM at a time, plus (retain modes) the forward's Linear inputs for the backward's weight
            # gradients: 1.1 GB at the benchmark batch; what does not fit is simply cut again.  Owned by the model: released with it.
            mb = int(os.environ.get("PLANK_X3_SCRATCH_MB", "2048"))
            ws = torch.empty(mb * (1 << 20) + 256, dtype=torch.uint8, device=self._flat.device)
            self._x3_ws = ws
        base = (ws.data_ptr() + 255) // 256 * 256
        # backward segments: mode 2 keeps the cut dY of every dX GEMM for the segment's grouped weight-gradient launch; the
        # forward: mode 3 keeps the cut input of every Linear for the same launch (include/plank_hip.h pa_gemm_split_config)
        keep = os.environ.get("PLANK_X3_RETAIN", "1")
        mode = 1 if (not retain or keep == "0") else 3 if (retain == "fwd" and keep != "bwd") else 1 if retain == "fwd" else 2
        L.check(L.lib().pa_gemm_split_config(mode, C.c_void_p(base), C.c_int64(ws.numel() - (base - ws.data_ptr()))), "pa_gemm_split_config")
        # this model's weight images: learnt by the first step, re-cut by refresh_transposed() whenever the parameters changed
        if retain and getattr(self, "_x3_cache", None) is None:
            # OFF by default (PLANK_X3_WCACHE_MB=0): measured on MI355X the split launches it removes (28.0 -> 21.8 ms over 23 steps)
            # cost less than what the GEMMs lose reading weight images cut a whole step earlier instead of just before the launch
            # (pair GEMM 54.9 -> 62.6 us, gemm3s 18.2 -> 23.3 us): 12.07 -> 12.57 ms per step.  -1: sized from the parameters.
            mb = int(os.environ.get("PLANK_X3_WCACHE_MB", "0"))
            if mb < 0:
                mb = max(64, (self._numel * 12 >> 20) + 64)
            if mb > 0:
                buf = torch.empty(mb * (1 << 20) + 256, dtype=torch.uint8, device=self._flat.device)
                cb = (buf.data_ptr() + 255) // 256 * 256
                h = C.c_void_p()
                L.check(L.lib().pa_gemm_split_cache_create(C.c_void_p(cb), C.c_int64(buf.numel() - (cb - buf.data_ptr())), C.byref(h)),
                        "pa_gemm_split_cache_create")
                self._x3_cache = (h, buf)
        cache = getattr(self, "_x3_cache", None)
        if retain and cache is not None:
            L.check(L.lib().pa_gemm_split_cache_use(cache[0]), "pa_gemm_split_cache_use")

    def _pa_activation(self):
        return {"relu": 1, "gelu": 2}[self.activation]

    def _ensure_handle(self):
        self._require_gpu()
        if self._handle is not None:
            return
        rows = (C.c_int32 * 5)(*[self._shapes[f"input_embeddings.{k}.weight"][0] for k in INPUT_KEYS])
        cfg = _ModelCfg(self.num_model, self.num_head, self.num_feedforward, self.num_encoder_layers,
                        self.num_decoder_layers, self.vocab_size, self.num_output_dof, rows, self.eps_layer, 1e-5,
                        int(self.has_enc_norm), self.dropout, int(self.token.PAD), int(self.token.END),
                        self._pa_dtype(), self._pa_activation())
        h = C.c_void_p()
        L.check(L.lib().pa_model_create(C.byref(cfg), C.byref(h)), "pa_model_create")
        self._handle = h
        self._rebind()

    def new_bound_handle(self):
        """A second runtime handle over the SAME parameter buffers (its own batch / workspace / decode state): what lets
        two halves of a decode batch run as independent kernel streams (decode.GreedyDecoder lanes).  The caller owns it
        (pa_model_destroy)."""
        self._ensure_handle()
        rows = (C.c_int32 * 5)(*[self._shapes[f"input_embeddings.{k}.weight"][0] for k in INPUT_KEYS])
        cfg = _ModelCfg(self.num_model, self.num_head, self.num_feedforward, self.num_encoder_layers,
                        self.num_decoder_layers, self.vocab_size, self.num_output_dof, rows, self.eps_layer, 1e-5,
                        int(self.has_enc_norm), self.dropout, int(self.token.PAD), int(self.token.END), self._pa_dtype(),
                        self._pa_activation())
        h = C.c_void_p()
        L.check(L.lib().pa_model_create(C.byref(cfg), C.byref(h)), "pa_model_create")
        pf = self._ptr_table(self._flat, 4)
        pl = self._ptr_table(self._shadow, 2) if self.compute_dtype == "bf16" else pf
        L.check(L.lib().pa_model_bind(h, pf, pl, None), "pa_model_bind")
        return h

    def _ptr_table(self, flat, esz):
        n = len(self._order)
        arr = (C.c_void_p * n)()
        base = flat.data_ptr()
        for i, k in enumerate(self._order):
            arr[i] = base + self._offsets[k] * esz
        return arr

    def _rebind(self, grads=None):
        pf = self._ptr_table(self._flat, 4)
        if self.compute_dtype == "bf16":
            if self._shadow is None:
                self._shadow = torch.empty(self._numel, dtype=torch.bfloat16, device=self._flat.device)
                self._shadow_version = -1
            pl = self._ptr_table(self._shadow, 2)
        else:
            pl = pf
        g = grads if grads is not None else self._gflat
        gr = self._ptr_table(g, 4) if g is not None else None
        L.check(L.lib().pa_model_bind(self._handle, pf, pl, gr), "pa_model_bind")
        self._bound_grads = g
        self._setup_transposed_shadow()

    def _setup_transposed_shadow(self):
        """Shadow of W^T (compute dtype) for every 2-D Linear weight whose transpose stays 16-byte aligned: the backward
        dX = dY W then runs as a k-contiguous GEMM (csrc/runtime.hip linear_dx).  bf16: derived from the bf16 parameter shadow,
        plus vocab_head^T padded to a whole K tile and the packed cross-attention K|V weights of all decoder layers.  f32
        (round 4): derived from the parameters themselves - the strided-operand form of the exact-f32 GEMM (register-staged
        4 x 4 transposes) ran at 0.4 of the f32 MFMA peak and was a quarter of the f32 step."""
        if self._shadowT is not None:
            return
        dev = self._flat.device
        bf = self.compute_dtype == "bf16"
        esz, vec = (2, 8) if bf else (4, 4)
        src_buf = self._shadow if bf else self._flat
        self._shadowT = torch.zeros(self._numel, dtype=torch.bfloat16 if bf else torch.float32, device=dev)
        n = len(self._order)
        tab = (C.c_void_p * n)()
        descs, tiles = [], 0
        for i, k in enumerate(self._order):
            s = self._shapes[k]
            is_linear_w = len(s) == 2 and ("proj" in k or "linear" in k or k == "pointer_head.weight")
            if bf and k == "vocab_head.weight" and s[1] % 8 == 0:
                # [vocab][d] -> W^T [d][ldv] with the vocabulary padded to a whole K tile (pad columns stay zero): the
                # runtime contracts d(logits) [rows][ldv] with it (csrc/runtime.hip bwd_heads)
                ldv = (s[0] + 63) // 64 * 64
                self._vocabT = torch.zeros(s[1] * ldv, dtype=torch.bfloat16, device=dev)
                tab[i] = self._vocabT.data_ptr()
                d = _TrDesc(self._shadow.data_ptr() + self._offsets[k] * 2, self._vocabT.data_ptr(), s[0], s[1], s[1], ldv, tiles, 0)
                tiles += ((s[0] + 63) // 64) * ((s[1] + 63) // 64)
                descs.append(d)
                continue
            if not is_linear_w or s[0] % vec or s[1] % vec:
                tab[i] = None
                continue
            off = self._offsets[k] * esz
            tab[i] = self._shadowT.data_ptr() + off
            d = _TrDesc(src_buf.data_ptr() + off, self._shadowT.data_ptr() + off, s[0], s[1], s[1], s[0], tiles, 0)
            tiles += ((s[0] + 63) // 64) * ((s[1] + 63) // 64)
            descs.append(d)
        # cross-attention K/V weights of all decoder layers, transposed and packed: kvT[k][l*2d + n] = W_in_l[d + n][k]
        nd, dm = self.num_decoder_layers, self.num_model
        self._kvT = None
        if bf and nd > 0 and dm % 8 == 0 and os.environ.get("PLANK_CROSS_KV", "1") != "0":
            self._kvT = torch.zeros(dm * nd * 2 * dm, dtype=torch.bfloat16, device=dev)
            for li in range(nd):
                k = f"decoder.layers.{li}.multihead_attn.in_proj_weight"
                src = self._shadow.data_ptr() + (self._offsets[k] + dm * dm) * 2
                dst = self._kvT.data_ptr() + li * 2 * dm * 2
                descs.append(_TrDesc(src, dst, 2 * dm, dm, dm, nd * 2 * dm, tiles, 0))
                tiles += ((2 * dm + 63) // 64) * ((dm + 63) // 64)
        arr = (_TrDesc * len(descs))(*descs)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self._tr_descs = (host.to(dev), len(descs), tiles)
        L.check(L.lib().pa_model_bind_transposed(self._handle, tab), "pa_model_bind_transposed")
        L.check(L.lib().pa_model_bind_cross_kv_t(self._handle, L.ptr(self._kvT) if self._kvT is not None else None),
                "pa_model_bind_cross_kv_t")

    def refresh_transposed(self):
        """Re-derive the W^T shadow from the bf16 shadow (one batched transpose launch) on the current stream.  Only the BACKWARD
        reads W^T (dX = dY W as a k-contiguous GEMM, csrc/runtime.hip linear_dx); PLANK_TRANSPOSE_STREAM=1 (opt-in) sends the launch
        to a side stream behind the kernels enqueued so far and lets the backward wait for its event (wait_transposed), so the next
        forward does not queue behind the 40 us (bf16; f32: 70 us).  MEASURED SLOWER on MI355X / ROCm 7.2: 4.72 ms per step against
        4.56 (same session, twice; profiles/r06_transpose_stream.txt) - a second active stream costs the main one more than the
        40 us it hides, as every two-stream arrangement tried since round 4 did."""
        if self._tr_descs is not None:
            d, n, tiles = self._tr_descs
            dt = L.PA_BF16 if self.compute_dtype == "bf16" else L.PA_F32
            if os.environ.get("PLANK_TRANSPOSE_STREAM", "0") == "1":
                if self._tr_stream is None:
                    self._tr_stream, self._tr_event = torch.cuda.Stream(device=self._flat.device), torch.cuda.Event()
                self._tr_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._tr_stream):
                    L.check(L.lib().pa_transpose_many(L.ptr(d), n, tiles, dt, L.stream()), "pa_transpose_many")
                    self._tr_event.record(self._tr_stream)
                self._tr_pending = True
            else:
                L.check(L.lib().pa_transpose_many(L.ptr(d), n, tiles, dt, L.stream()), "pa_transpose_many")
        # 'x3': the cut images of the weights (and of their transposes) follow the parameters (pa_gemm_split_cache_refresh)
        cache = getattr(self, "_x3_cache", None)
        if cache is not None:
            self.wait_transposed()               # (the images of the transposes are cut from W^T)
            L.check(L.lib().pa_gemm_split_cache_refresh(cache[0], L.stream()), "pa_gemm_split_cache_refresh")

    def wait_transposed(self):
        """The current stream waits for the side-stream transposes (refresh_transposed): called before anything that reads W^T (the
        backward) or rewrites their source (the optimizer's update of the bf16 shadow / the parameters, _refresh_shadow's cast)."""
        if self._tr_pending:
            torch.cuda.current_stream().wait_event(self._tr_event)

    def _ensure_grads(self):
        if self._gflat is None:
            self._gflat = torch.zeros(self._numel, dtype=torch.float32, device=self._flat.device)

    def _param_version(self):
        """Staleness stamp of the low-precision shadows.  After .cuda()/.to() every parameter is re-pointed at the flat
        buffer with ``p.data = view`` and owns its version counter, so in-place updates through the parameters
        (torch.optim.Adam, load_state_dict, p.add_()) never touch ``_flat._version``: the stamp is the sum over all of
        them.  Updates through the C ABI (FusedAdam) bump nothing and call mark_shadow_fresh()/invalidate_shadow()."""
        v = self._flat._version
        for p in self._params.values():
            v += p._version
        return v

    def invalidate_shadow(self):
        self._shadow_version = -1

    def _refresh_shadow(self):
        v = self._param_version()
        if v != self._shadow_version:
            self.wait_transposed()
            if self.compute_dtype == "bf16":
                L.check(L.lib().pa_cast(L.ptr(self._shadow), L.PA_BF16, L.ptr(self._flat), L.PA_F32,
                                        C.c_int64(self._numel), L.stream()), "pa_cast")
            self._shadow_version = v
            self.refresh_transposed()

    def mark_shadow_fresh(self):
        """Called by the fused optimizer, which refreshes the bf16 shadow inside its own kernel."""
        self._shadow_version = self._param_version()
        self.refresh_transposed()

    # ---------------------------------------------------------------------------------- batches
    def _make_batch(self, batch, with_output=True):
        iv = batch["input_value"]
        B, S = iv.shape
        keep = []
        b = _Batch()
        for i, k in enumerate(INPUT_KEYS):
            t = batch.get(k)
            if t is None:
                b.input_idx[i] = None
                continue
            t = t.to(device=self._flat.device, dtype=torch.int64).contiguous()
            keep.append(t)
            b.input_idx[i] = t.data_ptr()
        m = batch["input_mask"].to(self._flat.device).contiguous()
        m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
        keep.append(m)
        b.input_mask = m.data_ptr()
        if self.unpad:
            # pack the encoder to its valid rows (padded positions never reach the loss).  A batch that went through
            # prepare_batch() carries the packing already; otherwise it is computed here, which costs the one
            # device->host read (n_valid) of the step.
            pack = batch.get("_pack")
            if pack is None or pack[0].device != m.device:
                pack = self._pack(m)
            cu, rowmap, n_valid = pack
            keep += [cu, rowmap]
            b.cu_in, b.rowmap, b.n_valid = cu.data_ptr(), rowmap.data_ptr(), n_valid
            self._last_pack = (cu, rowmap, n_valid, B, S)
            groups = batch.get("_groups")
            if groups is not None and with_output:
                for j, grp in enumerate(groups["in"]):      # encoder rows grouped by table row, per input table
                    if grp is not None and grp[0].device == m.device:
                        keep += [grp[0], grp[1]]
                        b.in_order[j], b.in_seg[j] = grp[0].data_ptr(), grp[1].data_ptr()
                og = groups.get("out")
                if og is not None and og[0][0].device == m.device and groups.get("out_T") == batch["output_value"].shape[1]:
                    for j, grp in enumerate(og):            # decoder rows grouped by value / coord / pos table row
                        keep += [grp[0], grp[1]]
                        b.out_order[j], b.out_seg[j] = grp[0].data_ptr(), grp[1].data_ptr()
        else:
            self._last_pack = None
        T = self.max_output_length
        if with_output:
            ov = batch["output_value"].to(device=self._flat.device, dtype=torch.int64).contiguous()
            ol = batch["output_label"].to(device=self._flat.device, dtype=torch.int64).contiguous()
            om = batch["output_mask"].to(self._flat.device).contiguous()
            om = om.view(torch.uint8) if om.dtype == torch.bool else om.to(torch.uint8)
            keep += [ov, ol, om]
            T = ov.shape[1]
            b.output_value, b.output_label, b.output_mask = ov.data_ptr(), ol.data_ptr(), om.data_ptr()
        b.B, b.S, b.T = B, S, T
        return b, keep

    def _pack(self, mask_u8, n_valid=None):
        """``n_valid``: the number of unmasked encoder rows when the caller already knows it on the HOST (a dataloader that
        padded the batch does; ``batch["_n_valid"]``) - the one device -> host read of a step is then skipped.  Contract: it
        must equal ``(~input_mask).sum()`` of THIS batch."""
        B, S = mask_u8.shape
        cu = torch.empty(2 * B + 1, dtype=torch.int32, device=mask_u8.device)
        rowmap = torch.empty(B * S, dtype=torch.int32, device=mask_u8.device)
        L.check(L.lib().pa_pack_rows(L.ptr(mask_u8), B, S, L.ptr(cu), L.ptr(rowmap), L.stream()), "pa_pack_rows")
        if n_valid is None:
            return cu, rowmap, int(cu[B])
        # A host-supplied count sizes every packed encoder launch: a stale one (batch edited after collation) reads / writes
        # past the packed rows.  Checked against the device's own count on a model's first uses (a device -> host read each;
        # PLANK_CHECK_NVALID=1: always, 0: never) - ADVICE r5.
        chk = os.environ.get("PLANK_CHECK_NVALID")
        self._nvalid_checks = getattr(self, "_nvalid_checks", 0) + 1
        if chk != "0" and (chk == "1" or self._nvalid_checks <= 4):
            got = int(cu[B])
            if got != int(n_valid):
                raise ValueError(f"batch['_n_valid'] = {int(n_valid)} but input_mask has {got} unmasked rows")
        return cu, rowmap, int(n_valid)

    def prepare_batch(self, batch, groups=True):
        """Move a collated batch to the model's device and attach the encoder row packing (``_pack``: valid-row
        offsets per sample, packed-row -> position map, number of valid rows) and - for training (``groups``) - the
        token rows grouped by embedding-table row (``_groups``).  Doing this when the batch is built (dataloader /
        before the timed region) keeps the training loop free of device->host reads; forward() accepts unprepared
        batches too."""
        self._require_gpu()
        dev = self._flat.device
        out = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
        if self.unpad:
            msk = out["input_mask"].contiguous()
            msk = msk.view(torch.uint8) if msk.dtype == torch.bool else msk.to(torch.uint8)
            out["_pack"] = self._pack(msk, batch.get("_n_valid"))
            # group the token rows by embedding-table row: the table gradients then are segment sums instead of millions
            # of atomics (pa_embed_segment_bwd).  Batch-only information, like the packing: ONE launch for all eight
            # tables (pa_group_rows, a stable counting sort per table), no torch op.
            if groups:
                cu, rowmap, n_valid = out["_pack"]
                out["_groups"] = self._group_rows(out, rowmap, n_valid)
        return out

    def _group_rows(self, batch, rowmap, n_valid):
        """{'in': [(order, seg) | None] * 5, 'out': [(order, seg)] * 3 | None, 'out_T': T} on the model's device."""
        dev = self._flat.device
        descs, slots = [], []

        def add(kind, idx, n, rows, T=0, dof=0, tok_ld=0, rmap=None):
            order = torch.empty(n, dtype=torch.int32, device=dev)
            seg = torch.empty(rows + 1, dtype=torch.int32, device=dev)
            descs.append(L.GroupDesc(idx.data_ptr() if idx is not None else None, rmap.data_ptr() if rmap is not None else None,
                                     n, rows, kind, T, dof, tok_ld, order.data_ptr(), seg.data_ptr()))
            slots.append((order, seg))
            return order, seg

        keep = []
        gin = []
        for key in INPUT_KEYS:
            t = batch.get(key)
            if t is None or n_valid <= 0:
                gin.append(None)
                continue
            t = t.to(device=dev, dtype=torch.int64).contiguous()
            keep.append(t)
            gin.append(add(0, t, n_valid, self._shapes[f"input_embeddings.{key}.weight"][0], rmap=rowmap))
        groups = {"in": gin, "out": None}
        ov = batch.get("output_value")
        if ov is not None and ov.shape[1] >= 2:
            ov = ov.to(device=dev, dtype=torch.int64).contiguous()
            keep.append(ov)
            Bq, Tq = ov.shape
            dof, n = self.num_output_dof, Bq * (Tq - 1)
            groups["out"] = [add(1, ov, n, self._shapes["input_embeddings.input_value.weight"][0], Tq, dof, Tq),
                             add(2, None, n, dof, Tq, dof, Tq),
                             add(3, None, n, (Tq + dof - 1) // dof, Tq, dof, Tq)]
            groups["out_T"] = Tq
        if descs:
            arr = (L.GroupDesc * len(descs))(*descs)
            L.check(L.lib().pa_group_rows(arr, len(descs), L.stream()), "pa_group_rows")
        groups["_keep"] = keep
        return groups

    def _workspace(self, B, S, T):
        need = int(L.lib().pa_model_train_ws_bytes(self._handle, B, S, T))
        if need < 0:
            L.check(need, "pa_model_train_ws_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self._flat.device)
        return self._ws

    # ---------------------------------------------------------------------------------- train step
    def train_step(self, batch):
        """reference models.py:190-233."""
        self._ensure_handle()
        self._refresh_shadow()
        b, keep = self._make_batch(batch, True)
        ws = self._workspace(b.B, b.S, b.T)
        base = (ws.data_ptr() + 255) // 256 * 256
        stats = torch.empty(L.lib().pa_model_stats_floats(), dtype=torch.float32, device=self._flat.device)   # include/plank_hip.h: f32[8]
        self._step_seed = (self._step_seed * 1664525 + 1013904223 + int(torch.initial_seed())) & 0xFFFFFFFF
        self._split(True, retain="fwd")
        try:
            L.check(L.lib().pa_model_train_fwd(self._handle, C.byref(b), C.c_void_p(base),
                                               C.c_int64(ws.numel() - (base - ws.data_ptr())),
                                               C.c_uint32(self._step_seed), 1 if self.training else 0,
                                               L.ptr(stats), L.stream()), "pa_model_train_fwd")
        finally:
            self._split(False)
        self._live = (b, keep, stats)
        if self._hook_leaf is None or self._hook_leaf.device != stats.device:
            self._hook_leaf = torch.zeros((), device=stats.device, requires_grad=True)
        if torch.is_grad_enabled():
            loss = _TrainStepFn.apply(self._hook_leaf, self, stats)
        else:
            loss = stats[4]
        return {"loss": loss, "accuracy": stats[5]}

    def _run_backward(self, gloss):
        self._ensure_grads()
        self.wait_transposed()
        accumulate = any(p.grad is not None for p in self._params.values())
        target = self._gflat
        if accumulate:
            if self._gtmp is None:
                self._gtmp = torch.empty_like(self._gflat)
            target = self._gtmp
        target.zero_()
        if self._bound_grads is not target:
            self._rebind(target)
        nseg = int(L.lib().pa_model_train_num_segments(self._handle))
        # upstream d(loss): read in place by the first backward kernel (no copy launch, no host sync)
        up = gloss if (gloss.dtype == torch.float32 and gloss.is_cuda and gloss.numel() == 1) else \
            gloss.reshape(1).to(device=self._flat.device, dtype=torch.float32)
        if not up.is_contiguous():
            up = up.contiguous()
        self._upstream_keep = up                     # alive until the kernels that read it have been enqueued behind it
        L.check(L.lib().pa_model_set_upstream(self._handle, L.ptr(up)), "pa_model_set_upstream")
        slices = self.segment_slices()
        # gradient accumulation: the fresh micro-batch contribution (target = _gtmp) is what gets exchanged - the
        # buffer accumulated so far was already summed over the ranks - and is added to _gflat after the last slice
        hook = self._grad_hook
        self._sync_target = target
        # with the library's side stream the gradients of segment s are final once segment s + lag is enqueued
        lag = int(L.lib().pa_model_grad_lag(self._handle))
        for s in range(nseg):
            self._split(True, retain=True)
            try:
                L.check(L.lib().pa_model_train_bwd(self._handle, s, s + 1, C.c_float(1.0), L.stream()),
                        "pa_model_train_bwd")
            finally:
                self._split(False)
            if hook is not None and s - lag >= 0:
                hook(s - lag, *slices[s - lag])
        if hook is not None:
            for s in range(max(0, nseg - lag), nseg):       # the last segment joined everything
                hook(s, *slices[s])
        if accumulate:
            self._gflat.add_(self._gtmp)        # (the hook of the last segment waited for every collective)
        self._sync_target = None
        for k, p in self._params.items():
            if p.grad is None and p.requires_grad:
                n = p.numel()
                p.grad = self._gflat[self._offsets[k]: self._offsets[k] + n].view(self._shapes[k])

    # ---------------------------------------------------------------------------------- eval (greedy decode)
    def parse_sequence(self, sequence):
        """reference models.py:258-265."""
        valid_mask = torch.cumsum(sequence == self.token.END, 0) == 0
        valid_seq = sequence[valid_mask]
        num_plank = len(valid_seq) // self.num_output_dof
        return valid_seq[:num_plank * self.num_output_dof].reshape(-1, self.num_output_dof)

    def eval_step(self, batch):
        """reference models.py:267-323: greedy autoregressive sampling (KV-cached HIP decode)."""
        from .decode import GreedyDecoder
        self._ensure_handle()
        self._refresh_shadow()
        if self._decoder is None:
            self._decoder = GreedyDecoder(self)
        output, attach = self._decoder.run(batch)
        predicts, groundtruths = [], []
        for i in range(output.shape[0]):
            predicts.append(self.parse_sequence(output[i]))
            groundtruths.append(self.parse_sequence(batch["output_value"][i].to(output.device)))
        return {"samples": output, "attach": attach, "predicts": predicts, "groundtruths": groundtruths}

    def forward(self, batch):
        """reference models.py:325-330."""
        return self.train_step(batch) if self.training else self.eval_step(batch)

    # ---------------------------------------------------------------------------------- introspection
    def debug_tensor(self, which):
        """Activations of the last training forward, for the parity tests (pa_model_tensor): 'memory', 'hiddens',
        'vocab_logits', 'ptr_logits', 'enc_ffn<l>' / 'dec_ffn<l>' (FFN hidden activations [rows, d_ff]; encoder rows are
        scattered back to [B*S, d_ff] with zeros at PAD when the step ran packed)."""
        names = {"memory": 0, "hiddens": 1, "vocab_logits": 2, "ptr_logits": 3}
        for l in range(self.num_encoder_layers):
            names[f"enc_ffn{l}"] = 16 + l
        for l in range(self.num_decoder_layers):
            names[f"dec_ffn{l}"] = 80 + l
        p, n = C.c_void_p(), C.c_int64()
        L.check(L.lib().pa_model_tensor(self._handle, names[which], C.byref(p), C.byref(n)), "pa_model_tensor")
        dt = torch.float32 if (which in ("vocab_logits", "ptr_logits") or self.compute_dtype == "f32") else torch.bfloat16
        esz = 4 if dt == torch.float32 else 2
        off = p.value - self._ws.data_ptr()
        t = self._ws[off: off + n.value * esz].view(dt).clone()
        if (which == "memory" or which.startswith("enc_ffn")) and getattr(self, "_last_pack", None) is not None:
            cu, rowmap, nv, B, S = self._last_pack                    # scatter the packed rows back to [B*S, width] (zeros at PAD)
            width = self.num_model if which == "memory" else self.num_feedforward
            full = torch.zeros(B * S, width, dtype=dt, device=t.device)
            full[rowmap[:nv].long()] = t.view(nv, width)
            return full.view(-1)
        return t


def build_model(cfg):
    """reference models.py:333-343.  Optional ``cfg.MODEL.COMPUTE_DTYPE`` ('f32' | 'bf16' | 'x3')."""
    model_cfg = cfg.MODEL
    dtype = getattr(model_cfg, "COMPUTE_DTYPE", None) if not isinstance(model_cfg, dict) else model_cfg.get("COMPUTE_DTYPE")
    return PlankModel(
        cfg.MODEL.NUM_MODEL, cfg.MODEL.NUM_HEAD, cfg.MODEL.NUM_FEEDFORWARD, cfg.MODEL.DROPOUT,
        cfg.MODEL.ACTIVATION, cfg.MODEL.NORMALIZE_BEFORE, cfg.MODEL.NUM_ENCODER_LAYERS,
        cfg.MODEL.NUM_DECODER_LAYERS, cfg.DATA.NUM_VIEW, cfg.DATA.NUM_TYPE, cfg.DATA.NUM_INPUT_DOF,
        cfg.DATA.NUM_OUTPUT_DOF, cfg.DATA.MAX_INPUT_LENGTH, cfg.DATA.MAX_OUTPUT_LENGTH, cfg.DATA.VOCAB_SIZE,
        cfg.TOKEN, compute_dtype=dtype)
