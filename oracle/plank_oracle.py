"""CPU ORACLE for the PlankAssembly encoder-decoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional, state-dict driven restatement of the algorithm
in the reference's ``plankassembly/models.py`` (and of the torch ``nn.Transformer*`` /
``F.multi_head_attention_forward`` semantics that file delegates to; pinned
pytorch=1.10.0 in the reference's environment.yml:95).  It is written with explicit
tensor arithmetic on the CPU (matmul / exp / sum) -- it does NOT assemble
``nn.Transformer*`` modules -- and every function cites the reference lines it follows.

Who may import this: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- as the checker / reported CPU baseline, never as the product
path.  ``plankassembly_amd`` never imports it; the product fails loudly without its HIP
extension.

Parity pinning: the reference has no tests, so this oracle is
pinned against golden vectors produced by importing the real reference model in the
build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.

Quirks that are reproduced on purpose:
* per-layer LayerNorm eps is ``float(NORMALIZE_BEFORE)`` (=1.0 for the shipped configs)
  and layers are POST-norm (reference models.py:60-61,66-67 pass ``normalize_before``
  in torch's ``layer_norm_eps`` positional slot); final norms use 1e-5, and the final
  encoder norm exists only if NORMALIZE_BEFORE is truthy (models.py:62).
* pointer logits are scaled by 1/d_model (models.py:150).
* training fills pointer logits j >= i with the VALUE 1e-6 (models.py:160-161).
* eval returns the un-gated vocab softmax while the prefix is shorter than 6
  (models.py:172-173) and applies the 1e-6 pointer-mask fill after gating (183-184).
* the decoder key-padding mask in training is the UNSHIFTED output_mask (198, 213).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

NEG_INF = float("-inf")


@dataclass
class OracleCfg:
    d_model: int = 512
    n_head: int = 8
    d_ff: int = 1024
    n_enc: int = 6
    n_dec: int = 6
    vocab: int = 514
    in_dof: int = 4
    out_dof: int = 6
    max_input_length: int = 1200
    max_output_length: int = 128
    normalize_before: bool = True
    pad: int = 513
    end: int = 512
    activation: str = "relu"                 # cfg.MODEL.ACTIVATION, handed to torch's Transformer layers (models.py:60-61,66-67)

    @property
    def eps_layer(self) -> float:           # the positional-argument slip
        return float(self.normalize_before)

    @property
    def has_enc_norm(self) -> bool:          # models.py:62
        return bool(self.normalize_before)

    @property
    def max_num_output(self) -> int:         # models.py:33
        return math.ceil(self.max_output_length / self.out_dof)


# ----------------------------------------------------------------------------- primitives
def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def layer_norm(x, w, b, eps):
    """Biased-variance LayerNorm over the last dim (torch nn.LayerNorm semantics)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def softmax_lastdim(x):
    m = x.max(dim=-1, keepdim=True).values
    e = torch.exp(x - m)
    return e / e.sum(dim=-1, keepdim=True)


def log_softmax_lastdim(x):
    m = x.max(dim=-1, keepdim=True).values
    z = x - m
    return z - torch.log(torch.exp(z).sum(dim=-1, keepdim=True))


def _drop(drop, site, x):
    """Dropout with GIVEN decisions: ``drop`` is None (eval / p = 0) or a callable (site, tensor) -> tensor that zeroes the
    dropped entries and scales the survivors by 1 / (1 - p), i.e. what ``F.dropout(x, p, training=True)`` does for one draw
    of its mask.  torch draws the masks from its Philox stream, which no other implementation can reproduce; the tests feed
    the decisions of the implementation under test instead (tests/dropout_masks.py), so that a training step UNDER dropout
    can be compared tensor by tensor.  Sites (torch nn/modules/transformer.py, TransformerEncoderLayer / DecoderLayer,
    norm_first=False; nn/functional.py multi_head_attention_forward "attn = dropout(attn, p=dropout_p)"):
    ``<layer>self_attn`` / ``<layer>multihead_attn`` - attention probabilities [B, H, Lq, Lk];
    ``<layer>dropout1`` / ``dropout2`` / ``dropout3`` - sublayer outputs before the residual add;
    ``<layer>dropout`` - the feed-forward hidden activation after the ReLU."""
    return x if drop is None else drop(site, x)


def mha(q_in, kv_in, p, prefix, n_head, add_mask, drop=None):
    """torch F.multi_head_attention_forward, batch_first; dropout-free unless ``drop`` hands in the decisions (_drop).

    ``add_mask`` is an additive float mask broadcastable to [B, H, Lq, Lk]
    (0 = keep, -inf = drop): the merge of attn_mask and key_padding_mask that torch
    performs (installed torch nn/functional.py "merge key padding and attention masks").
    Heads are contiguous d_head slices; scores are scaled by 1/sqrt(d_head).
    """
    d = q_in.shape[-1]
    dh = d // n_head
    w, b = p[prefix + "in_proj_weight"], p[prefix + "in_proj_bias"]
    q = linear(q_in, w[:d], b[:d])
    k = linear(kv_in, w[d:2 * d], b[d:2 * d])
    v = linear(kv_in, w[2 * d:], b[2 * d:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    q = q.view(B, Lq, n_head, dh).transpose(1, 2)
    k = k.view(B, Lk, n_head, dh).transpose(1, 2)
    v = v.view(B, Lk, n_head, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    if add_mask is not None:
        s = s + add_mask
    a = _drop(drop, prefix[:-1], softmax_lastdim(s))
    o = (a @ v).transpose(1, 2).reshape(B, Lq, d)
    return linear(o, p[prefix + "out_proj.weight"], p[prefix + "out_proj.bias"])


def key_padding_additive(mask_bool):
    """bool [B, L] (True = PAD) -> additive float [B, 1, 1, L]."""
    m = torch.zeros(mask_bool.shape, dtype=torch.float32)
    m = m.masked_fill(mask_bool, NEG_INF)
    return m[:, None, None, :]


def causal_additive(sz):
    """reference models.py:85-89."""
    return torch.triu(torch.full((sz, sz), NEG_INF), diagonal=1)


# ----------------------------------------------------------------------------- embeddings
def embed_input(p, batch):
    """reference models.py:103-112 -- sum over the ``input_*`` id tensors, batch-dict order."""
    out = 0
    for key, value in batch.items():
        if key[:5] != "input" or "mask" in key:
            continue
        out = out + p[f"input_embeddings.{key}.weight"][value]
    return out


def embed_output(p, cfg: OracleCfg, output):
    """reference models.py:114-138 -- shared value table + coord + pos, zero row prepended."""
    B, L = output.shape
    d = p["input_embeddings.input_value.weight"].shape[1]
    val = p["input_embeddings.input_value.weight"][output]
    t = torch.arange(L)
    coord = p["query_coord_embedding.weight"][t % cfg.out_dof]
    pos = p["query_pos_embedding.weight"][t // cfg.out_dof]
    emb = val + coord[None] + pos[None]
    return torch.cat((torch.zeros(B, 1, d, dtype=emb.dtype), emb), dim=1)


# ----------------------------------------------------------------------------- layers
def _act(cfg, relu, key, x):
    """The `activation` of torch's _ff_block (nn/modules/transformer.py: relu, or gelu = F.gelu in its exact erf form).  ``relu``
    (optional, tests only, ReLU models): callable (site key, pre-activation) -> activation, so that a test can evaluate the
    network on GIVEN ReLU branches (tests/test_headline_gpu.py ForcedBranches: the float64 oracle on the branches the f32 device
    run took - a pre-activation within f32 rounding of zero is a coin toss between precisions, and one flipped unit moves every
    upstream gradient)."""
    if cfg.activation == "gelu":
        return torch.nn.functional.gelu(x)
    return torch.relu(x) if relu is None else relu(key, x)


def encoder_layer(x, p, pre, cfg, add_mask, drop=None, relu=None):
    """torch TransformerEncoderLayer.forward, norm_first=False branch (_sa_block: dropout1(self_attn(x));
    _ff_block: dropout2(linear2(dropout(activation(linear1(x))))))."""
    x = layer_norm(x + _drop(drop, pre + "dropout1", mha(x, x, p, pre + "self_attn.", cfg.n_head, add_mask, drop)),
                   p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.eps_layer)
    h = _drop(drop, pre + "dropout", _act(cfg, relu, pre + "linear1", linear(x, p[pre + "linear1.weight"], p[pre + "linear1.bias"])))
    f = _drop(drop, pre + "dropout2", linear(h, p[pre + "linear2.weight"], p[pre + "linear2.bias"]))
    return layer_norm(x + f, p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.eps_layer)


def encode(p, cfg: OracleCfg, batch, drop=None, relu=None):
    """reference models.py:206 / 279."""
    x = embed_input(p, batch)
    add_mask = key_padding_additive(batch["input_mask"])
    for i in range(cfg.n_enc):
        x = encoder_layer(x, p, f"encoder.layers.{i}.", cfg, add_mask, drop, relu)
    if cfg.has_enc_norm:
        x = layer_norm(x, p["encoder.norm.weight"], p["encoder.norm.bias"], 1e-5)
    return x


def decoder_layer(x, memory, p, pre, cfg, self_mask, mem_mask, drop=None, relu=None):
    """torch TransformerDecoderLayer.forward, norm_first=False branch (dropout1 / dropout2 / dropout3 on the three sublayer
    outputs, dropout on the feed-forward hidden activation)."""
    x = layer_norm(x + _drop(drop, pre + "dropout1", mha(x, x, p, pre + "self_attn.", cfg.n_head, self_mask, drop)),
                   p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.eps_layer)
    x = layer_norm(x + _drop(drop, pre + "dropout2", mha(x, memory, p, pre + "multihead_attn.", cfg.n_head, mem_mask, drop)),
                   p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.eps_layer)
    h = _drop(drop, pre + "dropout", _act(cfg, relu, pre + "linear1", linear(x, p[pre + "linear1.weight"], p[pre + "linear1.bias"])))
    f = _drop(drop, pre + "dropout3", linear(h, p[pre + "linear2.weight"], p[pre + "linear2.bias"]))
    return layer_norm(x + f, p[pre + "norm3.weight"], p[pre + "norm3.bias"], cfg.eps_layer)


def decode(p, cfg: OracleCfg, tgt, memory, input_mask, tgt_pad_mask=None, drop=None, relu=None):
    """reference models.py:212-214 (train) / 293-294 (eval: tgt_pad_mask None)."""
    sz = tgt.shape[1]
    self_mask = causal_additive(sz)[None, None]
    if tgt_pad_mask is not None:
        self_mask = self_mask + key_padding_additive(tgt_pad_mask)
    mem_mask = key_padding_additive(input_mask)
    x = tgt
    for i in range(cfg.n_dec):
        x = decoder_layer(x, memory, p, f"decoder.layers.{i}.", cfg, self_mask, mem_mask, drop, relu)
    return layer_norm(x, p["decoder.norm.weight"], p["decoder.norm.bias"], 1e-5)


# ----------------------------------------------------------------------------- heads
def pointer_mask(cfg: OracleCfg, sz):
    """Closed form of reference models.py:91-101 (1 = allowed)."""
    i = torch.arange(sz)[:, None]
    j = torch.arange(sz)[None, :]
    bbox = (j < 6) & (j == i % 6)
    plank = (j >= 6) & ((j % 6) == ((i % 6) + 3) % 6)
    return ((bbox | plank) & (i >= 6)).to(torch.float32)


def head_logits(p, cfg: OracleCfg, h):
    """reference models.py:145-154."""
    vocab = linear(h, p["vocab_head.weight"], p["vocab_head.bias"])
    feat = linear(h, p["pointer_head.weight"], p["pointer_head.bias"])
    ptr = (feat @ h.transpose(1, 2)) / cfg.d_model
    prob = torch.sigmoid(linear(h, p["switch_head.weight"], p["switch_head.bias"]))
    return vocab, ptr, prob


def create_dist_train(p, cfg: OracleCfg, h, eps=1e-6):
    """reference models.py:156-166,186 -- log-probabilities [B, T, V+T]."""
    sz = h.shape[1]
    vocab, ptr, prob = head_logits(p, cfg, h)
    tri = torch.triu(torch.ones(sz, sz)) == 1
    ptr = ptr.masked_fill(tri[None], eps)
    vd = log_softmax_lastdim(vocab) + torch.log(torch.clamp(1 - prob, min=eps))
    pd = log_softmax_lastdim(ptr) + torch.log(torch.clamp(prob, min=eps))
    return torch.cat((vd, pd), dim=-1)


def create_dist_eval(p, cfg: OracleCfg, h, eps=1e-6):
    """reference models.py:168-186 -- probabilities [B, sz, V] (sz<6) or [B, sz, V+sz]."""
    sz = h.shape[1]
    vocab, ptr, prob = head_logits(p, cfg, h)
    vd = softmax_lastdim(vocab)
    if sz < 6:
        return vd
    tri = torch.triu(torch.ones(sz, sz)) == 1
    ptr = ptr.masked_fill(tri[None], NEG_INF)
    # row 0 is all -inf -> NaN in the reference too (never consumed for sz >= 6, row sz-1 is)
    m = ptr.max(dim=-1, keepdim=True).values
    e = torch.exp(ptr - m)
    pd = e / e.sum(dim=-1, keepdim=True)
    vd = vd * (1 - prob)
    pd = pd * prob
    pd = pd.masked_fill((pointer_mask(cfg, sz) == 0)[None], eps)
    return torch.cat((vd, pd), dim=-1)


# ----------------------------------------------------------------------------- train step
def train_forward(p, cfg: OracleCfg, batch, return_all=False, drop=None, relu=None):
    """reference models.py:190-233; dropout-free unless ``drop`` hands in the decisions of every site (_drop); ``relu``: _act."""
    memory = encode(p, cfg, batch, drop, relu)
    tgt = embed_output(p, cfg, batch["output_value"][:, :-1])
    hiddens = decode(p, cfg, tgt, memory, batch["input_mask"], batch["output_mask"], drop, relu)
    dists = create_dist_train(p, cfg, hiddens)
    label = batch["output_label"]
    valid = label != cfg.pad
    picked = dists.gather(-1, label.clamp(max=dists.shape[-1] - 1)[..., None])[..., 0]
    loss = -(picked * valid).sum() / valid.sum()
    predict = dists.argmax(dim=-1)
    correct = (valid & (predict == label)).sum()
    accuracy = float(correct) / (float(valid.sum()) + 1e-10)
    if return_all:
        return dict(loss=loss, accuracy=accuracy, memory=memory, hiddens=hiddens, dists=dists)
    return dict(loss=loss, accuracy=accuracy)


def adam_step(params, grads, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (reference trainer_complete.py:127-129); in place."""
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k in params:
        g = grads[k]
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-lr / bc1)


# ----------------------------------------------------------------------------- greedy decode
def sample(cfg: OracleCfg, last_dist, samples):
    """reference models.py:235-256 -- first-max argmax, pointer copy."""
    tok = last_dist.argmax(dim=-1)
    ptr = torch.full_like(tok, -1)
    is_ptr = tok >= cfg.vocab
    if bool(is_ptr.any()):
        idx = (tok - cfg.vocab).clamp(min=0)
        ptr = torch.where(is_ptr, idx, ptr)
        copied = samples.gather(1, idx[:, None].clamp(max=max(samples.shape[1] - 1, 0)))[:, 0] \
            if samples.shape[1] else tok
        tok = torch.where(is_ptr, copied, tok)
    return tok, ptr


def greedy_decode_recompute(p, cfg: OracleCfg, batch, max_steps=None, early_stop=True):
    """reference models.py:267-307 -- the O(T^2) loop exactly as the reference runs it."""
    memory = encode(p, cfg, batch)
    B = memory.shape[0]
    out = torch.empty(B, 0, dtype=torch.long)
    att = torch.empty(B, 0, dtype=torch.long)
    for _ in range(max_steps or cfg.max_output_length):
        tgt = embed_output(p, cfg, out)
        h = decode(p, cfg, tgt, memory, batch["input_mask"], None)
        dist = create_dist_eval(p, cfg, h)
        tok, ptr = sample(cfg, dist[:, -1], out)
        out = torch.cat((out, tok[:, None]), dim=1)
        att = torch.cat((att, ptr[:, None]), dim=1)
        if early_stop and bool(torch.all(torch.any(out == cfg.end, dim=1))):
            break
    return out, att


def greedy_decode_cached(p, cfg: OracleCfg, batch, max_steps=None, early_stop=True, return_margins=False):
    """Same result as :func:`greedy_decode_recompute` with a K/V cache (causal decoder in
    eval mode => row t only depends on rows <= t).  This is the form the HIP decode path
    implements and the form timed as the ``cached`` CPU baseline.  ``return_margins`` also
    returns [B, steps] = (largest - second largest) / largest probability of every sampled row: how
    far each argmax is from flipping (diagnostics for reduced-precision comparisons)."""
    d, H = cfg.d_model, cfg.n_head
    dh = d // H
    memory = encode(p, cfg, batch)
    B, S, _ = memory.shape
    mem_mask = key_padding_additive(batch["input_mask"])            # [B,1,1,S]
    cross = []
    for i in range(cfg.n_dec):
        pre = f"decoder.layers.{i}.multihead_attn."
        w, b = p[pre + "in_proj_weight"], p[pre + "in_proj_bias"]
        k = linear(memory, w[d:2 * d], b[d:2 * d]).view(B, S, H, dh).transpose(1, 2)
        v = linear(memory, w[2 * d:], b[2 * d:]).view(B, S, H, dh).transpose(1, 2)
        cross.append((k, v))
    steps = max_steps or cfg.max_output_length
    self_k = [torch.zeros(B, H, steps, dh) for _ in range(cfg.n_dec)]
    self_v = [torch.zeros(B, H, steps, dh) for _ in range(cfg.n_dec)]
    hid = torch.zeros(B, steps, d)
    out = torch.empty(B, 0, dtype=torch.long)
    att = torch.empty(B, 0, dtype=torch.long)
    scale = 1.0 / math.sqrt(dh)
    margins = []
    x_in = torch.zeros(B, d)                                         # BOS zero row
    for t in range(steps):
        x = x_in
        for i in range(cfg.n_dec):
            pre = f"decoder.layers.{i}."
            w, b = p[pre + "self_attn.in_proj_weight"], p[pre + "self_attn.in_proj_bias"]
            qkv = linear(x, w, b)
            q = qkv[:, :d].view(B, H, 1, dh)
            self_k[i][:, :, t] = qkv[:, d:2 * d].view(B, H, dh)
            self_v[i][:, :, t] = qkv[:, 2 * d:].view(B, H, dh)
            s = (q @ self_k[i][:, :, :t + 1].transpose(-1, -2)) * scale
            o = (softmax_lastdim(s) @ self_v[i][:, :, :t + 1]).reshape(B, d)
            o = linear(o, p[pre + "self_attn.out_proj.weight"], p[pre + "self_attn.out_proj.bias"])
            x = layer_norm(x + o, p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.eps_layer)
            w, b = p[pre + "multihead_attn.in_proj_weight"], p[pre + "multihead_attn.in_proj_bias"]
            q = linear(x, w[:d], b[:d]).view(B, H, 1, dh)
            ck, cv = cross[i]
            s = (q @ ck.transpose(-1, -2)) * scale + mem_mask
            o = (softmax_lastdim(s) @ cv).reshape(B, d)
            o = linear(o, p[pre + "multihead_attn.out_proj.weight"], p[pre + "multihead_attn.out_proj.bias"])
            x = layer_norm(x + o, p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.eps_layer)
            h = _act(cfg, None, pre + "linear1", linear(x, p[pre + "linear1.weight"], p[pre + "linear1.bias"]))
            f = linear(h, p[pre + "linear2.weight"], p[pre + "linear2.bias"])
            x = layer_norm(x + f, p[pre + "norm3.weight"], p[pre + "norm3.bias"], cfg.eps_layer)
        x = layer_norm(x, p["decoder.norm.weight"], p["decoder.norm.bias"], 1e-5)
        hid[:, t] = x
        dist = last_row_dist(p, cfg, hid[:, :t + 1])
        if return_margins:
            top2 = dist.topk(2, dim=-1).values
            margins.append((top2[:, 0] - top2[:, 1]) / top2[:, 0])
        tok, ptr = sample(cfg, dist, out)
        out = torch.cat((out, tok[:, None]), dim=1)
        att = torch.cat((att, ptr[:, None]), dim=1)
        if early_stop and bool(torch.all(torch.any(out == cfg.end, dim=1))):
            break
        # next decoder input: value + coord[t%6] + pos[t//6]  (models.py:120-132)
        x_in = (p["input_embeddings.input_value.weight"][tok]
                + p["query_coord_embedding.weight"][t % cfg.out_dof]
                + p["query_pos_embedding.weight"][t // cfg.out_dof])
    if return_margins:
        return out, att, torch.stack(margins, dim=1)
    return out, att


def last_row_dist(p, cfg: OracleCfg, hid, eps=1e-6):
    """Row sz-1 of :func:`create_dist_eval` computed from the hidden prefix only."""
    sz = hid.shape[1]
    i = sz - 1
    h = hid[:, i]
    vd = softmax_lastdim(linear(h, p["vocab_head.weight"], p["vocab_head.bias"]))
    if sz < 6:
        return vd
    feat = linear(h, p["pointer_head.weight"], p["pointer_head.bias"])
    ptr = torch.einsum("bd,bjd->bj", feat, hid) / cfg.d_model
    prob = torch.sigmoid(linear(h, p["switch_head.weight"], p["switch_head.bias"]))
    ptr = ptr.clone()
    ptr[:, i:] = NEG_INF
    pd = softmax_lastdim(ptr) * prob
    allowed = pointer_mask(cfg, sz)[i] != 0
    pd = torch.where(allowed[None], pd, torch.full_like(pd, eps))
    return torch.cat((vd * (1 - prob), pd), dim=-1)


def parse_sequence(cfg: OracleCfg, seq):
    """reference models.py:258-265."""
    valid = torch.cumsum(seq == cfg.end, 0) == 0
    s = seq[valid]
    n = len(s) // cfg.out_dof
    return s[: n * cfg.out_dof].reshape(-1, cfg.out_dof)
