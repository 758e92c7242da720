"""CPU simulation (analysis helper, not a test): how far can a bf16 greedy decode agree with the f32 reference's tokens,
and what would keeping the residual stream / LayerNorm / hidden cache / head logits in f32 buy?

The f32 oracle's cached decode loop (oracle/plank_oracle.py:339) is re-run with roundings to bf16 inserted where a bf16 device
path has them:
  every mode      weights bf16; every GEMM's A operand bf16 (MFMA inputs), f32 accumulation; Q / K / V and the K / V caches
                  bf16; attention output bf16; FFN hidden bf16
  'all_bf16'      additionally every tensor that crosses a kernel boundary is bf16: residual sums, LayerNorm outputs, the
                  hidden cache (what the HIP bf16 path stores today)
  'f32_resid'     residual stream, LayerNorm inputs / outputs and the hidden cache stay f32 (rounded only as a GEMM operand)
  'f32_heads'     = f32_resid + vocabulary / pointer / switch heads evaluated in f32 on the f32 hidden rows
Reported per mode: exact-prefix agreement with the f32 tokens, rows exact to the end, and the oracle's relative top-2 margin
at every row's first flip.      python tests/bf16_decode_sim.py [batch] [steps]"""
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import large_cases as LC                                   # noqa: E402
from oracle import plank_oracle as O                       # noqa: E402


def rb(x):
    return x.to(torch.bfloat16).to(torch.float32)


MODES = {   # name: (encoder, decoder residual path, vocabulary head, pointer head incl. the hidden cache it reads)
    "all_bf16": ("bf16", "bf16", "bf16", "bf16"),
    "f32_resid": ("resid", "resid", "bf16", "bf16"),
    "f32_heads": ("resid", "resid", "f32", "f32"),
    # which part matters: one change at a time from all_bf16, and the cheapest sufficient combination
    "enc_exact_f32_only": ("f32", "bf16", "bf16", "bf16"),
    "dec_resid_only": ("bf16", "resid", "bf16", "bf16"),
    "heads_only": ("bf16", "bf16", "f32", "f32"),
    "dec_resid+heads": ("bf16", "resid", "f32", "f32"),
    "enc_f32+dec_resid+heads": ("f32", "resid", "f32", "f32"),
    "enc_f32+dec_resid+vocab_f32": ("f32", "resid", "f32", "bf16"),
    "dec_resid+vocab_f32": ("bf16", "resid", "f32", "bf16"),
}


def run(sd, cfg, batch, mode, steps):
    enc_mode, dec_mode, vocab_mode, ptr_mode = MODES[mode]
    act = rb if enc_mode == "bf16" else (lambda x: x)       # tensors that cross a kernel boundary on the residual path (encoder)
    W = {k: (rb(v) if (v.dim() > 1 and "embedding" not in k) else v) for k, v in sd.items()}   # Linear weights bf16; tables / biases f32

    def lin(x, w, b):                                       # bf16 operands, f32 accumulate, f32 result (caller rounds)
        return rb(x) @ W[w].t() + sd[b]

    def ln(x, w, b, eps):
        return O.layer_norm(x, sd[w], sd[b], eps)

    d, H = cfg.d_model, cfg.n_head
    dh = d // H
    scale = 1.0 / math.sqrt(dh)

    def attend(q, k, v, mask):                              # q [B,H,1,dh] bf16 values, k / v bf16 caches
        s = (q @ k.transpose(-1, -2)) * scale
        if mask is not None:
            s = s + mask
        return rb((O.softmax_lastdim(s) @ v))

    # ---- encoder (the bf16 training-style forward in eval mode)
    x = act(O.embed_input(sd, batch))
    B, S, _ = x.shape
    kpm = O.key_padding_additive(batch["input_mask"])
    for i in range(cfg.n_enc):
        pre = f"encoder.layers.{i}."
        qkv = rb(lin(x, pre + "self_attn.in_proj_weight", pre + "self_attn.in_proj_bias"))
        q, k, v = [t.view(B, S, H, dh).transpose(1, 2) for t in qkv.split(d, dim=-1)]
        s = (q @ k.transpose(-1, -2)) * scale + kpm
        o = rb(O.softmax_lastdim(s) @ v).transpose(1, 2).reshape(B, S, d)
        z = act(x + lin(o, pre + "self_attn.out_proj.weight", pre + "self_attn.out_proj.bias"))
        x = act(ln(z, pre + "norm1.weight", pre + "norm1.bias", cfg.eps_layer))
        h = rb(torch.relu(lin(x, pre + "linear1.weight", pre + "linear1.bias")))
        z = act(x + lin(h, pre + "linear2.weight", pre + "linear2.bias"))
        x = act(ln(z, pre + "norm2.weight", pre + "norm2.bias", cfg.eps_layer))
    memory = act(ln(x, "encoder.norm.weight", "encoder.norm.bias", 1e-5)) if cfg.normalize_before else x
    if enc_mode == "f32":                                   # the encoder prologue on the exact-f32 path (once per decode)
        memory = O.encode(sd, cfg, batch)
    act = rb if dec_mode == "bf16" else (lambda x: x)       # ... and in the decoder
    cross = []
    for i in range(cfg.n_dec):
        pre = f"decoder.layers.{i}.multihead_attn."
        kv = rb(rb(memory) @ W[pre + "in_proj_weight"][d:].t() + sd[pre + "in_proj_bias"][d:])
        cross.append((kv[..., :d].view(B, S, H, dh).transpose(1, 2), kv[..., d:].view(B, S, H, dh).transpose(1, 2)))
    self_k = [torch.zeros(B, H, steps, dh) for _ in range(cfg.n_dec)]
    self_v = [torch.zeros(B, H, steps, dh) for _ in range(cfg.n_dec)]
    hid = torch.zeros(B, steps, d)
    out = torch.empty(B, 0, dtype=torch.long)
    att = torch.empty(B, 0, dtype=torch.long)
    x_in = torch.zeros(B, d)
    for t in range(steps):
        x = act(x_in)
        for i in range(cfg.n_dec):
            pre = f"decoder.layers.{i}."
            qkv = rb(lin(x, pre + "self_attn.in_proj_weight", pre + "self_attn.in_proj_bias"))
            q = qkv[:, :d].view(B, H, 1, dh)
            self_k[i][:, :, t] = qkv[:, d:2 * d].view(B, H, dh)
            self_v[i][:, :, t] = qkv[:, 2 * d:].view(B, H, dh)
            o = attend(q, self_k[i][:, :, :t + 1], self_v[i][:, :, :t + 1], None).reshape(B, d)
            z = act(x + lin(o, pre + "self_attn.out_proj.weight", pre + "self_attn.out_proj.bias"))
            x = act(ln(z, pre + "norm1.weight", pre + "norm1.bias", cfg.eps_layer))
            w, b = pre + "multihead_attn.in_proj_weight", pre + "multihead_attn.in_proj_bias"
            q = rb(rb(x) @ W[w][:d].t() + sd[b][:d]).view(B, H, 1, dh)
            o = attend(q, cross[i][0], cross[i][1], kpm).reshape(B, d)
            z = act(x + lin(o, pre + "multihead_attn.out_proj.weight", pre + "multihead_attn.out_proj.bias"))
            x = act(ln(z, pre + "norm2.weight", pre + "norm2.bias", cfg.eps_layer))
            h = rb(torch.relu(lin(x, pre + "linear1.weight", pre + "linear1.bias")))
            z = act(x + lin(h, pre + "linear2.weight", pre + "linear2.bias"))
            x = act(ln(z, pre + "norm3.weight", pre + "norm3.bias", cfg.eps_layer))
        x = act(ln(x, "decoder.norm.weight", "decoder.norm.bias", 1e-5))
        hid[:, t] = x
        # heads (oracle last_row_dist with the path's precision)
        # vocabulary head on the current row; pointer head = feature of the current row against the cached rows
        hv = x if vocab_mode == "f32" else rb(x)
        vd = O.softmax_lastdim(hv @ (sd if vocab_mode == "f32" else W)["vocab_head.weight"].t() + sd["vocab_head.bias"])
        hp = hid[:, :t + 1] if ptr_mode == "f32" else rb(hid[:, :t + 1])
        if t + 1 < 6:
            dist = vd
        else:                                               # oracle/plank_oracle.py last_row_dist with the two heads' own precisions
            feat = hp[:, t] @ (sd if ptr_mode == "f32" else W)["pointer_head.weight"].t() + sd["pointer_head.bias"]
            ptr = torch.einsum("bd,bjd->bj", feat, hp) / cfg.d_model
            prob = torch.sigmoid(O.linear(x, sd["switch_head.weight"], sd["switch_head.bias"]))
            ptr[:, t:] = O.NEG_INF
            pd_ = O.softmax_lastdim(ptr) * prob
            allowed = O.pointer_mask(cfg, t + 1)[t] != 0
            pd_ = torch.where(allowed[None], pd_, torch.full_like(pd_, 1e-6))
            dist = torch.cat((vd * (1 - prob), pd_), dim=-1)
        tok, ptr_i = O.sample(cfg, dist, out)
        out = torch.cat((out, tok[:, None]), dim=1)
        att = torch.cat((att, ptr_i[:, None]), dim=1)
        x_in = (sd["input_embeddings.input_value.weight"][tok] + sd["query_coord_embedding.weight"][t % cfg.out_dof]
                + sd["query_pos_embedding.weight"][t // cfg.out_dof])
    return out, att


def main():
    Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    c = LC.CASES["headline"]
    sd = LC.case_state_dict(c)
    cfg = LC.case_oracle_cfg(c)
    db = LC.case_batch(c, decode=True, batch_size=Bn)
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else c["max_out"]
    with torch.no_grad():
        s_ref, a_ref, marg = O.greedy_decode_cached(sd, cfg, db, max_steps=steps, early_stop=False, return_margins=True)
        for mode in (sys.argv[3].split(",") if len(sys.argv) > 3 else list(MODES)):
            s, a = run(sd, cfg, db, mode, steps)
            first, flips = [], []
            for i in range(Bn):
                neq = (s[i] != s_ref[i]) | (a[i] != a_ref[i])
                t = int(neq.nonzero()[0]) if bool(neq.any()) else steps
                first.append(t)
                if t < steps:
                    flips.append(float(marg[i, t]))
            agree = sum(first) / (Bn * steps)
            print(f"{mode:10s}: exact-prefix agreement {agree:.3f}, rows exact to the end {sum(t == steps for t in first)}/{Bn}, "
                  f"first flips at steps {first}, oracle margins there {['%.1e' % m for m in flips]}", flush=True)


if __name__ == "__main__":
    main()
