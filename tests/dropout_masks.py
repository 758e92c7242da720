"""The HIP library's dropout decisions, restated in numpy (TEST INFRASTRUCTURE).

Dropout in `plankassembly_amd/csrc` is counter-based: every decision is a pure function of (step seed, site, element
index), so the forward, the backward kernels that regenerate it and - here - a test can all compute the same mask.  This
module restates those functions (csrc/pa_device.h `mix32`, `drop_keep_rc`, `drop_row_hash`, `drop_key_hash`, `drop_keep2`;
csrc/model.h `site_seed`; the site numbering of csrc/runtime.hip `forward_enc_layer` / `forward_dec_layer`; the step-seed
recurrence of models.py `PlankModel.forward`) and hands the masks to the CPU oracle (`oracle.plank_oracle._drop`), so that a
training step UNDER dropout 0.2 - the benchmarked mode - is compared with the oracle tensor by tensor.
tests/test_kernels_gpu.py pins every function below against decisions extracted from the kernels themselves."""
import numpy as np
import torch

M32 = np.uint64(0xFFFFFFFF)


def mix32(x):
    x = np.asarray(x).astype(np.uint64) & M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def site_seed(base, site):
    return int(mix32(np.uint64((int(base) + 0x9E3779B9 * (int(site) + 1)) & 0xFFFFFFFF)))


def next_step_seed(prev, initial_seed):
    """models.py PlankModel.forward: the seed the NEXT training forward will hand to pa_model_train_fwd."""
    return (int(prev) * 1664525 + 1013904223 + int(initial_seed)) & 0xFFFFFFFF


def linear_keep(seed, rows, n_cols, p):
    """Epilogue dropout of a Linear whose output is [n_rows][n_cols] (pa_device.h drop_keep_rc, round 5): the SUM of two
    independent separable products, keep(row, col) <=> low32(A1[row] * C1[col] + A2[row] * C2[col]) >= p * 2^32, with A1 / C1 the
    attention dropout's 24-bit odd hashes of the output row (batch folded in: b * M + m) and the output column (bits 0..23 of a 32-bit
    mix) and A2 / C2 the top 24 bits of one more mixing round on those mixes (32 free bits per side: no two rows or columns share a mask).
    ``rows``: the output-row index of every row wanted (packed row numbers for the packed encoder).  bool [len(rows), n_cols]."""
    thr32 = np.uint64(int(float(np.float32(p)) * 4294967296.0))
    r = np.asarray(rows, dtype=np.uint64)
    k = np.arange(n_cols, dtype=np.uint64)
    odd = lambda x: (x & np.uint64(0xFFFFFF)) | np.uint64(0x800001)
    mr = mix32((r * np.uint64(0x9E3779B9) + np.uint64(seed)) & M32)
    mc = mix32((k * np.uint64(0x85EBCA6B) + np.uint64(seed ^ 0x5BD1E995)) & M32)
    a1, c1 = odd(mr), odd(mc)                                   # bits 0..23 of the mixes: the attention dropout's hashes
    mr2 = ((mr ^ (mr >> np.uint64(13))) * np.uint64(0x846CA68B)) & M32     # one more xorshift-multiply round each
    mc2 = ((mc ^ (mc >> np.uint64(13))) * np.uint64(0x7FEB352D)) & M32
    a2, c2 = odd(mr2 >> np.uint64(8)), odd(mc2 >> np.uint64(8))
    h = ((a1[..., None] * c1) + (a2[..., None] * c2)) & M32
    return h >= thr32


def linear_scale(p):
    return attn_scale(p)


def attn_keep(seed, row_index, n_keys, p):
    """Attention-probability dropout: keep(row, key) <=> low32(A[row] * C[key]) >= p * 2^32 (pa_device.h drop_keep2).
    ``row_index``: global query-row numbers ((b * H + h) * Lq + q) of the rows wanted, any shape.  bool [*row_index.shape, n_keys]."""
    thr32 = np.uint64(int(float(np.float32(p)) * 4294967296.0))
    r = np.asarray(row_index, dtype=np.uint64)
    a = (mix32((r * np.uint64(0x9E3779B9) + np.uint64(seed)) & M32) & np.uint64(0xFFFFFF)) | np.uint64(0x800001)
    k = np.arange(n_keys, dtype=np.uint64)
    c = (mix32((k * np.uint64(0x85EBCA6B) + np.uint64(seed ^ 0x5BD1E995)) & M32) & np.uint64(0xFFFFFF)) | np.uint64(0x800001)
    return ((a[..., None] * c) & M32) >= thr32


def attn_scale(p):
    thr32 = int(float(np.float32(p)) * 4294967296.0)
    return float(np.float32(1.0 / (1.0 - thr32 / 4294967296.0)))


def _mix32_t(x):
    """mix32 on int64 torch tensors (values < 2^32): multi-threaded, for the B = 16 x 8 heads x 1024 x 1024 attention masks."""
    m = 0xFFFFFFFF
    x = x ^ (x >> 16); x = (x * 0x7FEB352D) & m
    x = x ^ (x >> 15); x = (x * 0x846CA68B) & m
    return x ^ (x >> 16)


def attn_keep_torch(seed, B, H, Lq, Lk, p):
    """attn_keep for the dense row numbering (b * H + h) * Lq + q, as a bool torch tensor [B, H, Lq, Lk]."""
    thr32 = int(float(np.float32(p)) * 4294967296.0)
    r = torch.arange(B * H * Lq, dtype=torch.int64)
    a = (_mix32_t((r * 0x9E3779B9 + seed) & 0xFFFFFFFF) & 0xFFFFFF) | 0x800001
    k = torch.arange(Lk, dtype=torch.int64)
    c = (_mix32_t((k * 0x85EBCA6B + (seed ^ 0x5BD1E995)) & 0xFFFFFFFF) & 0xFFFFFF) | 0x800001
    out = torch.empty(B * H * Lq, Lk, dtype=torch.bool)
    step = max(1, (1 << 24) // Lk)                                   # bound the int64 temporaries to 128 MB
    for i in range(0, B * H * Lq, step):
        out[i:i + step] = ((a[i:i + step, None] * c[None, :]) & 0xFFFFFFFF) >= thr32
    return out.view(B, H, Lq, Lk)


class HipDropout:
    """Callable (site, tensor) -> tensor for `oracle.plank_oracle.train_forward(..., drop=...)`: applies the masks the HIP
    training step with step seed ``seed`` uses.  ``input_mask`` (bool [B, S], True = PAD) fixes the packed row numbers of the
    encoder rows when ``packed`` (models.py `unpad`); decoder rows are b * T + t."""

    def __init__(self, seed, p, n_head, input_mask, packed=True):
        self.seed, self.p, self.H = int(seed), float(p), int(n_head)
        valid = ~np.asarray(input_mask, dtype=bool)
        B, S = valid.shape
        if packed:
            flat = np.cumsum(valid.reshape(-1)).reshape(B, S) - 1          # packed row of every valid token, batch order
            self.enc_rows = np.where(valid, flat, 0)
        else:
            self.enc_rows = np.arange(B * S).reshape(B, S)
        self.enc_valid = valid
        self.sites_seen = []

    def _site(self, name):
        part = name.split(".")
        layer, what = int(part[2]), part[3]
        if part[0] == "encoder":
            return 8 * layer + {"self_attn": 0, "dropout1": 1, "dropout": 2, "dropout2": 3}[what], True
        return 1000 + 8 * layer + {"self_attn": 0, "dropout1": 1, "multihead_attn": 2, "dropout2": 3, "dropout": 4, "dropout3": 5}[what], False

    def __call__(self, name, x):
        site, enc = self._site(name)
        seed = site_seed(self.seed, site)
        self.sites_seen.append(name)
        if name.endswith("attn"):
            B, H, Lq, Lk = x.shape
            return x * attn_keep_torch(seed, B, H, Lq, Lk, self.p).to(x.dtype) * attn_scale(self.p)
        B, L, N = x.shape
        rows = self.enc_rows if enc else np.arange(B * L).reshape(B, L)
        keep = linear_keep(seed, rows.reshape(-1), N, self.p).reshape(B, L, N)
        if enc:
            keep = keep | ~self.enc_valid[:, :, None]                      # padded rows never reach the loss
        return x * torch.from_numpy(keep).to(x.dtype) * linear_scale(self.p)
