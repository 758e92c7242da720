"""Run the encoder self-attention forward/backward at the bench shape a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plankassembly_amd import ops
B, S, D, H = 16, 1024, 512, 8
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(B, S, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
valid = torch.randint(S // 2, S + 1, (B,), device="cuda", generator=g)
kpm = torch.arange(S, device="cuda")[None] >= valid[:, None]
do = torch.randn(B, S, D, device="cuda", generator=g).to(torch.bfloat16)
for _ in range(5):
    o, lse = ops.attn_fwd(q, k, v, H, kpm=kpm, drop_p=0.2, drop_seed=1)
    ops.attn_bwd(do, q, k, v, o, lse, H, kpm=kpm, drop_p=0.2, drop_seed=1)
torch.cuda.synchronize()
print("ok")
