import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from plankassembly_amd import ops
from plankassembly_amd.data import synth_batch
cfgd = bench.CONFIGS["headline"]
model = bench.build("bf16", cfgd["max_in"], cfgd["max_out"], 0.2, cfgd).train()
b = synth_batch(16, bench.cfg_spec(cfgd), seed=2022, device="cuda"); b.pop("name")
pb = model.prepare_batch(b)
cu, rowmap, n = pb["_pack"]
B = 16
print("cu", cu.tolist(), "base", None if cu._base is None else cu._base.shape)
order = ops.pack_order(cu)
lens = (cu[1:B + 1] - cu[:B]).tolist()
print("order", order.tolist()); print("lens", lens, "argsort", np.argsort(-np.array(lens), kind="stable").tolist())
d = 512
g = torch.Generator(device="cuda").manual_seed(3)
qkv = torch.randn(n, 3 * d, device="cuda", generator=g).to(torch.bfloat16)
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
fl = sum(4.0 * l * l * d for l in lens)
cu2, order2 = ops.pack_lengths(lens, "cuda")
for name, c_, o_ in (("pack_rows", cu, order), ("pack_lengths", cu2, order2), ("cu from pack_rows + argsort order", cu, order2)):
    t = bench.time_kernel(lambda: ops.attn_varlen_fwd(q, k, v, 8, c_, c_, B, 1024, 1024, order=o_, drop_p=0.2, drop_seed=5))
    print(name, "fwd %.1f us %.0f TF" % (t * 1e6, fl / t / 1e12))
