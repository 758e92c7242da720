import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plankassembly_amd import ops
dh=32; H=1; B=1; dm=H*dh; Lq=32; Lk=128
q=torch.zeros(B,Lq,dm).to(torch.bfloat16); k=torch.zeros(B,Lk,dm).to(torch.bfloat16)
bad={}
for key in range(Lk):
    v=torch.zeros(B,Lk,dm); v[0,key,:]=torch.arange(1,dm+1).float()
    o,_=ops.attn_fwd(q.cuda(),k.cuda(),v.to(torch.bfloat16).cuda(),H)
    got=(o[0,0].float().cpu()*Lk).round().int().tolist()
    if got!=list(range(1,dm+1)): bad[key]=got
print("bad keys", sorted(bad))
for kk in sorted(bad)[:12]: print(kk, bad[kk])
