import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plankassembly_amd import ops
torch.manual_seed(0)
def ref(q,k,v,H,causal):
    B,Lq,dm=q.shape; Lk=k.shape[1]; dh=dm//H
    qh=q.float().view(B,Lq,H,dh).transpose(1,2); kh=k.float().view(B,Lk,H,dh).transpose(1,2); vh=v.float().view(B,Lk,H,dh).transpose(1,2)
    s=qh@kh.transpose(-1,-2)/math.sqrt(dh)
    if causal: s=s+torch.triu(torch.full((Lq,Lk),float("-inf")),1)
    return (torch.softmax(s,-1)@vh).transpose(1,2).reshape(B,Lq,dm)
for dh in (16,32,64):
    for (Lq,Lk,causal) in ((64,64,False),(64,128,False),(64,192,False),(130,130,True),(64,100,False),(64,256,False)):
        H=2; B=1; dm=H*dh
        q=torch.randn(B,Lq,dm).to(torch.bfloat16); k=torch.randn(B,Lk,dm).to(torch.bfloat16); v=torch.randn(B,Lk,dm).to(torch.bfloat16)
        o,_=ops.attn_fwd(q.cuda(),k.cuda(),v.cuda(),H,causal=causal)
        r=ref(q,k,v,H,causal)
        err=(o.float().cpu()-r).abs()
        rows=err.view(B,Lq,dm).amax(dim=(0,2))
        bad=(rows>0.05).nonzero().flatten().tolist()
        print(f"dh={dh} Lq={Lq} Lk={Lk} causal={causal}: max err {float(err.max()):.3g}; bad rows {bad[:6]}..{bad[-3:] if bad else ''} n={len(bad)}")
