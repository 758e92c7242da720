"""How long is ONE block's serial chain?  Packed self-attention launches whose blocks are alone on their CU (1, 2, 4 elements
of one length: 8 q-tiles x 8 heads x B blocks <= 256) against full launches: if the lone chain already takes the packed
launch's time, the launch is bound by the chain's own latency, not by what shares its CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plankassembly_amd import ops
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "attn_balance.py")).read().split("for seed in")[0])
for L in (961, 513, 257, 129):
    for B in (1, 2, 4, 12, 16):
        run(f"{B} x {L}", [L] * B)
    run(f"1 x {L} no dropout", [L], drop=0.0)
