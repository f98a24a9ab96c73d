"""Diagnostic (GPU box): when the policy kernel's actions differ from run to run, do the action token's attention rows
(exported per layer and head) differ too, and from which layer on?  Narrows a race down to the part of the kernel it is in."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import numpy as np, torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA
g = FULL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 300
m = HyperVLA.from_synthetic(g, max_batch=B)
ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
w, _, _ = m.create_tasks(instruction_dict=ins, initial_state=st)
tok = m.encode_images(im)
head = torch.zeros(B, g.layers, g.heads, g.patches, device=m.device)
m._ctx.set_attention_outputs(0, head.data_ptr())
a0, _ = m.policy_from_tokens(tok, w)
a0, h0 = a0.clone(), head.clone()
first = {}
nbad = 0
for it in range(runs):
    head.zero_()
    a, _ = m.policy_from_tokens(tok, w)
    da = (a - a0).abs().reshape(B, -1).max(1).values
    dh = (head - h0).abs().amax(-1)                    # [B, L, heads]
    for b in torch.nonzero(da > 0).flatten().tolist():
        nbad += 1
        if nbad <= 6:                                # per 32-key block (= the wave that scored it): how the layer-0 rows moved
            for hd in range(2):
                r = (head[b, 0, hd] / h0[b, 0, hd]).reshape(-1, 32)
                d = (head[b, 0, hd] - h0[b, 0, hd]).abs().reshape(-1, 32)
                print(f"  run {it} episode {b} layer 0 head {hd}: per-wave ratio new/old min..max",
                      [f"{float(r[wv].min()):.4f}..{float(r[wv].max()):.4f}" for wv in range(r.shape[0])],
                      "sum", float(head[b, 0, hd].sum()), float(h0[b, 0, hd].sum()))
        lay = [(l, [round(float(v), 6) for v in dh[b, l]]) for l in range(g.layers) if float(dh[b, l].max()) > 0]
        key = tuple((l, tuple(i for i, v in enumerate(vals) if v > 0)) for l, vals in lay)
        first[key] = first.get(key, 0) + 1
print(f"B={B}: {nbad} differing episode-runs of {runs * B}")
for k, v in sorted(first.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v:5d} x  layers / heads whose action-row attention differs: {k if k else 'none (attention rows identical)'}")
