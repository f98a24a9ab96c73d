"""Diagnostic (GPU box): run-to-run stability of the T5 encoder and of the fine-tune forward (loss) / backward (gradients)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import numpy as np, torch
from hypervla import synthetic as syn
from hypervla.config import FULL, T5_MID
from hypervla.model import HyperVLA
from hypervla.train import FineTuner
g, B = FULL, 8
m = HyperVLA.from_synthetic(g, max_batch=B)
m.load_language_encoder(syn.synthetic_t5_params(T5_MID), T5_MID)
tok = syn.synthetic_token_ids(B, T5_MID, g.lang_tokens)
ref = m.encode_instructions(tok)["token_embedding"].clone()
bad = sum(int(not torch.equal(m.encode_instructions(tok)["token_embedding"], ref)) for _ in range(200))
print("T5: runs differing", bad, "of 200")
ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
batch = syn.synthetic_action_batch(B, g)
for enc in (False, True):
    ft = FineTuner(m, B, train_encoder=enc)
    x = im if enc else m.encode_images(im)
    l0 = ft.forward_backward(ins, st, x, batch).clone(); g0 = ft.grads.clone()
    badl = badg = 0; worst = 0.0
    for _ in range(30):
        l = ft.forward_backward(ins, st, x, batch)
        badl += int(not torch.equal(l, l0))
        d = float((ft.grads - g0).abs().max() / g0.abs().max())
        badg += int(d > 0); worst = max(worst, d)
    print(f"fine-tune (train_encoder={enc}): loss differing {badl}/30, grads differing {badg}/30 (worst rel {worst:.1e}; split-K atomics are expected to move last bits)")
