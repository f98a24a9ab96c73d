"""Diagnostic (GPU box): which stage makes an episode of a B=256 batch differ from the same episode in a batch of 64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import numpy as np, torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA
g, B = FULL, 256
m = HyperVLA.from_synthetic(g, max_batch=B)
ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
li = ins["language_instruction"]
def run(idx):
    sub_ins = {"language_instruction": {k: np.asarray(v)[idx] for k, v in li.items()}}
    sub_st = {"patch_embeddings": st["patch_embeddings"][idx], "pad_mask_dict": {"image_primary": np.ones((len(idx), 1))}}
    w, tasks, _ = m.create_tasks(instruction_dict=sub_ins, initial_state=sub_st)
    theta, ctx = [t.cpu().numpy() for t in w.export()]
    tok = m.encode_images(im[idx])
    a, l = m.policy_from_tokens(tok, w)
    return ctx, theta, tok.cpu().numpy(), a.cpu().numpy()
full = run(np.arange(B))
for lo in (0, 64, 128, 192):
    sub = run(np.arange(lo, lo + 64))
    for nm, f, s in zip(("ctx", "theta", "tokens", "actions"), full, sub):
        d = np.abs(f[lo:lo + 64] - s).reshape(64, -1).max(1)
        print(lo, nm, "episodes differing:", np.nonzero(d)[0] + lo, "max", d.max())
