"""CPU study: the policy's attention probabilities (unnormalised p = exp(s - max), in (0, 1]) rounded ONCE to fp16 instead of hi + lo:
action error against the exact float64 policy, README geometry, synthetic weights, unit-variance tokens (CPU, ~1 minute; round 6).
Result: action MAE 1.07e-4, max 6.4e-4, gripper logit max 4.4e-4 -- as much as the whole image encoder spends -- so the four vector
instructions per probability that the hi + lo split costs policy_kernel are not available for an instruction diet."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'hyper-vla_amd'))
from oracle import hvla_ref_np as onp
from hypervla import synthetic as syn
from hypervla.config import FULL, generated_leaves, encoder_leaves
g = FULL
B = 16
P = syn.synthetic_params(g)
ins, st = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g)
bp, _ = onp.create_tasks(P, g, generated_leaves(g), ins, st)
rng = np.random.default_rng(0)
tokens = rng.standard_normal((B, g.patches, g.enc_dim))
exact = onp.policy(bp, g, tokens)
orig = onp.mha
def mha_q(x, mask, p, prefix, heads, sink=None, tag=""):
    F = np.float64
    B_, S, Dm = x.shape
    q = np.einsum("bsd,dhk->bshk", x, np.asarray(p[prefix + "query/kernel"], F)) + np.asarray(p[prefix + "query/bias"], F)
    k = np.einsum("bsd,dhk->bshk", x, np.asarray(p[prefix + "key/kernel"], F)) + np.asarray(p[prefix + "key/bias"], F)
    v = np.einsum("bsd,dhk->bshk", x, np.asarray(p[prefix + "value/kernel"], F)) + np.asarray(p[prefix + "value/bias"], F)
    hd = q.shape[-1]
    q = q / np.sqrt(hd)
    w = np.einsum("bqhd,bkhd->bhqk", q, k)
    w = np.where(np.asarray(mask) != 0, w, onp.F32_MIN)
    w = w - w.max(-1, keepdims=True)
    w = np.exp(w)
    den = w.sum(-1, keepdims=True)                       # the kernel adds the f32 values up, rounds only the MFMA operand
    w16 = w.astype(np.float16).astype(F)
    o = np.einsum("bhqk,bkhd->bqhd", w16, v) / den.transpose(0, 2, 1, 3)
    return np.einsum("bqhd,hdo->bqo", o, np.asarray(p[prefix + "out/kernel"], F)) + np.asarray(p[prefix + "out/bias"], F)
onp.mha = mha_q
got = onp.policy(bp, g, tokens)
onp.mha = orig
a0, l0 = exact[0], exact[1]
a1, l1 = got[0], got[1]
d = np.abs(a1[..., :6] - a0[..., :6])
print("P rounded once to fp16: action MAE %.3e max %.3e; logit max %.3e" % (d.mean(), d.max(), np.abs(l1 - l0).max()))
