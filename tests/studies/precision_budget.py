"""CPU study: which 16-bit rounding sites of the GPU image encoder cost how much action error.

    python tests/studies/precision_budget.py [--episodes 8] [--style synthetic|trained] [--sites all|h,w,qkv,p,o,g]

The float64 numpy/torch restatement (oracle/) is run once exactly and once per configuration with the GPU path's
rounding emulated at the chosen sites (operands rounded to fp16 / bf16, everything else float64), and the predicted
actions are compared (policy evaluated exactly in both runs, so only the encoder's rounding shows).  Diagnostic tool:
imports oracle/ and is never imported by the product.

Sites (per encoder layer):  h = LayerNorm outputs (A operand of QKV and fc1),  w = all four weight matrices,
qkv = stored q / k / v,  p = softmax probabilities,  o = attention output (A operand of out-proj),
g = GELU output (A operand of fc2).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "hyper-vla_amd"), ROOT):
    sys.path.insert(0, p)

from hypervla import synthetic as syn                                      # noqa: E402
from hypervla.config import FULL, MID, encoder_leaves, generated_leaves    # noqa: E402
from oracle import hvla_ref_np as onp                                      # noqa: E402

F = torch.float64


def rounder(kind):
    if kind == "f16":
        return lambda t: t.to(torch.float16).to(F)
    if kind == "bf16":
        return lambda t: t.to(torch.bfloat16).to(F)
    if kind == "f16x2":       # hi + lo fp16 pair (two MFMAs per product on that operand)
        def r(t):
            hi = t.to(torch.float16).to(F)
            return hi + (t - hi).to(torch.float16).to(F)
        return r
    return lambda t: t


def encoder(hp, g, enc_shapes, images_u8, sites, kind="f16", layer_kinds=None, layers_on=None, corr=None, collect=None, fold=False,
            exact_w=()):
    # exact_w: {(layer, "wq" | "wo" | "w1" | "w2")}: that matrix is NOT rounded (hi + lo 16-bit planes: two MFMA passes)
    """DINOv2 forward in float64 with the listed operand sites rounded (oracle/hvla_ref_np.py::dinov2 restated on torch
    for speed)."""
    E, H = g.enc_dim, g.enc_heads
    hd = E // H

    def get(path):
        return torch.as_tensor(np.asarray(hp["encoder_image_encoder_" + "_".join(path)], np.float64).reshape(enc_shapes[path]))

    x = torch.as_tensor(onp.normalize_images(images_u8))
    B = x.shape[0]
    p_, G = g.patch, g.grid
    x = x.reshape(B, G, p_, G, p_, 3).permute(0, 1, 3, 2, 4, 5).reshape(B, G * G, p_ * p_ * 3)
    x = x @ get(("embeddings", "patch_embeddings", "projection", "kernel")).reshape(p_ * p_ * 3, E) + get(("embeddings", "patch_embeddings", "projection", "bias"))
    x = torch.cat([get(("embeddings", "cls_token")).expand(B, 1, E), x], 1) + get(("embeddings", "position_embeddings"))
    stats = {}

    def ln(x, s, b):
        m = x.mean(-1, keepdim=True)
        v = (x * x).mean(-1, keepdim=True) - m * m
        return (x - m) * torch.rsqrt(v.clamp_min(0) + 1e-6) * s + b

    def mm(a, W, Wr, key):
        """a @ round(W), plus the first-order compensation of the weight rounding: mean activation x (W - round(W)),
        with the mean taken per image (dynamic) or from a calibration run (static: a bias offset)."""
        y = a @ Wr(W)
        if collect is not None:
            collect[key] = a.mean((0, 1))
        if corr == "dynamic":
            y = y + a.mean(1, keepdim=True) @ (W - Wr(W))
        elif isinstance(corr, tuple) and corr[0] == "bands":          # ("bands", n, layers): the mean row per band of 256 / n patch rows
            nb_, lay = corr[1], corr[2]
            if key[0] in lay:
                P_ = a.shape[1] - 1
                mb = a[:, 1:].reshape(a.shape[0], nb_, P_ // nb_, a.shape[2]).mean(2, keepdim=True).expand(-1, -1, P_ // nb_, -1).reshape(a.shape[0], P_, a.shape[2])
                mrow = torch.cat([a.mean(1, keepdim=True), mb], 1)
            else:
                mrow = a.mean(1, keepdim=True)
            y = y + mrow @ (W - Wr(W))
        elif isinstance(corr, dict):
            y = y + corr[key] @ (W - Wr(W))
        return y

    def ln_mm(x, s, b, W, bias, Wr, Xr):
        """LayerNorm folded into the GEMM that consumes it (study "foldln"): the 16-bit operand is x itself, the LN scale
        is folded into W, mean / rstd come in as a rank-1 correction of the f32 accumulator, the weight rounding is
        compensated per image with the mean row of x:  r (x16 W'16 + xbar dW' - mu colsum(W')) + (b + beta W)."""
        m = x.mean(-1, keepdim=True)
        v = (x * x).mean(-1, keepdim=True) - m * m
        r = torch.rsqrt(v.clamp_min(0) + 1e-6)
        Wp = s[:, None] * W
        W16 = Wr(Wp)
        x16 = Xr(x)
        acc = x16 @ W16 + x16[:, 1:].mean(1, keepdim=True) @ (Wp - W16)
        return r * (acc - m * Wp.sum(0)) + (bias + b @ W)

    for i in range(g.enc_layers):
        k_i = layer_kinds[i] if layer_kinds else kind
        rd = rounder(k_i)
        act = sites if layers_on is None or i in layers_on else set()
        R = {s: (rd if (s in act or (s[0] == "w" and "w" in act)) and (i, s) not in exact_w else (lambda t: t)) for s in "h w wq wo w1 w2 qkv p o g".split()}
        L = ("encoder", "layer", str(i))
        if fold:
            Wqkv = torch.cat([get(L + ("attention", "attention", n_, "kernel")) for n_ in ("query", "key", "value")], 1)
            bqkv = torch.cat([get(L + ("attention", "attention", n_, "bias")) for n_ in ("query", "key", "value")], 0)
            qkv_ = ln_mm(x, get(L + ("norm1", "scale")), get(L + ("norm1", "bias")), Wqkv, bqkv, R["wq"], R["h"])
        h = R["h"](ln(x, get(L + ("norm1", "scale")), get(L + ("norm1", "bias"))))
        q = mm(h, get(L + ("attention", "attention", "query", "kernel")), R["wq"], (i, "h1")) + get(L + ("attention", "attention", "query", "bias"))
        k = mm(h, get(L + ("attention", "attention", "key", "kernel")), R["wq"], (i, "h1")) + get(L + ("attention", "attention", "key", "bias"))
        v = mm(h, get(L + ("attention", "attention", "value", "kernel")), R["wq"], (i, "h1")) + get(L + ("attention", "attention", "value", "bias"))
        if fold:
            q, k, v = qkv_[..., :E], qkv_[..., E:2 * E], qkv_[..., 2 * E:]
        q = R["qkv"](q * (np.log2(np.e) / 8.0)) / (np.log2(np.e) / 8.0) * (1.0 / 8.0) * 8.0   # stored pre-scaled by log2e/8
        k, v = R["qkv"](k), R["qkv"](v)
        q, k, v = (t.reshape(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
        w_ = (q @ k.transpose(-1, -2)) / np.sqrt(hd)
        w_ = torch.exp(w_ - w_.amax(-1, keepdim=True))
        den = w_.sum(-1, keepdim=True)
        o = (R["p"](w_) @ v) / den                      # GPU: P rounded, row sum in f32 of the unrounded exponentials
        o = R["o"](o.transpose(1, 2).reshape(B, -1, E))
        o = mm(o, get(L + ("attention", "output", "dense", "kernel")), R["wo"], (i, "o")) + get(L + ("attention", "output", "dense", "bias"))
        x = x + o * get(L + ("layer_scale1", "lambda1"))
        h = R["h"](ln(x, get(L + ("norm2", "scale")), get(L + ("norm2", "bias"))))
        a = mm(h, get(L + ("mlp", "fc1", "kernel")), R["w1"], (i, "h2")) + get(L + ("mlp", "fc1", "bias"))
        if fold:
            a = ln_mm(x, get(L + ("norm2", "scale")), get(L + ("norm2", "bias")), get(L + ("mlp", "fc1", "kernel")), get(L + ("mlp", "fc1", "bias")), R["w1"], R["h"])
        stats.setdefault("fc1_absmax", []).append(float(a.abs().max()))
        a = R["g"](0.5 * a * (1.0 + torch.erf(a / np.sqrt(2.0))))
        stats.setdefault("g_absmax", []).append(float(a.abs().max()))
        a = mm(a, get(L + ("mlp", "fc2", "kernel")), R["w2"], (i, "g")) + get(L + ("mlp", "fc2", "bias"))
        x = x + a * get(L + ("layer_scale2", "lambda1"))
        stats.setdefault("x_absmax", []).append(float(x.abs().max()))
    return ln(x, get(("layernorm", "scale")), get(("layernorm", "bias")))[:, 1:], stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=8)
    ap.add_argument("--style", default="synthetic")
    ap.add_argument("--geometry", default="full")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--study", default="sites", choices=["sites", "weights", "bias", "foldln", "hilo"])
    ap.add_argument("--images", default="noise", choices=["noise", "structured"])
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    g = FULL if a.geometry == "full" else MID
    B = a.episodes
    hp = syn.synthetic_params(g) if a.style == "synthetic" else syn.synthetic_params_trained_like(g)
    leaves, enc_shapes = generated_leaves(g), dict(encoder_leaves(g))
    ins, st = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g)
    im = syn.synthetic_images(B, g) if a.images == "noise" else syn.synthetic_images_structured(B, g)
    bp, _ = onp.create_tasks(hp, g, leaves, ins, st)
    t0 = time.time()
    tok0, stats = encoder(hp, g, enc_shapes, im[:, 0], set())
    act0, logit0, _ = onp.policy(bp, g, tok0.numpy())
    print(f"exact run {time.time() - t0:.1f}s; residual |x|max per layer {['%.1f' % v for v in stats['x_absmax']]}; "
          f"fc1 pre-act |max| {['%.1f' % v for v in stats['fc1_absmax']]}; gelu out |max| {['%.1f' % v for v in stats['g_absmax']]}", flush=True)
    allsites = {"h", "w", "qkv", "p", "o", "g"}
    runs = [("all f16", allsites, "f16", None)]
    for s in sorted(allsites):
        runs.append((f"only {s} f16", {s}, "f16", None))
    runs += [("all bf16", allsites, "bf16", None),
             ("all f16 but w", allsites - {"w"}, "f16", None),
             ("all f16 but h", allsites - {"h"}, "f16", None),
             ("all f16 but h,w", allsites - {"h", "w"}, "f16", None)]
    if a.study == "bias":
        cal = syn.synthetic_images(8, g, rank=77) if a.images == "noise" else syn.synthetic_images_structured(8, g, rank=77)
        means = {}
        encoder(hp, g, enc_shapes, cal[:, 0], allsites, "f16", collect=means)     # calibration run on OTHER images
        for name, corr in (("f16, no compensation", None), ("f16 + static bias corr", means), ("f16 + per-image corr", "dynamic")):
            tok, _ = encoder(hp, g, enc_shapes, im[:, 0], allsites, "f16", corr=corr)
            act, logit, _ = onp.policy(bp, g, tok.numpy())
            d = np.abs(act[..., :6] - act0[..., :6])
            dt = (tok - tok0).numpy()
            print(f"{name:24s} action MAE {d.mean():.2e} max {d.max():.2e} p99 {np.quantile(d, 0.99):.2e} | logit max {np.abs(logit - logit0).max():.2e} | "
                  f"token rms {np.sqrt((dt * dt).mean()):.2e} max {np.abs(dt).max():.2e}", flush=True)
        return
    if a.study == "hilo":
        # which matrices, kept exact (hi + lo planes, two MFMA passes) on top of the per-image compensation,
        # bring the worst action error of the structured fixture under 1e-3, and what each costs (ms per step at B = 256:
        # qkv 0.20, out 0.11, fc1 0.31, fc2 0.27 per layer)
        Ls = g.enc_layers
        cases = [("f16 + per-image corr (product)", set()),
                 ("+ fc2 exact, last 2 layers", {(l, "w2") for l in (Ls - 2, Ls - 1)}),
                 ("+ fc2 exact, last 4 layers", {(l, "w2") for l in range(Ls - 4, Ls)}),
                 ("+ fc1, fc2 exact, last 2 layers", {(l, m) for l in (Ls - 2, Ls - 1) for m in ("w1", "w2")}),
                 ("+ all four exact, last layer", {(Ls - 1, m) for m in ("wq", "wo", "w1", "w2")}),
                 ("+ all four exact, last 2 layers", {(l, m) for l in (Ls - 2, Ls - 1) for m in ("wq", "wo", "w1", "w2")}),
                 ("+ all four exact, first 2 layers", {(l, m) for l in (0, 1) for m in ("wq", "wo", "w1", "w2")}),
                 ("+ fc2 exact, all layers", {(l, "w2") for l in range(Ls)}),
                 ("+ out-proj exact, all layers", {(l, "wo") for l in range(Ls)}),
                 ("+ all exact (activations only)", {(l, m) for l in range(Ls) for m in ("wq", "wo", "w1", "w2")})]
        Lall = set(range(Ls))
        for name, cr in (("mean row per 128 rows, all layers", ("bands", 2, Lall)), ("mean row per 64 rows, all layers", ("bands", 4, Lall)),
                         ("mean row per 16 rows, all layers", ("bands", 16, Lall)), ("mean row per 64 rows, first 3 layers", ("bands", 4, {0, 1, 2})),
                         ("mean row per 16 rows, first 3 layers", ("bands", 16, {0, 1, 2}))):
            tok, _ = encoder(hp, g, enc_shapes, im[:, 0], allsites, "f16", corr=cr)
            act, logit, _ = onp.policy(bp, g, tok.numpy())
            d = np.abs(act[..., :6] - act0[..., :6])
            print(f"{name:36s} action MAE {d.mean():.2e} max {d.max():.2e} p99 {np.quantile(d, 0.99):.2e}", flush=True)
        for name, ex in cases:
            tok, _ = encoder(hp, g, enc_shapes, im[:, 0], allsites, "f16", corr="dynamic", exact_w=ex)
            act, logit, _ = onp.policy(bp, g, tok.numpy())
            d = np.abs(act[..., :6] - act0[..., :6])
            print(f"{name:36s} action MAE {d.mean():.2e} max {d.max():.2e} p99 {np.quantile(d, 0.99):.2e}", flush=True)
        return
    if a.study == "foldln":
        for name, kw in (("f16 + per-image corr (LN pass)", dict(corr="dynamic")), ("f16, LN folded into QKV / fc1", dict(corr="dynamic", fold=True))):
            tok, _ = encoder(hp, g, enc_shapes, im[:, 0], allsites, "f16", **kw)
            act, logit, _ = onp.policy(bp, g, tok.numpy())
            d = np.abs(act[..., :6] - act0[..., :6])
            dt = (tok - tok0).numpy()
            print(f"{name:32s} action MAE {d.mean():.2e} max {d.max():.2e} p99 {np.quantile(d, 0.99):.2e} | logit max {np.abs(logit - logit0).max():.2e} | "
                  f"token rms {np.sqrt((dt * dt).mean()):.2e} max {np.abs(dt).max():.2e}", flush=True)
        return
    if a.study == "weights":
        runs = [(f"only {s} f16", {s}, "f16", None, None) for s in ("wq", "wo", "w1", "w2")]
        runs += [(f"w f16 layers {lo}-{lo + 2}", {"w"}, "f16", None, set(range(lo, lo + 3))) for lo in (0, 3, 6, 9)]
        runs += [("all f16, w f16x2", allsites - {"w"}, "f16", None, None)]
    else:
        runs = [r + (None,) for r in runs]
    for name, sites, kind, lk, lon in runs:
        tok, _ = encoder(hp, g, enc_shapes, im[:, 0], sites, kind, lk, lon)
        act, logit, _ = onp.policy(bp, g, tok.numpy())
        d = np.abs(act[..., :6] - act0[..., :6])
        dt = (tok - tok0).numpy()
        print(f"{name:22s} action MAE {d.mean():.2e} max {d.max():.2e} | logit max {np.abs(logit - logit0).max():.2e} | "
              f"token rms {np.sqrt((dt * dt).mean()):.2e} max {np.abs(dt).max():.2e}", flush=True)


if __name__ == "__main__":
    main()
