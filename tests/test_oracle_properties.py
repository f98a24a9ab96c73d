"""CPU: invariants the reference *code* guarantees (SURVEY.md §8c pin 3), checked on the oracle."""
import numpy as np
import pytest

from hypervla import synthetic as syn
from hypervla.config import TINY, encoder_leaves, generated_leaves
from oracle import hvla_ref_np as onp

G = TINY
LEAVES, ENC = generated_leaves(G), dict(encoder_leaves(G))


def _inputs(B=3):
    return (syn.synthetic_params(G), syn.synthetic_instructions(B, G), syn.synthetic_initial_state(B, G),
            syn.synthetic_images(B, G))


def test_zero_kernel_gives_head_bias_for_any_context():
    """hypernetwork.py:75-77 + model.py:330-346: with zero head kernels the generated params are the bias."""
    P, ins, st, _ = _inputs()
    for lf in LEAVES:
        P[lf.head_name + "/kernel"] = np.zeros_like(P[lf.head_name + "/kernel"])
    bp, _ = onp.create_tasks(P, G, LEAVES, ins, st)
    for lf in LEAVES:
        want = np.broadcast_to(P[lf.head_name + "/bias"].reshape(lf.shape), bp[lf.flat_name].shape)
        np.testing.assert_array_equal(bp[lf.flat_name], want)


def test_padded_language_tokens_cannot_influence_context():
    """hypernetwork.py:151-157 (attend_to_padding=False)."""
    P, ins, st, _ = _inputs()
    li = ins["language_instruction"]
    c0 = onp.context_embedding(P, G, li["token_embedding"], li["attention_mask"], st["patch_embeddings"][:, 0])
    emb = li["token_embedding"].copy()
    emb[li["attention_mask"] == 0] += 100.0
    c1 = onp.context_embedding(P, G, emb, li["attention_mask"], st["patch_embeddings"][:, 0])
    np.testing.assert_allclose(c0, c1, atol=1e-12)
    emb2 = li["token_embedding"].copy()
    emb2[:, 0] += 1.0                                        # a real token does matter
    c2 = onp.context_embedding(P, G, emb2, li["attention_mask"], st["patch_embeddings"][:, 0])
    assert np.abs(c2 - c0).max() > 1e-6


def test_patch_tokens_do_not_see_the_action_token():
    """base_vit.py:209-214: patch rows are independent of pos_embedding[256] / the action token."""
    P, ins, st, im = _inputs(2)
    bp, _ = onp.create_tasks(P, G, LEAVES, ins, st)
    sink0, sink1 = {}, {}
    tok = onp.dinov2(P, G, ENC, onp.normalize_images(im[:, 0]))[:, 1:]
    onp.policy(bp, G, tok, sink0)
    bp2 = {k: v.copy() for k, v in bp.items()}
    bp2["encoder_pos_embedding"][:, :, -1] += 3.0
    onp.policy(bp2, G, tok, sink1)
    key = f"pol/Transformer_0/encoderblock_{G.layers - 1}/out"
    np.testing.assert_allclose(sink0[key][:-1], sink1[key][:-1], atol=1e-12)
    assert np.abs(sink0[key][-1] - sink1[key][-1]).max() > 1e-6


def test_action_ranges_and_shape():
    P, ins, st, im = _inputs()
    bp, _ = onp.create_tasks(P, G, LEAVES, ins, st)
    act, logit, _, _ = onp.sample_actions(P, G, ENC, bp, im)
    assert act.shape == (3, G.horizon, G.action_dim)
    assert np.abs(act[..., :6]).max() <= G.max_action                      # action_heads.py:469-470
    assert set(np.unique(act[..., 6])) <= {0.0, 1.0}                        # :536
    np.testing.assert_array_equal(act[..., 6], (logit >= 0).astype(float))


def test_batched_equals_per_episode():
    P, ins, st, im = _inputs()
    bp, _ = onp.create_tasks(P, G, LEAVES, ins, st)
    act, *_ = onp.sample_actions(P, G, ENC, bp, im)
    b = 1
    ins1 = {"language_instruction": {k: v[b:b + 1] for k, v in ins["language_instruction"].items()}}
    bp1, _ = onp.create_tasks(P, G, LEAVES, ins1, {"patch_embeddings": st["patch_embeddings"][b:b + 1]})
    act1, *_ = onp.sample_actions(P, G, ENC, bp1, im[b:b + 1])
    np.testing.assert_allclose(act1[0], act[b], atol=1e-12)


def test_ensemble_temp0_is_running_mean_of_aligned_predictions():
    """action_ensemble.py:18-26."""
    rng = np.random.default_rng(0)
    H = 4
    ens = onp.Ensembler(H, 0.0)
    hist = []
    for t in range(9):
        a = rng.normal(size=(5, H, 7))
        hist.append(a)
        got = ens(a)
        use = hist[-H:]
        want = np.mean([p[:, len(use) - 1 - i] for i, p in enumerate(use)], axis=0)
        np.testing.assert_allclose(got, want, atol=1e-12)


def test_unnormalise_respects_mask():
    """hypervla_interface.py:220-230."""
    stats = syn.synthetic_dataset_statistics(G)["libero"]["action"]
    a = np.random.default_rng(1).normal(size=(4, 7))
    out = onp.unnormalize(a, stats)
    np.testing.assert_allclose(out[:, :6], a[:, :6] * stats["std"][:6] + stats["mean"][:6])
    np.testing.assert_array_equal(out[:, 6], a[:, 6])


def test_mix_loss_numpy_vs_torch_and_gradient_oracle():
    """A13 forward: the two restatements of MixActionHead.loss agree; the autograd gradient oracle (round-2
    fine-tune step) is consistent with a central finite difference of the float64 loss."""
    import torch
    from oracle import hvla_ref_torch as ot
    P, ins, st, im = _inputs(3)
    batch = syn.synthetic_action_batch(3, G)
    bp, _ = onp.create_tasks(P, G, LEAVES, ins, st)
    tok = onp.dinov2(P, G, ENC, onp.normalize_images(im[:, 0]))[:, 1:]
    act, logit, _ = onp.policy(bp, G, tok)
    per, mean = onp.mix_loss(G, act[..., :6], logit, batch["action"], batch["timestep_pad_mask"], batch["action_pad_mask"])
    per_t, mean_t, grads = ot.train_loss_and_grads(P, G, LEAVES, ins, st, tok, batch)
    np.testing.assert_allclose(per, per_t.numpy(), atol=1e-10)
    assert abs(mean - float(mean_t)) < 1e-10 and np.isfinite(per).all() and (per > 0).all()
    key = LEAVES[5].head_name + "/bias"              # a generated LayerNorm/bias leaf
    idx = 3
    eps = 1e-5
    vals = []
    for sgn in (+1, -1):
        P2 = dict(P)
        v = P[key].astype(np.float64).copy()
        v[idx] += sgn * eps
        P2[key] = v
        bp2, _ = onp.create_tasks(P2, G, LEAVES, ins, st)
        a2, l2, _ = onp.policy(bp2, G, tok)
        vals.append(onp.mix_loss(G, a2[..., :6], l2, batch["action"], batch["timestep_pad_mask"], batch["action_pad_mask"])[1])
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - float(grads[key][idx])) <= 1e-6 * max(1.0, abs(fd))


def test_t5_restatement_matches_transformers_torch_t5():
    """The float64 T5 encoder restatement against transformers' own torch T5EncoderModel (same architecture as the
    FlaxT5EncoderModel the reference calls) on random weights, incl. padding and the relative-position buckets."""
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    from hypervla import synthetic as syn
    from hypervla.config import T5_TINY, T5Geometry
    from oracle import hvla_ref_np as onp
    for t in (T5_TINY, T5Geometry(vocab=500, d_model=96, d_kv=16, heads=6, d_ff=160, layers=3)):
        cfg = tr.T5Config(vocab_size=t.vocab, d_model=t.d_model, d_kv=t.d_kv, d_ff=t.d_ff, num_layers=t.layers,
                          num_heads=t.heads, relative_attention_num_buckets=t.buckets,
                          relative_attention_max_distance=t.max_distance, feed_forward_proj="relu",
                          layer_norm_epsilon=t.eps, dropout_rate=0.0)
        m = tr.T5EncoderModel(cfg).eval().double()
        tp = syn.synthetic_t5_params(t)
        sd = {"shared.weight": tp["shared/embedding"], "encoder.embed_tokens.weight": tp["shared/embedding"],
              "encoder.final_layer_norm.weight": tp["encoder/final_layer_norm/weight"],
              "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight":
                  tp["encoder/block/0/layer/0/SelfAttention/relative_attention_bias/embedding"]}
        for i in range(t.layers):
            b, T = f"encoder/block/{i}/layer/", f"encoder.block.{i}.layer."
            for nm in "qkvo":
                sd[T + f"0.SelfAttention.{nm}.weight"] = tp[b + f"0/SelfAttention/{nm}/kernel"].T
            sd[T + "0.layer_norm.weight"] = tp[b + "0/layer_norm/weight"]
            sd[T + "1.DenseReluDense.wi.weight"] = tp[b + "1/DenseReluDense/wi/kernel"].T
            sd[T + "1.DenseReluDense.wo.weight"] = tp[b + "1/DenseReluDense/wo/kernel"].T
            sd[T + "1.layer_norm.weight"] = tp[b + "1/layer_norm/weight"]
        m.load_state_dict({k: torch.as_tensor(np.ascontiguousarray(v)).double() for k, v in sd.items()}, strict=True)
        tok = syn.synthetic_token_ids(3, t)
        with torch.no_grad():
            ref = m(input_ids=torch.as_tensor(tok["input_ids"]),
                    attention_mask=torch.as_tensor(tok["attention_mask"])).last_hidden_state.numpy()
        got = onp.t5_encoder(tp, t, tok["input_ids"], tok["attention_mask"])
        assert np.abs(got - ref).max() <= 1e-5
        blk = m.encoder.block[0].layer[0].SelfAttention
        for T_ in (8, 32, 77, 200):
            rp = torch.arange(T_)[None, :] - torch.arange(T_)[:, None]
            np.testing.assert_array_equal(blk._relative_position_bucket(rp, True, t.buckets, t.max_distance).numpy(),
                                          onp.t5_relative_buckets(T_, t.buckets, t.max_distance))


def test_image_preprocessing_restatement_properties():
    """tf.image.resize(lanczos3, antialias) + crop_and_resize restatement (no TensorFlow here: parity unpinned): span
    weights are normalised, a same-size resize is the identity on uint8, constants stay constant through both stages,
    down-scaling by an integer factor of a smooth ramp keeps the ramp."""
    rng = np.random.default_rng(3)
    for n_in, n_out in ((640, 224), (480, 224), (224, 224), (128, 224)):
        spans = onp.resize_spans(n_in, n_out)
        assert len(spans) == n_out
        for s, w in spans:
            assert 0 <= s and s + len(w) <= n_in and abs(float(w.sum()) - 1.0) < 1e-5
    sq = rng.integers(0, 256, (40, 40, 3), dtype=np.uint8)
    np.testing.assert_array_equal(onp.preprocess_image(sq, 40), sq)
    const = np.full((48, 64, 3), 77, np.uint8)
    assert (onp.preprocess_image(const, 28) == 77).all() and (onp.preprocess_image(const, 28, crop=True) == 77).all()
    ramp = np.broadcast_to((np.arange(96, dtype=np.float32) * 2.0)[None, :, None], (96, 96, 3))
    small = onp.resize_lanczos3(ramp, 32)
    want = (np.arange(32) + 0.5) * 3.0 * 2.0 - 1.0                       # value of the ramp at each output pixel's centre
    np.testing.assert_allclose(small[16, 4:-4, 0], want[4:-4], atol=0.05)
    # the crop keeps the centre: pixel (S-1)/2 maps to (S-1)/2
    img = rng.integers(0, 256, (31, 31, 3)).astype(np.float32)
    c = onp.crop_and_resize_bilinear(img, 31)
    np.testing.assert_allclose(c[15, 15], img[15, 15], atol=1e-3)


def test_resize_with_pad_restatement_properties():
    """tf.image.resize_with_pad(image, 256, 320) restatement (hypervla_interface.py:90-95; parity unpinned): a 480 x 640
    frame is halved to 240 x 320 and sits between 8 zero rows; halving with half-pixel centres averages 2 x 2 blocks; a
    frame that already is 256 x 320 passes through; a tall frame is padded left and right instead."""
    rng = np.random.default_rng(5)
    f = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    p = onp.resize_with_pad_bilinear(f)
    assert p.shape == (256, 320, 3) and p.dtype == np.float32
    assert (p[:8] == 0).all() and (p[248:] == 0).all()
    blocks = f.astype(np.float32).reshape(240, 2, 320, 2, 3).mean(axis=(1, 3))
    np.testing.assert_allclose(p[8:248], blocks, atol=1e-4)
    same = rng.integers(0, 256, (256, 320, 3), dtype=np.uint8)
    np.testing.assert_array_equal(onp.resize_with_pad_bilinear(same), same.astype(np.float32))
    tall = np.full((512, 320, 3), 200, np.uint8)
    q = onp.resize_with_pad_bilinear(tall)
    assert (q[:, :80] == 0).all() and (q[:, 240:] == 0).all() and (q[:, 80:240] == 200).all()
    out = onp.preprocess_image(f, 224, padded_resize=True)
    assert out.shape == (224, 224, 3) and out.dtype == np.uint8 and out[:4].max() <= 2 and out[112].mean() > 60
