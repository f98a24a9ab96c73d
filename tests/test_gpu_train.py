"""GPU: fine-tune step (A13; frozen and trained image encoder) through the C ABI against the autograd gradient oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    from hypervla import synthetic as syn
    from hypervla.config import MID, encoder_leaves, generated_leaves
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner
    from oracle import hvla_ref_np as onp, hvla_ref_torch as ot
    g, B = MID, 4
    model = HyperVLA.from_synthetic(g, max_batch=B)
    P = model.params
    leaves = generated_leaves(g)
    ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
    batch = syn.synthetic_action_batch(B, g)
    tok = onp.dinov2(P, g, dict(encoder_leaves(g)), onp.normalize_images(im[:, 0]))[:, 1:]      # frozen tokens (oracle)
    per, loss, grads = ot.train_loss_and_grads(P, g, leaves, ins, st, tok, batch)
    ft = FineTuner(model, B, ema_start_step=0)          # EMA from the first update on (the reference's default is 5000)
    return dict(g=g, B=B, model=model, ft=ft, ins=ins, st=st, im=im, batch=batch, tok=tok, per=per.numpy(),
                loss=float(loss), grads={k: v.numpy() for k, v in grads.items()}, P=P)


def test_train_forward_loss(setup):
    s = setup
    loss = s["ft"].forward_backward(s["ins"], s["st"], s["tok"].astype(np.float32), s["batch"], forward_only=True).clone()
    np.testing.assert_allclose(loss.cpu().numpy(), s["per"], rtol=2e-4, atol=2e-5)
    # the forward pass has no atomic reduction: the loss is bit-reproducible (the split-K weight gradients are not)
    again = s["ft"].forward_backward(s["ins"], s["st"], s["tok"].astype(np.float32), s["batch"], forward_only=True)
    assert torch.equal(loss, again)


def test_train_gradients_match_autograd(setup):
    from hypervla.train import unpack_params
    s = setup
    ft = s["ft"]
    loss = ft.forward_backward(s["ins"], s["st"], s["tok"].astype(np.float32), s["batch"])
    np.testing.assert_allclose(loss.cpu().numpy(), s["per"], rtol=2e-4, atol=2e-5)
    got = unpack_params(s["g"], ft.grads.cpu().numpy())
    worst = []
    gmax = max(np.abs(v).max() for v in s["grads"].values())
    for k, ref in s["grads"].items():
        d = np.abs(got[k].reshape(ref.shape) - ref).max()
        # leaves whose true gradient is ~0 (key biases: softmax is invariant to them) are held to the global scale
        scale = max(np.abs(ref).max(), 1e-4 * gmax)
        worst.append((d / scale, k, d, scale))
    worst.sort(reverse=True)
    print("worst relative gradient errors:", [(f"{r:.2e}", k) for r, k, _, _ in worst[:5]])
    assert worst[0][0] <= 2e-3, worst[:5]


def test_unclipped_targets_when_the_checkpoint_says_so(setup):
    """action_head_kwargs.clip_target=False (what load_pretrained injects for checkpoints without the key,
    hypervla/model.py:157-163): the loss and its gradient use the raw target (action_heads.py:499-500).  The synthetic
    batch has targets beyond +-max_action, so the two settings differ."""
    import dataclasses
    from hypervla.config import default_config, generated_leaves
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner, unpack_params
    from oracle import hvla_ref_torch as ot
    s = setup
    g2 = dataclasses.replace(s["g"], clip_target=False)
    assert (np.abs(s["batch"]["action"][..., :6]) > g2.max_action).any()
    m2 = HyperVLA(default_config(g2), s["P"], None, None, max_batch=s["B"])
    assert m2.geometry.clip_target is False
    per, loss, grads = ot.train_loss_and_grads(s["P"], g2, generated_leaves(g2), s["ins"], s["st"], s["tok"], s["batch"])
    assert abs(float(loss) - s["loss"]) > 1e-3                      # not the clipped loss
    ft = FineTuner(m2, s["B"])
    got_loss = ft.forward_backward(s["ins"], s["st"], s["tok"].astype(np.float32), s["batch"])
    np.testing.assert_allclose(got_loss.cpu().numpy(), per.numpy(), rtol=2e-4, atol=2e-5)
    got = unpack_params(g2, ft.grads.cpu().numpy())
    gmax = max(float(v.abs().max()) for v in grads.values())
    for k in ("output_head_action_head_continuous_head_kernel/kernel", "Transformer_0/encoder_norm/scale"):
        ref = grads[k].numpy()
        d = np.abs(got[k].reshape(ref.shape) - ref).max()
        assert d <= 2e-3 * max(np.abs(ref).max(), 1e-4 * gmax), (k, d)
    # the stand-alone loss entry point follows the same switch
    act, lg = m2.policy_from_tokens(s["tok"].astype(np.float32), m2.create_tasks(instruction_dict=s["ins"], initial_state=s["st"])[0])
    l2, _ = m2.action_loss(act, lg, s["batch"])
    np.testing.assert_allclose(l2.cpu().numpy(), per.numpy(), rtol=2e-4, atol=2e-5)


def test_adamw_step_matches_reference_update(setup):
    """clip-by-global-norm -> AdamW (bf16 mu, v5 weight-decay mask) -> EMA against a numpy restatement of the
    optax chain (octo/utils/train_utils.py:411-426; scripts/train.py:618-625) for the first step."""
    from hypervla.config import generated_leaves
    from hypervla.train import pack_params, train_param_layout
    s = setup
    ft, g = s["ft"], s["g"]
    ft.forward_backward(s["ins"], s["st"], s["tok"].astype(np.float32), s["batch"])
    p0 = ft.params.cpu().numpy().astype(np.float64)
    gr = ft.grads.cpu().numpy().astype(np.float64)
    e0 = ft.ema.cpu().numpy().astype(np.float64)
    lr, wd, b1, b2, eps = 1e-3, 0.05, 0.9, 0.999, 1e-8
    ft.step_count = 0
    ft.mu.zero_(); ft.nu.zero_()
    ft.apply(lr=lr)
    norm = np.sqrt((gr * gr).sum())
    gc = gr * min(1.0, 1.0 / norm)
    mu = (1 - b1) * gc
    nu = (1 - b2) * gc * gc
    upd = (mu / (1 - b1)) / (np.sqrt(nu / (1 - b2)) + eps)
    layout, total = train_param_layout(g)
    mask = np.zeros(total, bool)
    G = ft.G
    cols = np.zeros(G, bool)
    for l in generated_leaves(g):
        if "kernel" in l.flat_name:
            cols[l.offset:l.offset + l.size] = True
    for name, off, shape in layout:
        if name == "W_cat":
            mask[off:off + int(np.prod(shape))] = np.tile(cols, shape[0])
        if name == "b_cat":
            mask[off:off + G] = cols
    upd = upd + wd * p0 * mask
    want = p0 - lr * upd
    got = ft.params.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    np.testing.assert_allclose(ft.ema.cpu().numpy(), 0.999 * e0 + 0.001 * want, atol=2e-6)
    assert abs(float(ft.mu.float().abs().sum()) - np.abs(mu).sum()) <= 1e-2 * np.abs(mu).sum()


def test_loss_decreases_over_steps(setup):
    s = setup
    from hypervla.train import FineTuner
    ft = FineTuner(s["model"], s["B"], peak_lr=1e-3)
    losses = []
    for i in range(8):
        losses.append(float(ft.step(s["ins"], s["st"], s["im"], s["batch"], lr=1e-3)))
    assert losses[-1] < losses[0], losses


@pytest.mark.timeout(900)
def test_full_geometry_gradients(golden_dir):
    """README geometry (vit_t policy on 256 DINOv2-base tokens, 201 500 generated parameters), B = 2."""
    from hypervla import synthetic as syn
    from hypervla.config import FULL, generated_leaves
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner, unpack_params
    from oracle import hvla_ref_torch as ot
    g, B = FULL, 2
    model = HyperVLA.from_synthetic(g, max_batch=B)
    ins, st = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g)
    batch = syn.synthetic_action_batch(B, g)
    tok = np.random.default_rng(9).standard_normal((B, g.patches, g.enc_dim)).astype(np.float32)   # frozen-encoder tokens
    params = {k: v for k, v in model.params.items() if not k.startswith("encoder_image_encoder_")}
    per, loss, grads = ot.train_loss_and_grads(params, g, generated_leaves(g), ins, st, tok, batch)
    ft = FineTuner(model, B)
    got_loss = ft.forward_backward(ins, st, tok, batch).cpu().numpy()
    np.testing.assert_allclose(got_loss, per.numpy(), rtol=3e-4, atol=3e-5)
    got = unpack_params(g, ft.grads.cpu().numpy())
    gmax = max(float(v.abs().max()) for v in grads.values())
    worst = max((np.abs(got[k].reshape(v.shape) - v.numpy()).max() / max(float(v.abs().max()), 1e-4 * gmax), k) for k, v in grads.items())
    print("full geometry worst relative gradient error", worst)
    assert worst[0] <= 3e-3, worst


# ------------------------------------------------------------------ trained image encoder (README.md:55)
def _encoder_case(g, B, tol, seed_rank=0):
    from hypervla import synthetic as syn
    from hypervla.config import encoder_leaves, generated_leaves
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner, unpack_params
    from oracle import hvla_ref_torch as ot
    model = HyperVLA.from_synthetic(g, max_batch=B)
    ins, st, im = syn.synthetic_instructions(B, g, seed_rank), syn.synthetic_initial_state(B, g, seed_rank), syn.synthetic_images(B, g, seed_rank)
    batch = syn.synthetic_action_batch(B, g, seed_rank)
    per, loss, grads = ot.train_loss_and_grads(model.params, g, generated_leaves(g), ins, st, None, batch, images=im,
                                               enc_shapes=dict(encoder_leaves(g)))
    ft = FineTuner(model, B, train_encoder=True)
    got_loss = ft.forward_backward(ins, st, im, batch).cpu().numpy()
    np.testing.assert_allclose(got_loss, per.numpy(), rtol=3e-4, atol=3e-5)
    got = unpack_params(g, ft.grads.cpu().numpy(), train_encoder=True)
    assert set(got) == set(grads), set(got) ^ set(grads)
    gmax = max(float(v.abs().max()) for v in grads.values())
    rel = sorted(((np.abs(got[k].reshape(v.shape) - v.numpy()).max() / max(float(v.abs().max()), 1e-4 * gmax), k)
                  for k, v in grads.items()), reverse=True)
    print("worst relative gradient errors (trained encoder):", [(f"{r:.2e}", k) for r, k in rel[:6]])
    enc_rel = [r for r in rel if r[1].startswith("encoder_image_encoder_")]
    assert len(enc_rel) == len(encoder_leaves(g))
    assert any(float(grads[k].abs().max()) > 0 for _, k in enc_rel)
    assert rel[0][0] <= tol, rel[:6]
    return model, ft, (ins, st, im, batch)


def test_encoder_gradients_match_autograd():
    """Every DINOv2 leaf (patch conv, cls / position embeddings, LayerScale, 12x attention + MLP, final LayerNorm) and
    every hypernetwork leaf against float64 autograd through transformers' Dinov2Model + the restated policy."""
    from hypervla.config import MID
    _encoder_case(MID, 3, 2e-3)


@pytest.mark.timeout(1200)
def test_encoder_gradients_full_geometry():
    from hypervla.config import FULL
    _encoder_case(FULL, 1, 3e-3)


@pytest.mark.parametrize("strategy", ["v5", "v1"])
def test_two_group_optimizer_and_delta_decay(strategy):
    """multi_transform{generated: AdamW(lr, mask), shared: AdamW(base_lr, mask)} under one global-norm clip, plus the pull
    towards the pretrained encoder (train_utils.py:330-426, scripts/train.py:465-471), for both weight-decay strategies:
    v5 (README run) decays the kernel-generating output heads and EVERY image-encoder leaf, v1 every leaf whose path
    contains "kernel" (which includes the bias of a head that generates a kernel).
    Under v5 with base_weight_decay > 0 a shared bias / norm / LayerScale leaf that sits at its pretrained value sees
    +wd p - wd p0 = 0, i.e. it does not drift."""
    from hypervla import synthetic as syn
    from hypervla.config import MID, generated_leaves
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner, train_param_layout
    g, B = MID, 2
    model = HyperVLA.from_synthetic(g, max_batch=B)
    ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
    batch = syn.synthetic_action_batch(B, g)
    wd, bwd, lr, blr, b1, b2, eps = 0.05, 0.01, 1e-3, 2e-4, 0.9, 0.999, 1e-8
    ft = FineTuner(model, B, weight_decay=wd, train_encoder=True, base_weight_decay=bwd, weight_decay_strategy=strategy)
    assert ft.base_peak_lr == 3e-5                                      # hypervla_pretrain_config.py:292-298
    layout, total = train_param_layout(g, True)
    nh, G = ft.n_hyper, ft.G
    moved = torch.zeros_like(ft.params[nh:])
    for name, off, shape in layout:                                       # move the encoder KERNELS away from the pretrained point
        if off >= nh and "kernel" in name:
            moved[off - nh:off - nh + int(np.prod(shape))] = 0.01
    ft.params[nh:] += moved * torch.randn_like(moved)
    ft.forward_backward(ins, st, im, batch)
    p0 = ft.params.cpu().numpy().astype(np.float64)
    pre = ft.params0.cpu().numpy().astype(np.float64)
    gr = ft.grads.cpu().numpy().astype(np.float64)
    ft.apply(lr=lr, base_lr=blr)
    gc = gr * min(1.0, 1.0 / np.sqrt((gr * gr).sum()))
    upd = ((1 - b1) * gc / (1 - b1)) / (np.sqrt((1 - b2) * gc * gc / (1 - b2)) + eps)
    cols = np.zeros(G, bool)
    for l in generated_leaves(g):
        if "kernel" in l.flat_name:
            cols[l.offset:l.offset + l.size] = True
    want = p0.copy()
    for name, off, shape in layout:
        n = int(np.prod(shape))
        sl = slice(off, off + n)
        if off < nh:
            if strategy == "v5":
                m = np.tile(cols, shape[0]) if name == "W_cat" else cols if name == "b_cat" else np.zeros(n, bool)
            else:
                # v1: "kernel" in keystr(path) (train_utils.py:378-382); the path of a head's bias is
                # ['output_head_<leaf>']['bias'], which contains "kernel" when the generated leaf is a kernel
                m = np.ones(n, bool) if (name == "W_cat" or "kernel" in name) else cols if name == "b_cat" else np.zeros(n, bool)
            want[sl] = p0[sl] - lr * (upd[sl] + wd * p0[sl] * m)
        else:
            k = 1.0 if (strategy == "v5" or "kernel" in name) else 0.0
            want[sl] = p0[sl] - blr * (upd[sl] + bwd * k * p0[sl] - bwd * pre[off - nh:off - nh + n])
            if strategy == "v5" and "kernel" not in name:                 # no drift of an un-moved leaf beyond Adam's own step
                np.testing.assert_allclose(want[sl], p0[sl] - blr * upd[sl], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ft.params.cpu().numpy(), want, rtol=0, atol=2e-6)


def test_gradient_accumulation_and_ema_start(setup):
    """optax.MultiSteps under chain(clip, MultiSteps(adamw)) (train_utils.py:420-426): with k = 2 the parameters move on every
    second micro-batch only, by AdamW on the mean of the two CLIPPED gradients; the EMA is a copy of the parameters at
    update `ema_start_step` and an average after it (scripts/train.py:681-690)."""
    from hypervla import synthetic as syn
    from hypervla.train import FineTuner
    s = setup
    g, B = s["g"], s["B"]
    ft = FineTuner(s["model"], B, grad_accumulation_steps=2, ema_start_step=2, weight_decay=0.0)
    tok = s["tok"].astype(np.float32)
    b2 = syn.synthetic_action_batch(B, g, rank=1)
    p0 = ft.params.clone()
    ft.forward_backward(s["ins"], s["st"], tok, s["batch"])
    g1 = ft.grads.double().clone()
    assert ft.apply(lr=1e-3) is False and torch.equal(ft.params, p0) and ft.step_count == 0
    ft.forward_backward(s["ins"], s["st"], tok, b2)
    g2 = ft.grads.double().clone()
    assert ft.apply(lr=1e-3) is True and ft.step_count == 1
    clipn = lambda v: v * min(1.0, 1.0 / float(v.norm()))
    mean = (clipn(g1) + clipn(g2)) / 2
    want = p0.double() - 1e-3 * mean / (mean.abs() + 1e-8)              # first AdamW step: m / (sqrt(v) + eps) = g / (|g| + eps)
    # (where the two clipped gradients cancel to below Adam's eps the update g / (|g| + eps) amplifies f32 rounding of g)
    ok = mean.abs() > 1e-6
    err = ((ft.params.double() - want).abs() * ok)
    assert float(err.max()) <= 2e-6 and float(ok.double().mean()) > 0.9, (float(err.max()), float(ok.double().mean()))
    assert float((ft.params.double() - p0.double()).abs().max()) <= 1e-3 * (1 + 1e-3)       # |update| <= lr everywhere
    assert torch.equal(ft.ema, p0)                                       # update 1 < ema_start_step: untouched
    for _ in range(2):
        ft.forward_backward(s["ins"], s["st"], tok, s["batch"]); ft.apply(lr=1e-3)
    assert ft.step_count == 2 and torch.equal(ft.ema, ft.params)         # update 2 == start: a copy
    p2 = ft.params.clone()
    for _ in range(2):
        ft.forward_backward(s["ins"], s["st"], tok, s["batch"]); ft.apply(lr=1e-3)
    np.testing.assert_allclose(ft.ema.cpu().numpy(), (0.999 * p2.double() + 0.001 * ft.params.double()).cpu().numpy(), atol=2e-6)


@pytest.mark.timeout(900)
def test_config5_per_gpu_step_at_full_geometry():
    """BASELINE configs[4] per-GPU share: README geometry, 32 samples, image encoder trained (113 M parameters).  No oracle
    at this size; properties: the loss of the fused step equals the forward-only loss of the inference-precision model to
    f32 accuracy class, the flat gradient is finite and equals the mean of the two half-batch gradients (what `pmean` over
    two ranks would give, scripts/train.py:453-460), and one optimizer step moves both parameter groups."""
    from hypervla import synthetic as syn
    from hypervla.config import FULL
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner
    g, B = FULL, 32
    model = HyperVLA.from_synthetic(g, max_batch=B)
    ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
    batch = syn.synthetic_action_batch(B, g)
    ft = FineTuner(model, B, train_encoder=True)
    assert ft.n - ft.n_hyper > 85_000_000                                   # the DINOv2-base leaves (16 x 16 position table) are in the vector
    loss = ft.forward_backward(ins, st, im, batch).clone()
    assert torch.isfinite(loss).all() and torch.isfinite(ft.grads).all()
    gfull = ft.grads.double().clone()
    # the same step's forward against the inference path (fp16 encoder): per-sample losses agree to the encoder's precision
    w, tasks, _ = model.create_tasks(instruction_dict=ins, initial_state=st)
    act, inter = model.sample_actions(torch.as_tensor(im[:, 0]).to(model.device), ins, tasks, None, w)
    l_inf, _ = model.action_loss(act, inter["gripper_logits"], batch)
    np.testing.assert_allclose(loss.cpu().numpy(), l_inf.cpu().numpy(), rtol=5e-3, atol=5e-3)
    # gradient of the batch = mean of the gradients of its two halves
    half = FineTuner(model, B // 2, train_encoder=True)
    acc = torch.zeros_like(gfull)
    for lo in (0, B // 2):
        sl = slice(lo, lo + B // 2)
        sub_ins = {"language_instruction": {k: np.asarray(v)[sl] for k, v in ins["language_instruction"].items()}}
        sub_st = {"patch_embeddings": st["patch_embeddings"][sl]}
        sub_b = {k: v[sl] for k, v in batch.items()}
        half.forward_backward(sub_ins, sub_st, im[sl], sub_b)
        acc += half.grads.double() / 2
    scale = float(gfull.abs().max())
    assert float((acc - gfull).abs().max()) <= 2e-3 * scale, (float((acc - gfull).abs().max()), scale)
    before = ft.params.clone()
    ft.apply(lr=1e-4, base_lr=1e-5)                  # (the schedules themselves start their warm-up at 0)
    moved = (ft.params - before).abs()
    assert float(moved[:ft.n_hyper].max()) > 0 and float(moved[ft.n_hyper:].max()) > 0


def test_training_a_baked_position_table_must_be_asked_for():
    """The converter bakes HF's 37 x 37 DINOv2 position table to the run-time grid; the reference trains the original table
    through interpolate_pos_encoding, so FineTuner(train_encoder=True) refuses such a checkpoint unless told to go ahead."""
    from hypervla import synthetic as syn
    from hypervla.config import MID, default_config
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner
    cfg = default_config(MID)
    cfg["position_embeddings_baked_from"] = [37, 37]
    m = HyperVLA(cfg, syn.synthetic_params(MID), None, None, max_batch=2)
    FineTuner(m, 2)                                             # frozen encoder: fine
    with pytest.raises(ValueError, match="baked"):
        FineTuner(m, 2, train_encoder=True)
    FineTuner(m, 2, train_encoder=True, accept_baked_position_table=True)


def test_encoder_training_reduces_loss():
    from hypervla.config import MID
    from hypervla import synthetic as syn
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner
    g, B = MID, 4
    model = HyperVLA.from_synthetic(g, max_batch=B)
    ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
    batch = syn.synthetic_action_batch(B, g)
    ft = FineTuner(model, B, train_encoder=True)
    before = ft.params[ft.n_hyper:].clone()
    losses = [float(ft.step(ins, st, im, batch, lr=1e-3, base_lr=1e-4)) for _ in range(8)]
    assert losses[-1] < losses[0], losses
    assert float((ft.params[ft.n_hyper:] - before).abs().max()) > 0      # the encoder really moved


@pytest.mark.timeout(900)
def test_full_geometry_gradient_is_the_mean_of_per_sample_gradients():
    """Size-independent property at the README geometry with the encoder trained (no oracle needed): the loss is a mean
    over independent episodes (scripts/train.py:453-457), so grads(batch of 6) == mean_b grads(episode b alone)."""
    from hypervla import synthetic as syn
    from hypervla.config import FULL
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner
    g, B = FULL, 6
    model = HyperVLA.from_synthetic(g, max_batch=B)
    ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
    batch = syn.synthetic_action_batch(B, g)
    batch["timestep_pad_mask"][:] = True                       # keep every sample in the mean
    ft = FineTuner(model, B, train_encoder=True)
    loss = ft.forward_backward(ins, st, im, batch).clone()
    big = ft.grads.clone()
    one = FineTuner(model, 1, train_encoder=True)
    acc = torch.zeros_like(big)
    li = ins["language_instruction"]
    for b in range(B):
        sl = slice(b, b + 1)
        l1 = one.forward_backward({"language_instruction": {k: np.asarray(v)[sl] for k, v in li.items()}},
                                  {"patch_embeddings": st["patch_embeddings"][sl]}, im[sl],
                                  {k: v[sl] for k, v in batch.items()})
        assert abs(float(l1[0]) - float(loss[b])) <= 2e-5 * max(1.0, abs(float(loss[b])))
        acc += one.grads
    acc /= B
    scale = float(big.abs().max())
    assert float((acc - big).abs().max()) <= 2e-4 * scale, float((acc - big).abs().max()) / scale


def _dp_worker(rank, world, port, out_path):
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)       # one GPU on the test box: gloo moves the CUDA tensor
    from hypervla import synthetic as syn
    from hypervla.config import MID
    from hypervla.dp import episode_range
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner
    g, total = MID, 4
    lo, hi = episode_range(total, world, rank)
    sl = slice(lo, hi)
    model = HyperVLA.from_synthetic(g, max_batch=hi - lo)
    ins, st, im = syn.synthetic_instructions(total, g), syn.synthetic_initial_state(total, g), syn.synthetic_images(total, g)
    batch = syn.synthetic_action_batch(total, g)
    batch["timestep_pad_mask"][:] = True
    ft = FineTuner(model, hi - lo, train_encoder=True)
    li = ins["language_instruction"]
    ft.step({"language_instruction": {k: np.asarray(v)[sl] for k, v in li.items()}}, {"patch_embeddings": st["patch_embeddings"][sl]},
            im[sl], {k: v[sl] for k, v in batch.items()}, lr=1e-3, base_lr=1e-4)
    if rank == 0:
        np.save(out_path, np.stack([ft.params.cpu().numpy(), ft.grads.cpu().numpy()]))     # grads: after the all-reduce
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_data_parallel_step_equals_the_big_batch_step(tmp_path):
    """Config 5's data parallelism: two ranks with half the samples each, gradient all-reduce (mean) between
    hvla_train_step and hvla_train_apply (`pmean`, scripts/train.py:460) == one process with all the samples."""
    import torch.multiprocessing as mp
    from hypervla import synthetic as syn
    from hypervla.config import MID
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner
    import socket
    out = str(tmp_path / "rank0.npy")
    with socket.socket() as sk:                              # a free port, not a fixed one another run may still hold
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    g, total = MID, 4
    model = HyperVLA.from_synthetic(g, max_batch=total)
    ins, st, im = syn.synthetic_instructions(total, g), syn.synthetic_initial_state(total, g), syn.synthetic_images(total, g)
    batch = syn.synthetic_action_batch(total, g)
    batch["timestep_pad_mask"][:] = True
    ft = FineTuner(model, total, train_encoder=True)
    before = ft.params.cpu().numpy().copy()
    ft.step(ins, st, im, batch, lr=1e-3, base_lr=1e-4)
    single, gsingle = ft.params.cpu().numpy(), ft.grads.cpu().numpy()
    dp, gdp = np.load(out)
    gmax = np.abs(gsingle).max()
    assert np.abs(gdp - gsingle).max() <= 2e-4 * gmax, np.abs(gdp - gsingle).max() / gmax
    # Adam's first step is sign-like (lr * g / (|g| + eps)): compare the parameters where the gradient is not ~0
    big = np.abs(gsingle) > 1e-4 * gmax
    moved = np.abs(single - before)[big].max()
    assert moved > 0 and np.abs(dp - single)[big].max() <= 1e-2 * moved, (np.abs(dp - single)[big].max(), moved)


def test_bucketed_all_reduce_path_on_one_rank(tmp_path):
    """FineTuner.all_reduce_gradient cuts `pmean(grads)` (scripts/train.py:460) into the three buckets hvla_train_step
    finishes in turn and enqueues each all-reduce on a communication stream behind the bucket's event.  One GPU here, so a
    one-rank RCCL group: the reduction is the identity, which is the check -- every bucket goes through event, stream and
    collective and comes back unchanged, and the optimizer step behind it equals the step without the group."""
    import torch
    import torch.distributed as dist
    from hypervla import synthetic as syn
    from hypervla.config import MID
    from hypervla.model import HyperVLA
    from hypervla.train import FineTuner
    g, B = MID, 4
    model = HyperVLA.from_synthetic(g, max_batch=B)
    ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
    batch = syn.synthetic_action_batch(B, g)
    ref = FineTuner(model, B, train_encoder=True)
    ref.forward_backward(ins, st, im, batch)
    g_ref = ref.grads.clone()
    ref.apply(lr=1e-3, base_lr=1e-4)
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", world_size=1, rank=0)
    try:
        ft = FineTuner(model, B, train_encoder=True)
        assert [b[0] for b in ft.buckets] == ["image_encoder", "output_heads", "context_encoder"] and ft._bucket_id == [0, 1, 2]
        ft.forward_backward(ins, st, im, batch)
        ft.all_reduce_gradient(single_rank_too=True)                 # enqueued behind the step, nothing synchronised in between
        torch.cuda.synchronize()
        scale = float(g_ref.abs().max())
        assert float((ft.grads - g_ref).abs().max()) <= 1e-4 * scale   # (two runs of the step differ in the last bits: split-K sums)
        ft.forward_backward(ins, st, im, batch)
        torch.cuda.synchronize()
        g0 = ft.grads.clone()
        ft.all_reduce_gradient(single_rank_too=True)
        torch.cuda.synchronize()
        assert torch.equal(ft.grads, g0)                             # one rank: the reduction is the identity, bit for bit
        with pytest.raises(RuntimeError, match="bucket"):           # a frozen-encoder step produces no encoder bucket
            fz = FineTuner(model, B)
            fz.forward_backward(ins, st, model.encode_images(im), batch)
            model._ctx.train_wait_bucket(0, 0)
        ft.forward_backward(ins, st, im, batch)                       # events are re-recorded by every step
        ft.apply(lr=1e-3, base_lr=1e-4)
        # the same update; Adam's first step is lr * sign(g), so the few gradients whose last bits straddle 0 may differ
        assert float(((ft.params - ref.params).abs() > 1e-5).float().mean()) < 1e-2
    finally:
        dist.destroy_process_group()
