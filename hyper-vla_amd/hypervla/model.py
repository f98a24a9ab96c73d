"""``hypervla.model.HyperVLA`` — the reference's model API (hypervla/model.py:24-224) hosted on
libhvla.  Same surface, MI355X-native inside:

    model = HyperVLA.load_pretrained(path, step=None)      # model.py:139-224
    model = model.replace(params=ema_params)                # flax struct.dataclass .replace
    base_params, tasks, _ = model.create_tasks(instruction_dict=..., initial_state=...)   # :35-83
    actions, _ = model.sample_actions(images, instruction_dict, tasks, pad_mask, base_params, rng=key)  # :85-137

Differences a caller can observe, all by design:
  * ``create_tasks`` accepts B >= 1 episodes (the reference squeezes B == 1, model.py:81) and returns
    an opaque :class:`GeneratedWeights` handle that owns the device arena; ``.to_pytree()`` gives the
    reference's ``base_params`` pytree back.
  * of the per-step intermediates (every attention map, base_vit.py:117-118) the two slices a caller of the path reads
    (data/utils/hypervla_interface.py:208-217) are materialised on request: ``sample_actions(..., attention_maps=True)``
    adds ``"dino_cls_attention"`` [B, 12, 12, P] and ``"head_attention"`` [B, 4, 4, P] to its second value, which
    otherwise is ``{"gripper_logits": ...}``.
  * arrays may be numpy (copied to the device) or torch CUDA tensors (used in place); results come
    back in the same kind.

There is no CPU / eager fallback: constructing a model without a gfx950 device and a built
``libhvla.so`` raises.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Optional

import numpy as np

from . import _native
from .config import (FULL, Geometry, default_config, generated_leaves, geometry_from_config,
                     hypernet_param_shapes)


def _torch():
    import torch
    return torch


class GeneratedWeights:
    """Handle to one batch of generated policy weights on the device (what the reference calls
    ``base_params``).  Immutable from the caller's side; freed on garbage collection."""

    def __init__(self, model: "HyperVLA", handle, batch: int):
        self._model, self._h, self.batch = model, handle, batch

    def __del__(self):
        try:
            if self._h is not None:
                self._model._ctx.weights_free(self._h)
                self._h = None
        except Exception:
            pass

    def export(self):
        """(theta [B, G] in reference leaf order, context embedding [B, C]) as torch CUDA tensors."""
        torch = _torch()
        m = self._model
        dev = m.device
        theta = torch.empty(self.batch, m._ctx.num_generated, dtype=torch.float32, device=dev)
        ctx = torch.empty(self.batch, m.geometry.ctx_dim, dtype=torch.float32, device=dev)
        m._ctx.weights_export(self._h, theta.data_ptr(), ctx.data_ptr(), m._stream())
        return theta, ctx

    def to_pytree(self, squeeze: bool = False) -> Dict[str, Any]:
        """The reference's nested ``base_params`` dict (generated leaves only; the shared DINOv2 leaves
        live once on the device and are not broadcast, hypernetwork.py:233)."""
        theta = self.export()[0].cpu().numpy()
        tree: Dict[str, Any] = {}
        for lf in generated_leaves(self._model.geometry):
            v = theta[:, lf.offset:lf.offset + lf.size].reshape((self.batch,) + lf.shape)
            if squeeze:
                v = v[0]
            node = tree
            for key in lf.path[:-1]:
                node = node.setdefault(key, {})
            node[lf.path[-1]] = v
        return tree


class HyperVLA:
    def __init__(self, config: Dict, params: Dict[str, np.ndarray], example_batch: Optional[Dict] = None,
                 dataset_statistics: Optional[Dict] = None, device: int = 0, max_batch: int = 256,
                 enc_dtype: str = "f16", streams: int = 1, _shared_ctx=None):
        torch = _torch()
        self.config = config
        self.params = params
        self.example_batch = example_batch
        self.dataset_statistics = dataset_statistics
        self.geometry: Geometry = geometry_from_config(config)
        self.base_net_metadata = {"leaves": generated_leaves(self.geometry)}
        if not torch.cuda.is_available():
            raise RuntimeError("HyperVLA needs an MI355X (gfx950) device: torch.cuda.is_available() is False "
                               "and there is no CPU fallback")
        self.device = torch.device("cuda", device)
        self.max_batch, self.enc_dtype, self.streams = max_batch, enc_dtype, streams
        self._ctx = _native.Context(self.geometry, device, max_batch, enc_dtype, streams)
        want = hypernet_param_shapes(self.geometry)
        missing = [k for k in want if k not in params]
        if missing:
            raise ValueError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")
        self._ctx.load_weights({k: params[k] for k in want})

    def release_pooled_arenas(self) -> None:
        """Free the generated-weight arenas that freed `GeneratedWeights` parked for re-use (hvla_release_pooled_arenas:
        up to four per context, about 0.8 MB per episode each).  `create_tasks` allocates again when it next needs one."""
        self._ctx.release_pooled_arenas()

    # ------------------------------------------------------------------ construction
    @classmethod
    def load_pretrained(cls, checkpoint_path: str, step: Optional[int] = None, audit="raise", **kw) -> "HyperVLA":
        """Reads ``config.json``, ``example_batch.msgpack`` (flax msgpack, read without flax: hypervla/convert.py),
        ``dataset_statistics.json`` and the parameter file written by :meth:`save_pretrained` (``params_<step>.npz`` /
        ``params.npz``; flat '/'-joined flax names) -- the files of hypervla/model.py:152-214 except that
        the parameters are an npz instead of an Orbax step directory (exporter-only: INTEGRATION.md).

        `audit`: what the operand-range audit of the checkpoint's example batch does when activations leave the 16-bit operand
        type's range -- "raise" (default: refuse the checkpoint), "warn" (load it and warn; `model.operand_range` has the
        numbers) or False / "off" (no audit: no encoder pass at load time).  When the checkpoint has no example batch the audit
        cannot run; that is reported with a warning instead of passing silently."""
        if audit not in ("raise", "warn", "off", False, None):
            raise ValueError(f"audit must be 'raise', 'warn' or 'off', got {audit!r}")
        with open(os.path.join(checkpoint_path, "config.json")) as f:
            config = json.load(f)
        if "action_head_kwargs" not in config["base_net_kwargs"]:        # model.py:157-163
            config["base_net_kwargs"]["action_head_kwargs"] = dict(
                token_per_horizon=False, squash_continuous_action=True, clip_target=False, max_action=5.0)
        stats = None
        sp = os.path.join(checkpoint_path, "dataset_statistics.json")
        if os.path.exists(sp):
            with open(sp) as f:
                stats = _tree_map(np.array, json.load(f))
        cands = sorted(p for p in os.listdir(checkpoint_path) if p.startswith("params") and p.endswith(".npz"))
        if step is not None:
            cands = [p for p in cands if p == f"params_{step}.npz"]
        if not cands:
            raise FileNotFoundError(f"no params*.npz under {checkpoint_path} (step={step})")
        with np.load(os.path.join(checkpoint_path, cands[-1])) as z:
            params = {k: z[k] for k in z.files}
        from .convert import load_example_batch
        model = cls(config, params, load_example_batch(checkpoint_path), stats, **kw)
        if audit in ("raise", "warn"):         # a real checkpoint's activations must fit the 16-bit operand type
            import warnings
            try:
                if model.audit_operand_range() is None:
                    warnings.warn("operand-range audit skipped: the checkpoint has no example batch with image_primary frames; "
                                  "call model.audit_operand_range(images) on real observations", RuntimeWarning)
            except ValueError as e:             # operands out of range, or example frames the resize refuses: under 'warn' neither blocks the
                                                # load; a native / HIP failure of the audit itself is never downgraded
                if audit == "raise":
                    raise
                warnings.warn(f"{e} (audit='warn': loaded anyway)", RuntimeWarning)
        return model

    FP16_OPERAND_LIMIT = 32768.0               # half of fp16's largest finite value (65504)

    def audit_operand_range(self, images=None):
        """Run the image encoder once with a range audit of every 16-bit MFMA operand it writes (LayerNorm outputs, q / k /
        v, attention outputs, GELU outputs, all layers: `hvla_encode_audit`) and refuse a checkpoint whose activations leave
        the operand type's range: a trained DINOv2 carries outlier channels that the synthetic weights do not (DESIGN.md
        section 2).  `images` defaults to the checkpoint's example batch; returns {site: largest |operand|}, or None when
        there is nothing to audit on."""
        torch = _torch()
        if images is None:
            try:
                images = np.asarray(self.example_batch["observation"]["image_primary"])
            except (TypeError, KeyError):
                return None
        images = images[: self.max_batch]       # (sliced first, as it is -- ndarray or tensor, on any device: an example batch of many large frames is neither moved nor resized in full)
        img = self._dev(images, torch.uint8)
        if img.dim() == 5:
            img = img[:, 0].contiguous()
        g = self.geometry
        if tuple(img.shape[1:]) != (g.image_size, g.image_size, 3):
            try:
                img = self.preprocess_images(img)  # example frames of another size: the evaluators' resize (InferenceWrapper._resize_image)
            except _native.NativeError as e:       # frames the resize refuses are a property of the example batch (ValueError), not a device failure
                raise ValueError(f"the example frames {tuple(img.shape)} cannot be resized for the audit: {e}") from e
        img = img.contiguous()
        audit = self._ctx.encode_audit(img.data_ptr(), img.shape[0], self._stream())      # {site: (max |x|, non-finite count)}
        sites = {k: v[0] for k, v in audit.items()}
        bad = {k: v[1] for k, v in audit.items() if v[1]}
        limit = self.FP16_OPERAND_LIMIT if self.enc_dtype == "f16" else 3.0e38
        if bad or max(sites.values()) > limit:
            raise ValueError(f"encoder activations leave the {self.enc_dtype} operand range on the example batch: largest "
                             f"|operand| per site {sites}, non-finite values {bad}; load with enc_dtype='bf16' "
                             "(8 exponent bits) instead")
        self.operand_range = sites
        return sites

    def save_pretrained(self, step: int, checkpoint_path: str):
        os.makedirs(checkpoint_path, exist_ok=True)
        np.savez(os.path.join(checkpoint_path, f"params_{step}.npz"), **self.params)
        cp = os.path.join(checkpoint_path, "config.json")
        if not os.path.exists(cp):
            with open(cp, "w") as f:
                json.dump(self.config, f)
        sp = os.path.join(checkpoint_path, "dataset_statistics.json")
        if self.dataset_statistics is not None and not os.path.exists(sp):
            with open(sp, "w") as f:
                json.dump(_tree_map(lambda x: np.asarray(x).tolist(), self.dataset_statistics), f)
        ep = os.path.join(checkpoint_path, "example_batch.msgpack")          # hypervla/model.py:270-274
        if self.example_batch is not None and not os.path.exists(ep):
            from .convert import msgpack_serialize
            with open(ep, "wb") as f:
                f.write(msgpack_serialize(self.example_batch))

    @classmethod
    def from_synthetic(cls, geometry: Geometry = FULL, params: Optional[Dict[str, np.ndarray]] = None, **kw) -> "HyperVLA":
        from . import synthetic
        return cls(default_config(geometry), synthetic.synthetic_params(geometry) if params is None else params, None,
                   synthetic.synthetic_dataset_statistics(geometry), **kw)

    def replace(self, **changes) -> "HyperVLA":
        """flax ``struct.dataclass.replace``: the evaluators swap in EMA params this way
        (data/simpler/evaluate.py:440-444)."""
        if set(changes) - {"params", "config", "dataset_statistics", "example_batch"}:
            raise TypeError(f"cannot replace {set(changes)}")
        if "params" in changes or "config" in changes:
            return HyperVLA(changes.get("config", self.config), changes.get("params", self.params),
                            changes.get("example_batch", self.example_batch),
                            changes.get("dataset_statistics", self.dataset_statistics),
                            self.device.index or 0, self.max_batch, self.enc_dtype, self.streams)
        import copy
        other = copy.copy(self)
        for k, v in changes.items():
            setattr(other, k, v)
        return other

    # ------------------------------------------------------------------ helpers
    def _stream(self) -> int:
        return int(_torch().cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, dtype):
        torch = _torch()
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(np.asarray(a))).to(device=self.device, dtype=dtype).contiguous()

    # ------------------------------------------------------------------ observation preprocessing
    def preprocess_images(self, frames, crop: bool = False, padded_resize: bool = False):
        """`InferenceWrapper._resize_image` (data/utils/hypervla_interface.py:89-121) on the device: uint8 camera frames
        [B, H, W, 3] (or [H, W, 3]) -> uint8 [B, image_size, image_size, 3] CUDA tensor.  `padded_resize` first fits the
        frame into 256 x 320 with zero padding (`tf.image.resize_with_pad`, the "rtx" augmentation style)."""
        torch = _torch()
        f = self._dev(frames, torch.uint8)
        if f.dim() == 3:
            f = f[None].contiguous()
        if f.dim() != 4 or f.shape[-1] != 3:
            raise ValueError(f"frames must be [B, H, W, 3] uint8, got {tuple(f.shape)}")
        B, H, W, _ = f.shape
        S = self.geometry.image_size
        out = torch.empty(B, S, S, 3, dtype=torch.uint8, device=self.device)
        self._ctx.preprocess(f.data_ptr(), B, H, W, crop, out.data_ptr(), self._stream(), padded_resize)
        return out

    # ------------------------------------------------------------------ frozen instruction encoder (optional)
    def load_language_encoder(self, t5_params: Dict[str, np.ndarray], t5_geometry=None, max_batch: Optional[int] = None):
        """Put the frozen T5 encoder of `LanguageTokenizer('t5-base')` on the device
        (data/utils/language_tokenizer.py:9-28).  `t5_params`: FlaxT5EncoderModel tree flattened with '/'."""
        from .config import T5_BASE
        t = T5_BASE if t5_geometry is None else t5_geometry
        if t.d_model != self.geometry.lang_dim:
            raise ValueError(f"T5 d_model {t.d_model} != lang_dim {self.geometry.lang_dim}")
        self._ctx.t5_load(t, t5_params, self.geometry.lang_tokens, max_batch or self.max_batch)
        self.language_encoder = t
        return self

    def encode_instructions(self, tokens: Dict) -> Dict:
        """`token_to_embedding` (data/utils/language_tokenizer.py:24-28) on the device: {"input_ids", "attention_mask"}
        i64 [B, T] -> the same dict plus "token_embedding" f32 [B, T, lang_dim] (a CUDA tensor)."""
        torch = _torch()
        if getattr(self, "language_encoder", None) is None:
            raise RuntimeError("no language encoder loaded: call load_language_encoder(t5_params) first")
        ids = self._dev(tokens["input_ids"], torch.int64)
        mask = self._dev(tokens["attention_mask"], torch.int64)
        if ids.dim() != 2 or ids.shape != mask.shape:
            raise ValueError(f"input_ids {tuple(ids.shape)} / attention_mask {tuple(mask.shape)} must be [B, T]")
        B, T = ids.shape
        emb = torch.empty(B, T, self.geometry.lang_dim, dtype=torch.float32, device=self.device)
        self._ctx.t5_encode(ids.data_ptr(), mask.data_ptr(), emb.data_ptr(), B, T, self._stream())
        out = dict(tokens)
        out["token_embedding"] = emb
        return out

    # ------------------------------------------------------------------ the hot path
    def create_tasks(self, goals=None, instruction_dict: Dict = None, initial_state: Dict = None):
        """hypervla/model.py:35-83.  Returns (base_params handle, tasks dict, intermediates)."""
        torch = _torch()
        if instruction_dict is None or initial_state is None:
            raise ValueError("the built path is language + initial-image conditioned: instruction_dict and "
                             "initial_state are required (use_initial_image=True, README.md:42)")
        li = instruction_dict["language_instruction"]
        if "token_embedding" not in li and getattr(self, "language_encoder", None) is not None:
            li = self.encode_instructions(li)                      # scripts/train.py:407-415 does this inside the step
        g = self.geometry
        B = int(np.shape(li["input_ids"])[0])
        tok = self._dev(li["token_embedding"], torch.float32)
        mask = self._dev(li["attention_mask"], torch.int64)
        pe = initial_state["patch_embeddings"]
        cls = self._dev(pe[:, 0] if not isinstance(pe, torch.Tensor) else pe[:, 0], torch.float32)
        if tuple(tok.shape) != (B, g.lang_tokens, g.lang_dim) or tuple(mask.shape) != (B, g.lang_tokens) \
                or tuple(cls.shape) != (B, g.enc_dim):
            raise ValueError(f"bad shapes: token_embedding {tuple(tok.shape)}, attention_mask {tuple(mask.shape)}, "
                             f"patch_embeddings[:,0] {tuple(cls.shape)} for B={B}")
        h = self._ctx.generate(tok.data_ptr(), mask.data_ptr(), cls.data_ptr(), B, self._stream())
        torch.cuda.current_stream(self.device).synchronize()      # inputs may be temporaries
        tasks = {"pad_mask_dict": {"language_instruction": np.ones(B, dtype=bool)},     # model.py:51-70
                 "language_instruction": li}
        return GeneratedWeights(self, h, B), tasks, {}

    def sample_actions(self, images, instruction_dict=None, task=None, timestep_pad_mask=None,
                       base_params: GeneratedWeights = None, train: bool = False, rng=None,
                       image_embeddings=None, attention_maps: bool = False):
        """hypervla/model.py:85-137.  images uint8 [B, 1, H, W, 3] (or [B, H, W, 3]) -> actions
        [B, horizon, action_dim].  `attention_maps=True` also returns the two attention slices the reference's wrapper keeps
        from the intermediates (hypervla_interface.py:208-217): DINOv2's CLS-query attention over the patches, every layer and
        head, and the generated policy's action-token attention over the patches."""
        torch = _torch()
        if train:
            # train=True switches on nn.Dropout(dropout_rate) (base_vit.py:205, transformer.py:67,74,192,244) and the
            # image_embedding_noise (base_vit.py:123-127).  Every shipped config sets all of them to 0
            # (scripts/configs/hypervla_pretrain_config.py:333-339,376-380), where train=True is the same function as
            # train=False; a checkpoint that really asks for noise is refused rather than silently run without it.
            v = self.config["base_net_kwargs"]["vit_kwargs"]
            rates = {k: float(v.get(k, 0.0) or 0.0) for k in ("dropout_rate", "attention_dropout_rate", "image_embedding_noise")}
            rates["embedding_dropout_rate"] = float(self.config["base_net_kwargs"].get("embedding_dropout_rate", 0.0) or 0.0)
            if any(r > 0 for r in rates.values()):
                raise NotImplementedError(f"train=True with {rates}: dropout / embedding noise are not built (the README run uses 0)")
        if not isinstance(base_params, GeneratedWeights):
            raise TypeError("base_params must be the handle returned by create_tasks")
        g = self.geometry
        as_torch = isinstance(images, torch.Tensor)
        img = self._dev(images, torch.uint8)
        if img.dim() == 5:
            if img.shape[1] != 1:
                raise ValueError("window size must be 1 (images.squeeze(1), model.py:117)")
            img = img[:, 0].contiguous()
        B = base_params.batch
        if tuple(img.shape) != (B, g.image_size, g.image_size, 3):    # base_vit.py:86-89
            raise ValueError(f"Input image size must be {g.image_size}x{g.image_size}: got {tuple(img.shape)} for B={B}")
        actions = torch.empty(B, g.horizon, g.action_dim, dtype=torch.float32, device=self.device)
        logits = torch.empty(B, g.horizon, dtype=torch.float32, device=self.device)
        inter = {"gripper_logits": logits}
        if attention_maps:
            inter["dino_cls_attention"] = torch.empty(B, g.enc_layers, g.enc_heads, g.patches, dtype=torch.float32, device=self.device)
            inter["head_attention"] = torch.empty(B, g.layers, g.heads, g.patches, dtype=torch.float32, device=self.device)
            self._ctx.set_attention_outputs(inter["dino_cls_attention"].data_ptr(), inter["head_attention"].data_ptr())
        try:
            self._ctx.step(base_params._h, img.data_ptr(), actions.data_ptr(), logits.data_ptr(), B, self._stream())
        finally:
            if attention_maps:
                self._ctx.set_attention_outputs(0, 0)
        if as_torch:
            return actions, inter
        torch.cuda.current_stream(self.device).synchronize()
        return actions.cpu().numpy(), {k: v.cpu().numpy() for k, v in inter.items()}

    # stage-level entry points (policy-only variant of BASELINE config 2, parity tests)
    def encode_images(self, images):
        torch = _torch()
        g = self.geometry
        img = self._dev(images, torch.uint8)
        if img.dim() == 5:
            img = img[:, 0].contiguous()
        B = img.shape[0]
        tokens = torch.empty(B, g.patches, g.enc_dim, dtype=torch.float32, device=self.device)
        self._ctx.encode(img.data_ptr(), tokens.data_ptr(), B, self._stream())
        return tokens

    def encode_initial_image(self, images):
        """The evaluators' `DINO_encode_image(initial_image).last_hidden_state` on the device
        (data/simpler/evaluate.py:155-163,264-274): uint8 [B, (1,) H, W, 3] -> f32 [B, 1 + P, E] (row 0 = CLS), ready to
        be passed as ``initial_state["patch_embeddings"]``.  Uses the DINOv2 weights of this checkpoint (the
        evaluators use the pretrained ones, which are the same tensors unless the encoder was fine-tuned)."""
        torch = _torch()
        g = self.geometry
        img = self._dev(images, torch.uint8)
        if img.dim() == 5:
            img = img[:, 0].contiguous()
        B = img.shape[0]
        hidden = torch.empty(B, g.patches + 1, g.enc_dim, dtype=torch.float32, device=self.device)
        self._ctx.encode_hidden(img.data_ptr(), hidden.data_ptr(), B, self._stream())
        return hidden

    def policy_from_tokens(self, tokens, base_params: GeneratedWeights):
        torch = _torch()
        g = self.geometry
        tok = self._dev(tokens, torch.float32)
        B = base_params.batch
        if tuple(tok.shape) != (B, g.patches, g.enc_dim):
            raise ValueError(f"tokens must be [{B}, {g.patches}, {g.enc_dim}]")
        actions = torch.empty(B, g.horizon, g.action_dim, dtype=torch.float32, device=self.device)
        logits = torch.empty(B, g.horizon, dtype=torch.float32, device=self.device)
        self._ctx.policy(base_params._h, tok.data_ptr(), actions.data_ptr(), logits.data_ptr(), B, self._stream())
        return actions, logits


    def action_loss(self, actions, gripper_logits, batch):
        """Per-sample MixActionHead.loss (action_heads.py:474-522) of policy outputs against an OXE batch
        {"action" [B,1,H,7], "action_pad_mask" [B,1,H,7], "timestep_pad_mask" [B,1]}; returns (loss [B], mean)."""
        torch = _torch()
        g = self.geometry
        act = self._dev(actions, torch.float32)
        lg = self._dev(gripper_logits, torch.float32)
        B = act.shape[0]
        tgt = self._dev(np.asarray(batch["action"])[:, 0], torch.float32)
        am = self._dev(np.asarray(batch["action_pad_mask"])[:, 0].astype(np.uint8), torch.uint8)
        tm = self._dev(np.asarray(batch["timestep_pad_mask"])[:, 0].astype(np.uint8), torch.uint8)
        if tuple(tgt.shape) != (B, g.horizon, g.action_dim) or tuple(lg.shape) != (B, g.horizon):
            raise ValueError("bad loss shapes")
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        self._ctx.loss(act.data_ptr(), lg.data_ptr(), tgt.data_ptr(), tm.data_ptr(), am.data_ptr(), out.data_ptr(), B,
                       self._stream())
        return out, out.mean()


HyperVLAModel = HyperVLA        # the name BASELINE.json's north_star uses


def _tree_map(fn, tree):
    if isinstance(tree, dict):
        return {k: _tree_map(fn, v) for k, v in tree.items()}
    return fn(tree)
