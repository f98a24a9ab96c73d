"""Caller-side counterpart of the reference's ``InferenceWrapper``
(data/utils/hypervla_interface.py:18-304) for this library's :class:`HyperVLA`.

Given identical post-resize ``uint8[224,224,3]`` observations it reproduces the reference's
history / pad-mask bookkeeping (:123-139), un-normalisation (:219-242), temporal ensemble (:250-253,
== data/utils/action_ensemble.py:15-27 with temperature 0), euler -> axis-angle (:261-267), the
per-``policy_setup`` gripper rules (:269-299) and the 5-tuple it returns (:304).  The lanczos3 resize
(:89-121), padded_resize included, runs on the device (hvla_preprocess).
"""
from __future__ import annotations

import time
from collections import deque
from typing import Optional

import numpy as np


def euler2axangle(ai: float, aj: float, ak: float):
    """Static-frame xyz euler angles -> (axis, angle), via the unit quaternion (what
    transforms3d.euler.euler2axangle computes for its default axes)."""
    ci, si = np.cos(ai / 2.0), np.sin(ai / 2.0)
    cj, sj = np.cos(aj / 2.0), np.sin(aj / 2.0)
    ck, sk = np.cos(ak / 2.0), np.sin(ak / 2.0)
    w = cj * ci * ck + sj * si * sk
    x = cj * si * ck - sj * ci * sk
    y = cj * si * sk + sj * ci * ck
    z = cj * ci * sk - sj * si * ck
    n = np.sqrt(w * w + x * x + y * y + z * z)
    if n < 1e-8:
        return np.array([1.0, 0.0, 0.0]), 0.0
    w, x, y, z = w / n, x / n, y / n, z / n
    v = np.sqrt(x * x + y * y + z * z)
    if v < np.finfo(np.float64).eps * 3.0:
        return np.array([1.0, 0.0, 0.0]), 0.0
    return np.array([x, y, z]) / v, 2.0 * np.arctan2(v, w)


def device_unnormalization(stats: dict, normalization_type: str = "normal"):
    """(mean, std, mask) float32 / float32 / uint8 vectors for ``hvla_ensemble`` (include/hvla.h), which un-normalises with
    ``a * std + mean`` on the masked columns: the dataset's own mean / std for NormalizationType.NORMAL
    (data/utils/hypervla_interface.py:219-230), and for BOUNDS (:231-242) the pair that makes the same map out of
    ``(a + 1) * (p99 - p01 + 1e-8) / 2 + p01``."""
    if str(normalization_type).lower() in ("bounds",):
        p01, p99 = np.asarray(stats["p01"], np.float64), np.asarray(stats["p99"], np.float64)
        std = (p99 - p01 + 1e-8) / 2
        mean = p01 + std
        mask = np.asarray(stats.get("mask", np.ones_like(p01, dtype=bool)), bool)
    elif str(normalization_type).lower() in ("normal",):
        mean, std = np.asarray(stats["mean"], np.float64), np.asarray(stats["std"], np.float64)
        mask = np.asarray(stats.get("mask", np.ones_like(mean, dtype=bool)), bool)
    else:
        raise ValueError(f"Unknown normalization type: {normalization_type}")
    return mean.astype(np.float32), std.astype(np.float32), mask.astype(np.uint8)


class ActionEnsembler:
    """Temporal ensemble over the last ``pred_action_horizon`` predictions; works for one episode
    ([horizon, 7] inputs) or a batch ([B, horizon, 7])."""

    def __init__(self, pred_action_horizon: int, action_ensemble_temp: float = 0.0):
        self.pred_action_horizon = pred_action_horizon
        self.action_ensemble_temp = action_ensemble_temp
        self.action_history = deque(maxlen=pred_action_horizon)

    def reset(self):
        self.action_history.clear()

    def ensemble_action(self, cur_action):
        self.action_history.append(np.asarray(cur_action))
        n = len(self.action_history)
        aligned = [pred[..., n - 1 - idx, :] for idx, pred in enumerate(self.action_history)]
        w = np.exp(-self.action_ensemble_temp * np.arange(n))
        w = w / w.sum()
        return sum(wi * a for wi, a in zip(w, aligned))


_DATASET = {"google_robot": "fractal20220817_data", "widowx_bridge": "bridge_dataset", "libero": "libero"}


class InferenceWrapper:
    def __init__(self, model=None, policy_setup: str = "libero", horizon: int = 1, pred_action_horizon: int = 1,
                 exec_horizon: int = 1, image_size: int = 256, init_rng: int = 0, action_ensemble: bool = False,
                 crop: bool = False, save_attention_map: bool = False, padded_resize: bool = False):
        if policy_setup not in _DATASET:
            raise ValueError(f"Unknown policy setup: {policy_setup}")
        self.save_attention_map = save_attention_map          # hypervla_interface.py:74,208-217
        self.dino_attention_map = self.head_attention_map = None
        self.model, self.policy_setup = model, policy_setup
        self.image_size, self.horizon = image_size, horizon
        self.pred_action_horizon, self.exec_horizon = pred_action_horizon, exec_horizon
        self.action_ensemble, self.crop, self.padded_resize = action_ensemble, crop, padded_resize
        self.sticky_gripper_num_repeat = {"google_robot": 15, "widowx_bridge": 1}.get(policy_setup, 0)
        dataset = _DATASET[policy_setup]
        stats = model.dataset_statistics
        self.unnormalization_statistics = stats["action"] if "action" in stats else stats[dataset]["action"]
        dk = model.config["dataset_kwargs"]
        if "dataset_kwargs" in dk:
            self.normalization_type = dk["dataset_kwargs"]["action_proprio_normalization_type"]
        else:
            self.normalization_type = next(d["action_proprio_normalization_type"]
                                           for d in dk["dataset_kwargs_list"] if d["name"] == dataset)
        self.action_ensembler = ActionEnsembler(pred_action_horizon, 0.0) if action_ensemble else None
        self.image_history = deque(maxlen=horizon)
        self.task = self.task_description = self.base_params = self.instruction_dict = None
        self._reset_gripper()
        self.num_image_history = 0
        self.episode_step = 0

    def _reset_gripper(self):
        self.sticky_action_is_on, self.gripper_action_repeat = False, 0
        self.sticky_gripper_action, self.previous_gripper_action = 0.0, None

    def _resize_image(self, image: np.ndarray) -> np.ndarray:
        """hypervla_interface.py:89-121 on the device (optional resize_with_pad to 256 x 320, lanczos3 antialias resize,
        optional sqrt(0.9) crop)."""
        if image.shape[:2] == (self.image_size, self.image_size) and not self.crop and not self.padded_resize:
            return image                       # a same-size lanczos3 resize reproduces the uint8 frame
        if self.image_size != self.model.geometry.image_size:
            raise ValueError(f"image_size {self.image_size} != the model's {self.model.geometry.image_size}")
        return self.model.preprocess_images(np.asarray(image), crop=self.crop, padded_resize=self.padded_resize)[0].cpu().numpy()

    def initial_state_from_image(self, image: np.ndarray):
        """The dict the evaluators assemble before `reset` (data/simpler/evaluate.py:264-274), with the DINOv2
        `last_hidden_state` of the first frame computed on the device instead of by a host-side HF Flax model."""
        frame = self._resize_image(image)
        hidden = self.model.encode_initial_image(frame[None])
        return {"image_primary": frame, "patch_embeddings": hidden,
                "pad_mask_dict": {"image_primary": np.ones((1, 1))}}

    def reset(self, task_description: str, instruction_dict, initial_state=None) -> None:
        self.base_params, self.task, _ = self.model.create_tasks(instruction_dict=instruction_dict,
                                                                 initial_state=initial_state)
        self.instruction_dict, self.task_description = instruction_dict, task_description
        self.image_history.clear()
        if self.action_ensembler is not None:
            self.action_ensembler.reset()
        self.num_image_history = 0
        self._reset_gripper()
        self.episode_step = 0

    def unnormalize(self, raw_actions: np.ndarray) -> np.ndarray:
        s = self.unnormalization_statistics
        if self.normalization_type in ("normal", "NORMAL"):
            mask = np.asarray(s.get("mask", np.ones_like(s["mean"], dtype=bool)), bool)
            a = raw_actions[..., :len(mask)]
            return np.where(mask, a * s["std"] + s["mean"], a)
        if self.normalization_type in ("bounds", "BOUNDS"):
            mask = np.asarray(s.get("mask", np.ones_like(s["p01"], dtype=bool)), bool)
            a = raw_actions[..., :len(mask)]
            return np.where(mask, (a + 1) * (s["p99"] - s["p01"] + 1e-8) / 2 + s["p01"], a)
        raise ValueError(f"Unknown normalization type: {self.normalization_type}")

    def postprocess(self, raw_actions: np.ndarray):
        """raw model output [pred_action_horizon, 7] -> (raw_action [7], env action [7])."""
        raw_actions = self.unnormalize(np.asarray(raw_actions, np.float64))
        assert raw_actions.shape == (self.pred_action_horizon, 7)
        raw_action = self.action_ensembler.ensemble_action(raw_actions) if self.action_ensemble \
            else np.array(raw_actions[0])
        roll, pitch, yaw = np.asarray(raw_action[3:6], dtype=np.float64)
        ax, angle = euler2axangle(roll, pitch, yaw)
        rot = ax * angle
        if self.policy_setup == "google_robot":
            cur = float(raw_action[-1])
            rel = 0.0 if self.previous_gripper_action is None else self.previous_gripper_action - cur
            self.previous_gripper_action = cur
            if abs(rel) > 0.5 and not self.sticky_action_is_on:
                self.sticky_action_is_on, self.sticky_gripper_action = True, rel
            if self.sticky_action_is_on:
                self.gripper_action_repeat += 1
                rel = self.sticky_gripper_action
            if self.gripper_action_repeat == self.sticky_gripper_num_repeat:
                self.sticky_action_is_on, self.gripper_action_repeat, self.sticky_gripper_action = False, 0, 0.0
            grip = rel
        elif self.policy_setup == "widowx_bridge":
            grip = 2.0 * float(raw_action[-1] > 0.5) - 1.0
        else:
            grip = 2.0 * raw_action[-1] - 1.0
        action = np.concatenate([raw_action[:3], rot.astype(np.float32), np.array([grip]).astype(np.float32)])
        return raw_action, action

    def step(self, image: np.ndarray, task_description: Optional[str] = None, image_embeddings=None, *a, **k):
        if task_description is not None and task_description != self.task_description:
            raise ValueError("task changed: call reset(task_description, instruction_dict, initial_state)")
        assert image.dtype == np.uint8
        image = self._resize_image(image)
        self.image_history.append(image)
        self.num_image_history = min(self.num_image_history + 1, self.horizon)
        images = np.stack(self.image_history, axis=0)[None]
        n = len(self.image_history)
        pad_mask = np.ones(n, dtype=np.float64)
        pad_mask[: n - min(n, self.num_image_history)] = 0
        start = time.time()
        extra = {"attention_maps": True} if self.save_attention_map else {}     # the reference's signature otherwise
        raw_actions, inter = self.model.sample_actions(images, self.instruction_dict, self.task, pad_mask[None],
                                                       self.base_params, rng=None, image_embeddings=image_embeddings, **extra)
        end = time.time()                      # numpy in -> numpy out: sample_actions has synchronised
        if self.save_attention_map:            # hypervla_interface.py:208-217: [12, 12, P] and [4, 4, P] of this episode
            self.dino_attention_map = np.asarray(inter["dino_cls_attention"])[0]
            self.head_attention_map = np.asarray(inter["head_attention"])[0]
        raw_action, action = self.postprocess(np.asarray(raw_actions)[0])
        self.episode_step += 1
        return raw_action, action, image, (self.task_description, self.task), (end - start)
