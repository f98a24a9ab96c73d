"""Batched evaluation glue: many simulators, one batched model step.

The reference evaluators drive ONE episode at a time -- `env.reset()`, `model.reset(instruction, ...)`, then
`model.step(image)` / `env.step(action)` until success or truncation (data/simpler/evaluate.py:226-330,
data/libero/evaluate.py:156-197) -- and keep a sub-process vector environment with shared-memory observation buffers
next to them (data/utils/venv.py:183-297 `ShArray` / `_worker`, :357-520 `SubprocEnvWorker`, :523-976
`BaseVectorEnv`).  Here the two are put together the way the hot path wants them: E simulators step in their own
processes and write their camera frames into ONE shared `uint8 [E, H, W, 3]` block; each timestep that block goes to
the device once, is resized there (`hvla_preprocess`), and all E episodes take one `sample_actions` step with their own
generated weights; the per-episode caller logic (un-normalisation, temporal ensemble, euler -> axis-angle, gripper
rules) is the `InferenceWrapper`'s.  No simulator is part of this package: anything with `reset()` / `step(action)` /
`get_language_instruction()` works (`EnvLike`), and the tests use a toy one.
"""
from __future__ import annotations

import multiprocessing as mp
import time
from multiprocessing import shared_memory
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .interface import InferenceWrapper

FrameFn = Callable[[Any, Any], np.ndarray]          # (env, obs) -> uint8 [H, W, 3]


def _default_frame(env, obs) -> np.ndarray:
    return np.asarray(obs, dtype=np.uint8)


def _split_reset(r):
    return r if isinstance(r, tuple) and len(r) == 2 and isinstance(r[1], dict) else (r, {})


def _split_step(r):
    """gymnasium 5-tuple (obs, reward, terminated/success, truncated, info) or gym 4-tuple (obs, reward, done, info)."""
    if len(r) == 5:
        return r
    obs, reward, done, info = r
    return obs, reward, done, False, info


class DummyVectorEnv:
    """All simulators in this process (venv.py:880-925): same interface as `ShmemVectorEnv`."""

    def __init__(self, env_fns: Sequence[Callable[[], Any]], frame_shape: Tuple[int, int, int],
                 get_frame: FrameFn = _default_frame):
        self.envs = [fn() for fn in env_fns]
        self.get_frame = get_frame
        self.frames = np.zeros((len(self.envs),) + tuple(frame_shape), np.uint8)

    def __len__(self):
        return len(self.envs)

    def _ids(self, ids):
        return list(range(len(self.envs))) if ids is None else list(ids)

    def reset(self, ids=None, **kwargs) -> List[dict]:
        infos = []
        for i in self._ids(ids):
            obs, info = _split_reset(self.envs[i].reset(**kwargs))
            self.frames[i] = self.get_frame(self.envs[i], obs)
            infos.append(info)
        return infos

    def step(self, actions: np.ndarray, ids=None):
        ids = self._ids(ids)
        rew, done, trunc, infos = np.zeros(len(ids)), np.zeros(len(ids), bool), np.zeros(len(ids), bool), []
        for k, i in enumerate(ids):
            obs, rew[k], done[k], trunc[k], info = _split_step(self.envs[i].step(actions[k]))
            self.frames[i] = self.get_frame(self.envs[i], obs)
            infos.append(info)
        return rew, done, trunc, infos

    def call(self, name: str, *args, ids=None):
        return [getattr(self.envs[i], name)(*args) for i in self._ids(ids)]

    def close(self):
        for e in self.envs:
            if hasattr(e, "close"):
                e.close()
        self.envs = []


def _worker(conn, env_fn, get_frame, shm_name, index, frame_shape):
    """venv.py:215-297: one simulator; observations go to the shared block, everything else over the pipe."""
    shm = shared_memory.SharedMemory(name=shm_name)
    frames = np.ndarray(frame_shape, np.uint8, buffer=shm.buf)
    env = None
    try:
        env = env_fn()
        while True:
            cmd, data = conn.recv()
            if cmd == "step":
                obs, reward, done, trunc, info = _split_step(env.step(data))
                frames[index] = get_frame(env, obs)
                conn.send((float(reward), bool(done), bool(trunc), info))
            elif cmd == "reset":
                obs, info = _split_reset(env.reset(**data))
                frames[index] = get_frame(env, obs)
                conn.send(info)
            elif cmd == "call":
                name, args = data
                conn.send(getattr(env, name)(*args))
            elif cmd == "close":
                conn.send(None)
                break
            else:
                raise NotImplementedError(cmd)
    except (KeyboardInterrupt, EOFError):
        pass
    except Exception as e:                                   # hand the failure to the parent instead of dying silently
        try:
            conn.send(e)
        except Exception:
            pass
    finally:
        if env is not None and hasattr(env, "close"):
            env.close()
        del frames
        shm.close()
        conn.close()


class ShmemVectorEnv:
    """One process per simulator; frames land in one shared `uint8 [E, H, W, 3]` block (`self.frames`, zero-copy view),
    so a timestep's observations are a single contiguous host buffer for the device copy (venv.py:183-213, 357-520,
    928-976).  `env_fns` must be picklable under the chosen start method ("fork" by default on Linux: a simulator that
    initialises a GPU context in the parent must use "spawn")."""

    def __init__(self, env_fns: Sequence[Callable[[], Any]], frame_shape: Tuple[int, int, int],
                 get_frame: FrameFn = _default_frame, context: str = "fork"):
        n = len(env_fns)
        self._shape = (n,) + tuple(frame_shape)
        self._shm = shared_memory.SharedMemory(create=True, size=int(np.prod(self._shape)))
        self.frames = np.ndarray(self._shape, np.uint8, buffer=self._shm.buf)
        self.frames[:] = 0
        ctx = mp.get_context(context)
        self._conns, self._procs = [], []
        for i, fn in enumerate(env_fns):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker, args=(child, fn, get_frame, self._shm.name, i, self._shape), daemon=True)
            p.start()
            child.close()
            self._conns.append(parent)
            self._procs.append(p)

    def __len__(self):
        return len(self._conns)

    def _ids(self, ids):
        return list(range(len(self._conns))) if ids is None else list(ids)

    def _recv(self, i):
        r = self._conns[i].recv()
        if isinstance(r, Exception):
            raise RuntimeError(f"simulator {i} failed: {r!r}") from r
        return r

    def reset(self, ids=None, **kwargs) -> List[dict]:
        ids = self._ids(ids)
        for i in ids:
            self._conns[i].send(("reset", kwargs))
        return [self._recv(i) for i in ids]

    def step(self, actions: np.ndarray, ids=None):
        """All listed simulators step concurrently; returns (reward, done, truncated, infos) in `ids` order and leaves the
        new frames in `self.frames`."""
        ids = self._ids(ids)
        for k, i in enumerate(ids):
            self._conns[i].send(("step", np.asarray(actions[k])))
        out = [self._recv(i) for i in ids]
        return (np.array([o[0] for o in out]), np.array([o[1] for o in out], bool),
                np.array([o[2] for o in out], bool), [o[3] for o in out])

    def call(self, name: str, *args, ids=None):
        ids = self._ids(ids)
        for i in ids:
            self._conns[i].send(("call", (name, args)))
        return [self._recv(i) for i in ids]

    def close(self):
        for c in self._conns:
            try:
                c.send(("close", None))
                c.recv()
            except (BrokenPipeError, EOFError, OSError):
                pass
            c.close()
        for p in self._procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        self._conns, self._procs = [], []
        if self._shm is not None:
            self.frames = None
            self._shm.close()
            self._shm.unlink()
            self._shm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchEvaluator:
    """E episodes in lockstep.  Per episode it keeps an `InferenceWrapper` for the caller-side state (ensemble history,
    sticky gripper, ...), but the model is stepped ONCE per timestep for all of them."""

    def __init__(self, model, policy_setup: str = "libero", pred_action_horizon: int = 4, action_ensemble: bool = True,
                 crop: bool = False, image_size: Optional[int] = None, padded_resize: bool = False):
        self.model = model
        self.kw = dict(policy_setup=policy_setup, horizon=1, pred_action_horizon=pred_action_horizon,
                       image_size=image_size or model.geometry.image_size, action_ensemble=action_ensemble, crop=crop,
                       padded_resize=padded_resize)
        self.crop, self.padded_resize = crop, padded_resize

    def _frames_to_device(self, frames: np.ndarray):
        import torch
        g = self.model.geometry
        if tuple(frames.shape[1:3]) == (g.image_size, g.image_size) and not self.crop and not self.padded_resize:
            return self.model._dev(frames, torch.uint8)
        return self.model.preprocess_images(frames, crop=self.crop, padded_resize=self.padded_resize)   # on the device

    def run(self, venv, tokenize: Callable[[List[str]], Dict[str, np.ndarray]], max_steps: int,
            instructions: Optional[List[str]] = None, reset_kwargs: Optional[dict] = None,
            success_from: Callable[[bool, dict], bool] = lambda done, info: bool(done)) -> Dict[str, Any]:
        """One batch of episodes: reset every simulator, generate every episode's policy weights once
        (`create_tasks`), then step until each episode has succeeded or been truncated (or `max_steps`).

        `tokenize(list of instructions)` returns the `language_instruction` dict (`input_ids`, `attention_mask` and
        either `token_embedding` or, with a loaded T5 encoder, nothing more)."""
        E = len(venv)
        t_sim = t_model = 0.0
        t0 = time.perf_counter()
        venv.reset(**(reset_kwargs or {}))
        if instructions is None:
            instructions = venv.call("get_language_instruction")
        t_sim += time.perf_counter() - t0
        wrappers = [InferenceWrapper(self.model, **self.kw) for _ in range(E)]
        t0 = time.perf_counter()
        first = self._frames_to_device(venv.frames)
        hidden = self.model.encode_initial_image(first)                      # evaluate.py:264-274, all episodes at once
        inst = {"language_instruction": tokenize(list(instructions))}
        weights, task, _ = self.model.create_tasks(
            instruction_dict=inst, initial_state={"image_primary": first, "patch_embeddings": hidden,
                                                  "pad_mask_dict": {"image_primary": np.ones((E, 1))}})
        for w, ins in zip(wrappers, instructions):                           # what InferenceWrapper.reset leaves behind
            w.task_description, w.base_params, w.task, w.instruction_dict = ins, weights, task, inst
        t_model += time.perf_counter() - t0
        active = np.ones(E, bool)
        success = np.zeros(E, bool)
        steps = np.zeros(E, int)
        actions = np.zeros((E, 7), np.float64)          # what InferenceWrapper.postprocess returns (f64 container)
        raw_log: List[np.ndarray] = []
        for _ in range(max_steps):
            if not active.any():
                break
            t0 = time.perf_counter()
            dev = self._frames_to_device(venv.frames)                        # finished episodes keep their last frame
            raw, _ = self.model.sample_actions(dev, inst, task, None, weights)
            raw = raw.cpu().numpy() if hasattr(raw, "cpu") else np.asarray(raw)
            for i in np.nonzero(active)[0]:
                _, actions[i] = wrappers[i].postprocess(raw[i])
            raw_log.append(raw)
            t_model += time.perf_counter() - t0
            t0 = time.perf_counter()
            ids = np.nonzero(active)[0]
            _, done, trunc, infos = venv.step(actions[ids], ids=ids)
            t_sim += time.perf_counter() - t0
            for k, i in enumerate(ids):
                steps[i] += 1
                if success_from(done[k], infos[k]):
                    success[i], active[i] = True, False
                elif trunc[k]:
                    active[i] = False
        return {"success": success, "steps": steps, "instructions": list(instructions), "model_seconds": t_model,
                "sim_seconds": t_sim, "raw_actions": raw_log}
