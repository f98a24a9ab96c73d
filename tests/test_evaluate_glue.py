"""N4 glue: the shared-memory vector environment (CPU) and the batched evaluator (GPU: batched == one episode at a time)."""
import numpy as np
import pytest


class ToyEnv:
    """Frames depend on (seed, t) only, so every driver sees the same observations; it records what it was told to do."""

    def __init__(self, seed, size=48, goal=3, limit=6):
        self.seed, self.size, self.goal, self.limit = seed, size, goal, limit
        self.t, self.log = 0, []

    def _frame(self):
        return np.random.default_rng(1000 * self.seed + self.t).integers(0, 256, (self.size, self.size, 3), dtype=np.uint8)

    def reset(self, **kw):
        self.t, self.log = 0, []
        return self._frame(), {"seed": self.seed}

    def get_language_instruction(self):
        return f"move block {self.seed}"

    def step(self, action):
        self.t += 1
        self.log.append(np.array(action, dtype=np.float64))
        return self._frame(), float(self.t), self.t >= self.goal, self.t >= self.limit, {"t": self.t}

    def get_log(self):
        return np.array(self.log)


def _fn(seed, **kw):
    import functools
    return functools.partial(ToyEnv, seed, **kw)


@pytest.mark.parametrize("kind", ["dummy", "shmem"])
def test_vector_env_frames_steps_and_calls(kind):
    from hypervla.evaluate import DummyVectorEnv, ShmemVectorEnv
    fns = [_fn(s, goal=2 + s) for s in range(3)]
    venv = (DummyVectorEnv if kind == "dummy" else ShmemVectorEnv)(fns, (48, 48, 3))
    try:
        assert len(venv) == 3
        infos = venv.reset()
        assert [i["seed"] for i in infos] == [0, 1, 2]
        for s in range(3):
            np.testing.assert_array_equal(venv.frames[s], ToyEnv(s)._frame())
        assert venv.call("get_language_instruction") == ["move block 0", "move block 1", "move block 2"]
        rew, done, trunc, infos = venv.step(np.arange(21, dtype=np.float32).reshape(3, 7))
        assert rew.tolist() == [1.0, 1.0, 1.0] and not done.any() and [i["t"] for i in infos] == [1, 1, 1]
        before = venv.frames[0].copy()
        rew, done, trunc, _ = venv.step(np.ones((2, 7), np.float32), ids=[1, 2])      # a subset: env 0 keeps its frame
        np.testing.assert_array_equal(venv.frames[0], before)
        assert rew.tolist() == [2.0, 2.0] and done.tolist() == [False, False]
        e = ToyEnv(2); e.t = 2
        np.testing.assert_array_equal(venv.frames[2], e._frame())
        log = venv.call("get_log", ids=[2])[0]
        assert log.shape == (2, 7) and log[0].tolist() == list(range(14, 21)) and log[1].tolist() == [1.0] * 7
    finally:
        venv.close()


def test_shared_block_is_released():
    from multiprocessing import shared_memory
    from hypervla.evaluate import ShmemVectorEnv
    venv = ShmemVectorEnv([_fn(0)], (48, 48, 3))
    name = venv._shm.name
    venv.close()
    with pytest.raises(FileNotFoundError):
        shared_memory.SharedMemory(name=name)


def test_worker_failure_reaches_the_parent():
    from hypervla.evaluate import ShmemVectorEnv
    venv = ShmemVectorEnv([_fn(0)], (48, 48, 3))
    try:
        venv.reset()
        with pytest.raises(RuntimeError):
            venv.call("no_such_method")
    finally:
        venv.close()


@pytest.mark.gpu
def test_batched_evaluator_equals_one_episode_at_a_time():
    """E simulators in lockstep through one batched model step give every simulator the same actions (bit for bit) as the
    reference's loop -- reset, then step -- run for that episode alone (data/simpler/evaluate.py:226-330)."""
    import torch
    from hypervla.config import FULL
    from hypervla.evaluate import BatchEvaluator, ShmemVectorEnv
    from hypervla.interface import InferenceWrapper
    from hypervla.model import HyperVLA
    from hypervla.synthetic import synthetic_instructions
    g = FULL
    E = 5
    m = HyperVLA.from_synthetic(g, max_batch=8)
    base = synthetic_instructions(E, g)["language_instruction"]

    def tokenize(instrs):                      # instruction i -> row i of the seeded synthetic T5 embeddings
        idx = [int(s.split()[-1]) for s in instrs]
        return {k: np.asarray(v)[idx] for k, v in base.items()}

    fns = [_fn(s, size=96, goal=2 + s % 3, limit=4) for s in range(E)]
    venv = ShmemVectorEnv(fns, (96, 96, 3))
    try:
        ev = BatchEvaluator(m, policy_setup="widowx_bridge", pred_action_horizon=g.horizon, action_ensemble=True, crop=True)
        res = ev.run(venv, tokenize, max_steps=10)
        logs = venv.call("get_log")
    finally:
        venv.close()
    assert res["success"].tolist() == [True] * E
    assert res["steps"].tolist() == [2 + s % 3 for s in range(E)]
    assert res["model_seconds"] > 0 and res["sim_seconds"] > 0
    for s in range(E):
        env = ToyEnv(s, size=96, goal=2 + s % 3, limit=4)
        wr = InferenceWrapper(m, policy_setup="widowx_bridge", horizon=1, pred_action_horizon=g.horizon,
                              image_size=g.image_size, action_ensemble=True, crop=True)
        frame, _ = env.reset()
        ins = {"language_instruction": tokenize([env.get_language_instruction()])}
        wr.reset(env.get_language_instruction(), ins, wr.initial_state_from_image(frame))
        done = False
        while not done:
            _, act, _, _, _ = wr.step(frame)
            frame, _, done, trunc, _ = env.step(act)
        np.testing.assert_array_equal(np.array(env.log), logs[s])
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_batched_evaluator_against_the_oracle():
    """N4 checked against the ORACLE, not against the HIP path itself: what every simulator is told to do by
    `BatchEvaluator.run` equals the float64 restatement of the reference's per-episode loop -- DINOv2 hidden state of the
    first frame -> create_tasks -> per step sample_actions -> un-normalise / ensemble / axis-angle / gripper rule
    (data/simpler/evaluate.py:226-330, data/utils/hypervla_interface.py:141-304) -- run on the same frames.  The toy
    simulator's frames depend on (seed, t) only, so the oracle can replay an episode without the actions."""
    from hypervla import synthetic as syn
    from hypervla.config import FULL, encoder_leaves, generated_leaves
    from hypervla.evaluate import BatchEvaluator, DummyVectorEnv
    from hypervla.model import HyperVLA
    from oracle import hvla_ref_np as onp
    g, E, setup = FULL, 3, "widowx_bridge"
    m = HyperVLA.from_synthetic(g, max_batch=4)
    base = syn.synthetic_instructions(E, g)["language_instruction"]

    def tokenize(instrs):
        idx = [int(s.split()[-1]) for s in instrs]
        return {k: np.asarray(v)[idx] for k, v in base.items()}

    goals = [2, 3, 2]
    fns = [_fn(s, size=g.image_size, goal=goals[s], limit=4) for s in range(E)]      # model-sized frames: no resize in the loop
    venv = DummyVectorEnv(fns, (g.image_size, g.image_size, 3))
    try:
        ev = BatchEvaluator(m, policy_setup=setup, pred_action_horizon=g.horizon, action_ensemble=True, crop=False)
        res = ev.run(venv, tokenize, max_steps=6)
        logs = venv.call("get_log")
    finally:
        venv.close()
    assert res["steps"].tolist() == goals
    P, leaves, enc = m.params, generated_leaves(g), dict(encoder_leaves(g))
    stats = m.dataset_statistics["bridge_dataset"]["action"]
    for s in range(E):
        env = ToyEnv(s, size=g.image_size, goal=goals[s], limit=4)
        frames = []
        f, _ = env.reset()
        for t in range(goals[s]):
            frames.append(f)
            f = env.step(np.zeros(7))[0]
        frames = np.stack(frames)
        hidden = onp.dinov2(P, g, enc, onp.normalize_images(frames[:1]))
        ins = {"language_instruction": {k: np.asarray(v)[s:s + 1] for k, v in base.items()}}
        bp, _ = onp.create_tasks(P, g, leaves, ins, {"patch_embeddings": hidden})
        raws, logits = [], []
        for t in range(goals[s]):
            a, lg, _, _ = onp.sample_actions(P, g, enc, bp, frames[t:t + 1])
            raws.append(a[0]), logits.append(lg[0])
        _, want = onp.postprocess_episode(np.stack(raws), stats, setup, True, g.horizon)
        got = np.asarray(logs[s])
        assert got.shape == want.shape == (goals[s], 7)
        # translation: |d action| <= 1e-3 times the un-normalisation scale (std <= 0.5); rotation goes through euler -> axis-angle
        assert np.abs(got[:, :6] - want[:, :6]).max() <= 2e-3, np.abs(got[:, :6] - want[:, :6]).max()
        if np.abs(np.stack(logits)).min() > 1e-2:              # thresholded column: equal wherever no logit sits on the threshold
            np.testing.assert_array_equal(got[:, 6], want[:, 6])
