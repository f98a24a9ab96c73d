"""CPU, world_size 2 over gloo: the N > 1 path of bench.py (episode sharding, barrier, max-over-ranks)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist            # noqa: E402
import torch.multiprocessing as mp          # noqa: E402

from hypervla import synthetic as syn       # noqa: E402
from hypervla.config import TINY            # noqa: E402
from hypervla.dp import episode_range, max_over_ranks, whole_job_rate   # noqa: E402


def test_episode_ranges_partition():
    for total in (0, 1, 7, 8, 256, 8192):
        for world in (1, 2, 3, 8):
            r = [episode_range(total, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        episode_range(8, 2, 2)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = 3
        ins = syn.synthetic_instructions(B, TINY, rank)          # per-rank synthetic shard (seed + rank)
        digest = float(np.abs(ins["language_instruction"]["token_embedding"]).sum())
        dist.barrier()
        t = max_over_ranks(0.25 * (rank + 1))                     # slowest rank defines the job time
        gathered = [None] * world
        dist.all_gather_object(gathered, digest)
        q.put((rank, t, gathered, whole_job_rate(B, world, 10, t)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    out = sorted(q.get(timeout=90) for _ in ps)
    [p.join(30) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (r0, t0, g0, v0), (r1, t1, g1, v1) = out
    assert t0 == t1 == 0.5                                       # max over ranks
    assert g0 == g1 and g0[0] != g0[1]                           # different episodes on different ranks
    assert v0 == v1 == 2 * 3 * 10 / 0.5


def test_bench_self_launch_builds_the_launcher_command(monkeypatch):
    """`python bench.py --gpus 8` with no launcher: the ranks are started as a child `torch.distributed.run` on the loopback
    interface with the caller's own arguments, and its exit code is what the script returns; asking for more GPUs than the
    node has is refused before anything is spawned."""
    import importlib
    import subprocess
    import sys
    bench = importlib.import_module("bench")
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--batch", "1024", "--steps", "5"])
    monkeypatch.setenv("HVLA_BENCH_SHARE_GPU", "1")
    assert bench.self_launch(8) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "8", "--batch", "1024", "--steps", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.delenv("HVLA_BENCH_SHARE_GPU")
    seen.clear()
    monkeypatch.setattr(bench, "kfd_gpu_count", lambda: 4)
    assert bench.self_launch(8) == 2 and not seen          # fewer GPUs than asked for: refused, nothing spawned
    monkeypatch.setattr(bench, "kfd_gpu_count", lambda: None)
    assert bench.self_launch(8) == 7 and seen              # sysfs unreadable: the ranks report the problem themselves


def test_gpu_count_comes_from_sysfs_not_from_hip(tmp_path, monkeypatch):
    """The launcher parent must not initialise HIP: GPUs are KFD topology nodes with SIMDs, cut down by the
    *_VISIBLE_DEVICES variables."""
    import importlib
    bench = importlib.import_module("bench")
    for i, simd in enumerate((0, 1024, 1024, 1024)):          # node 0 is the CPU
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench.kfd_gpu_count(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.kfd_gpu_count(str(tmp_path)) == 2
    assert bench.kfd_gpu_count(str(tmp_path / "absent")) is None
