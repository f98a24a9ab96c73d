"""The host-side layout / packing code of the C-ABI library under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build;
GPU sanitizers are not available on the pool).  csrc/layout.h and csrc/pack.h hold all the index arithmetic hvla_create
runs on checkpoint tensors; tests/native/pack_sanitize.cpp drives them on the README geometry with exact-size buffers."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_layout_and_packing_under_asan_ubsan(tmp_path):
    exe = tmp_path / "pack_sanitize"
    build = subprocess.run(
        ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra",
         "-Werror", "-I", os.path.join(ROOT, "hyper-vla_amd", "csrc"), os.path.join(ROOT, "tests", "native", "pack_sanitize.cpp"),
         "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    dump = tmp_path / "half.bin"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe), str(dump)], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.strip().endswith("OK"), run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr
    # the library's software float -> binary16 conversion is numpy's (round to nearest even, subnormals, overflow to inf)
    rec = np.fromfile(dump, dtype=np.uint32).reshape(-1, 2)
    x = rec[:, 0].copy().view(np.float32)
    finite = ~np.isnan(x)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    np.testing.assert_array_equal(rec[finite, 1], want[finite])
    assert finite.sum() > 900000


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_launch_plan_arithmetic_under_asan_ubsan(tmp_path):
    """csrc/plan.h: the XCD tile order of the small GEMM kernels is a bijection with one contiguous run per XCD at every grid size
    (a 64-column slice of W / dW on at most two XCDs at the B = 256 launch shapes), and the non-temporal 16-bit epilogue, whose
    stores carry 32-bit BYTE offsets, is only chosen for outputs below 4 GiB (fc1 from B = 2721 on would wrap)."""
    exe = tmp_path / "plan_check"
    build = subprocess.run(
        ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra",
         "-Werror", "-I", os.path.join(ROOT, "hyper-vla_amd", "csrc"), os.path.join(ROOT, "tests", "native", "plan_check.cpp"),
         "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and run.stdout.strip().endswith("OK"), run.stdout + run.stderr
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr
