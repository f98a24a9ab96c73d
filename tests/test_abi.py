"""CPU: the C-ABI library loads and exports exactly what include/hvla.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hvla.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hvla_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from hypervla import _native
    assert _declared() == sorted(_native.EXPORTS)


def test_library_exports_every_declared_symbol():
    from hypervla import _native
    if not os.path.exists(_native.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_native.lib_path())
    for name in _declared():
        assert hasattr(lib, name), name
    _native.load_library()


def test_config_struct_layout_is_the_header_s(tmp_path):
    """The ctypes mirror of hvla_config against the C compiler's view of include/hvla.h: size and every field offset."""
    import subprocess
    from hypervla import _native
    names = [n for n, _ in _native.hvla_config._fields_]
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"hvla.h\"\nint main(void) { printf(\"%zu\", sizeof(hvla_config));\n"
    prog += "".join(f'printf(" %zu", offsetof(hvla_config, {n}));\n' for n in names) + "return 0; }\n"
    (tmp_path / "l.c").write_text(prog)
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(tmp_path / "l.c"), "-o", str(tmp_path / "l")], check=True)
    got = [int(x) for x in subprocess.run([str(tmp_path / "l")], check=True, capture_output=True, text=True).stdout.split()]
    assert got[0] == ctypes.sizeof(_native.hvla_config)
    assert got[1:] == [getattr(_native.hvla_config, n).offset for n in names]
    assert names[0] == "struct_size"


def test_a_config_struct_of_another_size_is_refused():
    """hvla_config.struct_size: a binder built against another header (a shorter or a longer struct) gets HVLA_E_SHAPE from
    hvla_create before anything of the struct is read -- with or without a GPU (the struct used to be copied blindly)."""
    from hypervla import _native
    lib = _native.load_library()
    cfg = _native.hvla_config()
    for bad in (0, ctypes.sizeof(_native.hvla_config) - 4, ctypes.sizeof(_native.hvla_config) + 4):
        cfg.struct_size = bad
        h = ctypes.c_void_p()
        assert lib.hvla_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -1 and not h.value      # HVLA_E_SHAPE


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hypervla import _native
    from hypervla.config import FULL
    from hypervla.model import HyperVLA
    with pytest.raises(_native.NativeError):
        _native.Context(FULL, 0, 4)                      # hvla_create -> HVLA_E_DEVICE
    with pytest.raises(RuntimeError):
        HyperVLA.from_synthetic(FULL)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hyper-vla_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "hvla_ref_" not in txt, f


def test_no_kernel_of_the_library_spills_or_uses_scratch(tmp_path):
    """Every gfx950 kernel in libhvla.so (the code objects inside its .hip_fatbin section, read with the ROCm LLVM tools): no VGPR spill,
    no private segment.  A scratch access in one of these kernels is not only slow: behind LDS-DMA or a prefetch in flight its
    reload waits for all of it (vmcnt retires in order) -- VERDICT r4 found two spilling instantiations in the round-4 library."""
    import struct
    import subprocess
    from hypervla import _native
    llvm = "/opt/rocm/lib/llvm/bin/"
    if not (os.path.exists(llvm + "llvm-objcopy") and os.path.exists(llvm + "llvm-readelf")):
        pytest.skip("ROCm LLVM tools not installed")
    if not os.path.exists(_native.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    fb = str(tmp_path / "fat.bin")
    subprocess.run([llvm + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, _native.lib_path(), str(tmp_path / "x.so")], check=True)
    data = open(fb, "rb").read()
    magic, pos, kernels, bad, sgpr = b"__CLANG_OFFLOAD_BUNDLE__", 0, 0, [], {}
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        p = i + 32
        for _ in range(struct.unpack_from("<Q", data, i + 24)[0]):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                co = str(tmp_path / "co.elf")
                open(co, "wb").write(data[i + off:i + off + size])
                notes = subprocess.run([llvm + "llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
                for blk in notes.split("- .agpr_count:")[1:]:
                    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
                    kernels += 1
                    if int(g("vgpr_spill_count")) or int(g("private_segment_fixed_size")):
                        bad.append((g("name"), g("vgpr_spill_count"), g("private_segment_fixed_size")))
                    sgpr[g("name")] = int(g("sgpr_spill_count"))
        pos = i + 24
    assert kernels > 100, kernels
    assert not bad, bad
    # SGPRs parked in VGPR lanes (v_writelane / v_readlane) are not memory, but each is an instruction pair per use -- held to NUMBERS
    # (VERDICT r5 item 8): the QKV GEMM and every kernel not listed has none; fc1's persistent form 13 (its GELU epilogue's table
    # addresses, outside the K loop); the residual GEMMs with the fused LayerNorm 62 and the patch embedding with it 53 (epilogue only:
    # the K loop's ISA has no v_readlane / v_writelane); the policy megakernel 43; the context encoder (once per episode) 73.
    limits = [("ctx_encoder_kernel", 73), ("policy_kernel", 43), ("gemm256p_kernelINS_5OpF16ELi1", 0), ("gemm256p_kernelINS_6OpBF16ELi1", 0),
              ("gemm256p_kernelINS_5OpF16ELi2", 13), ("gemm256p_kernelINS_6OpBF16ELi2", 13), ("gemm256p_kernel", 62)]
    print("SGPRs parked in VGPR lanes, largest:", sorted(sgpr.items(), key=lambda kv: -kv[1])[:6])
    for name, n in sgpr.items():
        allowed = next((lim for key, lim in limits if key in name), 0)
        assert n <= allowed, (name, n, allowed)
