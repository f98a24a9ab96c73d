"""Guards on the generated ISA (CPU: hipcc cross-compiles gfx950 without a GPU) for constructs whose correctness rests on what the compiler does NOT
do between two inline-asm statements."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _regs(token_text):
    """VGPR numbers named in an instruction's operand text: v12, v[12:15]."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", token_text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", token_text):
        out.add(int(a))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_streamed_attention_leaves_the_hand_loaded_query_registers_alone_until_the_first_wait(tmp_path):
    """attention_kernel<Op, false, 8> (round 6) requests its query fragments with hand-issued global_load_dwordx4 ("=&v" outputs) so that
    the compiler does not wait for them with vmcnt(0) -- which would also wait for every K / V chunk of the LDS-DMA stream.  Until the
    counted wait of the first hand-over the hardware writes those 16 registers behind the compiler's back: an instruction that reads, copies
    or overwrites them before that wait would work on garbage (round 5 met exactly this with loop-carried registers).  The ISA is checked:
    between the four loads and the first s_barrier of the kernel nothing else names their destination registers."""
    src = os.path.join(ROOT, "hyper-vla_amd", "csrc", "encoder.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-save-temps", "-o", "x.o"],
                   cwd=tmp_path, check=True, capture_output=True, timeout=900)
    asm = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert len(asm) == 1, asm
    text = open(os.path.join(tmp_path, asm[0])).read()
    checked = 0
    for op in ("5OpF16", "6OpBF16"):
        m = re.search(r"^(_ZN4hvla16attention_kernelINS_%sELb0ELi8EEE\w*):" % op, text, re.M)
        assert m, op
        body = text[m.end():]
        body = body[:body.index(".Lfunc_end")]
        lines = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]
        loads = [i for i, ln in enumerate(lines) if ln.startswith("global_load_dwordx4")][:4]
        assert len(loads) == 4 and loads[3] - loads[0] == 3, (op, loads)           # the four hand-issued loads, back to back
        dest = set()
        for i in loads:
            dest |= _regs(lines[i].split(",")[0])
        assert len(dest) == 16, (op, sorted(dest))
        barrier = next(i for i, ln in enumerate(lines) if ln.startswith("s_barrier"))
        assert barrier > loads[3]
        waits = [ln for ln in lines[loads[3] + 1:barrier] if ln.startswith("s_waitcnt vmcnt")]
        assert waits and all(w != "s_waitcnt vmcnt(0)" for w in waits), (op, waits)   # counted, never a full drain before the first hand-over
        first_wait = next(i for i in range(loads[3] + 1, barrier) if lines[i].startswith("s_waitcnt vmcnt"))
        for ln in lines[loads[3] + 1:first_wait]:
            assert not (_regs(ln) & dest), (op, ln)
        checked += 1
    assert checked == 2
