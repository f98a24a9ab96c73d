"""Diagnostic (GPU box): where the context-encoder kernel spends its time.  Workgroup 0 stamps the shader clock at every
phase boundary (libhvla_bench.so, hypernet.hip CTX_STAMP); this prints the phases of one launch in shader-clock kilocycles
and as shares of the launch, next to the time of a whole create_tasks call (T5 excluded: pre-embedded tokens)."""
import os, sys, ctypes
os.environ["HVLA_LIBRARY_FLAVOUR"] = "bench"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import numpy as np, torch
from hypervla import synthetic as syn, _native
from hypervla.config import FULL
from hypervla.model import HyperVLA
g = FULL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = HyperVLA.from_synthetic(g, max_batch=B)
ins, st = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g)
for _ in range(3):
    w, _, _ = m.create_tasks(instruction_dict=ins, initial_state=st)
torch.cuda.synchronize()
lib = _native.load_library()
buf = (ctypes.c_ulonglong * 64)()
lib.hvla_debug_ctx_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
lib.hvla_debug_ctx_stamps.restype = ctypes.c_int
assert lib.hvla_debug_ctx_stamps(buf) == 0
s = np.array(list(buf), dtype=np.int64)
L = g.ctx_layers
names = ["token projection", "CLS projection"] + [f"L{l} {n}" for l in range(L) for n in ("ln0", "qkv", "attention", "out", "ln1", "fc1", "fc2")]
d = np.diff(s[: 3 + 7 * L]) / 1000.0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    m.create_tasks(instruction_dict=ins, initial_state=st)
e1.record()
torch.cuda.synchronize()
print(f"create_tasks (context encoder + weight generation + bookkeeping): {e0.elapsed_time(e1) / 10:.3f} ms per call")
tot = {}
for n, v in zip(names, d):
    key = n.split(" ", 1)[1] if n.startswith("L") else n
    tot[key] = tot.get(key, 0.0) + v
print(f"B = {B}; workgroup 0, kilocycles per phase summed over the {L} layers (total {d.sum():.1f} kcycles)")
for k, v in tot.items():
    print(f"  {k:18s} {v:8.1f}  {100 * v / d.sum():5.1f} %")
