"""Diagnostic (GPU box): where one episode's time goes inside policy_kernel -- shader-clock stamps of episode 0 / wave 0 at the
phase boundaries (libhvla_bench.so, hvla_debug_policy_stamps), printed as a share of the kernel.

    python tools/policy_timeline.py [B]
"""
import ctypes as C, os, sys
os.environ["HVLA_LIBRARY_FLAVOUR"] = "bench"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = FULL
m = HyperVLA.from_synthetic(g, max_batch=B)
ins, st = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g)
w, tasks, _ = m.create_tasks(instruction_dict=ins, initial_state=st)
tok = torch.randn(B, g.patches, g.enc_dim, device=m.device)
act = torch.zeros(B, g.horizon, g.action_dim, device=m.device)
logit = torch.zeros(B, g.horizon, device=m.device)
lib = m._ctx.lib
n = 3 + 6 * g.layers
out = (C.c_ulonglong * n)()
lib.hvla_debug_policy_stamps.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.POINTER(C.c_ulonglong), C.c_int32]
rc = lib.hvla_debug_policy_stamps(m._ctx.h, w._h, tok.data_ptr(), act.data_ptr(), logit.data_ptr(), B, out, n)
assert rc == 0, rc
t = np.array(list(out), dtype=np.float64)
t -= t[0]
names = ["start", "projection"]
for l in range(g.layers):
    names += [f"L{l} q/k/v heads 0-1", f"L{l} attention 0-1", f"L{l} q/k/v heads 2-3", f"L{l} attention 2-3", f"L{l} out-projection", f"L{l} MLP"]
names += ["(layers done)"]
tot = t[-1]
print(f"episode 0, wave 0, B = {B}: {tot:.0f} clock ticks from start to the head")
for i in range(1, n):
    print(f"  {names[i]:24s} {t[i] - t[i - 1]:9.0f}  {100 * (t[i] - t[i - 1]) / tot:5.1f} %")
agg = {}
for i in range(2, n - 1):
    k = names[i].split(" ", 1)[1]
    k = "q/k/v tiles" if k.startswith("q/k/v") else ("attention" if k.startswith("attention") else k)
    agg[k] = agg.get(k, 0) + t[i] - t[i - 1]
print("  sums over the layers:", {k: f"{100 * v / tot:.1f} %" for k, v in agg.items()}, f"projection {100 * t[1] / tot:.1f} %")
