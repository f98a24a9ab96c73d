"""Diagnostic (GPU box, under rocprofv3 --kernel-trace --stats): N create_tasks calls at batch B, nothing else."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import torch, time
from hypervla import synthetic as syn
from hypervla.config import FULL as g
from hypervla.model import HyperVLA
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
m = HyperVLA.from_synthetic(g, max_batch=B)
ins, st = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g)
for _ in range(3): m.create_tasks(instruction_dict=ins, initial_state=st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): m.create_tasks(instruction_dict=ins, initial_state=st)
torch.cuda.synchronize()
print(f"B = {B}: {(time.perf_counter() - t0) * 100:.3f} ms per create_tasks (wall clock, 10 calls)")
