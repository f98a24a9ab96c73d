"""actions of the in-tree library vs the BASE variant on the same inputs: the grouped hand-overs change no arithmetic."""
import os, sys, hashlib, subprocess
ROOT = os.getcwd()
code = '''
import sys, os, hashlib
sys.path.insert(0, os.path.join(os.getcwd(), "hyper-vla_amd")); sys.path.insert(0, os.getcwd())
import numpy as np
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA
for B in (1, 3, 64, 256):
    m = HyperVLA.from_synthetic(FULL, max_batch=B)
    ins, st, im = syn.synthetic_instructions(B, FULL), syn.synthetic_initial_state(B, FULL), syn.synthetic_images(B, FULL)
    w, tasks, _ = m.create_tasks(instruction_dict=ins, initial_state=st)
    act, inter = m.sample_actions(im, ins, tasks, np.ones((B, 1)), w)
    print(B, hashlib.sha256(np.ascontiguousarray(act).tobytes()).hexdigest()[:16], hashlib.sha256(np.ascontiguousarray(inter["gripper_logits"]).tobytes()).hexdigest()[:16])
'''
outs = {}
for v in ("GROUP", "BASE"):
    subprocess.run(["cp", f"tmp_variants/lib_{v}.so", "hyper-vla_amd/lib/libhvla.so"], check=True)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    outs[v] = r.stdout
    print(v); print(r.stdout); print(r.stderr[-500:] if r.returncode else "")
subprocess.run(["cp", "tmp_variants/lib_GROUP.so", "hyper-vla_amd/lib/libhvla.so"], check=True)
print("SAME BITS" if outs["GROUP"] == outs["BASE"] and outs["GROUP"].strip() else "DIFFERENT")
