import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd")); sys.path.insert(0, ROOT)
import torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA
m = HyperVLA.from_synthetic(FULL, max_batch=256)
print("model ok", flush=True)
if "probe" in sys.argv:
    print(m._ctx.box_probe(torch.cuda.current_stream(m.device).cuda_stream), flush=True)
im = syn.synthetic_images(256, FULL)[:, 0]
t = m.encode_images(im); torch.cuda.synchronize(); print("encode ok", flush=True)
for i in range(3):
    t = m.encode_images(im); torch.cuda.synchronize(); print("encode", i, flush=True)
