"""Diagnostic (GPU box): throughput of N half-batch steps on N streams vs one stream, and what the dominant kernel's
HIP-event duration looks like under that concurrency."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import torch, numpy as np
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA
g = FULL
FC1 = 2 * 257 * 768 * 3072
def setup(B, rank):
    m = HyperVLA.from_synthetic(g, max_batch=B)
    ins, st = syn.synthetic_instructions(B, g, rank), syn.synthetic_initial_state(B, g, rank)
    im = torch.as_tensor(syn.synthetic_images(B, g, rank)[:, 0]).cuda().contiguous()
    w, _, _ = m.create_tasks(instruction_dict=ins, initial_state=st)
    return m, w, im, torch.empty(B, 4, 7, device="cuda"), torch.empty(B, 4, device="cuda")
def run(parts, B, iters=20):
    objs = [setup(B, i) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def step():
        for (m, w, im, act, lg), s in zip(objs, streams):
            m._ctx.step(w._h, im.data_ptr(), act.data_ptr(), lg.data_ptr(), B, s.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    for o in objs: o[0]._ctx.profile(1)
    t = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / iters
    ms, n = 0.0, 0
    for o in objs:
        r = o[0]._ctx.profile_read()["fc1_gemm"]; ms += r[0]; n += r[1]; o[0]._ctx.profile(0)
    per = ms / n
    print(f"{parts} stream(s) x B={B}: {dt*1e3:.2f} ms/step {parts*B/dt:.0f} actions/s; fc1 {per*1e3:.0f} us/launch -> {FC1*B/per/1e9:.0f} TF/s per launch")
run(1, 256); run(2, 128); run(2, 256); run(4, 64)
