import sys, time; sys.path.insert(0,'/root/repo/hyper-vla_amd'); sys.path.insert(0,'/root/repo')
import torch, numpy as np
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA
g=FULL
def setup(B, rank):
    m=HyperVLA.from_synthetic(g, max_batch=B)
    ins,st=syn.synthetic_instructions(B,g,rank),syn.synthetic_initial_state(B,g,rank)
    im=torch.as_tensor(syn.synthetic_images(B,g,rank)[:,0]).cuda().contiguous()
    w,_,_=m.create_tasks(instruction_dict=ins,initial_state=st)
    act=torch.empty(B,4,7,device='cuda'); lg=torch.empty(B,4,device='cuda')
    return m,w,im,act,lg
def run(parts, B, iters=20):
    objs=[setup(B,i) for i in range(parts)]
    streams=[torch.cuda.Stream() for _ in range(parts)]
    def step():
        for (m,w,im,act,lg),s in zip(objs,streams):
            m._ctx.step(w._h, im.data_ptr(), act.data_ptr(), lg.data_ptr(), B, s.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t)/iters
    print(f"{parts} stream(s) x B={B}: {dt*1e3:.2f} ms/step {parts*B/dt:.0f} actions/s")
run(1,256); run(2,128); run(2,256); run(1,512); run(4,64)
