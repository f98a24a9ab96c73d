"""Diagnostic (GPU box): is the fine-tune step launch-bound at small batch?  eager vs hipGraph replay of fwd+bwd+apply."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import numpy as np, torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA
from hypervla.train import FineTuner
g = FULL
for B, enc in ((8, False), (32, False), (8, True), (32, True)):
    m = HyperVLA.from_synthetic(g, max_batch=B)
    ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
    batch = syn.synthetic_action_batch(B, g)
    ft = FineTuner(m, B, train_encoder=enc)
    x = torch.as_tensor(im[:, 0]).to(m.device) if enc else m.encode_images(im)
    li = ins["language_instruction"]
    ins_d = {"language_instruction": {"token_embedding": torch.as_tensor(li["token_embedding"]).to(m.device),
                                      "attention_mask": torch.as_tensor(li["attention_mask"]).to(m.device), "input_ids": li["input_ids"]}}
    def step():
        ft.forward_backward(ins_d, st, x, batch)
        ft.apply(lr=1e-4)
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t) / 10
    # capture: the host-side staging of forward_backward (numpy -> device copies) cannot be captured, so capture the C calls
    tok, msk, cls, obs, tgt, am, tm = ft._keep
    ptrs = [tok.data_ptr(), msk.data_ptr(), cls.data_ptr(), None if enc else obs.data_ptr(), obs.data_ptr() if enc else None,
            tgt.data_ptr(), tm.data_ptr(), am.data_ptr()]
    side = torch.cuda.Stream(m.device)
    with torch.cuda.stream(side):
        m._ctx.train_step(ft.buf, ptrs, B, ft._hyper(0.0), m._stream())
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            m._ctx.train_step(ft.buf, ptrs, B, ft._hyper(0.0), m._stream())
            m._ctx.train_apply(ft.buf, ft._hyper(1e-4), m._stream())
    for _ in range(3): graph.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): graph.replay()
    torch.cuda.synchronize(); rep = (time.perf_counter() - t) / 10
    print(f"B={B} train_encoder={enc}: eager {eager*1e3:.2f} ms/step, graph replay {rep*1e3:.2f} ms/step")
