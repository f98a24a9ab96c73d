"""Diagnostic (GPU box): what the vendor GEMM library reaches on the encoder's shapes (not used by the product)."""
import torch, time
B = 256
M = B * 257
for nm, (N, K) in {"qkv": (2304, 768), "out": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}.items():
    for dt in (torch.float16, torch.bfloat16):
        a = torch.randn(M, K, device="cuda", dtype=dt)
        w = torch.randn(K, N, device="cuda", dtype=dt)
        wt = torch.randn(N, K, device="cuda", dtype=dt)
        for lay, f in (("NN", lambda: a @ w), ("NT", lambda: a @ wt.t())):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 20 * 1e3
            print(f"{nm} {dt} {lay} M={M} N={N} K={K}: {ms*1e3:.1f} us {2*M*N*K/ms/1e9:.1f} TF/s")
