"""bench.py — headline benchmark of the HyperVLA action-prediction path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch 256] [--enc-dtype f16|bf16]
    (N > 1 works as typed: without WORLD_SIZE in the environment the script starts its own ranks with
     `python -m torch.distributed.run --nproc-per-node N ...` as a CHILD process before anything touches the GPU;
     under a launcher that already set RANK / WORLD_SIZE it is one of the ranks.)

One "step" = one `sample_actions` over one batch of synthetic OXE-shaped observations: uint8 224x224
images -> DINOv2-base (in the loop, as the reference runs it, base_vit.py:109-133) -> generated vit_t
policy -> [4, 7] action chunks.  1 action = 1 sample-step.  The hypernetwork (`create_tasks`) runs once
per episode batch before the timed region (BASELINE.json configs[1]: "weight-gen once + vit_t inference,
batch 256"); images are resident in HBM when timing starts.  Episodes shard across ranks with no
data-path collective => weak scaling, value = sum of per-rank actions / max rank time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "hyper-vla_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402
import torch         # noqa: E402

PMC_ROUND = "r6"        # profiles/<round>_pmc_*: the committed rocprofv3 --pmc passes `traffic` is read from
PEAK_TFLOPS = 2500.0   # dense bf16/fp16 MFMA, MI355X_MICROARCH.md


def fc1_main_rows(B, g):
    """Rows the timed gemm256p_kernel launch of fc1 covers: csrc/encoder.hip tiles the batch image by image (tile row b =
    the 256 patch rows of image b; the B CLS rows run in gemm64_kernel right behind it) when a batch has more than 2047
    rows; below that the whole GEMM is one gemm64_kernel launch."""
    M = B * (g.patches + 1)
    if g.patches == 256 and M > 2047 and g.enc_mlp % 256 == 0:
        return B * g.patches
    return M


def big_gemm_symbols(B, g, enc_dtype, ncu=256):
    """{category: (kernel symbol as rocprofv3 prints it, FLOPs of one launch)} for the four dense products of an encoder layer at a
    batch whose images are tiled one 256-row tile each -- run_encoder's choices (csrc/encoder.hip, csrc/plan.h) restated, so that the
    bench line can NAME the instantiation whose launches it timed and look its counters up in the committed PMC passes.  None at
    the small batches (rows <= 2047: gemm64c kernels, no single dominant symbol worth a roofline)."""
    S, E, F = g.patches + 1, g.enc_dim, g.enc_mlp
    M = B * S
    if not (g.patches == 256 and M > 2047 and E % 64 == 0 and F % 64 == 0 and E >= 256):
        return None
    op = "hvla::OpF16" if enc_dtype == "f16" else "hvla::OpBF16"
    rows = B * g.patches
    big = lambda n, el: M * n * el >= (96 << 20)

    def lnx_persistent(nbm, nbn):
        nt = nbm * nbn
        if ncu % 8 or nt % ncu or nbm % 8:
            return False
        wx = ncu // 8
        return wx >= nbn and (wx % nbn == 0 or (nt // ncu) % nbn == 0)

    def sym(epi, n, el, lnx):
        nbn = (n + 255) // 256
        if lnx:
            pers, nt = lnx_persistent(B, nbn), big(n, 4)
        else:
            pers, nt = (B * nbn) % ncu == 0, big(n, el) and (epi == 3 or M * n * el < (1 << 32))
        t = lambda b: "true" if b else "false"
        return f"void hvla::gemm256p_kernel<{op}, {epi}, {t(pers)}, {t(lnx)}, {t(nt)}>(hvla::GemmArgs)"

    lnx = M * E * 4 < (1 << 32) and E <= 1024
    return {"qkv_gemm": (sym(1, 3 * E, 2, False), 2.0 * rows * E * 3 * E),
            "out_gemm": (sym(3, E, 4, lnx), 2.0 * rows * E * E),
            "fc1_gemm": (sym(2, F, 2, False), 2.0 * rows * E * F),
            "fc2_gemm": (sym(3, E, 4, lnx), 2.0 * rows * F * E)}


def pmc_rows(symbol, rnd):
    """{counter: mean per launch} of one kernel symbol from the committed rocprofv3 --pmc passes of this command (profiles/<round>_pmc_*.csv,
    written by tools/collect_profiles.sh + tools/pmc_summary.py, which cuts the name at 90 characters)."""
    import csv
    out = {}
    for nm in ("fetch_size", "write_size", "sq", "tcc"):
        try:
            for row in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_{nm}_by_kernel.csv"))):
                if row["kernel"] == symbol[:90]:
                    out[row["counter"]] = float(row["mean"])
        except Exception:
            pass
    return out


def box_probe(ctx, dev):
    """What THIS box sustains, measured in this process right before the timed region (the boxes of a pool differ by several
    per cent, and the driver's one number per round lands on one of them): shader clock under a chip-wide MFMA loop and that
    loop's TFLOP/s (hvla_box_probe: a probe kernel of its own in libhvla), and what a 1 GiB device-to-device copy moves."""
    clock_mhz, mfma_tflops, probe_ms = ctx.box_probe(torch.cuda.current_stream(dev).cuda_stream)
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(dev))
    for _ in range(4):
        dst.copy_(src)
    e1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    copy_tbps = 4 * 2 * float(1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e12
    del src, dst
    return {"shader_clock_mhz_under_mfma": round(clock_mhz, 1), "mfma_probe_tflops": round(mfma_tflops, 1),
            "mfma_probe_frac_of_peak": round(mfma_tflops / PEAK_TFLOPS, 4), "mfma_probe_ms": round(probe_ms, 3),
            "copy_1gib_tbps": round(copy_tbps, 3),
            "note": "probe kernels of their own, run in this process before the timed region: a step time is read against these"}


def algorithmic_flops(g):
    """Per sample-step, counted as the reference executes them (SURVEY.md §8d / BASELINE.md §2)."""
    S, E, F, P, D, M, L = g.seq, g.enc_dim, g.enc_mlp, g.patches, g.dim, g.mlp, g.layers
    enc_linear = g.enc_layers * 2 * S * (4 * E * E + 2 * E * F)
    enc_attn = g.enc_layers * 4 * S * S * E
    patch = 2 * P * g.patch_in * E
    pol = 2 * P * E * D + L * 2 * S * (4 * D * D + 2 * D * M) + L * 4 * S * S * D + 2 * D * (g.horizon * g.action_dim)
    return dict(encoder=enc_linear + enc_attn + patch, policy=pol, fc1=2 * S * E * F)


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(g, params, seconds=12.0, batch=8):
    """The float32 torch-CPU restatement of the same step (oracle/hvla_ref_torch.py, "port"; JAX itself
    is not installable here), timed on this host's cores on a bounded sample."""
    from hypervla import synthetic as syn
    from hypervla.config import encoder_leaves, generated_leaves
    from oracle import hvla_ref_torch as ot
    cores = usable_cores()
    torch.set_num_threads(cores)
    ref = ot.FullRef(params, g, generated_leaves(g), dict(encoder_leaves(g)), torch.float32)
    ins, st, im = syn.synthetic_instructions(batch, g), syn.synthetic_initial_state(batch, g), syn.synthetic_images(batch, g)
    theta, _ = ref.create_tasks(ins, st)
    ref.sample_actions(theta, im)                       # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        ref.sample_actions(theta, im)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and n >= 2:
            break
    return {"value": round(batch * n / dt, 3), "unit": "actions/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of batch {batch} (same workload, f32 torch-CPU restatement of the JAX graph, "
                      f"{dt:.1f} s)"}


def finetune_bench(a, model, rank, world, use_dist):
    """Config 5: one optimiser step per iteration on a per-GPU batch; `value` = samples/s over all ranks."""
    from hypervla import synthetic as syn
    from hypervla.dp import max_over_ranks, whole_job_rate
    from hypervla.train import FineTuner
    import torch.distributed as dist
    g, B, dev = model.geometry, a.batch, model.device
    ft = FineTuner(model, B, train_encoder=a.train_encoder)
    ins, st = syn.synthetic_instructions(B, g, rank), syn.synthetic_initial_state(B, g, rank)
    images = torch.as_tensor(syn.synthetic_images(B, g, rank)[:, 0]).to(dev).contiguous()
    batch = syn.synthetic_action_batch(B, g, rank)
    losses = []
    for _ in range(a.warmup):
        losses.append(float(ft.step(ins, st, images, batch)))
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = ft.step(ins, st, images, batch)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = max_over_ranks(time.perf_counter() - t0, dev)
    # the step's dominant kernel family, timed live on two more steps (HIP events on the launch stream around every batched GEMM
    # launch: hvla_train_profile): csrc/train.hip bgemm3_kernel, the split-bf16 GEMM of every forward / dX / dW product
    model._ctx.train_profile(True)
    for _ in range(2):
        ft.step(ins, st, images, batch)
    torch.cuda.synchronize(dev)
    gemm_ms, gemm_flops, gemm_n = model._ctx.train_profile_read()
    model._ctx.train_profile(False)
    f32_tflops = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    roofline = {"bound": "mfma", "kernel": "bgemm3_kernel (csrc/train.hip): every dense product of the step, f32 operands split hi + lo "
                                           "while staged, THREE v_mfma_f32_32x32x16_bf16 per product (a_hi b_hi + a_hi b_lo + a_lo b_hi)",
                "achieved": round(f32_tflops, 2), "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(f32_tflops / PEAK_TFLOPS, 4),
                "matrix_instruction_tflops": round(3.0 * f32_tflops, 2), "matrix_instruction_frac": round(3.0 * f32_tflops / PEAK_TFLOPS, 4),
                "mfma_per_product": 3, "traffic": None,
                "gemm_ms_per_step": round(gemm_ms / 2, 3), "gemm_launches_per_step": gemm_n // 2,
                "gemm_flops_per_step_f32_equivalent": gemm_flops / 2,
                "note": "achieved / frac = USEFUL work (f32-equivalent 2MNK) / summed duration of the GEMM launches against the bf16 peak; "
                        "the matrix cores execute three instructions per product (matrix_instruction_*); "
                        "the rest of the step is element-wise kernels and the optimiser"}
    if rank == 0:
        print(json.dumps({
            "roofline": roofline,
            "metric": "finetune_samples_per_sec", "value": round(whole_job_rate(B, world, a.steps, elapsed), 2),
            "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if a.train_encoder else "f32 (policy/hypernet fwd+bwd) + " + a.enc_dtype + " (frozen encoder)",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[4]: fine-tune step, DINOv2 encoder trained (README.md:55): "
                                    "fwd + bwd through encoder, policy and hypernetwork + grad all-reduce + "
                                    "clip/AdamW(bf16 mu, two groups)/EMA") if a.train_encoder else
                                   ("BASELINE configs[4] variant: fine-tune step with the image encoder frozen "
                                    "(encode + fwd + bwd + grad all-reduce + clip/AdamW(bf16 mu)/EMA), hypernetwork "
                                    "parameters only"), "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"dp{world}, RCCL all-reduce of {ft.n} f32 gradients"},
            "loss_first": round(losses[0], 5) if losses else None, "loss_last": round(float(last), 5)}))
    if use_dist:
        dist.destroy_process_group()


def kfd_gpu_count(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process would see, counted WITHOUT the HIP runtime (torch.cuda.device_count() falls back to
    hipGetDeviceCount() when amdsmi is unavailable, which initialises HSA in the launcher): KFD topology nodes with SIMDs,
    cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  None if sysfs is not readable (the
    ranks then fail with their own message)."""
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except Exception:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n):
    """`python bench.py --gpus N` typed directly (no launcher): start one rank per GPU as children of this process,
    which has not touched the GPU (no HIP call yet; a process that has must never exec or fork GPU work), let rank 0's
    JSON line through on the inherited stdout and return the launcher's exit code."""
    import socket
    import subprocess
    if os.environ.get("HVLA_BENCH_SHARE_GPU") != "1":
        have = kfd_gpu_count()                       # sysfs only: no HIP / HSA call in this process
        if have is not None and have < n:
            print(f"bench.py: --gpus {n} but this node exposes {have} GPU(s)", file=sys.stderr)
            return 2
    with socket.socket() as s:                       # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this image
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="episodes per GPU")
    ap.add_argument("--enc-dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--encoder", default="base", choices=["base", "small"],
                    help="DINOv2-base (README / reference parity, default) or DINOv2-small (E=384, BASELINE configs[1] wording)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (BASELINE config 3)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="2: hvla_step runs the two halves of the batch on two streams (more actions/s; per-kernel "
                         "durations then include time shared with the other half, so the roofline line is not comparable)")
    ap.add_argument("--latency-samples", type=int, default=0,
                    help="per-step latency samples for p50 (default: >= 100; the profiler passes of tools/collect_profiles.sh ask for fewer)")
    ap.add_argument("--ensemble", action="store_true",
                    help="include the device-side un-normalise + temporal ensemble in every step (always on with --graph)")
    ap.add_argument("--finetune", action="store_true",
                    help="BASELINE config 5: encode + fwd + bwd + RCCL grad all-reduce + AdamW + EMA (encoder frozen "
                         "unless --train-encoder)")
    ap.add_argument("--train-encoder", action="store_true",
                    help="with --finetune: fine_tune_pretrained_image_encoder=True (README.md:55)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus))
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    # test hook only (tests/test_gpu_parity.py): all ranks on GPU 0 over gloo, so that the N > 1 control flow can be run on
    # a one-GPU box (RCCL refuses two ranks on one device); never set by the driver
    share = os.environ.get("HVLA_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ      # under torch.distributed.run even at N = 1
    if use_dist:
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))   # RCCL; used only for barrier + max(time)

    from hypervla import synthetic as syn
    from hypervla.config import FULL, SMALL_E
    from hypervla.dp import max_over_ranks, whole_job_rate
    from hypervla.model import HyperVLA
    g, B = (SMALL_E if a.encoder == "small" else FULL), a.batch
    model = HyperVLA.from_synthetic(g, device=local, max_batch=B, enc_dtype=a.enc_dtype, streams=a.streams)
    dev = model.device
    if a.finetune:
        return finetune_bench(a, model, rank, world, use_dist)
    ins, st = syn.synthetic_instructions(B, g, rank), syn.synthetic_initial_state(B, g, rank)
    images = torch.as_tensor(syn.synthetic_images(B, g, rank)[:, 0]).to(dev).contiguous()   # resident in HBM
    ctx = model._ctx

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    t0 = time.perf_counter()
    w, tasks, _ = model.create_tasks(instruction_dict=ins, initial_state=st)      # once per episode batch
    torch.cuda.synchronize(dev)
    generate_ms = (time.perf_counter() - t0) * 1e3
    actions = torch.empty(B, g.horizon, g.action_dim, device=dev)
    logits = torch.empty(B, g.horizon, device=dev)
    stream = model._stream()

    ens = None
    if a.graph or a.ensemble:          # BASELINE configs[2]: the device-side un-normalise + temporal ensemble is
        stats = syn.synthetic_dataset_statistics(g)["bridge_dataset"]["action"]     # part of the captured step
        ens = (torch.as_tensor(stats["mean"]).to(dev), torch.as_tensor(stats["std"]).to(dev),
               torch.as_tensor(stats["mask"].astype(np.uint8)).to(dev), torch.empty(B, g.action_dim, device=dev))
        ctx.ensemble_reset(w._h, stream)

    def step():
        ctx.step(w._h, images.data_ptr(), actions.data_ptr(), logits.data_ptr(), B, stream)
        if ens is not None:
            ctx.ensemble(w._h, actions.data_ptr(), ens[0].data_ptr(), ens[1].data_ptr(), ens[2].data_ptr(),
                         ens[3].data_ptr(), stream)

    box = box_probe(ctx, dev) if rank == 0 else None
    for _ in range(a.warmup):
        step()
    ctx.launches()
    step()
    launches = ctx.launches()         # what the library enqueued for one eager step (hvla_launches)
    eager_step = step
    if a.graph:                        # capture one step (launches only, no allocation / sync inside hvla_step)
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(dev)
        with torch.cuda.stream(side):
            stream = model._stream()
            step()                     # warm the side stream
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                stream = model._stream()
                step()
        stream = model._stream()
        step = graph.replay
        step()
    # ---- which kernel symbol is the dominant one?  An un-timed pass with events around every category (hvla_profile mode 2), then the
    # categories of the symbol with the largest share are the ones whose launches carry events inside the timed region (mode 1).
    parts = 2 if (a.streams == 2 and B >= 64) else 1       # episodes per launch = B / parts
    symbols = big_gemm_symbols(B // parts, g, a.enc_dtype)
    dom_cats = ["fc1_gemm"]
    if symbols is not None:
        ctx.profile(2)
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize(dev)
        pre = ctx.profile_read()
        share = {}
        for cat, (name, _) in symbols.items():
            share[name] = share.get(name, 0.0) + pre[cat][0]
        dom_symbol = max(share, key=share.get)
        dom_cats = [c for c, (name, _) in symbols.items() if name == dom_symbol]
    ctx.profile_select(dom_cats)
    ctx.profile(0 if a.graph else 1)   # HIP events around the dominant kernel's launches only, on the launch stream
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if a.graph:                        # events cannot be recorded inside a replayed graph: time the dominant
        ctx.profile(1)                 # kernel on eager launches of the same step right after the timed region
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize(dev)
    dom = ctx.profile_read()
    ctx.profile(0)
    elapsed = max_over_ranks(elapsed, dev)
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- per-step latency distribution + full kernel breakdown (separate, un-timed passes)
    lat = []
    # p50 over >= 200 samples for a replayed graph (config 3) and for the small batches whose metric IS the latency (B <= 8),
    # over >= 100 at the headline batch (30 for the 60-120 ms steps of 1024 / 2048 episodes)
    n_lat = max(a.steps, 200) if (a.graph or B <= 8) else max(a.steps, 100 if B <= 512 else 30)
    if a.latency_samples > 0:
        n_lat = a.latency_samples
    for _ in range(n_lat):
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize(dev)
        lat.append((time.perf_counter() - t1) * 1e3)
    ctx.profile(2)
    for _ in range(3):
        eager_step()
    torch.cuda.synchronize(dev)
    br = ctx.profile_read()
    ctx.profile(0)
    breakdown = {k: round(v[0] / 3, 4) for k, v in br.items()}

    # policy-only variant (patch tokens resident -> actions), BASELINE config 2 "(P)"
    tokens = model.encode_images(images)
    torch.cuda.synchronize(dev)
    for _ in range(5):
        ctx.policy(w._h, tokens.data_ptr(), actions.data_ptr(), logits.data_ptr(), B, stream)
    torch.cuda.synchronize(dev)
    # 100 back-to-back launches between two events on the launch stream (torch's current stream = model._stream()): a wall
    # clock around 20 launches, as rounds 1-3 had it, adds the host's synchronisation to a 0.2 ms kernel (0.237 against 0.213 ms)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(dev))
    for _ in range(100):
        ctx.policy(w._h, tokens.data_ptr(), actions.data_ptr(), logits.data_ptr(), B, stream)
    e1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    pol_ms = e0.elapsed_time(e1) / 100

    fl = algorithmic_flops(g)
    rows = (B // parts) * (g.patches + 1)
    main_rows = fc1_main_rows(B // parts, g)
    if symbols is None:                 # small batch: the fc1 launch (gemm64c_kernel / gemm64_kernel), as rounds 1-5 reported it
        dom_ms_total, dom_n = dom["fc1_gemm"]
        dom_flops_total = 2.0 * main_rows * g.enc_dim * g.enc_mlp * dom_n
        kernel_name = f"gemm64c_kernel<Op,EPI_GELU> (encoder fc1: [B*257,{g.enc_dim}]x[{g.enc_dim},{g.enc_mlp}] + bias + erf-GELU)"
        dom_symbol = None
    else:
        dom_ms_total = sum(dom[c][0] for c in dom_cats)
        dom_n = sum(dom[c][1] for c in dom_cats)
        dom_flops_total = sum(symbols[c][1] * dom[c][1] for c in dom_cats)
        what = {"qkv_gemm": "QKV", "out_gemm": "attention out-projection + residual + fused LayerNorm", "fc1_gemm": "fc1 + erf-GELU",
                "fc2_gemm": "fc2 + residual + fused LayerNorm"}
        kernel_name = dom_symbol + " = " + " and ".join(what[c] for c in dom_cats) + " of every encoder layer"
    dom_ms = dom_ms_total / max(dom_n, 1)                     # average launch of the dominant symbol inside the timed region
    flops_per_launch = dom_flops_total / max(dom_n, 1)
    achieved = flops_per_launch / (dom_ms * 1e-3) / 1e12
    # per shape, from the un-timed all-category pass right behind the timed region (3 eager steps; events around every launch)
    by_shape = None
    if symbols is not None:
        by_shape = {}
        for c, (name, f) in symbols.items():
            ms_c, n_c = br[c]
            if n_c:
                by_shape[c] = {"frac": round(f / (ms_c / n_c * 1e-3) / 1e12 / PEAK_TFLOPS, 4), "launch_ms": round(ms_c / n_c, 4), "kernel": name}
    # counters of the dominant symbol from the committed rocprofv3 --pmc passes of this same command (NOT measured in this run):
    # HBM-side bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts half of a wide streaming read), MFMA-busy =
    # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)
    traffic = mfma_busy = hbm_tbps = None
    headline_cfg = B == 256 and a.enc_dtype == "f16" and a.encoder == "base" and a.streams == 1
    if dom_symbol is not None and headline_cfg:
        pm = pmc_rows(dom_symbol, PMC_ROUND)
        if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
            traffic = int((2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0)
            hbm_tbps = round(traffic / (dom_ms * 1e-3) / 1e12, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in pm and pm.get("GRBM_GUI_ACTIVE", 0) > 0:
            mfma_busy = round(pm["SQ_VALU_MFMA_BUSY_CYCLES"] / (pm["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    ms_per_step = elapsed / a.steps * 1e3
    out = {
        "metric": "actions_per_sec", "value": round(whole_job_rate(B, world, a.steps, elapsed), 2), "unit": "actions/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.enc_dtype, "data": "synthetic",
        "config": {"workload": ("BASELINE configs[2] (hipGraph-captured step incl. device-side ensemble): " if a.graph else
                                "BASELINE configs[1]: ") + "hypernet weight-gen once (untimed) + full sample_actions step "
                               f"(u8 224x224 -> DINOv2-{a.encoder} E={g.enc_dim} in the loop -> generated vit_t 4L/64d policy -> "
                               "[4,7] action chunk); 1 action = 1 sample-step",
                   "batch_per_gpu": B, "global_batch": world * B, "encoder": "DINOv2-base (reference parity, E=768)" if a.encoder == "base" else "DINOv2-small (E=384)",
                   "parallelism": f"episode-dp{world} (no collectives)",
                   "encoder_operands": a.enc_dtype + " (+ per-image first-order compensation of the weight rounding, "
                                       "kernel_ms_per_step.small_row_gemms)",
                   "policy_operands": "split-bf16 (bf16x3)",
                   "launch": "hipGraph replay" if a.graph else f"eager ({launches} launches per step, counted by the library)",
                   "streams": a.streams,
                   "ensemble": "device-side un-normalise + temporal ensemble (history = horizon) inside the step"
                               if ens is not None else "not in the step"},
        "box": box,
        "latency_samples": len(lat),
        "p50_step_latency_ms": round(float(np.median(lat)), 4),
        "roofline": {"bound": "mfma", "kernel": kernel_name,
                     "achieved": round(achieved, 2), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_TFLOPS, 4), "traffic": traffic,
                     "traffic_source": (f"committed rocprofv3 --pmc passes of this command (profiles/{PMC_ROUND}_pmc_*_by_kernel.csv); "
                                        "NOT measured in this run") if traffic is not None else None,
                     "traffic_unit": "bytes per average launch (HBM-side, PMC: 2*FETCH_SIZE+WRITE_SIZE)",
                     "mfma_busy": mfma_busy, "hbm_tbps": hbm_tbps, "hbm_frac_of_8tbps": round(hbm_tbps / 8.0, 4) if hbm_tbps else None,
                     "chosen_by": "largest share of the step among the kernel symbols of an un-timed all-category pass (hvla_profile mode 2) "
                                  "right before the timed region; its launches then carry HIP events inside the timed region",
                     "launch_ms": round(dom_ms, 4), "launches_timed": dom_n,
                     "flops_per_launch": flops_per_launch, "rows_per_launch": main_rows, "rows_total": rows,
                     "by_shape": by_shape},
        "step_tflops": round((fl["encoder"] + fl["policy"]) * B / (ms_per_step * 1e-3) / 1e12, 2),
        "step_frac_of_peak": round((fl["encoder"] + fl["policy"]) * B / (ms_per_step * 1e-3) / 1e12 / PEAK_TFLOPS, 4),
        "kernel_ms_per_step": breakdown,
        "policy_only": {"ms_per_step": round(pol_ms, 4), "actions_per_sec": round(B / (pol_ms * 1e-3), 1),
                        "input": "f32 patch tokens [B,256,768] resident in HBM"},
        "generate_ms": round(generate_ms, 3),
    }
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(g, model.params)
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
