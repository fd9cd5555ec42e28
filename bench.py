#!/usr/bin/env python3
"""Headline benchmark: denoiser train-step images/sec at 256x256 (latent 32x32), bf16, batch 32 per GPU.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torchrun environment: this process starts the N ranks itself (`python -m torch.distributed.run
--nnodes=1 --nproc-per-node N ... bench.py <same arguments>`, one process per GPU over RCCL) before it touches a GPU, and
exits with their status; under torchrun (RANK / WORLD_SIZE set) it is one of the ranks.  Fewer visible GPUs than N, or a
WORLD_SIZE that is not N, is an error - never a silent one-GPU measurement.

One step = DDPM add-noise -> full SD-v1-size UNet forward (text cross-attention, T=77) -> fused MSE ->
backward (bucketed RCCL gradient all-reduce overlapped on a side stream when N > 1) -> global-norm clip
+ fused AdamW.  Data: synthetic latents / text states resident in HBM, random-init weights (no network).
Prints ONE JSON line on rank 0 with `roofline` (contraction kernels vs the dense bf16 MFMA peak,
measured with HIP events on the launch stream), and at N = 1 also `cpu_baseline` (the fp32 CPU oracle of the same
step on a bounded sample), `sampler` (the 512x512 sampling loop: 16 images, classifier-free guidance, 50 DDIM steps,
one HIP graph per step shape), `lora` (the LoRA fine-tune step at the headline shape) and `e2e` (the train step from
pixels and strings: auto-encoder, tokenizer and text encoder inside the step; never part of `value`).
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0       # dense bf16, MI355X
FWD_GFLOP_PER_SAMPLE_256 = 180.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--latent", type=int, default=32, help="latent side (32 = 256x256 images)")
    ap.add_argument("--ctx-len", type=int, default=77)
    ap.add_argument("--prof-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32, help="batch of the CPU oracle step (32 = the GPU step's own batch, ~15 s on 16 cores)")
    ap.add_argument("--small", action="store_true", help="plumbing-size model (NOT the headline config)")
    ap.add_argument("--no-graph", action="store_true", help="run the eager step even on one GPU")
    ap.add_argument("--no-extra", action="store_true", help="skip the sampler and LoRA legs of the N = 1 line")
    ap.add_argument("--sampler-steps", type=int, default=50)
    ap.add_argument("--lora-steps", type=int, default=20)
    ap.add_argument("--e2e-steps", type=int, default=10)
    ap.add_argument("--lora", action="store_true",
                    help="LoRA fine-tune step instead of the headline full fine-tune: frozen base, rank-4 adapters on "
                         "attn{1,2}.to_{q,k,v,out} (no cpu_baseline leg)")
    return ap.parse_args()


def source_fingerprint():
    """sha256 over the kernel sources and the modules that choose the launches: a committed counter profile applies to a
    run only if it was collected from the same code."""
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "image_generator_amd")
    files = sorted(os.path.join(pkg, "csrc", f) for f in os.listdir(os.path.join(pkg, "csrc"))
                   if f.endswith((".hip", ".h", ".cpp")) and f != "lmdb_reader.cpp")      # the data-file reader launches nothing
    files += [os.path.join(pkg, f) for f in ("ops.py", "unet.py", "trainer.py")]
    for f in files:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def visible_gpu_count():
    """GPUs a rank of this benchmark would see, counted WITHOUT loading a GPU runtime into this process: a throw-away
    child asks torch (which honours HIP_/ROCR_/CUDA_VISIBLE_DEVICES and whatever devices the container was given - the
    kfd topology in /sys lists the host's GPUs, not the container's).  None when the child could not tell."""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except (subprocess.SubprocessError, ValueError, IndexError, OSError):
        return None


def launch_ranks(n, argv, entry=None, need_devices=None):
    """Start n ranks of this benchmark (one process per GPU) and return their exit status.  The launching process does
    nothing GPU-side, before or after: it counts devices through a child (`visible_gpu_count`), refuses with status 2
    when there are fewer than `need_devices` (default n: one per rank), and otherwise waits for the ranks."""
    need = n if need_devices is None else need_devices
    have = visible_gpu_count()
    if have is not None and have < need:
        print(f"[bench] --gpus {n} requested but only {have} GPU(s) are visible; not measuring fewer", file=sys.stderr)
        return 2
    if have is None:            # the ranks check for themselves (init_distributed refuses a LOCAL_RANK without a device)
        print("[bench] could not count the visible GPUs; every rank will check its own device", file=sys.stderr)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC for RCCL between processes
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), entry or os.path.abspath(__file__), *argv]
    print(f"[bench] {have if have is not None else '?'} GPU(s) visible; starting {n} rank(s): {' '.join(cmd[1:8])} ...", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def sampler_leg(steps, images=16, side=64):
    """512x512 sampling loop (image_generator.py:73-101 at batch 16): one step = a batch-32 denoiser evaluation +
    guidance + DDIM update, the whole loop timed (eager first step, graph capture and replays included)."""
    from image_generator_amd.generator import ImageGenerator
    torch.manual_seed(0)
    gen = ImageGenerator("cuda:0", text_encoder_kwargs=dict(num_layers=2))      # the text tower is outside this loop
    gen.eval()
    emb = torch.randn(2 * images, 77, 768, device="cuda:0").to(torch.bfloat16)
    lat = torch.randn(images, 4, side, side, device="cuda:0")
    gen.sample_latents(emb, lat, 3, 7.5, use_graph=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = gen.sample_latents(emb, lat, steps, 7.5, use_graph=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if not bool(torch.isfinite(out).all()):
        raise RuntimeError("sampler leg: non-finite latents")
    flops = 2 * images * 803.9e9 * (side / 64.0) ** 2
    return {"steps_per_s": steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "images": images,
            "resolution": side * 8, "guidance": 7.5, "hip_graph": True, "mfma_frac": flops * steps / dt / (MFMA_PEAK_TFLOPS * 1e12)}


def lora_leg(steps, B=32, L=32):
    """LoRA fine-tune step (frozen base, rank-4 adapters on attn{1,2}.to_{q,k,v,out}) at the headline shape."""
    from image_generator_amd.generator import inject_lora
    from image_generator_amd.trainer import DenoiserTrainer
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    torch.manual_seed(1234)
    with torch.device("cuda:0"):
        unet = UNet2DConditionModel(DenoiserConfig.sd_v1())
    adapters = inject_lora(unet, r=4, alpha=1.0)
    torch.manual_seed(4321)
    with torch.no_grad():
        for ad in adapters:                      # B = 0 at init would leave half of the arithmetic multiplying zeros
            ad.lora_B.weight.normal_(0, 0.02)
    tr = DenoiserTrainer(unet, lr=1e-4, weight_decay=1e-2, max_grad_norm=1.0)
    lat, noise = torch.randn(B, 4, L, L, device="cuda:0"), torch.randn(B, 4, L, L, device="cuda:0")
    ts = torch.randint(0, 1000, (B,), device="cuda:0")
    ctx = torch.randn(B, 77, 768, device="cuda:0").to(torch.bfloat16)
    tr.capture(lat, noise, ts, ctx)
    for _ in range(3):
        tr.train_step(lat, noise, ts, ctx)
    torch.cuda.synchronize()
    probe = tr.opt.flat[:4096].clone()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.train_step(lat, noise, ts, ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gn = tr.opt.grad_norm()
    if not (math.isfinite(float(loss)) and math.isfinite(gn) and gn > 0) or torch.equal(probe, tr.opt.flat[:4096]):
        raise RuntimeError("lora leg: optimizer skipped or diverged")
    return {"images_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "batch": B,
            "adapters": len(adapters), "trainable_params": int(tr.opt.flat.numel()), "hip_graph": True,
            "mfma_frac": 2 * FWD_GFLOP_PER_SAMPLE_256 * 1e9 * B * steps / dt / (MFMA_PEAK_TFLOPS * 1e12)}


def e2e_leg(steps, B=32, side=256):
    """The reference's whole loop body from pixels and strings (trainer.py:243-261 around image_generator.py:103-126):
    `ImageGenerator.train_step(images, captions)` - frozen auto-encoder, tokenizer, frozen text encoder, noise and
    timesteps drawn inside - then backward, clip and AdamW.  Eager step (captions of one length so that the shapes stay
    static); the headline `value` excludes the auto-encoder and the text tower, this figure shows what they add."""
    from image_generator_amd.generator import ImageGenerator
    from image_generator_amd.trainer import DenoiserTrainer
    torch.manual_seed(1234)
    gen = ImageGenerator("cuda:0")
    gen.train()
    tr = DenoiserTrainer(gen.unet, lr=1e-4, weight_decay=1e-2, max_grad_norm=1.0)
    images = torch.rand(B, 3, side, side, device="cuda:0") * 2 - 1
    tags = ["1girl", "solo", "long_hair", "blue_sky", "outdoors", "smile", "school_uniform", "looking_at_viewer", "tree", "cloud"]
    captions = [" ".join(tags[(i + j) % len(tags)] for j in range(10)) for i in range(B)]
    step = lambda: tr.train_step_from(lambda: gen.train_step(images, captions, cond_dropout_prob=0.0))
    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    probe = tr.opt.flat[:4096].clone()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the parts beside the denoiser step, timed alone on the same inputs
    def timed(fn, n=5):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    with torch.no_grad():
        vae_ms = timed(lambda: gen.vae.encode(images).latent_dist.sample())
    text_ms = timed(lambda: gen.encode_text(captions, padding="longest"))
    gn = tr.opt.grad_norm()
    if not (math.isfinite(float(loss)) and math.isfinite(gn) and gn > 0) or torch.equal(probe, tr.opt.flat[:4096]):
        raise RuntimeError("e2e leg: optimizer skipped or diverged")
    return {"images_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "batch": B, "resolution": side,
            "hip_graph": False, "vae_encode_ms": vae_ms, "tokenizer_text_encoder_ms": text_ms,
            "includes": "ImageGenerator.train_step(images, captions): VAE encode + tokenizer + TextEncoder (8 layers, 768 wide) + "
                        "noise / timesteps + denoiser forward; backward; clip + AdamW"}


def cpu_baseline(args):
    """fp32 CPU oracle (oracle/denoiser_ref.py), same architecture and step, bounded sample."""
    from oracle.denoiser_ref import DenoiserRef, DenoiserConfig, mse_objective
    from oracle import ops_ref
    torch.manual_seed(0)
    cores = torch.get_num_threads()
    cfg = DenoiserConfig.small() if args.small else DenoiserConfig.sd_v1()
    model = DenoiserRef(cfg)
    B, L = args.cpu_batch, args.latent
    lat, noise = torch.randn(B, 4, L, L), torch.randn(B, 4, L, L)
    t = torch.randint(0, 1000, (B,))
    ctx = torch.randn(B, args.ctx_len, cfg.cross_attention_dim)
    acp = ops_ref.sd_alphas_cumprod()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-2)
    t0 = time.perf_counter()
    noisy = ops_ref.ddpm_add_noise(lat, noise, t, acp)
    loss = mse_objective(model(noisy, t, ctx).sample, noise)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    dt = time.perf_counter() - t0
    return {"value": B / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"1 full train step (fwd+loss+bwd+clip+AdamW), batch {B}, fp32 torch CPU oracle, {dt:.1f} s"}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))        # nothing has touched a GPU in this process yet
    import image_generator_amd.trainer as trainer_mod
    from image_generator_amd.trainer import DenoiserTrainer
    init_distributed = trainer_mod.init_distributed       # (looked up late: the one-GPU rehearsal substitutes its own)
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    from image_generator_amd._lib import lib
    import torch.distributed as dist
    import ctypes

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={env_world}: refusing to report a line for another world size",
              file=sys.stderr)
        sys.exit(2)
    try:
        rank, world, local = init_distributed()
    except trainer_mod.NoDeviceForRank as e:    # a rank without a GPU of its own never measures on somebody else's
        print(f"[bench] {e}", file=sys.stderr)
        sys.exit(2)
    if world > 1 and dist.get_world_size() != args.gpus:
        raise RuntimeError(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    dev = torch.device("cuda", local)
    torch.manual_seed(1234)           # identical initial weights on every rank
    cfg = DenoiserConfig.small() if args.small else DenoiserConfig.sd_v1()
    with torch.device(dev):
        unet = UNet2DConditionModel(cfg)
    if args.lora:
        from image_generator_amd.generator import inject_lora
        adapters = inject_lora(unet, r=4, alpha=1.0)
        torch.manual_seed(4321)
        with torch.no_grad():               # the zero-initialised up-projections would make half the arithmetic trivial
            for ad in adapters:
                ad.lora_B.weight.normal_(0, 0.02)
    trainer = DenoiserTrainer(unet, lr=1e-4, weight_decay=1e-2, max_grad_norm=1.0)
    B, L = args.batch, args.latent
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    lat = torch.randn(B, 4, L, L, device=dev, generator=g) * 0.18215 * 5.0
    noise = torch.randn(B, 4, L, L, device=dev, generator=g)
    ctx = torch.randn(B, args.ctx_len, cfg.cross_attention_dim, device=dev, generator=g).to(torch.bfloat16)
    ts = torch.randint(0, 1000, (B,), device=dev, generator=g)

    graphed = world == 1 and not args.no_graph and trainer.reducer is None
    if graphed:
        trainer.capture(lat, noise, ts, ctx)     # one HIP graph for forward + backward + optimizer

    def step(i):
        return trainer.train_step(lat, noise, ts, ctx, epoch_float=1.0 + i * 1e-3)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # multi-GPU (eager) step: the forward pass replayed as a HIP graph from the second warm-up step on, so that the host
    # enqueues only the backward pass and the optimizer per step; the collectives are never captured
    fwd_graph = False
    want_fwd_graph = (not graphed and trainer.reducer is not None and not args.no_graph
                      and os.environ.get("MI355_DP_FWD_GRAPH") != "0")

    def capture_forward():
        try:
            trainer.capture_forward(lat, noise, ts, ctx)
            return True
        except Exception as e:              # the eager step measures the same arithmetic
            trainer._fwd_graph = trainer._fwd_loss = None
            print(f"[bench] forward graph not used: {type(e).__name__}: {e}", file=sys.stderr)
            return False

    for i in range(args.warmup):
        if i == 1 and want_fwd_graph:
            fwd_graph = capture_forward()
        loss = step(i)
    if want_fwd_graph and args.warmup < 2:  # fewer than two warm-up steps asked for: capture behind one eager step of
        if args.warmup == 0:                # our own, and replay once untimed so that no timed step is a first replay
            step(0)
        fwd_graph = capture_forward()
        step(1)
    barrier()
    probe = trainer.opt.flat[:65536].clone()      # to prove afterwards that the timed steps really updated the weights
    if trainer.reducer is not None:
        trainer.reducer.exposed = []              # (before, after) events around the compute stream's wait for the collectives
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    rank_ms = [dt / args.steps * 1e3]
    comm_exposed = None
    if trainer.reducer is not None and trainer.reducer.exposed:
        comm_exposed = sum(a.elapsed_time(b) for a, b in trainer.reducer.exposed) / args.steps
        trainer.reducer.exposed = None
    if world > 1:
        every = [torch.zeros(2, device=dev) for _ in range(world)]
        dist.all_gather(every, torch.tensor([dt, comm_exposed or 0.0], device=dev))
        rank_ms = [float(e[0]) / args.steps * 1e3 for e in every]
        comm_exposed = max(float(e[1]) for e in every)
        dt = max(float(e[0]) for e in every)
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    final_loss = float(loss)
    gnorm = trainer.opt.grad_norm()
    if not (math.isfinite(final_loss) and math.isfinite(gnorm) and gnorm > 0) or torch.equal(probe, trainer.opt.flat[:65536]):
        raise RuntimeError(f"invalid measurement: optimizer skipped or diverged (loss {final_loss}, grad norm {gnorm})")

    # contraction-kernel roofline: HIP events on the launch stream around every GEMM / conv / wgrad /
    # attention launch of `prof_steps` further steps of the same workload
    fams = ["gemm_conv", "wgrad", "attention", "norm", "elementwise"]
    prof = {}
    pair_us = 0.0
    if args.prof_steps > 0:
        trainer._graph = None                      # event timing needs the eager launches
        trainer._fwd_graph = trainer._fwd_loss = None
        step(args.warmup + args.steps)             # (untimed) first eager step after the graph: allocator warm-up
        torch.cuda.synchronize()
        # An event pair measures its kernels only while the stream is busy: on an idle stream the opening event is
        # stamped when the host submits it and the host's launch latency lands inside the pair.  A spin kernel in front
        # of each step lets the host run ahead; what a pair then still carries besides its kernels is the stream's
        # processing of the closing event, measured here (paired against bare tiny launches) and subtracted per launch below.
        stream = torch.cuda.current_stream().cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
        spin = int(20_000_000 * 70.0 / max(e0.elapsed_time(e1), 1e-3))      # cycles for ~70 ms (a profiler slows the host)
        cal = torch.zeros(32 << 20, device=dev)        # one clearing launch of 128 MB runs ~30 us, like the timed kernels
        torch.cuda._sleep(spin // 8)
        pair_us = max(0.0, float(lib.mi355_prof_event_overhead_us(stream, cal.data_ptr(), cal.numel(), 256)))
        del cal
        lib.mi355_prof_enable(1)
        for i in range(args.prof_steps):
            torch.cuda._sleep(spin)
            step(args.warmup + args.steps + 1 + i)
        torch.cuda.synchronize()
        msv, work, cnt = (ctypes.c_double * 5)(), (ctypes.c_double * 5)(), (ctypes.c_longlong * 5)()
        lib.mi355_prof_collect(msv, work, cnt)
        lib.mi355_prof_enable(0)
        for i, f in enumerate(fams):
            raw = msv[i] / args.prof_steps
            prof[f] = {"ms_per_step": max(raw - pair_us * 1e-3 * cnt[i] / args.prof_steps, 0.5 * raw), "raw_ms_per_step": raw,
                       "work_per_step": work[i] / args.prof_steps, "launches_per_step": cnt[i] / args.prof_steps}
    measured = None
    if rank == 0 and not args.small:
        # what this device sustains, beside the spec figures the fractions are quoted against: register-only bf16 MFMA
        # loops on pseudo-random operands (the clock the chip holds depends on the data) and a 1 GiB streaming copy
        def _timed(fn, n):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / n * 1e-3
        cs = torch.cuda.current_stream().cuda_stream
        sink = torch.zeros(4, device=dev)
        it = 80000
        t = _timed(lambda: lib.mi355_peak_mfma_bf16(sink.data_ptr(), 256, it, 1, cs), 20)
        src, dst = torch.ones(1 << 30, dtype=torch.uint8, device=dev), torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        tc = _timed(lambda: lib.mi355_peak_copy(src.data_ptr(), dst.data_ptr(), 1 << 30, cs), 10)
        measured = {"mfma_bf16_random_operands_TFLOPs": 256 * 4 * it * 8 * 16384 / t / 1e12, "copy_1GiB_TBs": 2 * (1 << 30) / tc / 1e12}
        del src, dst
    if rank == 0:
        roof = None
        if prof:
            gp = prof["gemm_conv"]
            tf = gp["work_per_step"] / (gp["ms_per_step"] * 1e-3) / 1e12 if gp["ms_per_step"] > 0 else 0.0
            # HBM-side bytes of the same family from the committed rocprofv3 --pmc passes over this workload
            # (tools/pmc_step.sh: counters cannot be read from inside the process), per library call like `achieved`
            traffic, traffic_src = None, None
            tf_path = os.path.join(ROOT, "profiles", "pmc", "step_traffic.json")
            if os.path.exists(tf_path) and not args.small and not args.lora and B == 32 and L == 32:
                rec = json.load(open(tf_path))
                if rec.get("source_fingerprint") == source_fingerprint():
                    fam = rec["families"]["gemm_conv"]
                    traffic = (fam["fetch_bytes_per_step"] + fam["write_bytes_per_step"]) / max(1.0, gp["launches_per_step"])
                    traffic_src = ("profiles/pmc/step_traffic.json (separate --pmc passes over this code, FETCH_SIZE x2 + "
                                   "WRITE_SIZE, bytes per call)")
                else:
                    traffic_src = "none: the committed counter profile was collected from other kernel sources"
            roof = {"bound": "mfma", "kernel": "gemm_dma / gemm_pp / gemm_wide kernel family (implicit-GEMM conv + linear, fwd and dgrad)",
                    "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS,
                    "traffic": traffic, "traffic_source": traffic_src, "event_pair_overhead_us": pair_us,
                    "measured_on_this_device": measured,
                    "frac_of_measured_mfma": (tf / measured["mfma_bf16_random_operands_TFLOPs"]) if measured else None,
                    "avg_launch_us": gp["ms_per_step"] * 1e3 / max(1.0, gp["launches_per_step"]),
                    "flops_per_launch": gp["work_per_step"] / max(1.0, gp["launches_per_step"]),
                    "families": {f: {"ms_per_step": round(p["ms_per_step"], 3), "raw_ms_per_step": round(p["raw_ms_per_step"], 3),
                                     "achieved": (p["work_per_step"] / (p["ms_per_step"] * 1e-3) / 1e12) if p["ms_per_step"] > 0 else 0.0,
                                     "unit": "TFLOP/s" if f in ("gemm_conv", "wgrad", "attention") else "TB/s",
                                     "launches": p["launches_per_step"]} for f, p in prof.items()}}
        step_flops = (2 if args.lora else 3) * FWD_GFLOP_PER_SAMPLE_256 * 1e9 * B * (L / 32.0) ** 2
        out = {"metric": "train images/sec (whole node) at 256^2 bf16", "value": value, "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic (random latents/text states in HBM, random-init weights)",
               "config": {"workload": ("small plumbing model" if args.small else
                                       "LoRA fine-tune step 256x256 (latent 32x32), frozen SD-v1 UNet 859.5M params, 128 rank-4 "
                                       "adapters (797k trainable), text cross-attention T=77" if args.lora else
                                       "denoiser train step 256x256 (latent 32x32), SD-v1 UNet 859.5M params, "
                                       "text cross-attention T=77"),
                          "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}",
                          "optimizer": "fused AdamW + global-norm clip", "final_loss": final_loss, "final_grad_norm": gnorm,
                          "hip_graph": bool(graphed), "forward_hip_graph": bool(fwd_graph)},
               "world_size": dist.get_world_size() if dist.is_initialized() else 1,
               "rank_ms_per_step": {"max": max(rank_ms), "min": min(rank_ms)},
               "comm_exposed_ms_per_step": comm_exposed,
               "gradient_exchange": trainer.exchange_summary(),
               "step_mfma_frac": step_flops / (ms * 1e-3) / (MFMA_PEAK_TFLOPS * 1e12) if not args.small else None,
               "roofline": roof}
        if world == 1 and not args.no_extra and not args.small and not args.lora:
            # the other two workloads of the metric, on the same device, after the headline measurement (its memory freed)
            del trainer, unet, lat, noise, ctx, ts, probe
            torch.cuda.empty_cache()
            out["sampler"] = sampler_leg(args.sampler_steps)
            torch.cuda.empty_cache()
            out["lora"] = lora_leg(args.lora_steps)
            torch.cuda.empty_cache()
            out["e2e"] = e2e_leg(args.e2e_steps)
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline and not args.lora:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
