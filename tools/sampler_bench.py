"""The LoRA fine-tune step and the 512x512 sampling loop on one MI355X (run on the GPU box):
  --mode sampler : 512x512 sampling loop, 16 images with classifier-free guidance (UNet batch 32), DDIM,
                   one HIP graph per step shape; reports sampler steps/s and the MFMA fraction
                   (25.7 TFLOP per step: 32 x 803.9 GFLOP forward at 64x64 latents)
  --mode lora    : LoRA fine-tune step at 256x256, batch 32 (frozen base, rank-4 adapters on
                   attn{1,2}.to_{q,k,v,out}); reports images/s
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def sampler(args):
    from image_generator_amd.generator import ImageGenerator
    torch.manual_seed(0)
    gen = ImageGenerator("cuda:0", text_encoder_kwargs=dict(num_layers=2))     # text tower depth is irrelevant here
    gen.eval()
    n, side = args.images, args.res // 8
    emb = torch.randn(2 * n, 77, 768, device="cuda:0").to(torch.bfloat16)
    lat = torch.randn(n, 4, side, side, device="cuda:0")
    for use_graph in (False, True):
        gen.sample_latents(emb, lat, 3, 7.5, use_graph=use_graph)            # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gen.sample_latents(emb, lat, args.steps, 7.5, use_graph=use_graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        flops = 2 * n * 803.9e9 * (side / 64.0) ** 2
        print(json.dumps({"mode": "sampler", "hip_graph": use_graph, "images": n, "res": args.res, "steps": args.steps,
                          "steps_per_s": args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                          "mfma_frac": flops * args.steps / dt / 2.5e15, "finite": bool(torch.isfinite(out).all())}))


def lora(args):
    from image_generator_amd.generator import inject_lora
    from image_generator_amd.trainer import DenoiserTrainer
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    torch.manual_seed(0)
    with torch.device("cuda:0"):
        unet = UNet2DConditionModel(DenoiserConfig.sd_v1())
    adapters = inject_lora(unet, r=4, alpha=1.0)
    with torch.no_grad():
        for ad in adapters:                      # B = 0 at init would leave half of the arithmetic multiplying zeros
            ad.lora_B.weight.normal_(0, 0.02)
    n_train = sum(p.numel() for p in unet.parameters() if p.requires_grad)
    tr = DenoiserTrainer(unet, lr=1e-4, weight_decay=1e-2, max_grad_norm=1.0)
    B = args.batch
    lat, noise = torch.randn(B, 4, 32, 32, device="cuda:0"), torch.randn(B, 4, 32, 32, device="cuda:0")
    ts = torch.randint(0, 1000, (B,), device="cuda:0")
    ctx = torch.randn(B, 77, 768, device="cuda:0").to(torch.bfloat16)
    for graph in (False, True):
        if graph:
            tr.capture(lat, noise, ts, ctx)
        for _ in range(3):
            tr.train_step(lat, noise, ts, ctx)
        torch.cuda.synchronize()
        probe = tr.opt.flat[:4096].clone()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = tr.train_step(lat, noise, ts, ctx)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"mode": "lora", "hip_graph": graph, "adapters": len(adapters), "trainable_params": n_train,
                          "batch": B, "images_per_s": B * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                          "mfma_frac": 2 * 180.3e9 * B * args.steps / dt / 2.5e15,
                          "loss": float(loss), "grad_norm": tr.opt.grad_norm(),
                          "weights_moved": not torch.equal(probe, tr.opt.flat[:4096])}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="sampler")
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tune", type=int, nargs=2, action="append", default=[], metavar=("KEY", "VALUE"),
                    help="mi355_tune_set(KEY, VALUE) before the run (kernel-variant switches, for A/B on one box)")
    a = ap.parse_args()
    for key, value in a.tune:
        from image_generator_amd._lib import lib
        lib.mi355_tune_set(key, value)
    (sampler if a.mode == "sampler" else lora)(a)
