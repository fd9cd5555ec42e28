"""Times the SD-v1 auto-encoder on one MI355X: encode of a 256x256 training batch, decode of 512x512 samples."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd.vae import AutoencoderKL, VaeConfig

d = torch.device("cuda:0")
torch.manual_seed(0)
vae = AutoencoderKL(VaeConfig.sd_v1()).to(d)
for name, fn in (("encode 32 x 256x256", lambda: vae.encode(torch.rand(32, 3, 256, 256, device=d) * 2 - 1).latent_dist.mode()),
                 ("decode 16 x 512x512", lambda: vae.decode(torch.randn(16, 4, 64, 64, device=d)).sample)):
    out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt*1e3:.1f} ms, output {tuple(out.shape)}, finite {bool(torch.isfinite(out).all())}, "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
