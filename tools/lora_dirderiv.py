"""Diagnosis: directional derivative of the full-size LoRA step along its own gradient, central differences at several
step sizes, for the trainer's gradient and for the plain autograd path's gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd.generator import inject_lora
from image_generator_amd.trainer import DenoiserTrainer
from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
from image_generator_amd.ops import mse_loss

D = "cuda:0"
small = "--small" in sys.argv
torch.manual_seed(77)
with torch.device(D):
    unet = UNet2DConditionModel(DenoiserConfig.small() if small else DenoiserConfig.sd_v1())
adapters = inject_lora(unet, r=4, alpha=1.0)
torch.manual_seed(78)
bstd = float(os.environ.get("BSTD", "0.02"))
with torch.no_grad():
    for ad in adapters:
        ad.lora_B.weight.normal_(0, bstd)
B, L, T = (4, 8, 16) if small else (32, 32, 77)
cd = unet.config.cross_attention_dim
g = torch.Generator(device=D).manual_seed(8)
lat = torch.randn(B, 4, L, L, device=D, generator=g) * 0.9
noise = torch.randn(B, 4, L, L, device=D, generator=g)
ctx = torch.randn(B, T, cd, device=D, generator=g).to(torch.bfloat16)
ts = torch.randint(0, 1000, (B,), device=D, generator=g)
tr = DenoiserTrainer(unet, lr=1e-3)
a = tr.acp[ts].view(-1, 1, 1, 1)
noisy = a.sqrt() * lat + (1 - a).sqrt() * noise

def loss_at(theta):
    with torch.no_grad():
        tr.opt.flat.copy_(theta)
        return float(((unet(noisy, ts, encoder_hidden_states=ctx).sample.double() - noise.double()) ** 2).mean())

p0 = tr.opt.flat.clone()
# autograd-path gradient
for p in unet.parameters():
    p.grad = None
unet._flat_grad = None
for m in unet.modules():
    if hasattr(m, "grad_dst"):
        m.grad_dst = None
    if getattr(m, "lora_fused", None) is not None:
        m.lora_fused.grad_dst = None
loss = mse_loss(unet(noisy, ts, encoder_hidden_states=ctx).sample, noise)
loss.backward()
ga = torch.zeros_like(p0)
for p in unet.parameters():
    if p.requires_grad:
        o = unet._offs[id(p)] - tr.opt.base
        ga[o:o + p.numel()] = p.grad.flatten()
tr2 = DenoiserTrainer(unet, lr=1e-3)
tr2.forward_backward(lat, noise, ts, ctx)
gt = tr2.opt.grad.clone()
print("loss", float(loss), "|g_auto|", float(ga.norm()), "|g_trainer|", float(gt.norm()),
      "cos", float(torch.dot(ga, gt) / (ga.norm() * gt.norm())))
# split by A / B
isA = torch.zeros_like(p0, dtype=torch.bool)
for ad in adapters:
    o = unet._offs[id(ad.lora_A.weight)] - tr.opt.base
    isA[o:o + ad.lora_A.weight.numel()] = True
for name, gvec in (("auto", ga), ("trainer", gt)):
    for part, mask in (("all", torch.ones_like(isA)), ("A only", isA), ("B only", ~isA)):
        d = gvec * mask
        n = float(d.norm())
        d = d / n
        out = []
        for h in (0.02, 0.05, 0.1, 0.2, 0.4):
            lp, lm = loss_at(p0 + h * d), loss_at(p0 - h * d)
            out.append(f"h={h}: {(lp - lm) / (2 * h) / n:.3f}")
        print(name, part, "|g|", f"{n:.5f}", " measured/predicted:", "  ".join(out))
