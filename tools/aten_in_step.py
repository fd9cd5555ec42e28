"""Which ATen operators still launch kernels inside one eager train step (headline configuration), with the Python frames
that call them; run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from image_generator_amd.trainer import DenoiserTrainer
from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig

d = torch.device("cuda:0")
torch.manual_seed(0)
cfg = DenoiserConfig.sd_v1()
with torch.device(d):
    unet = UNet2DConditionModel(cfg)
tr = DenoiserTrainer(unet, lr=1e-4)
B = 32
lat, noise = torch.randn(B, 4, 32, 32, device=d), torch.randn(B, 4, 32, 32, device=d)
ctx = torch.randn(B, 77, cfg.cross_attention_dim, device=d).to(torch.bfloat16)
ts = torch.randint(0, 1000, (B,), device=d)
for _ in range(3):
    tr.train_step(lat, noise, ts, ctx)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_step(lat, noise, ts, ctx)
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue
    frames = [f for f in (ev.stack or []) if "torch/" not in f and "built-in" not in f and "<frozen" not in f][:4]
    key = (ev.name, str(ev.input_shapes)[:70] if ev.input_shapes else "", " <- ".join(f.split("/")[-1] for f in frames))
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1; r[1] += ev.device_time_total
for (name, shp, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:14s} x{n:3d} {us:8.1f} us  {shp}  {where}")
