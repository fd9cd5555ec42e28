"""cProfile of the host side of eager training steps (headline configuration); run on the GPU box."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.train_step(lat, noise, ts, ctx)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
