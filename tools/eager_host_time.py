"""Host enqueue time against device time of the eager training step (the multi-GPU path runs eagerly): if the host needs
longer to enqueue a step than the device to run it, the step is host-bound."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd.trainer import DenoiserTrainer
from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig

from image_generator_amd.trainer import init_distributed
init_distributed()
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
if "--fwd-graph" in sys.argv:          # the forward pass replayed as a HIP graph, backward and optimizer eager
    tr.capture_forward(lat, noise, ts, ctx)
    for _ in range(2):
        tr.train_step(lat, noise, ts, ctx)
    torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    tr.train_step(lat, noise, ts, ctx)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/n:.2f} ms/step, wall {1e3*(t2-t0)/n:.2f} ms/step (reducer: {tr.reducer is not None}, forward graph: {tr._fwd_graph is not None})")
# forward / backward split of the host time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    noisy = lat
    pred = unet(noisy, ts, encoder_hidden_states=ctx).sample
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"forward only: host enqueue {1e3*(t1-t0)/n:.2f} ms, wall {1e3*(t2-t0)/n:.2f} ms")
