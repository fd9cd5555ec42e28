"""Two data-parallel ranks on the MI355X (both on cuda:0, gloo rendezvous): the real multi-rank train step - bucket
callbacks from the weight-gradient queue, bf16 staging with direct stores, all-reduce on the side stream, the optimizer
dividing by the world size - against ONE process training on the concatenated batch.  RCCL refuses two ranks on one
device, so the collective itself is carried by gloo through host memory (patched in the workers only); everything
around it is the product path."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
D = "cuda:0"
STEPS = 4


def _data():
    g = torch.Generator().manual_seed(41)
    lat, noise = torch.randn(8, 4, 16, 16, generator=g), torch.randn(8, 4, 16, 16, generator=g)
    ts, ctx = torch.randint(0, 1000, (8,), generator=g), torch.randn(8, 9, 64, generator=g).to(torch.bfloat16)
    return lat, noise, ts, ctx


def _model(lora):
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    torch.manual_seed(0)
    unet = UNet2DConditionModel(DenoiserConfig.small())
    if lora:
        from image_generator_amd.generator import inject_lora
        with torch.no_grad():
            for ad in inject_lora(unet, r=4, alpha=0.5):
                ad.lora_B.weight.normal_(0, 0.05)
    return unet.to(D)


def _train(sl, lora):
    from image_generator_amd.trainer import DenoiserTrainer
    unet = _model(lora)
    tr = DenoiserTrainer(unet, lr=1e-3, bucket_mb=0.25)
    lat, noise, ts, ctx = (x[sl].to(D) for x in _data())
    losses, norms = [], []
    for _ in range(STEPS):
        losses.append(float(tr.train_step(lat, noise, ts, ctx)))
        norms.append(tr.opt.grad_norm())
    return losses, norms, tr


def _worker(rank, world, port, out, lora):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    real = dist.all_reduce

    def through_host(t, op=dist.ReduceOp.SUM, group=None):
        if not t.is_cuda:
            return real(t, op=op, group=group)
        c = t.float().cpu()                      # waits for the stream the reducer launched this on
        real(c, op=op, group=group)
        t.copy_(c)
    dist.all_reduce = through_host
    try:
        losses, norms, tr = _train(slice(4 * rank, 4 * rank + 4), lora)
        assert tr.world == 2 and tr.reducer is not None and tr.reducer.active and not tr._fused_sumsq
        if lora:
            assert tr.opt.frozen_base and not tr._direct_stage
        else:
            assert tr._direct_stage and len(tr.reducer.buckets) >= 3
        torch.cuda.synchronize()
        out[rank] = (losses, norms, tr.opt.flat.detach().cpu())
    finally:
        dist.all_reduce = real
        dist.destroy_process_group()


@pytest.mark.parametrize("lora", [False, True], ids=["full", "lora"])
def test_two_ranks_follow_the_single_process_run_on_the_joint_batch(lora):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out, lora), nprocs=2, join=True)
    (l0, n0, p0), (l1, n1, p1) = out[0], out[1]
    assert torch.equal(p0, p1)                       # the replicas stay bit-identical: same reduced gradients, same update
    assert n0 == n1                                  # the norm is taken after the reduction
    ls, ns, tr = _train(slice(0, 8), lora)
    ps = tr.opt.flat.detach().cpu()
    for a, b, c in zip(l0, l1, ls):                  # mean over the joint batch = mean of the two local means
        assert abs(0.5 * (a + b) - c) <= 1e-2 * abs(c), (l0, l1, ls)
    for a, c in zip(n0, ns):                         # |mean of the two local gradients| = |gradient of the joint mean|
        assert math.isfinite(a) and abs(a - c) <= 2e-2 * c, (n0, ns)
    assert 0.5 * (l0[-1] + l1[-1]) < 0.5 * (l0[0] + l1[0])
    init = _flat_init(lora, tr)
    moved = float((ps - init).norm())
    assert moved > 0 and float((p0 - ps).norm()) <= 0.25 * moved, (float((p0 - ps).norm()), moved)


def _flat_init(lora, tr):
    """The flat master parameters the optimizer owns (everything, or the adapters) of the seed-0 model before any update."""
    unet = _model(lora)
    flat, _ = unet.flat_parameters()
    base, end = (tr.unet.trainable_range() if lora else (0, flat.numel()))
    return flat[base:end].detach().cpu().clone()
