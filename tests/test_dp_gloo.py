"""CPU, world_size 2 and 4 over gloo: the bucketed gradient reducer of the data-parallel step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Pack:
    on_grad_ready = None


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from image_generator_amd.trainer import GradBucketReducer
    sizes = [4096, 1024, 8192, 512, 2048, 16384]          # "GEMM weights"
    n_big, tail = sum(sizes), 300
    flat = torch.zeros(n_big + tail)
    packs, off = [], 0
    for n in sizes:
        packs.append((off, n, _Pack()))
        off += n
    for step in range(2):
        # step 0: the reduced gradients stay in the bf16 staging buffer (what the fused optimizer reads);
        # step 1: `unpack=True` also writes them back into the fp32 buffer
        red = GradBucketReducer(flat, n_big, packs, bucket_mb=0.03, unpack=step == 1)   # ~8k floats per bucket
        assert len(red.buckets) >= 2
        g = torch.Generator().manual_seed(100 * step + rank)
        local = torch.randn(n_big + tail, generator=g)
        flat[n_big:] = local[n_big:]
        # "backward": layers finish in reverse order, each writes its slice then signals
        for o, n, pk in reversed(packs):
            flat[o:o + n] = local[o:o + n]
            pk.on_grad_ready()
        red.finish()
        out[(rank, step)] = flat.clone() if step == 1 else red.stage_all.float()
        if step == 0:
            assert torch.equal(flat, local)                           # the fp32 buffer keeps the local gradients
    # step 2: two layers' producers stored their bf16 gradients straight into the staging buffer (`staged`: address ->
    # floats); the fp32 buffer holds garbage there, which the pack pass must not read
    red = GradBucketReducer(flat, n_big, packs, bucket_mb=0.03)
    g = torch.Generator().manual_seed(200 + rank)
    local = torch.randn(n_big + tail, generator=g)
    flat[:] = local
    staged = {}
    for idx in (1, 4):
        o, n, _ = packs[idx]
        red.stage_all[o:o + n] = local[o:o + n].to(torch.bfloat16)
        flat[o:o + n] = float("nan")
        staged[flat.data_ptr() + 4 * o] = n
    red.staged = staged
    for o, n, pk in reversed(packs):
        pk.on_grad_ready()
    red.finish()
    out[(rank, 2)] = red.stage_all.float()
    # step 3: a bucket boundary lands on the last signalling layer and a region without any such layer follows inside
    # [0, n_big) (the channel-padded output convolution, whose gradient arrives through autograd): that trailing bucket
    # has no callback to fire it and must still be reduced by finish()
    extra = 640
    flat2 = torch.zeros(n_big + extra + tail)
    red = GradBucketReducer(flat2, n_big + extra, packs, bucket_mb=0.03)
    assert red.buckets[-1][2] == [] and red.buckets[-1][1] - red.buckets[-1][0] == extra
    g = torch.Generator().manual_seed(300 + rank)
    local = torch.randn(n_big + extra + tail, generator=g)
    flat2[:] = local
    for o, n, pk in reversed(packs):
        pk.on_grad_ready()
    red.finish()
    out[(rank, 3)] = red.stage_all.float()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_allreduce(world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for step in range(3):
        locs = [torch.randn(4096 + 1024 + 8192 + 512 + 2048 + 16384 + 300, generator=torch.Generator().manual_seed(100 * step + r))
                for r in range(world)]
        want = sum(l.to(torch.bfloat16).float() for l in locs)        # bf16 pack, sum, unpack
        for r in range(world):
            got = out[(r, step)]
            assert torch.allclose(got, want, rtol=2e-2, atol=3e-2)
            assert torch.equal(out[(0, step)], got)                   # every rank holds the same reduced gradient
    n = 4096 + 1024 + 8192 + 512 + 2048 + 16384 + 640 + 300
    want = sum(torch.randn(n, generator=torch.Generator().manual_seed(300 + r)).to(torch.bfloat16).float() for r in range(world))
    for r in range(world):
        assert torch.allclose(out[(r, 3)], want, rtol=2e-2, atol=3e-2)      # the packless bucket included
