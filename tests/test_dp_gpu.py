"""The data-parallel train step on the MI355X with its gradient reducer switched on (one rank over RCCL): bucket
callbacks from the deferred weight-gradient queue, bf16 staging, and the optimizer reading the staged gradients."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
D = "cuda:0"


def test_dp_step_with_reducer_matches_plain_step(rccl_world1):
    from image_generator_amd.trainer import DenoiserTrainer
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    g = torch.Generator().manual_seed(21)
    lat, noise = torch.randn(4, 4, 8, 8, generator=g).to(D), torch.randn(4, 4, 8, 8, generator=g).to(D)
    ts, ctx = torch.randint(0, 1000, (4,), generator=g).to(D), torch.randn(4, 16, 64, generator=g).to(D).to(torch.bfloat16)

    def run(with_reducer):
        torch.manual_seed(0)
        unet = UNet2DConditionModel(DenoiserConfig.small()).to(D)
        if with_reducer:
            os.environ["MI355_FORCE_REDUCER"] = "1"
        try:
            tr = DenoiserTrainer(unet, lr=1e-3, bucket_mb=0.25)
        finally:
            os.environ.pop("MI355_FORCE_REDUCER", None)
        assert (tr.reducer is not None) == with_reducer
        if with_reducer:
            assert tr.reducer.active and len(tr.reducer.buckets) >= 3
        losses, norms = [], []
        for i in range(4):
            losses.append(float(tr.train_step(lat, noise, ts, ctx)))
            norms.append(tr.opt.grad_norm())
        return losses, norms, tr

    try:
        la, na, tr = run(True)
        # one more backward pass on each trainer from the same weights: the staged (bf16) gradients the optimizer
        # would consume match the fp32 buffer of the plain step to bf16 rounding, whether a range reached the staging
        # buffer through the pack pass or straight from an unsplit weight-gradient job
        assert tr._direct_stage
        tr.wgrad_queue.flush_blocks = 64
        tr.forward_backward(lat, noise, ts, ctx)
        st = tr.reducer.stage_all.float().clone()
        os.environ["MI355_NO_DIRECT_STAGE"] = "1"
        os.environ["MI355_FORCE_REDUCER"] = "1"
        try:
            tr2 = DenoiserTrainer(tr.unet, lr=1e-3, bucket_mb=0.25)
        finally:
            os.environ.pop("MI355_NO_DIRECT_STAGE", None)
            os.environ.pop("MI355_FORCE_REDUCER", None)
        assert not tr2._direct_stage
        tr2.forward_backward(lat, noise, ts, ctx)
        fg, st2 = tr2.opt.grad, tr2.reducer.stage_all.float()
        assert float(fg.norm()) > 0
        assert float((st2 - fg).norm() / fg.norm()) <= 2 ** -8          # pack pass only
        assert float((st - fg).norm() / fg.norm()) <= 2 ** -8           # direct bf16 stores + pack of the rest
        # unsplit jobs round the same fp32 sums the pack pass rounds: identical except where split tiles' atomics
        # reorder the fp32 additions
        assert float((st != st2).float().mean()) <= 0.05
    finally:
        torch.cuda.synchronize()
    lb, nb, _ = run(False)
    assert la[0] == lb[0]                                    # same forward before any update
    for a, b in zip(la, lb):
        assert abs(a - b) <= 1e-2 * abs(b), (la, lb)         # bf16-rounded gradients: trajectories stay together
    for a, b in zip(na, nb):
        assert math.isfinite(a) and abs(a - b) <= 2e-2 * b, (na, nb)
    assert la[-1] < la[0]


@pytest.mark.parametrize("lora", [False, True])
def test_dp_step_with_captured_forward_matches_eager_step(rccl_world1, lora):
    """DenoiserTrainer.capture_forward: the forward pass replayed as a HIP graph, the backward pass walking the tape
    recorded at capture (retain_graph) with the reducer's collectives eager.  The kernels and their order are those of
    the eager step, so losses, gradient norms and weights stay with the eager trainer's over several steps with
    changing inputs (tolerance: fp32 summation order of atomics only)."""
    from image_generator_amd.trainer import DenoiserTrainer
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    g = torch.Generator().manual_seed(29)
    batches = [(torch.randn(4, 4, 8, 8, generator=g).to(D), torch.randn(4, 4, 8, 8, generator=g).to(D),
                torch.randint(0, 1000, (4,), generator=g).to(D), torch.randn(4, 16, 64, generator=g).to(D).to(torch.bfloat16))
               for _ in range(5)]

    def run(fwd_graph):
        torch.manual_seed(0)
        unet = UNet2DConditionModel(DenoiserConfig.small())
        if lora:                                         # frozen base, rank-4 adapters as the only trainable parameters
            from image_generator_amd.generator import inject_lora
            with torch.no_grad():
                for ad in inject_lora(unet, r=4, alpha=0.5):
                    ad.lora_B.weight.normal_(0, 0.05)
        unet.to(D)
        os.environ["MI355_FORCE_REDUCER"] = "1"
        try:
            tr = DenoiserTrainer(unet, lr=1e-3, bucket_mb=0.25)
        finally:
            os.environ.pop("MI355_FORCE_REDUCER", None)
        assert tr.reducer is not None and tr.reducer.active
        losses, norms = [], []
        for i, b in enumerate(batches):
            if fwd_graph and i == 1:
                tr.capture_forward(*b)
            losses.append(float(tr.train_step(*b)))
            norms.append(tr.opt.grad_norm())
        assert (tr._fwd_graph is not None) == fwd_graph
        return losses, norms, tr.opt.flat.clone()

    try:
        la, na, wa = run(True)
    finally:
        torch.cuda.synchronize()
    lb, nb, wb = run(False)
    for a, b in zip(la + na, lb + nb):
        assert math.isfinite(a) and abs(a - b) <= 1e-5 * abs(b), (la, lb, na, nb)
    assert float((wa - wb).norm() / wb.norm()) <= 1e-6
    assert len(set(la)) == len(la)                       # every step saw its own batch


def test_lora_dp_step_with_reducer_matches_plain_step(rccl_world1):
    """LoRA fine-tune with the gradient reducer on (one rank over RCCL): the adapters are the reducer's only (tail)
    bucket, the frozen base has none, and the optimizer consumes the staged bf16 adapter gradients - the trajectory
    stays with the plain LoRA step's (bf16 rounding of the gradients only) and the base weights never move."""
    from image_generator_amd.generator import inject_lora
    from image_generator_amd.trainer import DenoiserTrainer
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    g = torch.Generator().manual_seed(23)
    lat, noise = torch.randn(4, 4, 8, 8, generator=g).to(D), torch.randn(4, 4, 8, 8, generator=g).to(D)
    ts, ctx = torch.randint(0, 1000, (4,), generator=g).to(D), torch.randn(4, 16, 64, generator=g).to(D).to(torch.bfloat16)

    def run(with_reducer):
        torch.manual_seed(0)
        unet = UNet2DConditionModel(DenoiserConfig.small())
        adapters = inject_lora(unet, r=4, alpha=0.5)
        with torch.no_grad():
            for ad in adapters:
                ad.lora_B.weight.normal_(0, 0.05)
        unet.to(D)
        base = {n: p.detach().clone() for n, p in unet.named_parameters() if not p.requires_grad}
        if with_reducer:
            os.environ["MI355_FORCE_REDUCER"] = "1"
        try:
            tr = DenoiserTrainer(unet, lr=1e-3)
        finally:
            os.environ.pop("MI355_FORCE_REDUCER", None)
        assert tr.opt.frozen_base and (tr.reducer is not None) == with_reducer
        if with_reducer:
            assert tr.reducer.active and not tr._direct_stage and tr.reducer.tail[1] - tr.reducer.tail[0] == tr.opt.grad.numel()
        losses, norms = [], []
        for _ in range(4):
            losses.append(float(tr.train_step(lat, noise, ts, ctx)))
            norms.append(tr.opt.grad_norm())
        for n, p in unet.named_parameters():
            if not p.requires_grad:
                assert torch.equal(p.detach(), base[n]), n
        return losses, norms, tr

    try:
        la, na, tr = run(True)
        st, fg = tr.reducer.stage_all.float(), tr.opt.grad
        assert float(fg.norm()) > 0 and float((st - fg).norm() / fg.norm()) <= 2 ** -8
    finally:
        torch.cuda.synchronize()
    lb, nb, _ = run(False)
    assert la[0] == lb[0]
    for a, b in zip(la, lb):
        assert abs(a - b) <= 1e-2 * abs(b), (la, lb)
    for a, b in zip(na, nb):
        assert math.isfinite(a) and a > 0 and abs(a - b) <= 2e-2 * b, (na, nb)
    assert la[-1] < la[0]


def test_fullsize_direct_staging_matches_pack_pass(rccl_world1):
    """SD-v1 denoiser at the benchmark size (batch 32, 32x32 latents) with the reducer on: the level-0 weight gradients
    are pixel-split there (fp32 atomics, packed afterwards), everything else is stored as bf16 by its own job.  The
    staged buffer equals, to bf16 rounding, the fp32 gradients of the same backward pass without direct staging, every
    range of the flat buffer is either staged by a job or packed, and both kinds occur."""
    from image_generator_amd.trainer import DenoiserTrainer
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    torch.manual_seed(98)
    with torch.device(D):
        unet = UNet2DConditionModel(DenoiserConfig.sd_v1())
    B, L, T = 32, 32, 77
    g = torch.Generator(device=D).manual_seed(8)
    lat = torch.randn(B, 4, L, L, device=D, generator=g) * 0.9
    noise = torch.randn(B, 4, L, L, device=D, generator=g)
    ctx = torch.randn(B, T, 768, device=D, generator=g).to(torch.bfloat16)
    ts = torch.randint(0, 1000, (B,), device=D, generator=g)

    try:
        out = []
        for direct in (True, False):
            os.environ["MI355_FORCE_REDUCER"] = "1"
            if not direct:
                os.environ["MI355_NO_DIRECT_STAGE"] = "1"
            try:
                tr = DenoiserTrainer(unet, lr=1e-4)
            finally:
                os.environ.pop("MI355_FORCE_REDUCER", None)
                os.environ.pop("MI355_NO_DIRECT_STAGE", None)
            assert tr._direct_stage == direct
            tr.opt.grad.zero_()
            tr.reducer.stage_all.fill_(float("nan"))            # every element must be written by a job or by the pack pass
            staged = {}
            if direct:
                keep = tr.reducer.finish

                def finish(keep=keep, tr=tr, staged=staged):
                    staged.update(tr.reducer.staged or {})
                    keep()
                tr.reducer.finish = finish
            tr.forward_backward(lat, noise, ts, ctx)
            torch.cuda.synchronize()
            out.append((tr.reducer.stage_all.float().clone(), tr.opt.grad.clone(), staged))
            n_layers = sum(len(l) for l in tr.reducer._layers)
            del tr
        (st_d, fg_d, staged), (st_p, fg_p, _) = out
        assert bool(torch.isfinite(st_d).all()) and bool(torch.isfinite(st_p).all())
        assert 0 < len(staged) < n_layers                          # unsplit jobs staged, pixel-split ones packed
        covered = sum(staged.values())
        assert 0.8 * st_d.numel() < covered < st_d.numel()
        assert float((st_p - fg_p).norm() / fg_p.norm()) <= 2 ** -8
        assert float((st_d - fg_p).norm() / fg_p.norm()) <= 2 ** -8
        # the fp32 buffer of the direct run holds only what no job staged
        assert float(fg_d.norm()) < float(fg_p.norm())
    finally:
        torch.cuda.synchronize()


def test_native_communicator_and_trainer_through_it(rccl_world1):
    """The library's own RCCL entry points (one rank): unique id -> communicator -> in-place bf16 / fp32 all-reduce on a
    given stream (the sum over one rank is the input, bit for bit, and work queued on that stream before and after it
    keeps its order), then the train step with the reducer all-reducing through that communicator: the same
    trajectory, bit for bit, as through the torch process group (same kernels, same buffers, one rank)."""
    import ctypes
    from image_generator_amd._lib import lib, check
    from image_generator_amd.trainer import DenoiserTrainer, NativeComm
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    assert lib.mi355_dp_unique_id_bytes() == 128
    comm = NativeComm.get()
    assert comm.world == 1 and lib.mi355_dp_comm_world(comm.handle) == 1
    assert lib.mi355_dp_comm_world(None) < 0 and b"communicator" in lib.mi355_last_error()
    side = torch.cuda.Stream(D)
    for dt, code in ((torch.bfloat16, 0), (torch.float32, 1)):
        x = torch.randn(1 << 20, device=D).to(dt)
        want = (x.float() * 2 + 1).to(dt)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            x.mul_(2)
            check(lib.mi355_dp_allreduce_sum(comm.handle, x.data_ptr(), x.numel(), code, side.cuda_stream), "allreduce")
            x.add_(1)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(x, want)
    assert lib.mi355_dp_allreduce_sum(comm.handle, x.data_ptr(), x.numel(), 5, side.cuda_stream) < 0     # unknown dtype

    g = torch.Generator().manual_seed(29)
    lat, noise = torch.randn(4, 4, 8, 8, generator=g).to(D), torch.randn(4, 4, 8, 8, generator=g).to(D)
    ts, ctx = torch.randint(0, 1000, (4,), generator=g).to(D), torch.randn(4, 16, 64, generator=g).to(D).to(torch.bfloat16)

    def run(native):
        torch.manual_seed(0)
        unet = UNet2DConditionModel(DenoiserConfig.small()).to(D)
        os.environ["MI355_FORCE_REDUCER"] = "1"
        if native:
            os.environ["MI355_DP_NATIVE"] = "1"
        try:
            tr = DenoiserTrainer(unet, lr=1e-3, bucket_mb=0.25)
        finally:
            os.environ.pop("MI355_FORCE_REDUCER", None)
            os.environ.pop("MI355_DP_NATIVE", None)
        assert (tr.reducer.native is not None) == native
        out = [float(tr.train_step(lat, noise, ts, ctx)) for _ in range(4)]
        return out, tr.opt.flat.clone()

    la, pa = run(True)
    lb, pb = run(False)
    assert la == lb and torch.equal(pa, pb)
    assert la[-1] < la[0]
