"""Train-step harness for the denoiser: the build's counterpart of the hot loop in
/root/reference/trainer.py:227-280 (GeneratorTrainer.train).

Step order kept from the reference: forward + loss -> non-finite guard -> backward -> gradient-norm
clip -> AdamW(lr 1e-4, wd 1e-2, trainer.py:102-103) -> CosineAnnealingWarmRestarts(T_0=10, T_mult=2,
eta_min=1e-6) stepped with the fractional epoch `epoch + idx/len` (trainer.py:105-106,261).

What is different on purpose (MI355X-first):
  * data parallelism is a full replica per GPU (the 860 M-parameter model with fp32 masters and AdamW
    state is ~14 GB of 288 GB) with bucketed gradient all-reduce on RCCL, issued from gradient-ready
    callbacks on a side HIP stream while backward continues - not the reference's FSDP full-shard
    (trainer.py:82-94) whose all-gathers exist only to save memory this machine does not lack;
  * bf16 compute with fp32 masters replaces fp16 autocast + GradScaler (trainer.py:71,243,256-259);
  * the non-finite guard (trainer.py:247-253) runs on the device inside the fused optimizer (a NaN/Inf
    gradient norm skips the update on every rank because the norm is computed after the all-reduce), so
    the step has no host synchronisation;
  * the gradient clip uses the true global norm (the reference clips the local FSDP shard of
    still-scaled gradients, trainer.py:257 - an artefact of its wrapper, not an algorithmic choice).
"""
from __future__ import annotations

import math
import os

import torch
import torch.distributed as dist

from ._lib import lib, check
from .ops import BF16, F32, _stream, mse_loss


def cosine_warm_restarts_lr(base_lr, epoch_float, t0=10, t_mult=2, eta_min=1e-6):
    """Closed form of torch's CosineAnnealingWarmRestarts.step(epoch_float) (trainer.py:105-106,261)."""
    e = epoch_float
    if t_mult == 1:
        t_i, t_cur = t0, e % t0
    elif e >= t0:
        n = int(math.log(e / t0 * (t_mult - 1) + 1, t_mult))
        t_cur = e - t0 * (t_mult ** n - 1) / (t_mult - 1)
        t_i = t0 * t_mult ** n
    else:
        t_i, t_cur = t0, e
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / t_i)) / 2


class NativeComm:
    """The library's own RCCL communicator (`mi355_dp_*`): all-reduces go straight onto the stream they are given, with
    no hop through a process-group stream.  The unique id travels over the torch process group that `init_distributed`
    made; one communicator per process, kept until exit."""

    _instance = None

    @classmethod
    def get(cls, group=None):
        if cls._instance is None:
            cls._instance = cls(group)
        return cls._instance

    def __init__(self, group=None):
        import ctypes
        n = int(lib.mi355_dp_unique_id_bytes())
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        buf = ctypes.create_string_buffer(n)
        if rank == 0:
            check(lib.mi355_dp_unique_id(buf, n), "dp_unique_id")
        box = [bytes(buf.raw) if rank == 0 else None]
        if self.world > 1:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
        handle = ctypes.c_void_p()
        check(lib.mi355_dp_comm_create(ctypes.byref(handle), box[0], n, rank, self.world), "dp_comm_create")
        self.handle = handle
        if int(lib.mi355_dp_comm_world(self.handle)) != self.world:
            raise RuntimeError("NativeComm: RCCL reports another world size than the process group")

    def all_reduce_sum(self, t):
        """In place, on the current stream."""
        if t.dtype not in (BF16, F32) or not t.is_contiguous():
            raise ValueError("NativeComm.all_reduce_sum: contiguous bf16 or fp32 tensors only")
        check(lib.mi355_dp_allreduce_sum(self.handle, t.data_ptr(), t.numel(), 0 if t.dtype == BF16 else 1, _stream()),
              "dp_allreduce_sum")

    def destroy(self):
        if self.handle is not None:
            torch.cuda.synchronize()
            check(lib.mi355_dp_comm_destroy(self.handle), "dp_comm_destroy")
            self.handle = None
            type(self)._instance = None


class GradBucketReducer:
    """Bucketed bf16 all-reduce of the flat fp32 gradient buffer, overlapped with backward.

    The flat buffer is cut into contiguous buckets.  GEMM-weight gradients are written directly into the
    buffer by the weight-gradient kernels; each layer's `on_grad_ready` callback decrements its bucket's
    counter and the bucket that reaches zero is packed to bf16, all-reduced and unpacked on `comm_stream`
    behind an event recorded on the compute stream.  `finish()` reduces the tail (small parameters) and
    makes the compute stream wait for all communication.
    """

    def __init__(self, flat_grad, n_big, grad_packs, group=None, bucket_mb=64, unpack=False, native=None):
        """flat_grad: fp32 gradient buffer; [0, n_big) holds the GEMM weights, the rest small parameters;
        grad_packs: (offset, numel, pack) per layer, `pack.on_grad_ready` is installed here;
        native: a `NativeComm` to all-reduce through instead of the torch process group."""
        self.group = group
        self.native = native
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.flat_grad = flat_grad
        self.device = self.flat_grad.device
        self.use_cuda = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.use_cuda else None
        bucket_elems = max(1, int(bucket_mb * (1 << 20) // 4))
        # bucket boundaries over [0, n_big) aligned to weight boundaries, then one tail bucket
        packs = sorted(grad_packs, key=lambda t: t[0])
        self.buckets = []      # (start, end, [packs])
        self._layers = []      # per bucket: (offset, numel) of its layers, ascending
        cur_start, cur_packs, cur_layers = 0, [], []
        for o, n, pk in packs:
            cur_packs.append(pk)
            cur_layers.append((o, n))
            if o + n - cur_start >= bucket_elems:
                self.buckets.append([cur_start, o + n, cur_packs])
                self._layers.append(cur_layers)
                cur_start, cur_packs, cur_layers = o + n, [], []
        if cur_packs or cur_start < n_big:
            self.buckets.append([cur_start, n_big, cur_packs])
            self._layers.append(cur_layers)
        # address -> floats of the gradients that their producer already stored as bf16 into `stage_all` (a
        # `WgradQueue.covered_map` whose jobs were given the staging buffer): the pack pass skips them
        self.staged = None
        self.tail = (n_big, self.flat_grad.numel())
        self.active = self.world > 1 or os.environ.get("MI355_FORCE_REDUCER") == "1"
        self._counts = []
        self._pack_tables = {}
        self.exposed = None       # a list while someone measures: (before, after) events around finish()'s wait per step
        self.unpack = unpack
        self.stage_all = torch.zeros(self.flat_grad.numel(), dtype=BF16, device=self.device)
        self._stage = [self.stage_all[s:e] for s, e, _ in self.buckets]
        self._tail_stage = self.stage_all[self.tail[0]:self.tail[1]]
        for bi, (s, e, pks) in enumerate(self.buckets):
            for pk in pks:
                pk.on_grad_ready = (lambda b=bi: self._ready(b))
        self.reset()

    def reset(self):
        self._counts = [len(pks) for _, _, pks in self.buckets]
        self._fired = [False] * len(self.buckets)

    def _pack(self, start, end, layers):
        """fp32 -> bf16 into the staging buffer, over [start, end) minus the layers whose gradients are staged already:
        one table-driven launch of the library (the tables are built once per set of staged layers and kept)."""
        g, st, staged = self.flat_grad, self.stage_all, self.staged
        ranges, cur = [], start
        if staged:
            base = g.data_ptr()
            for o, n in layers:
                if staged.get(base + 4 * o) == n:
                    if o > cur:
                        ranges.append((cur, o))
                    cur = o + n
        if cur < end:
            ranges.append((cur, end))
        if not ranges:
            return
        if not self.use_cuda or any((a % 8) or ((b - a) % 4) for a, b in ranges):
            for a, b in ranges:
                st[a:b].copy_(g[a:b])
            return
        key = (start, end, tuple(ranges))
        table = self._pack_tables.get(key)
        if table is None:
            rows = [[g.data_ptr() + 4 * a, st.data_ptr() + 2 * a, b - a] for a, b in ranges]
            table = self._pack_tables[key] = torch.tensor(rows, dtype=torch.int64).to(self.device)
        longest = max(b - a for a, b in ranges)
        check(lib.mi355_cast_ranges_f32_to_bf16(table.data_ptr(), table.shape[0], max(1, min(1024, longest // 8192)),
                                                _stream()), "cast_ranges")

    def _launch(self, start, end, stage, layers=()):
        g = self.flat_grad[start:end]
        if self.use_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._pack(start, end, layers)
                if self.native is not None:
                    self.native.all_reduce_sum(stage)
                else:
                    dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.group)
                if self.unpack:
                    g.copy_(stage)                               # back to fp32 (the optimizer divides by world)
        else:
            self._pack(start, end, layers)
            dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.group)
            if self.unpack:
                g.copy_(stage)

    def _launch_bucket(self, b):
        s, e, _ = self.buckets[b]
        self._fired[b] = True
        self._launch(s, e, self._stage[b], self._layers[b])

    def _ready(self, b):
        self._counts[b] -= 1
        if self._counts[b] == 0 and self.active and not self._fired[b]:
            self._launch_bucket(b)

    def finish(self):
        """Call after backward: flush buckets that never fired, reduce the tail, join the streams."""
        if self.active:
            # every bucket that has not gone out yet: layers that did not run backward this step, and a trailing bucket
            # that holds no signalling layer at all (e.g. only a channel-padded convolution, whose gradient reaches
            # the buffer through autograd)
            for b in range(len(self.buckets)):
                if not self._fired[b]:
                    self._launch_bucket(b)
            if self.tail[1] > self.tail[0]:
                self._launch(self.tail[0], self.tail[1], self._tail_stage)
            if self.use_cuda:
                cur = torch.cuda.current_stream(self.device)
                if self.exposed is not None:        # measurement: how long the compute stream sits in this wait
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                    cur.wait_stream(self.comm_stream)
                    e1.record(cur)
                    self.exposed.append((e0, e1))
                else:
                    cur.wait_stream(self.comm_stream)
        self.reset()


class FusedAdamW:
    """AdamW over the flat fp32 master buffer in one kernel: global-norm clip, decoupled weight decay,
    bias correction, and the bf16 operand copy written in the same pass.  Hyper-parameters are device
    scalars (no host sync, graph-capturable)."""

    def __init__(self, unet, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=1.0):
        self.unet = unet
        flat, flat16 = unet.flat_parameters()
        # everything, or (frozen base + LoRA adapters) the adapters' contiguous tail of the flat storage
        self.base, end = unet.trainable_range()
        self.frozen_base = self.base != 0
        self.flat, self.flat16 = flat[self.base:end], flat16[self.base:end]
        self.grad = unet.enable_direct_grads()
        assert self.grad.numel() == self.flat.numel()
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.t = 0
        dev = self.flat.device
        self.hyper = torch.zeros(9, dtype=F32, device=dev)
        self.state = torch.zeros(2, dtype=F32, device=dev)
        # partial sums of squares that unsplit weight-gradient jobs add to while they store their results
        self.sq_slots = torch.zeros(int(lib.mi355_wgrad_sumsq_slots()), dtype=F32, device=dev)
        self._sq_tables = {}
        # single-process full fine-tune: the bf16 buffer (laid out like the flat gradient buffer) that unsplit
        # weight-gradient jobs store into instead of the fp32 buffer; the update then reads 2 bytes per gradient
        self.stage16 = None
        from .ops import PinnedRing
        self._hyper_ring = PinnedRing(lambda: torch.zeros(9, dtype=F32).pin_memory(), 8)

    def set_hyper(self, lr=None, grad_prescale=1.0):
        """Stage the per-step scalars (host -> device copy, asynchronous on the current stream)."""
        if lr is not None:
            self.lr = lr
        self.t += 1
        h = self._hyper_ring.acquire()      # ring: an in-flight copy never sees a later step's values
        h[0], h[1], h[2], h[3], h[4] = self.lr, self.betas[0], self.betas[1], self.eps, self.wd
        h[5], h[6] = 1 - self.betas[0] ** self.t, 1 - self.betas[1] ** self.t
        h[7], h[8] = self.max_norm if self.max_norm else 0.0, grad_prescale
        self.hyper.copy_(h, non_blocking=True)
        self._hyper_ring.copied()
        self._hyper_ring.release()

    def _uncovered_table(self, covered):
        """Device table for `mi355_sumsq_ranges`: the parts of the flat gradient buffer outside `covered` (address,
        floats of weight gradients whose sums of squares are already in `sq_slots`), then the slots themselves.
        Tables are kept for the life of the optimizer (a captured step re-reads its table on every replay)."""
        key = tuple(covered)
        hit = self._sq_tables.get(key)
        if hit is not None:
            return hit
        lo = self.grad.data_ptr()
        hi = lo + 4 * self.grad.numel()
        rows, cur = [], lo
        for ptr, cnt in sorted(covered):
            if ptr < cur or ptr + 4 * cnt > hi:
                raise RuntimeError("FusedAdamW: overlapping or out-of-range weight-gradient destinations")
            if ptr > cur:
                rows.append((cur, (ptr - cur) // 4))
            cur = ptr + 4 * cnt
        if cur < hi:
            rows.append((cur, (hi - cur) // 4))
        pack = None
        if self.stage16 is not None:        # the same uncovered ranges, as fp32 -> bf16 cast records into the staging buffer
            if any(((a - lo) // 4) % 8 or c % 4 for a, c in rows):
                raise RuntimeError("FusedAdamW: gradient ranges are not aligned for the bf16 staging pass")
            pack = torch.tensor([[a, self.stage16.data_ptr() + (a - lo) // 2, c] for a, c in rows], dtype=torch.int64)
        rows.append((self.sq_slots.data_ptr(), -self.sq_slots.numel()))
        host = torch.tensor(rows, dtype=torch.int64)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FusedAdamW: the range table of a captured step must be built by an eager step first")
        table = host.to(self.flat.device)
        self._sq_tables[key] = (table, len(rows), None if pack is None else pack.to(self.flat.device))
        return self._sq_tables[key]

    def apply(self, grad16=None, covered=None):
        """The device-side part of the step (capturable).  grad16: bf16 gradients laid out like the flat buffer (the
        all-reduced staging buffer of a data-parallel step) to use instead of the fp32 ones.  covered: (address, floats)
        of the gradients whose sums of squares the weight-gradient launches left in `sq_slots`; the norm pass then reads
        only the rest of the buffer."""
        s = _stream()
        n = self.flat.numel()
        check(lib.mi355_zero_f32(self.state.data_ptr(), 1, s), "zero")     # state[1] keeps counting skipped steps
        if covered and grad16 is None:
            table, rows, pack = self._uncovered_table(covered)
            check(lib.mi355_sumsq_ranges(table.data_ptr(), rows, self.state.data_ptr(), s), "sumsq_ranges")
            if pack is not None:
                # the covered gradients are in the bf16 staging buffer already (stored there by their weight-gradient
                # jobs); cast the rest into it and update from 2-byte gradients: 28 instead of 30 bytes per parameter
                check(lib.mi355_cast_ranges_f32_to_bf16(pack.data_ptr(), pack.shape[0], 256, s), "cast_ranges")
                check(lib.mi355_adamw_step_g16(self.flat.data_ptr(), self.stage16.data_ptr(), self.m.data_ptr(),
                                               self.v.data_ptr(), self.flat16.data_ptr(), self.hyper.data_ptr(),
                                               self.state.data_ptr(), n, s), "adamw_g16")
            else:
                check(lib.mi355_adamw_step(self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                           self.flat16.data_ptr(), self.hyper.data_ptr(), self.state.data_ptr(), n, s), "adamw")
            if not self.frozen_base:
                self.unet.refresh_operands(cast=False)
            return
        if grad16 is not None and n % 8 == 0:
            assert grad16.numel() == n and grad16.dtype == BF16
            # fixed order of additions: every rank derives the same clip factor from the same reduced gradients
            check(lib.mi355_sumsq_bf16_ordered(grad16.data_ptr(), self.state.data_ptr(), self.sq_slots.data_ptr(),
                                               self.sq_slots.numel(), n, s), "sumsq_bf16_ordered")
            check(lib.mi355_adamw_step_g16(self.flat.data_ptr(), grad16.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                           self.flat16.data_ptr(), self.hyper.data_ptr(), self.state.data_ptr(), n, s),
                  "adamw_g16")
        else:
            if grad16 is not None:
                self.grad.copy_(grad16)
            check(lib.mi355_sumsq_f32(self.grad.data_ptr(), self.state.data_ptr(), n, s), "sumsq")
            check(lib.mi355_adamw_step(self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                       self.flat16.data_ptr(), self.hyper.data_ptr(), self.state.data_ptr(), n, s), "adamw")
        if not self.frozen_base:          # a frozen base keeps its operand copies; the adapters are read as fp32 masters
            self.unet.refresh_operands(cast=False)

    def step(self, lr=None, grad_prescale=1.0, grad16=None, covered=None):
        self.set_hyper(lr, grad_prescale)
        self.apply(grad16, covered)

    def full_grad(self, covered=None):
        """The last step's gradient as one fp32 vector laid out like the flat parameter buffer (a copy; for inspection
        and tests).  With bf16 staging on, the gradients of the `covered` layers (address, floats; the trainer's
        `_covered`) exist only in `stage16`: they are read from there, everything else from the fp32 buffer."""
        g = self.grad.clone()
        if self.stage16 is not None and covered:
            lo = self.grad.data_ptr()
            for ptr, cnt in covered:
                o = (ptr - lo) // 4
                g[o:o + cnt] = self.stage16[o:o + cnt].float()
        return g

    def skipped_steps(self):
        """Host read (synchronises) of how many updates the non-finite guard has skipped so far.  The reference prints
        `NaN detected` and skips the batch (trainer.py:251-253); here the skip happens on the device and this counter
        is its trace.  A skipped step still advances the bias-correction step count `t` (torch's GradScaler would not
        advance the optimizer step): after k skips the corrections are those of step t instead of t - k, a difference
        that vanishes as beta^t does."""
        return int(self.state[1])

    def grad_norm(self):
        """Host read of the last step's pre-clip global gradient norm (synchronises)."""
        return float(self.state[0].sqrt() * self.hyper[8])

    # ---- torch.optim.AdamW-shaped state (what `optimizer.state_dict()` holds in the reference checkpoint,
    # trainer.py:139-153): per-parameter `step`, `exp_avg`, `exp_avg_sq`, indexed in `unet.parameters()` order
    def _param_views(self, flat):
        views = []
        for p in self.unet.parameters():
            if self.frozen_base and not p.requires_grad:
                continue
            o, n = self.unet._offs[id(p)] - self.base, p.numel()
            if p.dim() == 4:
                co, ci, kh, kw = p.shape
                views.append(flat[o:o + n].view(co, kh, kw, ci).permute(0, 3, 1, 2))
            else:
                views.append(flat[o:o + n].view(p.shape))
        return views

    def state_dict(self):
        ms, vs = self._param_views(self.m), self._param_views(self.v)
        state = {i: {"step": torch.tensor(float(self.t)), "exp_avg": m.detach().cpu().contiguous(),
                     "exp_avg_sq": v.detach().cpu().contiguous()} for i, (m, v) in enumerate(zip(ms, vs))}
        # the installed torch's own AdamW supplies the param-group keys, so the file loads into a torch optimizer too
        group = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(()))], lr=self.lr, betas=tuple(self.betas), eps=self.eps,
                                  weight_decay=self.wd).state_dict()["param_groups"][0]
        group["params"] = list(range(len(ms)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        ms, vs = self._param_views(self.m), self._param_views(self.v)
        if len(sd["state"]) not in (0, len(ms)):
            raise ValueError(f"optimizer state has {len(sd['state'])} entries, the model has {len(ms)} parameters")
        steps = set()
        for i, st in sd["state"].items():
            i = int(i)
            if tuple(st["exp_avg"].shape) != tuple(ms[i].shape):
                raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} != parameter {tuple(ms[i].shape)}")
            ms[i].copy_(st["exp_avg"])
            vs[i].copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("optimizer state with per-parameter step counts is not supported (one fused step counter)")
        self.t = steps.pop() if steps else 0
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.wd = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]


class DenoiserTrainer:
    """One data-parallel denoiser train step: add-noise -> UNet -> fused MSE -> backward with overlapped
    bucketed all-reduce -> fused clip + AdamW.  `latents` play the role of the VAE output times 0.18215
    (image_generator.py:107-109) and `ctx` the text-encoder states (image_generator.py:121)."""

    def __init__(self, unet, lr=1e-4, weight_decay=1e-2, max_grad_norm=1.0, alphas_cumprod=None, bucket_mb=64,
                 group=None):
        self.unet = unet
        self.opt = FusedAdamW(unet, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.reducer = None
        if self.world > 1 or os.environ.get("MI355_FORCE_REDUCER") == "1":
            n_big = 0 if self.opt.frozen_base else unet._n_big       # adapters only: one small tail bucket
            # MI355_DP_NATIVE=1: all-reduce through the library's own communicator (enqueued on the reducer's stream
            # directly) instead of the torch process group
            native = NativeComm.get(group) if (os.environ.get("MI355_DP_NATIVE") == "1" and unet_on_gpu(unet)) else None
            self.reducer = GradBucketReducer(unet.enable_direct_grads(), n_big, unet._grad_packs, group=group,
                                             bucket_mb=bucket_mb, native=native)
        dev = self.opt.flat.device
        if alphas_cumprod is None:
            from .schedulers import sd_alphas_cumprod
            alphas_cumprod = sd_alphas_cumprod()
        self.acp = alphas_cumprod.to(device=dev, dtype=F32).contiguous()
        self.base_lr = lr
        from . import ops as _ops
        self._ops = _ops
        # The grouped launches stay on the main stream: overlapping them with the rest of the backward pass on a
        # side stream measured 45.3 ms/step against 40.0 (their long workgroups starve the dependent chain).
        side = self.reducer is None and os.environ.get("MI355_WGRAD_SIDE") == "1"
        # Flush size: with a reducer the gradients must become ready in many portions for the all-reduces to run under
        # the backward pass (4 096 blocks, about nine flushes per step); without one, every flush only adds a launch
        # tail, so a single-GPU step flushes four times less often (34.39 against 34.58 ms per step on one box; beyond
        # 16 384 blocks nothing more is gained).  MI355_WGRAD_FLUSH overrides both.
        flush = int(os.environ.get("MI355_WGRAD_FLUSH", "4096" if self.reducer is not None else "16384"))
        self.wgrad_queue = (_ops.WgradQueue(dev, flush_blocks=flush, side_stream=side)
                            if os.environ.get("MI355_NO_WGRAD_QUEUE") != "1" else None)
        self._flush_blocks = flush
        if self.wgrad_queue is not None:
            self.wgrad_queue.zero_split_outputs = True
        # norm-layer parameter gradients: one batched reduction per step instead of one launch per layer
        self.reduce_queue = _ops.ReduceQueue(dev) if os.environ.get("MI355_NO_REDUCE_QUEUE") != "1" else None
        # single-process full fine-tune: unsplit weight-gradient jobs sum their squares while they store, and the
        # optimizer's norm pass skips those gradients (most of the 3.4 GB buffer)
        self._fused_sumsq = (self.wgrad_queue is not None and self.reducer is None and not self.opt.frozen_base
                             and os.environ.get("MI355_NO_FUSED_SUMSQ") != "1")
        self._covered = None
        # ... and store their gradients as bf16 into a staging buffer laid out like the flat gradient buffer (what the
        # data-parallel step all-reduces and the reference's own reduce_dtype, trainer.py:86-90, is 16-bit too): the
        # weight-gradient launches write, and the optimizer reads, 2 bytes per gradient instead of 4.  `param.grad` of
        # those layers is then not materialised (MI355_NO_DIRECT_STAGE=1 restores the fp32 path).
        if self._fused_sumsq and os.environ.get("MI355_NO_DIRECT_STAGE") != "1" and self.opt.flat.numel() % 8 == 0:
            self.opt.stage16 = torch.zeros(self.opt.flat.numel(), dtype=BF16, device=dev)
        # data-parallel full fine-tune: unsplit weight-gradient jobs store bf16 straight into the reducer's staging
        # buffer, so the fp32 -> bf16 pack pass touches only what is left (the fp32 buffer and `param.grad` do not
        # hold those gradients then)
        self._direct_stage = (self.wgrad_queue is not None and self.reducer is not None and self.reducer.active
                              and self.reducer.use_cuda and not self.reducer.unpack and not self.opt.frozen_base
                              and os.environ.get("MI355_NO_DIRECT_STAGE") != "1")

    def _forward_loss(self, latents, noise, t, ctx):
        B = latents.shape[0]
        noisy = torch.empty_like(latents)
        check(lib.mi355_ddpm_add_noise(latents.data_ptr(), noise.data_ptr(), t.data_ptr(), self.acp.data_ptr(),
                                       noisy.data_ptr(), B, latents[0].numel(), _stream()), "ddpm_add_noise")
        pred = self.unet(noisy, t, encoder_hidden_states=ctx).sample
        return mse_loss(pred, noise)

    def forward_backward(self, latents, noise, t, ctx, loss_fn=None):
        """loss_fn: the caller builds the loss from the denoiser itself (see `train_step_from`)."""
        if loss_fn is not None and self._fwd_graph is not None:
            raise RuntimeError("forward_backward(loss_fn=...) runs the eager forward pass; drop the captured one first")
        if self._fwd_graph is not None:
            for dst, src in zip(self._static, (latents, noise, t, ctx)):
                if dst.data_ptr() != src.data_ptr():
                    # through .data: the recorded tape saved views of these buffers, and a counted in-place write
                    # would make autograd refuse them - the graph replay rewrites everything else it saved anyway
                    dst.data.copy_(src, non_blocking=True)
        # with the queue, pixel-split weight gradients are cleared per job and unsplit ones are plain stores
        self.unet.zero_small_grads(weights_too=self.wgrad_queue is None)
        fused = self._fused_sumsq and self.wgrad_queue is not None
        self._covered = None
        if fused:
            g = self.opt.grad
            check(lib.mi355_zero_f32(self.opt.sq_slots.data_ptr(), self.opt.sq_slots.numel(), _stream()), "zero")
            self.wgrad_queue.track_unsplit(g.data_ptr(), g.data_ptr() + 4 * g.numel(), slots=self.opt.sq_slots,
                                           stage16=self.opt.stage16)
        staged = self._direct_stage and self.wgrad_queue is not None
        if staged:
            g = self.opt.grad
            self.wgrad_queue.track_unsplit(g.data_ptr(), g.data_ptr() + 4 * g.numel(), stage16=self.reducer.stage_all)
            self.reducer.staged = self.wgrad_queue.covered_map
        self._ops.set_direct_step(True)
        try:
            if self._fwd_graph is not None:
                self._fwd_graph.replay()                       # refreshes every tensor the recorded tape saved
                loss = self._fwd_loss
            elif loss_fn is not None:
                loss = loss_fn()
            else:
                loss = self._forward_loss(latents, noise, t, ctx)
            self._ops.set_wgrad_queue(self.wgrad_queue)        # weight gradients are deferred and batched
            self._ops.set_reduce_queue(self.reduce_queue)
            loss.backward(retain_graph=self._fwd_graph is not None)
            if self.wgrad_queue is not None:
                self.wgrad_queue.join()
                self._covered = list(self.wgrad_queue.covered) if fused else None
            if self.reduce_queue is not None:
                self.reduce_queue.flush()
        except BaseException:
            # a step that dies half-way (the reference's loop catches per-batch exceptions and carries on,
            # trainer.py:277-280) must not leave half-counted buckets or queued jobs for the next one
            if self.wgrad_queue is not None:
                self.wgrad_queue.discard()
                self.wgrad_queue.track_unsplit()
            if self.reduce_queue is not None:
                self.reduce_queue.discard()
            if self.reducer is not None:
                self.reducer.reset()
                self.reducer.staged = None
            raise
        finally:
            if fused:
                self.wgrad_queue.track_unsplit()
            self._ops.set_wgrad_queue(None)
            self._ops.set_reduce_queue(None)
            self._ops.set_direct_step(False)
        self.unet.finish_grads()
        if self.reducer is not None:
            self.reducer.finish()
            self.reducer.staged = None
        if staged:
            self.wgrad_queue.track_unsplit()
        return loss.detach()

    def full_grad(self):
        """fp32 copy of the last step's whole gradient (see FusedAdamW.full_grad)."""
        return self.opt.full_grad(self._covered)

    def exchange_summary(self):
        """What the gradient exchange of a data-parallel step looks like (None on a single process): bucket count and
        bf16 bytes per all-reduce, the tail bucket of small parameters, the weight-gradient flush size that decides in how
        many portions the gradients become ready, and which communicator carries the all-reduces."""
        r = self.reducer
        if r is None:
            return None
        sizes = [2 * (e - s0) for s0, e, _ in r.buckets]
        return {"buckets": len(sizes), "bucket_bytes_bf16": {"min": min(sizes) if sizes else 0, "max": max(sizes) if sizes else 0,
                                                             "total": sum(sizes)},
                "tail_bytes_bf16": 2 * (r.tail[1] - r.tail[0]), "wgrad_flush_blocks": self._flush_blocks,
                "communicator": "library (mi355_dp_*)" if r.native is not None else "torch process group",
                "world": r.world}

    def train_step(self, latents, noise, t, ctx, epoch_float=None):
        lr = None if epoch_float is None else cosine_warm_restarts_lr(self.base_lr, epoch_float)
        if self._graph is not None:
            return self._replay(latents, noise, t, ctx, lr)
        loss = self.forward_backward(latents, noise, t, ctx)
        reduced = self.reducer.stage_all if (self.reducer is not None and self.reducer.active) else None
        self.opt.step(lr=lr, grad_prescale=1.0 / self.world, grad16=reduced, covered=self._covered)
        self.step_count += 1
        return loss

    def train_step_from(self, loss_fn, epoch_float=None):
        """One step on a loss the caller builds from this trainer's denoiser: the reference's loop body
        (trainer.py:243-261) with `loss_fn = lambda: model.train_step(images, raw_text)` - pixels through the frozen
        auto-encoder, captions through tokenizer and text encoder, noise and timesteps drawn inside
        (image_generator.py:103-126) - followed by this trainer's backward pass (deferred, grouped weight gradients;
        bucketed all-reduce when data-parallel), global-norm clip and fused AdamW.  Eager: the captured step takes
        latents and text states (`train_step`)."""
        if self._graph is not None:
            raise RuntimeError("train_step_from(): this trainer replays a captured step on latents; build another trainer")
        lr = None if epoch_float is None else cosine_warm_restarts_lr(self.base_lr, epoch_float)
        loss = self.forward_backward(None, None, None, None, loss_fn=loss_fn)
        reduced = self.reducer.stage_all if (self.reducer is not None and self.reducer.active) else None
        self.opt.step(lr=lr, grad_prescale=1.0 / self.world, grad16=reduced, covered=self._covered)
        self.step_count += 1
        return loss

    # ---- files (reference trainer.py:123-189): `weights/<name>` holds unet.state_dict(), `checkpoints/<name>` the
    # {"model", "optimizer", "step", "loss", "epoch"} dictionary; rank 0 writes, every rank loads
    step_count = 0
    best_loss = float("inf")

    def save_weights(self, output_dir, name="unet.pth"):
        if dist.is_initialized() and dist.get_rank() != 0:
            return None
        d = os.path.join(str(output_dir), "weights")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, name)
        torch.save({k: v.detach().cpu() for k, v in self.unet.state_dict().items()}, path)
        return path

    def save_checkpoint(self, output_dir, epoch, name="checkpoint.pth"):
        if dist.is_initialized() and dist.get_rank() != 0:
            return None
        d = os.path.join(str(output_dir), "checkpoints")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, name)
        torch.save({"model": {k: v.detach().cpu() for k, v in self.unet.state_dict().items()},
                    "optimizer": self.opt.state_dict(), "step": self.step_count, "loss": self.best_loss,
                    "epoch": epoch}, path)
        print(f"[INFO] Saved checkpoint to {path}")
        return path

    def load_checkpoint(self, path):
        """Returns the epoch to resume from, or None (with the reference's error message) when the file is missing
        or does not fit the model.  Tensors only: the file is read with `weights_only=True`."""
        if not os.path.exists(path):
            print(f"[ERROR] Checkpoint file {path} not found!")
            return None
        try:
            ck = torch.load(path, map_location="cpu", weights_only=True)
            self.unet.load_state_dict(ck["model"])
            self.opt.load_state_dict(ck["optimizer"])
            self.unet.refresh_operands()
            self.step_count = ck["step"] + 1
            self.best_loss = ck["loss"]
            self._graph = None                      # a captured step holds the old step counter's hyper-parameters
            self._fwd_graph = self._fwd_loss = None
            print(f"[INFO] Resumed from {path} | epoch: {ck['epoch']} | step: {self.step_count} | loss: {self.best_loss}")
            return ck["epoch"]
        except Exception as e:                      # the reference reports and carries on (trainer.py:186-189)
            print(f"[ERROR] Failed to load checkpoint: {e}")
            return None

    # ---- whole-step HIP graph (static shapes): forward, backward and optimizer replayed as one graph,
    # per-step scalars (lr, bias corrections) are staged into a device buffer before each replay
    _graph = None

    def capture(self, latents, noise, t, ctx):
        """Capture one full step for these shapes.  Two un-captured warm-up steps run first with lr = 0 and no weight
        decay (weights unchanged); the optimizer's moments and step count are put back afterwards, so capturing after
        `load_checkpoint()` or after eager steps resumes with the AdamW state intact, as the reference's resume does
        (trainer.py:176-181)."""
        if self.world > 1:
            raise RuntimeError("capture(): the graphed step is single-process; multi-GPU runs the eager step")
        self._static = tuple(x.clone() for x in (latents, noise, t, ctx))
        opt = self.opt
        keep = (opt.lr, opt.wd)
        saved_t = opt.t
        # fresh optimizer: the moments are zero and need no copy (2 x 3.4 GB for the full model otherwise)
        saved_mv = None if saved_t == 0 else (opt.m.clone(), opt.v.clone())
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):
                opt.lr, opt.wd = 0.0, 0.0
                self.forward_backward(*self._static)
                opt.step(lr=0.0, grad_prescale=1.0, covered=self._covered)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        opt.lr, opt.wd = keep
        if saved_mv is None:
            opt.m.zero_()
            opt.v.zero_()
        else:
            opt.m.copy_(saved_mv[0])
            opt.v.copy_(saved_mv[1])
            del saved_mv
        opt.t = saved_t
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._static_loss = self.forward_backward(*self._static)
            opt.apply(covered=self._covered)
        self._graph = g

    # ---- forward-only HIP graph for the data-parallel step
    _fwd_graph = None

    def capture_forward(self, latents, noise, t, ctx):
        """The forward pass (add-noise, denoiser, loss) of the eager step as one HIP graph for these shapes.  The backward
        pass stays eager - its bucket all-reduces are issued from callbacks as gradients complete, and no collective is
        ever captured - and walks the autograd tape recorded here again every step (`retain_graph`): the tensors the
        tape saved are the graph's own buffers, which each replay fills with the current step's activations.  Takes the
        forward's third of the host's enqueue work out of a step whose host and device times are close (one process
        per GPU has no idle device time to hide a slow host behind).  Call after at least one eager step."""
        if self._graph is not None:
            raise RuntimeError("capture_forward(): the whole step is already captured")
        self.unet._maybe_refresh()                   # (an operand refresh must not end up inside the graph)
        self._static = tuple(x.clone() for x in (latents, noise, t, ctx))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        self._ops.set_direct_step(True)
        try:
            # thread-local capture mode: the process group's watchdog thread may query events meanwhile
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                loss = self._forward_loss(*self._static)
        finally:
            self._ops.set_direct_step(False)
        self._fwd_graph, self._fwd_loss = g, loss

    def _replay(self, latents, noise, t, ctx, lr):
        for dst, src in zip(self._static, (latents, noise, t, ctx)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.opt.set_hyper(lr, grad_prescale=1.0)
        self._graph.replay()
        self.step_count += 1
        return self._static_loss


def unet_on_gpu(unet):
    return next(unet.parameters()).device.type == "cuda"


class NoDeviceForRank(RuntimeError):
    pass


def init_distributed():
    """One process per GPU (torchrun env): RCCL process group, returns (rank, world, local_rank).  A LOCAL_RANK without a
    visible GPU of its own raises `NoDeviceForRank` before any group is created (two ranks never share a device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    have = torch.cuda.device_count()
    if have and local >= have:
        raise NoDeviceForRank(f"rank {rank}: LOCAL_RANK {local} but only {have} GPU(s) are visible to this process")
    if (world > 1 or os.environ.get("MI355_FORCE_REDUCER") == "1") and "RANK" in os.environ and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local
