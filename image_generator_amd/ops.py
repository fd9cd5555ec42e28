"""Tensor-level wrappers over the C ABI plus the torch.autograd.Function glue.

PyTorch owns every buffer (allocation, streams, autograd bookkeeping); all arithmetic is done by
libmi355_denoiser.so.  Activations are bf16, token-major / NHWC.
"""
from __future__ import annotations

import ctypes
import os
import math

import torch

from ._lib import lib, check

BF16 = torch.bfloat16
F32 = torch.float32

_WORKSPACE = {}
WORKSPACE_BYTES = 256 << 20


def _stream():
    # raw hipStream_t of the current stream of the current device (the Python Stream object costs ~9 us per call,
    # which adds up over the ~2000 launches of a training step)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _ptr(t):
    return None if t is None else t.data_ptr()


def workspace(device):
    ws = _WORKSPACE.get(device)
    if ws is None:
        ws = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
        _WORKSPACE[device] = ws
    return ws


def _req(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: HIP device tensor required (no CPU path)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: must be contiguous")
    return t


# ------------------------------------------------------------------------------------------------
# raw calls
# ------------------------------------------------------------------------------------------------
def gemm(a, w, bias=None, rowbias=None, rowbias_div=1, residual=None, out_fp32=False, alpha=1.0,
         lora_t=None, lora_b=None, lora_alpha=0.0, out=None, lora_seg=0, lora_rmajor=False):
    """out[M,N] = alpha * a[M,K] @ w[N,K]^T (+ bias + rowbias + lora + residual).  a may have a row stride.
    lora_t fp32 [M, r] (or [M, S*r] with lora_seg = N/S for S row-stacked adapters) or bf16 [M, 8 | 16];
    lora_b [N, r] (or [r, N] if lora_rmajor)."""
    assert a.dtype == BF16 and w.dtype == BF16 and a.is_cuda
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1      # w may be a column slice of a wider matrix
    if out is None:
        out = torch.empty((M, N), dtype=F32 if out_fp32 else BF16, device=a.device)
    ws = workspace(a.device)
    r = ldt = t16 = 0
    if lora_t is not None:
        assert lora_b.dtype == F32 and lora_t.is_contiguous() and lora_b.is_contiguous()
        r = lora_b.shape[0] if lora_rmajor else lora_b.shape[1]
        ldt = lora_t.shape[1]
        t16 = int(lora_t.dtype == BF16)         # bf16 [M, 8 | 16]: zero / don't-care columns beyond the ranks in use
        assert t16 or lora_t.dtype == F32
        need = r * (N // lora_seg) if lora_seg else r
        assert lora_t.shape[0] == M and lora_b.numel() == N * r and (ldt in (8, 16) and ldt >= need if t16 else ldt == need)
    check(lib.mi355_gemm_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), M, N, K, out.data_ptr(), out.stride(0),
                              1 if out.dtype == F32 else 0, _ptr(bias), _ptr(rowbias), rowbias_div,
                              0 if rowbias is None else rowbias.stride(0), _ptr(residual),
                              0 if residual is None else residual.stride(0), alpha, _ptr(lora_t), ldt, _ptr(lora_b), r,
                              lora_seg, int(lora_rmajor), t16, lora_alpha, ws.data_ptr(), ws.numel(), _stream()), "gemm")
    return out


def gemm_cat(a1, a2, w, bias=None):
    """bf16 [M, N] = [a1 | a2] @ w[N, K1 + K2]^T (+ bias) with the concatenation read in place."""
    M, K1 = a1.shape
    K2 = a2.shape[1]
    N = w.shape[0]
    assert a1.dtype == BF16 and a2.dtype == BF16 and w.dtype == BF16 and a1.is_contiguous() and a2.is_contiguous()
    assert a2.shape[0] == M and w.shape[1] == K1 + K2 and w.stride(1) == 1
    out = torch.empty((M, N), dtype=BF16, device=a1.device)
    ws = workspace(a1.device)
    check(lib.mi355_gemm_cat_bf16(a1.data_ptr(), K1, a2.data_ptr(), K2, w.data_ptr(), w.stride(0), M, N, out.data_ptr(), N,
                                  _ptr(bias), ws.data_ptr(), ws.numel(), _stream()), "gemm_cat")
    return out


def gemm_split(a, w, n1):
    """(bf16 [M, n1], bf16 [M, N - n1]) = column blocks of a @ w[N, K]^T, written as two tensors."""
    M, K = a.shape
    N = w.shape[0]
    assert a.dtype == BF16 and w.dtype == BF16 and a.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K
    o1 = torch.empty((M, n1), dtype=BF16, device=a.device)
    o2 = torch.empty((M, N - n1), dtype=BF16, device=a.device)
    ws = workspace(a.device)
    check(lib.mi355_gemm_split_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), M, n1, N - n1, K, o1.data_ptr(),
                                    o2.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "gemm_split")
    return o1, o2


def gemm_geglu_bwd(dout, w_t, pre):
    """bf16 [M, 2F] = GEGLU backward of (dout[M, K] @ w_t[F, K]^T, pre[M, 2F]): the [M, F] product is never written."""
    M, K = dout.shape
    F = w_t.shape[0]
    assert dout.dtype == BF16 and w_t.dtype == BF16 and pre.dtype == BF16 and dout.stride(1) == 1 and w_t.stride(1) == 1
    assert w_t.shape[1] == K and pre.shape == (M, 2 * F) and pre.is_contiguous()
    dpre = torch.empty_like(pre)
    ws = workspace(dout.device)
    check(lib.mi355_gemm_geglu_bwd_bf16(dout.data_ptr(), dout.stride(0), w_t.data_ptr(), w_t.stride(0), M, F, K, pre.data_ptr(),
                                        dpre.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "gemm_geglu_bwd")
    return dpre


def gemm_bt(a, w, n1=0, pre=None):
    """a[M, K] @ w[K, N] with w read as stored (the data gradient of a linear layer from its [out, in] weight).
    n1 > 0: returns the column blocks (bf16 [M, n1], bf16 [M, N - n1]) as two tensors.  pre (bf16 [M, 2N]): returns the
    GEGLU backward [M, 2N] of the product instead of the product (see gemm_geglu_bwd)."""
    M, K = a.shape
    N = w.shape[1]
    assert a.dtype == BF16 and w.dtype == BF16 and a.stride(1) == 1 and w.stride(1) == 1 and w.shape[0] == K
    ws = workspace(a.device)
    out = out2 = dpre = None
    if pre is not None:
        assert pre.dtype == BF16 and pre.shape == (M, 2 * N) and pre.is_contiguous() and not n1
        dpre = torch.empty_like(pre)
    elif n1:
        out = torch.empty((M, n1), dtype=BF16, device=a.device)
        out2 = torch.empty((M, N - n1), dtype=BF16, device=a.device)
    else:
        out = torch.empty((M, N), dtype=BF16, device=a.device)
    check(lib.mi355_gemm_bt_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), M, N, K, _ptr(out), N, _ptr(out2), n1,
                                 _ptr(pre), _ptr(dpre), ws.data_ptr(), ws.numel(), _stream()), "gemm_bt")
    return dpre if pre is not None else ((out, out2) if n1 else out)


def conv2d_dgrad_bt(dy, w, cin, ksize, dil=False, out_hw=None):
    """Data gradient [B, Ho, Wo, cin] of a 3x3 / 1x1 convolution from dy [B, H, W, Cout] and the weight as stored,
    w [Cout][k][k][cin] (dil: the layer had stride 2; out_hw its input size)."""
    B, H, W, Co = dy.shape
    Ho, Wo = out_hw if out_hw is not None else (H, W)
    assert dy.dtype == BF16 and w.dtype == BF16 and dy.is_contiguous() and w.is_contiguous()
    dx = torch.empty((B, Ho, Wo, cin), dtype=BF16, device=dy.device)
    ws = workspace(dy.device)
    check(lib.mi355_conv2d_dgrad_bt_bf16(dy.data_ptr(), B, H, W, Co, w.data_ptr(), cin, ksize, int(dil), Ho, Wo, dx.data_ptr(),
                                         ws.data_ptr(), ws.numel(), _stream()), "conv2d_dgrad_bt")
    return dx


def gemm_rowstats(a, w, bias=None, residual=None):
    """(out, rowstat, parts): out = a w^T + bias + residual as `gemm`, and rowstat [M][parts][2] fp32 = per row and column
    tile the sum and sum of squares of the bf16 values stored - the statistics a LayerNorm of `out` needs, produced by
    the launch that writes `out` (see mi355_gemm_rowstats_bf16).  N % 160 == 0, K % 64 == 0."""
    import ctypes
    _req(a, BF16, "a"); _req(w, BF16, "w")
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=BF16, device=a.device)
    stat = torch.empty((M, N // 160, 2), dtype=F32, device=a.device)
    parts = ctypes.c_int(0)
    ws = workspace(a.device)
    check(lib.mi355_gemm_rowstats_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), M, N, K, out.data_ptr(), N,
                                       _ptr(bias), _ptr(residual), 0 if residual is None else residual.stride(0),
                                       stat.data_ptr(), ctypes.byref(parts), ws.data_ptr(), ws.numel(), _stream()), "gemm_rowstats")
    return out, stat.view(-1)[:M * parts.value * 2].view(M, parts.value, 2), parts.value


def ln_fold_weight(weight, gamma, beta, bias=None, out=None, scale=1.0):
    """(wp, colsum, bias_out) for `gemm_ln`: the fp32 weight [N][K] of the layer behind a LayerNorm with that norm's
    gamma / beta folded in (see mi355_ln_fold_weight).  out: the three destinations (row slices of larger buffers when
    several layers share the norm and run as one product).  scale: a factor on the layer's output folded in too."""
    N, K = weight.shape
    assert weight.dtype == F32 and weight.stride(1) == 1
    if out is not None:
        wp, cs, bo = out
        assert wp.shape == (N, K) and wp.is_contiguous() and cs.numel() == N and bo.numel() == N
    else:
        wp = torch.empty((N, K), dtype=BF16, device=weight.device)
        cs = torch.empty(N, dtype=F32, device=weight.device)
        bo = torch.empty(N, dtype=F32, device=weight.device)
    check(lib.mi355_ln_fold_weight(weight.data_ptr(), weight.stride(0), gamma.data_ptr(), beta.data_ptr(), _ptr(bias),
                                   wp.data_ptr(), cs.data_ptr(), bo.data_ptr(), N, K, float(scale), _stream()), "ln_fold_weight")
    return wp, cs, bo


def gemm_ln(a, wp, colsum, bias_out, rowstat, parts, eps, geglu=False):
    """LayerNorm(a) W^T + b from the un-normalised a [M][K], the folded weight triple of `ln_fold_weight` and the row
    statistics `gemm_rowstats` emitted with a; geglu: the gated form (N = 2F weight rows, out [M][F])."""
    M, K = a.shape
    N = wp.shape[0]
    out = torch.empty((M, N // 2 if geglu else N), dtype=BF16, device=a.device)
    ws = workspace(a.device)
    check(lib.mi355_gemm_ln_bf16(a.data_ptr(), a.stride(0), wp.data_ptr(), wp.stride(0), M, N, K, out.data_ptr(), out.shape[1],
                                 bias_out.data_ptr(), rowstat.data_ptr(), parts, colsum.data_ptr(), eps, int(geglu),
                                 ws.data_ptr(), ws.numel(), _stream()), "gemm_ln")
    return out


def gemm_skinny(a, w, alpha=1.0):
    """bf16 [M, W] = alpha * a[M, K] @ w[W, K]^T for W in (8, 16): one launch, no split-K workspace."""
    M, K = a.shape
    W = w.shape[0]
    assert a.dtype == BF16 and w.dtype == BF16 and a.stride(1) == 1 and w.is_contiguous() and w.shape[1] == K
    out = torch.empty((M, W), dtype=BF16, device=a.device)
    check(lib.mi355_gemm_skinny_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), M, W, K, out.data_ptr(), alpha, _stream()),
          "gemm_skinny")
    return out


def gemm_geglu(a, w, bias=None, keep_preact=False):
    """out[M,F] = (a @ w[:F]^T + b[:F]) * gelu(a @ w[F:]^T + b[F:]) in one launch.
    keep_preact: also return the bf16 [M, 2F] pre-activation (hidden | gate) for a backward pass."""
    M, K = a.shape
    F2 = w.shape[0]
    assert a.dtype == BF16 and w.dtype == BF16 and a.stride(1) == 1 and w.is_contiguous() and w.shape[1] == K
    out = torch.empty((M, F2 // 2), dtype=BF16, device=a.device)
    pre = torch.empty((M, F2), dtype=BF16, device=a.device) if keep_preact else None
    check(lib.mi355_gemm_geglu_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), K, M, F2 // 2, K, out.data_ptr(), F2 // 2,
                                    _ptr(bias), _ptr(pre), _stream()), "gemm_geglu")
    return (out, pre) if keep_preact else out


# Statistics of a following GroupNorm that the last convolution / product emitted with its output: (sums tensor
# [B * tiles][channels / unit][2], row tiles per sample, unit), or None.  The module that ran the layer attaches them to
# the output tensor (`take_gn_pre`), the GroupNorm that reads that tensor skips its statistics pass.
_LAST_GN_PRE = None
_GN_PRE_ON = os.environ.get("MI355_NO_GN_PRE") != "1"


def take_gn_pre():
    global _LAST_GN_PRE
    pre, _LAST_GN_PRE = _LAST_GN_PRE, None
    return pre


def _emit_gn_pre(gstat, rows, rows_per_sample, unit, M, N):
    """rows: output rows per row tile the launch emitted statistics for (0: none); the buffer was sized for 128-row tiles."""
    global _LAST_GN_PRE
    _LAST_GN_PRE = (gstat[:(M // rows) * (N // unit) * 2], rows_per_sample // rows, unit) if rows > 0 else None


def gemm_gnstats(a, w, bias, residual, rows_per_sample, unit):
    """`gemm` with bias / residual whose epilogue also emits the statistics of a following GroupNorm (see
    mi355_gemm_gnstats_bf16); they are left for `take_gn_pre` when the launch could emit them."""
    import ctypes
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=BF16, device=a.device)
    gstat = torch.empty(((M + 127) // 128) * (N // unit) * 2, dtype=F32, device=a.device)
    rows = ctypes.c_int(0)
    ws = workspace(a.device)
    check(lib.mi355_gemm_gnstats_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), M, N, K, out.data_ptr(), N, _ptr(bias),
                                      _ptr(residual), 0 if residual is None else residual.stride(0), rows_per_sample,
                                      gstat.data_ptr(), unit, ctypes.byref(rows), ws.data_ptr(), ws.numel(), _stream()), "gemm_gnstats")
    _emit_gn_pre(gstat, rows.value, rows_per_sample, unit, M, N)
    return out


def conv2d(x, w, cout, ksize, stride=1, ups=False, dil=False, out_hw=None, bias=None, rowbias=None, residual=None,
           out_fp32=False, gn_unit=0):
    """x [B,H,W,C] bf16 NHWC; w physical [Cout][k][k][C] bf16 -> [B,Ho,Wo,Cout].
    gn_unit > 0: also emit the statistics of a following GroupNorm (see `take_gn_pre`)."""
    B, H, W, C = x.shape
    assert x.dtype == BF16 and x.is_contiguous()
    pad = ksize // 2
    if out_hw is None:
        hv, wv = (2 * H, 2 * W) if ups else (H, W)
        out_hw = ((hv + 2 * pad - ksize) // stride + 1, (wv + 2 * pad - ksize) // stride + 1)
    Ho, Wo = out_hw
    out = torch.empty((B, Ho, Wo, cout), dtype=F32 if out_fp32 else BF16, device=x.device)
    ws = workspace(x.device)
    if gn_unit and _GN_PRE_ON and not out_fp32 and not dil and cout % gn_unit == 0 and (Ho * Wo) % 128 == 0:
        import ctypes
        gstat = torch.empty((B * Ho * Wo // 128) * (cout // gn_unit) * 2, dtype=F32, device=x.device)
        rows = ctypes.c_int(0)
        check(lib.mi355_conv2d_gnstats_bf16(x.data_ptr(), B, H, W, C, w.data_ptr(), cout, ksize, stride, int(ups), Ho, Wo,
                                            out.data_ptr(), _ptr(bias), _ptr(rowbias), 0 if rowbias is None else rowbias.stride(0),
                                            _ptr(residual), gstat.data_ptr(), gn_unit, ctypes.byref(rows), ws.data_ptr(), ws.numel(),
                                            _stream()), "conv2d_gnstats")
        _emit_gn_pre(gstat, rows.value, Ho * Wo, gn_unit, B * Ho * Wo, cout)
        return out
    check(lib.mi355_conv2d_bf16(x.data_ptr(), B, H, W, C, w.data_ptr(), cout, ksize, stride, int(ups), int(dil), Ho, Wo,
                                out.data_ptr(), int(out_fp32), _ptr(bias), _ptr(rowbias),
                                0 if rowbias is None else rowbias.stride(0), _ptr(residual), 1.0, ws.data_ptr(),
                                ws.numel(), _stream()), "conv2d")
    return out


def wgrad(dy, x, cin, cout, ksize=1, stride=1, ups=False, geom=None, out=None, accumulate=False, dbias=None):
    """fp32 [Cout][k*k][Cin].  Linear: dy [M,Cout], x [M,Cin].  Conv: geom=(B,H,W,Ho,Wo).
    dbias (optional fp32 [Cout]) receives the bias gradient from the same pass."""
    if geom is None:
        B, H, W, Ho, Wo = dy.shape[0], 1, 1, 1, 1
    else:
        B, H, W, Ho, Wo = geom
    if out is None:
        out = torch.empty((cout, ksize * ksize, cin), dtype=F32, device=dy.device)
    check(lib.mi355_wgrad_bf16(dy.data_ptr(), x.data_ptr(), out.data_ptr(), _ptr(dbias), B, H, W, cin, cout, ksize, stride,
                               int(ups), Ho, Wo, int(accumulate), _stream()), "wgrad")
    return out


class PinnedRing:
    """Round-robin pinned host buffers behind asynchronous host -> device copies.

    A copy out of pinned memory reads the buffer when the STREAM reaches it, not when the host enqueues it, so a slot may
    be refilled only after the copy that last read it has run.  Each slot keeps an event recorded right behind its copy;
    `acquire()` waits for that event on the host, which costs nothing unless the host really is a whole ring ahead of the
    device (an eager step with no synchronisation lets it run ahead without bound).  A copy enqueued during a stream
    capture is re-read on every graph replay: its slot is retired into `keep` for the life of the owner and the ring is
    refilled by the next eager `release()` (host memory cannot be pinned while a capture is open)."""

    def __init__(self, make, size):
        self._make, self._size = make, size
        self._bufs = [make() for _ in range(size)]
        self._evs = [None] * size
        self._i = 0
        self.keep = []

    def acquire(self):
        """The buffer to fill next, safe to overwrite."""
        ev = self._evs[self._i]
        if ev is not None:
            ev.synchronize()
            self._evs[self._i] = None
        return self._bufs[self._i]

    def copied(self):
        """Call on the stream, and right after, the copy out of the acquired buffer was enqueued."""
        if not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record()
            self._evs[self._i] = ev

    def release(self):
        """Move on to the next slot (after `copied()`)."""
        if torch.cuda.is_current_stream_capturing():
            self.keep.append(self._bufs.pop(self._i))
            self._evs.pop(self._i)
            if not self._bufs:
                raise RuntimeError("PinnedRing: ring exhausted during graph capture")
            self._i %= len(self._bufs)
        else:
            while len(self._bufs) < self._size:
                self._bufs.append(self._make())
                self._evs.append(None)
            self._i = (self._i + 1) % len(self._bufs)


class WgradQueue:
    """Deferred weight gradients: backward pushes (dy, x, destination) jobs, `flush()` runs them as ONE
    grouped launch.  Thousands of tiles in flight keep all 256 CUs busy without aggressive pixel splits
    (less atomic traffic) and remove ~230 small launches per step.  Destinations must be pre-zeroed.

    With `side_stream=True` the grouped launches run on a second stream, concurrently with the rest of the
    backward pass (whose small normalisation / projection kernels leave most CUs idle); `join()` makes the
    current stream wait for them.  Operands stay referenced until `join()`, so the allocator cannot hand their
    memory to a later kernel of the main stream while a grouped launch still reads it."""

    def __init__(self, device, flush_blocks=int(os.environ.get("MI355_WGRAD_FLUSH", "4096")), max_jobs=512, ring=64,
                 side_stream=False):
        self.device = device
        self.side = torch.cuda.Stream(device) if side_stream else None
        self._inflight = []
        self.jb = int(lib.mi355_wgrad_job_bytes())
        self.max_jobs, self.flush_blocks = max_jobs, flush_blocks
        self._ring = PinnedRing(lambda: torch.empty(max_jobs * (self.jb + 16), dtype=torch.uint8).pin_memory(), ring)
        self._host = self._ring.acquire()
        self._n = 0
        self._blocks = 0
        self._flops = 0.0
        self._refs = []
        self._cbs = []
        self._splits = ctypes.c_int(0)
        # longest-running jobs first in every launch (MI355_WGRAD_SORT=0: in the order they were queued)
        self.sort_jobs = os.environ.get("MI355_WGRAD_SORT", "1") != "0"
        # destinations of pixel-split jobs accumulate with atomics: the queue zeroes exactly those (one launch per
        # flush) when `zero_split_outputs` is set, so the owner need not clear the whole gradient buffer every step
        self.zero_split_outputs = False
        self._zero = []
        # `track_unsplit(lo, hi, ...)`: jobs that are not pixel-split and whose destination lies in the address range
        # [lo, hi) of an fp32 gradient buffer are listed in `covered` (address, floats) and
        #   slots   - add the sum of squares of what they store to these `mi355_wgrad_sumsq_slots()` floats (the owner of the gradient norm then
        #             skips their part of the buffer), and / or
        #   stage16 - store bf16 into the buffer of that address laid out like [lo, hi) INSTEAD of fp32 (the staging
        #             buffer of a bf16 gradient all-reduce: no separate pack pass over these gradients)
        self._tr_lo = self._tr_hi = 0
        self._tr_slots = self._tr_stage = None
        self.covered = []
        self.covered_map = {}       # address -> floats, the same records

    def track_unsplit(self, lo=0, hi=0, slots=None, stage16=None):
        self._tr_lo, self._tr_hi = lo, hi
        self._tr_slots = None if slots is None else slots.data_ptr()
        self._tr_stage = None if stage16 is None else stage16.data_ptr()
        self.covered = []
        self.covered_map = {}

    def push(self, dy, x, dw, dbias, B, H, W, cin, cout, ksize, stride, ups, Ho, Wo, callback=None):
        dwp = dw.data_ptr()
        tracked = self._tr_lo <= dwp < self._tr_hi
        slots = self._tr_slots if tracked else None
        dw16 = self._tr_stage + (dwp - self._tr_lo) // 2 if (tracked and self._tr_stage is not None) else None
        nb = lib.mi355_wgrad_job_fill(self._host.data_ptr() + self._n * self.jb, dy.data_ptr(), x.data_ptr(),
                                      dwp, _ptr(dbias), B, H, W, cin, cout, ksize, stride, int(ups), Ho, Wo,
                                      self._blocks, ctypes.byref(self._splits), slots, dw16)
        check(min(nb, 0), "wgrad_job_fill")
        if tracked and self._splits.value == 1:
            self.covered.append((dwp, cout * ksize * ksize * cin))
            self.covered_map[dwp] = cout * ksize * ksize * cin
        if self.zero_split_outputs and self._splits.value > 1:
            self._zero.append((dw.data_ptr(), dw.numel() * 4))
        self._n += 1
        self._blocks += nb
        self._flops += 2.0 * B * Ho * Wo * cout * ksize * ksize * cin
        self._refs.append((dy, x))
        if callback is not None:
            self._cbs.append(callback)
        if self._blocks >= self.flush_blocks or self._n == self.max_jobs:
            self.flush()

    def push_cat(self, dy, x1, x2, dw, dbias, callback=None):
        """The dense job whose input is the channel concatenation [x1 | x2] (2-d views, rows = pixels), read in place."""
        M, cout = dy.shape
        c1, c2 = x1.shape[1], x2.shape[1]
        dwp = dw.data_ptr()
        tracked = self._tr_lo <= dwp < self._tr_hi
        slots = self._tr_slots if tracked else None
        dw16 = self._tr_stage + (dwp - self._tr_lo) // 2 if (tracked and self._tr_stage is not None) else None
        nb = lib.mi355_wgrad_job_fill_cat(self._host.data_ptr() + self._n * self.jb, dy.data_ptr(), x1.data_ptr(), c1,
                                          x2.data_ptr(), c2, dwp, _ptr(dbias), M, cout, self._blocks,
                                          ctypes.byref(self._splits), slots, dw16)
        check(min(nb, 0), "wgrad_job_fill_cat")
        if tracked and self._splits.value == 1:
            self.covered.append((dwp, cout * (c1 + c2)))
            self.covered_map[dwp] = cout * (c1 + c2)
        if self.zero_split_outputs and self._splits.value > 1:
            self._zero.append((dw.data_ptr(), dw.numel() * 4))
        self._n += 1
        self._blocks += nb
        self._flops += 2.0 * M * cout * (c1 + c2)
        self._refs.append((dy, x1, x2))
        if callback is not None:
            self._cbs.append(callback)
        if self._blocks >= self.flush_blocks or self._n == self.max_jobs:
            self.flush()

    def flush(self):
        if self._n == 0:
            return
        nbytes = self._n * self.jb

        def launch():
            nz = len(self._zero)
            if self.sort_jobs and self._n > 1:
                check(min(lib.mi355_wgrad_jobs_sort(self._host.data_ptr(), self._n), 0), "wgrad_jobs_sort")
            table = torch.empty(nbytes + 16 * nz, dtype=torch.uint8, device=self.device)
            if nz:
                zt = self._host[nbytes:nbytes + 16 * nz].view(torch.int64)
                for i, (ptr, nb) in enumerate(self._zero):
                    zt[2 * i], zt[2 * i + 1] = ptr, nb
            table.copy_(self._host[:nbytes + 16 * nz], non_blocking=True)
            self._ring.copied()
            if nz:
                check(lib.mi355_zero_ranges(table.data_ptr() + nbytes, nz, _stream()), "zero_ranges")
            check(lib.mi355_wgrad_grouped(table.data_ptr(), self._n, self._blocks, self._flops, _stream()), "wgrad_grouped")
            return table

        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())      # every queued dy / x has been produced
            with torch.cuda.stream(self.side):
                self._inflight.append((launch(), self._refs))
        else:
            launch()
        for cb in self._cbs:
            cb()
        self._ring.release()
        self._host = self._ring.acquire()
        self._n, self._blocks, self._flops = 0, 0, 0.0
        self._refs, self._cbs, self._zero = [], [], []

    def discard(self):
        """Drop the queued jobs without running them (a step that raised half-way through backward)."""
        self._n, self._blocks, self._flops = 0, 0, 0.0
        self._refs, self._cbs, self._zero = [], [], []

    def join(self):
        """Flush, then make the current stream wait for every grouped launch issued so far."""
        self.flush()
        if self.side is not None and self._inflight:
            torch.cuda.current_stream().wait_stream(self.side)
            self._inflight = []


_WGRAD_QUEUE = None


def set_wgrad_queue(q):
    """Install (or remove with None) the queue used by layers in direct-gradient mode."""
    global _WGRAD_QUEUE
    _WGRAD_QUEUE = q


class ReduceQueue:
    """Deferred parameter-gradient reductions of the norm layers: in direct-gradient mode their backward kernels
    leave per-workgroup partial rows in scratch and push one record here; `flush()` reduces all of them (109 layers
    of the denoiser) with ONE launch into the flat gradient buffer instead of one small launch per layer."""

    def __init__(self, device, max_entries=256, ring=8):
        self.device = device
        self.max_entries = max_entries
        self._ring = PinnedRing(lambda: torch.empty(max_entries * 6, dtype=torch.int64).pin_memory(), ring)
        self._entries, self._refs, self._max_cols = [], [], 0

    def push(self, partial_ptr, out, out_hi, rows, n, split, scratch):
        self._entries.append((partial_ptr, out.data_ptr(), 0 if out_hi is None else out_hi.data_ptr(), rows, n, split))
        self._refs.append(scratch)          # the partial rows must outlive the flush
        self._max_cols = max(self._max_cols, n)
        if len(self._entries) == self.max_entries:
            self.flush()

    def discard(self):
        self._entries, self._refs, self._max_cols = [], [], 0

    def flush(self):
        n = len(self._entries)
        if n == 0:
            return
        host = self._ring.acquire()
        host[:6 * n] = torch.tensor(self._entries, dtype=torch.int64).view(-1)
        table = torch.empty(6 * n, dtype=torch.int64, device=self.device)
        table.copy_(host[:6 * n], non_blocking=True)
        self._ring.copied()
        check(lib.mi355_reduce_rows_batched(table.data_ptr(), n, self._max_cols, _stream()), "reduce_rows_batched")
        self._ring.release()
        self._keep_last = (table, self._refs)       # alive until the next flush (the launch is asynchronous)
        self._entries, self._refs, self._max_cols = [], [], 0


_REDUCE_QUEUE = None
# True while a DenoiserTrainer step runs its forward and backward pass: layers with per-step operands that the trainer
# prepared (the LoRA adapters' bf16 factor copies) use them only then
_DIRECT_STEP = False


def set_direct_step(on):
    global _DIRECT_STEP
    _DIRECT_STEP = bool(on)


def set_reduce_queue(q):
    global _REDUCE_QUEUE
    _REDUCE_QUEUE = q


def colsum(y2d, seg=None):
    M, N = y2d.shape
    seg = M if seg is None else seg
    out = torch.empty((M // seg, N), dtype=F32, device=y2d.device)
    scratch = torch.empty(int(lib.mi355_colsum_scratch_floats(M, N, seg)), dtype=F32, device=y2d.device)
    check(lib.mi355_colsum_bf16(y2d.data_ptr(), out.data_ptr(), scratch.data_ptr(), M, N, seg, _stream()), "colsum")
    return out


ATTN_PRESCALE = 1.4426950408889634       # log2(e): a prescaled query carries head_dim ** -0.5 * ATTN_PRESCALE


def attention_fwd(q, k, v, heads, kv_len=None, prescaled=False):
    """q [B,Nq,C], k/v [B,Nk,C] (last dim contiguous; row stride arbitrary multiple of 8).
    kv_len: optional int32 [B] device tensor, keys >= kv_len[b] are masked (key-padding mask).
    prescaled: q already carries head_dim ** -0.5 * log2(e) (mi355_attention_fwd_prescaled; no gradient path)."""
    B, Nq, C = q.shape
    Nk = k.shape[1]
    D = C // heads
    out = torch.empty((B, Nq, C), dtype=BF16, device=q.device)
    lse = torch.empty((B, heads, Nq), dtype=F32, device=q.device)
    if kv_len is not None:
        assert kv_len.dtype == torch.int32 and kv_len.is_cuda and kv_len.numel() == B
        check(lib.mi355_attention_fwd_masked(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                             kv_len.data_ptr(), B, heads, Nq, Nk, D, q.stride(1), k.stride(1), v.stride(1),
                                             C, 1.0 / math.sqrt(D), _stream()), "attention_fwd_masked")
        return out, lse
    if prescaled:
        check(lib.mi355_attention_fwd_prescaled(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), B,
                                                heads, Nq, Nk, D, q.stride(1), k.stride(1), v.stride(1), C, _stream()),
              "attention_fwd_prescaled")
        return out, lse
    check(lib.mi355_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), B, heads,
                                  Nq, Nk, D, q.stride(1), k.stride(1), v.stride(1), C, 1.0 / math.sqrt(D), _stream()),
          "attention_fwd")
    return out, lse


def attention_bwd(q, k, v, o, do, lse, heads, dq, dk, dv):
    B, Nq, C = q.shape
    Nk = k.shape[1]
    D = C // heads
    delta = torch.empty((B, heads, Nq), dtype=F32, device=q.device)
    assert dq.stride(1) == q.stride(1)
    check(lib.mi355_attention_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                                  delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, heads, Nq, Nk, D,
                                  q.stride(1), k.stride(1), v.stride(1), C, dk.stride(1), dv.stride(1), 1.0 / math.sqrt(D),
                                  _stream()), "attention_bwd")


def gelu_bf16(x):
    y = torch.empty_like(x)
    check(lib.mi355_gelu_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "gelu")
    return y


def layernorm_bf16(x2d, gamma, beta, eps):
    """LayerNorm over the last dim of a contiguous bf16 [M, C] matrix (forward only, no statistics kept)."""
    M, C = x2d.shape
    y = torch.empty_like(x2d)
    check(lib.mi355_layernorm_fwd(x2d.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), None, M, C, eps,
                                  _stream()), "layernorm_fwd")
    return y


def cast_bf16(x_f32, out=None):
    n = x_f32.numel()
    if out is None:
        out = torch.empty(x_f32.shape, dtype=BF16, device=x_f32.device)
    check(lib.mi355_cast_f32_to_bf16(x_f32.data_ptr(), out.data_ptr(), n, _stream()), "cast")
    return out


# ------------------------------------------------------------------------------------------------
# autograd functions
# ------------------------------------------------------------------------------------------------
class WeightPack:
    """bf16 operand copies of one (possibly fused) weight matrix, refreshed by the owning module.

    w16   : [N][K] bf16 (forward operand)
    w16_t : data-gradient operand: [K][N] for linear layers, [Cin][k][k][Cout] tap-flipped for convs.
    bt    : the data gradient reads w16 as stored (gemm_bt / conv2d_dgrad_bt); w16_t is then None and never rebuilt.
    """
    __slots__ = ("w16", "w16_t", "n", "k", "ksize", "cin", "cout", "cin_pad", "cout_pad", "bias_pad", "grad_out",
                 "bias_grad_out", "on_grad_ready", "bias_parts", "bt", "gn_unit")

    def __init__(self):
        self.gn_unit = 0          # > 0: the layer's output feeds a GroupNorm - emit its statistics in runs of this many channels
        self.bias_parts = 0       # number of trailing bias parameters in LinearFn's `weights` (fused projections)
        self.bt = False
        self.w16 = self.w16_t = None
        self.bias_pad = None      # fp32 bias padded to cout_pad (convolutions whose Cout is not a multiple of 8)
        self.grad_out = None      # optional fp32 [N][K] view the weight gradient is written into directly
        self.bias_grad_out = None  # optional fp32 [N] view for the bias gradient (both views are pre-zeroed by the owner)
        self.on_grad_ready = None


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) (+ residual); weights may be several row-stacked parameters (fused projections).
    With `pack.bias_parts = n` the last n entries of `weights` are the bias parameters whose concatenation
    `bias` is (a view of the flat storage), and each receives its slice of the bias gradient."""

    @staticmethod
    def forward(ctx, x, pack, bias, residual, out_fp32, *weights):
        x2 = x.reshape(-1, x.shape[-1])
        res2 = None if residual is None else residual.reshape(-1, residual.shape[-1])
        hw = x.shape[1] * x.shape[2] if x.dim() == 4 else 0
        if pack.gn_unit and _GN_PRE_ON and not out_fp32 and hw % 128 == 0 and hw > 0 and pack.n % pack.gn_unit == 0 and (res2 is None or res2.stride(1) == 1):
            # a 1x1 convolution on an NHWC map whose output feeds a GroupNorm: statistics out of the epilogue
            y = gemm_gnstats(x2, pack.w16.view(pack.n, pack.k), bias, res2, hw, pack.gn_unit)
        else:
            y = gemm(x2, pack.w16.view(pack.n, pack.k), bias=bias, residual=res2, out_fp32=out_fp32)
        ctx.save_for_backward(x2)
        ctx.pack = pack
        ctx.xshape = x.shape
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        nb = pack.bias_parts
        ctx.bshapes = [b.shape for b in weights[len(weights) - nb:]] if nb else []
        ctx.breq = [b.requires_grad for b in weights[len(weights) - nb:]] if nb else []
        weights = weights[:len(weights) - nb] if nb else weights
        ctx.wshapes = [w.shape for w in weights]
        ctx.wreq = [w.requires_grad for w in weights]
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        return LinearFn._backward(ctx, x2, dy, ctx.needs_input_grad)

    @staticmethod
    def _backward(ctx, x2, dy, needs):
        """Data, weight and bias gradients of y = x W^T (+ b) (+ residual); `needs` has LinearFn.forward's layout."""
        pack = ctx.pack
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != BF16:
            dy2 = dy2.to(BF16)
        dy2 = dy2.contiguous()
        dx = db = dres = None
        if needs[0]:
            if pack.bt:
                dx = gemm_bt(dy2, pack.w16.view(pack.n, pack.k)).view(ctx.xshape)
            else:
                dx = gemm(dy2, pack.w16_t.view(pack.k, pack.n)).view(ctx.xshape)
        want_db = ctx.has_bias and (needs[2] or any(ctx.breq))
        if ctx.has_res and needs[3]:
            dres = dy2.view(dy.shape)
        dws = [None] * len(ctx.wshapes)
        if any(ctx.wreq):
            if pack.grad_out is not None:
                # direct mode: the owner zeroed the flat gradient buffer; gradients are written in place
                if want_db and pack.bias_grad_out is None:
                    db = torch.zeros(pack.n, dtype=F32, device=dy2.device)
                dbias = (pack.bias_grad_out if pack.bias_grad_out is not None else db) if want_db else None
                if _WGRAD_QUEUE is not None:
                    _WGRAD_QUEUE.push(dy2, x2, pack.grad_out, dbias, dy2.shape[0], 1, 1, pack.k, pack.n, 1, 1, False, 1, 1,
                                      pack.on_grad_ready)
                else:
                    wgrad(dy2, x2, pack.k, pack.n, out=pack.grad_out, accumulate=2, dbias=dbias)
                    if pack.on_grad_ready is not None:
                        pack.on_grad_ready()
            else:
                if want_db:
                    db = torch.empty(pack.n, dtype=F32, device=dy2.device)
                dw = wgrad(dy2, x2, pack.k, pack.n, dbias=db).view(pack.n, pack.k)
                r0 = 0
                for i, shp in enumerate(ctx.wshapes):
                    if ctx.wreq[i]:
                        dws[i] = dw[r0:r0 + shp[0]].view(shp)
                    r0 += shp[0]
        elif want_db:
            db = colsum(dy2)[0]
        dbs = [None] * len(ctx.bshapes)
        if ctx.bshapes and db is not None:
            r0 = 0
            for i, shp in enumerate(ctx.bshapes):
                if ctx.breq[i]:
                    dbs[i] = db[r0:r0 + shp[0]]
                r0 += shp[0]
            db = None
        return (dx, None, db, dres, None, *dws, *dbs)


class GegluLinearFn(torch.autograd.Function):
    """GEGLU input projection in one GEMM: y = h * gelu(g) with (h | g) = x W^T + b.  The epilogue also stores the
    bf16 pre-activation for backward, so the [M][2F] matrix is written once and never re-read in the forward pass."""

    @staticmethod
    def forward(ctx, x, pack, bias, weight):
        x2 = x.reshape(-1, x.shape[-1])
        y, pre = gemm_geglu(x2, pack.w16.view(pack.n, pack.k), bias, keep_preact=True)
        ctx.save_for_backward(x2, pre)
        ctx.pack, ctx.xshape = pack, x.shape
        ctx.has_bias, ctx.has_res = bias is not None, False
        ctx.bshapes, ctx.breq = [], []
        ctx.wshapes, ctx.wreq = [weight.shape], [weight.requires_grad]
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, pre = ctx.saved_tensors
        dy = dy.contiguous()
        F2 = pre.shape[1]
        dpre = torch.empty_like(pre)
        check(lib.mi355_geglu_bwd(pre.data_ptr(), dy.data_ptr(), dpre.data_ptr(), pre.shape[0], F2 // 2, _stream()), "geglu_bwd")
        n = ctx.needs_input_grad
        dx, _, db, _, _, dw = LinearFn._backward(ctx, x2, dpre, (n[0], False, n[2], False, False, n[3]))
        return dx, None, db, dw


class FeedForwardFn(torch.autograd.Function):
    """The transformer feed-forward block as one node: out = (h * gelu(g)) W2^T + b2 + residual with (h | g) = x W1^T + b1.
    Forward: the gated GEMM (pre-activation kept) and the output projection with the residual in its epilogue, as the two
    separate nodes would.  Backward: the output projection's data gradient runs with the gate's backward in its epilogue
    (`mi355_gemm_geglu_bwd_bf16`), so the [M][4C] gradient of the gated activations is never written or re-read."""

    @staticmethod
    def forward(ctx, x, residual, pk1, b1, w1, pk2, b2, w2):
        from types import SimpleNamespace
        x2 = x.reshape(-1, x.shape[-1])
        y, pre = gemm_geglu(x2, pk1.w16.view(pk1.n, pk1.k), b1, keep_preact=True)
        res2 = None if residual is None else residual.reshape(-1, residual.shape[-1])
        out = gemm(y, pk2.w16.view(pk2.n, pk2.k), bias=b2, residual=res2)
        ctx.save_for_backward(x2, pre, y)
        ctx.c1 = SimpleNamespace(pack=pk1, xshape=x.shape, has_bias=b1 is not None, has_res=False, bshapes=[], breq=[],
                                 wshapes=[w1.shape], wreq=[w1.requires_grad])
        ctx.c2 = SimpleNamespace(pack=pk2, xshape=y.shape, has_bias=b2 is not None, has_res=residual is not None, bshapes=[],
                                 breq=[], wshapes=[w2.shape], wreq=[w2.requires_grad])
        return out.view(*x.shape[:-1], pk2.n)

    @staticmethod
    def backward(ctx, dout):
        x2, pre, y = ctx.saved_tensors
        n = ctx.needs_input_grad            # x, residual, pk1, b1, w1, pk2, b2, w2
        pk2 = ctx.c2.pack
        d2 = (dout if dout.dtype == BF16 else dout.to(BF16)).reshape(-1, pk2.n).contiguous()
        if pk2.bt:
            dpre = gemm_bt(d2, pk2.w16.view(pk2.n, pk2.k), pre=pre)
        else:
            dpre = gemm_geglu_bwd(d2, pk2.w16_t.view(pk2.k, pk2.n), pre)
        _, _, db2, dres, _, dw2 = LinearFn._backward(ctx.c2, y, d2.view(dout.shape), (False, False, n[6], n[1], False, n[7]))
        dx, _, db1, _, _, dw1 = LinearFn._backward(ctx.c1, x2, dpre, (n[0], False, n[3], False, False, n[4]))
        return dx, dres, None, db1, dw1, None, db2, dw2


class SplitColsFn(torch.autograd.Function):
    """Column slices of a [B, N] matrix as views (row stride N); backward concatenates the slice gradients."""

    @staticmethod
    def forward(ctx, y, sizes):
        ctx.sizes = sizes
        ctx.meta = (y.shape[0], y.dtype, y.device)
        outs, c0 = [], 0
        for n in sizes:
            outs.append(y[:, c0:c0 + n])
            c0 += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        B, dtype, dev = ctx.meta
        parts = [g if g is not None else torch.zeros((B, n), dtype=dtype, device=dev) for g, n in zip(grads, ctx.sizes)]
        return torch.cat(parts, dim=1), None


class ContextKVFn(torch.autograd.Function):
    """Text keys and values of SEVERAL cross-attention layers from one product: y = x W_all^T with W_all the row-stacked
    (to_k | to_v) weights of every layer (a packed bf16 operand the owner keeps current); each layer receives its
    [B, T, 2 C_i] column slice of y as a view.  Backward: x (the text states) needs no gradient; each layer's weight
    gradient is its own job, exactly as if the layer had projected alone (its slice gradient arrives contiguous: the
    attention backward kernels write dk / dv with strides of their own).
    `weights`: (to_k.weight, to_v.weight) of every layer, in the order of `packs` (the layers' fused operand packs)."""

    @staticmethod
    def forward(ctx, x, w_all, packs, *weights):
        from types import SimpleNamespace
        B, T, K = x.shape
        x2 = x.reshape(B * T, K)
        y = gemm(x2, w_all)
        ctx.save_for_backward(x2)
        ctx.layers = [SimpleNamespace(pack=pk, xshape=x.shape, has_bias=False, has_res=False, bshapes=[], breq=[],
                                      wshapes=[weights[2 * i].shape, weights[2 * i + 1].shape],
                                      wreq=[weights[2 * i].requires_grad, weights[2 * i + 1].requires_grad])
                      for i, pk in enumerate(packs)]
        outs, c0 = [], 0
        for pk in packs:
            outs.append(y[:, c0:c0 + pk.n].view(B, T, pk.n))
            c0 += pk.n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        (x2,) = ctx.saved_tensors
        dws = []
        for ns, g in zip(ctx.layers, grads):
            if g is None or not any(ns.wreq):
                dws += [None, None]
                continue
            r = LinearFn._backward(ns, x2, g, (False, False, False, False, False))
            dws += list(r[5:7])
        return (None, None, None, *dws)


class ConvFn(torch.autograd.Function):
    """NHWC convolution (3x3 / 1x1, stride 1/2, optional fused nearest-2x upsample of the input)
    with bias, per-sample broadcast bias (timestep embedding) and residual fused in the epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, rowbias, residual, pack, stride, ups, out_fp32=False):
        kb = bias if (bias is None or pack.bias_pad is None) else pack.bias_pad
        y = conv2d(x, pack.w16, pack.cout_pad, pack.ksize, stride=stride, ups=ups, bias=kb, rowbias=rowbias,
                   residual=residual, out_fp32=out_fp32, gn_unit=pack.gn_unit)
        ctx.save_for_backward(x)
        ctx.pack, ctx.stride, ctx.ups = pack, stride, ups
        ctx.flags = (bias is not None, rowbias is not None, residual is not None)
        ctx.wreq = weight.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        pack, stride, ups = ctx.pack, ctx.stride, ctx.ups
        pre_sums = _DX_COLSUMS.pop(dy.data_ptr(), None)      # left by the GroupNorm backward that produced dy, if any
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        dy = dy.contiguous()
        B, H, W, C = x.shape
        _, Ho, Wo, Co = dy.shape
        dx = dw = db = drb = dres = None
        if ctx.needs_input_grad[0]:
            hv, wv = (2 * H, 2 * W) if ups else (H, W)
            if pack.bt:
                dxv = conv2d_dgrad_bt(dy, pack.w16, pack.cin_pad, pack.ksize, dil=(stride == 2), out_hw=(hv, wv))
            else:
                dxv = conv2d(dy, pack.w16_t, pack.cin_pad, pack.ksize, stride=1, dil=(stride == 2), out_hw=(hv, wv))
            if ups:
                dx = torch.empty_like(x)
                check(lib.mi355_sumpool2x2_bf16(dxv.data_ptr(), dx.data_ptr(), B, H, W, C, _stream()), "sumpool2x2")
            else:
                dx = dxv
        want_db = ctx.flags[0] and ctx.needs_input_grad[2]
        dbp = None
        direct_b = False
        if ctx.wreq:
            geom = (B, H, W, Ho, Wo)
            if pack.grad_out is not None:
                direct_b = want_db and pack.bias_grad_out is not None
                if want_db and not direct_b:
                    dbp = torch.zeros(pack.cout_pad, dtype=F32, device=dy.device)
                dbias = (pack.bias_grad_out if direct_b else dbp) if want_db else None
                if _WGRAD_QUEUE is not None:
                    _WGRAD_QUEUE.push(dy, x, pack.grad_out, dbias, B, H, W, pack.cin_pad, pack.cout_pad, pack.ksize, stride,
                                      ups, Ho, Wo, pack.on_grad_ready)
                else:
                    wgrad(dy, x, pack.cin_pad, pack.cout_pad, pack.ksize, stride, ups, geom=geom, out=pack.grad_out,
                          accumulate=2, dbias=dbias)
                    if pack.on_grad_ready is not None:
                        pack.on_grad_ready()
            else:
                if want_db:
                    dbp = torch.empty(pack.cout_pad, dtype=F32, device=dy.device)
                dwp = wgrad(dy, x, pack.cin_pad, pack.cout_pad, pack.ksize, stride, ups, geom=geom, dbias=dbp)
                dwp = dwp.view(pack.cout_pad, pack.ksize, pack.ksize, pack.cin_pad)
                # logical OIHW view of the physical [O][kh][kw][I] buffer (channels_last), sliced to the true sizes
                dw = dwp.permute(0, 3, 1, 2)[:pack.cout, :pack.cin]
        dy2 = dy.view(-1, Co)
        if want_db and not direct_b:
            db = (dbp if dbp is not None else colsum(dy2)[0])[:pack.cout]
        if ctx.flags[1] and ctx.needs_input_grad[3]:
            if pre_sums is not None and pre_sums.shape == (B, Co):
                drb = pre_sums
            else:
                drb = colsum(dy2, seg=Ho * Wo)
        if ctx.flags[2] and ctx.needs_input_grad[4]:
            dres = dy
        return dx, dw, db, drb, dres, None, None, None, None


_GN_TWO_PASS = {}


def _gn_pre_usable(hw, C, groups, pre, C1=None, pre2=None):
    """Whether producer-emitted statistics can stand in for the statistics pass of this GroupNorm: the two-pass geometry,
    every group whole runs of the emitted unit, and - for a two-source input - statistics of both sources in one unit."""
    if pre is None or not _GN_PRE_ON:
        return False
    key = (hw, C, groups)
    if key not in _GN_TWO_PASS:
        _GN_TWO_PASS[key] = bool(lib.mi355_groupnorm_fwd_two_pass(hw, C, groups))
    unit = pre[2]
    if not _GN_TWO_PASS[key] or (C // groups) % unit:
        return False
    if C1 is not None:
        return pre2 is not None and pre2[2] == unit and C1 % unit == 0
    return True


def groupnorm_forward(x, x2, gamma, beta, y, stats, groups, eps, silu, pre=None, pre2=None):
    """GroupNorm(+SiLU) forward of x (or of the channel concatenation [x | x2]) into y / stats: from the producers'
    statistics when they came with the input(s), else with the statistics pass."""
    B, H, W, C1 = x.shape
    C = C1 + (x2.shape[-1] if x2 is not None else 0)
    if _gn_pre_usable(H * W, C, groups, pre, C1 if x2 is not None else None, pre2):
        check(lib.mi355_groupnorm_fwd_pre(x.data_ptr(), _ptr(x2), C1, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                          stats.data_ptr(), pre[0].data_ptr(), pre[1], None if x2 is None else pre2[0].data_ptr(),
                                          0 if x2 is None else pre2[1], pre[2], B, H * W, C, groups, eps, int(silu), _stream()),
              "groupnorm_fwd_pre")
        return
    scratch = torch.empty(int(lib.mi355_groupnorm_scratch_floats(B, H * W, C, groups)), dtype=F32, device=x.device)
    if x2 is None:
        check(lib.mi355_groupnorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                      scratch.data_ptr(), B, H * W, C, groups, eps, int(silu), _stream()), "groupnorm_fwd")
    else:
        check(lib.mi355_groupnorm_fwd_cat(x.data_ptr(), C1, x2.data_ptr(), C - C1, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                          stats.data_ptr(), scratch.data_ptr(), B, H * W, groups, eps, int(silu), _stream()),
              "groupnorm_fwd_cat")


def groupnorm_cat(x1, x2, gamma, beta, groups, eps, silu):
    """GroupNorm(+SiLU) of torch.cat([x1, x2], -1) for NHWC bf16 tensors, without building the concatenation (inference:
    no autograd node)."""
    B, H, W, C1 = x1.shape
    C2 = x2.shape[-1]
    assert x2.shape[:3] == x1.shape[:3] and x1.dtype == BF16 and x2.dtype == BF16 and x1.is_contiguous() and x2.is_contiguous()
    y = torch.empty((B, H, W, C1 + C2), dtype=BF16, device=x1.device)
    stats = torch.empty((B, groups, 2), dtype=F32, device=x1.device)
    groupnorm_forward(x1, x2, gamma, beta, y, stats, groups, eps, silu, getattr(x1, "_gn_pre", None), getattr(x2, "_gn_pre", None))
    return y


class CatNormShortcutFn(torch.autograd.Function):
    """Head of an up-path residual block on the channel concatenation [x1 | x2] (hidden state | skip connection) without
    building it, for training: returns (GroupNorm+SiLU of the concatenation, its 1x1 shortcut convolution).  Backward
    writes the two input gradients as separate tensors, the shortcut's data gradient added inside the normalisation
    kernel; the shortcut's weight gradient is one deferred job that reads both tensors in place.  A trained shortcut
    weight needs its preassigned gradient buffer (`cat_train_ready`)."""

    @staticmethod
    def forward(ctx, x1, x2, gamma, beta, groups, eps, silu, gdst, pack, sc_bias, sc_weight, pre1=None, pre2=None):
        B, H, W, C1 = x1.shape
        C2 = x2.shape[-1]
        C = C1 + C2
        y = torch.empty((B, H, W, C), dtype=BF16, device=x1.device)
        stats = torch.empty((B, groups, 2), dtype=F32, device=x1.device)
        groupnorm_forward(x1, x2, gamma, beta, y, stats, groups, eps, silu, pre1, pre2)
        w = pack.w16.view(pack.n, C)
        if C1 % 64 == 0 and C2 % 64 == 0 and pack.n % 160 == 0:
            res = gemm_cat(x1.view(-1, C1), x2.view(-1, C2), w, bias=sc_bias)
        else:
            res = gemm(x1.view(-1, C1), w[:, :C1], bias=sc_bias)
            res = gemm(x2.view(-1, C2), w[:, C1:], residual=res, out=res)
        ctx.save_for_backward(x1, x2, stats, gamma, beta)
        ctx.groups, ctx.silu, ctx.gdst, ctx.pack = groups, silu, gdst, pack
        ctx.has_bias = sc_bias is not None
        return y, res.view(B, H, W, pack.n)

    @staticmethod
    def backward(ctx, dy, dres):
        x1, x2, stats, gamma, beta = ctx.saved_tensors
        pack, G = ctx.pack, ctx.groups
        B, H, W, C1 = x1.shape
        C2 = x2.shape[-1]
        C = C1 + C2
        dy = dy.contiguous()
        dr2 = (dres if dres.dtype == BF16 else dres.to(BF16)).contiguous().view(-1, pack.n)
        needs = ctx.needs_input_grad
        # shortcut data gradients, one per source; the normalisation kernel adds them to its own
        if pack.bt:
            d1, d2 = gemm_bt(dr2, pack.w16.view(pack.n, C), n1=C1)
        else:
            wt = pack.w16_t.view(C, pack.n)
            if C1 % 320 == 0 and C % 160 == 0:      # (no tile of the product may straddle C1: see mi355_gemm_split_bf16)
                d1, d2 = gemm_split(dr2, wt, C1)
            else:
                d1, d2 = gemm(dr2, wt[:C1]), gemm(dr2, wt[C1:])
        dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
        direct = ctx.gdst is not None
        frozen = not (needs[2] or needs[3])
        dg, db = ctx.gdst if direct else ((None, None) if frozen else (torch.empty_like(gamma), torch.empty_like(beta)))
        nsc = int(lib.mi355_groupnorm_scratch_floats(B, H * W, C, G))
        scratch = torch.empty(nsc, dtype=F32, device=x1.device)
        deferred = direct and not frozen and _REDUCE_QUEUE is not None
        check(lib.mi355_groupnorm_bwd_cat(x1.data_ptr(), C1, x2.data_ptr(), C2, dy.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                                          beta.data_ptr(), dx1.data_ptr(), dx2.data_ptr(), None if (deferred or frozen) else _ptr(dg),
                                          None if (deferred or frozen) else _ptr(db), scratch.data_ptr(), B, H * W, G, int(ctx.silu),
                                          d1.data_ptr(), d2.data_ptr(), _stream()), "groupnorm_bwd_cat")
        if deferred:        # per-strip channel partials follow the group partials in scratch
            strips = nsc // (B * (2 * G + 2 * C))
            _REDUCE_QUEUE.push(scratch.data_ptr() + 4 * B * strips * G * 2, db, dg, B * strips, 2 * C, C, scratch)
        # shortcut weight / bias gradients
        dw = dbias = None
        want_db = ctx.has_bias and needs[9]
        if needs[10]:
            assert pack.grad_out is not None, "CatNormShortcutFn: the weight gradient needs its preassigned buffer"
            if want_db and pack.bias_grad_out is None:
                dbias = torch.zeros(pack.n, dtype=F32, device=dy.device)
            dbp = (pack.bias_grad_out if pack.bias_grad_out is not None else dbias) if want_db else None
            if _WGRAD_QUEUE is not None:
                _WGRAD_QUEUE.push_cat(dr2, x1.view(-1, C1), x2.view(-1, C2), pack.grad_out, dbp, pack.on_grad_ready)
            else:           # no deferred queue: the single-job entry point has no two-source form
                wgrad(dr2, torch.cat([x1.view(-1, C1), x2.view(-1, C2)], 1), C, pack.n, out=pack.grad_out, accumulate=2, dbias=dbp)
                if pack.on_grad_ready is not None:
                    pack.on_grad_ready()
        elif want_db:
            dbias = colsum(dr2)[0]
        gg, gb = (None, None) if (direct or frozen) else (dg, db)
        return dx1, dx2, gg, gb, None, None, None, None, None, dbias, dw, None, None


def cat_train_ready(pack, weight):
    """CatNormShortcutFn can produce the weight gradient of this shortcut (or none is wanted)."""
    return (not weight.requires_grad) or pack.grad_out is not None


# per-sample column sums of an input gradient that a GroupNorm backward produced along with it, keyed by the gradient's
# address: the convolution whose output that norm read takes its per-sample bias gradient from here (ConvFn.backward)
_DX_COLSUMS = {}


class GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu, gdst=None, passthrough=False, colsum=False, pre=None):
        """gdst: optional (dgamma, dbeta) fp32 views the parameter gradients are written into directly.
        passthrough: also return x itself, for the branch that reads x beside the norm (residual / shortcut);
        backward then adds that branch's gradient inside the normalisation kernel instead of a separate add.
        pre: the statistics x's producer emitted with it (`take_gn_pre`), if any."""
        B, H, W, C = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((B, groups, 2), dtype=F32, device=x.device)
        groupnorm_forward(x, None, gamma, beta, y, stats, groups, eps, silu, pre)
        ctx.save_for_backward(x, stats, gamma, beta)
        ctx.groups, ctx.silu, ctx.gdst = groups, silu, gdst
        # colsum: x is the output of a convolution with a per-sample bias; backward leaves that bias's gradient (the
        # per-sample column sums of dx) in _DX_COLSUMS for the convolution's backward
        ctx.colsum = colsum
        if passthrough:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        x, stats, gamma, beta = ctx.saved_tensors
        dy = dy.contiguous()
        if dpass is not None:
            dpass = (dpass if dpass.dtype == BF16 else dpass.to(BF16)).contiguous()
        B, H, W, C = x.shape
        G = ctx.groups
        dx = torch.empty_like(x)
        direct = ctx.gdst is not None
        frozen = not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])      # affine parameters without gradient
        dg, db = ctx.gdst if direct else ((None, None) if frozen else (torch.empty_like(gamma), torch.empty_like(beta)))
        nsc = int(lib.mi355_groupnorm_scratch_floats(B, H * W, C, G))
        scratch = torch.empty(nsc, dtype=F32, device=x.device)
        deferred = direct and not frozen and _REDUCE_QUEUE is not None
        if ctx.colsum:
            sums = torch.empty((B, C), dtype=F32, device=x.device)
            sc2 = torch.empty(nsc, dtype=F32, device=x.device)
            check(lib.mi355_groupnorm_bwd_colsum(x.data_ptr(), dy.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                 dx.data_ptr(), None if deferred else _ptr(dg), None if deferred else _ptr(db),
                                                 scratch.data_ptr(), B, H * W, C, G, int(ctx.silu), _ptr(dpass), sums.data_ptr(),
                                                 sc2.data_ptr(), _stream()), "groupnorm_bwd_colsum")
            _DX_COLSUMS.clear()             # at most one pending entry: the producing convolution's backward runs next
            _DX_COLSUMS[dx.data_ptr()] = sums
        else:
            check(lib.mi355_groupnorm_bwd(x.data_ptr(), dy.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          dx.data_ptr(), None if deferred else _ptr(dg), None if deferred else _ptr(db),
                                          scratch.data_ptr(), B, H * W, C, G, int(ctx.silu), _ptr(dpass), _stream()), "groupnorm_bwd")
        if deferred:        # per-strip channel partials follow the group partials in scratch
            strips = nsc // (B * (2 * G + 2 * C))
            _REDUCE_QUEUE.push(scratch.data_ptr() + 4 * B * strips * G * 2, db, dg, B * strips, 2 * C, C, scratch)
        if direct or frozen:
            return dx, None, None, None, None, None, None, None, None, None
        return dx, dg, db, None, None, None, None, None, None, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, gdst=None, passthrough=False):
        """gdst: optional fp32 [2][C] view (dgamma row, dbeta row) written directly.
        passthrough: also return x for the residual branch; its gradient is added inside the backward kernel."""
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        stats = torch.empty((M, 2), dtype=F32, device=x.device)
        check(lib.mi355_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                      M, C, eps, _stream()), "layernorm_fwd")
        ctx.save_for_backward(x, stats, gamma)
        ctx.gdst = gdst
        if passthrough:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        x, stats, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        if dpass is not None:
            dpass = (dpass if dpass.dtype == BF16 else dpass.to(BF16)).contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        direct = ctx.gdst is not None
        frozen = not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dgb = ctx.gdst if direct else (None if frozen else torch.empty((2, C), dtype=F32, device=x.device))
        nsc = int(lib.mi355_layernorm_bwd_scratch_floats(M, C))
        scratch = torch.empty(nsc, dtype=F32, device=x.device)
        deferred = direct and not frozen and _REDUCE_QUEUE is not None
        check(lib.mi355_layernorm_bwd(x.data_ptr(), dy.data_ptr(), stats.data_ptr(), gamma.data_ptr(), dx.data_ptr(),
                                      None if deferred else _ptr(dgb), scratch.data_ptr(), M, C, _ptr(dpass), _stream()),
              "layernorm_bwd")
        if deferred:
            _REDUCE_QUEUE.push(scratch.data_ptr(), dgb, None, nsc // (2 * C), 2 * C, 0, scratch)
        if direct or frozen:
            return dx, None, None, None, None, None
        return dx, dgb[0], dgb[1], None, None, None


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v per head, no mask.

    Three operand layouts (the gradient comes back in the same layout, so fused projections need no
    slicing / concatenation pass):
      a = qkv [B,N,3C], b = c = None : output of the fused q/k/v projection (self-attention)
      a = q [B,N,C], b = kv [B,T,2C], c = None : fused k/v projection of the text embeddings
      a = q, b = k, c = v : three separate tensors
    """

    @staticmethod
    def _split(a, b, c):
        if b is None:
            C = a.shape[2] // 3
            return a[:, :, :C], a[:, :, C:2 * C], a[:, :, 2 * C:]
        if c is None:
            C = a.shape[2]
            return a, b[:, :, :C], b[:, :, C:]
        return a, b, c

    @staticmethod
    def forward(ctx, a, b, c, heads):
        q, k, v = AttentionFn._split(a, b, c)
        o, lse = attention_fwd(q, k, v, heads)
        ctx.save_for_backward(a, b, c, o, lse)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, do):
        a, b, c, o, lse = ctx.saved_tensors
        do = do.contiguous()
        da = torch.empty_like(a)
        db = None if b is None else torch.empty_like(b)
        dc = None if c is None else torch.empty_like(c)
        q, k, v = AttentionFn._split(a, b, c)
        dq, dk, dv = AttentionFn._split(da, db, dc)
        attention_bwd(q, k, v, o, do, lse, ctx.heads, dq, dk, dv)
        return da, db, dc, None


class GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        F2 = x.shape[-1]
        M = x.numel() // F2
        y = torch.empty((*x.shape[:-1], F2 // 2), dtype=BF16, device=x.device)
        check(lib.mi355_geglu_fwd(x.data_ptr(), y.data_ptr(), M, F2 // 2, _stream()), "geglu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        F2 = x.shape[-1]
        M = x.numel() // F2
        dx = torch.empty_like(x)
        check(lib.mi355_geglu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), M, F2 // 2, _stream()), "geglu_bwd")
        return dx


class SiluF32ToBf16Fn(torch.autograd.Function):
    """fp32 in -> bf16 out (timestep-embedding MLP activations)."""

    @staticmethod
    def forward(ctx, x):
        y = torch.empty(x.shape, dtype=BF16, device=x.device)
        check(lib.mi355_silu_f32_to_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "silu")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dyf = dy.to(F32).contiguous()
        dx = torch.empty_like(x)
        check(lib.mi355_silu_bwd_f32(x.data_ptr(), dyf.data_ptr(), dx.data_ptr(), x.numel(), _stream()), "silu_bwd")
        return dx


class MseLossFn(torch.autograd.Function):
    """mean((pred - target)^2) with the gradient produced in the same pass
    (reference objective: /root/reference/generator/image_generator.py:125)."""

    @staticmethod
    def forward(ctx, pred, target):
        pred = _req(pred, F32, "mse pred")
        target = _req(target.to(F32), F32, "mse target")
        loss = torch.empty((), dtype=F32, device=pred.device)
        grad = torch.empty_like(pred) if pred.requires_grad else None
        check(lib.mi355_mse_loss_fused(pred.data_ptr(), target.data_ptr(), _ptr(grad), loss.data_ptr(), pred.numel(),
                                       _stream()), "mse_loss")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, dl):
        return ctx.grad * dl, None


def mse_loss(pred, target):
    return MseLossFn.apply(pred, target)


def timestep_embedding(t_float, dim):
    B = t_float.shape[0]
    out = torch.empty((B, dim), dtype=BF16, device=t_float.device)
    check(lib.mi355_timestep_embed(t_float.data_ptr(), out.data_ptr(), B, dim, _stream()), "timestep_embed")
    return out


def nchw_to_nhwc_bf16(x, cpad):
    B, C, H, W = x.shape
    x = _req(x.to(F32), F32, "nchw input")
    y = torch.empty((B, H, W, cpad), dtype=BF16, device=x.device)
    check(lib.mi355_nchw_f32_to_nhwc_bf16(x.data_ptr(), y.data_ptr(), B, C, cpad, H * W, _stream()), "nchw_to_nhwc")
    return y


class ToNHWCFn(torch.autograd.Function):
    """fp32 NCHW [B,C,H,W] -> bf16 NHWC [B,H,W,cpad] (zero-padded channels)."""

    @staticmethod
    def forward(ctx, x, cpad):
        ctx.c = x.shape[1]
        return nchw_to_nhwc_bf16(x, cpad)

    @staticmethod
    def backward(ctx, dy):
        return dy[..., :ctx.c].permute(0, 3, 1, 2).to(F32), None


class ToNCHWFn(torch.autograd.Function):
    """fp32 NHWC [B,H,W,Cs] -> fp32 NCHW [B,Cd,H,W] keeping the first Cd channels."""

    @staticmethod
    def forward(ctx, x, cd):
        B, H, W, Cs = x.shape
        y = torch.empty((B, cd, H, W), dtype=F32, device=x.device)
        check(lib.mi355_nhwc_f32_to_nchw_f32(x.data_ptr(), y.data_ptr(), B, Cs, cd, H * W, _stream()), "nhwc_to_nchw")
        ctx.cs = Cs
        return y

    @staticmethod
    def backward(ctx, dy):
        g = dy.permute(0, 2, 3, 1)
        return torch.nn.functional.pad(g, (0, ctx.cs - g.shape[-1])).contiguous(), None
