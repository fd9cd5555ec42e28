"""mi355_groupnorm_fwd_static (no scratch argument, no allocation) against mi355_groupnorm_fwd, through the C ABI.

Two calls in a row on the same GPU, the second on a smaller input: the built-in device array keeps the first call's
partial sums, and the second call must not see them.  The results must be bit-identical to the entry point that takes
a caller-provided scratch buffer (same kernels, same order of additions), and the over-size case must be refused with a
message instead of overrunning the array."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _run_both(B, HW, C, G, silu, seed):
    from image_generator_amd._lib import lib, check
    from image_generator_amd.ops import _stream
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, HW, C, generator=g) * 1.7 + 0.3).to(BF).to(d)
    gamma, beta = torch.randn(C, generator=g).to(d), torch.randn(C, generator=g).to(d)
    # every output buffer exists before the call under test: the entry point itself must not need an allocation
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    s0, s1 = torch.empty(B, G, 2, device=d), torch.empty(B, G, 2, device=d)
    scratch = torch.empty(int(lib.mi355_groupnorm_scratch_floats(B, HW, C, G)), device=d)
    check(lib.mi355_groupnorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y0.data_ptr(), s0.data_ptr(),
                                  scratch.data_ptr(), B, HW, C, G, 1e-5, int(silu), _stream()), "groupnorm_fwd")
    torch.cuda.synchronize()
    # torch's counter sees its own caching allocator only; the driver's free-byte figure (hipMemGetInfo) sees a hipMalloc
    # from anywhere in the process, the library included
    before = torch.cuda.memory_stats(d)["allocation.all.allocated"]
    free_before = torch.cuda.mem_get_info(d)[0]
    check(lib.mi355_groupnorm_fwd_static(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y1.data_ptr(), s1.data_ptr(),
                                         B, HW, C, G, 1e-5, int(silu), _stream()), "groupnorm_fwd_static")
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats(d)["allocation.all.allocated"] == before
    assert torch.cuda.mem_get_info(d)[0] == free_before, "device memory was allocated during the call"
    # and against plain fp32 math, so that "equal" cannot mean "equally wrong"
    xf = x.float().cpu().view(B, HW, G, C // G)
    mean, var = xf.mean((1, 3), keepdim=True), xf.var((1, 3), unbiased=False, keepdim=True)
    ref = ((xf - mean) / torch.sqrt(var + 1e-5)).view(B, HW, C) * gamma.cpu() + beta.cpu()
    if silu:
        ref = ref * torch.sigmoid(ref)
    return (y0, s0), (y1, s1), ref


def test_static_groupnorm_matches_scratch_entry_point_twice():
    # strip path (group width 10 channels): 4 x 1024 x 320 uses 4*32*32*2 floats of the array, then a smaller input
    for (B, HW, C, G, silu, seed) in [(4, 1024, 320, 32, True, 1), (2, 256, 320, 32, False, 2)]:
        (y0, s0), (y1, s1), ref = _run_both(B, HW, C, G, silu, seed)
        assert torch.equal(y0.view(torch.int16), y1.view(torch.int16)), "outputs differ from mi355_groupnorm_fwd"
        assert torch.equal(s0, s1), "statistics differ from mi355_groupnorm_fwd"
        err = (y1.float().cpu() - ref).abs()
        assert float((err / (1.0 + ref.abs())).max()) < 2 ** -6, float(err.max())


def test_static_groupnorm_runs_inside_a_stream_capture():
    """The callers it is for: under a stream capture an allocating HIP call is illegal and fails the capture.  The entry
    point is captured (after one eager call, which has looked the array's address up), replayed on new input values,
    and gives the scratch entry point's bits."""
    from image_generator_amd._lib import lib, check
    from image_generator_amd.ops import _stream
    d = torch.device("cuda:0")
    B, HW, C, G = 2, 1024, 320, 32
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, HW, C, generator=g).to(BF).to(d)
    gamma, beta = torch.randn(C, generator=g).to(d), torch.randn(C, generator=g).to(d)
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    s0, s1 = torch.empty(B, G, 2, device=d), torch.empty(B, G, 2, device=d)
    scratch = torch.empty(int(lib.mi355_groupnorm_scratch_floats(B, HW, C, G)), device=d)

    def static_call():
        check(lib.mi355_groupnorm_fwd_static(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y1.data_ptr(), s1.data_ptr(),
                                             B, HW, C, G, 1e-5, 1, _stream()), "groupnorm_fwd_static")
    static_call()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_call()
    x.copy_((torch.randn(B, HW, C, generator=g) * 2 - 0.5).to(BF))
    y1.zero_()
    graph.replay()
    check(lib.mi355_groupnorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y0.data_ptr(), s0.data_ptr(),
                                  scratch.data_ptr(), B, HW, C, G, 1e-5, 1, _stream()), "groupnorm_fwd")
    torch.cuda.synchronize()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16)) and torch.equal(s0, s1)


def test_static_groupnorm_rejects_oversize():
    from image_generator_amd._lib import lib
    from image_generator_amd.ops import _stream
    d = torch.device("cuda:0")
    one = torch.zeros(16, device=d)
    # 200 samples x 32 strips x 32 groups x 2 floats > 262144: refused before any launch (the pointers are never read)
    rc = lib.mi355_groupnorm_fwd_static(one.data_ptr(), one.data_ptr(), one.data_ptr(), one.data_ptr(), one.data_ptr(),
                                        200, 4096, 320, 32, 1e-5, 0, _stream())
    assert rc == -3
    assert b"built-in" in lib.mi355_last_error()
