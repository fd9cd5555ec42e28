"""Fetched / written bytes against operand bytes for the train step's contraction shapes (forward and data-gradient
forms, batch 32), one launch each on cold caches, for rocprofv3 --pmc passes:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir>/FETCH_SIZE -o p -- python tools/pmc_gemm_shapes.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir>/WRITE_SIZE -o p -- python tools/pmc_gemm_shapes.py
    python tools/pmc_gemm_shapes.py --reduce <dir>

Every shape is preceded by a marker (a 64 KiB streaming copy) and a 512 MiB clearing pass that empties L2 and the
Infinity Cache; the reducer sums the GEMM-family launches (tile kernel + split-K reduction) between two markers.
fetch = 2 x FETCH_SIZE KiB (gfx950 tallies wide reads at half their size), write = WRITE_SIZE KiB, both at the L2's memory side."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

B = 32
# (label, kind, rows-per-sample side, Cin, Cout): conv = 3x3 forward, dgrad = its data gradient from the weight as stored,
# dense / dense_bt = linear layer forward / data gradient, geglu = gated feed-forward input projection
SHAPES = [("conv", 32, 320, 320), ("conv", 32, 640, 320), ("conv", 32, 960, 320), ("conv", 16, 640, 640), ("conv", 16, 1280, 640),
          ("conv", 16, 1920, 640), ("conv", 8, 1280, 1280), ("conv", 8, 2560, 1280), ("conv", 4, 1280, 1280), ("conv", 4, 2560, 1280),
          ("dgrad", 16, 640, 640), ("dgrad", 8, 1280, 1280), ("dgrad", 8, 2560, 1280),
          ("dense", 32, 320, 960), ("dense", 32, 320, 320), ("geglu", 32, 320, 2560), ("dense", 32, 1280, 320),
          ("dense", 16, 640, 1920), ("dense", 16, 640, 640), ("geglu", 16, 640, 5120), ("dense", 16, 2560, 640),
          ("dense", 8, 1280, 3840), ("dense", 8, 1280, 1280), ("geglu", 8, 1280, 10240), ("dense", 8, 5120, 1280),
          ("dense_bt", 16, 640, 1920), ("dense_bt", 8, 1280, 3840), ("dense_bt", 8, 1280, 1280)]


def operand_bytes(kind, side, cin, cout):
    m = B * side * side
    if kind in ("conv", "dgrad"):
        return 2 * (m * cin + 9 * cin * cout), 2 * m * cout
    if kind == "geglu":
        return 2 * (m * cin + cin * cout), 2 * m * cout // 2
    if kind == "dense_bt":
        return 2 * (m * cout + cin * cout), 2 * m * cin
    return 2 * (m * cin + cin * cout), 2 * m * cout


def run():
    import torch
    from image_generator_amd import ops
    from image_generator_amd._lib import lib
    d = torch.device("cuda:0")
    BF = torch.bfloat16
    flush = torch.empty(128 << 20, dtype=torch.float32, device=d)
    ms, md = torch.zeros(16384, device=d), torch.zeros(16384, device=d)
    cs = torch.cuda.current_stream().cuda_stream
    for rep in range(2):                     # the first round loads the code objects; the reducer takes the last
        for kind, side, cin, cout in SHAPES:
            m = B * side * side
            if kind == "conv":
                x, w = torch.randn(B, side, side, cin, device=d).to(BF), torch.randn(cout, 3, 3, cin, device=d).to(BF)
                fn = lambda: ops.conv2d(x, w, cout, 3)
            elif kind == "dgrad":              # dy [.., cout] -> dx [.., cin] through w [cout][3][3][cin] as stored
                x, w = torch.randn(B, side, side, cout, device=d).to(BF), torch.randn(cout, 3, 3, cin, device=d).to(BF)
                fn = lambda: ops.conv2d_dgrad_bt(x, w, cin, 3)
            elif kind == "geglu":
                x, w = torch.randn(m, cin, device=d).to(BF), torch.randn(cout, cin, device=d).to(BF)
                fn = lambda: ops.gemm_geglu(x, w, None)
            elif kind == "dense":
                x, w = torch.randn(m, cin, device=d).to(BF), torch.randn(cout, cin, device=d).to(BF)
                fn = lambda: ops.gemm(x, w)
            else:                              # dy [m][cout] . w [cout][cin] as stored
                x, w = torch.randn(m, cout, device=d).to(BF), torch.randn(cout, cin, device=d).to(BF)
                fn = lambda: ops.gemm_bt(x, w)
            flush.zero_()
            lib.mi355_peak_copy(ms.data_ptr(), md.data_ptr(), 65536, cs)
            fn()
            torch.cuda.synchronize()
    lib.mi355_peak_copy(ms.data_ptr(), md.data_ptr(), 65536, cs)
    torch.cuda.synchronize()


def reduce(d):
    fam = ("gemm_dma_kernel", "gemm_wide_kernel", "gemm_pp_kernel", "gemm_kernel", "splitk_reduce_kernel")
    tot = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{d}/{counter}/**/*counter_collection.csv", recursive=True)[0]
        rows = sorted((r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
        seg, segs = None, []
        for r in rows:
            if "peak_copy_kernel" in r["Kernel_Name"]:
                if seg is not None:
                    segs.append(seg)
                seg = [0.0, []]
            elif seg is not None and any(k in r["Kernel_Name"] for k in fam):
                seg[0] += float(r["Counter_Value"]) * 1024.0
                seg[1].append(r["Kernel_Name"].split("(")[0].split(" ")[-1][:40])
        tot[counter] = segs[-len(SHAPES):]
    print("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python tools/pmc_gemm_shapes.py ; python tools/pmc_gemm_shapes.py --reduce <dir>")
    print("# one launch per shape on cold caches, batch 32; fetched = 2 x FETCH_SIZE KiB, written = WRITE_SIZE KiB (L2 memory side)")
    for (kind, side, cin, cout), fs, ws in zip(SHAPES, tot["FETCH_SIZE"], tot["WRITE_SIZE"]):
        rd, wr = operand_bytes(kind, side, cin, cout)
        fetched, written = 2.0 * fs[0], ws[0]
        print(f"{kind:8s} {side:2d}x{side:<2d} {cin:5d}->{cout:<5d}: reads {rd / 1e6:7.1f} MB, fetched {fetched / 1e6:7.1f} MB = {fetched / rd:5.2f} x; "
              f"writes {wr / 1e6:6.1f} MB, written {written / 1e6:7.1f} MB = {written / wr:5.2f} x   [{' + '.join(sorted(set(fs[1])))}]")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
        reduce(sys.argv[2])
    else:
        run()
