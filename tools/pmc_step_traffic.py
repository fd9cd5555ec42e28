"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py --no-graph to per-family HBM-side bytes
per launch and per step.  FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide reads at 64 bytes);
WRITE_SIZE is taken as reported.  Both are in KiB.  usage: pmc_step_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/>"""
import csv, glob, json, os, sys, collections

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

FAMS = [("gemm_conv", ("gemm_dma_kernel", "gemm_wide_kernel", "gemm_pp_kernel", "gemm_kernel", "splitk_reduce_kernel", "gemm_skinny_kernel")),
        ("wgrad", ("wgrad_",)), ("attention", ("attn_",)), ("adamw", ("adamw_kernel", "sumsq_")),
        ("norm", ("gn_", "ln_", "reduce_rows")), ]


def family(name):
    for fam, keys in FAMS:
        if any(k in name for k in keys):
            return fam
    return "other"


def load(d, counter):
    """Per-family totals over the launches BETWEEN the run's first and last optimizer launch (whole steps only: model
    initialisation, the first step's head and whatever follows the last optimizer launch - bench.py's own yardstick
    kernels - stay outside), and the number of steps that window holds."""
    f = glob.glob(f"{d}/{counter}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
    if len(marks) < 2:
        raise SystemExit(f"{counter}: fewer than two optimizer launches in the pass")
    tot, cnt = collections.Counter(), collections.Counter()
    for r in rows[marks[0] + 1:marks[-1] + 1]:
        fam = family(r["Kernel_Name"])
        tot[fam] += float(r["Counter_Value"]) * 1024.0
        cnt[fam] += 1
    return tot, cnt, len(marks) - 1


d = sys.argv[1]
ft, fc, fs = load(d, "FETCH_SIZE")
wt, wc, ws = load(d, "WRITE_SIZE")
from bench import source_fingerprint          # noqa: E402  (bench.py only reads it back if it matches the running code)

out = {"command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> -- python bench.py --no-graph --steps 2 --warmup 1 --prof-steps 0 --no-cpu-baseline --no-extra",
       "source_fingerprint": source_fingerprint(),
       "steps_in_run": fs, "window": "launches between the first and the last optimizer launch of the pass (whole steps; no initialisation, no yardstick kernels)",
       "note": "bytes at the L2's memory side (Infinity-Cache hits included); fetch = 2 x FETCH_SIZE (gfx950 wide-read correction)",
       "families": {}}
for fam in [f for f, _ in FAMS] + ["other"]:
    if fc[fam] == 0:
        continue
    launches = fc[fam] / max(fs, 1)
    fetch, write = 2.0 * ft[fam] / max(fs, 1), wt[fam] / max(ws, 1)
    out["families"][fam] = {"launches_per_step": launches, "fetch_bytes_per_step": fetch, "write_bytes_per_step": write,
                            "bytes_per_launch": (fetch + write) / max(launches, 1.0)}
print(json.dumps(out, indent=1))
