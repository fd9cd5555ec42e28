"""Reduce one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) over bench.py --no-graph to the matrix-pipe
utilisation of every contraction-kernel family of the train step: MFMA-busy cycles summed over the SIMDs divided by
(1024 SIMDs x the kernels' active cycles), GRBM_GUI_ACTIVE being the sum over the 8 XCDs.
usage: pmc_step_mfma.py <dir with the pass's csv> [marker kernel [command text]]
The marker (default adamw_kernel) is the launch that ends a step: ddim_step_cfg_kernel for the sampling loop."""
import csv, glob, json, sys, collections

FAMS = [("gemm_conv (fwd + dgrad)", ("gemm_dma_kernel", "gemm_wide_kernel", "gemm_pp_kernel", "gemm_kernel", "gemm_skinny_kernel")),
        ("wgrad (grouped)", ("wgrad_",)), ("attention fwd", ("attn_fwd",)), ("attention bwd", ("attn_bwd",))]


def family(name):
    for fam, keys in FAMS:
        if any(k in name for k in keys):
            return fam
    return None


f = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
busy, act, cnt = collections.Counter(), collections.Counter(), collections.Counter()
per_kernel = collections.defaultdict(lambda: [0.0, 0.0, 0])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
# whole steps only: the launches between the first and the last optimizer launch of the pass
MARKER = sys.argv[2] if len(sys.argv) > 2 else "adamw_kernel"
COMMAND = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --no-graph --steps 2 --warmup 1 --prof-steps 0 --no-cpu-baseline"
marks = sorted({int(r["Dispatch_Id"]) for r in rows if MARKER in r["Kernel_Name"]})
if len(marks) < 2:
    raise SystemExit("fewer than two step-ending launches in the pass")
steps = len(marks) - 1
for r in rows:
    if not (marks[0] < int(r["Dispatch_Id"]) <= marks[-1]):
        continue
    name, c, v = r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
    fam = family(name)
    if fam is None:
        continue
    short = name.split("(")[0][-44:]
    if c == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[fam] += v; per_kernel[short][0] += v
    elif c == "GRBM_GUI_ACTIVE":
        act[fam] += v; cnt[fam] += 1; per_kernel[short][1] += v; per_kernel[short][2] += 1
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- " + COMMAND,
       "steps_in_run": steps, "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)", "families": {}, "kernels": {}}
for fam, _ in FAMS:
    if act[fam]:
        out["families"][fam] = {"launches_per_step": cnt[fam] / max(steps, 1), "mfma_busy_Mcycles_per_step": busy[fam] / max(steps, 1) / 1e6,
                                "active_Mcycles_per_step": act[fam] / 8 / max(steps, 1) / 1e6, "mfma_utilisation": busy[fam] / (1024 * act[fam] / 8)}
for k, (b, a, n) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
    if a:
        out["kernels"][k] = {"launches_per_step": n / max(steps, 1), "active_Mcycles_per_step": a / 8 / max(steps, 1) / 1e6, "mfma_utilisation": b / (1024 * a / 8)}
print(json.dumps(out, indent=1))
