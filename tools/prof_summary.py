"""Summarise a rocprofv3 --kernel-trace --stats csv directory: per-kernel totals per step."""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {tot/1e6/steps:.2f} ms  ({f})")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{r['Name'][:64]:64s} calls/step={float(r['Calls'])/steps:7.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):5.1f}%")
