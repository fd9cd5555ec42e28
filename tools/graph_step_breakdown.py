"""Per-kernel time and idle gaps of ONE graphed training step, from a rocprofv3 kernel trace of bench.py:
the window between the ends of two consecutive AdamW launches, `back` steps before the last one (bench.py ends
with its event-timed eager steps, so back = 2 selects the last graph replay).
usage: graph_step_breakdown.py <trace.csv> [top] [back]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
ad = [i for i, e in enumerate(ev) if "adamw" in e[2]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
a, b = ad[-2 - back], ad[-1 - back]
win = ev[a + 1:b + 1]
t0, t1 = ev[a][1], ev[b][1]
busy = sum(e[1] - e[0] for e in win)
gaps = sum(max(0, win[i + 1][0] - win[i][1]) for i in range(len(win) - 1)) + max(0, win[0][0] - t0)
print(f"step window {(t1 - t0) / 1e6:.2f} ms, {len(win)} launches, kernel time {busy / 1e6:.2f} ms, idle gaps {gaps / 1e6:.2f} ms")
d = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    k = n.split("(")[0][-48:]
    d[k][0] += e - s
    d[k][1] += 1
for k, (t, c) in sorted(d.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{k:50s} x{c:4d} {t / 1e6:7.3f} ms")
