"""Runs bench.py as a child process and prints its peak resident set size (sizing of the cpu_baseline sample)."""
import resource, subprocess, sys
r = subprocess.run([sys.executable, "-u", "bench.py"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
print(line[-1][line[-1].find('"cpu_baseline"'):] if line else r.stderr[-2000:])
print("peak RSS of the bench process: %.1f GiB" % (resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 2 ** 20))
