"""bench.py's multi-rank control flow, exactly as the driver types it, on the one-GPU box: every branch that an 8-GPU node
would take first is taken here once per run.

  * `python bench.py --gpus 2` with one GPU visible: the launching process counts devices through a child, refuses with
    exit status 2 and a message, and measures nothing;
  * the launcher with one rank: counts, starts `python -m torch.distributed.run --nproc-per-node 1 ... bench.py`, the
    rank prints one JSON line;
  * two ranks sharing the GPU (tools/bench_two_ranks_rehearsal.py: the collective carried by gloo because RCCL refuses
    two ranks on one device, everything else bench.py's own code): launcher, reducer, forward-pass graph, timed loop,
    max over ranks, one JSON line from rank 0 with the gradient-exchange summary.

Each case is a fresh child process (a launching process must not have touched a GPU), run under a time limit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=420):
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, *cmd], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    return r, lines


def test_more_ranks_than_gpus_is_refused_with_status_2():
    import torch
    n = torch.cuda.device_count() + 1
    r, lines = _run(["bench.py", "--gpus", str(n), "--small", "--steps", "2", "--warmup", "2"])
    assert r.returncode == 2, (r.returncode, r.stderr[-2000:])
    assert "GPU(s) are visible" in r.stderr and not lines


def test_launcher_starts_one_rank_and_the_rank_reports():
    code = ("import sys, bench; sys.exit(bench.launch_ranks(1, ['--gpus', '1', '--small', '--steps', '2', '--warmup', '2', "
            "'--no-cpu-baseline', '--no-extra', '--prof-steps', '1']))")
    r, lines = _run(["-c", code])
    assert r.returncode == 0, r.stderr[-3000:]
    assert "starting 1 rank(s)" in r.stderr
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["value"] > 0 and lines[0]["gradient_exchange"] is None


def test_two_rank_rehearsal_reports_one_line_for_world_size_2():
    r, lines = _run(["tools/bench_two_ranks_rehearsal.py", "--small", "--steps", "2", "--warmup", "2", "--prof-steps", "1"])
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert len(lines) == 1, r.stdout[-2000:]
    line = lines[0]
    assert line["world_size"] == 2 and line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2"
    assert line["config"]["global_batch"] == 2 * line["config"]["per_gpu_batch"] and line["value"] > 0
    ex = line["gradient_exchange"]
    assert ex["world"] == 2 and ex["buckets"] >= 1 and ex["bucket_bytes_bf16"]["total"] > 0 and ex["wgrad_flush_blocks"] == 4096
    assert line["rank_ms_per_step"]["max"] >= line["rank_ms_per_step"]["min"] > 0
