"""bench.py's N > 1 plumbing, rehearsed on the one-GPU box: `python bench.py --gpus 2` must start its
two ranks itself (self_launch -> torch.distributed.run), run the data-parallel step with the reducer attached, and relay
exactly ONE JSON line from rank 0.  With MVLT_BENCH_BACKEND=gloo both ranks share cuda:0 (RCCL refuses two ranks on one
device); the driver's SCALE runs take the default backend (RCCL, one rank per GPU) through the same code."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_self_launch_relays_one_json_line():
    env = dict(os.environ, MVLT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-extra", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp2"
    assert "gloo all-reduce" in out["config"]["grad_exchange"] and "GLOBAL batch" in out["config"]["grad_exchange"]
    assert out["value"] > 0 and out["ms_per_step"] > 0
    loss = out["config"]["loss"]
    assert loss == loss and 5.0 < loss < 15.0          # ~ln(30522) + ln 2 on random-init weights


def test_bench_one_rank_over_rccl_with_and_without_the_collective():
    """`other_configs.one_rank_rccl` of the default bench line: the step with the gradient reducer over RCCL
    on ONE rank (MVLT_FORCE_DDP=1), and the same without the collective itself (MVLT_DDP_NULL_COLLECTIVE=1) -- exactly one JSON
    line each (RCCL prints a banner on stdout: bench.py re-points file descriptor 1 for the run), a finite loss, and the
    hardware-queue default in place."""
    for extra in ({}, {"MVLT_DDP_NULL_COLLECTIVE": "1"}):
        env = dict(os.environ, MVLT_FORCE_DDP="1", **extra)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GPU_MAX_HW_QUEUES"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extra",
                            "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
        assert len(lines) == 1, r.stdout[-2000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == 1 and out["value"] > 0
        loss = out["config"]["loss"]
        assert loss == loss and 5.0 < loss < 15.0
