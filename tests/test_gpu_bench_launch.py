"""`python bench.py --gpus N` launches its own ranks: the parent touches no GPU API, starts N fresh processes through
`python -m torch.distributed.run` as a child and exits with its code.  Dry run of the N = 2 path on the one GPU of the test box: gloo backend,
both ranks mapped onto GPU 0 (`--share-gpus`); on an 8-GPU node the driver's `python bench.py --gpus 8` takes the same path with RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpus", "--steps", "2", "--warmup", "1",
           "--cpu-forwards", "0", "--sustained-seconds", "0", "--no-kernel-timers"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints the one JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"]["world_size"] == 2 and line["ranks"]["backend"] == "gloo"
    assert line["scaling"] == "weak" and line["steps"] == 2 and line["value"] > 0
    assert line["config"]["members_per_gpu"] == 1  # weak scaling: 2 members in all, one per rank


def test_bench_strong_scaling_fixed_ensemble_over_two_ranks():
    """`--ensemble-size E` (north star: ">= 6x at 8 GPUs vs 1" is quoted on a FIXED ensemble, BASELINE configs[2]): 3 members dealt to 2
    ranks (2 + 1), and a 1-member ensemble on 2 ranks (rank 1 owns nothing: launches nothing, joins the gather)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for E, want in ((3, [2, 1]), (1, [1, 0])):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpus", "--steps", "1", "--warmup", "0",
               "--cpu-forwards", "0", "--sustained-seconds", "0", "--no-kernel-timers", "--ensemble-size", str(E), "--solver-steps", "4"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        line = json.loads(lines[0])
        assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["value"] > 0
        assert line["config"]["ensemble_size"] == E and line["config"]["members_on_rank"] == want and line["config"]["members_per_gpu"] is None
        assert abs(line["value"] - E * 1 * 1 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3  # value = E x lead steps x steps / time


def test_bench_rejects_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=3 does not match --gpus 2" in (r.stderr + r.stdout)
