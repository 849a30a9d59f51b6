"""GPU: `python bench.py --gpus 2` with NO launcher around it starts two ranks by itself and reports them.
One GPU here, so the collective backend is gloo (ranks share the device: control flow, sharding, barrier, max over ranks and the
rank-0 JSON line are what is checked -- RCCL itself needs a multi-GPU node)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=600):
    env = dict(os.environ, V3D_BENCH_BACKEND="gloo", V3D_BENCH_MAX_PIPELINE="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--no-cpu-baseline"] + extra,
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [
    ["--steps", "5", "--warmup", "2", "--windows", "1", "--no-roofline", "--no-h2d"],
    ["--workload", "waymo", "--steps", "3", "--warmup", "1", "--windows", "1", "--no-roofline", "--no-h2d"],
    ["--mode", "train", "--steps", "2", "--warmup", "1"],
], ids=["forward", "waymo", "train"])
def test_gpus_2_without_a_launcher_runs_two_ranks(extra):
    d = _run(extra)
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2
    assert d["value"] > 0 and d["scaling"] == "weak"


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0", V3D_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr
