"""-m gpu: the N > 1 path on real HIP compute.  Two ranks share the box's one GPU (RCCL refuses two ranks per device, so
the process group is gloo and tensors cross it through host memory; the sharding, the per-rank operator calls and the
exact-size exchanges are the code an 8-GPU node runs over RCCL).  bench.py --check-union makes rank 0 compare the
exchanged result of the strong-scaling job with a single-rank run of the whole job: the pair multiset for the join,
whole columns in input order for count_overlaps / coverage / nearest, whole outputs in key order for merge / subtract
(what the reference pins for its own partitioned forms, R/tests/integration_test.rs:3709-3755, :3783-3890, :3923-4020)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(extra, probe_rows="20000000"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device",
           "--backend", "gloo", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--probe-rows", probe_rows] + extra
    # a fresh child: the launcher starts before anything touches the GPU
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_strong_scaling_union_equals_single_rank_join():
    out = _run(["--scaling", "strong", "--check-union"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["gather"] is True
    assert out["union_check"] == "gathered pair set == single-rank join"


def test_two_ranks_weak_scaling_line_and_strong_extra():
    out = _run(["--scaling", "weak"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["strong"]["value"] > 0 and out["strong"]["gather"] is True


@pytest.mark.parametrize("workload,rows", [("count_coverage_100Mx1M", "20000000"), ("nearest_50Mx50M", "8000000"),
                                           ("merge_subtract_200M", "20000000")])
def test_two_ranks_sharded_operator_equals_single_rank_run(workload, rows):
    out = _run(["--workload", workload, "--scaling", "strong", "--check-union"], rows)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["gather"] is True and out["value"] > 0
    assert out["union_check"] == "exchanged result == single-rank run of the whole job"


def test_two_ranks_weak_scaling_of_a_sweep_workload():
    out = _run(["--workload", "merge_subtract_200M", "--scaling", "weak"], "8000000")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["gather"] is False and out["value"] > 0
