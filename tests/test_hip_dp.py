"""GPU test of the data-parallel step: two ranks (one process each, here sharing the single GPU of the test box and
talking over gloo -- RCCL needs one device per rank) run dp.GradientBuckets + the SyncBN all-reduces on identical shards and must
reproduce the single-process parameters after two SGD steps."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run(cmd, env):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("steps,tol", [(1, 2e-5), (2, 5e-4)])
def test_two_rank_step_matches_single_process(tmp_path, steps, tol):
    """one step: only summation-order noise (run-to-run 1e-6 of the parameter scale); two steps: that noise has been through a
    second forward/backward (run-to-run 4e-5)"""
    worker = os.path.join(HERE, "dp_worker.py")
    env = dict(os.environ, DP_WORKER_STEPS=str(steps))
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    single = str(tmp_path / "single.pt")
    run([sys.executable, worker, single], env)
    env2 = dict(env, SAUNET_DIST_BACKEND="gloo", SAUNET_SHARE_GPU="1")
    double = str(tmp_path / "double.pt")
    run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(29533 + steps), worker, double], env2)
    a, b = torch.load(single), torch.load(double)
    assert a["world"] == 1 and b["world"] == 2
    assert abs(a["losses"][0] - b["losses"][0]) < 1e-5 * abs(a["losses"][0])
    scale = float(a["params"].abs().max())
    assert float((a["params"] - b["params"]).abs().max()) < tol * scale
    # running statistics: identical except the unbiased-variance factor of the 6 SyncBN layers (n/(n-1) with the global count)
    assert float((a["running"] - b["running"]).abs().max()) < 1e-3 * float(a["running"].abs().max())
