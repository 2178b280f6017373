"""GPU test of the data-parallel step: K ranks, one process each, every rank on its OWN shard, run the real kernels +
dp.GradientBuckets (hooks, bucketed all-reduce, unpack) + the SyncBN all-reduces; rank 0's parameters and running statistics after
one and two SGD steps must equal the oracle's K-replica CPU emulation (oracle.saunet_ref.dp_emulate_step: per-shard losses averaged,
local statistics in the 144 BatchNorm layers, global statistics + the reference's accumulator in the 6 SyncBN layers).

Transports: RCCL with one GPU per rank when the box has >= 2 GPUs; otherwise the ranks share the single GPU and talk over gloo (the
collective calls, bucket logic and kernels are the same code; only the backend differs)."""
import os
import subprocess
import sys

import pytest
import torch

from oracle import saunet_ref as R, weights as Wt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def run(cmd, env):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def emulate(world, steps, shards=None):
    """the K-replica CPU emulation with the worker's weights, shards and optimiser (SGD: decay on conv weights only, train.py:166-196)"""
    import dp_worker as W
    spec = R.state_dict_spec()
    sd = Wt.make_state_dict(spec, W.SEED)
    keys = Wt.trainable_keys(spec)
    for k in keys:
        sd[k].requires_grad_(True)
    decay = [sd[k] for k, _, kind in spec if kind == "conv"]
    rest = [sd[k] for k, _, kind in spec if kind in ("bias", "gamma", "beta")]
    opt = torch.optim.SGD([dict(params=decay), dict(params=rest, weight_decay=0.0)], lr=W.LR, momentum=W.MOM, weight_decay=W.WD)
    shards = shards or [W.shard(r) for r in range(world)]
    per_shard = []
    for _ in range(steps):
        opt.zero_grad()
        loss, losses, _, _ = R.dp_emulate_step(sd, shards, True)
        loss.backward(); opt.step()
        per_shard.append([float(l) for l in losses])
    return sd, keys, per_shard


def launch(tmp_path, world, steps, backend):
    worker = os.path.join(HERE, "dp_worker.py")
    env = dict(os.environ, DP_WORKER_STEPS=str(steps))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if backend == "gloo":
        env.update(SAUNET_DIST_BACKEND="gloo", SAUNET_SHARE_GPU="1")
    out = str(tmp_path / ("dp_%s_%d.pt" % (backend, steps)))
    run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
         "--master-port", str(29533 + steps + (7 if backend == "nccl" else 0)), worker, out], env)
    return torch.load(out)


def check(res, world, steps, tol):
    import dp_worker as W
    assert res["world"] == world
    sd, keys, per_shard = emulate(world, steps)
    init = Wt.make_state_dict(R.state_dict_spec(), W.SEED)
    # per-rank losses of every step (each rank evaluates ITS shard)
    for r in range(world):
        for s in range(steps):
            assert abs(res["losses"][r][s] - per_shard[s][r]) < 2e-4 * max(1.0, abs(per_shard[s][r])), (r, s, res["losses"][r][s], per_shard[s][r])
    upd = max(float((sd[k].detach() - init[k]).abs().max()) for k in keys)          # scale of the parameter update
    worst = max(((float((res["params"][k] - sd[k].detach()).abs().max()), k) for k in keys))
    assert worst[0] < tol * upd, (worst, upd)
    # the test must be able to FAIL: replica 0 alone (no gradient averaging, local SyncBN statistics) lands far outside the tolerance
    solo, _, _ = emulate(1, steps, shards=[W.shard(0)])
    gap = max(float((solo[k].detach() - sd[k].detach()).abs().max()) for k in keys)
    assert gap > 10 * tol * upd, (gap, upd)
    # running statistics: local BatchNorm = replica 0's own batch; SyncBN = global batch through the (_tmp_running_*, _running_iter) accumulator
    for k in ("encoder.features.denseblock1.denselayer2.norm1.running_mean", "dec4.c3x3rb.1.running_var", "res1.bn1.running_mean",
              "res2.bn2.running_var", "res3.bn1._running_iter", "res1.bn2._tmp_running_mean"):
        ref = sd[k].detach().float()
        assert float((res["buffers"][k] - ref).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max())), k
    # parameters off the compute path keep grad None on every rank (single-process behaviour), so the optimiser never touches them
    assert any(k.startswith("encoder.classifier") for k in res["grad_none"])
    assert torch.equal(res["params"]["encoder.classifier.bias"], torch.zeros_like(res["params"]["encoder.classifier.bias"]))


@pytest.mark.parametrize("steps,tol", [(1, 2e-3), (2, 1.2e-2)])     # measured 1 step 4e-4 / 2 steps 7.5e-3 of the update scale (the second step amplifies
# float32 summation-order noise through a loss of ~4 at lr 1e-2); a missing all-reduce is 30-100x larger (checked below)
def test_two_ranks_different_shards_match_dp_emulation_gloo(tmp_path, steps, tol):
    check(launch(tmp_path, 2, steps, "gloo"), 2, steps, tol)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (>= 2 GPUs)")
@pytest.mark.parametrize("world", [2, 4])
def test_rccl_ranks_match_dp_emulation(tmp_path, world):
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs" % world)
    res = launch(tmp_path, world, 2, "nccl")
    assert res["backend"] == "nccl"
    check(res, world, 2, 1.2e-2)


def test_bench_rehearsal_two_ranks_share_the_gpu():
    """`bench.py --gpus 2 --share-gpu`: the bench's own N > 1 protocol -- self-launch of one rank per requested GPU, parameter
    broadcast, hook-launched bucket all-reduces overlapped with the backward kernels, the first-step bucket rebuild (encoder.classifier has no
    gradient), the 12 SyncBN statistic all-reduces of res1-3, barrier + max-over-ranks timing -- with the REAL HIP step on both ranks, on the one
    GPU of this box over gloo.  The line is marked as a rehearsal and carries no multi-GPU value.  /root/reference/train.py:272-277 (one replica
    per device), lib/nn/modules/batchnorm.py:98-139 (SyncBN exchange)."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SAUNET_SHARE_GPU", "SAUNET_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "2", "--batch", "4",
                        "--size", "128", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["value"] is None and "rehearsal" in rec and rec["rehearsal_slices_per_s"] > 0
    assert rec["config"]["global_batch"] == 8 and rec["config"]["parallelism"] == "dp2"
    comm = rec["comm"]
    assert comm["backend"] == "gloo" and comm["rccl_ranks"] == 0 and comm["physical_gpus"] >= 1
    assert comm["replicas_identical"] is True
    # 31.9 M float32 gradients (127.5 MB) in buckets of at most 32 MB, cut at parameter boundaries in reverse registration order: 6 buckets
    # of 13.8 - 33.4 MB (a parameter that would overflow the cap opens the next bucket; center / dec5 hold 4.7 - 7.1 M-element tensors)
    assert comm["buckets"] == 6 and len(comm["bucket_table_last_step"]) == 6
    assert sum(row["MB"] for row in comm["bucket_table_last_step"]) == pytest.approx(comm["allreduce_payload_bytes"] / 1e6, rel=1e-3)
    assert comm["syncbn_allreduces_per_step"] == 12                                     # 6 SyncBN layers x (forward statistics + backward sums)
    # overlap: the compute stream waited for less than the exchange took from the first launch to the last completion
    assert comm["exposed_allreduce_ms_last_step"] is not None and comm["allreduce_span_ms_last_step"] is not None
    assert comm["exposed_allreduce_ms_last_step"] < comm["allreduce_span_ms_last_step"], comm
    launches = [row["launch_ms"] for row in comm["bucket_table_last_step"]]
    assert launches == sorted(launches) and launches[-1] > launches[0]                  # buckets leave one by one as backward produces them


def test_bench_rehearsal_keeps_the_collective_roofline_census():
    """the roofline census of `bench.py` drives real training steps, i.e. SyncBN and gradient-bucket all-reduces -- with N > 1 it
    must run on EVERY rank.  Two ranks sharing this box's GPU over gloo with the census left ON: the run must finish (rank 0 alone used to hang in
    its first all-reduce) and rank 0's line must carry a populated `roofline`."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SAUNET_SHARE_GPU", "SAUNET_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--rehearsal-roofline", "--steps", "2", "--warmup", "2",
                        "--batch", "4", "--size", "128", "--no-cpu-baseline", "--no-launch-mix"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["comm"]["replicas_identical"] is True
    roof = rec["roofline"]
    assert "error" not in roof, roof
    assert roof["kernels"] and roof["ms"] > 0 and roof["launches"] > 0
