"""bench.py's N-rank path end to end WITHOUT a GPU: `bench.py --gpus 2 --dry-run` launched exactly as the driver launches the multi-GPU bench
(python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...), over gloo.  Covers the environment bootstrap, the
parameter broadcast of replicas that start different, the gradient buckets of the REAL SAUNet parameter set with their backward hooks, the
averaged write-back, the barrier / max-over-ranks timing protocol and the JSON contract incl. the per-bucket exchange table."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_rank_dry_run_of_the_bench_contract():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run",
           "--bucket-mb", "16"]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in out, k
    assert out["dry_run"] is True and out["value"] is None and out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 64
    comm = out["comm"]
    assert comm["backend"] == "gloo" and comm["bucket_mb"] == 16.0
    # 32.9 M float32 parameters (incl. the unused classifier, which only real backward passes drop) in >= 8 buckets of <= 16 MB (+ one oversized tensor)
    assert comm["allreduce_payload_bytes"] == 4 * 32896505
    assert comm["buckets"] >= 8 and len(comm["bucket_table_last_step"]) == comm["buckets"]
    launches = [row["host_launch_ms"] for row in comm["bucket_table_last_step"]]
    assert launches == sorted(launches)                            # buckets leave in reverse registration order, while backward is still running
    assert abs(sum(row["MB"] for row in comm["bucket_table_last_step"]) - comm["allreduce_payload_bytes"] / 1e6) < 0.1
    assert comm["averaged_gradient_max_err"] < 1e-5 and comm["replicas_identical"] is True


def test_bench_launches_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with NO torch.distributed.run in front (and no WORLD_SIZE) must still run two ranks, never
    print n_gpus != --gpus with rc 0 (the reference runs one replica per listed GPU: train.py:272-277, 404)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "4"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["comm"]["backend"] == "gloo"


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
