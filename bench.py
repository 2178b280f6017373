#!/usr/bin/env python
"""SAUNet training-step benchmark on MI355X (BASELINE.json metric: 2-D slices/s, train fwd+bwd, 256x256).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward + DualLoss + backward + gradient all-reduce (N>1) + fused SGD update of the full
SAUNet (DenseNet-121 encoder, shape stream, dual-attention decoder) on a device-resident synthetic batch
of 32 slices/GPU at 256x256, bf16 storage / fp32 accumulate (BASELINE config 2; configs 3-5 via flags).
Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

# algorithmic work per 256x256 slice (SURVEY.md section 8d / BASELINE.md section 2, conv+convT MACs x2, train = 3x fwd - conv0 dgrad)
TRAIN_GFLOP_PER_SLICE_256 = 216.1
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak
MFMA_F32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="slices per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--graph-dp", action="store_true", help="N>1: capture fwd+bwd (incl. the SyncBN all-reduces) in a hipGraph too; "
                    "default for N>1 is eager launches with the bucket all-reduces overlapped with backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "radam"])
    ap.add_argument("--roofline-only", action="store_true", help="run only the dominant-kernel probe (the command profiles/ was made with)")
    return ap.parse_args()


def kernel_roofline(S, dtype, batch, size):
    """Time the dominant kernel of the step on its own with HIP events on the launch stream.

    Dominant kernel (profiles/r01_f_step_kernel_stats.txt): dense_dgrad_kernel -- the DenseNet conv1 data gradient with the
    fused BatchNorm-backward epilogue (58 launches per step, one per dense layer; csrc/dense_dgrad.hip).  Probe geometry:
    dense block 1, layer 5 (Cin = 192 earlier channels) at B x (size/2)^2 pixels.  Algorithmic bytes per launch:
    g [P,128] read + x [P,Cin] read + dbuf [P,Cin] read and written = (128 + 3*Cin) * itemsize per pixel (+ 128*Cin weights);
    algorithmic FLOPs = 2*P*128*Cin."""
    HF = S.functional
    h = size // 2
    n, k, cin, ctot = batch, 128, 192, 256
    buf = torch.randn(n, ctot, h, h, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)    # concat buffer (x)
    dbuf = torch.randn(n, ctot, h, h, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)   # gradient buffer (y)
    g = torch.randn(n, k, h, h, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(k, cin, 1, 1, device="cuda") * 0.05)
    p = HF.BNParams(cin, "cuda")
    p.buf[0].uniform_(0.5, 1.5); p.buf[1].normal_(0, 0.3); p.buf[2].normal_(0, 0.3); p.buf[3].uniform_(0.5, 1.5)
    sums = torch.zeros(HF.STAT_R, 2, cin, dtype=torch.float64, device="cuda")
    def run():
        HF.conv_dgrad_raw(g, w, (n, cin, h, h), 1, 0, out=dbuf[:, :cin], bn_epi=(buf[:, :cin], p, True, sums, True))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    reps = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    P = n * h * h
    flops = 2.0 * P * k * cin
    nbytes = (P * (k + 3 * cin) + k * cin) * g.element_size()
    tflops = flops / (ms * 1e-3) / 1e12
    gbs = nbytes / (ms * 1e-3) / 1e9
    ai = flops / nbytes
    peak_tf = MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else MFMA_F32_PEAK_TFLOPS
    label = "dense_dgrad 1x1 128->%d +BN-bwd epilogue @%dx%d B%d" % (cin, h, h, n)
    # the bound that applies: min(MFMA peak, AI x HBM peak)
    hbm_bound_tf = ai * HBM_PEAK_GBS / 1e3
    if hbm_bound_tf < peak_tf:
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": None, "kernel": label, "ms": round(ms, 4), "algorithmic_bytes": int(nbytes),
                "tflops": round(tflops, 1), "flop_per_byte": round(ai, 1)}
    return {"bound": "mfma", "achieved": round(tflops, 1), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(tflops / peak_tf, 4),
            "traffic": None, "kernel": label, "ms": round(ms, 4), "algorithmic_bytes": int(nbytes),
            "gbs": round(gbs, 1), "flop_per_byte": round(ai, 1)}


def config_name(args, world):
    """which BASELINE.json config the command line corresponds to"""
    key = (args.size, args.batch, args.dtype)
    if key == (256, 32, "bf16"):
        return "configs[1]" if world == 1 else "configs[2]"
    if key == (256, 64, "bf16"):
        return "configs[3]"
    if key == (512, 8, "f32"):
        return "configs[4]"
    if key == (128, 2, "f32"):
        return "configs[0] geometry"
    return "custom geometry (not a BASELINE config)"


def pmc_traffic(kernel_label):
    """HBM bytes per launch of the probe kernel from the committed rocprofv3 PMC passes (profiles/*_pmc.json), or None."""
    path = os.path.join(ROOT, "profiles", "roofline_pmc.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        return rec.get("traffic_bytes_per_launch") if rec.get("kernel") == kernel_label else None
    except Exception:
        return None


def cpu_baseline(size):
    """The CPU oracle (oracle/saunet_ref.py, the pinned restatement of the reference's PyTorch path) timed on this
    host: B=2 slices, fwd+bwd+SGD, a bounded number of iterations."""
    from oracle import saunet_ref as R, weights as Wt
    # intra-op threads: a few dozen at most -- with one thread per core of a 256-core host the small-channel layers spend their
    # time in thread wake-ups and the same step runs ~100x slower (measured: 470 s/iteration with 256 threads)
    cores = max(1, min(os.cpu_count() or 1, 32))
    torch.set_num_threads(cores)
    spec = R.state_dict_spec()
    sd = Wt.make_state_dict(spec, 0)
    keys = Wt.trainable_keys(spec)
    for k in keys:
        sd[k].requires_grad_(True)
    B = 2
    img, seg, edge = Wt.synthetic_batch(B, size, size)
    canny = R.canny_branch(img)
    opt = torch.optim.SGD([sd[k] for k in keys], lr=5e-4, momentum=0.9)
    times = []
    t_start = time.time()
    for it in range(6):
        t0 = time.time()
        opt.zero_grad()
        loss, *_ = R.segmentation_step(sd, img, seg, edge, True, canny=canny)
        loss.backward(); opt.step()
        times.append(time.time() - t0)
        if time.time() - t_start > 20 and it >= 1:      # bounded sample: ~20 s of CPU work, at least one steady iteration
            break
        if time.time() - t_start > 60:
            break
    steady = sorted(times[1:])[len(times[1:]) // 2] if len(times) > 1 else times[0]
    return {"value": round(B / steady, 3), "unit": "slices/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle restatement (PyTorch CPU fp32), B=2 %dx%d, fwd+bwd+SGD, median of %d iters after 1 warm-up" % (size, size, len(times) - 1)}


def main():
    args = parse()
    import saunet_amd as S
    from saunet_amd import dp, data, optim
    if args.roofline_only:
        torch.cuda.set_device(0)
        r = kernel_roofline(S, torch.bfloat16 if args.dtype == "bf16" else torch.float32, args.batch, args.size)
        r["traffic"] = pmc_traffic(r["kernel"])
        print(json.dumps({"roofline": r}), flush=True)
        return
    rank, local, world = dp.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    S.set_compute_dtype(dtype)
    torch.manual_seed(304)
    net = S.SAUNet(num_classes=4).to(dev)
    sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4).train()
    dp.broadcast_parameters(net)
    opt = optim.create_optimizers(net, args.optimizer, lr=5e-4, momentum=0.9, weight_decay=1e-4)[0]
    use_graph = (not args.no_graph) and (world == 1 or args.graph_dp)
    buckets = dp.GradientBuckets(list(net.parameters()), bucket_mb=32.0, overlap=not use_graph) if world > 1 else None
    img, seg, edge = data.synthetic_batch(args.batch, args.size, args.size, seed=304 + 1000 * rank, device=dev)
    feed = {"image": img, "mask": (seg, edge)}

    def fwd_bwd():
        sm.zero_grad(set_to_none=True)
        loss, _ = sm(feed, 1)
        loss.backward()
        return loss.detach()     # never keep the autograd graph (and its AccumulateGrad nodes) alive across steps

    def tail():
        if buckets is not None:
            buckets.finish()
        opt.step(upload=False)

    opt_ready = False
    def eager_step():
        nonlocal opt_ready
        loss = fwd_bwd()
        opt.upload_hyper()
        tail()
        opt_ready = True
        return loss

    # warm-up (also creates optimiser state and fills allocator pools)
    for _ in range(max(args.warmup, 2)):
        loss = eager_step()
    del loss
    torch.cuda.synchronize()
    mode = "eager" if world == 1 else "eager, bucketed all-reduce overlapped with backward"
    graph = None
    if use_graph:
        try:
            from saunet_amd.graph import GraphedStep
            opt.upload_hyper()

            def captured():
                loss = fwd_bwd()
                if world == 1:
                    tail()
                return loss

            # weight re-packing is recorded INSIDE the graph (functional.PackedWeights.prepack under capture), so every replay trains
            # on the weights its predecessor's optimiser step produced
            graph = GraphedStep(captured, warmup=1)
            mode = "hipgraph(fwd+bwd+opt)" if world == 1 else "hipgraph(fwd+bwd)+eager(allreduce+opt)"
            graph.replay(); torch.cuda.synchronize()
        except Exception as e:  # capture unsupported -> measured eagerly, and said so in the JSON
            graph = None
            mode = "eager (graph capture failed: %s)" % (str(e).splitlines()[0][:80])
            torch.cuda.synchronize()

    def step():
        if graph is None:
            return eager_step()
        loss = graph.replay()
        if world > 1:
            tail()
        return loss

    for _ in range(2):
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    final_loss = float(loss.detach().float())

    if rank == 0:
        ms = dt / args.steps * 1e3
        slices = args.batch * world * args.steps / dt
        scale = (args.size / 256.0) ** 2
        out = {
            "metric": "2D slices/sec (train fwd+bwd+allreduce+SGD) at %dx%d" % (args.size, args.size),
            "value": round(slices, 2), "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (ellipse phantom, z-scored; random-init weights)",
            "config": {"workload": "ACDC %dx%d batch=%d/GPU SAUNet %s, %s" % (args.size, args.size, args.batch, args.dtype, config_name(args, world)),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "mode": mode, "optimizer": args.optimizer,
                       "loss": round(final_loss, 5)},
            "achieved_tflops_algorithmic": round(slices * TRAIN_GFLOP_PER_SLICE_256 * scale / 1e3, 2),
        }
        if not args.no_roofline:
            try:
                out["roofline"] = kernel_roofline(S, dtype, args.batch, args.size)
                out["roofline"]["traffic"] = pmc_traffic(out["roofline"]["kernel"])
            except Exception as e:
                out["roofline"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.size)
            except Exception as e:
                out["cpu_baseline"] = {"error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
