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
# what a register-only mfma_f32_32x32x16_bf16 loop SUSTAINS on this chip under its power cap with activation-like operands (profiles/r06_mfma_ceiling.txt:
# 1810 TFLOP/s on random normal data at 1.79 GHz / 1310 W, 1893 on relu-like x small weights, 2469 on zeros at 2.39 GHz / 877 W); matrix-bound entries
# carry `frac_of_sustained` next to `frac` (which stays priced against the dense peak)
MFMA_BF16_SUSTAINED_TFLOPS = 1850.0
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
    ap.add_argument("--dry-run", action="store_true", help="host-side rehearsal of the N>1 path WITHOUT a GPU (gloo): real SAUNet parameter set, "
                    "synthetic gradients, bucketed all-reduce overlapped with the backward hooks, timing protocol and JSON line; no kernel runs")
    ap.add_argument("--rehearsal-roofline", action="store_true", help="with --share-gpu: keep the live roofline census (a COLLECTIVE step on every rank) in the rehearsal")
    ap.add_argument("--share-gpu", action="store_true", help="REHEARSAL of the N>1 path on a 1-GPU box: the N ranks run the real HIP step on the SAME GPU and "
                    "exchange gradients / SyncBN statistics over gloo (SAUNET_SHARE_GPU=1, SAUNET_DIST_BACKEND=gloo): real kernels, real bucket / hook / "
                    "overlap logic and the bench's own timing protocol; the line is marked `rehearsal` and carries no multi-GPU `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "radam"])
    ap.add_argument("--roofline-only", action="store_true", help="run only the dominant-kernel probe (the command profiles/ was made with)")
    ap.add_argument("--no-launch-mix", action="store_true", help="skip the 58-geometry launch-mix timing of the dominant kernel")
    ap.add_argument("--infer", action="store_true", help="measure the inference forward (hipGraph, BN folded) instead of the training step")
    ap.add_argument("--no-extras", action="store_true", help="skip fp32_companion / other_configs / val_dice (A/B and profiling runs)")
    ap.add_argument("--bucket-mb", type=float, default=32.0, help="N>1: gradient all-reduce bucket size")
    return ap.parse_args()


DENSE_BLOCKS = ((6, 64), (12, 128), (24, 256), (16, 512))     # (layers, first Cin) of DenseNet-121's four blocks; growth 32, bottleneck 128


def _time_dense_dgrad(S, dtype, n, h, cin, ctot, reps, bufs=None):
    """ms per launch of the dominant kernel at one layer geometry (HIP events on the launch stream = torch's current stream)."""
    HF = S.functional
    k = 128
    if bufs is None:
        bufs = (torch.randn(n, ctot, h, h, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last),
                torch.randn(n, ctot, h, h, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last),
                torch.randn(n, k, h, h, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last))
    buf, dbuf, g = bufs
    w = torch.nn.Parameter(torch.randn(k, cin, 1, 1, device="cuda") * 0.05)
    p = HF.BNParams(cin, "cuda")
    p.buf[0].uniform_(0.5, 1.5); p.buf[1].normal_(0, 0.3); p.buf[2].normal_(0, 0.3); p.buf[3].uniform_(0.5, 1.5)
    sums = torch.zeros(HF.STAT_R, 2, cin, dtype=torch.float64, device="cuda")

    def run():
        HF.conv_dgrad_raw(g, w, (n, cin, h, h), 1, 0, out=dbuf[:, :cin], bn_epi=(buf[:, :cin], p, True, sums, True))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    P = n * h * h
    nbytes = (P * (k + 3 * cin) + k * cin) * g.element_size()
    return e0.elapsed_time(e1) / reps, nbytes, 2.0 * P * k * cin, bufs


def kernel_roofline(S, dtype, batch, size, launch_mix=True):
    """Time the dominant kernel of the step on its own with HIP events on the launch stream.

    Dominant kernel (profiles/r02_step_kernel_stats.txt): dense_dgrad_kernel -- the DenseNet conv1 data gradient with the
    fused BatchNorm-backward epilogue (58 launches per step, one per dense layer; csrc/dense_dgrad.hip).  Probe geometry:
    dense block 1, layer 5 (Cin = 192 earlier channels) at B x (size/2)^2 pixels.  Algorithmic bytes per launch:
    g [P,128] read + x [P,Cin] read + dbuf [P,Cin] read and written = (128 + 3*Cin) * itemsize per pixel (+ 128*Cin weights);
    algorithmic FLOPs = 2*P*128*Cin.  `launch_mix`: the same measurement over ALL 58 layer geometries of the step
    (sum of bytes / sum of time) -- the fraction the kernel reaches over its real launch mix, not at its best geometry."""
    h = size // 2
    n, k, cin, ctot = batch, 128, 192, 256
    ms, nbytes, flops, bufs = _time_dense_dgrad(S, dtype, n, h, cin, ctot, 30)
    P = n * h * h
    itemsize = 2 if dtype == torch.bfloat16 else 4
    tflops = flops / (ms * 1e-3) / 1e12
    gbs = nbytes / (ms * 1e-3) / 1e9
    ai = flops / nbytes
    peak_tf = MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else MFMA_F32_PEAK_TFLOPS
    label = "dense_dgrad 1x1 128->%d +BN-bwd epilogue @%dx%d B%d" % (cin, h, h, n)
    # the bound that applies: min(MFMA peak, AI x HBM peak)
    hbm_bound_tf = ai * HBM_PEAK_GBS / 1e3
    if hbm_bound_tf < peak_tf:
        out = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
               "traffic": None, "kernel": label, "ms": round(ms, 4), "algorithmic_bytes": int(nbytes),
               "tflops": round(tflops, 1), "flop_per_byte": round(ai, 1)}
    else:
        out = {"bound": "mfma", "achieved": round(tflops, 1), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(tflops / peak_tf, 4),
               "traffic": None, "kernel": label, "ms": round(ms, 4), "algorithmic_bytes": int(nbytes),
               "gbs": round(gbs, 1), "flop_per_byte": round(ai, 1)}
    del bufs
    if launch_mix:
        tot_ms = tot_bytes = 0.0
        per_block = []
        for b, (layers, c0) in enumerate(DENSE_BLOCKS):
            hb = size // (2 << b)
            bufs, bms, bby = None, 0.0, 0.0
            for l in range(layers):
                m_, by_, _, bufs = _time_dense_dgrad(S, dtype, n, hb, c0 + 32 * l, c0 + 32 * layers, 8, bufs)
                bms += m_; bby += by_
            del bufs
            per_block.append({"block": b + 1, "launches": layers, "ms": round(bms, 4), "GBs": round(bby / (bms * 1e-3) / 1e9, 1)})
            tot_ms += bms; tot_bytes += bby
        mix_gbs = tot_bytes / (tot_ms * 1e-3) / 1e9
        out["launch_mix"] = {"launches": sum(l for l, _ in DENSE_BLOCKS), "ms_per_step": round(tot_ms, 4), "algorithmic_bytes": int(tot_bytes),
                             "achieved": round(mix_gbs, 1), "frac": round(mix_gbs / HBM_PEAK_GBS, 4), "per_block": per_block}
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# Live per-kernel census of one real training step: which kernels dominate is READ from the committed rocprofv3
# summary of this round, and each of them is timed here over its real launch mix -- not one favourable geometry of one kernel.
STATS_FILE = os.path.join("profiles", "r06_f_step_kernel_stats.txt")


def _norm_symbol(name):
    """rocprofv3 row / launch-log name -> 'xyz_kernel<unsigned short, 64, ...>' (namespace, return type and argument list dropped)"""
    n = name.strip().replace("void saunet::", "").replace("saunet::", "")
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    n = n[:cut].strip()
    if "_kernel" not in n and "<" not in n:
        n += "_kernel"
    return n


def _family(sym):
    return sym.split("<")[0]


def read_kernel_ranking(path=None):
    """(symbols, families) of the committed per-step kernel table: [(symbol, us_per_step, calls_per_step)], largest first.  None if absent."""
    path = path or os.path.join(ROOT, STATS_FILE)
    try:
        rows = []
        section = "symbols"
        for line in open(path):
            if line.startswith("# [families]"):
                section = "families"
            elif line.startswith("# [symbols]"):
                section = "symbols"
            if line.startswith("#") or line.startswith("calls") or not line.strip() or section != "symbols":
                continue
            f = line.split(None, 5)
            if len(f) < 6 or not f[0].isdigit():
                continue
            rows.append((_norm_symbol(f[5]), float(f[4]), float(f[0])))
        steps = None
        for line in open(path):
            if "/step = total /" in line:
                steps = float(line.split("/step = total /")[1].split()[0])
        rows = [(sym, us, calls / (steps or 1.0)) for sym, us, calls in rows if "rocclr" not in sym and "at::native" not in sym]
        rows.sort(key=lambda r: -r[1])
        fam = {}
        for sym, us, calls in rows:
            e = fam.setdefault(_family(sym), [0.0, 0.0, 0])
            e[0] += us; e[1] += calls; e[2] += 1
        fams = sorted(((k, v[0], v[1], v[2]) for k, v in fam.items()), key=lambda r: -r[1])
        return rows, fams
    except Exception:
        return None


_SECONDARY = ("wgrad_reduce", "bn_bwd_correct_ab", "conv3x3_mm_finish", "bn_prologue_finalize")


def _main_symbol(log):
    """a library call may launch a helper next to its main kernel ('conv_tile_wgrad_grouped_kernel<..>+wgrad_reduce_multi'): the call's
    HIP-event time is attributed to the main kernel and the helpers are listed under `includes`"""
    parts = [_norm_symbol(x) for x in log.split("+") if x]
    main = [x for x in parts if not _family(x).replace("_kernel", "").startswith(_SECONDARY)]
    return (main[0] if main else parts[0]), [x for x in parts if x != (main[0] if main else parts[0])]


def _call_work(name, args, esz):
    """(algorithmic bytes, FLOPs) of one library call, SURVEY 8(d) accounting: every operand read once, every result written once (+ the
    read of the accumulate target), weights once; None for entries that are not priced."""
    def obj(a):
        return getattr(a, "_obj", a)
    if name in ("saunet_conv2d_forward", "saunet_conv2d_forward_ex", "saunet_conv2d_forward_bnpro"):
        d = obj(args[0])
        taps = 16 if d.transposed else d.KH * d.KW
        pin, pout = d.N * d.H * d.W, d.N * d.Ho * d.Wo
        by = (pin * d.Cin + pout * d.Cout + taps * d.Cin * d.Cout) * esz
        if name == "saunet_conv2d_forward_ex" and args[9] is not None:
            e = obj(args[9])
            if e.bn_x:
                by += pout * d.Cout * esz                       # the tensor the fused BatchNorm backward normalised
            if e.accumulate:
                by += pout * d.Cout * esz                       # y is read before it is written
        mac_px = pin if d.transposed else pout                  # conv-transpose: 16 taps per INPUT pixel
        return by, 2.0 * mac_px * taps * d.Cin * d.Cout
    if name == "saunet_dense_layer_backward_conv2":
        l = obj(args[0]); P = l.N * l.H * l.W
        chunk = 3 * 32 if l.dz2 else 32                          # gradient chunk (+ its activations and the corrected copy)
        return (P * (chunk + 128 + 128) + 9 * 32 * 128) * esz, 2.0 * P * 288 * 128
    if name == "saunet_dense_layer_backward_conv1":
        l = obj(args[0]); P = l.N * l.H * l.W
        cw = l.Cin - l.c_begin                                   # a layer pair's first launch touches the top chunk only
        return (P * (3 * 128 + 3 * cw) + 128 * cw) * esz, 2.0 * P * 128 * cw
    if name == "saunet_dense_layer_backward_conv1_pair":
        # G, z1 of the lower layer read, its dz1 written, the upper layer's dz1 read; x read, dbuf read + written over the lower layer's channels;
        # both layers' weight rows of those channels
        lo = obj(args[1]); P = lo.N * lo.H * lo.W
        return (P * (4 * 128 + 3 * lo.Cin) + 2 * 128 * lo.Cin) * esz, 2.0 * P * 128 * lo.Cin * 2
    if name == "saunet_bn_backward_apply":
        P, Cc = args[24], args[25]
        return P * Cc * esz * (3 + (1 if args[17] else 0) + (1 if args[5] else 0) + (1 if args[20] else 0)), 0.0
    if name == "saunet_bn_backward_reduce":
        return args[15] * args[16] * esz * (2 + (1 if args[5] else 0)), 0.0
    if name == "saunet_affine_act":
        return args[10] * args[11] * esz * (2 + (1 if args[5] else 0)), 0.0
    if name == "saunet_affine_act_pool":
        return args[8] * args[9] * esz * 2, 0.0
    if name in ("saunet_conv2d_wgrad", "saunet_conv2d_wgrad_deferred"):
        d = obj(args[0])
        taps = 16 if d.transposed else d.KH * d.KW
        pin, pout = d.N * d.H * d.W, d.N * d.Ho * d.Wo
        return (pin * d.Cin + pout * d.Cout) * esz + taps * d.Cin * d.Cout * 4, 2.0 * (pin if d.transposed else pout) * taps * d.Cin * d.Cout
    if name == "saunet_conv2d_wgrad_grouped":
        g = obj(args[0]); P = g.N * g.H * g.W
        by = fl = 0.0
        for i in range(g.count):
            it = g.item[i]
            by += P * (it.Cin + it.Cout) * esz + g.KH * g.KH * it.Cin * it.Cout * 4
            fl += 2.0 * P * g.KH * g.KH * it.Cin * it.Cout
        return by, fl
    if name == "saunet_bn_backward_correct_ab":
        return args[14] * args[15] * esz * 3, 0.0
    if name == "saunet_bn_backward_coeff_correct":
        return args[21] * (args[18] - args[17]) * esz * 3, 0.0
    return None


def live_kernel_census(S, step_fn, dtype):
    """ONE eager training step with HIP events (recorded on the launch stream = torch's current stream) around every call into
    libsaunet_hip.so.  The GPU is held back by a spin kernel while the host queues the step, so the launches execute back to back as they do
    inside the captured graph and an event pair brackets a kernel's execution + its launch boundary, not host gaps.  The kernel(s) behind
    a call come from the library's own launch log (saunet_launch_log), so the attribution follows the real dispatch.
    -> {symbol: {"ms", "launches", "algorithmic_bytes", "flops", "priced_ms", "includes"}} for this step."""
    L = S.lib
    handle = L.load()
    orig = L.call
    esz = 2 if dtype == torch.bfloat16 else 4
    recs = []

    def traced(name, *args):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        handle.saunet_launch_log()                         # drop anything logged outside a traced call
        e0.record()
        orig(name, *args)
        e1.record()
        log = handle.saunet_launch_log()
        recs.append((name, (log or b"").decode(), e0, e1, _call_work(name, args, esz)))

    step_fn(); torch.cuda.synchronize()                    # allocator pools / packings warm
    # calibrate the spin kernel, then hold the GPU for ~1.5x the host time of an eager step
    t0 = time.perf_counter(); step_fn(); host_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record(); torch.cuda._sleep(1000000); c1.record(); torch.cuda.synchronize()
    per_ms = 1000000 / max(c0.elapsed_time(c1), 1e-3)
    L.call = traced
    nulls = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
    try:
        torch.cuda._sleep(int(per_ms * min(1.5 * host_ms + 10.0, 400.0)))
        step_fn()
        for a_, b_ in nulls:                               # empty event pairs in the same queue: what the bracketing itself costs
            a_.record(); b_.record()
    finally:
        L.call = orig
    torch.cuda.synchronize()
    live_kernel_census.event_pair_overhead_us = round(sorted(a_.elapsed_time(b_) for a_, b_ in nulls)[len(nulls) // 2] * 1e3, 2)
    out = {}
    for name, log, e0, e1, work in recs:
        if not log:
            continue
        sym, inc = _main_symbol(log)
        ms = e0.elapsed_time(e1)
        r = out.setdefault(sym, {"ms": 0.0, "launches": 0, "algorithmic_bytes": 0.0, "flops": 0.0, "priced_ms": 0.0, "roof_ms": 0.0, "includes": set()})
        r["ms"] += ms; r["launches"] += 1; r["includes"].update(inc)
        if work is not None:
            r["algorithmic_bytes"] += work[0]; r["flops"] += work[1]; r["priced_ms"] += ms
            # the launch's own roofline time: whichever of the two bounds is the longer for THIS geometry (a family mixes HBM-bound and
            # matrix-bound launches; its aggregate flop/byte would price all of them against one peak)
            peak_tf = MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else MFMA_F32_PEAK_TFLOPS
            t_hbm, t_mfma = work[0] / (HBM_PEAK_GBS * 1e9), work[1] / (peak_tf * 1e12)
            r["roof_ms"] += max(t_hbm, t_mfma) * 1e3
            bb = r.setdefault("by_bound", {}).setdefault("hbm" if t_hbm >= t_mfma else "mfma", [0.0, 0, 0.0, 0.0])
            bb[0] += ms; bb[1] += 1; bb[2] += work[0]; bb[3] += work[1]
    return out


def _roofline_entry(label, ms, launches, nbytes, flops, dtype, extra=None):
    peak_tf = MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else MFMA_F32_PEAK_TFLOPS
    e = {"kernel": label, "ms": round(ms, 4), "launches": int(launches), "algorithmic_bytes": int(nbytes), "traffic": None}
    if ms <= 0 or nbytes <= 0:
        e.update(bound=None, achieved=None, peak=None, unit=None, frac=None)
        return e
    gbs = nbytes / (ms * 1e-3) / 1e9
    tfl = flops / (ms * 1e-3) / 1e12
    ai = flops / nbytes
    if ai * HBM_PEAK_GBS / 1e3 < peak_tf:
        e.update(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), tflops=round(tfl, 1))
    else:
        e.update(bound="mfma", achieved=round(tfl, 1), peak=peak_tf, unit="TFLOP/s", frac=round(tfl / peak_tf, 4), gbs=round(gbs, 1))
        if dtype == torch.bfloat16:
            e["frac_of_sustained"] = round(tfl / MFMA_BF16_SUSTAINED_TFLOPS, 4)
    e["flop_per_byte"] = round(ai, 1)
    if extra:
        e.update(extra)
    return e


def census_roofline(S, step_fn, dtype, args):
    """`roofline` of the bench line.  The kernel ranking is read from the committed rocprofv3 summary of this round (profiles/r06_*): its
    top three SYMBOLS are reported one by one, its largest FAMILY (all template instances of one kernel) gives the headline `frac` -- each
    timed live over its real launch mix of one step (live_kernel_census), algorithmic bytes per SURVEY 8(d), HBM traffic from the committed
    whole-step PMC passes (profiles/step_pmc.json `per_kernel`, FETCH_SIZE x 2 + WRITE_SIZE per the guide's gfx950 correction)."""
    rank = read_kernel_ranking()
    census = live_kernel_census(S, step_fn, dtype)
    pmc = {}
    try:
        with open(os.path.join(ROOT, "profiles", "step_pmc.json")) as f:
            rec = json.load(f)
        if rec.get("config") == [args.size, args.batch, args.dtype]:
            pmc = {_norm_symbol(k): v for k, v in rec.get("per_kernel", {}).items()}
    except Exception:
        pass
    if rank is None:
        # no committed table (a fresh checkout mid-round): rank by the live census itself and say so
        syms = sorted(((k, v["ms"] * 1e3, v["launches"]) for k, v in census.items()), key=lambda r: -r[1])
        fam = {}
        for sym, us, calls in syms:
            e = fam.setdefault(_family(sym), [0.0, 0.0, 0]); e[0] += us; e[1] += calls; e[2] += 1
        fams = sorted(((k, v[0], v[1], v[2]) for k, v in fam.items()), key=lambda r: -r[1])
        source = "live census (no committed %s)" % STATS_FILE
    else:
        syms, fams = rank
        source = STATS_FILE

    def entry_for(symbols, label, rocprof_us):
        ms = sum(census[s_]["ms"] for s_ in symbols if s_ in census)
        pms = sum(census[s_]["priced_ms"] for s_ in symbols if s_ in census)
        n = sum(census[s_]["launches"] for s_ in symbols if s_ in census)
        by = sum(census[s_]["algorithmic_bytes"] for s_ in symbols if s_ in census)
        fl = sum(census[s_]["flops"] for s_ in symbols if s_ in census)
        inc = sorted(set().union(*[census[s_]["includes"] for s_ in symbols if s_ in census])) if any(s_ in census for s_ in symbols) else []
        # bytes are known for the priced calls only: the rate is taken over THEIR time (ms stays the whole symbol's time)
        e = _roofline_entry(label, pms if pms > 0 else ms, n, by, fl, dtype)
        e["ms"] = round(ms, 4)
        roof = sum(census[s_]["roof_ms"] for s_ in symbols if s_ in census)
        # sum over the launches of max(bytes / HBM peak, FLOPs / MFMA peak) over their measured time: the per-launch roofline fraction
        e["frac_per_launch_bound"] = round(roof / pms, 4) if pms > 0 else None
        e["priced_fraction_of_ms"] = round(pms / ms, 3) if ms > 0 else None
        e["rocprof_ms_per_step"] = round(rocprof_us / 1e3, 4) if rocprof_us is not None else None
        # the same launches split by THEIR OWN bound: a family that mixes HBM-bound and matrix-bound geometries is
        # priced twice, each part against the peak that bounds it
        split = {}
        for s_ in symbols:
            for bnd, (bms, bn, bby, bfl) in census.get(s_, {}).get("by_bound", {}).items():
                a = split.setdefault(bnd, [0.0, 0, 0.0, 0.0]); a[0] += bms; a[1] += bn; a[2] += bby; a[3] += bfl
        if split:
            peak_tf_ = MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else MFMA_F32_PEAK_TFLOPS
            e["by_bound"] = {}
            for bnd, (bms, bn, bby, bfl) in split.items():
                if bms <= 0:
                    continue
                if bnd == "hbm":
                    e["by_bound"]["hbm"] = {"launches": bn, "ms": round(bms, 4), "achieved": round(bby / (bms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                            "unit": "GB/s", "frac": round(bby / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                else:
                    e["by_bound"]["mfma"] = {"launches": bn, "ms": round(bms, 4), "achieved": round(bfl / (bms * 1e-3) / 1e12, 1), "peak": peak_tf_,
                                             "unit": "TFLOP/s", "frac": round(bfl / (bms * 1e-3) / 1e12 / peak_tf_, 4)}
                    if dtype == torch.bfloat16:
                        e["by_bound"]["mfma"]["frac_of_sustained"] = round(bfl / (bms * 1e-3) / 1e12 / MFMA_BF16_SUSTAINED_TFLOPS, 4)
        tr = [pmc[s_]["traffic_bytes_per_step"] for s_ in symbols if s_ in pmc]
        e["traffic"] = int(sum(tr)) if tr else None
        if inc:
            e["includes"] = inc
        return e
    kernels = [entry_for([sym], sym, us) for sym, us, _ in syms[:3]]
    top_family = fams[0][0]
    members = [sym for sym, _, _ in syms if _family(sym) == top_family]
    if rank is None:
        members = [k for k in census if _family(k) == top_family]
    out = entry_for(members, "%s (%d template instances, whole launch mix of the step)" % (top_family, len(members)), fams[0][1])
    out["ranking_source"] = source
    out["timing"] = ("HIP events on the launch stream around every library call of ONE eager step queued behind a spin kernel (launches run back to back); "
                     "an event pair brackets the kernel AND its launch boundary (1.5-3 us per launch on this chip), so `ms` sits above the "
                     "rocprofv3 kernel-only duration by about that much per launch")
    out["empty_event_pair_us"] = getattr(live_kernel_census, "event_pair_overhead_us", None)
    out["kernels"] = kernels
    out["families"] = [{"family": k, "rocprof_ms_per_step": round(us / 1e3, 4), "launches": round(c, 1)} for k, us, c, _ in fams[:6]]
    return out


def weakest_family_roofline(S, dtype, batch, size):
    """Second roofline entry: the WEAKEST large MFMA-priced kernel family of the step next to the healthiest one.  Three candidates
    are timed live with HIP events at the step's geometry and the one with the lowest fraction of the dense matrix-core peak is reported (all
    three are listed under `candidates`):
      * dec3.c3x3rb weight gradient (512 -> 128 at (size/4)^2; conv3x3_wgrad_mm_kernel + its partial-gradient reduce) -- round 3's weakest,
      * res1 forward (64 -> 64 at size^2; conv3x3_res_fwd_kernel: LDS-read bound, 288 FLOP/B sits AT the ridge, so the HBM fraction is given too),
      * dec2.c3x3rb forward (256 -> 64 at (size/2)^2; conv3x3_mm_kernel with the 64-wide tile).
    FLOPs = 2 * P * 9 * Cin * Cout.  `same_layer_forward` keeps the LDS-DMA forward of dec3 for comparison."""
    HF = S.functional
    peak = MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else MFMA_F32_PEAK_TFLOPS
    esz = 2 if dtype == torch.bfloat16 else 4

    def timed(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def probe(cin, cout, h, wgrad):
        x = torch.randn(batch, cin, h, h, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
        w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device="cuda") * 0.03)
        fl = 2.0 * batch * h * h * 9 * cin * cout
        if wgrad:
            dy = torch.randn(batch, cout, h, h, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
            ms = timed(lambda: (HF.GRADS.reset(), HF.conv_wgrad_raw(x, dy, w, 1, 1)))
            HF.GRADS.reset()
        else:
            out = HF.new_act(batch, cout, h, h, dtype, "cuda")
            st = torch.zeros(HF.STAT_R, 2, cout, dtype=torch.float64, device="cuda")
            ms = timed(lambda: HF.conv_forward_raw(x, w, None, 1, 1, out=out, stats=st))
        tf = fl / (ms * 1e-3) / 1e12
        byts = batch * h * h * (cin + cout) * esz
        return {"ms": round(ms, 4), "achieved": round(tf, 1), "frac": round(tf / peak, 4), "flops": fl,
                "frac_of_sustained": round(tf / MFMA_BF16_SUSTAINED_TFLOPS, 4) if dtype == torch.bfloat16 else None,
                "hbm_frac": round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    cands = [
        dict(probe(512, 128, size // 4, True), kernel="conv3x3_wgrad_mm 3x3 512->128 @%dx%d B%d (dec3.c3x3rb weight gradient, incl. its partial-gradient reduce)" % (size // 4, size // 4, batch)),
        dict(probe(64, 64, size, False), kernel="conv3x3_res_fwd 3x3 64->64 @%dx%d B%d (res1 forward)" % (size, size, batch)),
        dict(probe(256, 64, size // 2, False), kernel="conv3x3_mm<64> 3x3 256->64 @%dx%d B%d (dec2.c3x3rb forward)" % (size // 2, size // 2, batch)),
    ]
    fwd3 = probe(512, 128, size // 4, False)
    worst = min(cands, key=lambda c: c["frac"])
    return {"bound": "mfma", "achieved": worst["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": worst["frac"], "ms": worst["ms"], "kernel": worst["kernel"],
            "flops": worst["flops"], "hbm_frac": worst["hbm_frac"], "candidates": cands,
            "same_layer_forward": {"kernel": "conv3x3_mm_kernel 512->128 (LDS-DMA staged, dec3)", "ms": fwd3["ms"], "achieved": fwd3["achieved"], "frac": fwd3["frac"]}}


def step_roofline(args, ms_per_step):
    """Whole-step roofline against SURVEY 8(d)'s ideal-fusion algorithmic bytes (1.086 GB per 256x256 slice in bf16, 2.171 GB in float32:
    every conv reads its input once and writes its output once, forward + backward) and the measured HBM traffic of the committed
    whole-step PMC passes (profiles/step_pmc.json, produced by scripts/collect_step_pmc.sh for the default configuration)."""
    per_slice = (1.086e9 if args.dtype == "bf16" else 2.171e9) * (args.size / 256.0) ** 2
    alg = per_slice * args.batch
    gbs = alg / (ms_per_step * 1e-3) / 1e9
    out = {"algorithmic_bytes": int(alg), "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic_bytes": None, "traffic_ratio": None,
           "mfma_frac": round(args.batch * TRAIN_GFLOP_PER_SLICE_256 * (args.size / 256.0) ** 2 / ms_per_step / (MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS), 4)}
    try:
        with open(os.path.join(ROOT, "profiles", "step_pmc.json")) as f:
            rec = json.load(f)
        if rec.get("config") == [args.size, args.batch, args.dtype]:
            out["traffic_bytes"] = int(rec["traffic_bytes_per_step"])
            out["traffic_ratio"] = round(rec["traffic_bytes_per_step"] / alg, 3)
            out["traffic_source"] = rec.get("source")
    except Exception:
        pass
    return out


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
    """The CPU oracle (oracle/saunet_ref.py, the pinned restatement of the reference's PyTorch path) timed on this host with
    SURVEY 8(d)'s protocol: fwd + bwd + SGD, 3 warm-up + 10 timed iterations at B=2 and (bounded: 1 + 4) at B=8, median s/iter."""
    from oracle import saunet_ref as R, weights as Wt
    # intra-op threads: a few dozen at most -- with one thread per core of a 256-core host the small-channel layers spend their
    # time in thread wake-ups and the same step runs ~100x slower (measured: 470 s/iteration with 256 threads)
    ncpu = os.cpu_count() or 1
    phys = physical_cores()
    spec = R.state_dict_spec()
    keys = Wt.trainable_keys(spec)
    # thread-count sweep (SURVEY 8d asks for the host's physical cores; more threads than the layers can feed only add wake-up latency):
    # 1 warm-up + 2 timed B=2 iterations per setting, the protocol then runs at the fastest
    sweep = {}
    for t in [t for t in (8, 16, 32, 64, 128) if t <= phys] or [max(1, min(phys, 8))]:
        torch.set_num_threads(t)
        sd = Wt.make_state_dict(spec, 0)
        for k in keys:
            sd[k].requires_grad_(True)
        img, seg, edge = Wt.synthetic_batch(2, size, size)
        canny = R.canny_branch(img)
        ts = []
        for it in range(3):
            t0 = time.time()
            loss, *_ = R.segmentation_step(sd, img, seg, edge, True, canny=canny)
            loss.backward()
            for k in keys:
                sd[k].grad = None
            ts.append(time.time() - t0)
        sweep[t] = round(min(ts[1:]), 4)
        if sweep[t] > 4.0 and len(sweep) > 1:      # far off the optimum already: stop burning the budget (the all-cores point is taken below)
            break
    if phys not in sweep and phys <= 256:
        # SURVEY 8(d) asks for ALL physical cores: that point is always on the line next to the fastest setting (one warm-up + one timed
        # forward + backward at B = 2: with one thread per core the small-channel layers spend their time in thread wake-ups)
        torch.set_num_threads(phys)
        sd = Wt.make_state_dict(spec, 0)
        for k in keys:
            sd[k].requires_grad_(True)
        img, seg, edge = Wt.synthetic_batch(2, size, size)
        canny = R.canny_branch(img)
        ts = []
        for it in range(2):
            t0 = time.time()
            loss, *_ = R.segmentation_step(sd, img, seg, edge, True, canny=canny)
            loss.backward()
            for k in keys:
                sd[k].grad = None
            ts.append(time.time() - t0)
            if ts[-1] > 60.0:
                break
        sweep[phys] = round(ts[-1], 4)
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    runs = []
    for B, warm, timed, budget in ((2, 3, 10, 40.0), (8, 1, 4, 40.0)):
        sd = Wt.make_state_dict(spec, 0)
        for k in keys:
            sd[k].requires_grad_(True)
        img, seg, edge = Wt.synthetic_batch(B, size, size)
        canny = R.canny_branch(img)
        opt = torch.optim.SGD([sd[k] for k in keys], lr=5e-4, momentum=0.9)
        times, t_start = [], time.time()
        for it in range(warm + timed):
            t0 = time.time()
            opt.zero_grad()
            loss, *_ = R.segmentation_step(sd, img, seg, edge, True, canny=canny)
            loss.backward(); opt.step()
            times.append(time.time() - t0)
            if time.time() - t_start > budget and it >= warm:      # bounded sample
                break
        steady = sorted(times[warm:]) if len(times) > warm else times[-1:]
        med = steady[len(steady) // 2]
        runs.append({"B": B, "s_per_iter": round(med, 4), "slices_per_s": round(B / med, 3), "warmup": min(warm, len(times) - len(steady)), "timed": len(steady)})
    best = max(runs, key=lambda r: r["slices_per_s"])
    return {"value": best["slices_per_s"], "unit": "slices/s", "cores": torch.get_num_threads(), "host_cpus": ncpu, "physical_cores": phys, "kind": "port",
            "all_physical_cores": ({"threads": phys, "s_per_fwd_bwd_b2": sweep[phys], "slices_per_s": round(2.0 / sweep[phys], 3)} if phys in sweep else None),
            "runs": runs, "thread_sweep_s_per_fwd_bwd_b2": {str(k): v for k, v in sweep.items()},
            "sample": "oracle restatement (PyTorch CPU fp32) %dx%d, fwd+bwd+SGD, median s/iter: B=2 3 warm-up + 10 timed, B=8 1 + 4 (bounded); "
                      "%d intra-op threads (fastest of the sweep) on a host with %d physical cores / %d logical CPUs" % (size, size, torch.get_num_threads(), phys, ncpu)}


def physical_cores():
    """physical cores of this host (unique (physical id, core id) pairs of /proc/cpuinfo; logical CPUs when that is not available)"""
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        return len(pairs) or (os.cpu_count() or 1)
    except Exception:
        return os.cpu_count() or 1


def quick_train_bench(S, dev, dtype, batch, size, steps=5, warmup=2, seed=304):
    """one more configuration measured live: build the net, warm up eagerly, capture fwd + bwd + fused SGD as a hipGraph, time `steps` replays"""
    from saunet_amd import data, optim
    from saunet_amd.graph import GraphedStep
    S.set_compute_dtype(dtype)
    torch.manual_seed(seed)
    net = S.SAUNet(num_classes=4).to(dev)
    sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4).train()
    opt = optim.create_optimizers(net, "sgd", lr=5e-4, momentum=0.9, weight_decay=1e-4)[0]
    img, seg, edge = data.synthetic_batch(batch, size, size, seed=seed, device=dev)
    feed = {"image": img, "mask": (seg, edge)}

    def fn():
        sm.zero_grad(set_to_none=True)
        loss, _ = sm(feed, 1)
        loss.backward()
        opt.step(upload=False)
        return loss.detach()
    for _ in range(warmup):
        opt.upload_hyper(); fn()
    torch.cuda.synchronize()
    g = GraphedStep(fn, warmup=1, optimizers=[opt])
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"value": round(batch * steps / dt, 2), "unit": "slices/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "loss": round(float(loss.float()), 5)}
    del g, net, sm, opt, feed, img, seg, edge
    S.functional.notify_params_changed(); S.functional.PACKS.clear()
    torch.cuda.empty_cache()
    return out


def extras(args, S, dev):
    """Driver-visible breadth measured live in the default run: the float32 companion of the headline line (the
    precision at which the 1e-3 / Dice-1e-4 parity with the reference holds), the other BASELINE configurations a single GPU can run, and the
    trained-weights validation Dice of float32 vs bf16 storage (saunet_amd.dice)."""
    out = {}
    bf16, f32 = torch.bfloat16, torch.float32
    try:
        c = quick_train_bench(S, dev, f32, args.batch, args.size)
        c["workload"] = "ACDC %dx%d batch=%d/GPU SAUNet f32 (same step as the headline line, float32 storage and MFMA)" % (args.size, args.size, args.batch)
        out["fp32_companion"] = c
    except Exception as e:
        out["fp32_companion"] = {"error": str(e)[:200]}
    oc = {}
    try:
        ia = argparse.Namespace(**vars(args)); ia.steps, ia.warmup, ia.dtype = 10, 2, "bf16"
        S.set_compute_dtype(bf16)
        r = infer_bench(ia, S, dev, bf16)
        oc["inference_bf16_b%d_%d" % (args.batch, args.size)] = {"value": r["value"], "unit": "slices/s", "ms_per_step": r["ms_per_step"], "steps": r["steps"]}
        S.functional.notify_params_changed(); S.functional.PACKS.clear(); torch.cuda.empty_cache()
    except Exception as e:
        oc["inference"] = {"error": str(e)[:200]}
    for key, dt_, b_, s_ in (("configs[3] 256x256 batch=64 bf16", bf16, 64, 256), ("configs[4] 512x512 batch=8 f32", f32, 8, 512)):
        try:
            oc[key] = quick_train_bench(S, dev, dt_, b_, s_)
        except Exception as e:
            oc[key] = {"error": str(e)[:200]}
    out["other_configs"] = oc
    try:
        from saunet_amd import dice
        # three of the five seeds of tests/test_hip_dice.py (the full table: profiles/r04_dice.json); the reference arm is DATA from the committed
        # fixture tests/golden/dice_ref.npz (the real reference trained on the CPU from the same weights / batches, oracle/make_golden_dice.py)
        ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "dice_ref.npz")
        d = dice.paired_study(seeds=(304, 305, 306, 307, 308), ref_npz=ref if os.path.exists(ref) else None)      # all five seeds of the reference fixture: CI half-widths 0.012-0.018 (three seeds: 0.03-0.05)
        out["val_dice"] = {"protocol": d["protocol"], "seeds": d["seeds"],
                           "rows": [{k: r[k] for k in ("seed", "f32", "bf16", "ref") if k in r} for r in d["rows"]],
                           "mean_f32": d["mean_dice_f32"], "mean_bf16": d["mean_dice_bf16"], "mean_ref": d.get("mean_dice_ref"),
                           "paired_mean_difference_and_ci95": d["pairs"],
                           "first_losses_f32_vs_ref": [[r["loss_first_f32"][:3], r.get("loss_first_ref", [])[:3]] for r in d["rows"]]}
    except Exception as e:
        out["val_dice"] = {"error": str(e)[:200]}
    return out


def infer_bench(args, S, dev, dtype):
    """--infer: eval-mode forward (SegmentationModule test branch semantics: logits -> softmax/argmax on the device) captured in a hipGraph."""
    from saunet_amd import data
    from saunet_amd.graph import GraphedStep
    torch.manual_seed(304)
    net = S.SAUNet(num_classes=4).to(dev).eval()
    img, seg, edge = data.synthetic_batch(args.batch, args.size, args.size, seed=304, device=dev)

    def fwd():
        with torch.no_grad():
            logits, edge_out = net(img)
            return S.functional.softmax_argmax(logits)
    for _ in range(max(args.warmup, 2)):
        fwd()
    torch.cuda.synchronize()
    mode = "hipgraph(forward+softmax/argmax)"
    try:
        g = GraphedStep(fwd, warmup=1, changes_params=False)
        run = g.replay
    except Exception as e:
        run, mode = fwd, "eager (graph capture failed: %s)" % (str(e).splitlines()[0][:80])
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / args.steps * 1e3
    scale = (args.size / 256.0) ** 2
    return {"metric": "2D slices/sec (inference forward) at %dx%d" % (args.size, args.size), "value": round(args.batch * args.steps / dt, 2),
            "unit": "slices/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (ellipse phantom, z-scored; random-init weights)",
            "config": {"workload": "ACDC %dx%d batch=%d SAUNet %s inference (BN folded, softmax+argmax on device)" % (args.size, args.size, args.batch, args.dtype),
                       "global_batch": args.batch, "parallelism": "dp1", "mode": mode},
            "achieved_tflops_algorithmic": round(args.batch * args.steps / dt * 72.15 * scale / 1e3, 2)}


def dry_run(args, S, dp):
    """`--dry-run`: everything of the N-rank path that does not need a GPU, end to end over gloo -- environment bootstrap, parameter broadcast,
    gradient buckets in reverse registration order with their post-accumulate hooks, overlapped all-reduce + averaged write-back, the timing
    protocol (barrier, K steps, max over ranks) and rank 0's JSON line with the `comm` record.  The step itself is a stand-in (a gradient of
    the right shape for every SAUNet parameter); the CPU suite runs this with two ranks (tests/test_bench_dry.py)."""
    rank, local, world = dp.init_from_env(backend="gloo")
    if world != args.gpus:
        raise SystemExit("--gpus %d but %d rank(s) were launched" % (args.gpus, world))
    torch.manual_seed(304 + rank)                     # different replicas before the broadcast
    net = S.SAUNet(num_classes=4)
    dp.broadcast_parameters(net)
    params = [p for p in net.parameters() if p.requires_grad]
    buckets = dp.GradientBuckets(params, bucket_mb=args.bucket_mb, overlap=True) if world > 1 else None
    if buckets is not None:
        buckets.time_finish = True
    lr = 5e-4

    def step(it):
        for p in params:
            p.grad = None
        loss = sum((p * float(rank + 1 + it)).sum() for p in params)
        loss.backward()                               # hooks launch each bucket's all-reduce as soon as it is complete
        if buckets is not None:
            buckets.finish()
        with torch.no_grad():
            for p in params:
                p.add_(p.grad, alpha=-lr)
        return loss.detach()
    for it in range(max(args.warmup, 1)):
        step(it)
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        loss = step(it)
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    # every rank must hold the same averaged gradient and the same parameters afterwards
    want = sum(float(r + 1 + args.steps - 1) for r in range(world)) / world
    gerr = max(float((p.grad - want).abs().max()) for p in params[:8] + params[-8:])
    chk = torch.tensor([float(params[0].detach().double().sum()), float(params[-1].detach().double().sum())], dtype=torch.float64)
    lo, hi = chk.clone(), chk.clone()
    if world > 1:
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN); torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        payload = sum(p.numel() for p in params) * 4
        print(json.dumps({
            "metric": "DRY RUN of the N-rank host path (no GPU work): 2D slices/sec (train fwd+bwd+allreduce+SGD) at %dx%d" % (args.size, args.size),
            "value": None, "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "none (stand-in gradients)", "dry_run": True,
            "config": {"workload": "dry run: SAUNet parameter set, stand-in step", "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "mode": "eager, bucketed all-reduce overlapped with backward hooks"},
            "comm": {"backend": torch.distributed.get_backend() if world > 1 else None, "rccl_ranks": 0,
                     "buckets": len(buckets.buckets) if buckets is not None else 0, "bucket_mb": args.bucket_mb, "allreduce_payload_bytes": payload,
                     "bucket_table_last_step": buckets.bucket_table() if buckets is not None else [],
                     "averaged_gradient_max_err": gerr, "replicas_identical": bool(torch.equal(lo, hi))}}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` WITHOUT a launcher (N > 1, no WORLD_SIZE in the environment): re-execute this command line as N ranks through
    torch.distributed.run on 127.0.0.1 -- one process per GPU, as /root/reference/train.py:272-277,404 intends one replica per device -- and
    exit with its status.  Without this the run would silently measure ONE GPU and print n_gpus = 1."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    if args.share_gpu:
        env.update(SAUNET_SHARE_GPU="1", SAUNET_DIST_BACKEND="gloo")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if env_world == 0 and args.gpus > 1 and not (args.roofline_only or args.infer):
        self_launch(args)
    if env_world not in (0, args.gpus) and not (args.roofline_only or args.infer):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch exactly one rank per requested GPU" % (args.gpus, env_world))
    import saunet_amd as S
    from saunet_amd import dp, data, optim
    if args.roofline_only:
        torch.cuda.set_device(0)
        r = kernel_roofline(S, torch.bfloat16 if args.dtype == "bf16" else torch.float32, args.batch, args.size, launch_mix=not args.no_launch_mix)
        r["traffic"] = pmc_traffic(r["kernel"])
        print(json.dumps({"roofline": r}), flush=True)
        return
    if args.infer:
        torch.cuda.set_device(0)
        dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        S.set_compute_dtype(dtype)
        print(json.dumps(infer_bench(args, S, torch.device("cuda", 0), dtype)), flush=True)
        return
    if args.dry_run:
        return dry_run(args, S, dp)
    if args.share_gpu:
        os.environ.update(SAUNET_SHARE_GPU="1", SAUNET_DIST_BACKEND="gloo")
    rank, local, world = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but %d rank(s) were launched" % (args.gpus, world))
    if world > 1 and ((torch.distributed.get_backend() != "nccl" and not os.environ.get("SAUNET_DIST_BACKEND")) or torch.distributed.get_world_size() != args.gpus):
        raise SystemExit("expected an RCCL ('nccl') group of %d ranks, got backend %s with %d" % (args.gpus, torch.distributed.get_backend(), torch.distributed.get_world_size()))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    S.set_compute_dtype(dtype)
    torch.manual_seed(304)
    net = S.SAUNet(num_classes=4).to(dev)
    sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4).train()
    dp.broadcast_parameters(net)
    opt = optim.create_optimizers(net, args.optimizer, lr=5e-4, momentum=0.9, weight_decay=1e-4)[0]
    use_graph = (not args.no_graph) and world == 1      # N > 1: eager launches, bucket all-reduces overlapped with the backward kernels
    buckets = dp.GradientBuckets(list(net.parameters()), bucket_mb=args.bucket_mb, overlap=not use_graph) if world > 1 else None
    if buckets is not None:
        buckets.time_finish = True
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
            def captured():
                loss = fwd_bwd()
                if world == 1:
                    tail()
                return loss

            # weight re-packing is recorded INSIDE the graph (functional.PackedWeights.prepack under capture), so every replay trains
            # on the weights its predecessor's optimiser step produced
            graph = GraphedStep(captured, warmup=1, optimizers=[opt])
            mode = "hipgraph(fwd+bwd+opt)"
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
    comm = None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
        # exposed gradient-exchange time of the LAST step on this rank: how long the compute stream sat in finish() (waits on the
        # all-reduce stream + unpack kernels); everything else of the all-reduce ran hidden behind backward kernels
        exposed = None
        if buckets is not None and buckets.exposed_events is not None:
            exposed = buckets.exposed_events[0].elapsed_time(buckets.exposed_events[1])
        payload = sum(p.numel() for p in buckets.params) * 4 if buckets is not None else 0
        table = buckets.bucket_table() if buckets is not None else []
        span = (max(r["done_ms"] for r in table) - min(r["launch_ms"] for r in table)) if table and all("done_ms" in r for r in table) else None
        # every replica must hold the same parameters after the same number of averaged updates (weights are broadcast once, then only
        # all-reduced gradients move them): float64 checksum of every parameter, min == max over ranks
        chk = torch.stack([p.detach().double().sum() for p in net.parameters()])
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN); torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        comm = {"backend": torch.distributed.get_backend(), "rccl_ranks": torch.distributed.get_world_size() if torch.distributed.get_backend() == "nccl" else 0,
                "buckets": len(buckets.buckets) if buckets is not None else 0, "bucket_mb": args.bucket_mb, "allreduce_payload_bytes": payload,
                "exposed_allreduce_ms_last_step": None if exposed is None else round(exposed, 3),
                "allreduce_span_ms_last_step": None if span is None else round(span, 3),
                "bucket_table_last_step": table,
                "syncbn_allreduces_per_step": S.functional.SYNCBN_ALLREDUCES["last_step"],
                "replicas_identical": bool(torch.equal(lo, hi)), "physical_gpus": torch.cuda.device_count()}
    final_loss = float(loss.detach().float())

    # The live census drives REAL training steps (SyncBN all-reduces, hook-launched bucket all-reduces, buckets.finish()): with N > 1 it is
    # a collective operation, so EVERY rank runs it (rank 0 alone deadlocked the group); only rank 0's table is reported.
    census_out = None
    want_roofline = not args.no_roofline and (not (args.share_gpu and world > 1) or args.rehearsal_roofline)
    if want_roofline and world > 1:
        try:
            census_out = census_roofline(S, eager_step, dtype, args)
        except Exception as e:
            import traceback
            census_out = {"error": str(e)[:200], "where": traceback.format_exc()[-400:]}
        torch.cuda.synchronize()
        torch.distributed.barrier()

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
        if comm is not None:
            out["comm"] = comm
        if args.share_gpu and world > 1:
            # N ranks time-share ONE GPU: the aggregate is not an N-GPU measurement and must never be read as one
            out["rehearsal"] = "%d ranks share 1 GPU over gloo: real HIP step + bucketed all-reduce / SyncBN exchange under the bench's timing protocol; not a scaling measurement" % world
            out["rehearsal_slices_per_s"] = out["value"]
            out["value"] = None
        if want_roofline:
            try:
                # the dominant kernels of THIS round's committed profile, timed live over one real step's launch mix
                out["roofline"] = census_out if census_out is not None else census_roofline(S, eager_step, dtype, args)
                if "error" in out["roofline"]:
                    raise RuntimeError(out["roofline"]["error"])
                # round 1-4's headline kernel stays on the line as a fourth entry (one geometry + its 58-launch mix)
                dd = kernel_roofline(S, dtype, args.batch, args.size, launch_mix=not args.no_launch_mix)
                dd["traffic"] = pmc_traffic(dd["kernel"])
                out["roofline"]["dense_dgrad_probe"] = dd
                out["roofline"]["step"] = step_roofline(args, ms)     # per-GPU step: weak scaling, every rank does this work
                if rank == 0:
                    out["roofline"]["weakest_large_family"] = weakest_family_roofline(S, dtype, args.batch, args.size)
            except Exception as e:
                import traceback
                out["roofline"] = {"error": str(e)[:200], "where": traceback.format_exc()[-400:]}
        if world == 1 and not args.no_extras:
            # release the headline configuration first: the extras build their own nets / graphs
            del graph, net, sm, opt, feed, img, seg, edge
            S.functional.notify_params_changed(); S.functional.PACKS.clear()
            torch.cuda.empty_cache()
            out.update(extras(args, S, dev))
            S.set_compute_dtype(dtype)
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
