"""Per-phase cycle stamps of one kernel launch (profiling build: python -m saunet_amd._build --timing; run with
SAUNET_HIP_LIB=scripts/_ab/libsaunet_timing.so).  usage: phase_timing.py <case> ; cases: conv2fwd conv2wgrad conv1wgrad conv1dgrad conv1fwd dec3wgrad dec3fwd dec5fwd"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional
case = sys.argv[1]
dt = torch.bfloat16
n = 32
def act(c, h): return torch.randn(n, c, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
unit = "conv_tile"
if case in ("conv2fwd", "conv2fwd3", "conv2fwd4"):
    hh = {"conv2fwd": 128, "conv2fwd3": 32, "conv2fwd4": 16}[case]
    x = act(128, hh); w = torch.nn.Parameter(torch.randn(32, 128, 3, 3, device="cuda") * 0.03)
    sc = torch.rand(128, device="cuda") + 0.5; sh = torch.randn(128, device="cuda") * 0.1
    out = HF.new_act(n, 32, hh, hh, dt, "cuda"); st = torch.zeros(HF.STAT_R, 2, 32, dtype=torch.float64, device="cuda")
    run = lambda: HF.conv_forward_raw(x, w, None, 1, 1, pro=(sc, sh, True), out=out, stats=st)
elif case in ("conv2fwdbn1", "conv2fwdbn2"):
    unit = "dense_fwd"
    hh = {"conv2fwdbn1": 128, "conv2fwdbn2": 64}[case]
    z1 = act(128, hh); w = torch.nn.Parameter(torch.randn(32, 128, 3, 3, device="cuda") * 0.03)
    gamma = torch.rand(128, device="cuda") + 0.5; beta = torch.randn(128, device="cuda") * 0.1
    st2 = HF.bn_stats(z1); buf = HF.new_act(n, 96, hh, hh, dt, "cuda"); params = HF.BNParams(128, "cuda")
    rm, rv = torch.zeros(128, device="cuda"), torch.ones(128, device="cuda"); st_out = HF.new_stats(96, "cuda")
    run = lambda: HF.conv_forward_bnpro(z1, w, 1, 1, st2, n * hh * hh, 0, None, gamma, beta, rm, rv, 0.1, 1e-5, params.buf, out=buf[:, 32:64], stats=st_out[:, :, 32:64])
elif case in ("conv2wgrad", "conv1wgrad", "dec3wgrad"):
    cin, h, cout, k = {"conv2wgrad": (128, 128, 32, 3), "conv1wgrad": (192, 128, 128, 1), "dec3wgrad": (512, 64, 128, 3)}[case]
    x = act(cin, h); dy = act(cout, h); w = torch.nn.Parameter(torch.randn(cout, cin, k, k, device="cuda") * 0.03)
    sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda") * 0.1
    run = lambda: (HF.GRADS.reset(), HF.conv_wgrad_raw(x, dy, w, 1, k // 2, pro=(sc, sh, True) if case != "dec3wgrad" else None))
elif case in ("sc1", "sc2", "sc3"):
    h, cnt = {"sc1": (128, 6), "sc2": (64, 12), "sc3": (32, 24)}[case]
    probs = []
    for i in range(cnt):
        sc = torch.rand(128, device="cuda") + 0.5; sh = torch.randn(128, device="cuda") * 0.1
        probs.append((act(128, h), act(32, h), torch.nn.Parameter(torch.zeros(32, 128, 3, 3, device="cuda")), (sc, sh)))
    run = lambda: (HF.GRADS.reset(), HF.conv_wgrad_grouped(probs, 3, 1, True))
elif case in ("dec4convtwgrad", "dec2convtwgrad"):
    ci, h = {"dec4convtwgrad": (512, 16), "dec2convtwgrad": (128, 64)}[case]
    x = act(ci, h); dy = act(ci, 2 * h); w = torch.nn.Parameter(torch.randn(ci, ci, 4, 4, device="cuda") * 0.03)
    run = lambda: (HF.GRADS.reset(), HF.conv_wgrad_raw(x, dy, w, 2, 1, transposed=True))
elif case in ("dec3wgradmm",):
    x = act(512, 64); dy = act(128, 64); w = torch.nn.Parameter(torch.randn(128, 512, 3, 3, device="cuda") * 0.03)
    run = lambda: (HF.GRADS.reset(), HF.conv_wgrad_raw(x, dy, w, 1, 1))
elif case in ("dec3mm", "dec5mm", "dec4mm", "dec2mm"):
    unit = "conv_mm"
    cin, h, cout = {"dec3mm": (512, 64, 128), "dec5mm": (1536, 16, 512), "dec4mm": (1024, 32, 256), "dec2mm": (256, 128, 64)}[case]
    x = act(cin, h); w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device="cuda") * 0.02)
    out = HF.new_act(n, cout, h, h, dt, "cuda"); st = torch.zeros(HF.STAT_R, 2, cout, dtype=torch.float64, device="cuda")
    run = lambda: HF.conv_forward_raw(x, w, None, 1, 1, out=out, stats=st)
elif case in ("dec3fwd", "dec5fwd"):
    cin, h, cout = {"dec3fwd": (512, 64, 128), "dec5fwd": (1536, 16, 512)}[case]
    x = act(cin, h); w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device="cuda") * 0.02)
    out = HF.new_act(n, cout, h, h, dt, "cuda"); st = torch.zeros(HF.STAT_R, 2, cout, dtype=torch.float64, device="cuda")
    run = lambda: HF.conv_forward_raw(x, w, None, 1, 1, out=out, stats=st)
elif case in ("conv1fwd", "conv1fwd3", "conv1fwd4"):
    unit = "conv_igemm"
    ci_, h_ = {"conv1fwd": (192, 128), "conv1fwd3": (640, 32), "conv1fwd4": (768, 16)}[case]
    x = act(ci_, h_); w = torch.nn.Parameter(torch.randn(128, ci_, 1, 1, device="cuda") * 0.03)
    sc = torch.rand(ci_, device="cuda") + 0.5; sh = torch.randn(ci_, device="cuda") * 0.1
    out = HF.new_act(n, 128, h_, h_, dt, "cuda"); st = torch.zeros(HF.STAT_R, 2, 128, dtype=torch.float64, device="cuda")
    run = lambda: HF.conv_forward_raw(x, w, None, 1, 0, pro=(sc, sh, True), out=out, stats=st)
elif case in ("conv1small3", "conv1small4"):
    unit = "dense_fwd"
    ci_, h_ = {"conv1small3": (640, 32), "conv1small4": (768, 16)}[case]
    buf = act(1024, h_); x = buf[:, :ci_]; w = torch.nn.Parameter(torch.randn(128, ci_, 1, 1, device="cuda") * 0.03)
    gam = torch.rand(ci_, device="cuda") + 0.5; bet = torch.randn(ci_, device="cuda") * 0.1
    HF.STATS.reset()
    stats = HF.bn_stats(x); xh = torch.zeros(5, 1024, device="cuda")
    HF.L.call("saunet_bn_xhat", ci_ - 32, stats[0, 0].data_ptr(), stats[0, 1].data_ptr(), stats.shape[0], stats.stride(0), float(n * h_ * h_), 1e-5, xh.data_ptr(), xh.stride(0), HF.L.stream())
    prm = HF.BNParams(ci_, "cuda"); rm = torch.zeros(ci_, device="cuda"); rv = torch.ones(ci_, device="cuda")
    out = HF.new_act(n, 128, h_, h_, dt, "cuda"); st = torch.zeros(HF.STAT_R, 2, 128, dtype=torch.float64, device="cuda")
    run = lambda: HF.conv_forward_bnpro(x, w, 1, 0, stats, n * h_ * h_, ci_ - 32, xh, gam, bet, rm, rv, 0.1, 1e-5, prm.buf, out=out, stats=st)
elif case in ("k4lds3", "k4lds4"):
    unit = "dense_dgrad"
    import ctypes as C
    cin, ctot, h = {"k4lds3": (640, 1024, 32), "k4lds4": (768, 1024, 16)}[case]
    L = S.lib
    buf = act(ctot, h); dbuf = act(ctot, h); g = act(128, h); z1 = act(128, h); dz1 = act(128, h)
    w = torch.nn.Parameter(torch.randn(128, cin, 1, 1, device="cuda") * 0.05)
    p1 = HF.BNParams(cin, "cuda"); p1.buf[0].uniform_(0.5, 1.5); p1.buf[1].normal_(0, 0.3); p1.buf[2].normal_(0, 0.3); p1.buf[3].uniform_(0.5, 1.5)
    p2 = HF.BNParams(128, "cuda"); p2.buf[0].uniform_(0.5, 1.5); p2.buf[1].normal_(0, 0.3); p2.buf[2].normal_(0, 0.3); p2.buf[3].uniform_(0.5, 1.5)
    s1 = torch.zeros(HF.STAT_R, 2, cin, dtype=torch.float64, device="cuda"); s2 = torch.randn(HF.STAT_R, 2, 128, dtype=torch.float64, device="cuda")
    ab = torch.zeros(HF.STAT_R, 2, ctot, dtype=torch.float64, device="cuda"); xh = torch.zeros(5, ctot, device="cuda"); dgb = torch.zeros(2, 128, device="cuda")
    w1p = HF.PACKS.get(w, L.PACK_DGRAD, dt)
    d = L.DenseLayerBwd()
    d.N, d.H, d.W, d.Cin, d.Ctot = n, h, h, cin, ctot
    d.buf, d.dbuf, d.xhat, d.ld_xhat = buf.data_ptr(), dbuf.data_ptr(), xh.data_ptr(), xh.stride(0)
    d.ab, d.ab_replicas, d.ab_rstride, d.count = ab.data_ptr(), ab.shape[0], ab.stride(0), float(n * h * h)
    d.z1, d.g, d.dz1, d.w1_dgrad, d.p1, d.p2 = z1.data_ptr(), g.data_ptr(), dz1.data_ptr(), w1p.data_ptr(), p1.buf.data_ptr(), p2.buf.data_ptr()
    d.sums2, d.sums2_replicas, d.sums2_rstride = s2.data_ptr(), s2.shape[0], s2.stride(0)
    d.sums1, d.sums1_replicas, d.sums1_rstride = s1.data_ptr(), s1.shape[0], s1.stride(0)
    d.dgamma2, d.dbeta2 = dgb[0].data_ptr(), dgb[1].data_ptr()
    run = lambda: L.call("saunet_dense_layer_backward_conv1", C.byref(d), L.stream())
elif case in ("k3corr1", "k3corr2"):
    unit = "dense_dgrad"
    import ctypes as C
    cin, ctot, h = {"k3corr1": (160, 256, 128), "k3corr2": (320, 512, 64)}[case]
    L = S.lib
    buf = act(ctot, h); dbuf = act(ctot, h); g = act(128, h); z1 = act(128, h); dz2 = act(32, h)
    xh = torch.rand(5, ctot, device="cuda"); ab = torch.randn(HF.STAT_R, 2, ctot, dtype=torch.float64, device="cuda")
    w2 = torch.nn.Parameter(torch.randn(32, 128, 3, 3, device="cuda") * 0.05)
    p2 = HF.BNParams(128, "cuda"); p2.buf[0].uniform_(0.5, 1.5); p2.buf[1].normal_(0, 0.3); p2.buf[2].normal_(0, 0.3); p2.buf[3].uniform_(0.5, 1.5)
    s2 = torch.zeros(HF.STAT_R, 2, 128, dtype=torch.float64, device="cuda")
    w2p = HF.PACKS.get(w2, L.PACK_DGRAD, dt)
    d = L.DenseLayerBwd()
    d.N, d.H, d.W, d.Cin, d.Ctot = n, h, h, cin, ctot
    d.buf, d.dbuf, d.xhat, d.ld_xhat = buf.data_ptr(), dbuf.data_ptr(), xh.data_ptr(), xh.stride(0)
    d.ab, d.ab_replicas, d.ab_rstride, d.count = ab.data_ptr(), ab.shape[0], ab.stride(0), float(n * h * h)
    d.z1, d.g, d.dz2, d.w2_dgrad, d.p2 = z1.data_ptr(), g.data_ptr(), dz2.data_ptr(), w2p.data_ptr(), p2.buf.data_ptr()
    d.sums2, d.sums2_replicas, d.sums2_rstride = s2.data_ptr(), s2.shape[0], s2.stride(0)
    run = lambda: L.call("saunet_dense_layer_backward_conv2", C.byref(d), L.stream())
elif case in ("k4pair3", "k4pair2"):
    unit = "dense_dgrad"
    import ctypes as C
    cin, ctot, h = {"k4pair3": (640, 1024, 32), "k4pair2": (320, 512, 64)}[case]
    L = S.lib
    buf = act(ctot, h); dbuf = act(ctot, h); g = act(128, h)
    xh = torch.zeros(5, ctot, device="cuda"); ab = torch.zeros(HF.STAT_R, 2, ctot, dtype=torch.float64, device="cuda")
    keep = []
    def desc(c):
        z1 = act(128, h); dz1 = act(128, h)
        w = torch.nn.Parameter(torch.randn(128, c, 1, 1, device="cuda") * 0.05)
        p1 = HF.BNParams(c, "cuda"); p1.buf[0].uniform_(0.5, 1.5); p1.buf[1].normal_(0, 0.3); p1.buf[2].normal_(0, 0.3); p1.buf[3].uniform_(0.5, 1.5)
        p2 = HF.BNParams(128, "cuda"); p2.buf[0].uniform_(0.5, 1.5); p2.buf[1].normal_(0, 0.3); p2.buf[2].normal_(0, 0.3); p2.buf[3].uniform_(0.5, 1.5)
        s1 = torch.zeros(HF.STAT_R, 2, c, dtype=torch.float64, device="cuda"); s2 = torch.randn(HF.STAT_R, 2, 128, dtype=torch.float64, device="cuda")
        dgb = torch.zeros(2, 128, device="cuda"); w1p = HF.PACKS.get(w, L.PACK_DGRAD, dt)
        d = L.DenseLayerBwd()
        d.N, d.H, d.W, d.Cin, d.Ctot = n, h, h, c, ctot
        d.buf, d.dbuf, d.xhat, d.ld_xhat = buf.data_ptr(), dbuf.data_ptr(), xh.data_ptr(), xh.stride(0)
        d.ab, d.ab_replicas, d.ab_rstride, d.count = ab.data_ptr(), ab.shape[0], ab.stride(0), float(n * h * h)
        d.z1, d.g, d.dz1, d.w1_dgrad, d.p1, d.p2 = z1.data_ptr(), g.data_ptr(), dz1.data_ptr(), w1p.data_ptr(), p1.buf.data_ptr(), p2.buf.data_ptr()
        d.sums2, d.sums2_replicas, d.sums2_rstride = s2.data_ptr(), s2.shape[0], s2.stride(0)
        d.sums1, d.sums1_replicas, d.sums1_rstride = s1.data_ptr(), s1.shape[0], s1.stride(0)
        d.dgamma2, d.dbeta2 = dgb[0].data_ptr(), dgb[1].data_ptr()
        keep.extend([z1, dz1, w, p1, p2, s1, s2, dgb, w1p])
        return d
    dhi, dlo = desc(cin + 32), desc(cin)
    run = lambda: L.call("saunet_dense_layer_backward_conv1_pair", C.byref(dhi), C.byref(dlo), L.stream())
elif case in ("conv1dgrad", "conv1dgrad3", "conv1dgrad4"):
    unit = "dense_dgrad"
    cin, ctot, h = {"conv1dgrad": (192, 256, 128), "conv1dgrad3": (640, 1024, 32), "conv1dgrad4": (768, 1024, 16)}[case]
    buf = act(ctot, h); dbuf = act(ctot, h); g = act(128, h)
    w = torch.nn.Parameter(torch.randn(128, cin, 1, 1, device="cuda") * 0.05)
    p = HF.BNParams(cin, "cuda"); p.buf[0].uniform_(0.5, 1.5); p.buf[1].normal_(0, 0.3); p.buf[2].normal_(0, 0.3); p.buf[3].uniform_(0.5, 1.5)
    sums = torch.zeros(HF.STAT_R, 2, cin, dtype=torch.float64, device="cuda")
    run = lambda: HF.conv_dgrad_raw(g, w, (n, cin, h, h), 1, 0, out=dbuf[:, :cin], bn_epi=(buf[:, :cin], p, True, sums, True))
elif case in ("conv2dgrad", "conv2dgrad3", "conv2dgrad4"):
    unit = "dense_dgrad"
    h = {"conv2dgrad": 128, "conv2dgrad3": 32, "conv2dgrad4": 16}[case]
    dbuf = act(256, h); z1 = act(128, h)
    w = torch.nn.Parameter(torch.randn(32, 128, 3, 3, device="cuda") * 0.05)
    p = HF.BNParams(128, "cuda"); p.buf[0].uniform_(0.5, 1.5); p.buf[1].normal_(0, 0.3); p.buf[2].normal_(0, 0.3); p.buf[3].uniform_(0.5, 1.5)
    sums = torch.zeros(HF.STAT_R, 2, 128, dtype=torch.float64, device="cuda")
    run = lambda: HF.conv_dgrad_raw(dbuf[:, 64:96], w, z1.shape, 1, 1, bn_epi=(z1, p, True, sums))
for _ in range(3):
    run()
torch.cuda.synchronize()
lib = S.lib.load()
N = 2000
buf_ = (ctypes.c_ulonglong * N)()
getattr(lib, "saunet_debug_timing_" + unit)(buf_, N)
vals = [(v >> 56, v & 0x00ffffffffffffff) for v in buf_ if v]
t0 = vals[0][1]; prev = t0
import collections
agg = collections.defaultdict(list)
last = None
for slot, t in vals:
    if last is not None:
        agg[(last, slot)].append(t - prev)
    prev = t; last = slot
print("case", case, "stamps", len(vals), "total cycles", vals[-1][1] - t0)
for k, v in sorted(agg.items()):
    v2 = sorted(v)
    print("  %2d -> %2d : n=%3d  median %6d  min %6d  max %6d  sum %8d" % (k[0], k[1], len(v), v2[len(v2) // 2], v2[0], v2[-1], sum(v)))
