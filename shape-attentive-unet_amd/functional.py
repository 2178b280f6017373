"""Autograd layer over the C ABI: every forward/backward below is a sequence of libsaunet_hip.so calls.

PyTorch supplies tensors, streams and the autograd tape only.  Activations are logical NCHW tensors in
``torch.channels_last`` memory (= NHWC), possibly channel slices of a wider buffer (``ld`` > C).
Backward passes are written by hand so that producer/consumer fusion survives differentiation:
  * BatchNorm statistics are taken in the producing convolution's epilogue,
  * BatchNorm+ReLU is applied in the consuming convolution's operand load inside DenseNet,
  * concatenation is a write into a channel slice.
"""
import ctypes as C
import os
import weakref

import torch

from . import lib as L

_CL = torch.channels_last


# ------------------------------------------------------------------------------------------------ views
def nhwc(t):
    """Return t as a dense channels_last tensor (no copy if it already is one / a channel slice)."""
    if t.dim() != 4:
        raise RuntimeError("saunet_amd expects 4-D NCHW tensors, got %s" % (tuple(t.shape),))
    n, c, h, w = t.shape
    s = t.stride()
    if c == 1:
        if t.is_contiguous() or t.is_contiguous(memory_format=_CL):
            return t
    if s[1] == 1 and s[2] == w * s[3] and s[0] == h * w * s[3] and s[3] >= c:
        return t
    return t.contiguous(memory_format=_CL)


def ld_of(t):
    """channel stride (elements between consecutive pixels) of an NHWC view"""
    n, c, h, w = t.shape
    if w > 1:
        return t.stride(3)
    if h > 1:
        return t.stride(2)
    return t.stride(0) if n > 1 else c


def new_act(n, c, h, w, dtype, device, zero=False):
    t = torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=_CL)
    return t.zero_() if zero else t


def _check_dev(t):
    if not t.is_cuda:
        raise RuntimeError("saunet_amd ops run only on the GPU through libsaunet_hip.so (no CPU fallback)")


# ------------------------------------------------------------------------------------------------ raw op wrappers
class ZeroArena:
    """Zero-initialised scratch handed out as slices of a few big buffers (one memset each) instead of hundreds of
    tiny torch.zeros() fills per step: BN statistic accumulators (float64) and weight-gradient buffers (float32)."""

    def __init__(self, dtype, chunk):
        self.dtype, self.chunk = dtype, chunk
        self.buf, self.off = None, 0

    def reset(self):
        self.buf, self.off = None, 0

    def take(self, n, device):
        n_al = (n + 31) // 32 * 32  # keep every slice 128/256-byte aligned
        if n_al > self.chunk:
            return torch.zeros(n, dtype=self.dtype, device=device)
        if self.buf is None or self.buf.device != torch.device(device) or self.off + n_al > self.chunk:
            self.buf = torch.zeros(self.chunk, dtype=self.dtype, device=device)
            self.off = 0
        out = self.buf[self.off:self.off + n]
        self.off += n_al
        return out


STATS = ZeroArena(torch.float64, 1 << 21)      # 16 MB chunks
GRADS = ZeroArena(torch.float32, 1 << 24)      # 64 MB chunks


def zeros_f64(*shape, device):
    n = 1
    for s_ in shape:
        n *= s_
    return STATS.take(n, device).view(*shape)


STAT_R = 16   # replicated statistic accumulators (see saunet_conv_desc.stat_replicas)


def new_stats(c, device):
    """zeroed float64 accumulators [R, 2, C]: row 0 = sum, row 1 = sum of squares (or sum g / sum g*xhat in backward)"""
    return zeros_f64(STAT_R, 2, c, device=device)


def collapse_stats(st):
    """sum the replicas into replica 0 and return it as a flat [2*C] view"""
    r, two, c = st.shape
    L.call("saunet_sum_replicas", st.data_ptr(), 2 * c, r, st.stride(0), L.stream())
    return st[0].reshape(2 * c)


def capturing():
    """True while the current HIP stream is being captured into a hipGraph."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


SYNCBN_ALLREDUCES = {"count": 0, "last_step": 0}      # SyncBN statistic exchanges issued since the last begin_step() / during the last full step


def begin_step():
    """Called at the start of every SAUNet forward: new scratch arenas, all weight packings refreshed in bulk."""
    SYNCBN_ALLREDUCES["last_step"], SYNCBN_ALLREDUCES["count"] = SYNCBN_ALLREDUCES["count"], 0
    stale = len(_PENDING_AB)
    _PENDING_AB.clear()
    if stale:
        # a folded transition parked its coefficient sums and no dense block consumed them: the gradients of that backward pass were wrong
        raise RuntimeError("%d folded transition backward(s) of the previous step were never settled by a dense block" % stale)
    del _DEFERRED[:]              # (a backward pass that died mid-way: its partial gradients are gone with the arenas)
    _DENSE_BASES.clear()          # a reserved concat buffer nobody adopted (exception, standalone stem / transition) must not outlive its step
    STATS.reset(); GRADS.reset()
    PACKS.prepack()


def notify_params_changed():
    """Parameters / running statistics were changed through raw pointers without the host seeing it -- i.e. a captured
    hipGraph containing an optimiser step or a training-mode BatchNorm was replayed.  Marks every weight packing dirty and
    drops the eval-mode BatchNorm coefficient cache.  (Eager fused optimiser steps and training-mode finalizes do this
    themselves; only ``graph.replay()`` needs the explicit call -- saunet_amd.graph.GraphedStep makes it.)"""
    PACKS.invalidate()
    _EVAL_BN.clear()
    INFER.clear()


class PackedWeights:
    """Per-parameter cache of MFMA-friendly weight packings.  Entries are keyed by the Parameter object (weakref)
    and its version counter; `prepack` re-packs every packing seen so far in a handful of multi-tensor launches."""

    def __init__(self):
        self.cache = {}
        self.dirty = True
        self.generation = 0      # bumped whenever parameters / running statistics may have changed through a raw pointer

    def invalidate(self):
        """the fused optimisers update parameters through raw pointers (no version bump): they call this"""
        self.dirty = True
        self.generation += 1

    def _dims(self, w, mode):
        if mode in (L.PACK_CONVT_FWD, L.PACK_CONVT_DGRAD):
            ci, co, kh, kw = w.shape
        else:
            co, ci, kh, kw = w.shape
        return co, ci, kh, kw

    def prepack(self):
        # under stream capture the re-pack launches must become part of the graph: a replayed optimiser step changes the master
        # weights through raw pointers, so every replay has to refresh the packings itself (in place, addresses are stable)
        if not self.dirty and not capturing():
            return
        live = []
        for key, ent in list(self.cache.items()):
            w = ent[2]()
            if w is None or w.data_ptr() != ent[3]:
                del self.cache[key]
                continue
            live.append((key, ent, w))
        by_dtype = {}
        for key, ent, w in live:
            by_dtype.setdefault((key[2], w.device), []).append((key, ent, w))
        for (dtype, dev), items in by_dtype.items():
            with torch.cuda.device(dev):
                for s0 in range(0, len(items), 64):
                    pl = L.PackList()
                    chunk = items[s0:s0 + 64]
                    pl.count = len(chunk)
                    for i, (key, ent, w) in enumerate(chunk):
                        co, ci, kh, kw = self._dims(w, key[1])
                        pl.mode[i] = key[1]
                        pl.dims[i][0], pl.dims[i][1], pl.dims[i][2], pl.dims[i][3] = co, ci, kh, kw
                        pl.src[i] = w.data_ptr(); pl.dst[i] = ent[1].data_ptr()
                        self.cache[key] = (w._version, ent[1], ent[2], ent[3])
                    L.call("saunet_pack_weight_multi", C.byref(pl), L.BF16 if dtype == torch.bfloat16 else L.F32, L.stream())
        self.dirty = False

    def get(self, w, mode, dtype):
        # only leaf Parameters are cached, identified by object (weakref) + version counter: an address or an id()
        # can be recycled by another tensor, a live object cannot
        cacheable = isinstance(w, torch.nn.Parameter)
        key = (id(w), mode, dtype)
        ent = self.cache.get(key) if cacheable else None
        ver = w._version
        if ent is not None and ent[2]() is w and ent[0] == ver and ent[1].device == w.device and ent[3] == w.data_ptr() \
                and not self.dirty:
            return ent[1]
        if ent is not None and ent[2]() is w and ent[1].device == w.device and ent[1].numel() == w.numel():
            out = ent[1]      # re-pack in place (keeps addresses stable for graph replay)
        else:
            out = torch.empty(w.numel(), dtype=dtype, device=w.device)
        wd = w.detach()
        if not wd.is_contiguous() or wd.dtype != torch.float32:
            wd = wd.contiguous().float()
        co, ci, kh, kw = self._dims(w, mode)
        L.call("saunet_pack_weight", mode, L.BF16 if dtype == torch.bfloat16 else L.F32, wd.data_ptr(), co, ci, kh, kw,
               out.data_ptr(), L.stream())
        if cacheable:
            if len(self.cache) > 4096:
                self.cache = {k: v for k, v in self.cache.items() if v[2]() is not None}
            self.cache[key] = (ver, out, weakref.ref(w), w.data_ptr())
        return out

    def clear(self):
        self.cache.clear()
        self.dirty = True


PACKS = PackedWeights()


class InferenceCache:
    """Derived inference tensors (BatchNorm folded into convolution weights, packed for the MFMA kernels) keyed by the tensors they
    derive from: identity + version counter of each + PACKS.generation (bumped by fused optimiser steps, training-mode BatchNorm and
    notify_params_changed).  Entries are never modified in place: when a source changes a NEW entry is built, and once a hipGraph has
    been captured while the cache was in use, replaced entries are kept alive (a captured inference graph reads their addresses)."""

    def __init__(self):
        self.entries = {}
        self.retired = []
        self.seen_capture = False

    def get(self, tensors, build):
        key = tuple(id(t) for t in tensors)
        vers = tuple(t._version for t in tensors) + (PACKS.generation,)
        ent = self.entries.get(key)
        if capturing():
            self.seen_capture = True
        if ent is not None and ent[0] == vers and all(r() is t for r, t in zip(ent[1], tensors)):
            return ent[2]
        val = build()
        if capturing():
            # built inside a capture: its buffers belong to the graph's memory pool and hold nothing until the first replay -- serve it to
            # this capture only, never from the cache to a later eager call
            self.retired.append(val)
            return val
        if ent is not None and self.seen_capture:
            self.retired.append(ent[2])
        if len(self.entries) > 4096:
            self.entries = {k: v for k, v in self.entries.items() if all(r() is not None for r in v[1])}
        self.entries[key] = (vers, [weakref.ref(t) for t in tensors], val)
        return val

    def clear(self):
        if self.seen_capture:
            self.retired.extend(v[2] for v in self.entries.values())
        self.entries = {}


INFER = InferenceCache()


def _pack_now(w, mode, dtype):
    """pack a float32 weight tensor (not a Parameter) into a fresh buffer"""
    wd = w.detach()
    if not wd.is_contiguous() or wd.dtype != torch.float32:
        wd = wd.contiguous().float()
    if mode in (L.PACK_CONVT_FWD, L.PACK_CONVT_DGRAD):
        ci, co, kh, kw = wd.shape
    else:
        co, ci, kh, kw = wd.shape
    out = torch.empty(wd.numel(), dtype=dtype, device=wd.device)
    L.call("saunet_pack_weight", mode, L.BF16 if dtype == torch.bfloat16 else L.F32, wd.data_ptr(), co, ci, kh, kw, out.data_ptr(), L.stream())
    return out


def folded_conv_bn(weight, bias, bn, transposed, dtype):
    """Inference: (packed w', b') with BatchNorm folded in -- w' = w * gamma/sqrt(var+eps) per output channel, b' = beta - mean*scale
    (+ scale * conv bias).  Cached until the weights or the running statistics change."""
    srcs = [weight, bn.weight, bn.bias, bn.running_mean, bn.running_var] + ([bias] if bias is not None else [])

    def build():
        p = bn_finalize(None, 1, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, False)
        wf = weight.detach() * (p.scale.view(1, -1, 1, 1) if transposed else p.scale.view(-1, 1, 1, 1))
        bf = p.shift.clone() if bias is None else torch.addcmul(p.shift, bias.detach(), p.scale)
        return _pack_now(wf, L.PACK_CONVT_FWD if transposed else L.PACK_FWD, dtype), bf

    return INFER.get(srcs + [_DTYPE_TOKEN[dtype]], build)


_DTYPE_TOKEN = {torch.float32: torch.zeros(0), torch.bfloat16: torch.zeros(0)}    # distinct key objects per storage dtype


def _desc(x, cout, ldy, ho, wo, kh, kw, stride, pad, transposed=False, pro_relu=False, cin=None):
    d = L.ConvDesc()
    d.dtype = L.dtype_code(x)
    d.N, d.H, d.W = x.shape[0], x.shape[2], x.shape[3]
    d.Cin = x.shape[1] if cin is None else cin
    d.ldx = ld_of(x)
    d.Ho, d.Wo, d.Cout, d.ldy = ho, wo, cout, ldy
    d.KH, d.KW, d.stride, d.pad = kh, kw, stride, pad
    d.transposed = 1 if transposed else 0
    d.pro_relu = 1 if pro_relu else 0
    d.stat_replicas, d.stat_rstride = 1, 0
    return d


def _attach_workspace(d, device):
    """Caller-owned scratch of the forward entry points (saunet_conv2d_forward_workspace): allocated from torch's stream-ordered caching
    allocator -- inside a hipGraph capture from the graph's private pool -- and handed over in the descriptor.  Only the 8 x 8 `center`
    geometry asks for one today, so the query is skipped for everything else.  Returns the tensor (keep it referenced until the launch)."""
    if d.H != 8 or d.W != 8 or d.KH != 3:
        return None
    need = L.load().saunet_conv2d_forward_workspace(C.byref(d))
    if need <= 0:
        return None
    ws = torch.empty(int(need), dtype=torch.uint8, device=device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), int(need)
    return ws


def conv_out_hw(h, w, kh, kw, stride, pad, transposed):
    if transposed:
        return 2 * h, 2 * w
    return (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1


def conv_forward_raw(x, weight, bias, stride, pad, transposed=False, pro=None, out=None, stats=None, act_relu=False, packed=None):
    """y = conv(prologue(x), weight) + bias.  pro = (scale, shift, relu) or None.
    out: optional pre-allocated (channel-slice) destination.  stats = [R, 2, Cout] float64 accumulators (may be a
    channel slice of a wider [R, 2, Ctot] buffer).  packed: the already packed (forward layout, storage dtype) form of `weight`
    -- inference passes its cached folded weights here; `weight` then only supplies the shape."""
    _check_dev(x)
    x = nhwc(x)
    if transposed:
        cin_w, cout, kh, kw = weight.shape
    else:
        cout, cin_w, kh, kw = weight.shape
    if cin_w != x.shape[1]:
        raise RuntimeError("conv: weight expects %d input channels, got %d" % (cin_w, x.shape[1]))
    ho, wo = conv_out_hw(x.shape[2], x.shape[3], kh, kw, stride, pad, transposed)
    if out is None:
        out = new_act(x.shape[0], cout, ho, wo, x.dtype, x.device)
    wp = packed if packed is not None else PACKS.get(weight, L.PACK_CONVT_FWD if transposed else L.PACK_FWD, x.dtype)
    d = _desc(x, cout, ld_of(out), ho, wo, kh, kw, stride, pad, transposed, bool(pro and pro[2]))
    if stats is not None:
        d.stat_replicas, d.stat_rstride = stats.shape[0], stats.stride(0)
    d.epi_relu = 1 if act_relu else 0
    _ws = _attach_workspace(d, x.device)
    L.call("saunet_conv2d_forward", C.byref(d), x.data_ptr(), wp.data_ptr(), L.ptr(bias),
           L.ptr(pro[0]) if pro else None, L.ptr(pro[1]) if pro else None, out.data_ptr(),
           stats[0, 0].data_ptr() if stats is not None else None, stats[0, 1].data_ptr() if stats is not None else None, L.stream())
    return out


def conv_forward_bnpro(x, weight, stride, pad, bn_stats_in, count, c_lo, xhat, gamma, beta, rmean, rvar, momentum, eps, params, out=None, stats=None):
    """y = conv(relu(BN(x)), weight) with the BatchNorm coefficients derived inside the convolution kernel from the raw statistics
    `bn_stats_in` ([R, 2, >=Cin] float64 accumulators of x's channels) -- saunet_conv2d_forward_bnpro: no finalize launch.  Channels below
    c_lo take their normalisation from the `xhat` rows ([5, ld]); `params` ([4, Cin] float32) receives scale / shift / mean / invstd."""
    _check_dev(x)
    x = nhwc(x)
    cout, cin_w, kh, kw = weight.shape
    if cin_w != x.shape[1]:
        raise RuntimeError("conv: weight expects %d input channels, got %d" % (cin_w, x.shape[1]))
    ho, wo = conv_out_hw(x.shape[2], x.shape[3], kh, kw, stride, pad, False)
    if out is None:
        out = new_act(x.shape[0], cout, ho, wo, x.dtype, x.device)
    wp = PACKS.get(weight, L.PACK_FWD, x.dtype)
    d = _desc(x, cout, ld_of(out), ho, wo, kh, kw, stride, pad, False, True)
    if stats is not None:
        d.stat_replicas, d.stat_rstride = stats.shape[0], stats.stride(0)
    p = L.BnPrologue()
    p.sum, p.sumsq = bn_stats_in[0, 0].data_ptr(), bn_stats_in[0, 1].data_ptr()
    p.replicas, p.rstride, p.count, p.eps, p.momentum = bn_stats_in.shape[0], bn_stats_in.stride(0), float(count), float(eps), float(momentum)
    p.c_lo, p.ld_xhat, p.xhat = c_lo, (xhat.stride(0) if xhat is not None else 0), L.ptr(xhat)
    p.gamma, p.beta, p.params, p.running_mean, p.running_var = gamma.data_ptr(), beta.data_ptr(), params.data_ptr(), L.ptr(rmean), L.ptr(rvar)
    L.call("saunet_conv2d_forward_bnpro", C.byref(d), x.data_ptr(), wp.data_ptr(), None, C.byref(p), out.data_ptr(),
           stats[0, 0].data_ptr() if stats is not None else None, stats[0, 1].data_ptr() if stats is not None else None, L.stream())
    return out


DENSE_BNPRO = True    # consumer-side BatchNorm finalize in the DenseNet layers (False: per-layer saunet_bn_finalize launches; module attribute for A/B runs and tests)


def igemm_ok(t, cin, cout):
    """mirror of igemm_supported() in csrc/conv_igemm.hip: is this (view, Cin, Cout) on the MFMA path?"""
    epc = 8 if t.dtype == torch.bfloat16 else 4
    return cin % epc == 0 and cout % epc == 0 and cin >= epc and cout >= 8 and ld_of(t) % epc == 0


DGRAD_ACCUMULATE = os.environ.get("SAUNET_DGRAD_ACCUMULATE", "1") != "0"      # A/B switch: 0 = separate add pass behind every accumulating data gradient


def conv_dgrad_raw(dy, weight, x_shape, stride, pad, transposed=False, out=None, bn_epi=None, accumulate=False):
    """dx of y = conv(x, weight): a forward convolution over dy with re-packed weights.
    bn_epi = (bn_x, BNParams, relu, sums): fuse the reduction pass of the BatchNorm backward that consumes dx.
    accumulate (without bn_epi): out += dx where the library has a kernel for it (saunet_conv2d_accumulate_supported), else dx is computed
    into a fresh tensor and added by a separate pass -- `out` holds the sum either way."""
    dy = nhwc(dy)
    n, cin, h, w = x_shape
    if out is None:
        out = new_act(n, cin, h, w, dy.dtype, dy.device)
    if transposed:
        # y = convT(x): dx[n,ih,iw,ci] = sum dy[n,2ih-1+kh,2iw-1+kw,co] w[ci,co,kh,kw]  -> stride-2 pad-1 4x4 conv over dy
        wp = PACKS.get(weight, L.PACK_CONVT_DGRAD, dy.dtype)
        d = _desc(dy, cin, ld_of(out), h, w, 4, 4, 2, 1)
    else:
        cout, _, kh, kw = weight.shape
        if stride != 1:
            raise RuntimeError("dgrad: only stride-1 convolutions need an input gradient on this path")
        wp = PACKS.get(weight, L.PACK_DGRAD, dy.dtype)
        d = _desc(dy, cin, ld_of(out), h, w, kh, kw, 1, kh - 1 - pad)
    _ws = _attach_workspace(d, dy.device)
    if accumulate and bn_epi is None:
        if transposed or not DGRAD_ACCUMULATE or not L.load().saunet_conv2d_accumulate_supported(C.byref(d)):
            tmp = conv_dgrad_raw(dy, weight, x_shape, stride, pad, transposed)
            copy_channels(tmp, out, accumulate=True)
            return out
        e = L.BnEpilogue()
        e.bn_x, e.ld_bn_x, e.relu, e.accumulate = None, 0, 0, 1
        e.scale = e.shift = e.mean = e.invstd = None
        e.sums, e.sums_replicas, e.sums_rstride = None, 1, 0
        L.call("saunet_conv2d_forward_ex", C.byref(d), dy.data_ptr(), wp.data_ptr(), None, None, None, out.data_ptr(), None, None,
               C.byref(e), L.stream())
        return out
    if bn_epi is not None:
        bx, p, relu, sums = bn_epi[:4]
        e = L.BnEpilogue()
        e.bn_x, e.ld_bn_x, e.relu = bx.data_ptr(), ld_of(bx), 1 if relu else 0
        e.accumulate = int(bn_epi[4]) if len(bn_epi) > 4 and bn_epi[4] else 0          # 1: y += scale * g; 2: y = scale * g
        e.relu_mask = bn_epi[5].data_ptr() if len(bn_epi) > 5 and bn_epi[5] is not None else None      # ReLU decisions as bits (a residual block's output)
        e.scale, e.shift, e.mean, e.invstd = p.scale.data_ptr(), p.shift.data_ptr(), p.mean.data_ptr(), p.invstd.data_ptr()
        e.sums = sums.data_ptr()
        e.sums_replicas, e.sums_rstride = (sums.shape[0], sums.stride(0)) if sums.dim() == 3 else (1, 0)
        L.call("saunet_conv2d_forward_ex", C.byref(d), dy.data_ptr(), wp.data_ptr(), None, None, None, out.data_ptr(), None, None,
               C.byref(e), L.stream())
        return out
    L.call("saunet_conv2d_forward", C.byref(d), dy.data_ptr(), wp.data_ptr(), None, None, None, out.data_ptr(), None, None,
           L.stream())
    return out


DENSE_WGRAD_BATCH_REDUCE = True   # the partial-gradient reductions of a dense block in one launch (False: one per convolution)


class _ShapeOnly:
    """stand-in for a weight tensor where only its shape matters (derived weight-gradient problems)"""

    def __init__(self, shape):
        self.shape = torch.Size(shape)

    def numel(self):
        return self.shape.numel()


def conv_wgrad_raw(x, dy, weight, stride, pad, transposed=False, pro=None, pending=None):
    """pending: optional list collecting the deferred cross-workgroup reductions of the tiled kernels (flush_wgrad_reductions runs them in
    one launch); the returned gradient is complete only after that flush."""
    x = nhwc(x); dy = nhwc(dy)
    return _conv_wgrad_impl(x, dy, weight, stride, pad, transposed, pro, pending)


WGRAD_BIAS_FUSED = os.environ.get("SAUNET_WGRAD_BIAS_FUSED", "1") != "0"      # bias gradient of the few-output 1x1 layers inside their weight-gradient pass (A/B, tests)


def conv_wgrad_bias_raw(x, dy, weight, stride, pad):
    """(dw, dbias) of a plain convolution in ONE pass where the library has a kernel for it (saunet_conv2d_wgrad_bias_supported: pointwise,
    Cout <= 4), else None -- the caller then runs conv_wgrad_raw + channel_sum."""
    if not WGRAD_BIAS_FUSED:
        return None
    x = nhwc(x); dy = nhwc(dy)
    cout, _, kh, kw = weight.shape
    d = _desc(x, cout, ld_of(dy), dy.shape[2], dy.shape[3], kh, kw, stride, pad, False, False)
    lib = L.load()
    if lib.saunet_conv2d_wgrad_bias_supported(C.byref(d)) != 1:
        return None
    need = lib.saunet_conv2d_wgrad_workspace(C.byref(d))
    if need < 0:
        return None
    dw = GRADS.take(weight.numel(), x.device).view(weight.shape)         # zeroed arena: both gradients are accumulated into
    db = GRADS.take(cout, x.device)
    ws = torch.empty(max(need, 4) // 4, dtype=torch.float32, device=x.device)
    L.call("saunet_conv2d_wgrad_bias", C.byref(d), x.data_ptr(), dy.data_ptr(), None, None, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), need, L.stream())
    return dw, db


# Deferred weight-gradient reductions (round 6): every layer that goes through _conv_wgrad_impl with a leaf parameter leaves its per-group partial
# gradients in its workspace and the reductions of the WHOLE backward pass run as one saunet_wgrad_reduce_multi launch from an autograd final
# callback (30 small launches per step otherwise).  Not taken outside a backward pass, for derived weights (their gradient is consumed at once by
# the ops that derive them) or when the parameter already holds a gradient (AccumulateGrad would add the incomplete tensor immediately).
# SAUNET_WGRAD_DEFER=0 restores the per-layer reductions (A/B, tests).  dp.GradBuckets flushes before it packs a bucket.
WGRAD_DEFER = os.environ.get("SAUNET_WGRAD_DEFER", "1") != "0"
_DEFERRED = []       # (saunet_wgrad_pending, workspace tensor) of the backward pass in flight


def _defer_ok(weight):
    return (WGRAD_DEFER and isinstance(weight, torch.Tensor) and weight.is_leaf and weight.grad is None
            and torch._C._current_graph_task_id() >= 0)


def flush_deferred_wgrads():
    """Run the reductions collected so far (no-op when there are none).  Called by the autograd engine at the end of the backward pass, on the
    stream that surrounds the user's backward() call (which the engine has synchronised with the streams the gradients were produced on)."""
    if _DEFERRED:
        items = list(_DEFERRED)
        del _DEFERRED[:]
        flush_wgrad_reductions(items)


def flush_wgrad_reductions(pending):
    """dw += sum over groups of the partial gradients, for every entry collected by conv_wgrad_raw(..., pending=list): ONE launch per 64."""
    for i in range(0, len(pending), L.WGRAD_REDUCE_MAX):
        chunk = pending[i:i + L.WGRAD_REDUCE_MAX]
        lst = L.WgradReduceList()
        lst.count = len(chunk)
        for j, entry in enumerate(chunk):
            lst.item[j] = entry[0]
        L.call("saunet_wgrad_reduce_multi", C.byref(lst), L.stream())
    del pending[:]


DENSE_WGRAD_GROUPED = True   # all weight gradients of a dense block as two grouped launches (False: per-layer launches)
DENSE_COEFF_CORRECT = True   # coefficient + chunk correction of the linear BatchNorm backward in one launch (False: two)
# Layer pairs in the fused dense backward (round 5): two consecutive layers add their conv1 data gradients to dbuf in ONE pass over buf / dbuf
# where the library supports it (saunet_dense_layer_backward_pair_supported: the low-resolution blocks); SAUNET_DENSE_BWD_PAIRS=0 restores the
# per-layer sequence (A/B, tests).
DENSE_BWD_PAIRS = os.environ.get("SAUNET_DENSE_BWD_PAIRS", "1") != "0"
DENSE_TRANSITION_FOLD = os.environ.get("SAUNET_TRANSITION_FOLD", "1") != "0"  # the transition's BN-backward apply folded into the block's linear form
DENSE_BWD_FUSED = os.environ.get("SAUNET_DENSE_BWD_FUSED", "1") != "0"       # bf16 training: two launches per dense layer in backward (saunet_dense_layer_backward_conv2 / _conv1; False: the round-4 four)


def conv_wgrad_grouped(problems, ksize, pad, pro_relu):
    """Weight gradients of several stride-1 convolutions over the SAME map geometry in ONE launch (saunet_conv2d_wgrad_grouped).
    problems: list of (x, dy, weight, (pro_scale, pro_shift) or None).  Returns the list of gradients, or None when the geometry is not
    served by the tiled kernels (the caller then falls back to one conv_wgrad_raw per problem)."""
    if not problems:
        return []
    x0 = nhwc(problems[0][0])
    out = []
    for i0 in range(0, len(problems), L.WGRAD_GROUP_MAX):
        chunk = problems[i0:i0 + L.WGRAD_GROUP_MAX]
        g = L.WgradGroup()
        g.dtype, g.N, g.H, g.W = L.dtype_code(x0), x0.shape[0], x0.shape[2], x0.shape[3]
        g.KH, g.pad, g.pro_relu, g.count = ksize, pad, 1 if pro_relu else 0, len(chunk)
        dws, keep = [], []
        for i, (x, dy, weight, pro) in enumerate(chunk):
            x = nhwc(x); dy = nhwc(dy)
            dw = GRADS.take(weight.numel(), x.device).view(weight.shape)
            it = g.item[i]
            it.x, it.dy, it.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
            it.pro_scale, it.pro_shift = (pro[0].data_ptr(), pro[1].data_ptr()) if pro is not None else (None, None)
            it.Cin, it.ldx, it.Cout, it.lddy = x.shape[1], ld_of(x), dy.shape[1], ld_of(dy)
            dws.append(dw); keep.append((x, dy))
        need = L.load().saunet_conv2d_wgrad_grouped_workspace(C.byref(g))
        if need < 0:
            if out:
                raise RuntimeError("grouped weight gradient: chunk %d of one geometry is unsupported (%d)" % (i0, need))
            return None
        ws = torch.empty(max(need, 4) // 4, dtype=torch.float32, device=x0.device)
        L.call("saunet_conv2d_wgrad_grouped", C.byref(g), ws.data_ptr(), need, L.stream())
        out += dws
    return out


# ConvTranspose2d(k4 s2 p1) weight gradients straight from the NHWC tensors on the tiled kernel (four parity sub-problems with 2x2 taps, csrc/conv_tile.hip
# KS == 2); "0" = the round-2 path (im2col of dy + pointwise weight gradient), kept for maps that are not multiples of 16 and as the A/B reference
CONVT_WGRAD_DIRECT = True


def _conv_wgrad_impl(x, dy, weight, stride, pad, transposed=False, pro=None, pending=None):
    direct_t = (CONVT_WGRAD_DIRECT and transposed and pro is None and x.dtype == torch.bfloat16 and tuple(weight.shape[2:]) == (4, 4)
                and stride == 2 and pad == 1            # (tile_wgrad_convt_supported's own conditions, mirrored)
                and x.shape[2] % 16 == 0 and x.shape[3] % 16 == 0 and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0 and ld_of(x) % 8 == 0
                and ld_of(dy) % 8 == 0 and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0)
    if transposed and not direct_t and pro is None and weight.shape[2:] == (4, 4) and weight.shape[1] % 8 == 0 and weight.shape[0] % 8 == 0:
        # ConvTranspose2d(k=4, s=2, p=1):  dW[ci][co][kh][kw] = sum x[n,ih,iw,ci] * dy[n, 2ih-1+kh, 2iw-1+kw, co]  is the weight
        # gradient of a POINTWISE conv from im2col(dy; k=4, s=2, p=1) (K order kh,kw,co) to x: one gather pass over dy,
        # then the transposing-read matrix-core kernel instead of the generic atomic split-K one (3-5x faster).
        ci, co = weight.shape[0], weight.shape[1]
        cols = im2col(dy, 4, 4, 2, 1)
        dwp = _conv_wgrad_impl(cols, x, _ShapeOnly((ci, 16 * co, 1, 1)), 1, 0)
        return dwp.view(ci, 4, 4, co).permute(0, 3, 1, 2).contiguous()
    if (not transposed and pro is None and tuple(weight.shape[2:]) == (3, 3) and stride == 1 and pad == 1 and (x.shape[2] % 16 or x.shape[3] % 16)
            and (x.shape[0] * x.shape[2] * x.shape[3]) % 256 == 0 and x.shape[1] % 8 == 0 and weight.shape[0] % 8 == 0):
        # 3x3 convs on maps smaller than the 16x16 pixel tile (the 8x8 `center`): im2col (a few MB) + pointwise weight gradient
        co, ci = weight.shape[0], weight.shape[1]
        dwp = _conv_wgrad_impl(im2col(x, 3, 3, 1, 1), dy, _ShapeOnly((co, 9 * ci, 1, 1)), 1, 0)
        return dwp.view(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()
    dw = GRADS.take(weight.numel(), x.device).view(weight.shape)
    if transposed:
        _, cout, kh, kw = weight.shape
    else:
        cout, _, kh, kw = weight.shape
    d = _desc(x, cout, ld_of(dy), dy.shape[2], dy.shape[3], kh, kw, stride, pad, transposed, bool(pro and pro[2]))
    need = L.load().saunet_conv2d_wgrad_workspace(C.byref(d))
    if need < 0:
        raise RuntimeError("saunet_conv2d_wgrad_workspace failed (%d)" % need)
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device) if need > 0 else None
    if pending is None and ws is not None and _defer_ok(weight):
        # the cross-workgroup reduction of this layer joins ONE launch at the end of the backward pass (flush_deferred_wgrads).  Only the raw
        # pointers are kept: a second reference to dw would make AccumulateGrad clone the (still incomplete) gradient instead of adopting it;
        # the GRADS arena owns the memory until the next begin_step()
        pd = L.WgradPending()
        L.call("saunet_conv2d_wgrad_deferred", C.byref(d), x.data_ptr(), dy.data_ptr(), L.ptr(pro[0]) if pro else None,
               L.ptr(pro[1]) if pro else None, dw.data_ptr(), L.ptr(ws), need, C.byref(pd), L.stream())
        if pd.groups > 0:
            if not _DEFERRED:
                torch.autograd.Variable._execution_engine.queue_callback(flush_deferred_wgrads)
            _DEFERRED.append((pd, ws))
        return dw
    if pending is not None and ws is not None:
        pd = L.WgradPending()
        L.call("saunet_conv2d_wgrad_deferred", C.byref(d), x.data_ptr(), dy.data_ptr(), L.ptr(pro[0]) if pro else None,
               L.ptr(pro[1]) if pro else None, dw.data_ptr(), L.ptr(ws), need, C.byref(pd), L.stream())
        if pd.groups > 0:
            pending.append((pd, ws, dw))        # the workspace and the gradient stay alive until the flush
        return dw
    L.call("saunet_conv2d_wgrad", C.byref(d), x.data_ptr(), dy.data_ptr(), L.ptr(pro[0]) if pro else None,
           L.ptr(pro[1]) if pro else None, dw.data_ptr(), L.ptr(ws), need, L.stream())
    return dw


def im2col(x, kh, kw, stride, pad):
    """[N,C,H,W] (NHWC memory) -> [N, kh*kw*C, Ho, Wo] with K order (kh, kw, c); no gradient (used on the input image)."""
    x = nhwc(x)
    n, c, h, w = x.shape
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, False)
    out = new_act(n, kh * kw * c, ho, wo, x.dtype, x.device)
    L.call("saunet_im2col", L.dtype_code(x), x.data_ptr(), n, h, w, c, ld_of(x), kh, kw, stride, pad, out.data_ptr(), ld_of(out), L.stream())
    return out


def channel_sum(t):
    t = nhwc(t)
    n, c, h, w = t.shape
    acc = zeros_f64(c, device=t.device)
    L.call("saunet_channel_sum", L.dtype_code(t), t.data_ptr(), n * h * w, c, ld_of(t), acc.data_ptr(), L.stream())
    return acc.float()


def bn_stats(x, stats=None):
    x = nhwc(x)
    n, c, h, w = x.shape
    if stats is None:
        stats = new_stats(c, x.device)
    L.call("saunet_bn_stats", L.dtype_code(x), x.data_ptr(), n * h * w, c, ld_of(x), stats[0, 0].data_ptr(), stats[0, 1].data_ptr(),
           stats.shape[0], stats.stride(0), L.stream())
    return stats


class BNParams:
    """scale/shift/mean/invstd vectors [4,C] float32 produced by bn_finalize."""
    __slots__ = ("buf",)

    def __init__(self, c, device):
        self.buf = torch.empty(4, c, dtype=torch.float32, device=device)

    scale = property(lambda s: s.buf[0])
    shift = property(lambda s: s.buf[1])
    mean = property(lambda s: s.buf[2])
    invstd = property(lambda s: s.buf[3])


_EVAL_BN = {}


def bn_finalize(stats, count, gamma, beta, rmean, rvar, momentum, eps, training, conv_bias=None):
    """stats: [R, 2, C] accumulators (or a channel slice of them); None in eval mode.
    Eval-mode results depend on the parameters only and are cached per BatchNorm (150 tiny launches per inference forward
    otherwise); the cache is dropped whenever a training-mode finalize or a fused optimiser step may have changed them."""
    c = gamma.shape[0]
    if training and rmean is not None:
        PACKS.generation += 1                  # running statistics are about to change through a raw pointer
    key = None
    # never read or fill the cache while a hipGraph is being captured: the captured forward must contain its own finalize
    # launches (gamma / beta / running statistics change between replays) and must not reference cache-owned buffers
    if not training and stats is None and conv_bias is None and rmean is not None and not capturing():
        key = (id(rmean), id(gamma))
        vers = (gamma._version, beta._version, rmean._version, rvar._version, float(eps), PACKS.generation, gamma.device)
        ent = _EVAL_BN.get(key)
        if ent is not None and ent[0] == vers and ent[1]() is rmean and ent[2]() is gamma:
            return ent[3]
    p = BNParams(c, gamma.device)
    reps, rstr = (stats.shape[0], stats.stride(0)) if stats is not None else (1, 0)
    L.call("saunet_bn_finalize", c, stats[0, 0].data_ptr() if stats is not None else None,
           stats[0, 1].data_ptr() if stats is not None else None, reps, rstr, float(count), L.ptr(conv_bias), gamma.data_ptr(),
           beta.data_ptr(), float(eps), float(momentum), L.ptr(rmean), L.ptr(rvar), p.scale.data_ptr(), p.shift.data_ptr(),
           p.mean.data_ptr(), p.invstd.data_ptr(), 1 if training else 0, L.stream())
    if key is not None:
        if len(_EVAL_BN) > 2048:
            _EVAL_BN.clear()
        _EVAL_BN[key] = (vers, weakref.ref(rmean), weakref.ref(gamma), p)
    return p


def relu_mask_ok(x, residual=None):
    """the bit-mask forms (saunet_affine_act_mask / saunet_bn_backward_*_masked) serve bf16 tensors with 8-channel chunks"""
    ts = [x] + ([residual] if residual is not None else [])
    return RELU_MASK and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0 and all(ld_of(t) % 8 == 0 and t.data_ptr() % 16 == 0 for t in ts)


RELU_MASK = os.environ.get("SAUNET_RELU_MASK", "1") != "0"      # residual blocks keep their ReLU decisions as bits for the backward pass (A/B, tests)


def affine_act(x, scale, shift, relu, residual=None, out=None, pool=None, mask=None):
    """pool: optional float32 [N, C] tensor that receives the global average pool of the OUTPUT (SE squeeze fused into this pass)
    mask: optional uint8 [N*H*W*C/8] tensor that receives the ReLU decisions as bits (relu_mask_ok(x, residual) must hold)"""
    x = nhwc(x)
    n, c, h, w = x.shape
    if out is None:
        out = new_act(n, c, h, w, x.dtype, x.device)
    if mask is not None:
        if not relu or pool is not None:
            raise ValueError("affine_act: the ReLU bit mask needs relu=True and no pool")
        if residual is not None:
            residual = nhwc(residual)
        L.call("saunet_affine_act_mask", L.dtype_code(x), x.data_ptr(), ld_of(x), L.ptr(scale), L.ptr(shift), L.ptr(residual),
               ld_of(residual) if residual is not None else 0, out.data_ptr(), ld_of(out), n * h * w, c, mask.data_ptr(), L.stream())
        return out
    if pool is not None:
        epc = 8 if x.dtype == torch.bfloat16 else 4
        if residual is None and c % epc == 0 and ld_of(x) % epc == 0 and ld_of(out) % epc == 0 and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0:
            L.call("saunet_affine_act_pool", L.dtype_code(x), x.data_ptr(), ld_of(x), scale.data_ptr(), shift.data_ptr(), 1 if relu else 0,
                   out.data_ptr(), ld_of(out), n * h * w, c, pool.data_ptr(), h * w, L.stream())
            return out
        out = affine_act(x, scale, shift, relu, residual, out)
        L.call("saunet_global_avgpool", L.dtype_code(out), out.data_ptr(), n, h * w, c, ld_of(out), pool.data_ptr(), L.stream())
        return out
    if residual is not None:
        residual = nhwc(residual)
    L.call("saunet_affine_act", L.dtype_code(x), x.data_ptr(), ld_of(x), L.ptr(scale), L.ptr(shift), L.ptr(residual),
           ld_of(residual) if residual is not None else 0, 1 if relu else 0, out.data_ptr(), ld_of(out), n * h * w, c, L.stream())
    return out


BN_FINALIZE_FUSED = os.environ.get("SAUNET_BN_FINALIZE_FUSED", "1") != "0"     # consumer-side finalize in the BN-apply pass (A/B, tests)


def bn_affine_act(z, stats, count, gamma, beta, rmean, rvar, momentum, eps, relu, residual=None, out=None, pool=None, mask=None, conv_bias=None):
    """y = act(BN(z) (+residual)) in training mode from the raw statistics `stats` of z, returns (y, BNParams).  One launch where the library
    serves the geometry (saunet_affine_act_bn / _pool_bn: the finalize happens in the pass's prologue), else bn_finalize + affine_act."""
    z = nhwc(z)
    n, c, h, w = z.shape
    epc = 8 if z.dtype == torch.bfloat16 else 4
    if residual is not None:
        residual = nhwc(residual)
    views = [z] + ([residual] if residual is not None else []) + ([out] if out is not None else [])
    ok = (BN_FINALIZE_FUSED and z.is_cuda and stats is not None and z.dtype in (torch.bfloat16, torch.float32) and c % epc == 0 and c <= 4096
          and all(ld_of(t) % epc == 0 and t.data_ptr() % 16 == 0 for t in views) and not (pool is not None and (residual is not None or mask is not None)))
    if not ok:
        p = bn_finalize(stats, count, gamma, beta, rmean, rvar, momentum, eps, True, conv_bias=conv_bias)
        return affine_act(z, p.scale, p.shift, relu, residual, out=out, pool=pool, mask=mask), p
    if rmean is not None:
        PACKS.generation += 1                  # running statistics are about to change through a raw pointer
    p = BNParams(c, z.device)
    if out is None:
        out = new_act(n, c, h, w, z.dtype, z.device)
    pro = L.BnPrologue()
    pro.sum, pro.sumsq = stats[0, 0].data_ptr(), stats[0, 1].data_ptr()
    pro.replicas, pro.rstride, pro.count, pro.eps, pro.momentum = stats.shape[0], stats.stride(0), float(count), float(eps), float(momentum)
    pro.c_lo, pro.ld_xhat, pro.xhat = 0, 0, None
    pro.gamma, pro.beta, pro.params, pro.running_mean, pro.running_var = gamma.data_ptr(), beta.data_ptr(), p.buf.data_ptr(), L.ptr(rmean), L.ptr(rvar)
    if pool is not None:
        L.call("saunet_affine_act_pool_bn", L.dtype_code(z), z.data_ptr(), ld_of(z), C.byref(pro), L.ptr(conv_bias), 1 if relu else 0, out.data_ptr(), ld_of(out),
               n * h * w, c, pool.data_ptr(), h * w, L.stream())
    else:
        L.call("saunet_affine_act_bn", L.dtype_code(z), z.data_ptr(), ld_of(z), C.byref(pro), L.ptr(conv_bias), L.ptr(residual),
               ld_of(residual) if residual is not None else 0, 1 if relu else 0, out.data_ptr(), ld_of(out), n * h * w, c, L.ptr(mask), L.stream())
    return out, p


def bn_backward(dy, x, p, relu, count, training, residual=None, dx=None, accumulate=False, want_dres=False, sync_group=None,
                presums=None, mask=None):
    """Returns (dx, dres, dgamma, dbeta).  x is the tensor BN normalised (pre-affine).
    sync_group: all-reduce the two per-channel sums over that process group between the reduce and the
    apply kernel (SynchronizedBatchNorm semantics, lib/nn/modules/batchnorm.py:98-139); `count` must then
    already be the global element count."""
    dy = nhwc(dy); x = nhwc(x)
    n, c, h, w = x.shape
    P = n * h * w
    dev = x.device
    if residual is not None:
        residual = nhwc(residual)
    rp, rl = L.ptr(residual), (ld_of(residual) if residual is not None else 0)
    dt = L.dtype_code(x)
    if presums is not None:
        # dy is already g = dy*[relu mask] and the two sums were taken in the producing dgrad kernel's epilogue
        st, relu = presums, False
    elif mask is not None:
        # the ReLU decisions of the forward pass as bits (affine_act(mask=...)): the residual is not read again
        st = new_stats(c, dev)
        L.call("saunet_bn_backward_reduce_masked", dt, dy.data_ptr(), ld_of(dy), x.data_ptr(), ld_of(x), mask.data_ptr(), p.scale.data_ptr(),
               p.shift.data_ptr(), p.mean.data_ptr(), p.invstd.data_ptr(), st.data_ptr(), st.shape[0], st.stride(0), P, c, L.stream())
    else:
        st = new_stats(c, dev)
        L.call("saunet_bn_backward_reduce", dt, dy.data_ptr(), ld_of(dy), x.data_ptr(), ld_of(x), rp, rl, p.scale.data_ptr(),
               p.shift.data_ptr(), p.mean.data_ptr(), p.invstd.data_ptr(), 1 if relu else 0, st.data_ptr(), st.shape[0], st.stride(0),
               P, c, L.stream())
    sums, sreps, srstr = st, st.shape[0], st.stride(0)       # the apply kernel adds the replicas itself (once per block, via LDS)
    local = None
    if sync_group is not None and training:
        sums, sreps, srstr = collapse_stats(st), 1, 0
        # dx needs the GLOBAL sums (the shared statistics couple the ranks); dgamma / dbeta are gradients of the LOCAL loss and
        # are averaged with all other parameter gradients afterwards, so they must come from the local sums
        local = sums.clone()
        torch.distributed.all_reduce(sums, group=sync_group)
        SYNCBN_ALLREDUCES["count"] += 1
    if dx is None:
        dx = new_act(n, c, h, w, x.dtype, dev)
    dres = new_act(n, c, h, w, x.dtype, dev) if want_dres else None
    dgb = torch.empty(2, c, dtype=torch.float32, device=dev)
    if mask is not None and presums is None:
        L.call("saunet_bn_backward_apply_masked", dt, dy.data_ptr(), ld_of(dy), x.data_ptr(), ld_of(x), mask.data_ptr(), p.scale.data_ptr(),
               p.shift.data_ptr(), p.mean.data_ptr(), p.invstd.data_ptr(), sums.data_ptr(), sreps, srstr, float(count),
               1 if training else 0, 1 if accumulate else 0, dx.data_ptr(), ld_of(dx), L.ptr(dres),
               ld_of(dres) if dres is not None else 0, dgb[0].data_ptr(), dgb[1].data_ptr(), P, c, L.stream())
        if local is not None:
            return dx, dres, local[c:].float(), local[:c].float()
        return dx, dres, dgb[0], dgb[1]
    L.call("saunet_bn_backward_apply", dt, dy.data_ptr(), ld_of(dy), x.data_ptr(), ld_of(x), rp, rl, p.scale.data_ptr(),
           p.shift.data_ptr(), p.mean.data_ptr(), p.invstd.data_ptr(), 1 if relu else 0, sums.data_ptr(), sreps, srstr, float(count),
           1 if training else 0, 1 if accumulate else 0, dx.data_ptr(), ld_of(dx), L.ptr(dres),
           ld_of(dres) if dres is not None else 0, dgb[0].data_ptr(), dgb[1].data_ptr(), P, c, L.stream())
    if local is not None:
        return dx, dres, local[c:].float(), local[:c].float()
    return dx, dres, dgb[0], dgb[1]


def copy_channels(src, dst, accumulate=False):
    src = nhwc(src)
    n, c, h, w = src.shape
    L.call("saunet_copy_channels", L.dtype_code(src), L.dtype_code(dst), src.data_ptr(), ld_of(src), dst.data_ptr(), ld_of(dst),
           n * h * w, c, 1 if accumulate else 0, L.stream())
    return dst


# ------------------------------------------------------------------------------------------------ autograd functions
class _Conv(torch.autograd.Function):
    """Plain convolution / transposed convolution (no normalisation)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, transposed):
        x = nhwc(x)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, transposed, bias is not None)
        return conv_forward_raw(x, weight, bias, stride, pad, transposed)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad, transposed, has_bias = ctx.cfg
        dy = nhwc(dy)
        dx = conv_dgrad_raw(dy, weight, x.shape, stride, pad, transposed) if ctx.needs_input_grad[0] else None
        if has_bias and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and not transposed and x.is_cuda:
            both = conv_wgrad_bias_raw(x, dy, weight, stride, pad)       # few-output 1x1 layers: the bias gradient rides in the weight-gradient pass
            if both is not None:
                return dx, both[0], both[1], None, None, None
        dw = conv_wgrad_raw(x, dy, weight, stride, pad, transposed) if ctx.needs_input_grad[1] else None
        db = channel_sum(dy) if has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None, None


def conv2d(x, weight, bias=None, stride=1, padding=0):
    return _Conv.apply(x, weight, bias, stride, padding, False)


def conv_transpose2d(x, weight, bias=None):
    return _Conv.apply(x, weight, bias, 2, 1, True)


def _bump(bn):
    """num_batches_tracked += 1 (skipped when the owning SAUNet increments all counters with one fused add)"""
    if bn.training and bn.num_batches_tracked is not None and not getattr(bn, "_nbt_fused", False):
        bn.num_batches_tracked.add_(1)


class _ConvBNAct(torch.autograd.Function):
    """y = act(BN(conv(x)) [+ residual]) with the batch statistics taken in the conv epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, rmean, rvar, residual, stride, pad, transposed, relu, momentum, eps, training, group,
                sync_bufs=None, out=None, pool=None):
        x = nhwc(x)
        cout = weight.shape[1] if transposed else weight.shape[0]
        stats = new_stats(cout, x.device) if training else None
        z = conv_forward_raw(x, weight, bias, stride, pad, transposed, stats=stats)
        count = z.shape[0] * z.shape[2] * z.shape[3]
        y = None
        if group is not None and training:
            # SynchronizedBatchNorm across replicas (lib/nn/modules/batchnorm.py:98-139): global batch statistics = all-reduce of
            # (sum, sumsq), count scales with the world, inv_std = clamp(var, eps)^-1/2, running statistics through the reference's
            # (_tmp_running_*, _running_iter) accumulator
            if bias is not None:
                raise RuntimeError("synchronised batch norm behind a biased convolution is not on the SAUNet path")
            flat = collapse_stats(stats)
            torch.distributed.all_reduce(flat, group=group)
            SYNCBN_ALLREDUCES["count"] += 1
            count *= torch.distributed.get_world_size(group)
            p = BNParams(cout, x.device)
            tm, tv, it = sync_bufs if sync_bufs is not None else (None, None, None)
            PACKS.generation += 1
            L.call("saunet_syncbn_finalize", cout, flat[:cout].data_ptr(), flat[cout:].data_ptr(), 1, 0, float(count), gamma.data_ptr(),
                   beta.data_ptr(), float(eps), float(momentum), L.ptr(tm), L.ptr(tv), L.ptr(it), L.ptr(rmean), L.ptr(rvar),
                   p.scale.data_ptr(), p.shift.data_ptr(), p.mean.data_ptr(), p.invstd.data_ptr(), L.stream())
        elif training:
            y, p = bn_affine_act(z, stats, count, gamma, beta, rmean, rvar, momentum, eps, relu, residual, out=out, pool=pool, conv_bias=bias)
        else:
            p = bn_finalize(stats, count, gamma, beta, rmean, rvar, momentum, eps, training, conv_bias=bias)
        if y is None:
            y = affine_act(z, p.scale, p.shift, relu, residual, out=out, pool=pool)
        ctx.save_for_backward(x, weight, z, p.buf, residual if residual is not None else z.new_empty(0))
        ctx.cfg = (stride, pad, transposed, relu, training, count, bias is not None, residual is not None, group)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, z, pbuf, residual = ctx.saved_tensors
        stride, pad, transposed, relu, training, count, has_bias, has_res, group = ctx.cfg
        p = BNParams.__new__(BNParams); p.buf = pbuf
        need_res = has_res and ctx.needs_input_grad[7]
        dz, dres, dgamma, dbeta = bn_backward(dy, z, p, relu, count, training, residual if has_res else None, want_dres=need_res,
                                              sync_group=group)
        dx = conv_dgrad_raw(dz, weight, x.shape, stride, pad, transposed) if ctx.needs_input_grad[0] else None
        dw = conv_wgrad_raw(x, dz, weight, stride, pad, transposed)
        db = None
        if has_bias:
            # a bias in front of a training-mode BatchNorm has an exactly zero gradient (sum of dz over the batch is 0)
            cout = dz.shape[1]
            db = (GRADS.take(cout, dz.device) if dz.is_cuda else torch.zeros(cout, dtype=torch.float32, device=dz.device)) if training else channel_sum(dz)
        return dx, dw, db, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None, None, None, None, None


def conv_bn_act(x, weight, bias, bn, relu=True, residual=None, stride=1, padding=0, transposed=False, out=None, pool=None):
    """bn: an nn.BatchNorm2d-like module (weight, bias, running_mean, running_var, momentum, eps, training).
    out: optional destination (a channel slice of a concat buffer -- see cat_alias).
    pool: optional float32 [N, Cout] tensor that receives the global average pool of the result (the SE squeeze rides on the BN-apply pass).
    A bn with a truthy ``sync`` attribute (SynchronizedBatchNorm2d) reduces its statistics over the default
    process group when one with more than one rank exists."""
    _bump(bn)
    if transposed:
        stride, padding = 2, 1  # the only transposed geometry on the path: ConvTranspose2d(k=4, s=2, p=1)
    if not bn.training and not torch.is_grad_enabled() and residual is None and x.is_cuda:
        # inference: BatchNorm folded into the convolution -- w' = w * gamma/sqrt(var+eps) per output channel, b' = the BN shift
        # (incl. the conv bias), ReLU in the conv epilogue: one kernel, no normalisation pass
        x = nhwc(x)
        if isinstance(weight, torch.nn.Parameter):
            wp, bf = folded_conv_bn(weight, bias, bn, transposed, x.dtype)          # folded + packed once, cached across calls
            y = conv_forward_raw(x, weight, bf, stride, padding, transposed, act_relu=relu, out=out, packed=wp)
        else:        # derived weight tensors (the stem's im2col matrix): fold on the fly
            p = bn_finalize(None, 1, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, False)
            wf = weight * (p.scale.view(1, -1, 1, 1) if transposed else p.scale.view(-1, 1, 1, 1))
            bf = p.shift if bias is None else torch.addcmul(p.shift, bias, p.scale)
            y = conv_forward_raw(x, wf, bf, stride, padding, transposed, act_relu=relu, out=out)
        if pool is not None:
            L.call("saunet_global_avgpool", L.dtype_code(y), y.data_ptr(), y.shape[0], y.shape[2] * y.shape[3], y.shape[1], ld_of(y), pool.data_ptr(), L.stream())
        return y
    group, sync_bufs = None, None
    if getattr(bn, "sync", False) and bn.training and torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1:
        group = torch.distributed.group.WORLD
        sync_bufs = (bn._tmp_running_mean, bn._tmp_running_var, bn._running_iter)
    return _ConvBNAct.apply(x, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, stride, padding,
                            transposed, relu, bn.momentum, bn.eps, bn.training, group, sync_bufs, out, pool)


def _sync_finalize(stats, count_local, gamma, beta, rmean, rvar, momentum, eps, group, sync_bufs):
    """SynchronizedBatchNorm finalize (lib/nn/modules/batchnorm.py:98-139) from local raw statistics: all-reduce of (sum, sumsq), global count,
    the reference's (_tmp_running_*, _running_iter) moving average.  -> (BNParams, global count)"""
    c = gamma.shape[0]
    flat = collapse_stats(stats)
    torch.distributed.all_reduce(flat, group=group)
    SYNCBN_ALLREDUCES["count"] += 1
    count = count_local * torch.distributed.get_world_size(group)
    p = BNParams(c, gamma.device)
    tm, tv, it = sync_bufs if sync_bufs is not None else (None, None, None)
    PACKS.generation += 1
    L.call("saunet_syncbn_finalize", c, flat[:c].data_ptr(), flat[c:].data_ptr(), 1, 0, float(count), gamma.data_ptr(),
           beta.data_ptr(), float(eps), float(momentum), L.ptr(tm), L.ptr(tv), L.ptr(it), L.ptr(rmean), L.ptr(rvar),
           p.scale.data_ptr(), p.shift.data_ptr(), p.mean.data_ptr(), p.invstd.data_ptr(), L.stream())
    return p, count


def _sync_args(bn1, bn2):
    """(process group, bn1's accumulator buffers, bn2's) when the block's SynchronizedBatchNorm layers have to exchange statistics, else (None, None, None)"""
    synced = getattr(bn1, "sync", False) and bn1.training and torch.distributed.is_available() and torch.distributed.is_initialized() \
        and torch.distributed.get_world_size() > 1
    if not synced:
        return None, None, None
    return (torch.distributed.group.WORLD, (bn1._tmp_running_mean, bn1._tmp_running_var, bn1._running_iter),
            (bn2._tmp_running_mean, bn2._tmp_running_var, bn2._running_iter))


class _BasicBlock(torch.autograd.Function):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + x), 3x3 stride-1 convolutions (models/resnet.py:30-60, the shape stream's res1-3), with bn1 + ReLU
    applied in conv2's operand load: the activated intermediate is never materialised (forward: one read + one write of the full-resolution
    tensor less) and bn1's backward reduction rides in conv2's data-gradient epilogue (one two-tensor pass less).  Single-process statistics
    only: SynchronizedBatchNorm across ranks keeps the unfused path (it needs the all-reduce between reduce and apply)."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, mom1, eps1, mom2, eps2, training, group=None, sb1=None, sb2=None):
        x = nhwc(x)
        c = w1.shape[0]
        st1 = new_stats(c, x.device) if training else None
        z1 = conv_forward_raw(x, w1, None, 1, 1, stats=st1)
        count = z1.shape[0] * z1.shape[2] * z1.shape[3]
        st2 = new_stats(c, x.device) if training else None
        # (measured and rejected, round 6: bn1's finalize in conv2's operand-load prologue -- saunet_conv2d_forward_bnpro -- and the same for the
        # transitions' 1x1: step +0.05 ms same-box; thousands of full-resolution workgroups each repeat the replica sums before their first load)
        sync = group is not None and training
        if sync:         # SynchronizedBatchNorm across ranks: the statistics are all-reduced between the convolution and its BatchNorm
            p1, gcount = _sync_finalize(st1, count, g1, b1, rm1, rv1, mom1, eps1, group, sb1)
        else:
            p1, gcount = bn_finalize(st1, count, g1, b1, rm1, rv1, mom1, eps1, training), count
        z2 = conv_forward_raw(z1, w2, None, 1, 1, pro=(p1.scale, p1.shift, True), stats=st2)
        # the ReLU decisions of the block's output as bits (1/16 of the tensor): bn2's backward passes then read them instead of the skip tensor
        mask = torch.empty(z2.numel() // 8, dtype=torch.uint8, device=x.device) if relu_mask_ok(z2, x) else None
        if sync:
            p2, _ = _sync_finalize(st2, count, g2, b2, rm2, rv2, mom2, eps2, group, sb2)
            y = affine_act(z2, p2.scale, p2.shift, True, x, mask=mask)
        elif training:
            y, p2 = bn_affine_act(z2, st2, count, g2, b2, rm2, rv2, mom2, eps2, True, x, mask=mask)
        else:
            p2 = bn_finalize(st2, count, g2, b2, rm2, rv2, mom2, eps2, training)
            y = affine_act(z2, p2.scale, p2.shift, True, x, mask=mask)
        ctx.save_for_backward(x, w1, w2, z1, z2, p1.buf, p2.buf, mask if mask is not None else z2.new_empty(0))
        ctx.cfg = (training, gcount, mask is not None, group if sync else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, z1, z2, p1b, p2b, mask = ctx.saved_tensors
        training, count, masked, group = ctx.cfg
        p1 = BNParams.__new__(BNParams); p1.buf = p1b
        p2 = BNParams.__new__(BNParams); p2.buf = p2b
        if masked and nhwc(dy).data_ptr() % 16 == 0 and ld_of(nhwc(dy)) % 8 == 0:
            dz2, dres, dg2, db2 = bn_backward(dy, z2, p2, True, count, training, None, want_dres=True, mask=mask, sync_group=group)
        else:
            dz2, dres, dg2, db2 = bn_backward(dy, z2, p2, True, count, training, x, want_dres=True, sync_group=group)
        dw2 = conv_wgrad_raw(z1, dz2, w2, 1, 1, pro=(p1.scale, p1.shift, True))
        s1 = new_stats(z1.shape[1], z1.device)
        da1 = conv_dgrad_raw(dz2, w2, z1.shape, 1, 1, bn_epi=(z1, p1, True, s1))
        dz1, _, dg1, db1 = bn_backward(da1, z1, p1, True, count, training, dx=da1, presums=s1, sync_group=group)
        dw1 = conv_wgrad_raw(x, dz1, w1, 1, 1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv_dgrad_raw(dz1, w1, x.shape, 1, 1, out=dres, accumulate=True)      # on top of the skip branch's gradient
        return dx, dw1, dg1, db1, None, None, dw2, dg2, db2, None, None, None, None, None, None, None, None, None, None


class _BasicBlockConv(torch.autograd.Function):
    """conv1x1_k(BasicBlock(x)) as ONE autograd node (round 6).  In the shape stream every residual block feeds exactly one biased 1x1 convolution
    (res1 -> d1, res2 -> d2, res3 -> d3: /root/reference/models/models.py:337-356), so the gradient of the block's output has a single producer: the
    data gradient of that convolution.  Its epilogue applies the block's ReLU decisions (the bits kept by saunet_affine_act_mask) and takes the two
    BatchNorm-backward sums of bn2 while the tile is on chip (saunet_bn_epilogue.relu_mask), so bn2's reduce pass over (dy, z2) disappears, and the
    masked gradient it writes IS the skip branch's gradient: the apply pass no longer writes a second copy.  Three full-resolution tensor passes
    fewer per block than _BasicBlock + _Conv.  Forward = _BasicBlock.forward followed by the convolution."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, mom1, eps1, mom2, eps2, training, wk, bk, group=None, sb1=None, sb2=None):
        x = nhwc(x)
        c = w1.shape[0]
        st1 = new_stats(c, x.device) if training else None
        z1 = conv_forward_raw(x, w1, None, 1, 1, stats=st1)
        count = z1.shape[0] * z1.shape[2] * z1.shape[3]
        st2 = new_stats(c, x.device) if training else None
        sync = group is not None and training
        if sync:         # SynchronizedBatchNorm across ranks: the statistics are all-reduced between the convolution and its BatchNorm
            p1, gcount = _sync_finalize(st1, count, g1, b1, rm1, rv1, mom1, eps1, group, sb1)
        else:
            p1, gcount = bn_finalize(st1, count, g1, b1, rm1, rv1, mom1, eps1, training), count
        z2 = conv_forward_raw(z1, w2, None, 1, 1, pro=(p1.scale, p1.shift, True), stats=st2)
        mask = torch.empty(z2.numel() // 8, dtype=torch.uint8, device=x.device)
        if sync:
            p2, _ = _sync_finalize(st2, count, g2, b2, rm2, rv2, mom2, eps2, group, sb2)
            y = affine_act(z2, p2.scale, p2.shift, True, x, mask=mask)
        elif training:
            y, p2 = bn_affine_act(z2, st2, count, g2, b2, rm2, rv2, mom2, eps2, True, x, mask=mask)
        else:
            p2 = bn_finalize(st2, count, g2, b2, rm2, rv2, mom2, eps2, training)
            y = affine_act(z2, p2.scale, p2.shift, True, x, mask=mask)
        out = conv_forward_raw(y, wk, bk, 1, 0)
        ctx.save_for_backward(x, w1, w2, z1, z2, p1.buf, p2.buf, mask, y, wk)
        ctx.cfg = (training, gcount, bk is not None, group if sync else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w1, w2, z1, z2, p1b, p2b, mask, y, wk = ctx.saved_tensors
        training, count, has_bias, group = ctx.cfg
        p1 = BNParams.__new__(BNParams); p1.buf = p1b
        p2 = BNParams.__new__(BNParams); p2.buf = p2b
        dout = nhwc(dout)
        both = conv_wgrad_bias_raw(y, dout, wk, 1, 0) if has_bias else None
        if both is not None:
            dwk, dbk = both
        else:
            dwk = conv_wgrad_raw(y, dout, wk, 1, 0)
            dbk = channel_sum(dout) if has_bias else None
        # the 1x1 data gradient with bn2's backward reduction in its epilogue: g = dy * [block output > 0] (the skip branch's gradient), sum g, sum g * xhat2
        s2 = new_stats(z2.shape[1], z2.device)
        g = conv_dgrad_raw(dout, wk, y.shape, 1, 0, bn_epi=(z2, p2, True, s2, 0, mask))
        dz2, _, dg2, db2 = bn_backward(g, z2, p2, True, count, training, presums=s2, sync_group=group)
        dw2 = conv_wgrad_raw(z1, dz2, w2, 1, 1, pro=(p1.scale, p1.shift, True))
        s1 = new_stats(z1.shape[1], z1.device)
        da1 = conv_dgrad_raw(dz2, w2, z1.shape, 1, 1, bn_epi=(z1, p1, True, s1))
        dz1, _, dg1, db1 = bn_backward(da1, z1, p1, True, count, training, dx=da1, presums=s1, sync_group=group)
        dw1 = conv_wgrad_raw(x, dz1, w1, 1, 1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv_dgrad_raw(dz1, w1, x.shape, 1, 1, out=g, accumulate=True)        # on top of the skip branch's gradient (g is not read again)
        return dx, dw1, dg1, db1, None, None, dw2, dg2, db2, None, None, None, None, None, None, None, dwk, dbk, None, None, None


# SynchronizedBatchNorm inside the fused residual block (round 6): the statistic exchanges of bn1 / bn2 sit between the fused launches exactly where the
# unfused conv_bn_act pair has them (forward: after each convolution; backward: between the sums and the apply pass), so N > 1 runs keep the fused
# block's passes.  SAUNET_SYNC_BLOCK_FUSED=0 restores the two conv_bn_act calls under SyncBN (A/B, tests).
SYNC_BLOCK_FUSED = os.environ.get("SAUNET_SYNC_BLOCK_FUSED", "1") != "0"
BLOCK_CONV_FUSED = os.environ.get("SAUNET_BLOCK_CONV_FUSED", "1") != "0"     # residual block + the 1x1 convolution behind it as one node (A/B, tests)


def basic_block_conv1x1(x, blk, conv):
    """conv(blk(x)) for a BasicBlock module and the biased 1x1 nn.Conv2d that is its ONLY consumer; one fused node where the library serves it
    (bf16 storage, 8-channel chunks, local batch statistics), else the two calls."""
    x = nhwc(x)
    c, ck = blk.conv1.weight.shape[0], conv.weight.shape[0]
    group, sb1, sb2 = _sync_args(blk.bn1, blk.bn2)
    ok = (BLOCK_CONV_FUSED and FUSED_BASIC_BLOCK and RELU_MASK and (group is None or SYNC_BLOCK_FUSED) and x.is_cuda and x.dtype == torch.bfloat16
          and (blk.bn1.training or torch.is_grad_enabled()) and blk.bn1.training == blk.bn2.training and torch.is_grad_enabled()
          and tuple(conv.weight.shape[2:]) == (1, 1) and conv.stride == (1, 1) and c % 8 == 0 and ck % 8 == 0 and c >= 8 and ck >= 8
          and ld_of(x) % 8 == 0 and x.data_ptr() % 16 == 0 and blk.conv1.weight.shape[1] == c)
    if not ok:
        return conv2d(blk(x), conv.weight, conv.bias)
    _bump(blk.bn1); _bump(blk.bn2)
    return _BasicBlockConv.apply(x, blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.bn1.running_mean, blk.bn1.running_var, blk.conv2.weight,
                                 blk.bn2.weight, blk.bn2.bias, blk.bn2.running_mean, blk.bn2.running_var, blk.bn1.momentum, blk.bn1.eps,
                                 blk.bn2.momentum, blk.bn2.eps, blk.bn1.training, conv.weight, conv.bias, group, sb1, sb2)


# Default since round 3 (FUSED_BASIC_BLOCK = False restores the two conv_bn_act calls).  Rounds 1 and 2 measured it SLOWER (33.45 -> 33.86 ms per
# step): the three full-resolution passes it saves (0.3 ms) were lost to the BN-backward epilogue of the narrow 3x3 kernels, whose partial sums
# went through float atomics on LDS (12 cycles per active lane).  With the per-wave slot fold those epilogues are cheap: 30.23 -> 30.05 ms.
FUSED_BASIC_BLOCK = True


def basic_block(x, conv1, bn1, conv2, bn2):
    """BasicBlock forward; the fused form when it applies (training or grad mode on the GPU, batch statistics local to this process)."""
    group, sb1, sb2 = _sync_args(bn1, bn2)
    if not FUSED_BASIC_BLOCK or (group is not None and not SYNC_BLOCK_FUSED) or not x.is_cuda or not (bn1.training or torch.is_grad_enabled()) \
            or bn1.training != bn2.training:
        out = conv_bn_act(x, conv1.weight, None, bn1, relu=True, padding=1)
        return conv_bn_act(out, conv2.weight, None, bn2, relu=True, residual=x, padding=1)
    _bump(bn1); _bump(bn2)
    return _BasicBlock.apply(x, conv1.weight, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, conv2.weight, bn2.weight, bn2.bias,
                             bn2.running_mean, bn2.running_var, bn1.momentum, bn1.eps, bn2.momentum, bn2.eps, bn1.training, group, sb1, sb2)


class _BNAct(torch.autograd.Function):
    """Stand-alone BatchNorm (+ReLU) over a materialised tensor; optional precomputed statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, stats, relu, momentum, eps, training, out=None):
        x = nhwc(x)
        n, c, h, w = x.shape
        count = n * h * w
        if training and stats is None:
            stats = bn_stats(x)
        if training:
            y, p = bn_affine_act(x, stats, count, gamma, beta, rmean, rvar, momentum, eps, relu, out=out)
        else:
            p = bn_finalize(None, count, gamma, beta, rmean, rvar, momentum, eps, training)
            y = affine_act(x, p.scale, p.shift, relu, out=out)
        ctx.save_for_backward(x, p.buf)
        ctx.cfg = (relu, training, count)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pbuf = ctx.saved_tensors
        relu, training, count = ctx.cfg
        p = BNParams.__new__(BNParams); p.buf = pbuf
        dx, _, dgamma, dbeta = bn_backward(dy, x, p, relu, count, training)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


def batch_norm_act(x, bn, relu=False, stats=None, out=None):
    _bump(bn)
    return _BNAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, stats, relu, bn.momentum, bn.eps, bn.training, out)


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ho, wo, out=None):
        x = nhwc(x)
        n, c, h, w = x.shape
        y = out if out is not None else new_act(n, c, ho, wo, x.dtype, x.device)
        L.call("saunet_bilinear_forward", L.dtype_code(x), x.data_ptr(), n, h, w, c, ld_of(x), y.data_ptr(), ho, wo, ld_of(y), L.stream())
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = nhwc(dy)
        n, c, h, w = ctx.shape
        dx = new_act(n, c, h, w, dy.dtype, dy.device)
        L.call("saunet_bilinear_backward", L.dtype_code(dy), dy.data_ptr(), n, dy.shape[2], dy.shape[3], c, ld_of(dy), dx.data_ptr(),
               h, w, ld_of(dx), 0, L.stream())
        return dx, None, None, None


def interpolate_bilinear(x, size=None, scale_factor=None, out=None):
    """F.interpolate(mode='bilinear', align_corners=True).  out: optional destination (channel slice of a concat buffer)."""
    if size is None:
        size = (int(x.shape[2] * scale_factor), int(x.shape[3] * scale_factor))
    return _Bilinear.apply(x, int(size[0]), int(size[1]), out)


class _Pool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, is_max):
        x = nhwc(x)
        n, c, h, w = x.shape
        y = new_act(n, c, h // 2, w // 2, x.dtype, x.device)
        L.call("saunet_pool2x2_forward", L.dtype_code(x), 1 if is_max else 0, x.data_ptr(), n, h, w, c, ld_of(x), y.data_ptr(), ld_of(y), L.stream())
        ctx.is_max = is_max
        ctx.save_for_backward(x if is_max else x.new_empty(0))
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = nhwc(dy)
        n, c, h, w = ctx.shape
        dx = new_act(n, c, h, w, dy.dtype, dy.device)
        L.call("saunet_pool2x2_backward", L.dtype_code(dy), 1 if ctx.is_max else 0, x.data_ptr() if ctx.is_max else None, dy.data_ptr(),
               n, h, w, c, ld_of(x) if ctx.is_max else 0, ld_of(dy), dx.data_ptr(), ld_of(dx), 0, L.stream())
        return dx, None


def max_pool2x2(x):
    return _Pool2x2.apply(x, True)


def avg_pool2x2(x):
    return _Pool2x2.apply(x, False)


class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        n, c, h, w = x.shape
        y = new_act(n, c, h, w, x.dtype, x.device)
        L.call("saunet_sigmoid_forward", L.dtype_code(x), x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), n * h * w, c, L.stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = nhwc(dy)
        n, c, h, w = y.shape
        dx = new_act(n, c, h, w, y.dtype, y.device)
        L.call("saunet_sigmoid_backward", L.dtype_code(y), y.data_ptr(), ld_of(y), dy.data_ptr(), ld_of(dy), dx.data_ptr(), ld_of(dx),
               n * h * w, c, 0, L.stream())
        return dx


def sigmoid(x):
    return _Sigmoid.apply(x)


class _Cat(torch.autograd.Function):
    """torch.cat(dim=1) as channel-slice writes; backward hands out slice views of the incoming gradient."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [nhwc(x) for x in xs]
        n, _, h, w = xs[0].shape
        cs = [x.shape[1] for x in xs]
        out = new_act(n, sum(cs), h, w, xs[0].dtype, xs[0].device)
        o = 0
        for x, c in zip(xs, cs):
            copy_channels(x, out[:, o:o + c])
            o += c
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = nhwc(dy)
        outs, o = [], 0
        for c in ctx.cs:
            outs.append(dy[:, o:o + c])
            o += c
        return tuple(outs)


def cat(xs):
    return _Cat.apply(*xs)


class _CatAlias(torch.autograd.Function):
    """torch.cat(dim=1) of tensors that were PRODUCED in place: every xs[i] already is the channel slice [o_i, o_i + C_i) of `buf`
    (its producer was given that slice as destination), so the forward copies nothing; backward hands out slice views of dy."""

    @staticmethod
    def forward(ctx, buf, *xs):
        o, cs = 0, []
        for x in xs:
            c = x.shape[1]
            if x.data_ptr() != buf[:, o:o + c].data_ptr() or ld_of(x) != ld_of(buf) or x.shape[0] != buf.shape[0] or x.shape[2:] != buf.shape[2:]:
                raise RuntimeError("cat_alias: piece %d is not the slice [%d, %d) of the concat buffer" % (len(cs), o, o + c))
            cs.append(c); o += c
        if o != buf.shape[1]:
            raise RuntimeError("cat_alias: pieces cover %d of %d channels" % (o, buf.shape[1]))
        ctx.cs = cs
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, dy):
        dy = nhwc(dy)
        outs, o = [], 0
        for c in ctx.cs:
            outs.append(dy[:, o:o + c])
            o += c
        return (None,) + tuple(outs)


def cat_alias(buf, xs):
    return _CatAlias.apply(buf, *xs)


class _Cast(torch.autograd.Function):
    """storage-dtype conversion (bf16 <-> float32) of an activation; the gradient is cast back."""

    @staticmethod
    def forward(ctx, x, dtype, out=None):
        x = nhwc(x)
        n, c, h, w = x.shape
        ctx.src_dtype = x.dtype
        y = out if out is not None else new_act(n, c, h, w, dtype, x.device)
        copy_channels(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = nhwc(dy)
        n, c, h, w = dy.shape
        dx = new_act(n, c, h, w, ctx.src_dtype, dy.device)
        copy_channels(dy, dx)
        return dx, None, None


def cast(x, dtype, out=None):
    """out: optional destination of dtype `dtype` (a channel slice of a concat buffer); written even when no conversion is needed."""
    if x.dtype == dtype and out is None:
        return x
    return _Cast.apply(x, dtype, out)


class _GateMul(torch.autograd.Function):
    """x * (alpha + 1) with a one-channel alpha (GSConv.py:55)."""

    @staticmethod
    def forward(ctx, x, alpha):
        x = nhwc(x); alpha = nhwc(alpha)
        n, c, h, w = x.shape
        y = new_act(n, c, h, w, x.dtype, x.device)
        L.call("saunet_gate_mul_forward", L.dtype_code(x), x.data_ptr(), ld_of(x), alpha.data_ptr(), y.data_ptr(), ld_of(y), n * h * w, c, L.stream())
        ctx.save_for_backward(x, alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, alpha = ctx.saved_tensors
        dy = nhwc(dy)
        n, c, h, w = x.shape
        dx = new_act(n, c, h, w, x.dtype, x.device)
        dal = new_act(n, 1, h, w, x.dtype, x.device)
        L.call("saunet_gate_mul_backward", L.dtype_code(x), x.data_ptr(), ld_of(x), alpha.data_ptr(), dy.data_ptr(), ld_of(dy),
               dx.data_ptr(), ld_of(dx), dal.data_ptr(), n * h * w, c, L.stream())
        return dx, dal


def gate_mul(x, alpha):
    return _GateMul.apply(x, alpha)


class _GatedConv(torch.autograd.Function):
    """Fused GatedSpatialConv2d (GSConv.py:16-57) for bf16 storage and training-mode batch norm: two forward passes
    (z, then y/alpha) and three backward passes that recompute the gate chain per pixel (csrc/gate.hip)."""

    @staticmethod
    def forward(ctx, feat, gate, g0, b0, rm0, rv0, w1, b1, w2, b2, g1, bb1, rm1, rv1, wm, mom0, eps0, mom1, eps1, training):
        feat = nhwc(feat); gate = nhwc(gate)
        _check_dev(feat)
        n, c, h, w = feat.shape
        P, dev, c1 = n * h * w, feat.device, c + 1
        st0 = st1 = None
        if training:
            st0 = new_stats(c1, dev)
            bn_stats(feat, st0[:, :, :c]); bn_stats(gate, st0[:, :, c:])
            st1 = new_stats(1, dev)
        p0 = bn_finalize(st0, P, g0, b0, rm0, rv0, mom0, eps0, training)
        z = torch.empty(P, dtype=torch.float32, device=dev)
        L.call("saunet_gate_forward_z", L.BF16, c, feat.data_ptr(), ld_of(feat), gate.data_ptr(), ld_of(gate), P, p0.buf.data_ptr(),
               w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), z.data_ptr(),
               st1[0, 0].data_ptr() if training else None, st1[0, 1].data_ptr() if training else None,
               st1.shape[0] if training else 1, st1.stride(0) if training else 0, L.stream())
        p1 = bn_finalize(st1, P, g1, bb1, rm1, rv1, mom1, eps1, training)
        y = new_act(n, c, h, w, feat.dtype, dev)
        alpha = new_act(n, 1, h, w, feat.dtype, dev)
        L.call("saunet_gate_forward_out", L.BF16, c, feat.data_ptr(), ld_of(feat), z.data_ptr(), P, p1.buf.data_ptr(), wm.data_ptr(),
               y.data_ptr(), ld_of(y), alpha.data_ptr(), L.stream())
        ctx.save_for_backward(feat, gate, z, p0.buf, p1.buf, w1, b1, w2, wm)
        ctx.set_materialize_grads(False)
        ctx.training = training
        return y, alpha

    @staticmethod
    def backward(ctx, dy, dalpha):
        feat, gate, z, p0, p1, w1, b1, w2, wm = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("fused gated convolution: backward needs training-mode batch norm (use the unfused path)")
        n, c, h, w = feat.shape
        P, dev, c1 = n * h * w, feat.device, c + 1
        dy = nhwc(dy) if dy is not None else new_act(n, c, h, w, feat.dtype, dev, zero=True)
        if dalpha is not None:
            dalpha = nhwc(dalpha).contiguous()
        nbytes = L.load().saunet_gate_backward_workspace(P)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        q = torch.empty(P, dtype=torch.float32, device=dev)
        sizes = [c * c, 2, 3, c1 * c1, c1, c1, 1, 2 * c1, 3 * c1]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        dwm, dbn1, K, dw1, db1, dw2, db2, dbn0, E = torch.split(flat, sizes)
        st = L.stream()
        L.call("saunet_gate_backward_q", L.BF16, c, dy.data_ptr(), ld_of(dy), feat.data_ptr(), ld_of(feat), z.data_ptr(), L.ptr(dalpha), P,
               p1.data_ptr(), wm.data_ptr(), q.data_ptr(), dwm.data_ptr(), dbn1.data_ptr(), K.data_ptr(), ws.data_ptr(), nbytes, st)
        L.call("saunet_gate_backward_sums", L.BF16, c, feat.data_ptr(), ld_of(feat), gate.data_ptr(), ld_of(gate), q.data_ptr(), z.data_ptr(), P,
               K.data_ptr(), p0.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(),
               db2.data_ptr(), dbn0.data_ptr(), E.data_ptr(), ws.data_ptr(), nbytes, st)
        dfeat = new_act(n, c, h, w, feat.dtype, dev)
        dgate = new_act(n, 1, h, w, feat.dtype, dev)
        L.call("saunet_gate_backward_apply", L.BF16, c, dy.data_ptr(), ld_of(dy), feat.data_ptr(), ld_of(feat), gate.data_ptr(), ld_of(gate),
               q.data_ptr(), z.data_ptr(), P, K.data_ptr(), E.data_ptr(), p0.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
               p1.data_ptr(), wm.data_ptr(), dfeat.data_ptr(), ld_of(dfeat), dgate.data_ptr(), ld_of(dgate), st)
        return (dfeat, dgate, dbn0[:c1], dbn0[c1:], None, None, dw1.view_as(w1), db1, dw2.view_as(w2), db2, dbn1[0:1], dbn1[1:2], None, None,
                dwm.view_as(wm), None, None, None, None, None)


def gated_conv_fusable(feat, gate, m):
    """the fused kernels cover bf16 storage, C in {8,16,32}, no module bias, 16-byte aligned feature rows, and (for
    autograd) training-mode batch norm"""
    c = feat.shape[1]
    if feat.dtype != torch.bfloat16 or gate.dtype != torch.bfloat16 or c not in (8, 16, 32) or m.bias is not None or gate.shape[1] != 1:
        return False
    if not m.training and torch.is_grad_enabled():
        return False
    f = nhwc(feat)
    return ld_of(f) % 8 == 0 and f.data_ptr() % 16 == 0


def gated_conv(feat, gate, m):
    """m: a GatedSpatialConv2d-like module (``_gate_conv`` Sequential + own 1x1 weight).  Returns (y, alpha)."""
    g = m._gate_conv
    _bump(g[0]); _bump(g[4])
    return _GatedConv.apply(feat, gate, g[0].weight, g[0].bias, g[0].running_mean, g[0].running_var, g[1].weight, g[1].bias,
                            g[3].weight, g[3].bias, g[4].weight, g[4].bias, g[4].running_mean, g[4].running_var, m.weight,
                            g[0].momentum, g[0].eps, g[4].momentum, g[4].eps, m.training)


class _ExpandBNAct(torch.autograd.Function):
    """Conv2d(1, C, 1x1) -> BatchNorm -> ReLU on a one-channel float32 map as ONE per-pixel affine map (csrc/expand.hip)."""

    @staticmethod
    def forward(ctx, a, w, b, gamma, beta, rmean, rvar, momentum, eps, training, relu, out_dtype, out=None):
        a = nhwc(a)
        _check_dev(a)
        if not a.is_contiguous():          # the kernels index the one-channel map as a dense [P] vector
            a = a.contiguous()
        n, _, h, wd = a.shape
        P, dev, c = n * h * wd, a.device, w.shape[0]
        stats = bn_stats(a) if training else None
        coef = torch.empty(4, c, dtype=torch.float32, device=dev)
        mv = torch.empty(2, dtype=torch.float32, device=dev)
        L.call("saunet_expand_coeff", c, stats[0, 0].data_ptr() if training else None, stats[0, 1].data_ptr() if training else None,
               stats.shape[0] if training else 1, stats.stride(0) if training else 0, float(P), w.data_ptr(), L.ptr(b), gamma.data_ptr(),
               beta.data_ptr(), float(eps), float(momentum), L.ptr(rmean), L.ptr(rvar), coef.data_ptr(), mv.data_ptr(), 1 if training else 0, L.stream())
        y = out if out is not None else new_act(n, c, h, wd, out_dtype, dev)
        L.call("saunet_expand_forward", L.dtype_code(y), a.data_ptr(), P, c, coef.data_ptr(), y.data_ptr(), ld_of(y), 1 if relu else 0, L.stream())
        ctx.save_for_backward(a, coef, mv)
        ctx.cfg = (relu, training, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, coef, mv = ctx.saved_tensors
        relu, training, has_bias = ctx.cfg
        if not training:
            raise RuntimeError("fused expand: backward needs training-mode batch norm (use the unfused path)")
        dy = nhwc(dy)
        n, c, h, wd = dy.shape
        P, dev = n * h * wd, dy.device
        sums = new_stats(c, dev)
        flat = torch.empty(4 * c + 2, dtype=torch.float32, device=dev)
        dw, db, dg, dbeta, D = flat[:c], flat[c:2 * c], flat[2 * c:3 * c], flat[3 * c:4 * c], flat[4 * c:]
        da = torch.empty_like(a)
        L.call("saunet_expand_backward", L.dtype_code(dy), dy.data_ptr(), ld_of(dy), a.data_ptr(), P, c, coef.data_ptr(), mv.data_ptr(), 1 if relu else 0,
               sums.data_ptr(), sums.shape[0], sums.stride(0), dw.data_ptr(), db.data_ptr(), dg.data_ptr(), dbeta.data_ptr(), D.data_ptr(), da.data_ptr(),
               L.stream())
        return da, dw.view(c, 1, 1, 1), (db if has_bias else None), dg, dbeta, None, None, None, None, None, None, None, None


def expand_fusable(x, conv, bn):
    c = conv.out_channels
    return (conv.in_channels == 1 and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and c in (32, 64)
            and x.dtype == torch.float32 and x.is_cuda and (bn.training or not torch.is_grad_enabled()))


def expand_bn_act(x, conv, bn, relu=True, out_dtype=None, out=None):
    _bump(bn)
    return _ExpandBNAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, bn.training,
                              relu, out_dtype or x.dtype, out)


class _DualAttTail(torch.autograd.Function):
    """out = (S + 1) * F * sigmoid(fc2(relu(fc1(avgpool(F)))))  (attention_blocks.py:50-57, 237)."""

    @staticmethod
    def forward(ctx, F, S, w1, b1, w2, b2, pooled=None):
        F = nhwc(F); S = nhwc(S)
        n, c, h, w = F.shape
        cr = w1.shape[0]
        dev = F.device
        if pooled is None:      # the producer of F did not pool it on the way out: one extra read of F
            pooled = torch.empty(n, c, dtype=torch.float32, device=dev)
            L.call("saunet_global_avgpool", L.dtype_code(F), F.data_ptr(), n, h * w, c, ld_of(F), pooled.data_ptr(), L.stream())
        hidden = torch.empty(n, cr, dtype=torch.float32, device=dev)
        se = torch.empty(n, c, dtype=torch.float32, device=dev)
        w1c, w2c = w1.detach().reshape(cr, c), w2.detach().reshape(c, cr)
        L.call("saunet_se_excite", pooled.data_ptr(), n, c, cr, w1c.data_ptr(), b1.data_ptr(), w2c.data_ptr(), b2.data_ptr(),
               hidden.data_ptr(), se.data_ptr(), L.stream())
        out = new_act(n, c, h, w, F.dtype, dev)
        L.call("saunet_att_combine_forward", L.dtype_code(F), F.data_ptr(), ld_of(F), S.data_ptr(), se.data_ptr(), out.data_ptr(),
               ld_of(out), n, h * w, c, L.stream())
        ctx.save_for_backward(F, S, w1, w2, pooled, hidden, se)
        return out

    @staticmethod
    def backward(ctx, dout):
        F, S, w1, w2, pooled, hidden, se = ctx.saved_tensors
        dout = nhwc(dout)
        n, c, h, w = F.shape
        cr = w1.shape[0]
        dev = F.device
        dF = new_act(n, c, h, w, F.dtype, dev)
        dS = new_act(n, 1, h, w, F.dtype, dev)
        dse = torch.empty(n, c, dtype=torch.float32, device=dev)
        dt = L.dtype_code(F)
        L.call("saunet_att_combine_backward", dt, F.data_ptr(), ld_of(F), S.data_ptr(), se.data_ptr(), dout.data_ptr(), ld_of(dout),
               dF.data_ptr(), ld_of(dF), dS.data_ptr(), dse.data_ptr(), n, h * w, c, L.stream())
        g = GRADS.take(2 * c * cr + c + cr, dev) if F.is_cuda else torch.zeros(2 * c * cr + c + cr, dtype=torch.float32, device=dev)
        dw1, db1 = g[:cr * c], g[cr * c:cr * c + cr]
        dw2, db2 = g[cr * c + cr:2 * cr * c + cr], g[2 * cr * c + cr:]
        dpooled = torch.empty(n, c, dtype=torch.float32, device=dev)
        w1c, w2c = w1.detach().reshape(cr, c), w2.detach().reshape(c, cr)
        L.call("saunet_se_excite_backward", pooled.data_ptr(), hidden.data_ptr(), se.data_ptr(), dse.data_ptr(), n, c, cr,
               w1c.data_ptr(), w2c.data_ptr(), dpooled.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), L.stream())
        L.call("saunet_add_pooled_grad", dt, dF.data_ptr(), ld_of(dF), dpooled.data_ptr(), n, h * w, c, L.stream())
        return dF, dS, dw1.view(w1.shape), db1, dw2.view(w2.shape), db2, None


def dual_att_tail(F, S, fc1, fc2, pooled=None):
    """pooled: the global average pool of F [N, C] float32 if its producer already computed it (conv_bn_act(..., pool=...))"""
    return _DualAttTail.apply(F, S, fc1.weight, fc1.bias, fc2.weight, fc2.bias, pooled)


POOL_MODES = {"avg": 0, "max": 1, "avgmax": 2, "avgmaxc": 3}


class _GlobalPool(torch.autograd.Function):
    """adaptive_avgmax_pool2d (models/adaptive_avgmax_pool.py:19-40): global avg / max / avgmax / avgmaxc in one pass."""

    @staticmethod
    def forward(ctx, x, mode):
        x = nhwc(x)
        _check_dev(x)
        n, c, h, w = x.shape
        dev = x.device
        out = torch.empty(n, c * (2 if mode == 3 else 1), dtype=torch.float32, device=dev)
        arg = torch.empty(n, c, dtype=torch.int32, device=dev) if mode != 0 else None
        need = L.load().saunet_global_pool_workspace(n, h * w, c)
        ws = torch.empty(max(need, 4) // 4, dtype=torch.float32, device=dev)
        L.call("saunet_global_pool_forward", L.dtype_code(x), mode, x.data_ptr(), n, h * w, c, ld_of(x), out.data_ptr(), L.ptr(arg),
               ws.data_ptr(), need, L.stream())
        ctx.save_for_backward(arg if arg is not None else out.new_empty(0))
        ctx.cfg = (mode, n, c, h, w, x.dtype)
        return out.view(n, -1, 1, 1).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        mode, n, c, h, w, dtype = ctx.cfg
        dy = dy.reshape(n, -1).to(torch.float32).contiguous()
        dx = new_act(n, c, h, w, dtype, dy.device)
        L.call("saunet_global_pool_backward", L.dtype_code(dx), mode, dy.data_ptr(), arg.data_ptr() if mode != 0 else None, n, h * w, c,
               dx.data_ptr(), ld_of(dx), L.stream())
        return dx, None


def adaptive_avgmax_pool2d(x, pool_type="avg"):
    """[N,C,H,W] -> [N,C,1,1] ('avg', 'max', 'avgmax') or [N,2C,1,1] ('avgmaxc'); an unknown pool_type falls back to 'avg' with the
    reference's message (adaptive_avgmax_pool.py:35-37)."""
    if pool_type not in POOL_MODES:
        print("Invalid pool type %s specified. Defaulting to average pooling." % pool_type)
        pool_type = "avg"
    return _GlobalPool.apply(x, POOL_MODES[pool_type])


def _as_f32(t):
    t = nhwc(t)
    if t.dtype == torch.float32:
        return t
    n, c, h, w = t.shape
    return copy_channels(t, new_act(n, c, h, w, torch.float32, t.device))


class _DualLoss(torch.autograd.Function):
    """dice + weighted CE + BCE(edge) and the pixel_acc metrics from one pass (loss.py:149-159).
    The loss always runs on float32 copies of the (4+1)-channel heads, whatever the storage dtype."""

    @staticmethod
    def forward(ctx, logits, edge, seg_t, edge_t):
        ctx.in_dtypes = (logits.dtype, edge.dtype)
        logits = _as_f32(logits); edge = _as_f32(edge)
        n, c, h, w = logits.shape
        if c != 4:
            raise RuntimeError("DualLoss kernel is specialised for 4 classes, got %d" % c)
        P = n * h * w
        dev = logits.device
        seg_t = seg_t.to(device=dev, dtype=torch.int64).contiguous()
        edge_t = edge_t.to(device=dev, dtype=torch.float32).contiguous()
        if seg_t.numel() != P or edge_t.numel() != P:
            raise RuntimeError("DualLoss: target sizes %s / %s do not match logits %s" % (tuple(seg_t.shape), tuple(edge_t.shape), tuple(logits.shape)))
        sums = STATS.take(32, dev) if logits.is_cuda else torch.zeros(32, dtype=torch.float64, device=dev)
        L.call("saunet_dual_loss_forward", L.F32, logits.data_ptr(), ld_of(logits), edge.data_ptr(), seg_t.data_ptr(), edge_t.data_ptr(), P,
               sums.data_ptr(), L.stream())
        out = torch.empty(5, dtype=torch.float32, device=dev)
        L.call("saunet_dual_loss_finalize", sums.data_ptr(), P, out.data_ptr(), out[1:].data_ptr(), L.stream())
        ctx.save_for_backward(logits, edge, seg_t, edge_t, sums)
        loss, metrics = out[0], out[1:]
        ctx.mark_non_differentiable(metrics)
        return loss, metrics

    @staticmethod
    def backward(ctx, dloss, _dmetrics):
        logits, edge, seg_t, edge_t, sums = ctx.saved_tensors
        n, c, h, w = logits.shape
        dl = new_act(n, c, h, w, torch.float32, logits.device)
        de = new_act(n, 1, h, w, torch.float32, logits.device)
        dloss = dloss.to(torch.float32).contiguous()
        L.call("saunet_dual_loss_backward", L.F32, logits.data_ptr(), ld_of(logits), edge.data_ptr(), seg_t.data_ptr(),
               edge_t.data_ptr(), n * h * w, sums.data_ptr(), dloss.data_ptr(), dl.data_ptr(), ld_of(dl), de.data_ptr(), L.stream())
        if ctx.in_dtypes[0] != torch.float32:
            dl = copy_channels(dl, new_act(n, c, h, w, ctx.in_dtypes[0], dl.device))
        if ctx.in_dtypes[1] != torch.float32:
            de = copy_channels(de, new_act(n, 1, h, w, ctx.in_dtypes[1], de.device))
        return dl, de, None, None


def dual_loss(logits, edge, seg_t, edge_t):
    """-> (loss scalar tensor, metrics tensor [acc, j1, j2, j3])."""
    return _DualLoss.apply(logits, edge, seg_t, edge_t)


def softmax_argmax(logits, want_prob=True, want_label=True):
    """Inference head on the device: (softmax over the class dimension as float32 [N,C,H,W], argmax labels int64 [N,H,W]); no gradient.
    Replaces torch.softmax(...) + .argmax(1) of the reference's test / eval branches (models/models.py:96-109, train.py:47)."""
    _check_dev(logits)
    lg = nhwc(logits.detach())
    n, c, h, w = lg.shape
    prob = new_act(n, c, h, w, torch.float32, lg.device) if want_prob else None
    label = torch.empty((n, h, w), dtype=torch.int64, device=lg.device) if want_label else None
    L.call("saunet_softmax_argmax", L.dtype_code(lg), lg.data_ptr(), ld_of(lg), n * h * w, c, L.ptr(prob), ld_of(prob) if prob is not None else 0,
           L.ptr(label), L.stream())
    return prob, label


_METRIC_KIND = {torch.float32: 0, torch.bfloat16: 1, torch.int64: 2, torch.uint8: 3, torch.bool: 3}


def pixel_metrics(pred, label, num_class):
    """SegmentationModuleBase.pixel_acc (models/models.py:51-74) on the device: pred [N, C, H, W] class scores (float32 / bf16 / int64 / uint8,
    NCHW or channels_last memory), label [N, H, W] -> float32 [C]: acc over the labelled pixels, then the Jaccard of classes 1 .. C-1."""
    _check_dev(pred)
    if pred.dim() != 4 or pred.shape[1] != num_class:
        raise RuntimeError("pixel_metrics: pred must be [N, %d, H, W], got %s" % (num_class, tuple(pred.shape)))
    if pred.dtype not in _METRIC_KIND:
        raise RuntimeError("pixel_metrics: unsupported prediction dtype %s" % pred.dtype)
    pred = pred.detach()
    if not (pred.is_contiguous() or pred.is_contiguous(memory_format=_CL)):
        pred = pred.contiguous()
    n, c, h, w = pred.shape
    sn, sc, sh, sw = pred.stride()
    if sh != w * sw:                                    # one pixel stride must address the whole map
        pred = pred.contiguous(); sn, sc, sh, sw = pred.stride()
    lab = label.detach().to(device=pred.device, dtype=torch.int64).contiguous()
    if tuple(lab.shape) != (n, h, w):
        raise RuntimeError("pixel_metrics: label must be [N, H, W] = %s, got %s" % ((n, h, w), tuple(lab.shape)))
    counts = torch.zeros(2 + 3 * (c - 1), dtype=torch.int64, device=pred.device)
    out = torch.empty(c, dtype=torch.float32, device=pred.device)
    L.call("saunet_pixel_metrics", _METRIC_KIND[pred.dtype], pred.data_ptr(), sn, sc, sw, lab.data_ptr(), n, h * w, c, counts.data_ptr(), out.data_ptr(), L.stream())
    return out


def binary_jaccard(pred, label):
    """SegmentationModuleBase.jaccard (models/models.py:76-78) on the device -> 0-d float32 tensor."""
    _check_dev(pred)
    if pred.dtype not in _METRIC_KIND:
        raise RuntimeError("binary_jaccard: unsupported prediction dtype %s" % pred.dtype)
    p = pred.detach().contiguous()
    lab = label.detach().to(device=p.device, dtype=torch.int64).contiguous()
    if lab.numel() != p.numel():
        raise RuntimeError("binary_jaccard: %d predictions for %d labels" % (p.numel(), lab.numel()))
    sums = torch.zeros(3, dtype=torch.int64, device=p.device)
    out = torch.empty(1, dtype=torch.float32, device=p.device)
    L.call("saunet_binary_jaccard", _METRIC_KIND[p.dtype], p.data_ptr(), lab.data_ptr(), p.numel(), sums.data_ptr(), out.data_ptr(), L.stream())
    return out[0]


def canny(image, low=10, high=100, dtype=None):
    """image: float32 [N,3,H,W] (contiguous NCHW, as the loader delivers it) -> [N,1,H,W] in {0,255}."""
    _check_dev(image)
    img = image.detach().to(torch.float32).contiguous()
    n, c, h, w = img.shape
    if c != 3:
        raise RuntimeError("canny expects 3-channel input")
    dtype = dtype or image.dtype
    out = torch.empty((n, 1, h, w), dtype=dtype, device=img.device)
    work = torch.empty((n, 3, h, w), dtype=torch.int32, device=img.device)
    L.call("saunet_canny", L.BF16 if dtype == torch.bfloat16 else L.F32, img.data_ptr(), n, h, w, int(low), int(high), out.data_ptr(),
           work.data_ptr(), L.stream())
    return out


# ------------------------------------------------------------------------------------------------ relu
_CONST = {}


def _const_vec(c, device, value):
    key = (c, str(device), value)
    v = _CONST.get(key)
    if v is None:
        v = torch.full((c,), float(value), dtype=torch.float32, device=device)
        _CONST[key] = v
    return v


class _Relu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        y = affine_act(x, None, None, True)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = nhwc(dy)
        n, c, h, w = y.shape
        dev = y.device
        one, zero = _const_vec(c, dev, 1.0), _const_vec(c, dev, 0.0)
        dx = new_act(n, c, h, w, y.dtype, dev)
        dummy = torch.zeros(2 * c, dtype=torch.float64, device=dev)
        # identity-statistics BN backward in eval mode == dy * [y > 0]
        L.call("saunet_bn_backward_apply", L.dtype_code(y), dy.data_ptr(), ld_of(dy), y.data_ptr(), ld_of(y), None, 0, one.data_ptr(),
               zero.data_ptr(), zero.data_ptr(), one.data_ptr(), 1, dummy.data_ptr(), 1, 0, 1.0, 0, 0, dx.data_ptr(), ld_of(dx), None, 0, None, None,
               n * h * w, c, L.stream())
        return dx


def relu(x):
    return _Relu.apply(x)


# ------------------------------------------------------------------------------------------------ DenseNet
# A dense block's input can be produced IN PLACE: the producer (stem, transition) allocates the whole concat buffer of the block that consumes its
# output, writes its channels into the first slice and registers the buffer here; _DenseBlock.forward then adopts it instead of allocating a
# buffer and copying x0 into it (4 copy launches, ~0.1 ms per step).  Keyed by the slice's address; an entry lives from producer to consumer.
_DENSE_BASES = {}
# Hand-off of a transition's share of the linear BN backward to the dense block in front of it (round 5): the block's forward registers its
# concat buffer here when its backward will run the fused two-launch layers; the transition's backward then stores  d(buf) = scale * g  from
# its data-gradient epilogue (no apply pass over the C-channel tensor) and leaves the coefficient sums in _PENDING_AB[buf pointer]; the block's
# backward starts its running sums `ab` from them.  The registration travels ON THE OBJECT (an address can be reused): dense_block()
# tags the statistics tensor it returns with (buffer address, norm1 eps), transition() folds only when the tag names ITS buffer and its own
# eps equals the block's (the deferred correction uses the block's xhat rows).  _PENDING_AB is keyed by the forward buffer's address while
# both autograd nodes hold that buffer alive; the block's backward pops its entry, begin_step() refuses left-overs.
_LAST_FUSED_BLOCK = [None]
_PENDING_AB = {}


def reserve_dense_input(n, c, h, w, ctot, dtype, device):
    """-> the channel slice [:, :c] of a fresh [n, ctot, h, w] NHWC buffer, registered for adoption by the next _DenseBlock.forward"""
    base = new_act(n, ctot, h, w, dtype, device)
    view = base[:, :c]
    if len(_DENSE_BASES) > 16:
        _DENSE_BASES.clear()
    _DENSE_BASES[view.data_ptr()] = base
    return view


def _dense_bwd_fused_ok(buf, training, growth, bottleneck, c0, nl):
    """the block's backward will run saunet_dense_layer_backward_conv2 / _conv1 (bf16 storage, training-mode statistics, DenseNet-121 widths)"""
    # (the last three terms mirror dense_layer_check in csrc/dense_dgrad.hip: Cin <= 2048, pixel count below 2^31, 16-byte aligned rows)
    return bool(DENSE_BWD_FUSED and training and buf.is_cuda and buf.dtype == torch.bfloat16 and growth == 32 and bottleneck == 128 and c0 % 8 == 0
                and ld_of(buf) == buf.shape[1] and 0 < nl <= L.DENSE_LAYERS_MAX
                and c0 + growth * (nl - 1) <= 2048 and buf.shape[0] * buf.shape[2] * buf.shape[3] < (1 << 31) and buf.data_ptr() % 16 == 0)


class _DenseBlock(torch.autograd.Function):
    """One DenseNet block: L x [BN-ReLU-conv1x1(->128)-BN-ReLU-conv3x3(->32)] over a growing concat.

    The concat is ONE preallocated NHWC buffer; every layer's 32 new channels are written into their
    slice by the conv kernel itself, together with their batch statistics (computed once, reused by every
    later norm1 -- only gamma/beta differ per layer).  BN+ReLU are applied inside the consuming conv's
    operand load, so normalised tensors are never materialised.  torchvision _DenseBlock/_DenseLayer
    (third-party, used at /root/reference/models/models.py:271,306-313)."""

    @staticmethod
    def forward(ctx, x0, training, cfgs, *tensors):
        # tensors: per layer (n1w, n1b, c1w, n2w, n2b, c2w) then per layer (n1rm, n1rv, n2rm, n2rv)
        nl = len(cfgs)
        params, bufs = tensors[:6 * nl], tensors[6 * nl:]
        x0 = nhwc(x0)
        n, c0, h, w = x0.shape
        growth = params[5].shape[0]
        ctot = c0 + growth * nl
        dev = x0.device
        base = _DENSE_BASES.pop(x0.data_ptr(), None)
        if base is not None and tuple(base.shape) == (n, ctot, h, w) and base.dtype == x0.dtype and ld_of(x0) == ctot and base.data_ptr() == x0.data_ptr():
            buf = base                                  # x0 already is the first channel slice of this block's buffer
        else:
            buf = new_act(n, ctot, h, w, x0.dtype, dev)
            copy_channels(x0, buf[:, :c0])
        stats = new_stats(ctot, dev)
        count = n * h * w
        if training:
            bn_stats(buf[:, :c0], stats[:, :, :c0])
        saved = []
        nl_loop = nl
        # training on the GPU: BatchNorm coefficients are derived inside the consuming convolution (saunet_conv2d_forward_bnpro) -- two
        # launches per layer instead of four; the per-channel normalisation of the concat channels (xh rows: xs, xt, mean, invstd, var) is
        # published by the first kernel that needs it and reused by every later norm1
        bnpro = training and DENSE_BNPRO and buf.is_cuda and nl_loop > 0
        xh = GRADS.take(5 * ctot, dev).view(5, ctot) if buf.is_cuda else torch.zeros(5, ctot, dtype=torch.float32, device=dev)
        if bnpro:
            PACKS.generation += 1                  # running statistics change through raw pointers (as in bn_finalize)
            L.call("saunet_bn_xhat", c0, stats[0, 0].data_ptr(), stats[0, 1].data_ptr(), stats.shape[0], stats.stride(0), float(count), float(cfgs[0][1]),
                   xh.data_ptr(), xh.stride(0), L.stream())
            for l in range(nl_loop):
                n1w, n1b, c1w, n2w, n2b, c2w = params[6 * l:6 * l + 6]
                n1rm, n1rv, n2rm, n2rv = bufs[4 * l:4 * l + 4]
                mom, eps, mom2, eps2 = cfgs[l]
                cin = c0 + growth * l
                p1, p2 = BNParams(cin, dev), BNParams(c1w.shape[0], dev)
                st2 = new_stats(c1w.shape[0], dev)
                z1 = conv_forward_bnpro(buf[:, :cin], c1w, 1, 0, stats, count, cin - growth if l > 0 else cin, xh, n1w, n1b, n1rm, n1rv, mom, eps,
                                        p1.buf, stats=st2)
                conv_forward_bnpro(z1, c2w, 1, 1, st2, count, 0, None, n2w, n2b, n2rm, n2rv, mom2, eps2, p2.buf, out=buf[:, cin:cin + growth],
                                   stats=stats[:, :, cin:cin + growth])
                saved += [z1, p1.buf, p2.buf]
            nl_loop = 0
        for l in range(nl_loop):
            n1w, n1b, c1w, n2w, n2b, c2w = params[6 * l:6 * l + 6]
            n1rm, n1rv, n2rm, n2rv = bufs[4 * l:4 * l + 4]
            mom, eps, mom2, eps2 = cfgs[l]
            cin = c0 + growth * l
            p1 = bn_finalize(stats[:, :, :cin] if training else None, count, n1w, n1b, n1rm, n1rv, mom, eps, training)
            st2 = new_stats(c1w.shape[0], dev) if training else None
            z1 = conv_forward_raw(buf[:, :cin], c1w, None, 1, 0, pro=(p1.scale, p1.shift, True), stats=st2)
            p2 = bn_finalize(st2, count, n2w, n2b, n2rm, n2rv, mom2, eps2, training)
            conv_forward_raw(z1, c2w, None, 1, 1, pro=(p2.scale, p2.shift, True), out=buf[:, cin:cin + growth],
                             stats=stats[:, :, cin:cin + growth] if training else None)
            saved += [z1, p1.buf, p2.buf]
        # xhat = x*xs + xt for every concat channel (gamma=1, beta=0): what the deferred backward correction needs
        if training and not bnpro:
            one, zero = _const_vec(ctot, dev, 1.0), _const_vec(ctot, dev, 0.0)
            L.call("saunet_bn_finalize", ctot, stats[0, 0].data_ptr(), stats[0, 1].data_ptr(), stats.shape[0], stats.stride(0), float(count),
                   None, one.data_ptr(), zero.data_ptr(),
                   float(cfgs[0][1]), 0.0, None, None, xh[0].data_ptr(), xh[1].data_ptr(), None, None, 1, L.stream())
        if _dense_bwd_fused_ok(buf, training, growth, params[2].shape[0] if nl else 0, c0, nl):
            _LAST_FUSED_BLOCK[0] = (buf.data_ptr(), float(cfgs[0][1]))
            if bnpro:
                # the xhat rows of a concat channel are published by the first conv1 that normalises it -- nobody inside the block does that for
                # the LAST layer's 32 channels, but a transition that folds its BatchNorm backward into this block needs them for its correction
                lo = ctot - growth
                L.call("saunet_bn_xhat", growth, stats[0, 0, lo:].data_ptr(), stats[0, 1, lo:].data_ptr(), stats.shape[0], stats.stride(0), float(count),
                       float(cfgs[0][1]), xh[0, lo:].data_ptr(), xh.stride(0), L.stream())
        ctx.save_for_backward(buf, xh, *params, *saved)
        ctx.meta = (nl, c0, growth, count, training)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)           # no zero-filled "gradient" of the statistics tensor in backward
        return buf, stats

    @staticmethod
    def backward(ctx, dbuf, _dstats):
        nl, c0, growth, count, training = ctx.meta
        t = ctx.saved_tensors
        buf, xh, params, saved = t[0], t[1], t[2:2 + 6 * nl], t[2 + 6 * nl:]
        n, ctot, h, w = buf.shape
        dev = buf.device
        dbuf = nhwc(dbuf)
        if not dbuf.is_contiguous(memory_format=_CL) or dbuf.shape[1] != ctot:
            dbuf = dbuf.contiguous(memory_format=_CL)
        P = n * h * w
        dt = L.dtype_code(buf)
        # "linear" BN backward: every consumer adds s*g into dbuf from its dgrad epilogue; the -(A + B*xhat) terms
        # are accumulated per channel and applied ONCE per 32-channel chunk right before that chunk is consumed
        # two (A, B) pairs: saunet_bn_backward_coeff_correct reads one and writes the other (ping-pong)
        ABB = GRADS.take(4 * ctot, dev).view(2, 2, ctot) if buf.is_cuda else torch.zeros(2, 2, ctot, dtype=torch.float32, device=dev)
        AB = ABB[0]
        merged = training and DENSE_COEFF_CORRECT and buf.is_cuda      # coeff of layer l + correct of chunk l-1 in one launch

        def correct(lo, hi):
            if training:
                d, x = dbuf[:, lo:hi], buf[:, lo:hi]
                L.call("saunet_bn_backward_correct", dt, d.data_ptr(), ld_of(d), x.data_ptr(), ld_of(x), AB[0, lo:hi].data_ptr(),
                       AB[1, lo:hi].data_ptr(), xh[0, lo:hi].data_ptr(), xh[1, lo:hi].data_ptr(), P, hi - lo, L.stream())

        grads = [None] * (6 * nl)
        pending_ab = _PENDING_AB.pop(buf.data_ptr(), None)          # a transition behind this block left its coefficient sums (see _Transition.backward)
        if _dense_bwd_fused_ok(buf, training, growth, params[2].shape[0] if nl else 0, c0, nl) and ld_of(dbuf) == ctot:
            return _DenseBlock._backward_fused(ctx, buf, dbuf, xh, params, saved, grads, pending_ab)
        if pending_ab is not None:      # (the switch was flipped between forward and backward: settle the transition's correction at once)
            for lo in range(0, ctot, 256):
                hi = min(lo + 256, ctot)
                dsl, xsl = dbuf[:, lo:hi], buf[:, lo:hi]
                L.call("saunet_bn_backward_correct_ab", dt, dsl.data_ptr(), ld_of(dsl), xsl.data_ptr(), ld_of(xsl), dsl.data_ptr(), ld_of(dsl),
                       pending_ab[0, 0, lo:hi].data_ptr(), pending_ab.shape[0], pending_ab.stride(0), ctot, float(count), xh[0, lo:hi].data_ptr(),
                       xh[1, lo:hi].data_ptr(), P, hi - lo, L.stream())
        pend = [] if (buf.is_cuda and DENSE_WGRAD_BATCH_REDUCE) else None   # the 2 x L partial-gradient reductions: one launch
        # default: both weight gradients of every layer are deferred to the end of the block and issued as two GROUPED launches (every layer's
        # dz1 / corrected gradient chunk stays alive until then: 24 x 8 MB at block 3 -- nothing against 288 GB)
        grouped = buf.is_cuda and DENSE_WGRAD_GROUPED
        wg1, wg2 = [], []
        for l in reversed(range(nl)):
            n1w, n1b, c1w, n2w, n2b, c2w = params[6 * l:6 * l + 6]
            z1, p1b, p2b = saved[3 * l:3 * l + 3]
            p1 = BNParams.__new__(BNParams); p1.buf = p1b
            p2 = BNParams.__new__(BNParams); p2.buf = p2b
            cin = c0 + growth * l
            xin = buf[:, :cin]
            if not merged:
                correct(cin, cin + growth)       # (merged: done by the previous iteration's coeff launch; the last chunk has no consumer in the block)
            dz2 = dbuf[:, cin:cin + growth]
            if grouped:
                wg2.append((l, z1, dz2, c2w, (p2.scale, p2.shift)))
                dw2 = None
            else:
                dw2 = conv_wgrad_raw(z1, dz2, c2w, 1, 1, pro=(p2.scale, p2.shift, True), pending=pend)
            s2 = new_stats(z1.shape[1], dev)
            da2 = conv_dgrad_raw(dz2, c2w, z1.shape, 1, 1, bn_epi=(z1, p2, True, s2))
            dz1, _, dg2, db2 = bn_backward(da2, z1, p2, True, count, training, dx=da2, presums=s2)
            if grouped:
                wg1.append((l, xin, dz1, c1w, (p1.scale, p1.shift)))
                dw1 = None
            else:
                dw1 = conv_wgrad_raw(xin, dz1, c1w, 1, 0, pro=(p1.scale, p1.shift, True), pending=pend)
            s1 = new_stats(cin, dev)
            conv_dgrad_raw(dz1, c1w, (n, cin, h, w), 1, 0, out=dbuf[:, :cin], bn_epi=(xin, p1, True, s1, True))
            dgb = torch.empty(2, cin, dtype=torch.float32, device=dev)
            if merged and l > 0:
                nxt = ABB[1] if AB.data_ptr() == ABB[0].data_ptr() else ABB[0]
                lo = cin - growth
                d_, x_ = dbuf[:, lo:cin], buf[:, lo:cin]
                L.call("saunet_bn_backward_coeff_correct", dt, cin, s1.data_ptr(), s1.shape[0], s1.stride(0), float(count), p1.scale.data_ptr(),
                       AB[0].data_ptr(), AB[1].data_ptr(), nxt[0].data_ptr(), nxt[1].data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(),
                       d_.data_ptr(), ld_of(d_), x_.data_ptr(), ld_of(x_), lo, cin, xh[0].data_ptr(), xh[1].data_ptr(), P, L.stream())
                AB = nxt
            else:
                L.call("saunet_bn_backward_coeff", cin, s1.data_ptr(), s1.shape[0], s1.stride(0), float(count), p1.scale.data_ptr(), AB[0].data_ptr(), AB[1].data_ptr(),
                       dgb[0].data_ptr(), dgb[1].data_ptr(), 1 if training else 0, L.stream())
            grads[6 * l:6 * l + 6] = [dgb[0], dgb[1], dw1, dg2, db2, dw2]
        if grouped:
            for plist, slot, ks, pd in ((wg2, 5, 3, 1), (wg1, 2, 1, 0)):
                dws = conv_wgrad_grouped([(x_, dy_, w_, pro_) for (_l, x_, dy_, w_, pro_) in plist], ks, pd, True)
                if dws is None:        # geometry off the tiled kernels (maps that are not multiples of the 16-pixel tile): per-layer launches
                    dws = [conv_wgrad_raw(x_, dy_, w_, 1, pd, pro=(pro_[0], pro_[1], True), pending=pend) for (_l, x_, dy_, w_, pro_) in plist]
                for (l_, *_rest), dw_ in zip(plist, dws):
                    grads[6 * l_ + slot] = dw_
        if pend:
            flush_wgrad_reductions(pend)
        correct(0, c0)
        dx0 = dbuf[:, :c0] if ctx.needs_input_grad[0] else None
        return (dx0, None, None) + tuple(grads) + (None,) * (4 * nl)


    @staticmethod
    def _backward_fused(ctx, buf, dbuf, xh, params, saved, grads, pending_ab=None):
        """Round 5: two launches per layer (saunet_dense_layer_backward_conv2 / _conv1) instead of four.  The BatchNorm-backward apply of
        norm2 happens in the conv1 data gradient's operand load, the deferred chunk correction of the linear BN1 backward in the conv2
        data gradient's (a separate streaming pass only on the maps that run the LDS-DMA staged conv2 kernel), and the per-layer
        coefficient launch is gone: the conv1 kernel's epilogue adds scale * (sum g, sum g*xhat) to the block's running float64 sums `ab`."""
        nl, c0, growth, count, training = ctx.meta
        n, ctot, h, w = buf.shape
        dev = buf.device
        P = n * h * w
        st = L.stream()
        ab = pending_ab if pending_ab is not None else new_stats(ctot, dev)
        g = new_act(n, 128, h, w, buf.dtype, dev)                   # scratch between the two launches of a layer
        wg1, wg2 = [], []
        bl = L.DenseBn1List()
        bl.count, bl.replicas = nl, STAT_R
        keep = []
        def descriptor():
            d = L.DenseLayerBwd()
            d.N, d.H, d.W, d.Ctot = n, h, w, ctot
            d.buf, d.dbuf, d.xhat, d.ld_xhat = buf.data_ptr(), dbuf.data_ptr(), xh.data_ptr(), xh.stride(0)
            d.ab, d.ab_replicas, d.ab_rstride, d.count = ab.data_ptr(), ab.shape[0], ab.stride(0), float(count)
            d.g = g.data_ptr()
            return d

        def fill(d, l):
            """descriptor of layer l + its bookkeeping (weight-gradient work lists, BatchNorm gradient slots)"""
            n1w, n1b, c1w, n2w, n2b, c2w = params[6 * l:6 * l + 6]
            z1, p1b, p2b = saved[3 * l:3 * l + 3]
            cin = c0 + growth * l
            dz1 = new_act(n, 128, h, w, buf.dtype, dev)
            # the last chunk has no consumer inside the block: nothing to correct unless a transition's sums are pending
            dz2 = new_act(n, growth, h, w, buf.dtype, dev) if (l + 1 < nl or pending_ab is not None) else None
            s2, s1 = new_stats(128, dev), new_stats(cin, dev)
            dgb2 = torch.empty(2, 128, dtype=torch.float32, device=dev)
            dgb1 = torch.empty(2, cin, dtype=torch.float32, device=dev)
            w2p, w1p = PACKS.get(c2w, L.PACK_DGRAD, buf.dtype), PACKS.get(c1w, L.PACK_DGRAD, buf.dtype)
            d.Cin, d.c_begin = cin, 0
            d.z1, d.dz1, d.dz2 = z1.data_ptr(), dz1.data_ptr(), (dz2.data_ptr() if dz2 is not None else None)
            d.w2_dgrad, d.w1_dgrad, d.p1, d.p2 = w2p.data_ptr(), w1p.data_ptr(), p1b.data_ptr(), p2b.data_ptr()
            d.sums2, d.sums2_replicas, d.sums2_rstride = s2.data_ptr(), s2.shape[0], s2.stride(0)
            d.sums1, d.sums1_replicas, d.sums1_rstride = s1.data_ptr(), s1.shape[0], s1.stride(0)
            d.dgamma2, d.dbeta2 = dgb2[0].data_ptr(), dgb2[1].data_ptr()
            wg2.append((l, z1, dz2 if dz2 is not None else dbuf[:, cin:cin + growth], c2w, (p2b[0], p2b[1])))
            wg1.append((l, buf[:, :cin], dz1, c1w, (p1b[0], p1b[1])))
            bl.sums[l], bl.rstride[l], bl.cin[l] = s1.data_ptr(), s1.stride(0), cin
            bl.dgamma[l], bl.dbeta[l] = dgb1[0].data_ptr(), dgb1[1].data_ptr()
            keep.extend([s1, s2, w2p, w1p])
            grads[6 * l], grads[6 * l + 1], grads[6 * l + 3], grads[6 * l + 4] = dgb1[0], dgb1[1], dgb2[0], dgb2[1]

        d, d_lo = descriptor(), descriptor()
        lib = L.load()
        l = nl - 1
        while l >= 0:
            fill(d, l)
            L.call("saunet_dense_layer_backward_conv2", C.byref(d), st)
            if DENSE_BWD_PAIRS and l >= 1 and growth == 32 and lib.saunet_dense_layer_backward_pair_supported(C.byref(d)) == 1:
                # layers l and l - 1 share one pass over buf / dbuf (saunet_hip.h: layer pairs): the top chunk of layer l first -- it is what
                # layer l - 1's conv2 data gradient reads -- then both layers' contributions to the channels below it
                d.c_begin = d.Cin - growth
                L.call("saunet_dense_layer_backward_conv1", C.byref(d), st)
                fill(d_lo, l - 1)
                L.call("saunet_dense_layer_backward_conv2", C.byref(d_lo), st)
                L.call("saunet_dense_layer_backward_conv1_pair", C.byref(d), C.byref(d_lo), st)
                l -= 2
            else:
                L.call("saunet_dense_layer_backward_conv1", C.byref(d), st)
                l -= 1
        L.call("saunet_dense_bn1_grads", C.byref(bl), st)
        pend = [] if DENSE_WGRAD_BATCH_REDUCE else None
        for plist, slot, ks, pd in ((wg2, 5, 3, 1), (wg1, 2, 1, 0)):
            dws = conv_wgrad_grouped([(x_, dy_, w_, pro_) for (_l, x_, dy_, w_, pro_) in plist], ks, pd, True) if DENSE_WGRAD_GROUPED else None
            if dws is None:        # geometry off the tiled kernels (maps that are not multiples of the 16-pixel tile): per-layer launches
                dws = [conv_wgrad_raw(x_, dy_, w_, 1, pd, pro=(pro_[0], pro_[1], True), pending=pend) for (_l, x_, dy_, w_, pro_) in plist]
            for (l_, *_rest), dw_ in zip(plist, dws):
                grads[6 * l_ + slot] = dw_
        if pend:
            flush_wgrad_reductions(pend)
        dt = L.dtype_code(buf)
        for lo in range(0, c0, 256):      # the block's input channels: corrected in place, they are the gradient handed upstream
            hi = min(lo + 256, c0)
            dsl, xsl = dbuf[:, lo:hi], buf[:, lo:hi]
            L.call("saunet_bn_backward_correct_ab", dt, dsl.data_ptr(), ctot, xsl.data_ptr(), ctot, dsl.data_ptr(), ctot, ab[0, 0, lo:hi].data_ptr(),
                   ab.shape[0], ab.stride(0), ctot, float(count), xh[0, lo:hi].data_ptr(), xh[1, lo:hi].data_ptr(), P, hi - lo, st)
        dx0 = dbuf[:, :c0] if ctx.needs_input_grad[0] else None
        return (dx0, None, None) + tuple(grads) + (None,) * (4 * nl)


def dense_block_infer(x0, layers):
    """Inference form of a DenseNet block: per layer ONE prologue'd 1x1 conv whose weights carry norm2 (w' = w * s2, bias = t2, ReLU in the
    epilogue -> the 3x3 conv needs no prologue) and one plain 3x3 conv writing its 32 channels into the concat buffer.  norm1 cannot be
    folded (a ReLU sits between it and conv1 and every layer has its own gamma/beta over the shared concat channels): it stays the operand
    prologue, with its eval-mode coefficients cached (bn_finalize)."""
    x0 = nhwc(x0)
    n, c0, h, w = x0.shape
    growth = layers[0].conv2.weight.shape[0]
    ctot = c0 + growth * len(layers)
    buf = new_act(n, ctot, h, w, x0.dtype, x0.device)
    copy_channels(x0, buf[:, :c0])
    cin = c0
    for m in layers:
        p1 = bn_finalize(None, 1, m.norm1.weight, m.norm1.bias, m.norm1.running_mean, m.norm1.running_var, m.norm1.momentum, m.norm1.eps, False)
        w1p, b1 = folded_conv_bn(m.conv1.weight, None, m.norm2, False, x0.dtype)
        a2 = conv_forward_raw(buf[:, :cin], m.conv1.weight, b1, 1, 0, pro=(p1.scale, p1.shift, True), act_relu=True, packed=w1p)
        conv_forward_raw(a2, m.conv2.weight, None, 1, 1, out=buf[:, cin:cin + growth])
        cin += growth
    return buf


def dense_block(x0, layers, training):
    """layers: list of modules with norm1, conv1, norm2, conv2.  Returns (concat buffer, its channel statistics)."""
    if not training and not torch.is_grad_enabled() and x0.is_cuda:
        return dense_block_infer(x0, layers), None
    eps = {float(m.norm1.eps) for m in layers}       # (norm2 of a layer normalises its own 128 channels: its eps / momentum travel per layer)
    if len(eps) > 1:
        # the block shares one set of normalised-input rows (invstd computed once per concat channel) between all its norm1 layers, forward and
        # backward: that is only the BatchNorm of every layer when they agree on eps (torchvision's _DenseLayer always does)
        raise RuntimeError("dense_block: the norm1 layers of one dense block must share eps, got %s" % sorted(eps))
    params, bufs, cfgs = [], [], []
    for m in layers:
        params += [m.norm1.weight, m.norm1.bias, m.conv1.weight, m.norm2.weight, m.norm2.bias, m.conv2.weight]
        bufs += [m.norm1.running_mean, m.norm1.running_var, m.norm2.running_mean, m.norm2.running_var]
        cfgs.append((m.norm1.momentum, m.norm1.eps, m.norm2.momentum, m.norm2.eps))
        _bump(m.norm1); _bump(m.norm2)
    _LAST_FUSED_BLOCK[0] = None
    buf, stats = _DenseBlock.apply(x0, training, tuple(cfgs), *params, *bufs)
    if _LAST_FUSED_BLOCK[0] is not None and stats is not None:
        stats._saunet_fused_block = _LAST_FUSED_BLOCK[0]      # read by transition(): (address of the block's concat buffer, its norm1 eps)
        _LAST_FUSED_BLOCK[0] = None
    return buf, stats


# Transition with the average pool in front of the 1x1 convolution (round 6): the two commute, so conv, data gradient and weight gradient run on
# a quarter of the pixels and the full-resolution C/2 tensor never exists (saunet_bn_relu_avgpool2 / _backward).  SAUNET_TRANSITION_POOL_FIRST=0
# restores conv -> pool (A/B, tests).
TRANSITION_POOL_FIRST = os.environ.get("SAUNET_TRANSITION_POOL_FIRST", "1") != "0"


def _pool_first_ok(buf):
    epc = 8 if buf.dtype == torch.bfloat16 else 4
    n, c, h, w = buf.shape
    return (TRANSITION_POOL_FIRST and buf.is_cuda and buf.dtype in (torch.bfloat16, torch.float32) and h % 2 == 0 and w % 2 == 0 and c % epc == 0
            and c <= 2048 and ld_of(buf) % epc == 0 and buf.data_ptr() % 16 == 0)


class _Transition(torch.autograd.Function):
    """BN-ReLU-conv1x1(C -> C/2)-AvgPool2 over a dense block's concat buffer (statistics already known)."""

    @staticmethod
    def forward(ctx, buf, stats, gamma, beta, rmean, rvar, weight, momentum, eps, training, reserve=0, fold=False):
        buf = nhwc(buf)
        n, c, h, w = buf.shape
        count = n * h * w
        p = bn_finalize(stats if training else None, count, gamma, beta, rmean, rvar, momentum, eps, training)
        co = weight.shape[0]
        if reserve and reserve > co:            # the pooled output is the first slice of the next dense block's concat buffer
            y = reserve_dense_input(n, co, h // 2, w // 2, reserve, buf.dtype, buf.device)
        else:
            y = new_act(n, co, h // 2, w // 2, buf.dtype, buf.device)
        if _pool_first_ok(buf):
            a = new_act(n, c, h // 2, w // 2, buf.dtype, buf.device)
            L.call("saunet_bn_relu_avgpool2", L.dtype_code(buf), buf.data_ptr(), ld_of(buf), p.scale.data_ptr(), p.shift.data_ptr(), a.data_ptr(), ld_of(a),
                   n, h, w, c, L.stream())
            conv_forward_raw(a, weight, None, 1, 0, out=y)
            ctx.save_for_backward(buf, weight, p.buf, a)
            ctx.meta = (count, training, bool(fold), True)
            return y
        z = conv_forward_raw(buf, weight, None, 1, 0, pro=(p.scale, p.shift, True))
        L.call("saunet_pool2x2_forward", L.dtype_code(z), 0, z.data_ptr(), n, h, w, z.shape[1], ld_of(z), y.data_ptr(), ld_of(y), L.stream())
        ctx.save_for_backward(buf, weight, p.buf, buf.new_empty(0))
        ctx.meta = (count, training, bool(fold), False)
        return y

    @staticmethod
    def backward(ctx, dy):
        buf, weight, pbuf, a = ctx.saved_tensors
        count, training, fold, pool_first = ctx.meta
        p = BNParams.__new__(BNParams); p.buf = pbuf
        dy = nhwc(dy)
        n, c, h, w = buf.shape
        co = weight.shape[0]
        if pool_first:
            dw = conv_wgrad_raw(a, dy, weight, 1, 0)                       # quarter-resolution operands, no prologue: `a` is the pooled activation
            dap = conv_dgrad_raw(dy, weight, a.shape, 1, 0)
            sb = new_stats(c, buf.device)
            folded = bool(fold and training and DENSE_BWD_FUSED and DENSE_TRANSITION_FOLD)
            da = new_act(n, c, h, w, buf.dtype, buf.device)
            L.call("saunet_bn_relu_avgpool2_backward", L.dtype_code(buf), dap.data_ptr(), ld_of(dap), buf.data_ptr(), ld_of(buf), p.scale.data_ptr(),
                   p.shift.data_ptr(), p.mean.data_ptr(), p.invstd.data_ptr(), 1 if folded else 0, da.data_ptr(), ld_of(da),
                   sb.data_ptr(), sb.shape[0], sb.stride(0), n, h, w, c, L.stream())
            if folded:
                ab = new_stats(c, buf.device)
                dgb = torch.empty(2, c, dtype=torch.float32, device=buf.device)
                L.call("saunet_bn_backward_coeff_ab", c, sb.data_ptr(), sb.shape[0], sb.stride(0), p.scale.data_ptr(), ab.data_ptr(), c,
                       dgb[0].data_ptr(), dgb[1].data_ptr(), L.stream())
                _PENDING_AB[buf.data_ptr()] = ab
                return da, None, dgb[0], dgb[1], None, None, dw, None, None, None, None, None
            dbuf, _, dg, db = bn_backward(da, buf, p, True, count, training, dx=da, presums=sb)
            return dbuf, None, dg, db, None, None, dw, None, None, None, None, None
        dz = new_act(n, co, h, w, dy.dtype, dy.device)
        L.call("saunet_pool2x2_backward", L.dtype_code(dy), 0, None, dy.data_ptr(), n, h, w, co, 0, ld_of(dy), dz.data_ptr(), ld_of(dz), 0, L.stream())
        dw = conv_wgrad_raw(buf, dz, weight, 1, 0, pro=(p.scale, p.shift, True))
        sb = new_stats(c, buf.device)
        if fold and training and DENSE_BWD_FUSED and DENSE_TRANSITION_FOLD:
            # linear form: d(buf) = scale * g  straight from the data gradient's epilogue; the -(A + B * xhat) half is applied by the dense block's
            # backward with every other consumer's share (chunk by chunk, as the gradient is consumed) -- no apply pass over the C-channel tensor
            da = conv_dgrad_raw(dz, weight, buf.shape, 1, 0, bn_epi=(buf, p, True, sb, 2))
            ab = new_stats(c, buf.device)
            dgb = torch.empty(2, c, dtype=torch.float32, device=buf.device)
            L.call("saunet_bn_backward_coeff_ab", c, sb.data_ptr(), sb.shape[0], sb.stride(0), p.scale.data_ptr(), ab.data_ptr(), c,
                   dgb[0].data_ptr(), dgb[1].data_ptr(), L.stream())
            _PENDING_AB[buf.data_ptr()] = ab
            return da, None, dgb[0], dgb[1], None, None, dw, None, None, None, None, None
        da = conv_dgrad_raw(dz, weight, buf.shape, 1, 0, bn_epi=(buf, p, True, sb))
        dbuf, _, dg, db = bn_backward(da, buf, p, True, count, training, dx=da, presums=sb)
        return dbuf, None, dg, db, None, None, dw, None, None, None, None, None


def transition(buf, stats, m, training, reserve=0):
    """reserve: total channel count of the dense block that consumes the result (0: a plain tensor)"""
    _bump(m.norm)
    # fold the BatchNorm backward into the dense block in front (section 4 of DESIGN.md) only when `stats` is the object that block's forward
    # returned for exactly this buffer and the two normalisations agree on eps (the block's xhat rows serve the deferred correction)
    tag = getattr(stats, "_saunet_fused_block", None)
    fold = bool(tag is not None and tag[0] == nhwc(buf).data_ptr() and tag[1] == float(m.norm.eps))
    return _Transition.apply(buf, stats, m.norm.weight, m.norm.bias, m.norm.running_mean, m.norm.running_var, m.conv.weight,
                             m.norm.momentum, m.norm.eps, training, int(reserve), fold)


def mask_to_edges(seg, num_classes=3):
    """seg: int64 [N,H,W] on the device -> float32 [N,1,H,W] edge ground truth (ac17_dataloader.py:231-258 on the GPU)."""
    _check_dev(seg)
    seg = seg.to(torch.int64).contiguous()
    n, h, w = seg.shape
    out = torch.empty((n, 1, h, w), dtype=torch.float32, device=seg.device)
    L.call("saunet_mask_to_edges", seg.data_ptr(), n, h, w, int(num_classes), out.data_ptr(), L.stream())
    return out
