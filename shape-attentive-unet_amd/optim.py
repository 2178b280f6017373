"""Optimiser side of the hot path: parameter grouping, cosine schedule and fused multi-tensor updates.

Mirrors /root/reference/train.py:166-216 (group_weight, create_optimizers, adjust_learning_rate) and
/root/reference/radam.py:15-78 (RAdam).  Updates run as ONE kernel launch per <=96 tensors
(csrc/optim.hip); hyper-parameters live in a device array so a captured hipGraph can be replayed after
only that array is rewritten.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from . import lib as L
from . import functional as HF


def group_weight(module):
    """decay: weights of every Linear / _ConvNd; no decay: their biases and all BatchNorm gamma/beta."""
    group_decay, group_no_decay = [], []
    seen = set()
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.modules.conv._ConvNd)):
            for p, dst in ((m.weight, group_decay), (m.bias, group_no_decay)):
                if p is not None and id(p) not in seen:
                    seen.add(id(p)); dst.append(p)
        elif isinstance(m, nn.modules.batchnorm._BatchNorm):
            for p in (m.weight, m.bias):
                if p is not None and id(p) not in seen:
                    seen.add(id(p)); group_no_decay.append(p)
    return [dict(params=group_decay), dict(params=group_no_decay, weight_decay=0.0)]


def cosine_lr(lr0, epoch, num_epoch):
    return lr0 * 0.5 * (1 + math.cos(3.14159 * epoch / num_epoch))  # the reference's constant, train.py:211


def adjust_learning_rate(optimizers, cur_iter, lr0, num_epoch):
    lr = cosine_lr(lr0, cur_iter, num_epoch)
    for opt in optimizers:
        for g in opt.param_groups:
            g["lr"] = lr
    return lr


def _tensor_lists(cols):
    """cols: list of equally long lists of tensors -> list of TensorList structs (<=96 tensors each)."""
    n = len(cols[0])
    out = []
    for s in range(0, n, 96):
        tl = L.TensorList()
        cnt = min(96, n - s)
        tl.count = cnt
        for k, col in enumerate(cols):
            for i in range(cnt):
                tl.ptrs[k][i] = col[s + i].data_ptr()
        for i in range(cnt):
            tl.numel[i] = cols[0][s + i].numel()
        out.append(tl)
    return out


class _FusedBase(torch.optim.Optimizer):
    """Private bookkeeping (device hyper-parameter arrays, last uploaded values) lives in ``self._priv`` keyed by the
    group index -- never in ``param_groups`` -- so ``state_dict()`` holds only what torch.optim would put there."""
    grad_scale = 1.0

    def _p(self, group):
        priv = self.__dict__.setdefault("_priv", {})
        for i, g in enumerate(self.param_groups):
            if g is group:
                return priv.setdefault(i, {})
        raise KeyError("unknown param group")

    def _hyper(self, group, n=8):
        pv = self._p(group)
        h = pv.get("hyper")
        if h is None or h.device != group["params"][0].device:
            h = torch.zeros(n, dtype=torch.float32, device=group["params"][0].device)
            pv["hyper"] = h
        return h

    def _upload(self, group, values):
        """Hyper-parameters go to the device array with an ASYNCHRONOUS copy from pinned memory, and only when they changed:
        a pageable-memory copy_ blocks the host until the stream has drained, i.e. once per step in eager mode."""
        values = tuple(float(v) for v in values)
        h = self._hyper(group)
        pv = self._p(group)
        if pv.get("vals") == values and pv.get("dev") == h.data_ptr():
            return
        host = torch.tensor(values, dtype=torch.float32)
        if h.is_cuda:
            host = host.pin_memory()          # the caching host allocator keeps the block alive until the copy has run
        h.copy_(host, non_blocking=True)
        pv["vals"], pv["dev"] = values, h.data_ptr()

    # ---- hipGraph protocol (saunet_amd.graph.GraphedStep): a captured step() launches the update kernels with whatever the device
    # hyper-parameter array holds at replay time, so the host refreshes that array before every replay and keeps the step counters
    def pre_replay(self):
        self.upload_hyper()

    def post_replay(self, n=1):
        for st in self.state.values():
            if "step" in st:
                st["step"] += n

    def capture_rollback(self):
        """stream capture ran step()'s host code once without executing anything on the device"""
        self.post_replay(-1)

    def _live(self, group):
        ps = [p for p in group["params"] if p.grad is not None]
        for p in ps:
            if not p.is_cuda:
                raise RuntimeError("fused optimisers run on the GPU only (no CPU fallback)")
            if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or not p.is_contiguous():
                raise RuntimeError("fused optimisers need contiguous float32 params/grads")
        return ps


class FusedSGD(_FusedBase):
    """torch.optim.SGD(momentum, weight_decay, nesterov=False) semantics (train.py:190-196)."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    def upload_hyper(self):
        for g in self.param_groups:
            # the kernel's "first step" flag (buf = grad, do not read the buffer) stays 0: momentum buffers are created ZERO-filled, for
            # which buf = momentum*0 + grad is the same value -- and a hyper array captured by a hipGraph must not carry a per-step flag
            self._upload(g, [g["lr"], g["momentum"], g["weight_decay"], 0.0, self.grad_scale, 0, 0, 0])

    @torch.no_grad()
    def step(self, closure=None, upload=True):
        if upload:
            self.upload_hyper()
        for g in self.param_groups:
            ps = self._live(g)
            if not ps:
                continue
            bufs = []
            for p in ps:
                st = self.state[p]
                if "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                bufs.append(st["momentum_buffer"])
            for tl in _tensor_lists([ps, [p.grad for p in ps], bufs]):
                L.call("saunet_sgd_step", C.byref(tl), self._hyper(g).data_ptr(), L.stream())
        HF.PACKS.invalidate()


class FusedRAdam(_FusedBase):
    """RAdam exactly as /root/reference/radam.py:15-78 (train.sh's optimiser); weight decay is NOT passed by the
    reference's create_optimizers (train.py:202-206), so the default is 0."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @staticmethod
    def schedule(step, lr, beta1, beta2):
        beta2_t = beta2 ** step
        n_sma_max = 2 / (1 - beta2) - 1
        n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)
        if n_sma >= 5:
            step_size = lr * math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) / (1 - beta1 ** step)
        else:
            step_size = lr / (1 - beta1 ** step)
        return n_sma, step_size

    def _group_step(self, g):
        """steps taken so far = the per-parameter counter of any parameter with state (all move together)"""
        for p in g["params"]:
            st = self.state.get(p)
            if st and "step" in st:
                return int(st["step"])
        return 0

    def upload_hyper(self):
        for g in self.param_groups:
            step = self._group_step(g) + 1
            b1, b2 = g["betas"]
            n_sma, step_size = self.schedule(step, g["lr"], b1, b2)
            self._upload(g, [b1, b2, g["eps"], g["weight_decay"] * g["lr"], step_size, 1.0 if n_sma >= 5 else 0.0, self.grad_scale, 0])

    @torch.no_grad()
    def step(self, closure=None, upload=True):
        if upload:
            self.upload_hyper()
        for g in self.param_groups:
            ps = self._live(g)
            if not ps:
                continue
            ea, es = [], []
            for p in ps:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = 0               # radam.py:38 keeps the step count per parameter
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ea.append(st["exp_avg"]); es.append(st["exp_avg_sq"])
            for tl in _tensor_lists([ps, [p.grad for p in ps], ea, es]):
                L.call("saunet_radam_step", C.byref(tl), self._hyper(g).data_ptr(), L.stream())
            for p in ps:
                self.state[p]["step"] += 1
        HF.PACKS.invalidate()


class FusedAdam(_FusedBase):
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False) as /root/reference/train.py:197-201 builds it."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    _group_step = FusedRAdam._group_step

    def upload_hyper(self):
        for g in self.param_groups:
            step = self._group_step(g) + 1
            b1, b2 = g["betas"]
            self._upload(g, [b1, b2, g["eps"], g["weight_decay"], g["lr"] / (1.0 - b1 ** step), 1.0 / math.sqrt(1.0 - b2 ** step),
                             self.grad_scale, 0])

    @torch.no_grad()
    def step(self, closure=None, upload=True):
        if upload:
            self.upload_hyper()
        for g in self.param_groups:
            ps = self._live(g)
            if not ps:
                continue
            ea, es = [], []
            for p in ps:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ea.append(st["exp_avg"]); es.append(st["exp_avg_sq"])
            for tl in _tensor_lists([ps, [p.grad for p in ps], ea, es]):
                L.call("saunet_adam_step", C.byref(tl), self._hyper(g).data_ptr(), L.stream())
            for p in ps:
                self.state[p]["step"] += 1
        HF.PACKS.invalidate()


def create_optimizers(unet, optimizer="sgd", lr=5e-4, momentum=0.9, weight_decay=1e-4):
    """train.py:187-207 (defaults = train.py's argparse defaults)."""
    groups = group_weight(unet)
    name = optimizer.lower()
    if name == "sgd":
        return [FusedSGD(groups, lr=lr, momentum=momentum, weight_decay=weight_decay)]
    if name == "adam":
        return [FusedAdam(groups, lr=lr, betas=(0.9, 0.999))]
    if name == "radam":
        return [FusedRAdam(groups, lr=lr, betas=(0.9, 0.999))]
    raise ValueError("optimizer %s: train.py:187-207 knows sgd, adam and radam" % optimizer)
