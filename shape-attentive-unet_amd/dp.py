"""Data parallelism for the SAUNet step: one process per GPU, gradients all-reduced over RCCL/xGMI.

Replaces the reference's single-process ``UserScatteredDataParallel`` + ``patch_replication_callback``
(/root/reference/lib/nn/parallel/data_parallel.py:48-62, lib/nn/modules/replicate.py:70-94,
train.py:272-277), which re-broadcasts all weights every step and reduces gradients to GPU 0.
Here every rank keeps a replica; gradients are packed into a few large buckets (xGMI is point-to-point:
few, large messages) in reverse-topological order and each bucket's all-reduce is launched as soon as its
last gradient has been produced, overlapping the remaining backward kernels.  SyncBN statistics of the
three shape-stream ResBlocks are all-reduced inside functional.conv_bn_act.

Semantics: per-replica loss on the local shard, gradients averaged over ranks,
local batch statistics for the 144 nn.BatchNorm2d layers, global statistics for the 6 SyncBN layers.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import lib as L
from . import functional as HF


def init_from_env(backend=None):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and join the default group.
    Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SAUNET_SHARE_GPU") == "1" and torch.cuda.is_available():
        local %= torch.cuda.device_count()          # test rig: several ranks on one GPU (needs SAUNET_DIST_BACKEND=gloo)
    backend = backend or os.environ.get("SAUNET_DIST_BACKEND") or None
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


def shard_indices(n, rank, world, epoch=0, seed=304, shuffle=True, drop_last=True):
    """Disjoint per-rank shards of a (seeded) permutation, equal length on every rank."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist() if shuffle else list(range(n))
    per = n // world if drop_last else (n + world - 1) // world
    if not drop_last:
        idx += idx[: per * world - n]
    return idx[rank * per:(rank + 1) * per]


def broadcast_parameters(module, src=0):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def _copy(tl_cols, pack, scale):
    """pack=True: flat <- grads ; pack=False: grads <- flat * scale."""
    tensors, flats = tl_cols
    if tensors[0].is_cuda:
        for s in range(0, len(tensors), 96):
            tl = L.TensorList()
            cnt = min(96, len(tensors) - s)
            tl.count = cnt
            for i in range(cnt):
                tl.ptrs[0][i] = tensors[s + i].data_ptr()
                tl.ptrs[1][i] = flats[s + i].data_ptr()
                tl.numel[i] = tensors[s + i].numel()
            L.call("saunet_bucket_copy", C.byref(tl), 1 if pack else 0, float(scale), L.stream())
    else:  # host tensors (gloo tests of the bucketing logic): plain copies
        for t, f in zip(tensors, flats):
            if pack:
                f.copy_(t.reshape(-1))
            else:
                t.copy_((f * scale).view_as(t))


class GradientBuckets:
    """Bucketed, backward-overlapped gradient averaging.

    Parameters that never receive a gradient (``encoder.classifier``: registered like in the reference, off the compute path)
    are detected on the first step and dropped from the buckets: their ``.grad`` stays None on every rank -- exactly as in a
    single-process run, where the optimiser skips them -- and no bucket waits for a gradient that never comes."""

    def __init__(self, params, bucket_mb=32.0, group=None, overlap=True):
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cap = int(bucket_mb * 1024 * 1024 / 4)
        self.overlap = overlap
        self.hooks = []
        self.first_step_done = False
        self.exposed_events = None        # (start, end) HIP events around the wait+unpack of finish(), if timing is enabled
        self.time_finish = False
        self.trace = None                 # time_finish: per-bucket record of the last step, see bucket_table()
        self._t0 = None
        self._build(self.params)

    def _build(self, params):
        # reverse registration order ~ order in which backward produces gradients (final ... conv0)
        self.remove_hooks()
        self.params = list(params)
        self.buckets, cur, cur_n = [], [], 0
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > self.cap:
                self.buckets.append(cur); cur, cur_n = [], 0
            cur.append(p); cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
        self.flat = [None] * len(self.buckets)
        self.bucket_of = {}
        for b, ps in enumerate(self.buckets):
            for p in ps:
                self.bucket_of[id(p)] = b
        if self.overlap and self.world > 1:
            for p in self.params:
                self.hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    def reset(self):
        self.pending = [len(ps) for ps in self.buckets]
        self.handles = [None] * len(self.buckets)

    def _views(self, b):
        ps = self.buckets[b]
        n = sum(p.numel() for p in ps)
        if self.flat[b] is None or self.flat[b].device != ps[0].device:
            self.flat[b] = torch.empty(n, dtype=torch.float32, device=ps[0].device)
        views, o = [], 0
        for p in ps:
            views.append(self.flat[b][o:o + p.numel()]); o += p.numel()
        return views

    def _present(self, b):
        """(gradients, flat views) of the bucket's parameters that have a gradient"""
        views = self._views(b)
        pairs = [(p.grad, v) for p, v in zip(self.buckets[b], views) if p.grad is not None]
        return [g for g, _ in pairs], [v for _, v in pairs]

    def _launch(self, b):
        HF.flush_deferred_wgrads()    # weight gradients whose cross-workgroup reduction was deferred to the end of backward must be complete now
        grads, views = self._present(b)
        if len(grads) != len(self.buckets[b]):
            self.flat[b].zero_()      # first step only: slots of gradient-less parameters travel as zeros (every rank agrees)
        if grads:
            _copy((grads, views), True, 1.0)
        if self.time_finish:
            import time
            if self.trace is None or len(self.trace) != len(self.buckets) or all(t.get("waited") for t in self.trace):
                self.trace = [{} for _ in self.buckets]
                self._t0 = time.perf_counter()
            rec = self.trace[b]
            rec.update(bucket=b, numel=int(self.flat[b].numel()), host_launch_ms=(time.perf_counter() - self._t0) * 1e3)
            if self.flat[b].is_cuda:
                rec["ev_launch"] = torch.cuda.Event(enable_timing=True); rec["ev_launch"].record()
        self.handles[b] = dist.all_reduce(self.flat[b], group=self.group, async_op=True)

    def _on_grad(self, p):
        b = self.bucket_of[id(p)]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def finish(self):
        """Call after backward: launches whatever has not been sent, waits, writes the averaged gradients back."""
        if self.world <= 1:
            return
        for b in range(len(self.buckets)):
            if self.handles[b] is None:
                self._launch(b)
        ev = None
        if self.time_finish and self.flat[0].is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for b in range(len(self.buckets)):
            if self.time_finish and self.trace is not None:
                import time
                self.trace[b]["host_wait_start_ms"] = (time.perf_counter() - self._t0) * 1e3
            self.handles[b].wait()
            grads, views = self._present(b)
            if grads:
                _copy((grads, views), False, 1.0 / self.world)
            if self.time_finish and self.trace is not None:
                import time
                rec = self.trace[b]
                rec["host_wait_end_ms"] = (time.perf_counter() - self._t0) * 1e3
                rec["waited"] = True
                if self.flat[b].is_cuda:
                    rec["ev_done"] = torch.cuda.Event(enable_timing=True); rec["ev_done"].record()
        if ev is not None:
            ev[1].record()
            self.exposed_events = ev
        if not self.first_step_done:
            self.first_step_done = True
            used = [p for p in self.params if p.grad is not None]
            if len(used) != len(self.params):
                self._build(used)
                return
        self.reset()

    def bucket_table(self):
        """Per-bucket view of the LAST step's gradient exchange (time_finish=True): when each bucket's all-reduce was launched (relative to the first
        launch of the step) and when the compute stream had it back, so exposed vs hidden exchange time is a table, not one number.  On the GPU
        the times are HIP-event times on the compute stream; `exposed_ms` of bucket b = what the stream waited for it beyond its predecessor.
        Synchronises the device."""
        if not self.trace:
            return []
        rows, prev_done = [], None
        cuda = all("ev_launch" in t and "ev_done" in t for t in self.trace)
        if cuda:
            torch.cuda.synchronize()
        first = self.trace[min(range(len(self.trace)), key=lambda i: self.trace[i].get("host_launch_ms", 0.0))]
        for t in self.trace:
            row = {"bucket": t.get("bucket"), "MB": round(t.get("numel", 0) * 4 / 1e6, 2), "host_launch_ms": round(t.get("host_launch_ms", 0.0), 3),
                   "host_wait_ms": round(t.get("host_wait_end_ms", 0.0) - t.get("host_wait_start_ms", 0.0), 3)}
            if cuda:
                row["launch_ms"] = round(first["ev_launch"].elapsed_time(t["ev_launch"]), 3)
                row["done_ms"] = round(first["ev_launch"].elapsed_time(t["ev_done"]), 3)
                if self.exposed_events is not None:
                    start = first["ev_launch"].elapsed_time(self.exposed_events[0])
                    base = max(start, prev_done if prev_done is not None else start)
                    row["exposed_ms"] = round(max(0.0, row["done_ms"] - base), 3)
                prev_done = row["done_ms"]
            rows.append(row)
        return rows

    def remove_hooks(self):
        for h in self.hooks:
            h.remove()
        self.hooks = []


def all_reduce_scalars(values, group=None):
    """Average a small vector of logging scalars over ranks (the reference gathers them to GPU 0)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return values
    dist.all_reduce(values, group=group)
    return values / dist.get_world_size(group)
