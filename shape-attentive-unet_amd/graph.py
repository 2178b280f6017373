"""hipGraph capture of a whole SAUNet step (forward + loss + backward [+ all-reduce] [+ fused optimiser]).

The step has no host synchronisation (Canny runs on the device, metrics come out of the loss kernel, hyper-parameters are
read from a device array), so ~2000 launches replay as one graph.  Two things make a captured TRAINING step correct:

  * every weight re-packing launch is recorded inside the graph (functional.PackedWeights.prepack re-packs unconditionally
    while the stream is capturing), so each replay packs the master weights the previous replay's optimiser step produced;
  * after a replay the host-side caches are told that parameters changed behind their back
    (functional.notify_params_changed), so an eager forward / eval that follows re-packs and re-finalises.
"""
import torch

from . import functional as HF


class GraphedStep:
    """``fn()`` is run ``warmup`` times eagerly on a side stream (allocator pools, optimiser state, pack cache), then
    captured.  ``replay()`` launches the graph and returns whatever ``fn`` returned during capture (static tensors)."""

    def __init__(self, fn, warmup=1, changes_params=True, optimizers=()):
        """optimizers: the fused optimisers whose step(upload=False) is inside `fn`; their device hyper-parameter arrays (learning rate,
        Adam / RAdam bias-correction terms) are refreshed before every replay and their host step counters advanced after it."""
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs the GPU (no CPU fallback)")
        self.changes_params = changes_params
        self.optimizers = list(optimizers)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(warmup, 1)):
                for o in self.optimizers:          # every warm-up iteration is a real step: it needs its own bias-correction / step-size terms
                    o.upload_hyper()
                fn()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        for o in self.optimizers:
            o.upload_hyper()
        with torch.cuda.graph(self.graph):
            self.out = fn()
        for o in self.optimizers:
            o.capture_rollback()
        if changes_params:
            HF.notify_params_changed()

    def replay(self):
        for o in self.optimizers:
            o.pre_replay()
        self.graph.replay()
        for o in self.optimizers:
            o.post_replay()
        if self.changes_params:
            HF.notify_params_changed()
        return self.out
