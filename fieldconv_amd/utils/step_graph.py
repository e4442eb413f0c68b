"""Replaying a whole training step as one HIP graph.

A network step on a segmentation-sized mesh (about a thousand vertices) is ~180 kernel launches of a few
microseconds each: the host (Python, autograd bookkeeping, ctypes calls) takes longer to enqueue them than the GPU
takes to run them.  Every launch of this package goes to torch's current stream with pointer arguments only, no
host synchronisation and no allocation outside torch's caching allocator, so a step can be captured with stream
capture and replayed with a single hipGraphLaunch.  `torch.cuda.CUDAGraph` is the capture API (a hipGraph on ROCm).

The mesh tensors, features and parameters are *static*: the graph reads the addresses seen at capture time.  New
values are written into the same tensors (`x.copy_(...)`, in-place optimizer updates); one StepGraph per mesh.

Pitfall (PyTorch's, observed on this ROCm build as a segfault in capture_end): no tensor that still carries an autograd
graph from an EARLIER eager run of the same step may be alive at capture time -- e.g. a kept `loss`, or `loss.clone()`,
which is differentiable and holds the graph too.  Its AccumulateGrad nodes are bound to the stream they were created
on (the default stream), and the engine then synchronises that stream with the capturing one.  Keep `loss.detach()`.
"""
import torch


class StepGraph:
    """Capture `fn()` -- forward, loss and `torch.autograd.grad` (or `.backward()`) on static tensors -- once and
    replay it.  `fn` returns a tensor or a tuple of tensors; `replay()` returns the same objects with refreshed
    contents.

        graphed = StepGraph(lambda: step(data))      # runs fn a few times on a side stream, then captures
        loss, *grads = graphed.replay()
    """

    def __init__(self, fn, warmup=3):
        if not torch.cuda.is_available():
            raise RuntimeError('StepGraph needs a ROCm device: HIP graphs replay device work only')
        self.graph = torch.cuda.CUDAGraph()
        # warm-up on a side stream (capture must not run on the default stream): fills the support-graph / packed
        # filter caches and the allocator's pools, so that the captured run only launches kernels
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph):
            out = fn()
            # detached: an output that still holds its autograd graph would keep the captured step's intermediates
            # (and their backward nodes, bound to the capture stream) alive past the capture
            self.outputs = tuple(t.detach() for t in out) if isinstance(out, (tuple, list)) else out.detach()
            del out

    def replay(self):
        self.graph.replay()
        return self.outputs
