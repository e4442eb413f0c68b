"""Bucketed gradient all-reduce for data-parallel training (BASELINE configs[4]: one mesh per GPU, replicated
parameters; the reference has no multi-GPU code).

Every parameter's `.grad` becomes a view into ONE flat float32 buffer (the same layout FusedAdam uses; an existing
FusedAdam's buffer is adopted instead of allocating a second one), so that a step's gradient exchange is one
`all_reduce` per bucket of at most `bucket_bytes` -- a single collective for the ~1 M parameters (4 MB) of the
networks in this package.  xGMI rings are per-link bound: few, large messages.  Backend "nccl" is RCCL on ROCm; with
"gloo" and device tensors (several test ranks on one GPU) the bucket is staged through the host.
"""
import torch
import torch.distributed as dist


class GradientBuckets:
    def __init__(self, params_or_optimizer, group=None, bucket_bytes=64 << 20, average=False):
        self.group = group
        self.average = average
        flat = getattr(params_or_optimizer, '_grad', None)          # FusedAdam: gradients already live in one buffer
        self.params, self.views = [], []
        if flat is not None:
            self.flat = flat
            self.params = list(params_or_optimizer.param_groups[0]['params'])
            self.views = list(params_or_optimizer._views)
        else:
            params = [p for p in params_or_optimizer if p.requires_grad]
            if not params:
                raise ValueError('GradientBuckets needs at least one parameter')
            dev, dt = params[0].device, params[0].dtype
            if any(p.device != dev or p.dtype != dt for p in params):
                raise ValueError('GradientBuckets: parameters of one dtype on one device')
            sizes = [p.numel() for p in params]
            offs, total = [], 0
            for n in sizes:                     # every tensor starts on a 16-byte boundary
                offs.append(total)
                total += (n + 3) // 4 * 4
            self.flat = torch.zeros(total, dtype=dt, device=dev)
            for p, o, n in zip(params, offs, sizes):
                p.grad = self.flat[o:o + n].view(p.shape)
                self.params.append(p)
                self.views.append(p.grad)
        per = max(1, int(bucket_bytes) // self.flat.element_size())
        self.buckets = [self.flat[i:i + per] for i in range(0, self.flat.numel(), per)]

    def zero(self):
        """Gradients accumulate into the flat buffer: zero it, run backward (one small add kernel per parameter)."""
        self.flat.zero_()

    def begin(self):
        """Alternative to zero(): detach the parameters from the buffer for this backward pass, so that autograd ASSIGNS every
        gradient instead of adding it to a zeroed view -- no fill and no add kernel per parameter (a hundred launches of a few
        microseconds each on the networks of this package); collect() then packs them with one multi-tensor copy."""
        for p in self.params:
            p.grad = None

    def collect(self, accumulate=False):
        """After backward (following begin()): fresh gradients into the flat buffer (added to it with accumulate=True: several
        backward passes per step), parameters' .grad point into the buffer again."""
        views, grads = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                if not accumulate:
                    v.zero_()                          # a parameter the loss does not depend on
            elif p.grad is not v:
                views.append(v)
                grads.append(p.grad)
            p.grad = v
        if views:
            if accumulate:
                torch._foreach_add_(views, grads)
            else:
                torch._foreach_copy_(views, grads)

    def all_reduce(self, async_op=False):
        """Sum (or average) the gradients over the group; returns the Work handles when async_op (empty on the
        host-staged path, which completes here)."""
        self.collect()                      # no-op unless gradients were assigned since begin() / zero_grad(set_to_none=True)
        world = dist.get_world_size(self.group)
        works = []
        staged = self.flat.is_cuda and dist.get_backend(self.group) == 'gloo'
        for b in self.buckets:
            if staged:
                h = b.cpu()
                dist.all_reduce(h, group=self.group)
                b.copy_(h)
            else:
                w = dist.all_reduce(b, group=self.group, async_op=async_op)
                if async_op:
                    works.append(w)
        if self.average and world > 1:
            if works:
                for w in works:
                    w.wait()
                works = []
            self.flat.div_(world)
        return works
