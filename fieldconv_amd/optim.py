"""Fused Adam for the networks built from this package.

All parameters live in one flat float32 buffer (each `p.data` becomes a view of it, each `p.grad` a view of one flat
gradient buffer), so that an optimizer step is one elementwise HIP kernel (csrc/fc_optim.hip through fc_adam_step) and
`zero_grad` at most one fill.  The step counter lives on the device: the update is capturable in a HIP graph together with the
forward and backward passes (fieldconv_amd.utils.StepGraph).  Arithmetic of torch.optim.Adam (L2 weight decay, no amsgrad).
"""
import ctypes

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = [p for p in params]
        if not params:
            raise ValueError('FusedAdam needs at least one parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError('FusedAdam keeps one flat buffer: one parameter group')
        ps = self.param_groups[0]['params']
        dev = ps[0].device
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev or not p.is_cuda:
                raise ValueError('FusedAdam: float32 parameters on one ROCm device (complex filters are stored as real pairs '
                                 'by the modules of this package)')
        sizes = [p.numel() for p in ps]
        offs, total = [], 0
        for n in sizes:                     # every tensor starts on a 16-byte boundary
            offs.append(total)
            total += (n + 3) // 4 * 4
        self._n = total
        self._flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self._m = torch.zeros(total, dtype=torch.float32, device=dev)
        self._v = torch.zeros(total, dtype=torch.float32, device=dev)
        self._step = torch.zeros(1, dtype=torch.float32, device=dev)
        self._offsets, self._sizes = offs, sizes
        self._views = []                    # the gradient views, one per parameter
        with torch.no_grad():
            for p, o, n in zip(ps, offs, sizes):
                view = self._flat[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self._grad[o:o + n].view(p.shape)
                self._views.append(p.grad)
        self._built = True

    def add_param_group(self, param_group):
        if getattr(self, '_built', False):
            raise RuntimeError('FusedAdam keeps every parameter in one flat buffer laid out at construction: '
                               'parameter groups cannot be added afterwards')
        super().add_param_group(param_group)

    def state_dict(self):
        """torch.optim.Adam's layout: per-parameter `step`, `exp_avg`, `exp_avg_sq` (copies of the flat buffers' slices),
        so that a checkpoint written here resumes under torch.optim.Adam and vice versa -- exactly when every parameter got a
        gradient in every step (see _pack_grads); every entry carries the shared step count."""
        ps = self.param_groups[0]['params']
        step = self._step.detach().clone().reshape(())
        state = {i: {'step': step.clone(), 'exp_avg': self._m[o:o + n].view(p.shape).clone(),
                     'exp_avg_sq': self._v[o:o + n].view(p.shape).clone()}
                 for i, (p, o, n) in enumerate(zip(ps, self._offsets, self._sizes))}
        group = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        group['params'] = list(range(len(ps)))
        return {'state': state, 'param_groups': [group]}

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        groups = state_dict['param_groups']
        ps = self.param_groups[0]['params']
        if len(groups) != 1 or len(groups[0]['params']) != len(ps):
            raise ValueError('FusedAdam.load_state_dict: expected one parameter group with '
                             f'{len(ps)} parameters')
        for k, v in groups[0].items():
            if k != 'params' and k in self.param_groups[0]:
                self.param_groups[0][k] = v
        state = state_dict['state']
        steps = set()
        for i, (p, o, n) in enumerate(zip(ps, self._offsets, self._sizes)):
            st = state.get(i, state.get(str(i)))
            if st is None:                      # a parameter that never received a gradient under torch.optim.Adam
                self._m[o:o + n].zero_()
                self._v[o:o + n].zero_()
                continue
            self._m[o:o + n].copy_(st['exp_avg'].reshape(-1))
            self._v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(float(st['step']))
        if len(steps) > 1:
            raise ValueError('FusedAdam keeps one step counter: the checkpoint has parameters at different steps')
        self._step.fill_(steps.pop() if steps else 0.0)

    def zero_grad(self, set_to_none=True):
        """set_to_none=True (the default, as in torch.optim): the parameters are detached from the flat buffer, autograd assigns
        every gradient, and step() packs them with one multi-tensor copy -- no fill and no add kernel per parameter tensor
        (the segmentation network's training step: 2.99 -> 2.54 ms eager, 1.81 -> 1.71 ms as one HIP graph).
        set_to_none=False: one fill, the gradient views stay in place and backward accumulates into them."""
        if set_to_none:
            for p in self.param_groups[0]['params']:
                p.grad = None
        else:
            self._grad.zero_()

    def _pack_grads(self):
        """Gradients that autograd assigned (after zero_grad(set_to_none=True)) go into the flat buffer; .grad points into the
        buffer again.  A parameter WITHOUT a gradient counts as having a zero gradient: the one elementwise kernel updates the
        whole flat buffer with one shared step counter, so its moments decay and weight decay applies, where torch.optim.Adam
        would skip it (and keep a per-parameter step).  The two agree whenever every parameter handed to the optimizer receives
        a gradient every step -- true for the networks of this package (the `phase` of ftype 0 / 2 layers is a buffer, not a
        parameter); frozen or unused parameters should not be given to FusedAdam."""
        lo, hi = self._grad.data_ptr(), self._grad.data_ptr() + 4 * self._n
        views, grads = [], []
        for p, v in zip(self.param_groups[0]['params'], self._views):
            gr = p.grad
            if gr is None:
                v.zero_()
            elif not (lo <= gr.data_ptr() < hi):
                if gr.shape != v.shape or gr.dtype != v.dtype or gr.device != v.device:
                    raise RuntimeError('FusedAdam: a parameter\'s .grad does not match the parameter')
                views.append(v)
                grads.append(gr)
            else:
                continue
            p.grad = v
        if views:
            torch._foreach_copy_(views, grads)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self._pack_grads()
        lib = _lib.load()
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(self._flat.device):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(lib.fc_adam_step(p(self._flat), p(self._grad), p(self._m), p(self._v), p(self._step), self._n, float(g['lr']),
                                        float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), st),
                       'fc_adam_step')
        return loss
