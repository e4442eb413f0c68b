#!/usr/bin/env python3
"""Host-side cost of a config-3 step (LiftBlock, four FCResNetBlocks, ECHOBlock) WITHOUT a GPU: the library's enqueueing entry
points are replaced by recorders (as in tests/test_host_logic.py::test_config3_step_takes_at_most_40_foreign_calls), the tensors are
small CPU tensors, so what is timed is Python + autograd + ctypes + allocator work per step.  cProfile of the steady-state step."""
import cProfile
import ctypes
import os
import pstats
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fieldconv_amd import _lib, functional          # noqa: E402
from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock          # noqa: E402
from oracle.torch_composites import FCPrecomp          # noqa: E402
from test_host_logic import _CountingLibrary          # noqa: E402

header = open(os.path.join(ROOT, 'include', 'fieldconv_hip.h')).read()
counting = _CountingLibrary(_lib.load(), header)
_lib.load = lambda path=None: counting
functional.on_device = lambda t: True
functional._on = lambda d: functional._NO_GUARD
functional._stream = lambda: ctypes.c_void_p(0)
functional._require_device = lambda t, w: None
from fieldconv_amd.data import sphere_support          # noqa: E402

N, k, nf, B, R, n_cls = 64, 8, 48, 2, 6, 8
data = sphere_support(N, k)
edges, sten, ln, wxp = FCPrecomp(B, R, data.epsilon)(data)
net = torch.nn.ModuleDict(dict(lift=LiftBlock(3, nf, n_rings=R, ftype=1), r1=FCResNetBlock(nf, nf, band_limit=B, n_rings=R),
                               r2=FCResNetBlock(nf, nf, band_limit=B, n_rings=R), r3=FCResNetBlock(nf, nf, band_limit=B, n_rings=R),
                               r4=FCResNetBlock(nf, nf, band_limit=B, n_rings=R),
                               echo=ECHOBlock(nf, n_cls, n_des=48, n_bins=3, band_limit=B, n_rings=R)))
params = list(net.parameters())
pos = torch.randn(N, 3)
labels = torch.randint(0, n_cls, (N,))
torch.set_num_threads(1)


def step():
    x = net['lift'](pos, edges, sten[..., B:B + 2])
    for name in ('r1', 'r2', 'r3', 'r4'):
        x = net[name](x, edges, sten)
    logits = net['echo'](x, edges, sten, ln, wxp)
    loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), labels)
    return torch.autograd.grad(loss, params, allow_unused=True)


for mode in (os.environ.get('MODES', '1,0').split(',')):
    os.environ['FIELDCONV_BLOCK_CALLS'] = mode
    for _ in range(20):
        step()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    print(f'FIELDCONV_BLOCK_CALLS={mode}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms of host time per step (CPU tensors of {N} vertices)')
    if os.environ.get('PROFILE', '1') == '1' and mode == '1':
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(100):
            step()
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(22)
