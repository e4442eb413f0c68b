#!/usr/bin/env python3
"""Per-mesh preprocessing cost, ms: the fused build (FCPrecomp -> graph + records, no dense stencil) against the two-step
path (FCPrecomp writes the (E,R,F) stencil, SupportGraph analyses it), config 3 and config 2 meshes."""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fieldconv_amd.data import sphere_support
from fieldconv_amd.graph import SupportGraph
from fieldconv_amd.transforms import FCPrecomp

dev = torch.device('cuda:0')
for N, k, support in ((1024, 128, 'all'), (20000, 32, 'p95')):
    data = sphere_support(N, k, support=support).to(dev)
    pre = FCPrecomp(2, 6, data.epsilon)

    def timed(fn, n=30):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, out
    args = (data.logMag, data.logAng, data.w, data.supp_edges, data.xp)
    os.environ.pop('FIELDCONV_EAGER_STENCIL', None)
    t_fused, (edges, sten, ln, wxp) = timed(lambda: pre._compute(*args))
    os.environ['FIELDCONV_EAGER_STENCIL'] = '1'
    t_pre, (edges2, sten2, _, _) = timed(lambda: pre._compute(*args))
    t_graph, _ = timed(lambda: SupportGraph(edges2, sten2, N))
    os.environ.pop('FIELDCONV_EAGER_STENCIL', None)
    print(f'N={N} k={k} E={edges.shape[0]}: fused FCPrecomp+graph {t_fused:.2f} ms | two-step: FCPrecomp {t_pre:.2f} + SupportGraph '
          f'{t_graph:.2f} = {t_pre + t_graph:.2f} ms')
