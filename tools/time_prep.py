#!/usr/bin/env python3
"""Per-mesh preprocessing cost: FCPrecomp (stencil assembly), SupportGraph (CSR + records) and EdgeCSR, ms each."""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fieldconv_amd.data import sphere_support
from fieldconv_amd.graph import EdgeCSR, SupportGraph
from fieldconv_amd.transforms import FCPrecomp

dev = torch.device('cuda:0')
for N, k in ((1024, 128), (20000, 32)):
    data = sphere_support(N, k).to(dev)
    pre = FCPrecomp(2, 6, data.epsilon)

    def timed(fn, n=20):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, out
    t_pre, (edges, sten, ln, wxp) = timed(lambda: pre._compute(data.logMag, data.logAng, data.w, data.supp_edges, data.xp))
    t_graph, _ = timed(lambda: SupportGraph(edges, sten, N))
    t_csr, _ = timed(lambda: EdgeCSR(edges, N))
    print(f'N={N} k={k} E={edges.shape[0]}: FCPrecomp {t_pre:.2f} ms, SupportGraph {t_graph:.2f} ms, EdgeCSR {t_csr:.2f} ms')
