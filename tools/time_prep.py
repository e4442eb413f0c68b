#!/usr/bin/env python3
"""Time the support-graph preprocessing (and FCPrecomp) on the config-2 mesh."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fieldconv_amd.data import sphere_support
from fieldconv_amd.graph import SupportGraph
from fieldconv_amd.transforms import FCPrecomp
N, k, B, R = int(os.environ.get('N', 20000)), int(os.environ.get('K', 32)), 2, 6
dev = torch.device('cuda:0')
data = sphere_support(N, k).to(dev)
pre = FCPrecomp(B, R, data.epsilon)
for name, fn in (('FCPrecomp', lambda: pre(data)), ):
    for _ in range(3): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = fn()
    torch.cuda.synchronize(); print(name, 'ms', (time.perf_counter() - t0) / 10 * 1e3)
edges, sten, _, _ = out
for fact in (True, False):
    for _ in range(3): g = SupportGraph(edges, sten, N, allow_factored=fact)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g = SupportGraph(edges, sten, N, allow_factored=fact)
    torch.cuda.synchronize(); print('SupportGraph factored=%s ms' % fact, (time.perf_counter() - t0) / 10 * 1e3)
