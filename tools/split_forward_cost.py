"""What the two-launch forward of dist.overlap_forward costs on one GPU (no exchange): config-2 layer step with the forward
pass in one launch and split at 18 504 of 20 000 targets.   python tools/split_forward_cost.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fieldconv_amd.data import sphere_support          # noqa: E402
from fieldconv_amd.graph import get_graph              # noqa: E402
from fieldconv_amd.nn import FieldConv                 # noqa: E402
from fieldconv_amd.transforms import FCPrecomp         # noqa: E402

dev = torch.device('cuda:0')
N, k, C, B, R = 20000, 32, 48, 2, 6
data = sphere_support(N, k, seed=0, support='p95').to(dev)
edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
from fieldconv_amd.graph import FactoredStencil     # noqa: E402
graph = get_graph(edges, sten, N).view()
sten = FactoredStencil.wrap(sten, graph)
conv = FieldConv(C, C, band_limit=B, n_rings=R, ftype=1).to(dev)
x = torch.randn(N, C, dtype=torch.complex64, device=dev, requires_grad=True)
gy = torch.randn(N, C, dtype=torch.complex64, device=dev)
params = [x] + list(conv.parameters())


def run(steps):
    for _ in range(steps):
        torch.autograd.grad(conv(x, edges, sten), params, grad_outputs=gy)


splits = [int(a) for a in sys.argv[1:]] or [18504]
run(300)                                # the clock governor needs ~100 ms of load to reach the sustained clock
for n_first in [0] + splits + [0] + splits:
    name = f'split at {n_first}' if n_first else 'one launch'
    graph.forward_split = (n_first, lambda: None) if n_first else None
    run(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(200)
    torch.cuda.synchronize()
    print(f'{name:16s} {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per forward+backward')
