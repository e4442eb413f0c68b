#!/usr/bin/env python3
"""How long the host needs to ENQUEUE one FieldConv fwd+bwd step, against how long the GPU needs to run it."""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fieldconv_amd.data import sphere_support
from fieldconv_amd.nn import FieldConv
from fieldconv_amd.transforms import FCPrecomp

N, k, C, B, R = 20000, 32, 48, 2, 6
dev = torch.device('cuda:0')
data = sphere_support(N, k).to(dev)
edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
conv = FieldConv(C, C, band_limit=B, n_rings=R).to(dev)
params = list(conv.parameters())
g = torch.Generator().manual_seed(0)
x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev).requires_grad_(True)
gy = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev)


def step():
    y = conv(x, edges, sten)
    torch.autograd.grad(y, [x] + params, grad_outputs=gy)


for _ in range(10):
    step()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'host enqueue {t_enq / n * 1e6:.0f} us/step, wall {t_all / n * 1e6:.0f} us/step')
