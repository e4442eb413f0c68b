#!/usr/bin/env python3
"""Small driver for rocprofv3: runs the config-2 FieldConv layer fwd+bwd a few times (no CPU baseline,
no distributed set-up), so counter passes stay short.   python tools/profile_step.py [steps] [graph: geo|rand]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fieldconv_amd.data import random_support, sphere_support      # noqa: E402
from fieldconv_amd.nn import FieldConv                              # noqa: E402
from fieldconv_amd.transforms import FCPrecomp                      # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
kind = sys.argv[2] if len(sys.argv) > 2 else 'geo'
N, k, C, B, R = 20000, 32, 48, 2, 6
dev = torch.device('cuda:0')
data = (sphere_support(N, k) if kind == 'geo' else random_support(N, k)).to(dev)
edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
torch.manual_seed(0)
conv = FieldConv(C, C, band_limit=B, n_rings=R, ftype=1).to(dev)
g = torch.Generator().manual_seed(0)
x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev).requires_grad_(True)
gy = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev)
for _ in range(steps):
    y = conv(x, edges, sten)
    torch.autograd.grad(y, [x] + list(conv.parameters()), grad_outputs=gy)
torch.cuda.synchronize()
print('done', steps)
