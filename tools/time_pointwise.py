#!/usr/bin/env python3
"""Time the TangentLin / TangentNonLin entry points alone (HIP events, median of reps)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fieldconv_amd.functional import tangent_lin, tangent_nonlin

N, C = int(os.environ.get('N', 20000)), int(os.environ.get('C', 48))
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev).requires_grad_(True)
gy = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev)
Re = (torch.randn(C, C, generator=g) * 0.1).to(dev).requires_grad_(True)
Im = (torch.randn(C, C, generator=g) * 0.1).to(dev).requires_grad_(True)
b = (torch.randn(C, generator=g) * 0.1).to(dev).requires_grad_(True)


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


y_lin = tangent_lin(x, Re, Im)
y_non = tangent_nonlin(x, b)
print('lin fwd     %7.1f us' % timeit(lambda: tangent_lin(x, Re, Im)))
print('lin fwd+bwd %7.1f us' % timeit(lambda: torch.autograd.grad(tangent_lin(x, Re, Im), (x, Re, Im), gy)))
print('non fwd     %7.1f us' % timeit(lambda: tangent_nonlin(x, b)))
print('non fwd+bwd %7.1f us' % timeit(lambda: torch.autograd.grad(tangent_nonlin(x, b), (x, b), gy)))
