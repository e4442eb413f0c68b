#!/usr/bin/env python3
"""What a custom autograd node costs the host by itself (GPU box): chains of dummy Functions with 11 / 2 tensor inputs whose passes only
allocate their outputs, against chains of FCResNetBlocks; enqueue time per node."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device('cuda:0')
N, C = 1024, 48


class Dummy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *params):
        ctx.save_for_backward(x, *params)
        return torch.empty_like(x)

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        return (torch.empty_like(saved[0]),) + tuple(torch.empty_like(p) for p in saved[1:])


class DummyLight(torch.autograd.Function):          # parameters kept on ctx (no version-checked unpacking), gradients carved from one buffer
    @staticmethod
    def forward(ctx, x, *params):
        ctx.save_for_backward(x)
        ctx.params = params
        return torch.empty_like(x)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        flat = torch.empty(sum(p.numel() for p in ctx.params), device=x.device)
        parts = flat.split([p.numel() for p in ctx.params])
        return (torch.empty_like(x),) + tuple(q.view(p.shape) for q, p in zip(parts, ctx.params))


def enqueue_us(fn, n=5, reps=9):
    for _ in range(5):
        fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    torch.cuda.synchronize()
    return best


x0 = torch.complex(torch.randn(N, C, device=dev), torch.randn(N, C, device=dev))
gy = torch.complex(torch.randn(N, C, device=dev), torch.randn(N, C, device=dev))


def chain(cls, n_nodes, n_params):
    sets = [[torch.randn(48, 48, 6, device=dev, requires_grad=True) for _ in range(n_params)] for _ in range(n_nodes)]
    flat = [p for s in sets for p in s]

    def run():
        x = x0.detach().requires_grad_(True)
        y = x
        for s in sets:
            y = cls.apply(y, *s)
        torch.autograd.grad(y, [x] + flat, grad_outputs=gy)
    return run


for cls in (Dummy, DummyLight):
    for n_params in (10, 1):
        t1, t5 = enqueue_us(chain(cls, 1, n_params)), enqueue_us(chain(cls, 5, n_params))
        print(f'{cls.__name__:10s} {n_params:2d} parameter tensors: 1 node {t1:7.1f} us, 5 nodes {t5:7.1f} us -> {(t5 - t1) / 4:6.1f} us per extra node')

from fieldconv_amd.data import sphere_support          # noqa: E402
from fieldconv_amd.nn import FCResNetBlock          # noqa: E402
from fieldconv_amd.transforms import FCPrecomp          # noqa: E402
data = sphere_support(N, 128).to(dev)
edges, sten, ln, wxp = FCPrecomp(2, 6, data.epsilon)(data)
blocks = [FCResNetBlock(C, C, band_limit=2, n_rings=6).to(dev) for _ in range(5)]


def block_chain(n):
    params = [p for b in blocks[:n] for p in b.parameters()]

    def run():
        x = x0.detach().requires_grad_(True)
        y = x
        for b in blocks[:n]:
            y = b(y, edges, sten)
        torch.autograd.grad(y, [x] + params, grad_outputs=gy)
    return run


t1, t5 = enqueue_us(block_chain(1)), enqueue_us(block_chain(5))
print(f'FCResNetBlock: 1 block {t1:7.1f} us, 5 blocks {t5:7.1f} us -> {(t5 - t1) / 4:6.1f} us per extra block (forward + backward)')


def fwd_chain(n):
    def run():
        with torch.no_grad():
            y = x0
            for b in blocks[:n]:
                y = b(y, edges, sten)
    return run


f1, f5 = enqueue_us(fwd_chain(1)), enqueue_us(fwd_chain(5))
print(f'FCResNetBlock forward only (no_grad): {(f5 - f1) / 4:6.1f} us per block')
