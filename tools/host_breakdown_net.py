#!/usr/bin/env python3
"""Where the HOST's time per config-3 step goes (on the GPU box): enqueue time -- a few repetitions issued behind a synchronisation,
timed until the last call returns -- of the network's parts on their own."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fieldconv_amd.data import sphere_support          # noqa: E402
from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock          # noqa: E402
from fieldconv_amd.transforms import FCPrecomp          # noqa: E402

N, k, nf, B, R, n_cls = 1024, 128, 48, 2, 6, 8
dev = torch.device('cuda:0')
data = sphere_support(N, k).to(dev)
edges, sten, ln, wxp = FCPrecomp(B, R, data.epsilon)(data)
torch.manual_seed(0)
lift = LiftBlock(3, nf, n_rings=R, ftype=1).to(dev)
blocks = [FCResNetBlock(nf, nf, band_limit=B, n_rings=R).to(dev) for _ in range(4)]
echo = ECHOBlock(nf, n_cls, n_des=nf, n_bins=3, band_limit=B, n_rings=R).to(dev)
pos = torch.randn(N, 3, device=dev)
labels = torch.randint(0, n_cls, (N,), device=dev)
x0 = torch.complex(torch.randn(N, nf, device=dev), torch.randn(N, nf, device=dev))
gy = torch.complex(torch.randn(N, nf, device=dev), torch.randn(N, nf, device=dev))
lsten = sten[..., B:B + 2]


def enqueue_ms(fn, n=5, reps=7):
    for _ in range(5):
        fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    torch.cuda.synchronize()
    return best


def part_lift():
    y = lift(pos, edges, lsten)
    torch.autograd.grad(y, list(lift.parameters()), grad_outputs=gy)


def part_block():
    x = x0.detach().requires_grad_(True)
    y = blocks[0](x, edges, sten)
    torch.autograd.grad(y, [x] + list(blocks[0].parameters()), grad_outputs=gy)


def part_block_fwd():
    with torch.no_grad():
        blocks[0](x0, edges, sten)


def part_echo():
    x = x0.detach().requires_grad_(True)
    logits = echo(x, edges, sten, ln, wxp)
    loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), labels)
    torch.autograd.grad(loss, [x] + list(echo.parameters()))


def part_echo_desc():
    from fieldconv_amd.blocks import echo_block_descriptors
    from fieldconv_amd.graph import get_graph
    x = x0.detach().requires_grad_(True)
    d = echo_block_descriptors(echo, x, get_graph(edges, sten, N), ln, wxp)
    torch.autograd.grad(d, [x], grad_outputs=torch.ones_like(d))


def part_mlp():
    d = torch.zeros(N, nf * 37, device=dev, requires_grad=True)
    h = torch.relu(echo.lin1(d))
    h = torch.relu(echo.lin2(h))
    logits = echo.lin3(h)
    loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), labels)
    torch.autograd.grad(loss, [d] + list(echo.lin1.parameters()) + list(echo.lin2.parameters()) + list(echo.lin3.parameters()))


def whole():
    x = lift(pos, edges, lsten)
    for b in blocks:
        x = b(x, edges, sten)
    logits = echo(x, edges, sten, ln, wxp)
    loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), labels)
    params = list(lift.parameters()) + [p for b in blocks for p in b.parameters()] + list(echo.parameters())
    torch.autograd.grad(loss, params)


def empty_launches():
    for _ in range(10):
        torch.empty(16, device=dev)


for name, fn in (('whole step', whole), ('LiftBlock fwd+bwd', part_lift), ('one FCResNetBlock fwd+bwd', part_block),
                 ('one FCResNetBlock fwd (no_grad)', part_block_fwd), ('ECHOBlock + loss fwd+bwd', part_echo),
                 ('ECHOBlock native half (conv+modReLU+ECHO) fwd+bwd', part_echo_desc), ('MLP + loss (torch) fwd+bwd', part_mlp),
                 ('10 x torch.empty', empty_launches)):
    print(f'{name:55s} {enqueue_ms(fn) * 1e3:8.1f} us of host time')
