#!/usr/bin/env python3
"""Time ECHOBlock's dense tail (fc_echo_head_forward / _backward) alone: back-to-back calls, microseconds per pass.
   SHAPE=net (config 3: N=1024, D=1392, C=48, Q=8) | dp (config 5: N=4999, D=156, C=16, Q=64); WHICH=f|b|fb; under rocprofv3 --kernel-trace --stats
   the per-kernel averages of one direction."""
import ctypes
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fieldconv_amd import _lib
from fieldconv_amd.functional import _p, _stream

shape = os.environ.get('SHAPE', 'net')
N, D, C, Q = (1024, 1392, 48, 8) if shape == 'net' else (4999, 156, 16, 64)
H1, H2 = 128, 64
dev = torch.device('cuda:0')
lib = _lib.load()
g = torch.Generator().manual_seed(0)
d = torch.rand(N, D, generator=g).to(dev)
x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev)
W = [torch.randn(*s, generator=g).to(dev) * 0.05 for s in ((H1, D), (H1,), (H2, H1), (H2,), (Q, H2), (Q,), (Q, C), (Q,))]
G = [torch.empty_like(w) for w in W]
hp = _lib.FcEchoHeadParams(D, H1, H2, C, Q, *[w.data_ptr() for w in W], *[t.data_ptr() for t in G])
h1, h2, y = torch.empty(N, H1, device=dev), torch.empty(N, H2, device=dev), torch.empty(N, Q, device=dev)
nf, nb = lib.fc_echo_head_forward_workspace_bytes(N, ctypes.byref(hp)), lib.fc_echo_head_backward_workspace_bytes(N, ctypes.byref(hp))
wsf, wsb = torch.empty(max(nf, 16), dtype=torch.uint8, device=dev), torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
gy = torch.randn(N, Q, generator=g).to(dev)
g_d, gx, g_h1 = torch.empty(N, D, device=dev), torch.empty(N, C, dtype=torch.cfloat, device=dev), torch.empty(N, H1, device=dev)


def fwd():
    rc = lib.fc_echo_head_forward(_p(d), _p(x), ctypes.byref(hp), _p(h1), _p(h2), _p(y), _p(wsf), nf, N, _stream())
    assert rc == 0, rc


def bwd():
    rc = lib.fc_echo_head_backward(_p(d), _p(x), _p(h1), _p(h2), _p(gy), ctypes.byref(hp), _p(g_d), _p(gx), _p(g_h1), _p(wsb), nb, N, _stream())
    assert rc == 0, rc


def per_call_us(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


fwd()
which = os.environ.get('WHICH', 'fb')
print(f'{shape}: N={N} D={D} C={C} Q={Q}; workspaces {nf} / {nb} bytes')
if 'f' in which:
    print('forward  %.1f us per pass (3 launches back to back)' % per_call_us(fwd))
if 'b' in which:
    print('backward %.1f us per pass' % per_call_us(bwd))
