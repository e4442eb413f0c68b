"""Development: the ring-major backward kernels against the oracle over a list of shapes (run with FC_BWD_RING=2 so that small
meshes take them too).  Usage on the GPU box: FC_BWD_RING=2 python tools/check_ring_bwd.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import fieldconv_oracle as orc          # noqa: E402  (checker)
from test_gpu_fullsize import precomp_case, run_conv   # noqa: E402
from fieldconv_amd.graph import SupportGraph         # noqa: E402
from fieldconv_amd import _lib                       # noqa: E402
from fieldconv_amd.functional import make_dims       # noqa: E402
import ctypes                                        # noqa: E402

dev = torch.device('cuda:0')
shapes = [(96, 10, 8, 6, 2, 6), (96, 10, 8, 8, 2, 6), (200, 9, 16, 16, 2, 6), (200, 9, 48, 48, 2, 6), (333, 7, 24, 40, 1, 4),
          (500, 8, 12, 16, 2, 6), (700, 8, 32, 32, 3, 6), (100, 6, 6, 8, 2, 3), (1000, 12, 48, 8, 2, 6), (1000, 12, 8, 48, 2, 6)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1].split(','))]


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


for (N, k, I, O, B, R) in shapes:
    edges, sten, x, gy, W = precomp_case(N, k, I, O, B, R, seed=N + I)
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    flags = _lib.load().fc_records_flags(ctypes.byref(make_dims(graph, I, O, B)), 1 if graph.factored else 0)
    y, gx, gW = run_conv(graph, x, W, gy, dev)
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    egw = np.abs(gW.cpu().numpy() - gW_ref).max(axis=(1, 2))        # per o
    print(f'N={N} k={k} I={I} O={O} B={B} R={R} flags={flags}: gx {rel(gx.cpu().numpy(), gx_ref):.2e}  gW {rel(gW.cpu().numpy(), gW_ref):.2e}',
          ' worst o', int(egw.argmax()), ' per f', ['%.1e' % v for v in np.abs(gW.cpu().numpy() - gW_ref).max(axis=(0, 1, 2)) / np.abs(gW_ref).max()],
          ' per r', ['%.1e' % v for v in np.abs(gW.cpu().numpy() - gW_ref).max(axis=(0, 1, 3)) / np.abs(gW_ref).max()], flush=True)

# the segmentation-net fixture's graph (ragged random graph, reference FCPrecomp stencil)
from conftest import load_golden   # noqa: E402
c = load_golden('net.npz')['segmentation_net']
edges, sten = torch.from_numpy(c['edges']), torch.from_numpy(c['sten'])
N = c['pos'].shape[0]
B, R = int(c['B']), int(c['R'])
for (I, O, zero_frac) in ((8, 6, 0.0), (8, 6, 0.5), (8, 8, 0.3), (48, 48, 0.3)):
    g = torch.Generator().manual_seed(I * 100 + O)
    x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
    x[torch.rand(N, I, generator=g) < zero_frac] = 0
    gy = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
    W = torch.complex(torch.randn(O, I, R, 2 * B + 1, generator=g), torch.randn(O, I, R, 2 * B + 1, generator=g)) / (I * R) ** 0.5
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    y, gx, gW = run_conv(graph, x, W, gy, dev)
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    deg_s = torch.bincount(edges[:, 0], minlength=N)
    d = np.abs(gW.cpu().numpy() - gW_ref) / np.abs(gW_ref).max()
    print(f'net graph I={I} O={O} zeros={zero_frac}: factored={graph.factored} gx {rel(gx.cpu().numpy(), gx_ref):.2e} gW {rel(gW.cpu().numpy(), gW_ref):.2e}',
          'per r', ['%.1e' % v for v in d.max(axis=(0, 1, 3))], 'per f', ['%.1e' % v for v in d.max(axis=(0, 1, 2))],
          'min/max out-degree', int(deg_s.min()), int(deg_s.max()), flush=True)
