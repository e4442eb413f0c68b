"""Forward time of one FieldConv layer against the mesh size, for the kernel family the environment selects (FC_RING=0
frequency-major, FC_RING=2 ring-major at every size, default: ring-major from 257 tiles).  python tools/ring_threshold.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fieldconv_amd.data import sphere_support          # noqa: E402
from fieldconv_amd.nn import FieldConv                 # noqa: E402
from fieldconv_amd.transforms import FCPrecomp         # noqa: E402

dev = torch.device('cuda:0')
k, C, B, R = int(os.environ.get('K', 32)), int(os.environ.get('C', 48)), int(os.environ.get('B', 2)), 6
conv = FieldConv(C, C, band_limit=B, n_rings=R).to(dev)
out = []
for N in (2048, 3072, 4096, 5000, 6000, 6890, 7500, 8192, 10000, 12288):
    data = sphere_support(N, k, seed=0, support='p95').to(dev)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    x = torch.randn(N, C, dtype=torch.complex64, device=dev)
    with torch.no_grad():
        for _ in range(300):
            conv(x, edges, sten)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            conv(x, edges, sten)
        torch.cuda.synchronize()
    out.append('%d:%.1f' % (N, (time.perf_counter() - t0) / 300 * 1e6))
print('FC_RING=%s forward us (incl. 9 us filter packing): ' % os.environ.get('FC_RING', 'default') + '  '.join(out))
