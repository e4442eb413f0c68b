"""Per-kernel times of one FieldConv layer over channel counts and band limits (HIP events around every launch; development):
looks for shapes that fall onto a slow kernel variant.   python tools/shape_sweep.py [N] [k]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fieldconv_amd.data import sphere_support          # noqa: E402
from fieldconv_amd.functional import kernel_timer      # noqa: E402
from fieldconv_amd.nn import FieldConv                 # noqa: E402
from fieldconv_amd.transforms import FCPrecomp         # noqa: E402

dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 28
R = 6
data = sphere_support(N, k, seed=0, support='p95').to(dev)
print(f'N={N} k={k} R={R}: us per launch (forward / backward data / backward filter), Medges/s of the three together')
for B in (1, 2, 3):
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    E = edges.shape[0]
    for C in (16, 32, 48, 64):
        conv = FieldConv(C, C, band_limit=B, n_rings=R).to(dev)
        x = torch.randn(N, C, dtype=torch.complex64, device=dev, requires_grad=True)
        gy = torch.randn(N, C, dtype=torch.complex64, device=dev)
        params = [x] + list(conv.parameters())
        for _ in range(100):
            torch.autograd.grad(conv(x, edges, sten), params, grad_outputs=gy)
        kernel_timer.reset(pairs=200)
        kernel_timer.stride = 1
        kernel_timer.enabled = True
        for _ in range(40):
            torch.autograd.grad(conv(x, edges, sten), params, grad_outputs=gy)
        torch.cuda.synchronize()
        kernel_timer.enabled = False
        t = {n: sum(v) / len(v) * 1e3 for n, v in kernel_timer.elapsed_ms().items()}
        tot = sum(t.values())
        print(f'  B={B} C={C:2d}: ' + ' / '.join(f'{t.get(n, 0):6.1f}' for n in ('fc_forward', 'fc_backward_data', 'fc_backward_filter'))
              + f'   {E / tot:7.0f}')
