"""Duration of each of the first steps of the config-2 layer after a synchronisation (HIP events around every step)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.build()
from fieldconv_amd.data import sphere_support
from fieldconv_amd.nn import FieldConv
from fieldconv_amd.transforms import FCPrecomp
dev = torch.device('cuda:0')
N, k, C, B, R = 20000, 32, 48, 2, 6
data = sphere_support(N, k, support='p95').to(dev)
edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
conv = FieldConv(C, C, band_limit=B, n_rings=R).to(dev)
x = torch.randn(N, C, dtype=torch.cfloat, device=dev).requires_grad_(True)
gy = torch.randn(N, C, dtype=torch.cfloat, device=dev)
params = list(conv.parameters())
def step():
    y = conv(x, edges, sten)
    torch.autograd.grad(y, [x] + params, grad_outputs=gy)
for rep in range(2):
    torch.cuda.synchronize()
    time.sleep(0.5 if rep else 0)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 60 * 1e3
    ms = [a.elapsed_time(b) for a, b in ev]
    print('rep', rep, 'wall/step %.3f' % wall, ' '.join('%.3f' % m for m in ms[:12]), '...', ' '.join('%.3f' % m for m in ms[-6:]))
