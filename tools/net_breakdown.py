import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from fieldconv_amd.data import sphere_support
from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock
from fieldconv_amd.transforms import FCPrecomp
from fieldconv_amd.graph import get_graph
N, k, nf, B, R, n_classes = 1024, 128, 48, 2, 6, 8
dev = torch.device('cuda:0')
data = sphere_support(N, k).to(dev)
pre = FCPrecomp(B, R, data.epsilon)
lift = LiftBlock(3, nf, n_rings=R, ftype=1).to(dev)
blk = FCResNetBlock(nf, nf, band_limit=B, n_rings=R).to(dev)
echo = ECHOBlock(nf, n_classes, n_des=48, n_bins=3, band_limit=B, n_rings=R).to(dev)
g = torch.Generator().manual_seed(0)
pos = torch.randn(N, 3, generator=g).to(dev).requires_grad_(True)
def T(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
edges, sten, ln, wxp = pre(data)
print('FCPrecomp            %.2f ms' % T(lambda: pre(data)))
def build():
    e, s, _, _ = pre(data); get_graph(e, s, N)
print('FCPrecomp + graph    %.2f ms' % T(build))
x0 = lift(pos, edges, sten[..., B:B + 2]).detach().requires_grad_(True)
def f_lift():
    y = lift(pos, edges, sten[..., B:B + 2]); torch.autograd.grad(y, [pos] + list(lift.parameters()), grad_outputs=torch.ones_like(y))
print('LiftBlock fwd+bwd    %.2f ms' % T(f_lift))
def f_blk():
    y = blk(x0, edges, sten); torch.autograd.grad(y, [x0] + list(blk.parameters()), grad_outputs=torch.ones_like(y))
print('FCResNetBlock f+b    %.2f ms' % T(f_blk))
def f_echo():
    y = echo(x0, edges, sten, ln, wxp); torch.autograd.grad(y.sum(), [x0] + list(echo.parameters()), allow_unused=True)
print('ECHOBlock fwd+bwd    %.2f ms' % T(f_echo))
