"""Does a result depend on what the allocator's free memory holds?  Runs the LiftBlock (factor-table path) step with the free blocks filled
with zeros, then with huge values: any difference = a kernel reads memory nobody wrote."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fieldconv_amd.data import sphere_support
from fieldconv_amd.nn import LiftBlock
from fieldconv_amd.transforms import FCPrecomp
dev = torch.device('cuda:0')
N, k, B, R = 300, 20, 2, 6


def fill(val):
    bl = []
    for s in [1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 1 << 10, 512]:
        for _ in range(8):
            bl.append(torch.full((max(s // 4, 1),), val, device=dev))
    del bl
    torch.cuda.synchronize()


def run(use_table):
    data = sphere_support(N, k, seed=N).to(dev)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    lift = sten[..., B:B + 2]
    torch.manual_seed(N)
    mod = LiftBlock(3, 24, n_rings=R, ftype=1).to(dev)
    with torch.no_grad():
        mod.nonlin.bias.uniform_(-0.3, 0.1)
    g = torch.Generator().manual_seed(N)
    pos = torch.randn(N, 3, generator=g).to(dev).requires_grad_(True)
    gy = torch.complex(torch.randn(N, 24, generator=g), torch.randn(N, 24, generator=g)).to(dev)
    y = mod(pos, edges, lift if use_table else sten.columns(0, 2))
    outs = (y.detach(),) + torch.autograd.grad(y, [pos] + list(mod.parameters()), grad_outputs=gy)
    return [o.detach().cpu().clone() for o in outs]


for use_table in (True, False):
    fill(0.0)
    a = run(use_table)
    fill(3e38)
    b = run(use_table)
    fill(float('nan'))
    c = run(use_table)
    for i, (x, y, z) in enumerate(zip(a, b, c)):
        same = torch.equal(x, y)
        nan = bool(torch.isnan(torch.view_as_real(z) if z.is_complex() else z).any())
        print('table' if use_table else 'dense', i, tuple(x.shape), 'equal(zeros, huge)=', same, 'nan with NaN fill=', nan,
              'maxdiff=', float((torch.view_as_real(x) - torch.view_as_real(y)).abs().max()) if x.is_complex() else float((x - y).abs().max()))
