#!/usr/bin/env python3
"""Development: replay one FCResNetBlock case of tools/fuzz/fuzz_components.py (same seed, same generator stream) and compare
OPERATOR BY OPERATOR -- each HIP operator gets the float64 reference's intermediate values as inputs -- to tell a defect of one
kernel from the block's conditioning (modReLU's backward amplifies rounding of its input by 1/|h|).
    CASE=47 python tools/fuzz/replay_block_case.py        (GPU box; seed 31337, 60 cases per family as in the sweep)"""
import os, sys
import numpy as np, torch
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tools','fuzz'))
os.environ['FUZZ_ONLY']='none'
sys.argv=['x','31337','0']
import importlib.util
spec=importlib.util.spec_from_file_location('fz', os.path.join(ROOT,'tools','fuzz','fuzz_components.py'))
fz=importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(fz)
except SystemExit:
    pass
# replay the generator: the resnet cases come after 5 families x 60 cases; simply re-run the families to advance rng identically
fz.rng=np.random.default_rng(31337)
for name, fn in (('echo', fz.fuzz_echo), ('trans_field', fz.fuzz_trans_field), ('precomp+graph', fz.fuzz_precomp_and_graph), ('small-mesh conv', fz.fuzz_small_mesh_conv), ('pointwise', fz.fuzz_pointwise)):
    for c in range(60):
        try: fn()
        except Exception as e: pass
want=int(os.environ.get('CASE','47'))
from fieldconv_amd.nn import FCResNetBlock, FieldConv
from oracle import reference_port_torch as port
from oracle import torch_composites as tc
rng=fz.rng
for c in range(want+1):
    N, k = int(rng.integers(8, 160)), int(rng.integers(3, 30))
    Cin, Cout = int(rng.integers(1, 72)), int(rng.integers(1, 72))
    B, R, ft, front = int(rng.integers(1, 4)), int(rng.integers(2, 9)), int(rng.integers(0, 3)), bool(rng.integers(0, 2))
    data, eps, edges, sten, _, _ = fz.mesh(N, k, B, R)
    N = data.num_nodes
    x, gy = fz.cplx(N, Cin), fz.cplx(N, Cout)
    torch.manual_seed(int(rng.integers(1 << 30)))
    m = FCResNetBlock(Cin, Cout, band_limit=B, n_rings=R, ftype=ft, frontload=front)
    with torch.no_grad():
        m.nonlin1.bias.normal_(0, 0.3); m.nonlin2.bias.normal_(0, 0.3)
    if c < want:
        # the original consumes no further rng in the passing path; failing path neither
        continue
print('case', want, N, k, Cin, Cout, B, R, ft, front, 'E', edges.shape[0])
dev=torch.device('cuda:0')
names=[n for n,_ in m.named_parameters()]
def reference(real, cdt):
    pd = {n_: p.detach().to(real).requires_grad_(True) for n_, p in m.named_parameters()}
    def conv(xx, pre):
        ph = pd.get(pre + '.phase', getattr(getattr(m, pre), 'phase').to(real))
        return port.field_conv(xx, edges, sten.to(cdt), pd[pre + '.zonal'], pd[pre + '.spherical'], ph, ft, B)
    xr = x.to(cdt).requires_grad_(True)
    c1 = conv(xr, 'conv1'); c1.retain_grad()
    h1 = tc.tangent_nonlin(c1, pd['nonlin1.bias']); h1.retain_grad()
    h = conv(h1, 'conv2'); h.retain_grad()
    yr = tc.tangent_nonlin(tc.tangent_lin(xr, pd['res.Re'], pd['res.Im']) + h, pd['nonlin2.bias'])
    yr.backward(gy.to(cdt))
    return dict(c1=c1.detach(), h1=h1.detach(), h=h.detach(), g_c1=c1.grad, g_h1=h1.grad, g_h=h.grad, gx=xr.grad, **{n_: pd[n_].grad for n_ in names})
r64=reference(torch.float64, torch.complex128)
r32=reference(torch.float32, torch.complex64)
def rel2(a,b):
    a=np.asarray(a,dtype=np.complex128); b=np.asarray(b,dtype=np.complex128)
    return float(np.linalg.norm((a-b).ravel())/max(np.linalg.norm(b.ravel()),1e-30))
for kx in ('c1','h1','h','g_h','g_h1','g_c1','gx'):
    print('fp32 port', kx, '%.2e' % rel2(r32[kx].numpy(), r64[kx].numpy()))
# our conv2 alone: input h1 (fp64 ref cast), cotangent g_h (ref cast): gx vs ref g_h1
md=m.to(dev)
h1d=r64['h1'].to(torch.complex64).to(dev).requires_grad_(True)
y2=md.conv2(h1d, edges.to(dev), sten.to(dev))
g=torch.autograd.grad(y2, [h1d]+list(md.conv2.parameters()), grad_outputs=r64['g_h'].to(torch.complex64).to(dev))
print('ours conv2 alone: y', '%.2e' % rel2(y2.detach().cpu().numpy(), r64['h'].numpy()), 'gx', '%.2e' % rel2(g[0].cpu().numpy(), r64['g_h1'].numpy()))
for n_,gg in zip([n for n,_ in md.conv2.named_parameters()], g[1:]):
    print('   conv2.'+n_, '%.2e' % rel2(gg.cpu().numpy(), r64['conv2.'+n_].numpy()), 'fp32 port %.2e' % rel2(r32['conv2.'+n_].numpy(), r64['conv2.'+n_].numpy()))
# our conv1 alone with ref cotangent g_c1
xd=x.to(dev).requires_grad_(True)
y1=md.conv1(xd, edges.to(dev), sten.to(dev))
g1=torch.autograd.grad(y1, [xd]+list(md.conv1.parameters()), grad_outputs=r64['g_c1'].to(torch.complex64).to(dev))
print('ours conv1 alone: y', '%.2e' % rel2(y1.detach().cpu().numpy(), r64['c1'].numpy()))
for n_,gg in zip([n for n,_ in md.conv1.named_parameters()], g1[1:]):
    print('   conv1.'+n_, '%.2e' % rel2(gg.cpu().numpy(), r64['conv1.'+n_].numpy()), 'fp32 port %.2e' % rel2(r32['conv1.'+n_].numpy(), r64['conv1.'+n_].numpy()))
# nonlin1 backward alone: ours with ref inputs
from fieldconv_amd.nn import TangentNonLin
c1d=r64['c1'].to(torch.complex64).to(dev).requires_grad_(True)
h1o=md.nonlin1(c1d)
gn=torch.autograd.grad(h1o,[c1d],grad_outputs=r64['g_h1'].to(torch.complex64).to(dev))[0]
print('ours nonlin1 bwd alone', '%.2e' % rel2(gn.cpu().numpy(), r64['g_c1'].numpy()))
print('min |c1| nonorigin', float(r64['c1'].abs()[r64['c1'].abs()>1e-7].min()), 'min |h1|>0', float(r64['h1'].abs()[r64['h1'].abs()>0].min()))
