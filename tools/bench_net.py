#!/usr/bin/env python3
"""Config 3: the segmentation network's topology (LiftBlock(3->48), four FCResNetBlocks, ECHOBlock(48->8))
forward + loss + backward on a synthetic mesh of 1024 vertices with ~128 neighbours each; ms per step."""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fieldconv_amd.data import sphere_support
from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock
from fieldconv_amd.transforms import FCPrecomp

N, k, nf, B, R, n_classes = int(os.environ.get('N', 1024)), int(os.environ.get('K', 128)), 48, 2, 6, 8
dev = torch.device('cuda:0')
data = sphere_support(N, k).to(dev)
pre = FCPrecomp(B, R, data.epsilon)
mods = torch.nn.ModuleDict(dict(
    lift=LiftBlock(3, nf, n_rings=R, ftype=1),
    r1=FCResNetBlock(nf, nf, band_limit=B, n_rings=R), r2=FCResNetBlock(nf, nf, band_limit=B, n_rings=R),
    r3=FCResNetBlock(nf, nf, band_limit=B, n_rings=R), r4=FCResNetBlock(nf, nf, band_limit=B, n_rings=R),
    echo=ECHOBlock(nf, n_classes, n_des=48, n_bins=3, band_limit=B, n_rings=R))).to(dev)
params = list(mods.parameters())
g = torch.Generator().manual_seed(0)
pos = torch.randn(N, 3, generator=g).to(dev)
labels = torch.randint(0, n_classes, (N,), generator=g).to(dev)


OPT = os.environ.get('OPT', '0') in ('1', 'fused')            # 1: torch.optim.Adam (capturable), fused: fieldconv_amd.optim.FusedAdam
if os.environ.get('OPT') == 'fused':
    from fieldconv_amd.optim import FusedAdam
    opt = FusedAdam(params, lr=1e-3)
elif OPT:
    opt = torch.optim.Adam(params, lr=1e-3, capturable=True, foreach=True)
    for p_ in params:
        p_.grad = torch.zeros_like(p_)


def step():
    edges, sten, ln, wxp = pre(data)                       # runs every forward in the reference, too
    x = mods['lift'](pos, edges, sten[..., B:B + 2])
    for name in ('r1', 'r2', 'r3', 'r4'):
        x = mods[name](x, edges, sten)
    logits = mods['echo'](x, edges, sten, ln, wxp)
    loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), labels)
    if OPT:                                             # a complete training step: gradients into .grad, Adam update
        opt.zero_grad(set_to_none=os.environ.get('SET_TO_NONE', '1') == '1')
        loss.backward()
        opt.step()
        return (loss.detach(),)
    return (loss,) + torch.autograd.grad(loss, params)


if os.environ.get('GRAPH', '0') == '1':             # the whole step as one HIP graph
    from fieldconv_amd.utils import StepGraph
    eager = [t.detach().clone() for t in step()]
    graphed = StepGraph(step)
    step = graphed.replay
    if not OPT:             # (with the optimizer in the graph every replay starts from different parameters)
        same = all(torch.equal(a, b) for a, b in zip(eager, step()))
        print('hipGraph replay bit-identical to eager:', same)
for _ in range(10):
    step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f'segmentation net N={N} k={k}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per fwd+bwd step')
