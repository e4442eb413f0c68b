"""Development: every FieldConv backward call inside the segmentation-net golden step, its gW against the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import fieldconv_oracle as orc          # noqa: E402
from conftest import load_golden                     # noqa: E402
import test_gpu_parity as T                          # noqa: E402
from fieldconv_amd import functional as Fn           # noqa: E402

dev = torch.device('cuda:0')
c = load_golden('net.npz')['segmentation_net']
calls = []
orig = Fn._launch_backward


def spy(lib, x, gy, graph, wpk_b, plan, wshape, st, params=None):
    out = orig(lib, x, gy, graph, wpk_b, plan, wshape, st, params=params)
    calls.append((x.detach().cpu().numpy().copy(), gy.detach().cpu().numpy().copy(), out[0].detach().cpu().numpy().copy(),
                  out[1].detach().cpu().numpy().copy(), wshape, plan.records))
    return out


Fn._launch_backward = spy
which = sys.argv[1] if len(sys.argv) > 1 else 'segmentation'
try:
    (T.test_segmentation_net_golden if which == 'segmentation' else T.test_correspondence_net_golden)(dev)
except AssertionError as e:
    print('test assertion:', str(e)[:200])
edges, sten = c['edges'], c['sten']
if which != 'segmentation':
    from oracle.torch_composites import FCPrecomp
    c = load_golden('net_correspondence.npz')['correspondence_net']

    class M:
        pass
    d = M()
    d.logMag, d.logAng, d.w, d.supp_edges, d.xp = (torch.from_numpy(np.ascontiguousarray(c[k])) for k in ('logMag', 'logAng', 'w', 'edges', 'xp'))
    e_, s_, _, _ = FCPrecomp(int(c['B']), int(c['R']), float(c['eps']))(d)
    edges, sten = e_.numpy(), s_.numpy()
for n, (x, gy, gx, gw, wshape, rec) in enumerate(calls):
    O, I, R, F = wshape
    W0 = np.zeros(wshape, dtype=np.complex64)
    _, gW_ref = orc.fieldconv_backward(x, edges, sten, W0, gy)
    d = np.abs(gw - gW_ref) / np.abs(gW_ref).max()
    print(f'call {n}: I={I} O={O} records={rec} gW err {d.max():.2e}  |x| range {np.abs(x).min():.1e}..{np.abs(x).max():.1e} zeros {float((x == 0).mean()):.2f}'
          f'  |gy| max {np.abs(gy).max():.1e}', 'per r', ['%.0e' % v for v in d.max(axis=(0, 1, 3))], 'per i', ['%.0e' % v for v in d.max(axis=(0, 2, 3))][:8], flush=True)
