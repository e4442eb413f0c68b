#!/usr/bin/env python3
"""Time fc_forward / fc_backward alone on the config-2 shape (HIP events, median of reps)."""
import os
import sys
import ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fieldconv_amd import _lib
from fieldconv_amd.data import random_support, sphere_support
from fieldconv_amd.functional import _csr, _p, _stream, make_dims
from fieldconv_amd.graph import SupportGraph
from fieldconv_amd.transforms import FCPrecomp

kind = sys.argv[1] if len(sys.argv) > 1 else 'geo'
N, k, C, B, R = int(os.environ.get('N', 20000)), int(os.environ.get('K', 32)), int(os.environ.get('C', 48)), int(os.environ.get('B', 2)), int(os.environ.get('R', 6))
dev = torch.device('cuda:0')
data = sphere_support(N, k) if kind == 'geo' else random_support(N, k)


def renumber(data, order):
    """The same mesh with vertex order[j] renamed j (edges regrouped by source, per-edge fields carried along)."""
    import numpy as np
    inv = np.empty(order.size, dtype=np.int64)
    inv[order] = np.arange(order.size)
    e = inv[data.supp_edges.numpy()]
    o = torch.from_numpy(np.argsort(e[:, 0], kind='stable'))
    data.supp_edges = torch.from_numpy(e)[o]
    data.logMag, data.logAng, data.xp = data.logMag[o], data.logAng[o], data.xp[o]
    data.w = data.w[torch.from_numpy(order)]
    return data


def distinct_per_tile(edges, N, by):
    """Mean number of distinct vertices at the other end of the edges of 16 consecutive vertices (by = 1: tiles of targets)."""
    import numpy as np
    e = edges.cpu().numpy()
    key = (e[:, by] // 16) * np.int64(N) + e[:, 1 - by]
    return np.unique(key).size / ((N + 15) // 16)


# ORDER=rcb16: compact 16-vertex tiles (recursive coordinate bisection of the point set); ORDER=random: a random numbering;
# default: the generator's (a Fibonacci lattice: consecutive vertices lie a golden angle apart on a latitude band)
ORDER = os.environ.get('ORDER', '')
if ORDER and kind == 'geo':
    import numpy as np
    from fieldconv_amd.data.synthetic import _fibonacci_sphere, _rcb_order
    if ORDER.startswith('rcb'):
        order, _ = _rcb_order(_fibonacci_sphere(N, 0, N, 0), max(1, N // int(ORDER[3:])))
    else:
        order = np.random.default_rng(0).permutation(N)
    data = renumber(data, order)
print('order %s: distinct sources per tile of 16 targets %.1f, distinct targets per tile of 16 sources %.1f' % (
    ORDER or 'generator', distinct_per_tile(data.supp_edges, N, 1), distinct_per_tile(data.supp_edges, N, 0)))
data = data.to(dev)
edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
graph = SupportGraph(edges, sten, N, allow_factored=os.environ.get('FACT', '1') == '1')
lib = _lib.load()
g = torch.Generator().manual_seed(0)
x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev)
gy = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev)
W = (torch.complex(torch.randn(C, C, R, 2 * B + 1, generator=g), torch.randn(C, C, R, 2 * B + 1, generator=g)) * 0.05).to(dev)
dims = make_dims(graph, C, C, B)
REC = lib.fc_records_flags(ctypes.byref(dims), 1 if (os.environ.get('FACT', '1') == '1' and graph.factored) else 0)
wf = torch.empty(lib.fc_packed_filter_floats_fwd(ctypes.byref(dims), REC), device=dev)
wb = torch.empty(lib.fc_packed_filter_floats_bwd(ctypes.byref(dims), REC), device=dev)
lib.fc_pack_filter(_p(W), _p(wf), _p(wb), ctypes.byref(dims), REC, _stream())
y = torch.empty(N, C, dtype=torch.cfloat, device=dev)
gx = torch.empty_like(x)
gw = torch.empty_like(W)
nb = lib.fc_backward_workspace_bytes(ctypes.byref(dims), REC)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
ct, cs = _csr(graph.rowptr_t, graph.nbr_t, graph.runs_t), _csr(graph.rowptr_s, graph.nbr_s, graph.runs_s)


FACT = os.environ.get('FACT', '1') == '1' and graph.factored


def fwd():
    if FACT:
        lib.fc_forward_factored(_p(x), _p(graph.rec_t), ctypes.byref(ct), _p(wf), _p(y), None, 0, ctypes.byref(dims), None, _stream())
    else:
        lib.fc_forward(_p(x), _p(graph.sten_t), ctypes.byref(ct), _p(wf), _p(y), ctypes.byref(dims), None, _stream())


def bwd_data():
    if FACT:
        lib.fc_backward_data_factored(_p(x), _p(gy), _p(graph.rec_s), ctypes.byref(cs), _p(wb), _p(gx), _p(ws), nb, ctypes.byref(dims), REC, _stream())
    else:
        lib.fc_backward_data(_p(x), _p(gy), _p(graph.sten_s), ctypes.byref(cs), _p(wb), _p(gx), _p(ws), nb, ctypes.byref(dims), _stream())


def bwd_gather():
    lib.fc_backward_gather(_p(gy), _p(graph.rec_s), ctypes.byref(cs), _p(wb), _p(ws), nb, ctypes.byref(dims), _stream())


def bwd_stream():
    lib.fc_backward_stream(_p(x), _p(wb), _p(gx), _p(ws), nb, ctypes.byref(dims), _stream())


def bwd_filter():
    lib.fc_backward_filter(_p(x), _p(ws), nb, ctypes.byref(dims), REC, _stream())


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


# the finishing launch (partial sums + parameter-gradient chain, ftype 1)
zon = torch.randn(C, C, R, generator=g).to(dev)
sphp = torch.randn(C, C, R, B, 2, generator=g).to(dev)
php = torch.randn(C, C, B + 1, generator=g).to(dev)
gz, gs, gp = torch.empty_like(zon), torch.empty_like(sphp), torch.empty_like(php)
fp = _lib.FcFilterParams(zon.data_ptr(), sphp.data_ptr(), php.data_ptr(), 1, gz.data_ptr(), gs.data_ptr(), gp.data_ptr(), None, 0, None)


def finish():
    lib.fc_backward_finish_params(None if os.environ.get('NO_GW') else _p(gw), _p(ws), nb, ctypes.byref(dims), REC, ctypes.byref(fp), _stream())


which = os.environ.get('WHICH', 'fb')
tag = os.environ.get('FC_DEBUG', '0') + '/' + os.environ.get('FC_DEBUG_BWD', '0') + (' factored' if FACT else ' dense')
if 'f' in which:
    print(f'FC_DEBUG={tag} {kind} fwd median/min us: %.1f %.1f' % timeit(fwd))
if 'b' in which:
    print(f'FC_DEBUG={tag} {kind} bwd_data median/min us: %.1f %.1f' % timeit(bwd_data))
    if FACT and lib.fc_backward_streams(ctypes.byref(dims), REC):
        print(f'FC_DEBUG={tag} {kind} bwd_gather median/min us: %.1f %.1f' % timeit(bwd_gather))
        print(f'FC_DEBUG={tag} {kind} bwd_stream(+gx) median/min us: %.1f %.1f' % timeit(bwd_stream))
    print(f'FC_DEBUG={tag} {kind} bwd_filter median/min us: %.1f %.1f' % timeit(bwd_filter))
    print(f'FC_DEBUG={tag} {kind} finish median/min us: %.1f %.1f' % timeit(finish))
    print('gx checksum: %.9e %.9e' % (gx.abs().double().sum().item(), gx[::97].real.double().sum().item()))
