#!/usr/bin/env python3
"""Host enqueue time vs wall time of the partitioned step with a single rank (RCCL path, empty halo): is the multi-GPU
step bound by the host's launch rate?   BENCH_FORCE_DIST-style set-up without the bench's reporting."""
import os
import sys
import time
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
from fieldconv_amd.data import sphere_partition
from fieldconv_amd.dist import HaloPlan, halo_exchange, overlap_backward
from fieldconv_amd.graph import get_graph
from fieldconv_amd.nn import FieldConv
from fieldconv_amd.transforms import FCPrecomp

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=dev)
B, R, C, k, n = 2, 6, 48, 32, 20000
data, n_owned, halo_global, bounds = sphere_partition(n, 1, 0, k=k, seed=0)
data = data.to(dev)
edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
plan = HaloPlan(n_owned, halo_global, bounds, device=dev)
conv = FieldConv(C, C, band_limit=B, n_rings=R).to(dev)
params = list(conv.parameters())
g = torch.Generator().manual_seed(1)
x = torch.complex(torch.randn(n_owned, C, generator=g), torch.randn(n_owned, C, generator=g)).to(dev).requires_grad_(True)
gy = torch.complex(torch.randn(n_owned, C, generator=g), torch.randn(n_owned, C, generator=g)).to(dev)
from fieldconv_amd.dist import GradientBuckets, overlap_forward     # noqa: E402
from fieldconv_amd.graph import FactoredStencil     # noqa: E402
graph = get_graph(edges, sten, data.num_nodes).view()      # hooks go on a per-use view
sten = FactoredStencil.wrap(sten, graph)
mode = sys.argv[1] if len(sys.argv) > 1 else 'dist'
capture = mode == 'dist_graph'          # the partitioned step (both exchanges, bucketed all-reduce) captured in ONE HIP graph
if capture:
    mode = 'dist'
if mode != 'plain':
    graph.restrict_targets(n_owned)
    overlap_backward(graph, plan)
    if mode == 'dist_fwd_overlap':
        overlap_forward(graph, plan, data.n_interior)
buckets = GradientBuckets(params) if mode != 'plain' else None
acc = {}


def lap(name, t):
    now = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (now - t)
    return now


def step():
    t = time.perf_counter()
    if mode == 'plain':
        y = conv(x, edges, sten)
        t = lap('forward', t)
        torch.autograd.grad(y, [x] + params, grad_outputs=gy)
        lap('backward', t)
        return
    xl = halo_exchange(x, plan, deferred=mode == 'dist_fwd_overlap')
    t = lap('halo_exchange', t)
    y = conv(xl, edges, sten)
    t = lap('forward', t)
    buckets.begin()
    x.grad = None
    t = lap('begin', t)
    y.backward(gy)
    t = lap('backward', t)
    buckets.collect()
    t = lap('collect', t)
    buckets.all_reduce()
    lap('all_reduce', t)


if capture:
    from fieldconv_amd.utils import StepGraph

    def captured():
        step()
        return x.grad
    try:
        sg = StepGraph(captured)
    except Exception as exc:                       # noqa: BLE001
        print(f'dist_graph: capture failed: {type(exc).__name__}: {str(exc)[:300]}')
        dist.destroy_process_group()
        sys.exit(0)
    eager_step = step
    step = sg.replay
    mode = 'dist_graph'
for _ in range(300):
    step()
torch.cuda.synchronize()
acc.clear()
n_steps = 200
t0 = time.perf_counter()
for _ in range(n_steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{mode}: host enqueue {(t1 - t0) / n_steps * 1e6:.0f} us/step, wall {(t2 - t0) / n_steps * 1e6:.0f} us/step; host by part: '
      + ', '.join(f'{k} {v / n_steps * 1e6:.0f}' for k, v in acc.items()))
dist.destroy_process_group()
