#!/usr/bin/env python3
"""Benchmark of the FieldConv hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one FieldConv layer forward + backward (input gradient, filter-parameter gradients, the
filter assembly and its autograd chain included) on a synthetic sphere mesh of 20 000 vertices per
GPU, k = 32 in-neighbours, C = 48 -> 48 channels, band_limit 2, n_rings 6, ftype 1, fp32 -- the
shape BASELINE.json's metric is quoted on.  With N > 1 (launched by torch.distributed.run, one rank
per GPU) the mesh has N x 20 000 vertices, is partitioned into N latitude bands, and every step
also runs the one-hop halo exchange (forward and transposed) and the all-reduce of the parameter
gradients over RCCL: weak scaling.  Inputs are resident in HBM before the timed region; support
graph preprocessing (CSR by target / by source, stencil permutation) is done once outside it and
reported separately, as it is shared by every convolution of a network.

Rank 0 prints one JSON line (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32 MFMA == fp32 vector peak on gfx950


def algorithmic_bytes(N, E, I, O, R, F):
    """Compulsory traffic of the operator contract with int32 indices, everything touched once
    (SURVEY.md 8(d) / BASELINE.md section 4)."""
    fwd = E * (8 * R * F + 4) + 4 * N + 8 * N * (I + O) + 8 * O * I * R * F
    bwd = E * (8 * R * F + 8) + 8 * N + 8 * N * (2 * I + O) + 16 * O * I * R * F
    return fwd, bwd


def algorithmic_flops(N, E, I, O, R, F):
    gather = 8 * E * I * R * F
    gemm = 8 * N * O * I * R * F
    return gather + gemm, gather + 2 * gemm       # fwd, bwd (H gather + two contractions)


def cpu_baseline(B, R, C, k, threads):
    """Reference algorithm (oracle/reference_port_torch.py) on the host cores, bounded sample."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import FieldConv
    from oracle import reference_port_torch as port
    from oracle.torch_composites import FCPrecomp              # the CPU leg's stencil comes from the oracle as well
    n_s = 2500
    torch.set_num_threads(threads)
    data = sphere_support(n_s, k=k, seed=1)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    g = torch.Generator().manual_seed(1)
    x = torch.complex(torch.randn(n_s, C, generator=g), torch.randn(n_s, C, generator=g)).requires_grad_(True)
    gy = torch.complex(torch.randn(n_s, C, generator=g), torch.randn(n_s, C, generator=g))
    conv = FieldConv(C, C, band_limit=B, n_rings=R, ftype=1)       # parameter container only (CPU)
    params = [conv.zonal, conv.spherical, conv.phase]

    def step():
        y = port.field_conv(x, edges, sten, conv.zonal, conv.spherical, conv.phase, 1, B)
        torch.autograd.grad(y, [x] + params, grad_outputs=gy)
    step()
    best = float('inf')
    for _ in range(4):                     # ~2 s per step on 128 cores: ~10 s of CPU work in all
        t0 = time.perf_counter()
        step()
        best = min(best, time.perf_counter() - t0)
    E = edges.shape[0]
    return {'value': E / best / 1e6, 'unit': 'Medges/s', 'cores': threads, 'kind': 'port',
            'sample': f'reference-structured torch CPU port (oracle/reference_port_torch.py), one FieldConv fwd+bwd on a '
                      f'{n_s}-vertex sphere mesh, k={k}, C={C}, B={B}, R={R} (E={E}); best of 4 after 1 warm-up, '
                      f'{best:.2f} s per step'}


def reduced_precision_run(args, conv, x, edges, sten, step):
    """Extra, reported separately (BASELINE configs[1] names bf16/fp32): the same workload with the contractions
    on single f16 halves (FC_MFMA=f16, fp32 accumulation) in a child process -- the mode is fixed per process --
    and its deviation from this process's fp32-grade result on the same seeded inputs."""
    import subprocess
    import tempfile
    try:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, 'rp.pt')
            cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(args.steps), '--warmup', str(args.warmup),
                   '--verts', str(args.verts), '--k', str(args.k), '--channels', str(args.channels), '--band-limit',
                   str(args.band_limit), '--n-rings', str(args.n_rings), '--support', args.support, '--no-cpu-baseline', '--no-extras', '--dump', path]
            env = dict(os.environ, FC_MFMA='f16')
            res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            child = json.loads(res.stdout.strip().splitlines()[-1])
            ref = torch.load(path)
        gx = step()[0].detach().cpu()
        y = conv(x, edges, sten).detach().cpu()
        err = lambda a, b: float((a - b).abs().max() / b.abs().max())
        return {'mfma': 'single f16 halves with per-row power-of-two scales, fp32 accumulation (FC_MFMA=f16)',
                'value': child['value'], 'unit': child['unit'], 'ms_per_step': child['ms_per_step'],
                'max_rel_err_y_vs_default': err(ref['y'], y), 'max_rel_err_gx_vs_default': err(ref['gx'], gx)}
    except Exception as exc:
        return {'value': None, 'note': f'failed: {type(exc).__name__}: {exc}'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)       # 0.5 ms each: long enough for steady clocks and a full launch queue
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--verts', type=int, default=20000, help='vertices per GPU')
    ap.add_argument('--k', type=int, default=32)
    ap.add_argument('--channels', type=int, default=48)
    ap.add_argument('--band-limit', type=int, default=2)
    ap.add_argument('--n-rings', type=int, default=6)
    ap.add_argument('--support', default='p95', choices=['p95', 'all'],
                    help="support radius: 'p95' = 95-percentile of the k-NN distances (SURVEY 8(d) G-geo: FCPrecomp drops 5 %% of the "
                         "edges, every ring populated); 'all' = above every k-NN distance (round-1 mesh: E = N*k, outer rings empty)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='only the metric of record (used by the reduced-precision child run)')
    ap.add_argument('--dump', default=None, help='write y and gx of one step to this file (reduced-precision child run)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f'--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}')
        args.gpus = world
    backend = os.environ.get('BENCH_BACKEND', 'nccl')                  # "gloo": several ranks on one GPU (test rigs only)
    if backend == 'gloo':
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    force_dist = os.environ.get('BENCH_FORCE_DIST', '0') == '1'        # exercise the RCCL path with a single rank
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        # RCCL prints a version banner to stdout when the communicator is created; keep stdout for the JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == 'nccl':
                dist.init_process_group('nccl', device_id=dev)
            else:
                dist.init_process_group(backend)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)          # the banner sits in C stdio's buffer
            except Exception:
                pass
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    import __graft_entry__
    __graft_entry__.build()
    from fieldconv_amd.data import sphere_partition
    from fieldconv_amd.dist import HaloPlan, halo_exchange, overlap_backward
    from fieldconv_amd.functional import kernel_timer
    from fieldconv_amd.graph import get_graph
    from fieldconv_amd.nn import FieldConv
    from fieldconv_amd.transforms import FCPrecomp

    B, R, C, k = args.band_limit, args.n_rings, args.channels, args.k
    F = 2 * B + 1
    n_total = args.verts * world
    data, n_owned, halo_global, bounds = sphere_partition(n_total, world, rank, k=k, seed=0, support=args.support)
    data = data.to(dev)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    n_local = data.num_nodes
    E = int(edges.shape[0])
    plan = HaloPlan(n_owned, halo_global, bounds, device=dev) if use_dist else None

    torch.manual_seed(1234)                                  # identical parameters on every rank
    conv = FieldConv(C, C, band_limit=B, n_rings=R, ftype=1).to(dev)
    params = list(conv.parameters())
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.complex(torch.randn(n_owned, C, generator=g), torch.randn(n_owned, C, generator=g))
    x[torch.rand(n_owned, C, generator=g) < 0.01] = 0        # 1 % exact zeros (origin-box path)
    x = x.to(dev).requires_grad_(True)
    gy = torch.complex(torch.randn(n_owned, C, generator=g), torch.randn(n_owned, C, generator=g)).to(dev)

    from fieldconv_amd.graph import SupportGraph
    SupportGraph(edges, sten, n_local)                      # first build pays one-off library initialisation
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    SupportGraph(edges, sten, n_local)                      # steady-state cost of the per-mesh preprocessing
    torch.cuda.synchronize()
    prep_ms = (time.perf_counter() - t0) * 1e3
    mesh_graph = get_graph(edges, sten, n_local)            # the cached instance every convolution will use
    if plan is not None and os.environ.get('BENCH_NO_OVERLAP', '0') != '1':
        overlap_backward(mesh_graph, plan)                  # gradient halo exchange under the filter-gradient kernel

    def _all_reduce(t, op=dist.ReduceOp.SUM):
        if backend == 'gloo':                       # host-staged (test rigs only)
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def step():
        xl = halo_exchange(x, plan) if plan is not None else x
        y = conv(xl, edges, sten)
        if plan is not None:
            y = y[:n_owned]
        grads = torch.autograd.grad(y, [x] + params, grad_outputs=gy)
        if use_dist:
            flat = torch.cat([t.reshape(-1) for t in grads[1:]])
            _all_reduce(flat)
        return grads

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    kernel_timer.reset(pairs=3 * (args.steps // 4 + 1))
    kernel_timer.stride = 4            # every 4th launch of each kernel inside the timed region carries a HIP-event pair
    kernel_timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_timer.enabled = False
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        _all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ecount = torch.tensor([E], device=dev, dtype=torch.int64)
        _all_reduce(ecount)
        E_total = int(ecount.item())
    else:
        E_total = E
    ms_per_step = elapsed / args.steps * 1e3
    value = E_total / (elapsed / args.steps) / 1e6

    if rank == 0:
        kt = {k_: sum(v) / len(v) for k_, v in kernel_timer.elapsed_ms().items() if v}
        fwd_b, bwd_b = algorithmic_bytes(n_local, E, C, C, R, F)
        fwd_f, bwd_f = algorithmic_flops(n_local, E, C, C, R, F)
        per_kernel = {}
        gemm_f = 8 * n_local * C * C * R * F
        wbytes = 8 * C * C * R * F
        # backward contract bytes split over its two kernels: the data kernel reads stencil, indices, gy, x, W and
        # writes gx; the filter kernel's contract traffic is x and the filter gradient (its H input is a temporary)
        for name, nbytes, nflops in (('fc_forward', fwd_b, fwd_f), ('fc_backward_data', bwd_b - wbytes - 8 * n_local * C, bwd_f - gemm_f),
                                     ('fc_backward_filter', wbytes + 8 * n_local * C, gemm_f)):
            if name in kt:
                sec = kt[name] * 1e-3
                per_kernel[name] = {'avg_ms': kt[name], 'algorithmic_bytes': nbytes, 'GBps': nbytes / sec / 1e9,
                                    'hbm_frac': nbytes / sec / 1e9 / HBM_PEAK_GBS, 'algorithmic_flops': nflops,
                                    'TFLOPs': nflops / sec / 1e12, 'mfma_f32_frac': nflops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS}
        dom = max(per_kernel, key=lambda n: per_kernel[n]['avg_ms']) if per_kernel else None
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if dom and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        roofline = None
        if dom:
            roofline = {'bound': 'hbm', 'kernel': dom + '_kernel', 'achieved': per_kernel[dom]['GBps'], 'peak': HBM_PEAK_GBS,
                        'unit': 'GB/s', 'frac': per_kernel[dom]['hbm_frac'], 'traffic': traffic,
                        'avg_launch_ms': per_kernel[dom]['avg_ms'], 'algorithmic_bytes_per_launch': per_kernel[dom]['algorithmic_bytes'],
                        'mfma_f32_frac': per_kernel[dom]['mfma_f32_frac']}
        out = {
            'metric': 'FieldConv fwd+bwd Medges/s (20k verts, k=32, C=48, M=2)',
            'value': value, 'unit': 'Medges/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[1] shape: one FieldConv layer fwd+bwd on a synthetic sphere mesh, '
                                   f'{args.verts} verts/GPU, k={k}, C={C}->{C}, band_limit={B}, n_rings={R}, ftype=1',
                       'verts_per_gpu': args.verts, 'edges_total': E_total, 'k': k, 'channels': C, 'band_limit': B, 'n_rings': R,
                       'parallelism': 'single GPU' if world == 1 else f'vertex partition x{world}, one-hop halo over RCCL',
                       'halo_rows_rank0': 0 if plan is None else plan.n_halo},
            'roofline': roofline,
            'kernels': per_kernel,
            'hbm_frac_fwd_bwd': (fwd_b + bwd_b) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'hbm_frac_kernels': (fwd_b + bwd_b) / (sum(v['avg_ms'] for v in per_kernel.values()) * 1e-3) / 1e9 / HBM_PEAK_GBS if per_kernel else None,
            'graph_preprocessing_ms': prep_ms,
        }
        if args.dump:
            gd = step()
            torch.save({'y': conv(x, edges, sten).detach().cpu(), 'gx': gd[0].detach().cpu()}, args.dump)
        if world == 1 and not use_dist and not args.no_extras:
            # extra (not the metric of record): one FCResNetBlock = 2 FieldConv + TangentLin + 2 modReLU, fwd+bwd
            from fieldconv_amd.nn import FCResNetBlock
            blk = FCResNetBlock(C, C, band_limit=B, n_rings=R, ftype=1).to(dev)
            bparams = list(blk.parameters())

            def bstep():
                yb = blk(x, edges, sten)
                torch.autograd.grad(yb, [x] + bparams, grad_outputs=gy)
            for _ in range(10):
                bstep()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(50):
                bstep()
            torch.cuda.synchronize()
            bms = (time.perf_counter() - tb) / 50 * 1e3
            out['fc_resnet_block'] = {'ms_per_step': bms, 'medges_per_s': 2 * E / (bms * 1e-3) / 1e6,
                                      'note': 'FCResNetBlock fwd+bwd, edges counted once per FieldConv (2 per block)'}
        if world == 1 and not use_dist and not args.no_extras and os.environ.get('FC_MFMA') is None:
            out['reduced_precision'] = reduced_precision_run(args, conv, x, edges, sten, step)
        if world == 1 and not args.no_cpu_baseline:
            ncpu = os.cpu_count() or 2
            threads = max(1, ncpu // 2)                 # physical cores (SMT siblings excluded)
            try:
                out['cpu_baseline'] = cpu_baseline(B, R, C, k, threads)
            except Exception as exc:                   # the GPU numbers stand on their own
                out['cpu_baseline'] = {'value': None, 'unit': 'Medges/s', 'cores': threads, 'kind': 'port',
                                       'sample': f'failed: {type(exc).__name__}: {exc}'}
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
