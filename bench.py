#!/usr/bin/env python3
"""Benchmark of the FieldConv hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode layer|dp|net] [--support p95|all]

--mode layer (default, the metric of record).  A step = one FieldConv layer forward + backward (input gradient,
filter-parameter gradients, the filter assembly and its autograd chain included) on a synthetic sphere mesh of
20 000 vertices per GPU, k = 32 nearest neighbours, support radius = 95-percentile of the k-NN distances (SURVEY 8(d)
G-geo: FCPrecomp drops the longest 5 % of the edges, every ring populated), C = 48 -> 48 channels, band_limit 2,
n_rings 6, ftype 1 -- the shape BASELINE.json's metric is quoted on.  With N > 1 (launched by torch.distributed.run,
one rank per GPU) the mesh has N x 20 000 vertices, is partitioned into N compact patches, and every step also runs
the one-hop halo exchange (forward and transposed) and one bucketed all-reduce of the parameter gradients over RCCL:
weak scaling.

--mode dp (BASELINE configs[4]).  Every rank holds its own FAUST-sized mesh (4 999 vertices, k = 28) and a replica of
the correspondence network's topology (reference correspondence.ipynb: LiftBlock, eight FCResNetBlocks with
TangentPerceptron meta-residuals, ECHOBlock) at C = 64, band_limit 3; a step = forward + loss + backward + ONE
bucketed all-reduce of all parameter gradients (fieldconv_amd.dist.GradientBuckets).  value = edges x convolutions per
second over all ranks.

--mode net (BASELINE configs[2]).  The segmentation network's topology (reference segmentation.ipynb: LiftBlock, four
FCResNetBlocks, ECHOBlock) forward + loss + backward on a 1 024-vertex mesh with ~128 neighbours, launched eagerly through
the block-level entry points (one foreign call per block and pass); extras: the step replayed as one HIP graph, composed
of per-operator calls, and on a different mesh every step with the preprocessing inside.

Inputs are resident in HBM before the timed region; support-graph preprocessing is done once outside it and reported
separately, as it is shared by every convolution of a network.  Rank 0 prints one JSON line (README / DESIGN.md).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32 MFMA == fp32 vector peak on gfx950
DTYPE = 'f32 (complex64 in/out; contractions on f16 MFMA with every operand split into two halves, fp32 accumulate)'


def algorithmic_bytes(N, E, I, O, R, F):
    """Compulsory traffic of the operator contract with int32 indices, everything touched once
    (SURVEY.md 8(d) / BASELINE.md section 4)."""
    fwd = E * (8 * R * F + 4) + 4 * N + 8 * N * (I + O) + 8 * O * I * R * F
    bwd = E * (8 * R * F + 8) + 8 * N + 8 * N * (2 * I + O) + 16 * O * I * R * F
    return fwd, bwd


def algorithmic_flops(N, E, I, O, R, F, factored):
    """Real FLOPs of what the kernels evaluate.  Gather: dense stencil = R*F complex multiply-adds per (edge, channel)
    (8 FLOP each); record-driven = (2+2B) complex products (6) for the rotated, phased copies plus 2F real-times-complex
    multiply-adds (4) for the two rings.  Contraction: N*O*I*R*F complex multiply-adds, once forward, twice backward."""
    B = (F - 1) // 2
    gather = E * I * (((2 + 2 * B) * 6 + 2 * F * 4) if factored else 8 * R * F)
    gemm = 8 * N * O * I * R * F
    return gather + gemm, gather + 2 * gemm


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(args, threads, shape=None, slab=4000):
    """The reference's algorithm (oracle/reference_port_torch.py: materialised (E,C,R,F) product, index-add, broadcast
    multiply-and-sum, torch autograd) on the host cores at the METRIC'S OWN config: the 20 000-vertex mesh, processed in
    target slabs of 4 000 vertices because the reference's temporaries need ~32 GB for the whole mesh (BASELINE.md
    section 3); every slab is a full forward + backward of its targets.  Plus a one-thread figure on one 1 000-target slab."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import FieldConv
    from oracle import reference_port_torch as port
    from oracle.torch_composites import FCPrecomp              # the CPU leg's stencil comes from the oracle as well
    B, R, C, k, N = shape or (args.band_limit, args.n_rings, args.channels, args.k, args.verts)
    data = sphere_support(N, k=k, seed=0, support=args.support)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    g = torch.Generator().manual_seed(1)
    x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g))
    gy = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g))
    conv = FieldConv(C, C, band_limit=B, n_rings=R, ftype=1)       # parameter container only (CPU)
    params = [conv.zonal, conv.spherical, conv.phase]
    order = torch.argsort(edges[:, 1], stable=True)
    e_t, s_t = edges[order], sten[order]
    rowptr = torch.searchsorted(e_t[:, 1].contiguous(), torch.arange(N + 1))

    def run(slab, lo_hi):
        total_e, t0 = 0, time.perf_counter()
        for lo in range(lo_hi[0], lo_hi[1], slab):
            hi = min(lo + slab, lo_hi[1])
            a, b = int(rowptr[lo]), int(rowptr[hi])
            es = e_t[a:b].clone()
            es[:, 1] -= lo
            xs = x.clone().requires_grad_(True)
            y = port.field_conv(xs, es, s_t[a:b], conv.zonal, conv.spherical, conv.phase, 1, B, n_out=hi - lo)
            torch.autograd.grad(y, [xs] + params, grad_outputs=gy[lo:hi])
            total_e += b - a
        return total_e, time.perf_counter() - t0

    torch.set_num_threads(threads)
    run(1000, (0, 1000))                                        # warm-up (allocator, thread pool)
    e_all, t_all = run(slab, (0, N))
    torch.set_num_threads(1)
    e_one, t_one = run(1000, (0, 1000))
    torch.set_num_threads(threads)
    return {'value': e_all / t_all / 1e6, 'unit': 'Medges/s', 'cores': threads, 'kind': 'port', 'cpu_model': cpu_model(),
            'one_thread': {'value': e_one / t_one / 1e6, 'unit': 'Medges/s', 'sample': f'targets 0..999 of the same mesh ({e_one} edges), '
                                                                                        f'{t_one:.1f} s'},
            'sample': f'reference-structured torch CPU port (oracle/reference_port_torch.py), one FieldConv fwd+bwd over the whole '
                      f'{N}-vertex mesh (E={e_all}, k={k}, C={C}, B={B}, R={R}) in target slabs of {slab} vertices, {threads} threads: '
                      f'{t_all:.1f} s'}


def env_report():
    """Every development switch of the package / the library / this script that is set in this process (fieldconv_amd/_env.py)."""
    from fieldconv_amd import _env
    return _env.active()


def describe_kernels(graph, I, O, B):
    """The library's own account of the kernels a FieldConv forward + backward over this graph launches (fc_describe_kernels)."""
    import ctypes
    from fieldconv_amd import _lib
    from fieldconv_amd.functional import make_dims
    buf = ctypes.create_string_buffer(1024)
    kind = 2 if getattr(graph, 'geo_t', None) is not None else (1 if graph.factored else 0)
    rc = _lib.load().fc_describe_kernels(ctypes.byref(make_dims(graph, I, O, B)), kind, buf, len(buf))
    return buf.value.decode() if rc == 0 else f'fc_describe_kernels failed ({rc})'


def child_run(args, env_extra, extra_args=(), dump=True):
    """The same workload in a child process (library switches are fixed per process); returns (json line, dumped y / gx)."""
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'out.pt')
        cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(args.steps), '--warmup', str(args.warmup),
               '--verts', str(args.verts), '--k', str(args.k), '--channels', str(args.channels), '--band-limit',
               str(args.band_limit), '--n-rings', str(args.n_rings), '--support', args.support, '--no-cpu-baseline', '--no-extras']
        cmd += list(extra_args)
        if dump:
            cmd += ['--dump', path]
        env = {k_: v for k_, v in os.environ.items() if not k_.startswith('FC_')}
        env.update(env_extra)
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        line = json.loads(res.stdout.strip().splitlines()[-1])
        ref = torch.load(path) if dump else None
    return line, ref


def other_mode(args, env_extra, what, y, gx):
    """Extra, reported separately: the same workload in another arithmetic mode of the library, and the deviation of its
    result from this process's default-mode result on the same seeded inputs."""
    try:
        child, ref = child_run(args, env_extra)
        err = lambda a, b: float((a - b).abs().max() / b.abs().max())
        return {'mfma': what, 'value': child['value'], 'unit': child['unit'], 'ms_per_step': child['ms_per_step'],
                'kernel_us': {k_: round(v['avg_ms'] * 1e3, 1) for k_, v in child.get('kernels', {}).items()},
                'max_rel_dev_y_vs_default': err(ref['y'], y), 'max_rel_dev_gx_vs_default': err(ref['gx'], gx)}
    except Exception as exc:
        return {'value': None, 'note': f'failed: {type(exc).__name__}: {exc}'}


def mode_leg(mode, keep):
    """`python bench.py --mode <mode>` (25 steps after 5, with its own bounded CPU baseline) in a child process -> the compact form of
    its line: value, ms per step (literal and settled), dominant-kernel roofline fraction, CPU baseline, and the extras named in `keep`."""
    try:
        cmd = [sys.executable, os.path.abspath(__file__), '--mode', mode, '--steps', '25', '--warmup', '5']
        env = {k_: v for k_, v in os.environ.items() if not k_.startswith('FC_')}
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        line = json.loads(res.stdout.strip().splitlines()[-1])
        roof = line.get('roofline') or {}
        leg = {'metric': line['metric'], 'value': line['value'], 'unit': line['unit'], 'ms_per_step': line['ms_per_step'],
               'settled_ms_per_step': (line.get('settled') or {}).get('ms_per_step'),
               'workload': (line.get('config') or {}).get('workload'),
               'roofline': {k_: roof.get(k_) for k_ in ('kernel', 'frac', 'achieved', 'unit', 'avg_launch_ms')} if roof else None,
               'cpu_baseline': line.get('cpu_baseline')}
        for k_ in keep:
            if k_ in line:
                v = line[k_]
                leg[k_] = {kk: v[kk] for kk in ('ms_per_step', 'value', 'block_level_calls', 'per_operator_calls') if kk in v} if isinstance(v, dict) else v
        return leg
    except Exception as exc:           # noqa: BLE001  (an extra: never the reason for a failed bench)
        return {'value': None, 'note': f'failed: {type(exc).__name__}: {exc}'[:300]}


def committed_counters(kernel_names):
    """Counter-derived figures cannot be taken inside the timed run (rocprofv3 serialises the kernels); they come from
    the committed rocprof passes of the same command, stamped with the digest of the sources the library was built
    from: a kernel change makes them stale, and stale numbers are not reported."""
    from fieldconv_amd.build import _source_digest
    out = {'source': 'profiles/pmc_counters.json', 'stale': None}
    path = os.path.join(ROOT, 'profiles', 'pmc_counters.json')
    if not os.path.exists(path):
        out['stale'] = 'missing'
        return out, {}
    try:
        blob = json.load(open(path))
    except Exception:
        out['stale'] = 'unreadable'
        return out, {}
    out['profiled_library_digest'] = blob.get('library_source_digest')
    out['stale'] = blob.get('library_source_digest') != _source_digest()
    out['command'] = blob.get('command')
    per = {}
    if not out['stale']:
        for name in kernel_names:
            per[name] = blob.get('kernels', {}).get(name, {})
    return out, per


def init_dist(dev, backend):
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


SETTLE_SECONDS = 0.3
PROTOCOL_NOTE = ('value / ms_per_step: literally the command line -- `warmup` untimed steps as the first GPU work of the process, fence, '
                 '`steps` timed steps, fence, no instrumentation inside.  Per-kernel times (kernels, roofline): HIP events on every 4th '
                 'launch in a separate pass of `steps` untimed steps straight behind the timed region.  `settled`: the same warmup + '
                 'steps repeated after settled.extra_untimed_steps further untimed steps (~0.3 s of load), i.e. at the clock a training '
                 'run lives at; reported beside it, never as value (--cold skips it)')


def timed_loop(step, steps, warmup, use_dist, dev, backend, before_timed=None, after_timed=None, settle=True, head_start_cycles=0):
    """-> (seconds for `steps` steps under the LITERAL protocol, max over ranks; info).

    The literal protocol -- `warmup` steps, fence, `steps` steps, fence, as the first GPU work of the process -- is the value
    of record and carries NO instrumentation.  The MI355X's clock governor takes ~100 ms of sustained load to reach the clock
    it then holds (tools/clock_trace.py: 2.0-2.1 GHz during the first 20 ms of work with a dip to 1.9 GHz after ~3 ms,
    2.4 GHz from ~80 ms on), so a 25-step run of 0.4 ms steps lies entirely inside that ramp.  With settle=True the same steps
    then keep the device busy for SETTLE_SECONDS (untimed) and the protocol is repeated: info['settled'] = (seconds, extra
    untimed steps before its warm-up) -- an extra, not the value.

    Per-kernel HIP events (an event pair is two barrier packets, ~5 us of idle GPU) are taken in a separate INSTRUMENTED
    pass of `steps` untimed steps straight behind each timed region (tags 'literal' / 'settled'): before_timed() arms the
    events, after_timed(tag) harvests them.  The literal pass's instrumented steps are steps warmup+steps .. warmup+2*steps of
    the process, i.e. still inside the clock ramp the literal region ran in."""
    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def protocol(tag):
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        if before_timed is not None:            # the instrumented pass: same steps, event pairs on a sample of the launches, untimed
            before_timed()
            if head_start_cycles:
                for _ in range(2):              # the per-operator path's own first steps (allocator growth, caches) are not sampled
                    step()
                before_timed()
                # network modes (the instrumented steps run the per-operator host path): the host gets a head start behind a spinning
                # kernel -- an event pair brackets a launch on the GPU's time line only while the launch queue is not empty (a
                # bracket the host is late for also counts the wait for its kernel)
                torch.cuda._sleep(head_start_cycles)
            for _ in range(steps):
                step()
            fence()
            if after_timed is not None:
                after_timed(tag)
        if use_dist:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            if backend == 'gloo':
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    literal = protocol('literal')
    if not settle:
        return literal, {'settled': None}
    n_settle = max(1, int(SETTLE_SECONDS / (literal / steps)))          # the same count on every rank (literal is the max over ranks)
    for _ in range(n_settle):
        step()
    return literal, {'settled': (protocol('settled'), n_settle + steps)}


def run_identity(use_dist, dev, backend, local_rank, extra):
    try:
        return _run_identity(use_dist, dev, backend, local_rank, extra)
    except Exception as exc:          # noqa: BLE001  (a report, never the reason for a failed bench)
        return {'error': f'{type(exc).__name__}: {exc}'[:300], 'ranks': dist.get_world_size() if use_dist else 1}


def _run_identity(use_dist, dev, backend, local_rank, extra):
    """What actually ran where: one record per rank (all-gathered), so that a multi-GPU line can be checked the first time an
    8-GPU node produces one -- the communicator's size as RCCL sees it, the device every rank sat on, its share of the mesh."""
    props = torch.cuda.get_device_properties(dev)
    me = {'rank': dist.get_rank() if use_dist else 0, 'local_rank': local_rank, 'device': props.name,
          'device_uuid': str(getattr(props, 'uuid', '')) or None, 'pci_bus_id': getattr(props, 'pci_bus_id', None),
          'compute_units': props.multi_processor_count}
    me.update(extra)
    if not use_dist:
        return {'backend': None, 'ranks': 1, 'per_rank': [me], 'distinct_devices': 1}
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, me)
    ids = {(g_['device_uuid'] or g_['pci_bus_id'] or g_['local_rank']) for g_ in gathered}
    return {'backend': dist.get_backend() + (' (= RCCL on ROCm)' if dist.get_backend() == 'nccl' else ''),
            'ranks': dist.get_world_size(), 'per_rank': gathered, 'distinct_devices': len(ids)}


def sum_over_ranks(value, use_dist, dev, backend):
    if not use_dist:
        return value
    t = torch.tensor([value], device=dev, dtype=torch.int64)
    if backend == 'gloo':
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    return int(t.item())


def _timeit_ms(fn, n_warm=10, n=50):
    for _ in range(n_warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def _host_enqueue_ms(fn, n=5, reps=5):
    """what the HOST needs to enqueue a step: a few steps issued behind a synchronisation, timed until the last call returns (the
    launch queue is far from full after five steps, so nothing here waits for the GPU)"""
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = (time.perf_counter() - t0) / n * 1e3
        best = dt if best is None else min(best, dt)
    torch.cuda.synchronize()
    return best


def host_and_replay_extras(out, step):
    """host enqueue time, the step replayed as one HIP graph, and the per-operator host path of the same step (single GPU only)"""
    out['host_enqueue_ms_per_step'] = {'block_level_calls': _host_enqueue_ms(step),
                                       'note': 'host time to enqueue one step (Python, autograd, ctypes, allocator): five steps issued '
                                               'behind a synchronisation, timed until the last call returns, best of five; the step is '
                                               'GPU-bound while this stays below the replayed time'}
    try:
        from fieldconv_amd.utils import StepGraph
        sg = StepGraph(lambda: step())
        out['graph_replay'] = {'ms_per_step': _timeit_ms(sg.replay, 20, 100),
                               'note': 'the same forward + loss + backward captured once and replayed (fieldconv_amd.utils.StepGraph): one '
                                       'mesh only -- not what a training run over a dataset can use'}
        out['eager_over_replay'] = out['ms_per_step'] / out['graph_replay']['ms_per_step']
    except Exception as exc:                      # noqa: BLE001
        out['graph_replay'] = {'error': repr(exc)[:200]}
    prev_calls = os.environ.get('FIELDCONV_BLOCK_CALLS')
    os.environ['FIELDCONV_BLOCK_CALLS'] = '0'
    try:
        out['host_enqueue_ms_per_step']['per_operator_calls'] = _host_enqueue_ms(step)
        out['per_operator_calls'] = {'ms_per_step': _timeit_ms(step), 'note': 'the same step with every block composed of per-operator autograd '
                                                                            'nodes and foreign calls (FIELDCONV_BLOCK_CALLS=0)'}
    finally:           # (back to what the timed region ran with: config.env must describe the measured value)
        if prev_calls is None:
            del os.environ['FIELDCONV_BLOCK_CALLS']
        else:
            os.environ['FIELDCONV_BLOCK_CALLS'] = prev_calls


# ------------------------------------------------------------------------------------------------ mode dp (config 5)
def run_dp(args, world, rank, dev, use_dist, backend):
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.dist import GradientBuckets
    from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock, TangentPerceptron
    from fieldconv_amd.transforms import FCPrecomp
    N, k, nf, B, R, n_cls = args.dp_verts, args.dp_k, args.dp_channels, args.dp_band_limit, args.n_rings, 64
    data = sphere_support(N, k=k, seed=100 + rank, support=args.support).to(dev)       # every rank its own mesh
    pre = FCPrecomp(B, R, data.epsilon)
    torch.manual_seed(1234)                                                           # identical replicas
    blocks = [FCResNetBlock(16, nf, band_limit=B, n_rings=R)] + [FCResNetBlock(nf, nf, band_limit=B, n_rings=R) for _ in range(6)] + \
             [FCResNetBlock(nf, 16, band_limit=B, n_rings=R, frontload=True)]
    net = torch.nn.ModuleDict(dict(
        lift=LiftBlock(3, 16, n_rings=R, ftype=1), blocks=torch.nn.ModuleList(blocks),
        res=torch.nn.ModuleList([TangentPerceptron(16, nf), TangentPerceptron(nf, nf), TangentPerceptron(nf, nf), TangentPerceptron(nf, 16)]),
        echo=ECHOBlock(16, nf, n_des=12, n_bins=2, band_limit=B, n_rings=R), lin1=torch.nn.Linear(nf, 256),
        lin2=torch.nn.Linear(256, n_cls))).to(dev)
    params = [p for p in net.parameters()]
    buckets = GradientBuckets(params)
    g = torch.Generator().manual_seed(200 + rank)
    pos = torch.randn(N, 3, generator=g).to(dev)
    labels = torch.randint(0, n_cls, (N,), generator=g).to(dev)
    n_convs = 2 * len(blocks) + 1

    def step():
        edges, sten, ln, wxp = pre(data)                       # runs every forward in the reference, too (memoised per mesh)
        x1 = net['lift'](pos, edges, sten[..., B:B + 2])
        bl, res = net['blocks'], net['res']
        x = bl[0](x1, edges, sten)
        x2 = bl[1](x, edges, sten) + res[0](x1)
        x = bl[2](x2, edges, sten)
        x3 = bl[3](x, edges, sten) + res[1](x2)
        x = bl[4](x3, edges, sten)
        x4 = bl[5](x, edges, sten) + res[2](x3)
        x = bl[6](x4, edges, sten)
        x = bl[7](x, edges, sten) + res[3](x4)
        h = net['echo'](x, edges, sten, ln, wxp)
        logits = net['lin2'](torch.relu(net['lin1'](h)))
        loss = torch.nn.functional.cross_entropy(logits, labels)
        buckets.begin()                                        # autograd assigns the gradients (no add kernel per parameter) ...
        loss.backward()
        buckets.collect()                                      # ... one multi-tensor copy packs them into the flat buffer
        if use_dist:
            buckets.all_reduce()                               # the one collective of a data-parallel step
        return loss

    from fieldconv_amd.functional import kernel_timer
    edges0 = pre(data)[0]
    E = int(edges0.shape[0])
    harvested = {}

    def arm_timer():
        # 17 convolutions per step, every 4th launch of each kernel family bracketed: 4 and 17 are coprime, so over the instrumented
        # pass every layer of the network is sampled equally often (the blocks run as per-operator calls while the timer is armed)
        kernel_timer.reset(pairs=3 * (args.steps * n_convs // 4 + 2))
        kernel_timer.stride = 4
        kernel_timer.enabled = True

    def harvest(tag):
        kernel_timer.enabled = False
        harvested[tag] = {k_: sum(v) / len(v) for k_, v in kernel_timer.elapsed_ms().items() if v}

    elapsed, info = timed_loop(step, args.steps, args.warmup, use_dist, dev, backend, before_timed=None if args.no_kernel_events else arm_timer, after_timed=harvest,
                               settle=not args.cold, head_start_cycles=4_000_000)
    E_total = sum_over_ranks(E, use_dist, dev, backend)
    identity = run_identity(use_dist, dev, backend, dev.index, {'vertices': N, 'edges': E})
    if rank != 0:
        return None
    from fieldconv_amd.graph import get_graph
    n_params = sum(p.numel() for p in params)
    # algorithmic bytes of every FieldConv of the network: (in, out) per layer
    F = 2 * B + 1
    layers = []
    for blk in blocks:
        layers += [(blk.conv1.in_channels, blk.conv1.out_channels), (blk.conv2.in_channels, blk.conv2.out_channels)]
    layers.append((net['echo'].conv.in_channels, net['echo'].conv.out_channels))
    tot = {'fc_forward': 0, 'fc_backward_data': 0, 'fc_backward_filter': 0}
    for (ci, co) in layers:
        fwd_b = E * (8 * R * F + 4) + 4 * N + 8 * N * (ci + co) + 8 * co * ci * R * F
        bwd_b = E * (8 * R * F + 8) + 8 * N + 8 * N * (2 * ci + co) + 16 * co * ci * R * F
        wb = 8 * co * ci * R * F
        tot['fc_forward'] += fwd_b
        tot['fc_backward_data'] += bwd_b - wb - 8 * N * ci
        tot['fc_backward_filter'] += wb + 8 * N * ci

    def kernel_report(kt):
        per = {}
        for name, nbytes in tot.items():
            if name in kt:
                per_launch = nbytes / len(layers)
                sec = kt[name] * 1e-3
                per[name] = {'avg_ms': kt[name], 'algorithmic_bytes_avg_per_launch': per_launch, 'launches_per_step': len(layers),
                             'GBps': per_launch / sec / 1e9, 'hbm_frac': per_launch / sec / 1e9 / HBM_PEAK_GBS}
        return per

    per_kernel = kernel_report(harvested.get('literal', {}))
    dom = max(per_kernel, key=lambda n: per_kernel[n]['avg_ms']) if per_kernel else None
    roofline = None
    if dom:
        roofline = {'bound': 'hbm', 'kernel': dom + '_kernel', 'achieved': per_kernel[dom]['GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': per_kernel[dom]['hbm_frac'], 'traffic': None, 'avg_launch_ms': per_kernel[dom]['avg_ms'],
                    'algorithmic_bytes_per_launch': per_kernel[dom]['algorithmic_bytes_avg_per_launch'],
                    'note': f'mean over the {len(layers)} FieldConv layers of the network (HIP events around every 4th launch of each '
                            'kernel family in the instrumented pass behind the timed region; algorithmic bytes of SURVEY 8(d) summed over the layers / their '
                            'number); traffic: no counter pass is committed for this mode'}
    settled = None
    if info['settled'] is not None:
        t_set, n_extra = info['settled']
        settled = {'value': E_total * n_convs / (t_set / args.steps) / 1e6, 'unit': 'Medges/s', 'ms_per_step': t_set / args.steps * 1e3,
                   'extra_untimed_steps': n_extra + args.warmup + args.steps,
                   'kernel_us': {k_: round(v['avg_ms'] * 1e3, 1) for k_, v in kernel_report(harvested.get('settled', {})).items()},
                   'note': 'the same warmup + steps again after extra_untimed_steps more steps (sustained clock): an extra'}
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        threads = max(1, (os.cpu_count() or 2) // 2)
        try:
            # one 64 -> 64 FieldConv layer (12 of the network's 17 are of that shape) on a mesh of this size: the unit of the
            # metric is edges per convolution, so a one-layer figure is comparable
            cpu = cpu_baseline(args, threads, shape=(B, R, nf, k, N), slab=1250)
            cpu['sample'] = 'ONE FieldConv layer of the network (C=%d, band_limit=%d), not the whole network: ' % (nf, B) + cpu['sample']
        except Exception as exc:
            cpu = {'value': None, 'unit': 'Medges/s', 'cores': threads, 'kind': 'port', 'sample': f'failed: {type(exc).__name__}: {exc}'}
    out = {
        'metric': 'FieldConv fwd+bwd Medges/s (config 5: correspondence-net replicas, one mesh per GPU, C=64, M=3)',
        'value': E_total * n_convs / (elapsed / args.steps) / 1e6, 'unit': 'Medges/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
        'config': {'workload': f'BASELINE configs[4] shape: correspondence-net topology (LiftBlock 3->16, 8 FCResNetBlocks at {nf} channels '
                               f'with TangentPerceptron meta-residuals, ECHOBlock, 2 linear layers; {n_params} parameters) forward + loss + '
                               f'backward on one {N}-vertex mesh per GPU, k={k}, band_limit={B}, n_rings={R}; edges counted once per '
                               f'FieldConv ({n_convs} per network)',
                   'verts_per_gpu': N, 'edges_per_mesh_rank0': E, 'convs_per_step': n_convs, 'parameters': n_params,
                   'parallelism': 'single GPU (replica)' if world == 1 and not use_dist else
                                  f'data-parallel x{world}: one mesh per GPU, one bucketed all-reduce of {4 * buckets.flat.numel()} gradient '
                                  f'bytes per step over RCCL',
                   'kernels': describe_kernels(get_graph(*pre(data)[:2], N), nf, nf, B), 'env': env_report()},
        'ranks': identity,
        'roofline': roofline, 'kernels': per_kernel, 'cpu_baseline': cpu, 'settled': settled, 'protocol': PROTOCOL_NOTE,
    }
    if world == 1 and not use_dist and not args.no_extras:
        host_and_replay_extras(out, step)
    return out


# ------------------------------------------------------------------------------------------------ mode net (config 3)
def cpu_baseline_block(N, k, C, B, R, threads, support):
    """CPU leg of --mode net: ONE FCResNetBlock of the network (2 of its 9 convolutions) forward + backward on the same mesh, composed of
    the oracle's reference-structured torch pieces (oracle/reference_port_torch.py: materialised (E,C,R,F) product, index-add, broadcast
    multiply-and-sum; oracle/torch_composites.py: TangentLin / TangentNonLin), torch autograd for the backward pass.  A bounded sample:
    the whole network is 9 convolutions of this size plus the lift and descriptor blocks (~10 GB of autograd-saved temporaries)."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import FCResNetBlock
    from oracle import reference_port_torch as port
    from oracle import torch_composites as tc
    data = sphere_support(N, k=k, seed=0, support=support)
    edges, sten, _, _ = tc.FCPrecomp(B, R, data.epsilon)(data)
    g = torch.Generator().manual_seed(1)
    x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).requires_grad_(True)
    gy = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g))
    blk = FCResNetBlock(C, C, band_limit=B, n_rings=R, ftype=1)            # parameter container only (CPU)
    params = list(blk.parameters())

    def run():
        t0 = time.perf_counter()
        h = tc.tangent_nonlin(port.field_conv(x, edges, sten, blk.conv1.zonal, blk.conv1.spherical, blk.conv1.phase, 1, B), blk.nonlin1.bias)
        y = tc.tangent_nonlin(tc.tangent_lin(x, blk.res.Re, blk.res.Im)
                              + port.field_conv(h, edges, sten, blk.conv2.zonal, blk.conv2.spherical, blk.conv2.phase, 1, B), blk.nonlin2.bias)
        torch.autograd.grad(y, [x] + params, grad_outputs=gy)
        return time.perf_counter() - t0
    torch.set_num_threads(threads)
    run()                                                                   # warm-up (allocator, thread pool)
    reps, total = 0, 0.0
    while total < 10.0 and reps < 5:
        total += run()
        reps += 1
    E = int(edges.shape[0])
    return {'value': 2 * E * reps / total / 1e6, 'unit': 'Medges/s', 'cores': threads, 'kind': 'port', 'cpu_model': cpu_model(),
            'sample': f'ONE FCResNetBlock (2 of the network\'s 9 convolutions, edges counted once per convolution) forward + backward on the '
                      f'{N}-vertex mesh (E={E}, C={C}, B={B}, R={R}), reference-structured torch CPU port (oracle/reference_port_torch.py, '
                      f'oracle/torch_composites.py), {threads} threads, {reps} repetitions: {total:.1f} s'}


def run_net(args, world, rank, dev, use_dist, backend):
    """BASELINE configs[2]: the segmentation network's topology (reference segmentation.ipynb:165-236: LiftBlock(3 -> 48), four
    FCResNetBlocks, ECHOBlock(48 -> 8, n_des = 48, n_bins = 3)) forward + loss + backward on a 1 024-vertex mesh with ~128 neighbours per
    vertex (what sample_n = 1024 / epsilon = 0.2 give, :89,120), launched EAGERLY -- the reference trains with batch size 1 on a different
    mesh every step (:137), so a captured HIP graph does not cover a training run; the replayed figure is an extra.  With N > 1 ranks every
    rank holds its own mesh and a replica (one bucketed all-reduce of the gradients per step), as in --mode dp."""
    from fieldconv_amd.blocks import cpp_nodes
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.dist import GradientBuckets
    from fieldconv_amd.functional import kernel_timer
    from fieldconv_amd.graph import get_graph
    from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock
    from fieldconv_amd.transforms import FCPrecomp
    N, k, nf, B, R, n_cls = args.net_verts, args.net_k, args.channels, args.band_limit, args.n_rings, 8
    data = sphere_support(N, k=k, seed=300 + rank).to(dev)
    pre = FCPrecomp(B, R, data.epsilon)
    torch.manual_seed(1234)
    net = torch.nn.ModuleDict(dict(
        lift=LiftBlock(3, nf, n_rings=R, ftype=1),
        blocks=torch.nn.ModuleList([FCResNetBlock(nf, nf, band_limit=B, n_rings=R) for _ in range(4)]),
        echo=ECHOBlock(nf, n_cls, n_des=nf, n_bins=3, band_limit=B, n_rings=R))).to(dev)
    params = list(net.parameters())
    buckets = GradientBuckets(params) if use_dist else None
    g = torch.Generator().manual_seed(400 + rank)
    pos = torch.randn(N, 3, generator=g).to(dev)
    labels = torch.randint(0, n_cls, (N,), generator=g).to(dev)
    n_convs = 2 * len(net['blocks']) + 1

    def forward_loss(d, p_, lab):
        edges, sten, ln, wxp = pre(d)                          # runs every forward in the reference, too (memoised per mesh)
        x = net['lift'](p_, edges, sten[..., B:B + 2])
        for blk in net['blocks']:
            x = blk(x, edges, sten)
        logits = net['echo'](x, edges, sten, ln, wxp)
        return torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), lab)

    def step():
        loss = forward_loss(data, pos, labels)
        if buckets is None:
            return (loss,) + torch.autograd.grad(loss, params)
        buckets.begin()
        loss.backward()
        buckets.collect()
        buckets.all_reduce()
        return (loss,)

    E = int(pre(data)[0].shape[0])
    harvested = {}

    def arm_timer():
        # nine convolutions per step, every 4th launch of each kernel family bracketed (4 and 9 are coprime: every layer is sampled
        # equally often).  While the timer is armed the blocks run as per-operator calls (the brackets go around single kernels): the
        # same kernels, another host path -- which is why this pass is separate from the timed region
        kernel_timer.reset(pairs=3 * (args.steps * n_convs // 4 + 2))
        kernel_timer.stride = 4
        kernel_timer.enabled = True

    def harvest(tag):
        kernel_timer.enabled = False
        harvested[tag] = {k_: sum(v) / len(v) for k_, v in kernel_timer.elapsed_ms().items() if v}

    elapsed, info = timed_loop(step, args.steps, args.warmup, use_dist, dev, backend, before_timed=None if args.no_kernel_events else arm_timer, after_timed=harvest,
                               settle=not args.cold, head_start_cycles=4_000_000)
    E_total = sum_over_ranks(E, use_dist, dev, backend)
    identity = run_identity(use_dist, dev, backend, dev.index, {'vertices': N, 'edges': E})
    if rank != 0:
        return None
    F = 2 * B + 1
    fwd_b, bwd_b = algorithmic_bytes(N, E, nf, nf, R, F)       # every one of the nine convolutions is nf -> nf on this mesh
    wb = 8 * nf * nf * R * F
    contract = {'fc_forward': fwd_b, 'fc_backward_data': bwd_b - wb - 8 * N * nf, 'fc_backward_filter': wb + 8 * N * nf}

    def kernel_report(kt):
        per = {}
        for name, nbytes in contract.items():
            if name in kt:
                sec = kt[name] * 1e-3
                per[name] = {'avg_ms': kt[name], 'algorithmic_bytes_per_launch': nbytes, 'launches_per_step': n_convs,
                             'GBps': nbytes / sec / 1e9, 'hbm_frac': nbytes / sec / 1e9 / HBM_PEAK_GBS}
        return per

    per_kernel = kernel_report(harvested.get('literal', {}))
    dom = max(per_kernel, key=lambda n: per_kernel[n]['avg_ms']) if per_kernel else None
    roofline = None
    if dom:
        conv_ms = sum(v['avg_ms'] for v in per_kernel.values()) * n_convs
        roofline = {'bound': 'hbm', 'kernel': dom + '_kernel', 'achieved': per_kernel[dom]['GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': per_kernel[dom]['hbm_frac'], 'traffic': None, 'avg_launch_ms': per_kernel[dom]['avg_ms'],
                    'algorithmic_bytes_per_launch': per_kernel[dom]['algorithmic_bytes_per_launch'],
                    'algorithmic_bytes_per_step_all_convolutions': n_convs * (fwd_b + bwd_b),
                    'convolution_kernels_ms_per_step': conv_ms,
                    'note': f'mean over the {n_convs} FieldConv layers of the network, all {nf} -> {nf} on this mesh (HIP events around every '
                            '4th launch of each kernel family in the instrumented pass behind the timed region; algorithmic bytes of SURVEY '
                            '8(d) per convolution); at 64 tiles of 16 vertices the kernels are launch-size bound -- up to 8 workgroups '
                            'share a tile (edge split) -- not bandwidth bound; traffic: no counter pass is committed for this mode'}
    settled = None
    if info['settled'] is not None:
        t_set, n_extra = info['settled']
        settled = {'value': E_total * n_convs / (t_set / args.steps) / 1e6, 'unit': 'Medges/s', 'ms_per_step': t_set / args.steps * 1e3,
                   'extra_untimed_steps': n_extra + args.warmup + args.steps,
                   'kernel_us': {k_: round(v['avg_ms'] * 1e3, 1) for k_, v in kernel_report(harvested.get('settled', {})).items()},
                   'note': 'the same warmup + steps again after extra_untimed_steps more steps (sustained clock): an extra'}
    out = {
        'metric': 'FieldConv fwd+bwd Medges/s (config 3: segmentation-net topology, 1 024 verts, k~128, C=48, M=2; edges x 9 convolutions)',
        'value': E_total * n_convs / (elapsed / args.steps) / 1e6, 'unit': 'Medges/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
        'config': {'workload': f'BASELINE configs[2] shape: segmentation-net topology (LiftBlock 3->{nf}, 4 FCResNetBlocks, ECHOBlock {nf}->{n_cls} '
                               f'with n_des={nf}, n_bins=3; {sum(p.numel() for p in params)} parameters) forward + loss + backward on one {N}-vertex '
                               f'mesh per GPU, k={k} ({E} support edges), band_limit={B}, n_rings={R}, launched eagerly through the block-level '
                               f'entry points; edges counted once per FieldConv ({n_convs} per network)',
                   'verts_per_gpu': N, 'edges_per_mesh_rank0': E, 'convs_per_step': n_convs,
                   'parallelism': 'single GPU' if world == 1 and not use_dist else f'data-parallel x{world}: one mesh per GPU, one bucketed '
                                  f'all-reduce of {4 * buckets.flat.numel()} gradient bytes per step over RCCL',
                   'step_launch': 'eager (a training run sees a different mesh every step: reference segmentation.ipynb:137)',
                   'kernels': describe_kernels(get_graph(*pre(data)[:2], N), nf, nf, B),
                   'autograd_nodes': 'C++ (fc_torch_nodes.so)' if cpp_nodes() is not None else 'Python (fieldconv_amd/blocks.py)',
                   'env': env_report()},
        'ranks': identity, 'roofline': roofline, 'kernels': per_kernel, 'settled': settled, 'protocol': PROTOCOL_NOTE,
    }
    if world == 1 and not use_dist and not args.no_extras:
        host_and_replay_extras(out, step)
        timeit = _timeit_ms
        # (3) a DIFFERENT mesh every step, preprocessing included: what the reference's training loop does (segmentation.ipynb:276-317)
        meshes = []
        for i in range(8):
            d_i = sphere_support(N, k=k, seed=500 + i).to(dev)
            gi = torch.Generator().manual_seed(600 + i)
            meshes.append((d_i, torch.randn(N, 3, generator=gi).to(dev), torch.randint(0, n_cls, (N,), generator=gi).to(dev)))
        from fieldconv_amd.graph import clear_cache
        turn = [0]

        def fresh_step():
            d_i, p_i, l_i = meshes[turn[0] % len(meshes)]
            turn[0] += 1
            pre._memo = None
            clear_cache()                               # every step builds its mesh's support graph anew
            return torch.autograd.grad(forward_loss(d_i, p_i, l_i), params)
        out['new_mesh_every_step'] = {'ms_per_step': timeit(fresh_step, 16, 48),
                                      'note': 'eight meshes of the same size in turn, FCPrecomp + support-graph build (fc_precomp_mark, one '
                                              'synchronisation, fc_precomp_graph) inside every step, no cache hit'}
    if world == 1 and not args.no_cpu_baseline:
        threads = max(1, (os.cpu_count() or 2) // 2)
        try:
            out['cpu_baseline'] = cpu_baseline_block(N, k, nf, B, R, threads, 'all')
        except Exception as exc:
            out['cpu_baseline'] = {'value': None, 'unit': 'Medges/s', 'cores': threads, 'kind': 'port', 'sample': f'failed: {type(exc).__name__}: {exc}'}
    else:
        out['cpu_baseline'] = None
    return out


# ------------------------------------------------------------------------------------------ mode layer (the metric)
def run_layer(args, world, rank, dev, use_dist, backend):
    from fieldconv_amd.data import sphere_partition
    from fieldconv_amd.dist import GradientBuckets, HaloPlan, halo_exchange, overlap_backward, overlap_forward
    from fieldconv_amd.functional import kernel_timer
    from fieldconv_amd.graph import get_graph
    from fieldconv_amd.nn import FieldConv
    from fieldconv_amd.transforms import FCPrecomp

    B, R, C, k = args.band_limit, args.n_rings, args.channels, args.k
    F = 2 * B + 1
    n_total = args.verts * world
    overlap = use_dist and os.environ.get('BENCH_NO_OVERLAP', '0') != '1'
    # Forward overlap (interior targets under the halo exchange) is opt-in here: at this size the partitioned step is bound by
    # the host's enqueue rate (tools/dist_overhead.py: 508 us of host time per step against 420 us of GPU work with it,
    # the second forward launch being 43 us of that), so the exchange it would hide already falls into the GPU's idle time.
    overlap_fwd = overlap and os.environ.get('BENCH_FORWARD_OVERLAP', '0') == '1'
    data, n_owned, halo_global, bounds = sphere_partition(n_total, world, rank, k=k, seed=0, support=args.support, interior_first=overlap_fwd)
    n_interior = data.n_interior
    data = data.to(dev)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    n_local = data.num_nodes
    E = int(edges.shape[0])
    plan = HaloPlan(n_owned, halo_global, bounds, device=dev) if use_dist else None

    torch.manual_seed(1234)                                  # identical parameters on every rank
    conv = FieldConv(C, C, band_limit=B, n_rings=R, ftype=1).to(dev)
    params = list(conv.parameters())
    buckets = GradientBuckets(params) if use_dist else None   # parameter gradients land in one flat buffer: one all-reduce, no cat
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.complex(torch.randn(n_owned, C, generator=g), torch.randn(n_owned, C, generator=g))
    x[torch.rand(n_owned, C, generator=g) < 0.01] = 0        # 1 % exact zeros (origin-box path)
    x = x.to(dev).requires_grad_(True)
    gy = torch.complex(torch.randn(n_owned, C, generator=g), torch.randn(n_owned, C, generator=g)).to(dev)

    mesh_graph = get_graph(edges, sten, n_local)            # the mesh's cached graph
    if plan is not None:
        # restriction and exchange hooks go on a per-use view (the cached graph is shared by every user of the mesh); the
        # convolution gets it through a stand-in for the stencil
        from fieldconv_amd.graph import FactoredStencil
        mesh_graph = mesh_graph.view()
        sten = FactoredStencil.wrap(sten, mesh_graph)
    if plan is not None:
        mesh_graph.restrict_targets(n_owned)                # halo vertices are sources only: no output rows, no padded gy
    if overlap:
        overlap_backward(mesh_graph, plan)                  # gradient halo exchange under the filter-gradient kernel
    if overlap_fwd:
        overlap_forward(mesh_graph, plan, n_interior)       # interior targets are convolved while the halo rows travel
    split_fwd = mesh_graph.forward_split is not None and 0 < mesh_graph.forward_split[0] < n_local      # two forward launches per step

    def step():
        xl = halo_exchange(x, plan, deferred=overlap_fwd) if plan is not None else x
        y = conv(xl, edges, sten)
        if buckets is None:
            return torch.autograd.grad(y, [x] + params, grad_outputs=gy)
        buckets.begin()
        x.grad = None
        y.backward(gy)
        buckets.collect()
        buckets.all_reduce()
        return (x.grad,) + tuple(p.grad for p in params)

    # Opt-in (BENCH_GRAPH_STEP=1): the whole partitioned step -- both halo exchanges, the convolution's kernels, the bucketed
    # all-reduce -- captured once in ONE HIP graph and replayed (RCCL's collectives are capturable through torch's NCCL
    # backend).  One rank over RCCL on one MI355X: 415 us per step against 454 us launched eagerly (the plain step: 389 us;
    # tools/dist_overhead.py).  Not the default with several ranks: a multi-rank capture has never run on this pool.
    graph_step = use_dist and os.environ.get('BENCH_GRAPH_STEP', '0') == '1'
    eager_step = step
    if graph_step:
        from fieldconv_amd.utils import StepGraph
        sg = StepGraph(lambda: eager_step()[0])
        step = sg.replay

    def measure_prep():
        """per-mesh preprocessing, timed AFTER the metric's timed regions (the protocol's warm-up is the first GPU work of the process)"""
        # per-mesh preprocessing = everything between the reference's data object and the first convolution launch: FCPrecomp's
        # selection, the two edge groupings and the per-edge records (one fused build).  Mean of 20 builds after 20 untimed ones:
        # the package keeps the graphs of the last 16 meshes, so only then does every build recycle the memory of an evicted
        # one, as it does in an epoch over a dataset (before that each build pays a 170 MB hipMalloc, ~2 ms).
        pre_fresh = FCPrecomp(B, R, data.epsilon)
        inputs = (data.logMag, data.logAng, data.w, data.supp_edges, data.xp)
        for _ in range(20):
            pre_fresh._compute(*inputs)
        torch.cuda.synchronize()
        mallocs0 = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
        t0 = time.perf_counter()
        for _ in range(20):
            pre_fresh._compute(*inputs)
        torch.cuda.synchronize()
        prep_ms = (time.perf_counter() - t0) / 20 * 1e3
        prep_mallocs = torch.cuda.memory_stats(dev).get('num_device_alloc', 0) - mallocs0      # hipMalloc calls inside the timed builds
        return prep_ms, prep_mallocs

    harvested = {}

    def arm_timer():
        if graph_step:
            return                          # a replayed graph runs no Python: the kernels are timed on eager steps afterwards
        kernel_timer.reset(pairs=4 * (args.steps // 3 + 1))
        # every 4th launch of each kernel in the instrumented pass carries a HIP-event pair (every 3rd when the forward pass
        # is two launches per step, so that interior and boundary launches are sampled alternately)
        kernel_timer.stride = 3 if split_fwd else 4
        kernel_timer.enabled = True

    def harvest(tag):                       # after the fence that ends a timed region
        kernel_timer.enabled = False
        if not graph_step:
            harvested[tag] = {k_: sum(v) / len(v) for k_, v in kernel_timer.elapsed_ms().items() if v}

    elapsed, info = timed_loop(step, args.steps, args.warmup, use_dist, dev, backend, before_timed=None if args.no_kernel_events else arm_timer, after_timed=harvest,
                               settle=not args.cold)
    if graph_step:                          # per-kernel times from a few eager steps behind the timed regions
        kernel_timer.reset(pairs=64)
        kernel_timer.stride = 1
        kernel_timer.enabled = True
        for _ in range(8):
            eager_step()
        torch.cuda.synchronize()
        kernel_timer.enabled = False
        harvested['literal'] = {k_: sum(v) / len(v) for k_, v in kernel_timer.elapsed_ms().items() if v}
    E_total = sum_over_ranks(E, use_dist, dev, backend)
    ms_per_step = elapsed / args.steps * 1e3
    value = E_total / (elapsed / args.steps) / 1e6
    # What the collectives cost a step: the same step with the three collectives taken out (halo rows left as they are, no
    # gradient exchange, no all-reduce -- same kernels, same shapes), timed at the sustained clock like `settled`; the
    # difference to the settled step is the communication a step does not hide (its host time included).
    def measure_comm():
        xl0 = torch.cat([x.detach(), torch.zeros(plan.n_halo, C, dtype=x.dtype, device=dev)]).requires_grad_(True)
        gyl = gy if mesh_graph.n_targets == n_owned else torch.cat([gy, torch.zeros(n_local - n_owned, C, dtype=gy.dtype, device=dev)])
        saved_hooks = (mesh_graph.on_gx, mesh_graph.forward_split)
        mesh_graph.on_gx, mesh_graph.forward_split = None, None

        def step_nocomm():
            return torch.autograd.grad(conv(xl0, edges, sten), [xl0] + params, grad_outputs=gyl)
        for _ in range(args.warmup + 5):
            step_nocomm()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_nocomm()
        torch.cuda.synchronize()
        nocomm_ms = (time.perf_counter() - t0) / args.steps * 1e3
        mesh_graph.on_gx, mesh_graph.forward_split = saved_hooks
        t = torch.tensor([nocomm_ms], dtype=torch.float64)
        if backend == 'gloo':
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        else:
            t = t.to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        nocomm_ms = float(t.item())
        ref_ms = (info['settled'][0] / args.steps * 1e3) if info['settled'] is not None else ms_per_step
        comm = {'collectives_per_step': 3, 'step_without_collectives_ms': nocomm_ms,
                'exposed_ms_per_step': ref_ms - nocomm_ms, 'relative_to': 'settled' if info['settled'] is not None else 'literal',
                'halo_bytes_forward_rank0': plan.n_halo * C * 8, 'halo_bytes_backward_rank0': int(plan.send_idx.numel()) * C * 8,
                'allreduce_bytes': 4 * buckets.flat.numel(),
                'note': 'exposed = (step with halo exchange forward, transposed exchange backward, bucketed all-reduce) - (the same '
                        'kernels without them), both at the sustained clock, max over ranks'}
        return comm

    comm = None
    if use_dist and not graph_step:
        hooks = (mesh_graph.on_gx, mesh_graph.forward_split)
        try:        # (an extra: whatever goes wrong here on a first multi-rank RCCL run must not cost the line its value)
            comm = measure_comm()
        except Exception as exc:        # noqa: BLE001
            mesh_graph.on_gx, mesh_graph.forward_split = hooks
            comm = {'error': f'{type(exc).__name__}: {exc}'[:300]}
    identity = run_identity(use_dist, dev, backend, dev.index, {'owned_vertices': int(n_owned), 'halo_rows': 0 if plan is None else plan.n_halo,
                                                                 'edges': E})
    prep_ms, prep_mallocs = measure_prep()
    if rank != 0:
        return None

    factored = bool(mesh_graph.factored)
    fwd_b, bwd_b = algorithmic_bytes(n_local, E, C, C, R, F)
    fwd_f, bwd_f = algorithmic_flops(n_local, E, C, C, R, F, factored)
    gemm_f = 8 * n_local * C * C * R * F
    wbytes = 8 * C * C * R * F
    # backward contract bytes split over its two kernels: the data kernel reads stencil, indices, gy, x, W and
    # writes gx; the filter kernel's contract traffic is x and the filter gradient (its H input is a temporary)
    contract = (('fc_forward', fwd_b, fwd_f), ('fc_backward_data', bwd_b - wbytes - 8 * n_local * C, bwd_f - gemm_f),
                ('fc_backward_filter', wbytes + 8 * n_local * C, gemm_f))
    # Large meshes run the H-streaming arrangement (fc_backward_streams): 'fc_backward_data' then brackets the GATHER kernel (records, indices,
    # the cotangent rows; H is a temporary) and 'fc_backward_filter' the STREAMING kernel + gx (x in, gx out, the filter in, its gradient
    # out; both contractions: their flops)
    streaming = False
    try:
        from fieldconv_amd import _lib as _l
        from fieldconv_amd.functional import make_dims
        import ctypes as _ct
        streaming = bool(factored and _l.load().fc_backward_streams(_ct.byref(make_dims(mesh_graph, C, C, B)), 1))
    except Exception:       # noqa: BLE001
        streaming = False
    if streaming:
        gather_b = E * (8 * R * F + 8) + 8 * n_local + 8 * n_local * C
        contract = (('fc_forward', fwd_b, fwd_f), ('fc_backward_data', gather_b, bwd_f - 2 * gemm_f),
                    ('fc_backward_filter', bwd_b - gather_b, 2 * gemm_f))

    def kernel_report(kt):
        kt = dict(kt)
        if split_fwd and 'fc_forward' in kt:
            kt['fc_forward'] *= 2               # interior + boundary launch
        per = {}
        for name, nbytes, nflops in contract:
            if name in kt:
                sec = kt[name] * 1e-3
                per[name] = {'avg_ms': kt[name], 'algorithmic_bytes': nbytes, 'GBps': nbytes / sec / 1e9,
                             'hbm_frac': nbytes / sec / 1e9 / HBM_PEAK_GBS, 'evaluated_flops': nflops,
                             'TFLOPs': nflops / sec / 1e12, 'flop_rate_vs_fp32_peak': nflops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS}
        return per

    per_kernel = kernel_report(harvested.get('literal', {}))
    counters_meta, counters = committed_counters(list(per_kernel))
    for name, c in counters.items():
        for key in ('mfma_busy', 'valu_busy', 'valu_insts', 'salu_insts', 'hbm_bytes_per_launch', 'hbm_read_bytes', 'hbm_write_bytes'):
            if key in c:
                per_kernel[name][key] = c[key]
    dom = max(per_kernel, key=lambda n: per_kernel[n]['avg_ms']) if per_kernel else None
    roofline = None
    if dom:
        kname = dom + '_kernel'
        if streaming and dom != 'fc_forward':
            kname = 'fc_backward_gather_kernel' if dom == 'fc_backward_data' else 'fc_backward_stream_kernel (+ fc_backward_gx_kernel)'
        roofline = {'bound': 'hbm', 'kernel': kname, 'achieved': per_kernel[dom]['GBps'], 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': per_kernel[dom]['hbm_frac'], 'traffic': per_kernel[dom].get('hbm_bytes_per_launch'),
                    'avg_launch_ms': per_kernel[dom]['avg_ms'], 'algorithmic_bytes_per_launch': per_kernel[dom]['algorithmic_bytes'],
                    'mfma_busy': per_kernel[dom].get('mfma_busy'), 'valu_busy': per_kernel[dom].get('valu_busy'),
                    'counters': counters_meta,
                    'note': 'HIP events on every 4th launch in the instrumented pass straight behind the literal timed region (the timed '
                            'region itself carries no events).  Not HBM-bound: the walks of the gathers pay for their vector arithmetic, '
                            'their row fetches (one per ~28 cycles and CU whatever the cache level) and their record reads one after '
                            'the other (tools/ubench/walk.hip, DESIGN 5), with the matrix pipe and HBM mostly idle; achieved = '
                            'algorithmic bytes / launch time as SURVEY 8(d) prescribes'}
        # the bound the kernels live under: of all vector instructions the dominant kernel issues, how many are the gather's
        # own arithmetic (2 + 2B complex products and 2F ring updates per edge, packed: 2 instructions per complex product)
        c = counters.get(dom, {})
        if c.get('valu_insts'):
            useful = E * ((2 + 2 * B) * 2 + 2 * F) if dom != 'fc_backward_filter' else None
            if streaming and dom == 'fc_backward_data':
                useful = E * (2 * F + 2 * F)         # gather kernel: F complex products with the record's phases + 2F ring updates per edge
            roofline['issue'] = {'vector_insts_per_launch': c['valu_insts'], 'scalar_insts_per_launch': c.get('salu_insts'),
                                 'gather_arithmetic_insts_per_launch': useful,
                                 'useful_frac': (useful / c['valu_insts']) if useful else None, 'valu_busy': c.get('valu_busy')}
    settled = None
    if info['settled'] is not None:
        t_set, n_extra = info['settled']
        per_set = kernel_report(harvested.get('settled', {}))
        settled = {'value': E_total / (t_set / args.steps) / 1e6, 'unit': 'Medges/s', 'ms_per_step': t_set / args.steps * 1e3,
                   'extra_untimed_steps': n_extra + args.warmup + args.steps,
                   'kernel_us': {k_: round(v['avg_ms'] * 1e3, 1) for k_, v in per_set.items()},
                   'hbm_frac_dominant_kernel': max((v['avg_ms'], v['hbm_frac']) for v in per_set.values())[1] if per_set else None,
                   'note': 'the same warmup + steps again after extra_untimed_steps more steps of the same workload (the literal run, '
                           'then ~0.3 s of load): the sustained-clock figure, an extra -- value is the literal run'}
    out = {
        'metric': 'FieldConv fwd+bwd Medges/s (20k verts, k=32, C=48, M=2)',
        'value': value, 'unit': 'Medges/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': DTYPE, 'data': 'synthetic',
        'config': {'workload': f'BASELINE configs[1] shape: one FieldConv layer fwd+bwd on a synthetic sphere mesh, '
                               f'{args.verts} verts/GPU, k={k} nearest neighbours, support radius = '
                               + ('95-percentile of the k-NN distances' if args.support == 'p95' else 'above every k-NN distance')
                               + f' ({E} edges kept on rank 0), C={C}->{C}, band_limit={B}, n_rings={R}, ftype=1',
                   'verts_per_gpu': args.verts, 'edges_total': E_total, 'k': k, 'channels': C, 'band_limit': B, 'n_rings': R,
                   'support': args.support, 'stencil_path': 'geometric records' if mesh_graph.geo_t is not None else
                   ('factored records' if factored else 'dense rows'),
                   'parallelism': 'single GPU' if world == 1 else f'vertex partition x{world}, one-hop halo over RCCL',
                   'halo_rows_rank0': 0 if plan is None else plan.n_halo,
                   'step_launch': 'one HIP graph per step (BENCH_GRAPH_STEP=1)' if graph_step else 'eager',
                   'kernels': describe_kernels(mesh_graph, C, C, B), 'env': env_report()},
        'ranks': identity, 'communication': comm,
        'roofline': roofline,
        'kernels': per_kernel,
        'hbm_frac_fwd_bwd': (fwd_b + bwd_b) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
        'hbm_frac_kernels': (fwd_b + bwd_b) / (sum(v['avg_ms'] for v in per_kernel.values()) * 1e-3) / 1e9 / HBM_PEAK_GBS if per_kernel else None,
        'mesh_preprocessing_ms': prep_ms, 'mesh_preprocessing_device_mallocs': prep_mallocs,
        'settled': settled, 'protocol': PROTOCOL_NOTE,
    }
    y_def = gx_def = None
    if args.dump or (world == 1 and not use_dist and not args.no_extras):
        gd = step()
        y_def, gx_def = conv(x, edges, sten).detach().cpu(), gd[0].detach().cpu()
    if args.dump:
        torch.save({'y': y_def, 'gx': gx_def}, args.dump)
    if world == 1 and not use_dist and not args.no_extras:
        # extras (not the metric of record)
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
        try:
            from fieldconv_amd.utils import StepGraph
            sg = StepGraph(step)                      # the same step, captured once and replayed as one HIP graph
            for _ in range(50):
                sg.replay()
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for _ in range(200):
                sg.replay()
            torch.cuda.synchronize()
            gms = (time.perf_counter() - tg) / 200 * 1e3
            out['graphed_step'] = {'ms_per_step': gms, 'value': E / (gms * 1e-3) / 1e6, 'unit': 'Medges/s',
                                   'note': 'the same forward + backward replayed as one HIP graph (fieldconv_amd.utils.StepGraph): '
                                           'no launch gaps, no host work per step; not the metric of record'}
        except Exception as exc:                      # noqa: BLE001  (an extra, never the reason for a failed bench)
            out['graphed_step'] = {'error': repr(exc)[:200]}
        if os.environ.get('FC_MFMA') is None:
            out['fp32_mfma'] = other_mode(args, {'FC_MFMA': 'f32'}, 'v_mfma_f32_16x16x4_f32 on fp32 operands throughout (FC_MFMA=f32)',
                                          y_def, gx_def)
            out['reduced_precision'] = other_mode(args, {'FC_MFMA': 'f16'}, 'single f16 halves with per-row power-of-two scales, fp32 '
                                                  'accumulation (FC_MFMA=f16): the bf16-class leg of SURVEY 8(d) -- the MFMA rate of bf16, '
                                                  'three more mantissa bits, the scales make up for the range', y_def, gx_def)
        # configs[2] and configs[4] of BASELINE.json, compact: the lines `--mode net` / `--mode dp` print, each from a child process of
        # its own behind everything that is timed here (segmentation.ipynb:165-236 / correspondence.ipynb Net; one rank)
        out['config3_segmentation_net'] = mode_leg('net', ('eager_over_replay', 'graph_replay', 'new_mesh_every_step', 'host_enqueue_ms_per_step'))
        out['config5_correspondence_net_one_rank'] = mode_leg('dp', ())
        if args.support == 'p95':
            try:
                other, _ = child_run(args, {}, extra_args=['--support', 'all'], dump=False)
                out['round1_mesh'] = {'value': other['value'], 'unit': other['unit'], 'ms_per_step': other['ms_per_step'],
                                      'edges_total': other['config']['edges_total'],
                                      'note': "the round-1 mesh (--support all: no edge dropped, E = N*k, rings 4-5 empty)"}
            except Exception as exc:
                out['round1_mesh'] = {'value': None, 'note': f'failed: {type(exc).__name__}: {exc}'}
    if world == 1 and not args.no_cpu_baseline:
        ncpu = os.cpu_count() or 2
        threads = max(1, ncpu // 2)                 # physical cores (SMT siblings excluded)
        try:
            out['cpu_baseline'] = cpu_baseline(args, threads)
        except Exception as exc:                   # the GPU numbers stand on their own
            out['cpu_baseline'] = {'value': None, 'unit': 'Medges/s', 'cores': threads, 'kind': 'port',
                                   'sample': f'failed: {type(exc).__name__}: {exc}'}
    else:
        out['cpu_baseline'] = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)       # 0.4 ms each: long enough for steady clocks and a full launch queue
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--mode', default='layer', choices=['layer', 'dp', 'net'])
    ap.add_argument('--verts', type=int, default=20000, help='vertices per GPU (mode layer)')
    ap.add_argument('--k', type=int, default=32)
    ap.add_argument('--channels', type=int, default=48)
    ap.add_argument('--band-limit', type=int, default=2)
    ap.add_argument('--n-rings', type=int, default=6)
    ap.add_argument('--dp-verts', type=int, default=4999, help='vertices per mesh (mode dp: FAUST-remeshed size)')
    ap.add_argument('--dp-k', type=int, default=28)
    ap.add_argument('--dp-channels', type=int, default=64)
    ap.add_argument('--dp-band-limit', type=int, default=3)
    ap.add_argument('--net-verts', type=int, default=1024, help='vertices per mesh (mode net: the reference samples 1 024 points)')
    ap.add_argument('--net-k', type=int, default=128, help='neighbours per vertex (mode net)')
    ap.add_argument('--support', default='p95', choices=['p95', 'all'],
                    help="support radius: 'p95' = 95-percentile of the k-NN distances (SURVEY 8(d) G-geo: FCPrecomp drops 5 %% of the "
                         "edges, every ring populated); 'all' = above every k-NN distance (round-1 mesh: E = N*k, outer rings empty)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cold', action='store_true', help='no clock-settling steps before the measurement (see timed_loop)')
    ap.add_argument('--no-extras', action='store_true', help='only the metric of record (used by the child runs)')
    ap.add_argument('--no-kernel-events', action='store_true', help='no instrumented pass behind the timed region (roofline: null) -- for runs under a profiler')
    ap.add_argument('--dump', default=None, help='write y and gx of one step to this file (child runs)')
    args = ap.parse_args()

    # Development switches of the library change what is computed (FC_DEBUG* skip whole phases): a benchmark line taken
    # with one of them set would be wrong or mislabelled, so it is refused outright.
    from fieldconv_amd import _env
    bad = sorted(k_ for k_ in os.environ if k_ in _env.WRONG_RESULTS)
    if bad:
        sys.exit('bench.py refuses to run with library debug switches set: ' + ', '.join(bad))
    # ... and a name with one of our prefixes that nothing reads is a typo or the switch of a removed kernel: the line would
    # claim a configuration that did not run.  The switches that ARE set travel in config.env.
    bad = _env.unknown()
    if bad:
        sys.exit('bench.py: unknown switch(es) ' + ', '.join(bad) + ' (known: ' + ', '.join(sorted(_env.SWITCHES)) + ')')

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` from a plain shell: start the N ranks ourselves.  Nothing in this process has touched the
        # GPU yet (importing torch and parsing arguments do not; the device count comes from sysfs), it never will, and the
        # ranks are CHILD processes (no exec of a GPU-initialised process): rank 0's JSON line is relayed as our last stdout
        # line, the exit code is the launcher's.
        sys.exit(self_launch(args.gpus))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        args.gpus = world
    backend = os.environ.get('BENCH_BACKEND', 'nccl')                  # "gloo": several ranks on one GPU (test rigs only)
    if backend == 'gloo':
        local_rank = local_rank % max(torch.cuda.device_count(), 1)

    # Build BEFORE anything touches the GPU; under a profiler never: rocprofv3's preloaded tool library has initialised the
    # GPU before this script starts, and a build would start hipcc children (which exec clang) from such a process -- the
    # exec hop this pool forbids.  Build first (python3 -c 'import __graft_entry__; __graft_entry__.build()'), then profile.
    from fieldconv_amd import build as _build
    if _build.under_profiler() and _build.needs_build():
        sys.exit('bench.py: libfieldconv_hip.so is missing or stale and this process runs under a profiler; build first with '
                 "python3 -c 'import __graft_entry__; __graft_entry__.build()' and profile again")
    import __graft_entry__
    __graft_entry__.build()

    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    force_dist = os.environ.get('BENCH_FORCE_DIST', '0') == '1'        # exercise the RCCL path with a single rank
    use_dist = world > 1 or force_dist
    if use_dist:
        init_dist(dev, backend)

    out = {'dp': run_dp, 'net': run_net, 'layer': run_layer}[args.mode](args, world, rank, dev, use_dist, backend)
    if rank == 0:
        from fieldconv_amd import _lib
        out['config']['library'] = ('development build (-DFC_DEV_SWITCHES: honours the FC_* switches in config.env)'
                                    if _lib.load().fc_dev_switches() else 'product build (reads no environment variable)')
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def visible_gpus():
    """GPUs a child process will see, WITHOUT initialising the HIP runtime here (torch.cuda.device_count() is hipGetDeviceCount):
    KFD topology nodes with SIMDs (CPUs are nodes without), narrowed by ROCR_/HIP_/CUDA_VISIBLE_DEVICES when set."""
    import glob
    n = 0
    for path in glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties'):
        try:
            for line in open(path):
                if line.startswith('simd_count') and int(line.split()[1]) > 0:
                    n += 1
        except (OSError, ValueError):
            pass
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            listed = len([t for t in v.split(',') if t.strip() != ''])
            n = min(n, listed)
    return n


def self_launch(n):
    """Run `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same arguments>` as a child and relay
    its output; -> exit code.  Fewer visible devices than ranks: refused, unless BENCH_BACKEND=gloo (test rigs: ranks share
    devices, host-staged collectives)."""
    import socket
    ndev = visible_gpus()                               # from sysfs and the *_VISIBLE_DEVICES variables: no HIP call in this process
    if ndev < n and os.environ.get('BENCH_BACKEND', 'nccl') != 'gloo':
        print(f'bench.py --gpus {n}: only {ndev} GPU(s) visible (RCCL needs one device per rank; BENCH_BACKEND=gloo shares devices '
              'for functional tests)', file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    last_json = None
    for line in proc.stdout:
        line = line.rstrip('\n')
        if line.startswith('{') and '"metric"' in line:
            last_json = line                            # held back: it must be the LAST line of our stdout
        else:
            print(line, flush=True)
    rc = proc.wait()
    if last_json is not None:
        print(last_json, flush=True)
    elif rc == 0:
        rc = 1
    return rc


if __name__ == '__main__':
    main()
