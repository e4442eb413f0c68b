"""autograd.Function wrappers that launch the HIP kernels through the C ABI.

The tensors are plain torch device buffers; only raw pointers, sizes and the current HIP stream
cross into libfieldconv_hip.so (include/fieldconv_hip.h).
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import FcCsr, FcDims, FcEpilogue, FcFilterParams, check


class KernelTimer:
    """Optional HIP-event bracket around the single-kernel launches `fc_forward` / `fc_backward`
    (each of those entry points enqueues exactly one kernel on the current stream).  bench.py turns
    it on for the timed region to obtain per-kernel durations for the roofline figure.  An event pair is a
    pair of barrier packets on the stream (about 5 us of idle GPU per bracket), so only every `stride`-th
    launch of a kernel is bracketed."""

    def __init__(self):
        self.enabled = False
        self.stride = 1
        self.events = {}
        self.count = {}
        self.pool = []

    def reset(self, pairs=0):
        """Forget earlier measurements; `pairs` event pairs are created now (hipEventCreate is not free and
        must not land in the timed region)."""
        self.events = {}
        self.count = {}
        self.pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(pairs)]

    def elapsed_ms(self):
        """{name: [ms, ...]}; call after torch.cuda.synchronize()."""
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in self.events.items()}


kernel_timer = KernelTimer()
KEEP_GW_EFF = False      # tests: also materialise gW_eff (O,I,R,F) when a module's backward pass only needs the parameter gradients
_ONE_CALL = os.environ.get('FIELDCONV_SEPARATE_CALLS', '0') != '1'      # development: one foreign call per kernel instead of per pass


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.on = False
        if kernel_timer.enabled:
            n = kernel_timer.count.get(self.name, 0)
            kernel_timer.count[self.name] = n + 1
            self.on = n % kernel_timer.stride == 0
        if self.on:
            if kernel_timer.pool:
                self.a, self.b = kernel_timer.pool.pop()
            else:
                self.a = torch.cuda.Event(enable_timing=True)
                self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.on:
            self.b.record()
            kernel_timer.events.setdefault(self.name, []).append((self.a, self.b))
        return False


def _require_device(t, what):
    if not t.is_cuda:
        raise RuntimeError(f'{what}: fieldconv_amd runs on a ROCm device only (got a {t.device} tensor); '
                           'there is no CPU fallback')


def on_device(t):
    """True for tensors the kernels can take (a ROCm device tensor)"""
    return t.is_cuda


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    """The current HIP stream of the current device as a raw pointer.  torch.cuda.current_stream() costs ~10 us of
    Python per call -- with ~25 launches per network forward that was a quarter of the host's enqueue time."""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(device):
    """Device guard for the launches below; free when `device` already is the current one (the usual case)."""
    if _cur_device is not None and device.index is not None and _cur_device() == device.index:
        return _NO_GUARD
    return torch.cuda.device(device)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _csr(rowptr, nbr, runs=None):
    return FcCsr(rowptr.data_ptr(), nbr.data_ptr() if nbr.numel() else None, runs.data_ptr() if runs is not None else None)


def make_dims(graph, I, O, B):
    return FcDims(graph.N, graph.E, int(I), int(O), graph.R, int(B))


def supported(graph, I, O, B):
    return bool(_lib.load().fc_supported(ctypes.byref(make_dims(graph, I, O, B))))


@_lib.keep_mode
class _FieldConvFn(torch.autograd.Function):
    """y = FieldConv(x; W_eff) on a preprocessed support graph (reference nn/field_conv.py:128-137)."""

    @staticmethod
    def forward(ctx, x, w_eff, graph):
        lib = _lib.load()
        x = x.contiguous()
        w_eff = w_eff.contiguous()
        O, I, R, F = w_eff.shape
        B = (F - 1) // 2
        plan = _conv_plan(lib, graph, I, O, B)
        with _on(x.device):
            st = _stream()

            def pack(pl, wpk_f, wpk_b):
                check(lib.fc_pack_filter(_p(w_eff), _p(wpk_f), _p(wpk_b) if wpk_b is not None else None, pl.dref, pl.records, st), 'fc_pack_filter')
            y, wpk_b = _run_forward(lib, x, graph, plan, O, st, pack)
        ctx.save_for_backward(x, wpk_b)
        ctx.graph = graph
        ctx.wshape = (O, I, R, F)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, wpk_b = ctx.saved_tensors
        O, I, R, F = ctx.wshape
        plan = _conv_plan(lib, ctx.graph, I, O, (F - 1) // 2)
        gy = gy.contiguous()
        with _on(x.device):
            gx, gw, _ = _launch_backward(lib, x, gy, ctx.graph, wpk_b, plan, ctx.wshape, _stream())
        return gx, gw, None


class _ConvPlan:
    """What every launch of one (graph, in, out, band limit) combination needs and never changes: the dims struct, the
    grouping structs, buffer sizes.  Cached on the graph: ~15 ctypes calls per convolution otherwise."""
    __slots__ = ('dims', 'dref', 'n_fwd', 'n_bwd', 'ws_fwd', 'ws_bwd', 'csr_t', 'csr_s', 'cref_t', 'cref_s', 'records')


def _conv_plan(lib, graph, I, O, B):
    key = (int(I), int(O), int(B))
    plan = graph._plans.get(key)
    if plan is not None:
        return plan
    plan = _ConvPlan()
    plan.dims = make_dims(graph, I, O, B)
    plan.dref = ctypes.byref(plan.dims)
    if not lib.fc_supported(plan.dref):
        raise _lib.FieldConvNativeError(
            f'FieldConv(in={I}, out={O}, n_rings={graph.R}, band_limit={B}) is outside the compiled HIP kernels '
            '(channels <= 64; (n_rings, band_limit) as listed in csrc/fc_kernels.hpp)')
    # which entry-point family the packed filter images are for (bit 0: record-driven; bit 1: the ring-major backward kernels)
    plan.records = lib.fc_records_flags(plan.dref, 1 if graph.factored else 0)
    plan.n_fwd = lib.fc_packed_filter_floats_fwd(plan.dref, plan.records)
    plan.n_bwd = lib.fc_packed_filter_floats_bwd(plan.dref, plan.records)
    plan.ws_bwd = lib.fc_backward_workspace_bytes(plan.dref, plan.records)
    plan.ws_fwd = 0
    if graph.factored and os.environ.get('FIELDCONV_NO_EDGE_SPLIT', '0') != '1':
        plan.ws_fwd = lib.fc_forward_workspace_bytes(plan.dref)      # non-zero on small meshes with wide supports
    plan.csr_t = _csr(graph.rowptr_t, graph.nbr_t, graph.runs_t)
    plan.csr_s = _csr(graph.rowptr_s, graph.nbr_s, graph.runs_s)
    plan.cref_t, plan.cref_s = ctypes.byref(plan.csr_t), ctypes.byref(plan.csr_s)
    graph._plans[key] = plan
    return plan


def _launch_forward(lib, x, graph, wpk_f, plan, O, st, addend=None, bias=None, out=None, row0=0):
    """-> y, or (pre-activation, activated) when a modReLU bias is given: the residual `addend` and the modReLU run in the
    kernel's epilogue (include/fieldconv_hip.h: fc_epilogue).  `plan` may cover the targets [row0, row0 + plan.dims.N) only
    (_row_plan); the rows then land in the tensors of `out` = (y, activated)."""
    n = plan.dims.N
    if out is None:
        y = torch.empty((n, O), dtype=torch.complex64, device=x.device)
        act = torch.empty_like(y) if bias is not None else None
    else:
        y, act = out[0][row0:row0 + n], (out[1][row0:row0 + n] if out[1] is not None else None)
        addend = addend[row0:row0 + n] if addend is not None else None
    epi = None
    if addend is not None or bias is not None:
        epi = ctypes.byref(FcEpilogue(addend.data_ptr() if addend is not None else None, bias.data_ptr() if bias is not None else None,
                                      act.data_ptr() if act is not None else None))
    nbytes = plan.ws_fwd
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    wsp = _p(ws) if ws is not None else None
    with _timed('fc_forward'):
        if graph.geo_t is not None:
            check(lib.fc_forward_geometric(_p(x), _p(graph.geo_t), plan.cref_t, _p(wpk_f), _p(y), wsp, nbytes, plan.dref, epi, st),
                  'fc_forward_geometric')
        elif graph.factored:
            check(lib.fc_forward_factored(_p(x), _p(graph.rec_t), plan.cref_t, _p(wpk_f), _p(y), wsp, nbytes, plan.dref, epi, st),
                  'fc_forward_factored')
        else:
            check(lib.fc_forward(_p(x), _p(graph.sten_t), plan.cref_t, _p(wpk_f), _p(y), plan.dref, epi, st), 'fc_forward')
    return y if bias is None else (y, act)


def _row_plan(lib, graph, whole, row0, nrows):
    """Forward-launch plan for the targets [row0, row0 + nrows) of `graph`: the same records and sources, the grouping arrays
    entered at row0 (fc_csr: row pointers are absolute slots, runs are per row).  The kernel variant -- and with it the
    layout of the packed filter image -- follows the number of rows of the launch, hence a plan (and an image) of its own."""
    key = ('rows', whole.dims.I, whole.dims.O, whole.dims.B, int(row0), int(nrows))
    plan = graph._plans.get(key)
    if plan is not None:
        return plan
    plan = _ConvPlan()
    ends = graph.rowptr_t[[row0, row0 + nrows]].tolist()
    plan.dims = FcDims(int(nrows), int(ends[1] - ends[0]), whole.dims.I, whole.dims.O, graph.R, whole.dims.B)
    plan.dref = ctypes.byref(plan.dims)
    plan.records = whole.records
    plan.n_fwd = lib.fc_packed_filter_floats_fwd(plan.dref, plan.records)
    plan.n_bwd = whole.n_bwd
    plan.ws_bwd = whole.ws_bwd
    plan.ws_fwd = 0
    if os.environ.get('FIELDCONV_NO_EDGE_SPLIT', '0') != '1':
        plan.ws_fwd = lib.fc_forward_workspace_bytes(plan.dref)
    plan.csr_t = _csr(graph.rowptr_t[row0:], graph.nbr_t, graph.runs_t[row0:] if graph.runs_t is not None else None)
    plan.csr_s, plan.cref_s = whole.csr_s, whole.cref_s
    plan.cref_t = ctypes.byref(plan.csr_t)
    graph._plans[key] = plan
    return plan


def _run_forward(lib, x, graph, plan, O, st, pack, addend=None, bias=None, params=None):
    """Filter images + forward launch(es).  pack(plan, wpk_f, wpk_b) enqueues the packing kernel.  -> (result of
    _launch_forward, wpk_b).  A graph with `forward_split` = (n_first, between) (dist/halo.py: overlap_forward) runs the
    targets [0, n_first) first, calls between() -- the wait for the halo rows of x, which only later targets read -- and
    then the rest."""
    dev = x.device
    wpk_b = torch.empty(plan.n_bwd, dtype=torch.float32, device=dev)
    split, nt = graph.forward_split, graph.n_targets
    between = split[1] if split is not None else None
    if split is not None and graph.factored and 0 < split[0] < nt:
        ranges = ((0, split[0]), (split[0], nt - split[0]))
    elif nt < graph.N and graph.factored:
        ranges = ((0, nt),)                       # a partitioned mesh: the rows behind n_targets have no in-edges (halo vertices)
    else:
        if nt < graph.N:
            raise _lib.FieldConvNativeError('restrict_targets needs a graph with factored records')
        wpk_f = torch.empty(plan.n_fwd, dtype=torch.float32, device=dev)
        if params is not None and between is None and _ONE_CALL and not kernel_timer.enabled:
            # filter images + forward launch in one foreign call (fc_forward_params)
            zonal, spherical, phase, ftype = params
            y = torch.empty((graph.N, O), dtype=torch.complex64, device=dev)
            act = torch.empty_like(y) if bias is not None else None
            epi = None
            if addend is not None or bias is not None:
                epi = ctypes.byref(FcEpilogue(addend.data_ptr() if addend is not None else None,
                                              bias.data_ptr() if bias is not None else None, act.data_ptr() if act is not None else None))
            fp = FcFilterParams(zonal.data_ptr(), spherical.data_ptr(), phase.data_ptr(), ftype, None, None, None)
            nbytes = plan.ws_fwd
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
            kind, recs = (2, graph.geo_t) if graph.geo_t is not None else ((1, graph.rec_t) if graph.factored else (0, graph.sten_t))
            check(lib.fc_forward_params(_p(x), _p(recs), plan.cref_t, kind, ctypes.byref(fp), _p(wpk_f), _p(wpk_b), _p(y),
                                        _p(ws) if ws is not None else None, nbytes, plan.dref, plan.records, epi, st), 'fc_forward_params')
            return (y if bias is None else (y, act)), wpk_b
        pack(plan, wpk_f, wpk_b)
        if between is not None:
            between()
        return _launch_forward(lib, x, graph, wpk_f, plan, O, st, addend=addend, bias=bias), wpk_b
    y = torch.empty((nt, O), dtype=torch.complex64, device=dev)
    act = torch.empty_like(y) if bias is not None else None
    if len(ranges) == 1 and between is not None:
        between()
    for row0, nrows in ranges:
        sub = _row_plan(lib, graph, plan, row0, nrows)
        wpk_f = torch.empty(sub.n_fwd, dtype=torch.float32, device=dev)
        pack(sub, wpk_f, wpk_b if row0 == 0 else None)
        _launch_forward(lib, x, graph, wpk_f, sub, O, st, addend=addend, bias=bias, out=(y, act), row0=row0)
        if row0 == 0 and len(ranges) == 2:
            between()
    return (y if bias is None else (y, act)), wpk_b


def _launch_backward(lib, x, gy, graph, wpk_b, plan, wshape, st, params=None, bias_sum=None):
    """-> (gx, gw_eff, parameter gradients or None).  params = (zonal, spherical, phase, ftype): also the VJP of the filter
    assembly.  One foreign call for the whole pass (fc_backward_all) unless something has to happen between the kernels: a
    partitioned mesh's gradient exchange (graph.on_gx) or the benchmark's per-kernel event brackets.
    bias_sum = (partials, n_parts, g_bias): the fused modReLU's bias-gradient partials, summed by the launch that finishes
    this pass (fc_filter_params' rider; needs params)."""
    O, I, R, F = wshape
    gx = torch.empty_like(x)
    # with module parameters only their gradients are wanted: the (O,I,R,F) tensor is never written (gw_eff = NULL)
    # (the development library's two-kernel finish, FC_SPLIT_FINISH=1, hands gW_eff from one kernel to the other)
    want_gw = params is None or KEEP_GW_EFF or os.environ.get('FC_SPLIT_FINISH', '0') not in ('', '0')
    gw = torch.empty((O, I, R, F), dtype=torch.complex64, device=x.device) if want_gw else None
    nbytes = plan.ws_bwd
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    sten = graph.rec_s if graph.factored else graph.sten_s
    wsp = _p(ws)
    pgrads = fp = None
    if params is not None:
        zonal, spherical, phase, ftype = params
        g_z = torch.empty_like(zonal)
        g_s = torch.empty_like(spherical)
        g_p = torch.empty_like(phase) if ftype == 1 else None
        pgrads = (g_z, g_s, g_p)
        fp = FcFilterParams(zonal.data_ptr(), spherical.data_ptr(), phase.data_ptr(), ftype, g_z.data_ptr(), g_s.data_ptr(),
                            g_p.data_ptr() if g_p is not None else None)
        if bias_sum is not None:
            fp.bias_partials, fp.bias_nparts, fp.g_bias = bias_sum[0].data_ptr(), bias_sum[1], bias_sum[2].data_ptr()
    gwp = _p(gw) if gw is not None else None
    if graph.on_gx is None and _ONE_CALL and not kernel_timer.enabled:
        check(lib.fc_backward_all(_p(x), _p(gy), _p(sten), plan.cref_s, plan.records, _p(wpk_b), _p(gx), gwp,
                                  ctypes.byref(fp) if fp is not None else None, wsp, nbytes, plan.dref, st), 'fc_backward_all')
        return gx, gw, pgrads
    if kernel_timer.enabled and graph.on_gx is None and graph.factored and lib.fc_backward_streams(plan.dref, plan.records):
        # the H-streaming arrangement's two halves bracketed apart (the per-kernel timing pass of bench.py): gather | stream + gx
        with _timed('fc_backward_data'):
            check(lib.fc_backward_gather(_p(gy), _p(sten), plan.cref_s, _p(wpk_b), wsp, nbytes, plan.dref, st), 'fc_backward_gather')
        with _timed('fc_backward_filter'):
            check(lib.fc_backward_stream(_p(x), _p(wpk_b), _p(gx), wsp, nbytes, plan.dref, st), 'fc_backward_stream')
        if fp is not None:
            check(lib.fc_backward_finish_params(gwp, wsp, nbytes, plan.dref, plan.records, ctypes.byref(fp), st), 'fc_backward_finish_params')
        else:
            check(lib.fc_backward_finish(_p(gw), wsp, nbytes, plan.dref, plan.records, st), 'fc_backward_finish')
        return gx, gw, pgrads
    with _timed('fc_backward_data'):
        if graph.factored:
            check(lib.fc_backward_data_factored(_p(x), _p(gy), _p(sten), plan.cref_s, _p(wpk_b), _p(gx), wsp, nbytes, plan.dref,
                                                plan.records, st), 'fc_backward_data_factored')
        else:
            check(lib.fc_backward_data(_p(x), _p(gy), _p(sten), plan.cref_s, _p(wpk_b), _p(gx), wsp, nbytes, plan.dref, st),
                  'fc_backward_data')
    if graph.on_gx is not None:         # gx is complete (in stream order): a partitioned mesh starts returning its halo rows
        graph.on_gx(gx)                 # now, under the filter-gradient kernel
    with _timed('fc_backward_filter'):
        check(lib.fc_backward_filter(_p(x), wsp, nbytes, plan.dref, plan.records, st), 'fc_backward_filter')
    if fp is not None:      # partial sums + parameter gradients in one launch
        check(lib.fc_backward_finish_params(gwp, wsp, nbytes, plan.dref, plan.records, ctypes.byref(fp), st), 'fc_backward_finish_params')
    else:
        check(lib.fc_backward_finish(_p(gw), wsp, nbytes, plan.dref, plan.records, st), 'fc_backward_finish')
    return gx, gw, pgrads


@_lib.keep_mode
class _FieldConvParamFn(torch.autograd.Function):
    """FieldConv straight from the module parameters: the filter assembly of reference
    nn/field_conv.py:10-33 and its autograd twin run as one small HIP kernel each
    (fc_pack_filter_params / fc_filter_param_grads) instead of a dozen torch ops."""

    @staticmethod
    def forward(ctx, x, zonal, spherical, phase, ftype, B, graph):
        lib = _lib.load()
        x = x.contiguous()
        zonal, spherical, phase = zonal.contiguous(), spherical.contiguous(), phase.contiguous()
        O, I, R = zonal.shape[0], zonal.shape[1], zonal.shape[2]
        F = 2 * B + 1
        plan = _conv_plan(lib, graph, I, O, B)
        with _on(x.device):
            st = _stream()

            def pack(pl, wpk_f, wpk_b):
                check(lib.fc_pack_filter_params(_p(zonal), _p(spherical), _p(phase), ftype, _p(wpk_f), _p(wpk_b) if wpk_b is not None else None, pl.dref,
                                                pl.records, st), 'fc_pack_filter_params')
            y, wpk_b = _run_forward(lib, x, graph, plan, O, st, pack, params=(zonal, spherical, phase, ftype))
        ctx.save_for_backward(x, wpk_b, zonal, spherical, phase)
        ctx.graph, ctx.ftype, ctx.wshape = graph, ftype, (O, I, R, F)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, wpk_b, zonal, spherical, phase = ctx.saved_tensors
        O, I, R, F = ctx.wshape
        plan = _conv_plan(lib, ctx.graph, I, O, (F - 1) // 2)
        gy = gy.contiguous()
        with _on(x.device):
            st = _stream()
            gx, _, (g_z, g_s, g_p) = _launch_backward(lib, x, gy, ctx.graph, wpk_b, plan, ctx.wshape, st,
                                                      params=(zonal, spherical, phase, ctx.ftype))
        return gx, g_z, g_s, g_p, None, None, None


@_lib.keep_mode
class _FieldConvActFn(torch.autograd.Function):
    """modReLU(FieldConv(x) [+ addend]) with the residual add and the modReLU in the convolution's epilogue (SURVEY 8 row
    f4; reference nn/fc_resnet_block.py:84-88, nn/tangent_nonlin.py:24-35).  The kernel leaves the pre-activation too; the
    backward pass is the modReLU's VJP on it (fc_tangent_nonlin_backward) followed by the convolution's."""

    @staticmethod
    def forward(ctx, x, zonal, spherical, phase, bias, addend, ftype, B, graph):
        lib = _lib.load()
        x = x.contiguous()
        zonal, spherical, phase = zonal.contiguous(), spherical.contiguous(), phase.contiguous()
        bias = bias.contiguous()
        addend = addend.contiguous() if addend is not None else None
        O, I, R = zonal.shape[0], zonal.shape[1], zonal.shape[2]
        F = 2 * B + 1
        plan = _conv_plan(lib, graph, I, O, B)
        with _on(x.device):
            st = _stream()
            def pack(pl, wpk_f, wpk_b):
                check(lib.fc_pack_filter_params(_p(zonal), _p(spherical), _p(phase), ftype, _p(wpk_f), _p(wpk_b) if wpk_b is not None else None, pl.dref,
                                                pl.records, st), 'fc_pack_filter_params')
            (pre, act), wpk_b = _run_forward(lib, x, graph, plan, O, st, pack, addend=addend, bias=bias,
                                             params=(zonal, spherical, phase, ftype))
        ctx.save_for_backward(x, wpk_b, zonal, spherical, phase, bias, pre)
        ctx.graph, ctx.ftype, ctx.wshape, ctx.has_addend = graph, ftype, (O, I, R, F), addend is not None
        return act

    @staticmethod
    def backward(ctx, g_act):
        lib = _lib.load()
        x, wpk_b, zonal, spherical, phase, bias, pre = ctx.saved_tensors
        O, I, R, F = ctx.wshape
        plan = _conv_plan(lib, ctx.graph, I, O, (F - 1) // 2)
        g_act = g_act.contiguous()
        N = pre.shape[0]
        with _on(x.device):
            st = _stream()
            g_pre = torch.empty_like(pre)
            g_bias = torch.empty_like(bias)
            nbytes = lib.fc_tangent_nonlin_backward_workspace_bytes(N, O)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            # the VJP's first kernel; its bias-gradient partials are summed by the convolution's finishing launch (one launch fewer)
            check(lib.fc_tangent_nonlin_backward_partial(_p(pre), _p(bias), _p(g_act), _p(g_pre), _p(ws), nbytes, N, O, st),
                  'fc_tangent_nonlin_backward_partial')
            gx, _, (g_z, g_s, g_p) = _launch_backward(lib, x, g_pre, ctx.graph, wpk_b, plan, ctx.wshape, st,
                                                      params=(zonal, spherical, phase, ctx.ftype),
                                                      bias_sum=(ws, lib.fc_tangent_nonlin_backward_groups(N), g_bias))
        return gx, g_z, g_s, g_p, g_bias, (g_pre if ctx.has_addend else None), None, None, None


def field_conv_act(x, zonal, spherical, phase, ftype, band_limit, graph, bias, addend=None):
    """modReLU(FieldConv(x) + addend) as one forward kernel (see _FieldConvActFn); `bias` (1,O) or (O,), `addend` (N,O)
    complex64 or None.  Layers wider than the kernels' channel block fall back to the separate operators."""
    _require_device(x, 'field_conv')
    O, I = zonal.shape[0], zonal.shape[1]
    if _run_time_path(x, graph):        # no specialised kernels for this (n_rings, band_limit), or double precision: separate operators
        h = field_conv_params(x, zonal, spherical, phase, ftype, band_limit, graph)
        if addend is not None:
            h = h + addend
        return tangent_nonlin(h, bias)
    blk = _channel_block(graph, I, O, band_limit)
    if I > blk or O > blk or os.environ.get('FIELDCONV_NO_FUSED_EPILOGUE', '0') == '1':
        h = field_conv_params(x, zonal, spherical, phase, ftype, band_limit, graph)
        if addend is not None:
            h = h + addend
        return tangent_nonlin(h, bias)
    if x.dtype != torch.complex64:
        raise ValueError('field_conv expects complex64 features')
    if x.dim() != 2 or x.shape[0] != graph.N or x.shape[1] != I:
        raise ValueError(f'x has shape {tuple(x.shape)}, expected ({graph.N}, {I})')
    if zonal.shape[2] != graph.R or 2 * band_limit + 1 != graph.F:
        raise ValueError(f'stencil is (E,{graph.R},{graph.F}) but the filter has n_rings={zonal.shape[2]}, band_limit={band_limit}')
    if bias.numel() != O:
        raise ValueError(f'bias has {bias.numel()} channels, the convolution {O}')
    if addend is not None and (addend.dtype != torch.complex64 or tuple(addend.shape) != (graph.n_targets, O)):
        raise ValueError(f'addend must be complex64 of shape ({graph.n_targets}, {O})')
    return _FieldConvActFn.apply(x, zonal, spherical, phase, bias.reshape(-1), addend, int(ftype), int(band_limit), graph)


def _dtype_code(t):
    """fc_dtype of a complex (or real) tensor: 0 = float32-based, 1 = float64-based."""
    return 1 if t.dtype in (torch.complex128, torch.float64) else 0


def _cgemm(lib, A, B, C, M, N, K, sam, sak, sbk, sbn, conj_b, alpha):
    """C (M,N) = alpha * A . op(B) on the matrix pipe (csrc/fc_cgemm.hip); strides in complex elements."""
    dt = _dtype_code(C)
    nbytes = lib.fc_cgemm_workspace_bytes(M, N, K, dt)          # > 0: a small output with a long contraction goes split along k
    ws = torch.empty(nbytes, dtype=torch.uint8, device=C.device) if nbytes else None
    check(lib.fc_cgemm(_p(A), _p(B), _p(C), M, N, K, sam, sak, sbk, sbn, 1 if conj_b else 0, float(alpha), dt,
                       _p(ws) if ws is not None else None, nbytes, _stream()), 'fc_cgemm')
    return C


@_lib.keep_mode
class _GenericFieldConvFn(torch.autograd.Function):
    """FieldConv for (n_rings, band_limit) pairs without specialised kernels (n_rings > 8, band_limit > 3 or 0) and for
    complex128 features of any shape (the reference's modules run under .double()): the gather and the scatter are run-time
    HIP kernels on dense stencil rows (csrc/fc_generic.hip), the three contractions with the filter complex GEMMs on the
    matrix pipe (csrc/fc_cgemm.hip), all in the tensors' own precision.  Same arithmetic as reference nn/field_conv.py:128-137
    and its autograd; any channel count."""

    @staticmethod
    def forward(ctx, x, w_eff, graph):
        lib = _lib.load()
        x = x.contiguous()
        O, I, R, F = w_eff.shape
        B = (F - 1) // 2
        nt, K = graph.n_targets, I * R * F
        with _on(x.device):
            contrib = _GenericFieldConvFn._gather(lib, x, graph, I, R, B)
            y = torch.empty((nt, O), dtype=x.dtype, device=x.device)
            _cgemm(lib, contrib, w_eff, y, nt, O, K, K, 1, 1, K, False, 1.0 / F)          # y = contrib . W^T / F
        ctx.save_for_backward(x, w_eff)
        ctx.graph = graph
        return y

    @staticmethod
    def _gather(lib, x, graph, I, R, B):
        F = 2 * B + 1
        contrib = torch.empty((graph.n_targets, I, R, F), dtype=x.dtype, device=x.device)
        by_t = _csr(graph.rowptr_t, graph.nbr_t, None)
        check(lib.fc_generic_gather(_p(x), _p(graph.sten_t), ctypes.byref(by_t), _p(contrib), graph.n_targets, I, R, B, _dtype_code(x),
                                    _stream()), 'fc_generic_gather')
        return contrib

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, w_eff = ctx.saved_tensors
        graph = ctx.graph
        O, I, R, F = w_eff.shape
        B = (F - 1) // 2
        nt, K = graph.n_targets, I * R * F
        gy = gy.contiguous()
        with _on(x.device):
            contrib = _GenericFieldConvFn._gather(lib, x, graph, I, R, B)      # recomputed, not kept between the passes
            g_contrib = torch.empty((nt, K), dtype=x.dtype, device=x.device)
            _cgemm(lib, gy, w_eff, g_contrib, nt, K, O, O, 1, K, 1, True, 1.0 / F)        # gy . conj(W) / F
            gw = torch.empty((O, I, R, F), dtype=x.dtype, device=x.device)
            _cgemm(lib, gy, contrib, gw, O, K, nt, 1, O, K, 1, True, 1.0 / F)             # gy^T . conj(contrib) / F
            gx = torch.empty_like(x)
            by_s = _csr(graph.rowptr_s, graph.nbr_s, None)
            check(lib.fc_generic_scatter(_p(x), _p(g_contrib), _p(graph.sten_s), ctypes.byref(by_s), _p(gx), graph.N, I, R, B, _dtype_code(x),
                                         _stream()), 'fc_generic_scatter')
        return gx, gw, None


def _generic_field_conv(x, w_eff, graph):
    if graph.sten_t is None or graph.sten_s is None:
        if x.dtype == torch.complex128:
            raise _lib.FieldConvNativeError('complex128 features need a complex128 stencil (the support graph at hand was built from '
                                            'float32 data: records only)')
        raise _lib.FieldConvNativeError('the run-time FieldConv path needs a support graph with dense stencil rows')
    if x.dtype not in (torch.complex64, torch.complex128) or w_eff.dtype != x.dtype or graph.sten_t.dtype != x.dtype:
        raise ValueError(f'field_conv expects features, filter and stencil of one complex dtype, got {x.dtype}, {w_eff.dtype}, '
                         f'{graph.sten_t.dtype}')
    if x.dim() != 2 or x.shape[0] != graph.N or x.shape[1] != w_eff.shape[1]:
        raise ValueError(f'x has shape {tuple(x.shape)}, expected ({graph.N}, {w_eff.shape[1]})')
    if w_eff.shape[2] != graph.R or w_eff.shape[3] != graph.F:
        raise ValueError(f'stencil is (E,{graph.R},{graph.F}) but the filter is {tuple(w_eff.shape)}')
    return _GenericFieldConvFn.apply(x, w_eff.contiguous(), graph)


def _run_time_path(x, graph):
    """True when a convolution takes the run-time kernels: no specialised kernels for the stencil's shape, or double precision."""
    return x.dtype == torch.complex128 or not _compiled(graph)


def _compiled(graph):
    return bool(_lib.load().fc_shape_compiled(graph.R, (graph.F - 1) // 2)) and graph.F % 2 == 1


MAX_CHANNELS = 64      # one channel per lane in the kernels' gather phases (csrc/fc_kernels.hpp: kMaxChannels)


@_lib.keep_mode
class _WideFieldConvFn(torch.autograd.Function):
    """FieldConv wider than the kernels' channel block (reference nn/field_conv.py:62 takes any width): channel blocks enqueued
    by ONE native call per pass (csrc/fc_wide.hip: fc_forward_wide / fc_backward_wide) -- block copies, filter packing from the
    parameter blocks, the sum over input blocks in the convolution's epilogue, block-wise parameter gradients; no Python loop and
    no concatenation.  Either the module parameters (params = (zonal, spherical, phase), w_eff None) or an explicit filter."""

    @staticmethod
    def forward(ctx, x, zonal, spherical, phase, w_eff, ftype, B, graph, blk):
        lib = _lib.load()
        x = x.contiguous()
        explicit = w_eff is not None
        if explicit:
            w_eff = w_eff.contiguous()
            O, I = w_eff.shape[0], w_eff.shape[1]
        else:
            zonal, spherical, phase = zonal.contiguous(), spherical.contiguous(), phase.contiguous()
            O, I = zonal.shape[0], zonal.shape[1]
        dims = make_dims(graph, I, O, B)
        records = 1 if graph.factored else 0
        kind, recs = (2, graph.geo_t) if graph.geo_t is not None else ((1, graph.rec_t) if graph.factored else (0, graph.sten_t))
        by_t = _csr(graph.rowptr_t, graph.nbr_t, graph.runs_t)
        with _on(x.device):
            nbytes = lib.fc_wide_workspace_bytes(ctypes.byref(dims), blk, records, 0)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            y = torch.empty((graph.N, O), dtype=torch.complex64, device=x.device)
            fp = None if explicit else FcFilterParams(zonal.data_ptr(), spherical.data_ptr(), phase.data_ptr(), ftype, None, None, None)
            check(lib.fc_forward_wide(_p(x), _p(recs), ctypes.byref(by_t), kind, ctypes.byref(fp) if fp is not None else None,
                                      _p(w_eff) if explicit else None, _p(y), _p(ws), nbytes, ctypes.byref(dims), records, blk, _stream()),
                  'fc_forward_wide')
        ctx.save_for_backward(x, *( (w_eff,) if explicit else (zonal, spherical, phase) ))
        ctx.graph, ctx.ftype, ctx.B, ctx.blk, ctx.explicit, ctx.io = graph, ftype, B, blk, explicit, (I, O)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        graph, blk = ctx.graph, ctx.blk
        I, O = ctx.io
        x = ctx.saved_tensors[0]
        gy = gy.contiguous()
        dims = make_dims(graph, I, O, ctx.B)
        records = 1 if graph.factored else 0
        sten = graph.rec_s if graph.factored else graph.sten_s
        by_s = _csr(graph.rowptr_s, graph.nbr_s, graph.runs_s)
        with _on(x.device):
            nbytes = lib.fc_wide_workspace_bytes(ctypes.byref(dims), blk, records, 1)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            gx = torch.empty_like(x)
            if ctx.explicit:
                w_eff = ctx.saved_tensors[1]
                gw = torch.empty_like(w_eff)
                check(lib.fc_backward_wide(_p(x), _p(gy), _p(sten), ctypes.byref(by_s), records, None, _p(w_eff), _p(gw), _p(gx), _p(ws), nbytes,
                                           ctypes.byref(dims), blk, _stream()), 'fc_backward_wide')
                return gx, None, None, None, gw, None, None, None, None
            zonal, spherical, phase = ctx.saved_tensors[1:]
            g_z, g_s = torch.empty_like(zonal), torch.empty_like(spherical)
            g_p = torch.empty_like(phase) if ctx.ftype == 1 else None
            fp = FcFilterParams(zonal.data_ptr(), spherical.data_ptr(), phase.data_ptr(), ctx.ftype, g_z.data_ptr(), g_s.data_ptr(),
                                g_p.data_ptr() if g_p is not None else None)
            check(lib.fc_backward_wide(_p(x), _p(gy), _p(sten), ctypes.byref(by_s), records, ctypes.byref(fp), None, None, _p(gx), _p(ws), nbytes,
                                       ctypes.byref(dims), blk, _stream()), 'fc_backward_wide')
        return gx, g_z, g_s, g_p, None, None, None, None, None


def _wide_checks(x, graph, I, R, F):
    if x.dtype != torch.complex64:
        raise ValueError('field_conv expects complex64 features')
    if x.dim() != 2 or x.shape[0] != graph.N or x.shape[1] != I:
        raise ValueError(f'x has shape {tuple(x.shape)}, expected ({graph.N}, {I})')
    if R != graph.R or F != graph.F:
        raise ValueError(f'stencil is (E,{graph.R},{graph.F}) but the filter has n_rings={R}, 2 band_limit + 1 = {F}')
    if graph.n_targets != graph.N or graph.forward_split is not None or graph.on_gx is not None:
        raise _lib.FieldConvNativeError('layers wider than the channel block are not available on a partitioned mesh '
                                        '(restricted targets / exchange hooks)')


def _channel_block(graph, I, O, B):
    """Widest channel block (<= 64) the compiled kernels take for this stencil shape: 64 in general, less when slab,
    partial sums and record ring would not fit the CU's LDS (e.g. 8 rings with more than 56 channels)."""
    cached = graph._plans.get(('block', B))
    if cached is not None:
        return cached
    lib = _lib.load()
    blk = MAX_CHANNELS
    while blk > 8 and not lib.fc_supported(ctypes.byref(FcDims(graph.N, graph.E, blk, blk, graph.R, int(B)))):
        blk -= 8
    graph._plans[('block', B)] = blk
    return blk


def field_conv_params(x, zonal, spherical, phase, ftype, band_limit, graph):
    """FieldConv from the raw module parameters (see FieldConv.forward).  Layers wider than the kernels' channel block
    (64, fewer for the largest ring counts) run as blocks of input x output channels, enqueued by one native call per pass
    (_WideFieldConvFn): the operator is linear in the input channels and independent across output channels."""
    _require_device(x, 'field_conv')
    O, I = zonal.shape[0], zonal.shape[1]
    if _run_time_path(x, graph):
        if zonal.shape[2] != graph.R or 2 * band_limit + 1 != graph.F:
            raise ValueError(f'stencil is (E,{graph.R},{graph.F}) but the filter has n_rings={zonal.shape[2]}, band_limit={band_limit}')
        from .nn.field_conv import effective_filter          # the (tiny) assembly and its autograd in torch
        return _generic_field_conv(x, effective_filter(zonal, spherical, phase, int(ftype), int(band_limit)), graph)
    blk = _channel_block(graph, I, O, band_limit)
    if I > blk or O > blk:              # channel blocks, enqueued natively (csrc/fc_wide.hip)
        _wide_checks(x, graph, I, zonal.shape[2], 2 * band_limit + 1)
        return _WideFieldConvFn.apply(x, zonal, spherical, phase, None, int(ftype), int(band_limit), graph, int(blk))
    if x.dtype != torch.complex64:
        raise ValueError('field_conv expects complex64 features')
    if x.dim() != 2 or x.shape[0] != graph.N or x.shape[1] != zonal.shape[1]:
        raise ValueError(f'x has shape {tuple(x.shape)}, expected ({graph.N}, {zonal.shape[1]})')
    if zonal.shape[2] != graph.R or 2 * band_limit + 1 != graph.F:
        raise ValueError(f'stencil is (E,{graph.R},{graph.F}) but the filter has n_rings={zonal.shape[2]}, band_limit={band_limit}')
    return _FieldConvParamFn.apply(x, zonal, spherical, phase, int(ftype), int(band_limit), graph)


def field_conv(x, w_eff, graph):
    """x (N,I) complex64, w_eff (O,I,R,F) complex64, graph: fieldconv_amd.graph.SupportGraph -> (N,O) complex64; complex128
    throughout takes the run-time kernels in double precision."""
    _require_device(x, 'field_conv')
    if x.dtype == torch.complex128:
        return _generic_field_conv(x, w_eff, graph)
    if x.dtype != torch.complex64 or w_eff.dtype != torch.complex64:
        raise ValueError('field_conv expects complex64 features and filters')
    if x.dim() != 2 or x.shape[0] != graph.N or x.shape[1] != w_eff.shape[1]:
        raise ValueError(f'x has shape {tuple(x.shape)}, expected ({graph.N}, {w_eff.shape[1]})')
    if w_eff.shape[2] != graph.R or w_eff.shape[3] != graph.F:
        raise ValueError(f'stencil is (E,{graph.R},{graph.F}) but the filter is {tuple(w_eff.shape)}')
    O, I = w_eff.shape[0], w_eff.shape[1]
    if not _compiled(graph):
        return _generic_field_conv(x, w_eff, graph)
    blk = _channel_block(graph, I, O, (graph.F - 1) // 2)
    if I > blk or O > blk:              # channel blocks, as in field_conv_params
        _wide_checks(x, graph, I, w_eff.shape[2], w_eff.shape[3])
        return _WideFieldConvFn.apply(x, None, None, None, w_eff, 0, (graph.F - 1) // 2, graph, int(blk))
    return _FieldConvFn.apply(x, w_eff, graph)


class _TangentLinFn(torch.autograd.Function):
    """reference nn/tangent_lin.py:27-29"""

    @staticmethod
    def forward(ctx, x, re_w, im_w):
        lib = _lib.load()
        x = x.contiguous()
        re_w = re_w.contiguous()
        im_w = im_w.contiguous()
        O, I = re_w.shape
        N = x.shape[0]
        with _on(x.device):
            y = torch.empty((N, O), dtype=torch.complex64, device=x.device)
            check(lib.fc_tangent_lin_forward(_p(x), _p(re_w), _p(im_w), _p(y), N, I, O, _stream()), 'fc_tangent_lin_forward')
        ctx.save_for_backward(x, re_w, im_w)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, re_w, im_w = ctx.saved_tensors
        O, I = re_w.shape
        N = x.shape[0]
        gy = gy.contiguous()
        with _on(x.device):
            gx = torch.empty_like(x)
            g_re = torch.empty_like(re_w)
            g_im = torch.empty_like(im_w)
            nbytes = lib.fc_tangent_lin_backward_workspace_bytes(N, I, O)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            check(lib.fc_tangent_lin_backward(_p(x), _p(gy), _p(re_w), _p(im_w), _p(gx), _p(g_re), _p(g_im), _p(ws), nbytes,
                                              N, I, O, _stream()), 'fc_tangent_lin_backward')
        return gx, g_re, g_im


class _TangentLinGemmFn(torch.autograd.Function):
    """reference nn/tangent_lin.py:27-29 as three complex GEMMs on the matrix pipe (fc_cgemm): double precision, and mixes wider
    than the 64 channels the LDS-resident fp32 kernel takes"""

    @staticmethod
    def forward(ctx, x, re_w, im_w):
        lib = _lib.load()
        x = x.contiguous()
        wc = torch.complex(re_w, im_w).contiguous()
        O, I = wc.shape
        N = x.shape[0]
        with _on(x.device):
            y = torch.empty((N, O), dtype=x.dtype, device=x.device)
            _cgemm(lib, x, wc, y, N, O, I, I, 1, 1, I, False, 1.0)                        # x . Wc^T
        ctx.save_for_backward(x, wc)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, wc = ctx.saved_tensors
        O, I = wc.shape
        N = x.shape[0]
        gy = gy.contiguous()
        with _on(x.device):
            gx = torch.empty_like(x)
            _cgemm(lib, gy, wc, gx, N, I, O, O, 1, I, 1, True, 1.0)                       # gy . conj(Wc)
            gw = torch.empty_like(wc)
            _cgemm(lib, gy, x, gw, O, I, N, 1, O, I, 1, True, 1.0)                        # gy^T . conj(x)
        return gx, gw.real.contiguous(), gw.imag.contiguous()


def tangent_lin(x, re_w, im_w):
    _require_device(x, 'tangent_lin')
    if x.dtype == torch.complex128:
        if re_w.dtype != torch.float64 or x.dim() != 2 or x.shape[1] != re_w.shape[1]:
            raise ValueError(f'tangent_lin: x {tuple(x.shape)} {x.dtype} does not match weights {tuple(re_w.shape)} {re_w.dtype}')
        return _TangentLinGemmFn.apply(x, re_w, im_w)
    if x.dtype != torch.complex64:
        raise ValueError('tangent_lin expects complex64 features')
    if x.dim() != 2 or x.shape[1] != re_w.shape[1]:
        raise ValueError(f'x has shape {tuple(x.shape)}, expected (N, {re_w.shape[1]})')
    O, I = re_w.shape
    if I > MAX_CHANNELS or O > MAX_CHANNELS:        # the fp32 kernel keeps the whole filter in LDS: wider mixes are plain GEMMs
        return _TangentLinGemmFn.apply(x, re_w, im_w)
    return _TangentLinFn.apply(x, re_w, im_w)


class _TangentNonLinFn(torch.autograd.Function):
    """reference nn/tangent_nonlin.py:24-35"""

    @staticmethod
    def forward(ctx, x, bias):
        lib = _lib.load()
        x = x.contiguous()
        b = bias.contiguous()
        N, C = x.shape
        with _on(x.device):
            y = torch.empty_like(x)
            check(lib.fc_tangent_nonlin_forward(_p(x), _p(b), _p(y), N, C, _stream()), 'fc_tangent_nonlin_forward')
        ctx.save_for_backward(x, b)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, b = ctx.saved_tensors
        N, C = x.shape
        gy = gy.contiguous()
        with _on(x.device):
            gx = torch.empty_like(x)
            gb = torch.empty_like(b)
            nbytes = lib.fc_tangent_nonlin_backward_workspace_bytes(N, C)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            check(lib.fc_tangent_nonlin_backward(_p(x), _p(b), _p(gy), _p(gx), _p(gb), _p(ws), nbytes, N, C, _stream()),
                  'fc_tangent_nonlin_backward')
        return gx, gb


class _TangentNonLinF64Fn(torch.autograd.Function):
    """reference nn/tangent_nonlin.py:24-35 in double precision (csrc/fc_pointwise_f64.hip)"""

    @staticmethod
    def forward(ctx, x, bias):
        lib = _lib.load()
        x = x.contiguous()
        b = bias.contiguous()
        N, C = x.shape
        with _on(x.device):
            y = torch.empty_like(x)
            check(lib.fc_tangent_nonlin_forward_f64(_p(x), _p(b), _p(y), N, C, _stream()), 'fc_tangent_nonlin_forward_f64')
        ctx.save_for_backward(x, b)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, b = ctx.saved_tensors
        N, C = x.shape
        gy = gy.contiguous()
        with _on(x.device):
            gx = torch.empty_like(x)
            gb = torch.empty_like(b)
            nbytes = lib.fc_tangent_nonlin_backward_workspace_bytes_f64(N, C)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            check(lib.fc_tangent_nonlin_backward_f64(_p(x), _p(b), _p(gy), _p(gx), _p(gb), _p(ws), nbytes, N, C, _stream()),
                  'fc_tangent_nonlin_backward_f64')
        return gx, gb


def tangent_nonlin(x, bias):
    _require_device(x, 'tangent_nonlin')
    if x.dtype == torch.complex128:
        if bias.dtype != torch.float64 or x.dim() != 2 or bias.numel() != x.shape[1]:
            raise ValueError(f'tangent_nonlin: x {tuple(x.shape)} {x.dtype} does not match bias {tuple(bias.shape)} {bias.dtype}')
        return _TangentNonLinF64Fn.apply(x, bias)
    if x.dtype != torch.complex64:
        raise ValueError('tangent_nonlin expects complex64 features')
    if x.dim() != 2 or bias.numel() != x.shape[1]:
        raise ValueError(f'x has shape {tuple(x.shape)} but bias has {bias.numel()} channels')
    return _TangentNonLinFn.apply(x, bias)


class _SoftAbsFn(torch.autograd.Function):
    """reference utils/field.py:29-37 as one kernel per pass (fc_soft_abs_forward / _backward)"""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = x.contiguous()
        with _on(x.device):
            y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            check(lib.fc_soft_abs_forward(_p(x), _p(y), x.numel(), _stream()), 'fc_soft_abs_forward')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, = ctx.saved_tensors
        gy = gy.contiguous()
        with _on(x.device):
            gx = torch.empty_like(x)
            check(lib.fc_soft_abs_backward(_p(x), _p(gy), _p(gx), x.numel(), _stream()), 'fc_soft_abs_backward')
        return gx


def soft_abs(x):
    """|x| outside the origin box, 0 inside (reference utils/field.py:29-37): one kernel per pass for complex64 device tensors, the
    branch-free torch formulation (utils/field.py) otherwise (double precision)."""
    if on_device(x) and x.dtype == torch.complex64:
        return _SoftAbsFn.apply(x)
    from .utils.field import softAbs
    return softAbs(x)


def _tkey(t):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version, t.dtype)


def echo_slot_order(csr, ln, wxp, dtype=torch.complex64):
    """ln / wxp in the slot orders of the two edge groupings (what the ECHO kernels stream): built once per mesh and kept with the
    grouping (a SupportGraph or an EdgeCSR) -- not four index_selects per forward and backward pass.  -> (ln_t, wxp_t, ln_s, wxp_s)"""
    key = ('echo_slots', _tkey(ln), _tkey(wxp), dtype)
    hit = csr._plans.get(key)
    if hit is None:
        l, w = ln.to(dtype), wxp.to(dtype)
        hit = csr._plans[key] = (l.index_select(0, csr.perm_t).contiguous(), w.index_select(0, csr.perm_t).contiguous(),
                                 l.index_select(0, csr.perm_s).contiguous(), w.index_select(0, csr.perm_s).contiguous(), (ln, wxp))
    return hit


class _EchoFn(torch.autograd.Function):
    """reference nn/echo.py:94-148 (ECHO.forward); any channel count: the entry points launch the channel blocks"""

    @staticmethod
    def forward(ctx, x, slots, csr, n_bins):
        lib = _lib.load()
        x = x.contiguous()
        N, C = x.shape
        dS = lib.fc_echo_hist_dim(n_bins)
        with _on(x.device):
            hist = torch.empty((N, C, dS), dtype=torch.complex64, device=x.device)
            desc = torch.empty((N, C, dS), dtype=torch.float32, device=x.device)
            by_t = _csr(csr.rowptr_t, csr.nbr_t, None)
            check(lib.fc_echo_forward(_p(x), _p(slots[0]), _p(slots[1]), ctypes.byref(by_t), _p(hist), _p(desc), N, csr.E, C, n_bins,
                                      _stream()), 'fc_echo_forward')
        ctx.save_for_backward(x, hist)
        ctx.csr, ctx.n_bins, ctx.slots = csr, n_bins, slots
        return desc

    @staticmethod
    def backward(ctx, g_desc):
        lib = _lib.load()
        x, hist = ctx.saved_tensors
        csr, slots = ctx.csr, ctx.slots
        N, C = x.shape
        g_desc = g_desc.contiguous()
        with _on(x.device):
            gx = torch.empty_like(x)
            gh = torch.empty_like(hist)
            by_s = _csr(csr.rowptr_s, csr.nbr_s, None)
            check(lib.fc_echo_backward(_p(x), _p(slots[2]), _p(slots[3]), ctypes.byref(by_s), _p(hist), _p(g_desc), _p(gx), _p(gh), N, csr.E, C,
                                       ctx.n_bins, _stream()), 'fc_echo_backward')
        return gx, None, None, None


class _EchoGenericFn(torch.autograd.Function):
    """ECHO descriptors outside the specialised kernels' range -- more than 8 raster bins per unit radius, or complex128 features
    (csrc/fc_lift_echo_generic.hip: run-time loops in the tensors' own precision)"""

    @staticmethod
    def forward(ctx, x, slots, csr, n_bins):
        lib = _lib.load()
        x = x.contiguous()
        N, C = x.shape
        dS = lib.fc_echo_hist_dim_generic(n_bins)
        if dS == 0:
            raise ValueError(f'ECHO: n_bins must be in 1..1024, got {n_bins}')
        dt = _dtype_code(x)
        with _on(x.device):
            hist = torch.empty((N, C, dS), dtype=x.dtype, device=x.device)
            desc = torch.empty((N, C, dS), dtype=x.real.dtype, device=x.device)
            nbytes = lib.fc_echo_generic_workspace_bytes(n_bins)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            by_t = _csr(csr.rowptr_t, csr.nbr_t, None)
            check(lib.fc_echo_forward_generic(_p(x), _p(slots[0]), _p(slots[1]), ctypes.byref(by_t), _p(hist), _p(desc), _p(ws), nbytes, N,
                                              csr.E, C, n_bins, dt, _stream()), 'fc_echo_forward_generic')
        ctx.save_for_backward(x, hist)
        ctx.csr, ctx.n_bins, ctx.slots = csr, n_bins, slots
        return desc

    @staticmethod
    def backward(ctx, g_desc):
        lib = _lib.load()
        x, hist = ctx.saved_tensors
        csr, slots = ctx.csr, ctx.slots
        N, C = x.shape
        g_desc = g_desc.contiguous()
        with _on(x.device):
            gx = torch.empty_like(x)
            gh = torch.empty_like(hist)
            nbytes = lib.fc_echo_generic_workspace_bytes(ctx.n_bins)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            by_s = _csr(csr.rowptr_s, csr.nbr_s, None)
            check(lib.fc_echo_backward_generic(_p(x), _p(slots[2]), _p(slots[3]), ctypes.byref(by_s), _p(hist), _p(g_desc), _p(gx), _p(gh),
                                               _p(ws), nbytes, N, csr.E, C, ctx.n_bins, _dtype_code(x), _stream()), 'fc_echo_backward_generic')
        return gx, None, None, None


def echo_descriptors(x, supp_edges, ln, wxp, n_bins):
    """ECHO descriptors |hist| (N, C, dS) of the tangent field x on the device (reference nn/echo.py:94-148); any channel count, any
    n_bins; complex64 features (the reference's own ECHO is float32-only) or complex128 (run-time kernels)."""
    _require_device(x, 'echo_descriptors')
    if x.dtype not in (torch.complex64, torch.complex128) or x.dim() != 2:
        raise ValueError('echo_descriptors expects complex features of shape (N, C)')
    n_bins = int(n_bins)
    if n_bins < 1:
        raise ValueError(f'ECHO: n_bins must be positive, got {n_bins}')
    from .graph import get_edge_csr
    csr = get_edge_csr(supp_edges, x.shape[0])
    slots = echo_slot_order(csr, ln, wxp, x.dtype)
    if x.dtype == torch.complex128 or _lib.load().fc_echo_hist_dim(n_bins) == 0:
        return _EchoGenericFn.apply(x, slots, csr, n_bins)
    return _EchoFn.apply(x, slots, csr, n_bins)


class _TransFieldFn(torch.autograd.Function):
    """reference nn/trans_field.py:78-113 (TransField.forward)"""

    @staticmethod
    def _stencil(lift_sten):
        # FCPrecomp's stand-in for supp_sten[..., B:B+2]: the kernels read the (E,8) factor table (sten_stride 0), no (E,R,2) array
        from .graph import LiftColumns
        if isinstance(lift_sten, LiftColumns):
            if lift_sten._dense is None:
                return lift_sten.factors, 0
            lift_sten = lift_sten._dense
        # (E,R,2) view of the full stencil (supp_sten[..., B:B+2], reference segmentation.ipynb:204) is read in place
        E, R, two = lift_sten.shape
        st = lift_sten.stride()
        if two == 2 and st[2] == 1 and st[0] == R * st[1] and st[1] >= 2:
            return lift_sten, int(st[1])
        return lift_sten.contiguous(), 2

    @staticmethod
    def forward(ctx, x, sten, stride, zonal_ang, zonal_mag, phase, csr, ftype):
        # (sten, stride) = _TransFieldFn._stencil(lift_sten), resolved by the caller
        lib = _lib.load()
        x = x.contiguous()
        N, Cin = x.shape
        O, _, R = zonal_ang.shape
        zonal_ang, zonal_mag, phase = zonal_ang.contiguous(), zonal_mag.contiguous(), phase.contiguous()
        with _on(x.device):
            y = torch.empty((N, O), dtype=torch.complex64, device=x.device)
            ang = torch.empty((N, Cin, R), dtype=torch.complex64, device=x.device)
            mag = torch.empty((N, Cin, R), dtype=torch.float32, device=x.device)
            s1sum = torch.empty((N, R), dtype=torch.complex64, device=x.device)
            by_t = _csr(csr.rowptr_t, csr.nbr_t, None)
            check(lib.fc_trans_field_forward(_p(x), _p(sten), ctypes.byref(by_t), _p(csr.perm_t), _p(zonal_ang), _p(zonal_mag),
                                             _p(phase), _p(y), _p(ang), _p(mag), _p(s1sum), N, csr.E, Cin, O, R, stride, _stream()),
                  'fc_trans_field_forward')
        ctx.save_for_backward(sten, zonal_ang, zonal_mag, phase, ang, mag, s1sum)
        ctx.csr, ctx.ftype, ctx.stride, ctx.Cin = csr, ftype, stride, Cin
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        sten, zonal_ang, zonal_mag, phase, ang, mag, s1sum = ctx.saved_tensors
        csr, Cin = ctx.csr, ctx.Cin
        O, _, R = zonal_ang.shape
        N = ang.shape[0]
        gy = gy.contiguous()
        with _on(gy.device):
            gx = torch.empty((N, Cin), dtype=torch.float32, device=gy.device)
            g_za, g_zm = torch.empty_like(zonal_ang), torch.empty_like(zonal_mag)
            g_ph = torch.empty_like(phase) if ctx.ftype != 0 else None
            nbytes = lib.fc_trans_field_backward_workspace_bytes(N, Cin, O, R)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=gy.device)
            by_s = _csr(csr.rowptr_s, csr.nbr_s, None)
            check(lib.fc_trans_field_backward(_p(sten), ctypes.byref(by_s), _p(csr.perm_s), _p(zonal_ang), _p(zonal_mag), _p(phase),
                                              _p(ang), _p(mag), _p(s1sum), _p(gy), _p(gx), _p(g_za), _p(g_zm),
                                              _p(g_ph) if g_ph is not None else None, _p(ws), nbytes, N, csr.E, Cin, O, R,
                                              ctx.stride, ctx.ftype, _stream()), 'fc_trans_field_backward')
        return gx, None, None, g_za, g_zm, g_ph, None, None


class _TransFieldGenericFn(torch.autograd.Function):
    """TransField outside the specialised kernels' range -- more than 4 scalar inputs, more than 64 output channels, more than 8 rings,
    or float64 (the reference's TransField / LiftBlock run under .double()): csrc/fc_lift_echo_generic.hip, run-time loops in the
    tensors' own precision, one native call per pass (no Python channel loops)."""

    @staticmethod
    def forward(ctx, x, sten, zonal_ang, zonal_mag, phase, csr, ftype):
        lib = _lib.load()
        x = x.contiguous()
        sten = sten.contiguous()                      # (E, R, >= 2) complex, the x's precision
        N, Cin = x.shape
        O, _, R = zonal_ang.shape
        stride = int(sten.shape[2])
        zonal_ang, zonal_mag, phase = zonal_ang.contiguous(), zonal_mag.contiguous(), phase.contiguous()
        cdt = torch.complex128 if x.dtype == torch.float64 else torch.complex64
        dt = _dtype_code(x)
        with _on(x.device):
            y = torch.empty((N, O), dtype=cdt, device=x.device)
            ang = torch.empty((N, Cin, R), dtype=cdt, device=x.device)
            mag = torch.empty((N, Cin, R), dtype=x.dtype, device=x.device)
            s1sum = torch.empty((N, R), dtype=cdt, device=x.device)
            by_t = _csr(csr.rowptr_t, csr.nbr_t, None)
            check(lib.fc_trans_field_forward_generic(_p(x), _p(sten), ctypes.byref(by_t), _p(csr.perm_t), _p(zonal_ang), _p(zonal_mag),
                                                     _p(phase), _p(y), _p(ang), _p(mag), _p(s1sum), N, csr.E, Cin, O, R, stride, dt, _stream()),
                  'fc_trans_field_forward_generic')
        ctx.save_for_backward(sten, zonal_ang, zonal_mag, phase, ang, mag, s1sum)
        ctx.csr, ctx.ftype, ctx.stride, ctx.Cin, ctx.dt = csr, ftype, stride, Cin, dt
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        sten, zonal_ang, zonal_mag, phase, ang, mag, s1sum = ctx.saved_tensors
        csr, Cin = ctx.csr, ctx.Cin
        O, _, R = zonal_ang.shape
        N = ang.shape[0]
        gy = gy.contiguous()
        with _on(gy.device):
            gx = torch.empty((N, Cin), dtype=mag.dtype, device=gy.device)
            g_za, g_zm = torch.empty_like(zonal_ang), torch.empty_like(zonal_mag)
            g_ph = torch.empty_like(phase) if ctx.ftype != 0 else None
            nbytes = lib.fc_trans_field_backward_generic_workspace_bytes(N, Cin, O, R, ctx.dt)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=gy.device)
            by_s = _csr(csr.rowptr_s, csr.nbr_s, None)
            check(lib.fc_trans_field_backward_generic(_p(sten), ctypes.byref(by_s), _p(csr.perm_s), _p(zonal_ang), _p(zonal_mag), _p(phase),
                                                      _p(ang), _p(mag), _p(s1sum), _p(gy), _p(gx), _p(g_za), _p(g_zm),
                                                      _p(g_ph) if g_ph is not None else None, _p(ws), nbytes, N, csr.E, Cin, O, R, ctx.stride,
                                                      ctx.ftype, ctx.dt, _stream()), 'fc_trans_field_backward_generic')
        return gx, None, g_za, g_zm, g_ph, None, None


def trans_field_specialised(x, lift_sten, zonal_ang):
    """True when the lane-mapped kernels of csrc/fc_trans_field.hip take this call: float32, <= 4 scalar inputs, <= 64 output channels,
    <= 8 rings"""
    O, Cin, R = zonal_ang.shape
    return x.dtype == torch.float32 and lift_sten.dtype == torch.complex64 and Cin <= 4 and O <= MAX_CHANNELS and R <= 8


def trans_field(x, supp_edges, lift_sten, zonal_ang, zonal_mag, phase, ftype):
    """TransField on the device: (N,Cin) real features -> (N,O) complex (reference nn/trans_field.py:78-113); float32 / complex64 or,
    like the reference's module under .double(), float64 / complex128 throughout; any sizes."""
    _require_device(x, 'trans_field')
    if x.dim() != 2 or (x.dtype, lift_sten.dtype) not in ((torch.float32, torch.complex64), (torch.float64, torch.complex128)):
        raise ValueError('trans_field expects float32 features (N, Cin) with a complex64 stencil (E, R, 2), or float64 with complex128')
    if zonal_ang.dtype != x.dtype or zonal_mag.dtype != x.dtype or phase.dtype != x.dtype:
        raise ValueError(f'trans_field: features are {x.dtype} but the filters are {zonal_ang.dtype} (module.double() / .float())')
    O, Cin, R = zonal_ang.shape
    if x.shape[1] != Cin or lift_sten.dim() != 3 or lift_sten.shape[1] != R or lift_sten.shape[2] < 2:
        raise ValueError('trans_field: feature / stencil shapes do not match the zonal filters')
    if lift_sten.shape[2] > 2:          # the reference reads columns 0 and 1 of whatever it is given (classification.ipynb:195
        lift_sten = lift_sten[..., :2]  # passes the full stencil); a strided view, read in place (FCPrecomp's stand-in: materialised)
    from .graph import LiftColumns, get_edge_csr
    csr = get_edge_csr(supp_edges, x.shape[0])
    if not trans_field_specialised(x, lift_sten, zonal_ang):
        if isinstance(lift_sten, LiftColumns):
            lift_sten = lift_sten.materialize()
        return _TransFieldGenericFn.apply(x, lift_sten, zonal_ang, zonal_mag, phase, csr, int(ftype))
    sten, stride = _TransFieldFn._stencil(lift_sten)
    return _TransFieldFn.apply(x, sten, stride, zonal_ang, zonal_mag, phase, csr, int(ftype))
