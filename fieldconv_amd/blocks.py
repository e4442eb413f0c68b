"""Whole network blocks as ONE autograd node and ONE foreign call per pass.

The reference's networks are stacks of three blocks -- FCResNetBlock (reference nn/fc_resnet_block.py:65-88), ECHOBlock
(nn/echo_block.py:73-103), LiftBlock (nn/lift_block.py:35-55) -- trained with batch size 1 on a different ~1k-vertex mesh every step
(segmentation.ipynb:120,137).  At that size the GPU runs a block in ~150 us while a per-operator binding spends ~300 us of host time on
it: five autograd nodes, a dozen `torch.empty`, sixteen ctypes calls.  The functions below hand a whole block pass to the library's
block-level entry points (csrc/fc_blocks.hip: fc_resnet_block_forward / _backward, fc_echo_block_*, fc_lift_block_*): same kernels, same
bits, two foreign calls and two allocations per pass.

Every function returns None when the block-level path does not apply (wide or run-time-path layers, double precision, a partitioned
mesh's exchange hooks, per-kernel timing, a development switch that selects separate operators); the modules then compose the
per-operator functions of functional.py as before.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import FcEchoBlockParams, FcFilterParams, FcLiftBlockParams, FcMesh, FcResnetBlockParams, check
from . import functional as Fn

_NODES = None


def cpp_nodes():
    """The block-level autograd nodes in C++ (fieldconv_amd/csrc_torch/fc_torch_nodes.cpp -> _native/fc_torch_nodes.so), bound to the
    library instance this process loaded; None when the extension is not built for these sources / this torch, or switched off
    (FIELDCONV_CPP_NODES=0): the Python nodes below are then used -- same calls, same results, more host time per node."""
    global _NODES
    if os.environ.get('FIELDCONV_CPP_NODES', '1') == '0':
        return None
    if _NODES is None:
        _NODES = False
        lib = _lib.load()
        if isinstance(lib, ctypes.CDLL):
            from .build import TORCH_NODES_PATH, torch_nodes_needs_build
            if os.path.exists(TORCH_NODES_PATH) and not torch_nodes_needs_build():
              try:
                import importlib.util
                spec = importlib.util.spec_from_file_location('fc_torch_nodes', TORCH_NODES_PATH)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                names = ('fc_resnet_block_forward', 'fc_resnet_block_backward', 'fc_echo_block_forward', 'fc_echo_block_backward',
                         'fc_lift_block_forward', 'fc_lift_block_backward', 'fc_lift_block_workspace_bytes', 'fc_lift_block_saved_bytes',
                         'fc_soft_abs_forward', 'fc_soft_abs_backward', 'fc_echo_head_forward', 'fc_echo_head_backward',
                         'fc_echo_head_forward_workspace_bytes', 'fc_echo_head_backward_workspace_bytes', 'fc_status_string')
                mod.bind({n: ctypes.cast(getattr(lib, n), ctypes.c_void_p).value for n in names})
                _NODES = mod
              except Exception as exc:          # noqa: BLE001  (an ABI or symbol mismatch with the installed torch: say so once, use the Python nodes)
                import warnings
                warnings.warn(f'fieldconv_amd: fc_torch_nodes.so could not be loaded / bound ({type(exc).__name__}: {exc}); '
                              'using the Python autograd nodes (more host time per block)')
            else:
                import warnings
                warnings.warn('fieldconv_amd: fc_torch_nodes.so is not built for these sources (python -m fieldconv_amd.build --nodes); '
                              'using the Python autograd nodes (more host time per block)')
    return _NODES or None


def _graph_ref(nodes, graph, B):
    """the mesh as the C++ nodes take it (built once per graph and band limit; holds the grouping arrays and records)"""
    key = ('graph_ref', int(B))
    ref = graph._plans.get(key)
    if ref is None:
        empty = torch.empty(0, dtype=torch.int32, device=graph.rowptr_t.device)
        if graph.geo_t is not None:
            kind, fwd, bwd = 2, graph.geo_t, graph.rec_s
        elif graph.factored:
            kind, fwd, bwd = 1, graph.rec_t, graph.rec_s
        else:
            kind, fwd, bwd = 0, graph.sten_t, graph.sten_s
        opt = lambda t: t if t is not None else empty
        ref = graph._plans[key] = nodes.GraphRef([graph.rowptr_t, opt(graph.nbr_t), opt(graph.runs_t), graph.rowptr_s, opt(graph.nbr_s),
                                                  opt(graph.runs_s), opt(fwd), opt(bwd)], graph.N, graph.E, graph.R, int(B), kind | (_lib.current_mode() << 8))
    return ref


def _resnet_sizes(lib, graph, mesh, C_in, C_mid, C_out, B):
    """(saved bytes, forward workspace bytes, backward workspace bytes) of an FCResNetBlock on this mesh, cached with the graph"""
    key = ('resnet', C_in, C_mid, C_out, int(B))
    sizes = graph._plans.get(key)
    if sizes is None:
        bp = FcResnetBlockParams(C_in, C_mid, C_out)
        bref = ctypes.byref(bp)
        with Fn._on(graph.rowptr_t.device):      # (the plans follow the CU count of the device the launches will run on, not of the current one)
            sizes = graph._plans[key] = (lib.fc_resnet_block_saved_bytes(mesh.ref, bref), lib.fc_resnet_block_workspace_bytes(mesh.ref, bref, 0),
                                         lib.fc_resnet_block_workspace_bytes(mesh.ref, bref, 1))
    return sizes


def enabled():
    """False under the development switches that ask for separate operators / calls / no edge split (read per call: tests flip them),
    and while the benchmark brackets single kernels with events"""
    env = os.environ
    return (env.get('FIELDCONV_BLOCK_CALLS', '1') != '0' and Fn._ONE_CALL and env.get('FIELDCONV_NO_FUSED_EPILOGUE', '0') != '1'
            and env.get('FIELDCONV_NO_EDGE_SPLIT', '0') != '1' and env.get('FC_SPLIT_FINISH', '0') in ('', '0')
            and not Fn.kernel_timer.enabled)


def _plain_graph(graph):
    """no exchange hooks, no restricted targets: nothing has to happen between the kernels of a pass"""
    return graph.on_gx is None and graph.forward_split is None and graph.n_targets == graph.N


class _Mesh:
    """fc_mesh of a support graph (cached on the graph per band limit; keeps the grouping structs alive)"""
    __slots__ = ('csr_t', 'csr_s', 'struct', 'ref')

    def __init__(self, graph, B):
        self.csr_t = Fn._csr(graph.rowptr_t, graph.nbr_t, graph.runs_t)
        self.csr_s = Fn._csr(graph.rowptr_s, graph.nbr_s, graph.runs_s)
        if graph.geo_t is not None:
            kind, fwd, bwd = 2, graph.geo_t, graph.rec_s
        elif graph.factored:
            kind, fwd, bwd = 1, graph.rec_t, graph.rec_s
        else:
            kind, fwd, bwd = 0, graph.sten_t, graph.sten_s
        self.struct = FcMesh(graph.N, graph.E, graph.R, int(B), kind, ctypes.pointer(self.csr_t), ctypes.pointer(self.csr_s),
                             fwd.data_ptr() if fwd is not None else None, bwd.data_ptr() if bwd is not None else None)
        self.ref = ctypes.byref(self.struct)


def _mesh(graph, B):
    key = ('mesh', int(B))
    m = graph._plans.get(key)
    if m is None:
        m = graph._plans[key] = _Mesh(graph, B)
    return m


def _filter_params(conv_tensors, ftype, grads=None):
    zonal, spherical, phase = conv_tensors
    if grads is None:
        return FcFilterParams(zonal.data_ptr(), spherical.data_ptr(), phase.data_ptr(), ftype, None, None, None)
    return FcFilterParams(zonal.data_ptr(), spherical.data_ptr(), phase.data_ptr(), ftype, grads[0].data_ptr(), grads[1].data_ptr(),
                          grads[2].data_ptr() if grads[2] is not None else None)


def _carve_alloc(shapes, dev, zero=False):
    """one flat float32 buffer and a view of it per shape (None entries stay None and take no room); pieces 16-byte aligned"""
    sizes = []
    for shp in shapes:
        if shp is not None:
            n = 1
            for d in shp:
                n *= d
            sizes.append((n + 3) // 4 * 4)
    flat = (torch.zeros if zero else torch.empty)(sum(sizes), dtype=torch.float32, device=dev)
    parts = iter(flat.split(sizes))
    out = []
    for shp in shapes:
        if shp is None:
            out.append(None)
        else:
            n = 1
            for d in shp:
                n *= d
            out.append(next(parts)[:n].view(shp) if n % 4 else next(parts).view(shp))
    return out


def _u8(nbytes, dev):
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)


# ----------------------------------------------------------------------------------------------------------------- FCResNetBlock
@_lib.keep_mode
class _ResnetBlockFn(torch.autograd.Function):
    """reference nn/fc_resnet_block.py:84-88 as fc_resnet_block_forward / fc_resnet_block_backward"""

    @staticmethod
    def forward(ctx, x, z1, s1, p1, b1, z2, s2, p2, b2, re_w, im_w, ftype, B, graph):
        lib = _lib.load()
        x = x.contiguous()
        tens = [t.contiguous() for t in (z1, s1, p1, b1, z2, s2, p2, b2, re_w, im_w)]
        z1, s1, p1, b1, z2, s2, p2, b2, re_w, im_w = tens
        C_mid, C_in, C_out = z1.shape[0], z1.shape[1], z2.shape[0]
        mesh = _mesh(graph, B)
        bp = FcResnetBlockParams(C_in, C_mid, C_out, _filter_params((z1, s1, p1), ftype), _filter_params((z2, s2, p2), ftype),
                                 b1.data_ptr(), b2.data_ptr(), re_w.data_ptr(), im_w.data_ptr(), None, None, None, None)
        sizes = _resnet_sizes(lib, graph, mesh, C_in, C_mid, C_out, B)
        dev = x.device
        with Fn._on(dev):
            out = torch.empty((graph.N, C_out), dtype=torch.complex64, device=dev)
            saved = _u8(sizes[0], dev)
            ws = _u8(sizes[1], dev)
            check(lib.fc_resnet_block_forward(Fn._p(x), mesh.ref, ctypes.byref(bp), Fn._p(out), Fn._p(saved), sizes[0], Fn._p(ws), sizes[1],
                                              Fn._stream()), 'fc_resnet_block_forward')
        ctx.save_for_backward(x, saved, *tens)
        ctx.graph, ctx.ftype, ctx.B, ctx.sizes = graph, ftype, B, sizes
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        x, saved, z1, s1, p1, b1, z2, s2, p2, b2, re_w, im_w = ctx.saved_tensors
        graph, ftype, sizes = ctx.graph, ctx.ftype, ctx.sizes
        C_mid, C_in, C_out = z1.shape[0], z1.shape[1], z2.shape[0]
        g_out = g_out.contiguous()
        mesh = _mesh(graph, ctx.B)
        dev = x.device
        has_phase = ftype == 1
        shapes = [z1.shape, s1.shape, p1.shape if has_phase else None, b1.shape, z2.shape, s2.shape, p2.shape if has_phase else None,
                  b2.shape, re_w.shape, im_w.shape]
        with Fn._on(dev):
            gx = torch.empty_like(x)
            g_z1, g_s1, g_p1, g_b1, g_z2, g_s2, g_p2, g_b2, g_re, g_im = _carve_alloc(shapes, dev)
            ws = _u8(sizes[2], dev)
            bp = FcResnetBlockParams(C_in, C_mid, C_out, _filter_params((z1, s1, p1), ftype, (g_z1, g_s1, g_p1)),
                                     _filter_params((z2, s2, p2), ftype, (g_z2, g_s2, g_p2)), b1.data_ptr(), b2.data_ptr(),
                                     re_w.data_ptr(), im_w.data_ptr(), g_b1.data_ptr(), g_b2.data_ptr(), g_re.data_ptr(), g_im.data_ptr())
            check(lib.fc_resnet_block_backward(Fn._p(x), Fn._p(g_out), mesh.ref, ctypes.byref(bp), Fn._p(saved), sizes[0], Fn._p(gx), Fn._p(ws),
                                               sizes[2], Fn._stream()), 'fc_resnet_block_backward')
        return gx, g_z1, g_s1, g_p1, g_b1, g_z2, g_s2, g_p2, g_b2, g_re, g_im, None, None, None


def _conv_ok(conv, graph, x, channels=None):
    """the block-level entry points take what the fused-epilogue path takes: compiled (n_rings, band_limit), float32, one channel block.
    x: the block's input; `channels`: the width this convolution sees when it is not the block's first"""
    if x.dtype != torch.complex64 or x.dim() != 2 or x.shape[0] != graph.N or (x.shape[1] if channels is None else channels) != conv.in_channels:
        return False
    key = ('block_ok', conv.in_channels, conv.out_channels, conv.B, conv.R)
    ok = graph._plans.get(key)
    if ok is None:          # a property of (graph, layer shape): decided once
        ok = not Fn._run_time_path(x, graph) and conv.R == graph.R and 2 * conv.B + 1 == graph.F
        if ok:
            blk = Fn._channel_block(graph, conv.in_channels, conv.out_channels, conv.B)
            ok = conv.in_channels <= blk and conv.out_channels <= blk
        graph._plans[key] = ok
    return ok


def _conv_tensors(conv):
    """(zonal, spherical, phase) of a FieldConv without nn.Module.__getattr__'s fallback chain (phase is a parameter for ftype 1, else a buffer)"""
    p = conv._parameters
    ph = p.get('phase')
    return p['zonal'], p['spherical'], ph if ph is not None else conv._buffers['phase']


def resnet_block(block, x, graph):
    """FCResNetBlock.forward through the block-level entry points, or None when they do not apply"""
    if not (enabled() and Fn.on_device(x) and _plain_graph(graph)):
        return None
    mods = block._modules
    c1, c2, res = mods['conv1'], mods['conv2'], mods['res']
    if c1.ftype != c2.ftype or c1.B != c2.B or not _conv_ok(c1, graph, x) or not _conv_ok(c2, graph, x, c1.out_channels):
        return None
    if res.in_channels > Fn.MAX_CHANNELS or res.out_channels > Fn.MAX_CHANNELS:
        return None
    z1, s1, p1 = _conv_tensors(c1)
    z2, s2, p2 = _conv_tensors(c2)
    rp = res._parameters
    nodes = cpp_nodes()
    if nodes is not None:                      # the same node in C++ (csrc_torch/fc_torch_nodes.cpp): no interpreter in either pass
        B = int(c1.B)
        sizes = graph._plans.get(('resnet', c1.in_channels, c1.out_channels, c2.out_channels, B))
        if sizes is None:
            sizes = _resnet_sizes(_lib.load(), graph, _mesh(graph, B), c1.in_channels, c1.out_channels, c2.out_channels, B)
        return nodes.resnet_block(x, z1, s1, p1, mods['nonlin1']._parameters['bias'], z2, s2, p2, mods['nonlin2']._parameters['bias'],
                                  rp['Re'], rp['Im'], _graph_ref(nodes, graph, B), int(c1.ftype), sizes[0], sizes[1], sizes[2])
    return _ResnetBlockFn.apply(x, z1, s1, p1, mods['nonlin1']._parameters['bias'], z2, s2, p2, mods['nonlin2']._parameters['bias'],
                                rp['Re'], rp['Im'], int(c1.ftype), int(c1.B), graph)


# --------------------------------------------------------------------------------------------------------------------- ECHOBlock
@_lib.keep_mode
class _EchoBlockFn(torch.autograd.Function):
    """reference nn/echo_block.py:93-94: ECHO(modReLU(conv(x))) as fc_echo_block_forward / fc_echo_block_backward"""

    @staticmethod
    def forward(ctx, x, zonal, spherical, phase, bias, ftype, B, n_des, n_bins, graph, slots):
        lib = _lib.load()
        x = x.contiguous()
        zonal, spherical, phase, bias = zonal.contiguous(), spherical.contiguous(), phase.contiguous(), bias.contiguous()
        C_in = zonal.shape[1]
        mesh = _mesh(graph, B)
        bp = FcEchoBlockParams(C_in, n_des, n_bins, _filter_params((zonal, spherical, phase), ftype), bias.data_ptr(), None)
        key = ('echo_block', C_in, n_des, n_bins, int(B))
        sizes = graph._plans.get(key)
        if sizes is None:
            bref = ctypes.byref(bp)
            with Fn._on(x.device):
                sizes = graph._plans[key] = (lib.fc_echo_block_saved_bytes(mesh.ref, bref), lib.fc_echo_block_workspace_bytes(mesh.ref, bref, 0),
                                             lib.fc_echo_block_workspace_bytes(mesh.ref, bref, 1), lib.fc_echo_hist_dim(n_bins))
        dev = x.device
        with Fn._on(dev):
            desc = torch.empty((graph.N, n_des, sizes[3]), dtype=torch.float32, device=dev)
            saved = _u8(sizes[0], dev)
            ws = _u8(sizes[1], dev)
            check(lib.fc_echo_block_forward(Fn._p(x), mesh.ref, Fn._p(slots[0]), Fn._p(slots[1]), ctypes.byref(bp), Fn._p(desc), Fn._p(saved),
                                            sizes[0], Fn._p(ws), sizes[1], Fn._stream()), 'fc_echo_block_forward')
        ctx.save_for_backward(x, saved, zonal, spherical, phase, bias)
        ctx.graph, ctx.ftype, ctx.B, ctx.sizes, ctx.n_des, ctx.n_bins, ctx.slots = graph, ftype, B, sizes, n_des, n_bins, slots
        return desc

    @staticmethod
    def backward(ctx, g_desc):
        lib = _lib.load()
        x, saved, zonal, spherical, phase, bias = ctx.saved_tensors
        graph, ftype, sizes, slots = ctx.graph, ctx.ftype, ctx.sizes, ctx.slots
        g_desc = g_desc.contiguous()
        mesh = _mesh(graph, ctx.B)
        dev = x.device
        shapes = [zonal.shape, spherical.shape, phase.shape if ftype == 1 else None, bias.shape]
        with Fn._on(dev):
            gx = torch.empty_like(x)
            # the module's bias has in_channels entries of which the first n_des act (reference nn/echo_block.py:57,93): the rest get zero
            g_z, g_s, g_p, g_b = _carve_alloc(shapes, dev, zero=bias.numel() > ctx.n_des)
            ws = _u8(sizes[2], dev)
            bp = FcEchoBlockParams(zonal.shape[1], ctx.n_des, ctx.n_bins, _filter_params((zonal, spherical, phase), ftype, (g_z, g_s, g_p)),
                                   bias.data_ptr(), g_b.data_ptr())
            check(lib.fc_echo_block_backward(Fn._p(x), Fn._p(g_desc), mesh.ref, Fn._p(slots[2]), Fn._p(slots[3]), ctypes.byref(bp), Fn._p(saved),
                                             sizes[0], Fn._p(gx), Fn._p(ws), sizes[2], Fn._stream()), 'fc_echo_block_backward')
        return gx, g_z, g_s, g_p, g_b, None, None, None, None, None, None


def echo_block_descriptors(block, x, graph, ln, wxp):
    """ECHOBlock's tangent-feature half -- ECHO(modReLU(conv(x))) -> (N, n_des, dS) -- or None when the block-level path does not apply"""
    if not (enabled() and Fn.on_device(x) and _plain_graph(graph)) or graph.perm_t is None or graph.nbr_t is None:
        return None
    conv, n_des, n_bins = block.conv, block.n_des, int(block.echo.n_bins)
    if not _conv_ok(conv, graph, x):
        return None
    if not 1 <= n_bins <= 8 or n_des != conv.out_channels:          # (more bins: the run-time ECHO kernels, composed path)
        return None
    slots = Fn.echo_slot_order(graph, ln, wxp)
    nodes = cpp_nodes()
    if nodes is not None:
        lib, B = _lib.load(), int(conv.B)
        key = ('echo_block', conv.in_channels, int(n_des), n_bins, B)
        sizes = graph._plans.get(key)
        if sizes is None:
            bref = ctypes.byref(FcEchoBlockParams(conv.in_channels, int(n_des), n_bins))
            mesh = _mesh(graph, B)
            sizes = graph._plans[key] = (lib.fc_echo_block_saved_bytes(mesh.ref, bref), lib.fc_echo_block_workspace_bytes(mesh.ref, bref, 0),
                                         lib.fc_echo_block_workspace_bytes(mesh.ref, bref, 1), lib.fc_echo_hist_dim(n_bins))
        z, sp, ph = _conv_tensors(conv)
        return nodes.echo_block(x, z, sp, ph, block._modules['nonlin']._parameters['bias'], _graph_ref(nodes, graph, B), list(slots[:4]),
                                int(conv.ftype), int(n_des), n_bins, sizes[3], sizes[0], sizes[1], sizes[2])
    return _EchoBlockFn.apply(x, conv.zonal, conv.spherical, conv.phase, block.nonlin.bias, int(conv.ftype), int(conv.B), int(n_des), n_bins,
                              graph, slots)


class _EchoTailFn(torch.autograd.Function):
    """ECHOBlock behind its descriptors (reference nn/echo_block.py:95-103): lin3(relu(lin2(relu(lin1(d))))) + res(softAbs(x)) as ONE autograd
    node.  The arithmetic is the reference's -- four dense layers on hipBLASLt through torch.addmm / torch.mm, softAbs through
    fc_soft_abs_* -- with the backward pass written out (linear and ReLU VJPs), so that the host pays one node instead of fourteen:
    on the reference's ~1k-vertex meshes the step is bound by what the host spends per autograd node."""

    @staticmethod
    def forward(ctx, d, x, w1, b1, w2, b2, w3, b3, wr, br):
        lib = _lib.load()
        x = x.contiguous()
        with Fn._on(x.device):
            a = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            check(lib.fc_soft_abs_forward(Fn._p(x), Fn._p(a), x.numel(), Fn._stream()), 'fc_soft_abs_forward')
            h1 = torch.addmm(b1, d, w1.t()).relu_()
            h2 = torch.addmm(b2, h1, w2.t()).relu_()
            y = torch.addmm(b3, h2, w3.t()).add_(torch.addmm(br, a, wr.t()))       # lin3(h2) + res(a), in the reference's order
        ctx.save_for_backward(d, x, a, h1, h2, w1, w2, w3, wr)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        d, x, a, h1, h2, w1, w2, w3, wr = ctx.saved_tensors
        g = g.contiguous()
        with Fn._on(x.device):
            gb = g.sum(0)                                   # lin3.bias and res.bias see the same cotangent
            g_w3 = g.t().mm(h2)
            g_h2 = g.mm(w3).mul_(h2 > 0)
            g_b2 = g_h2.sum(0)
            g_w2 = g_h2.t().mm(h1)
            g_h1 = g_h2.mm(w2).mul_(h1 > 0)
            g_b1 = g_h1.sum(0)
            g_w1 = g_h1.t().mm(d)
            g_d = g_h1.mm(w1)
            g_wr = g.t().mm(a)
            g_a = g.mm(wr)
            gx = torch.empty_like(x)
            check(lib.fc_soft_abs_backward(Fn._p(x), Fn._p(g_a), Fn._p(gx), x.numel(), Fn._stream()), 'fc_soft_abs_backward')
        return g_d, gx, g_w1, g_b1, g_w2, g_b2, g_w3, gb, g_wr, gb.clone()


def _head_params(d, x, w1, w2, w3, wr):
    return int(d.shape[1]), int(w1.shape[0]), int(w2.shape[0]), int(x.shape[1]), int(w3.shape[0])


def head_supported(d, x, w1, w2, w3, wr):
    """the native head's limits (fc_echo_head_params): the reference's 128 / 64 hidden units, at most 64 channels either side"""
    D, H1, H2, C, Q = _head_params(d, x, w1, w2, w3, wr)
    return H1 <= 128 and H2 <= 64 and C <= 64 and Q <= 64 and w1.shape[1] == D and w2.shape[1] == H1 and w3.shape[1] == H2 and \
        wr.shape == (Q, C)


class _EchoHeadFn(torch.autograd.Function):
    """The same tail as fc_echo_head_forward / fc_echo_head_backward (csrc/fc_head.hip): three launches per pass on the fp32 matrix pipe
    instead of ~30 (twelve ATen GEMMs -- the weight gradients contract over the vertices into a handful of tiles -- bias sums, masks, adds)."""

    @staticmethod
    def forward(ctx, d, x, w1, b1, w2, b2, w3, b3, wr, br):
        lib = _lib.load()
        d, x = d.contiguous(), x.contiguous()
        N = int(d.shape[0])
        D, H1, H2, C, Q = _head_params(d, x, w1, w2, w3, wr)
        hp = _lib.FcEchoHeadParams(D, H1, H2, C, Q, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(),
                                   wr.data_ptr(), br.data_ptr(), None, None, None, None, None, None, None, None)
        with Fn._on(x.device):
            nws = lib.fc_echo_head_forward_workspace_bytes(N, ctypes.byref(hp))
            buf = torch.empty(N * (H1 + H2) + (nws + 3) // 4, dtype=torch.float32, device=x.device)
            h1, h2, ws = buf[:N * H1].view(N, H1), buf[N * H1:N * (H1 + H2)].view(N, H2), buf[N * (H1 + H2):]
            y = torch.empty(N, Q, dtype=torch.float32, device=x.device)        # (its own storage: the caller may write into it)
            check(lib.fc_echo_head_forward(Fn._p(d), Fn._p(x), ctypes.byref(hp), Fn._p(h1), Fn._p(h2), Fn._p(y),
                                           Fn._p(ws) if nws else None, nws, N, Fn._stream()), 'fc_echo_head_forward')
        ctx.save_for_backward(d, x, h1, h2, w1, w2, w3, wr)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        d, x, h1, h2, w1, w2, w3, wr = ctx.saved_tensors
        g = g.contiguous()
        N = int(d.shape[0])
        D, H1, H2, C, Q = _head_params(d, x, w1, w2, w3, wr)
        with Fn._on(x.device):
            hp = _lib.FcEchoHeadParams(D, H1, H2, C, Q, w1.data_ptr(), None, w2.data_ptr(), None, w3.data_ptr(), None, wr.data_ptr(), None,
                                       None, None, None, None, None, None, None, None)
            nws = lib.fc_echo_head_backward_workspace_bytes(N, ctypes.byref(hp))

            def carve(sizes, extra=0):
                buf = torch.empty(sum(-(-s // 4) * 4 for s in sizes) + extra, dtype=torch.float32, device=x.device)
                parts, off = [], 0
                for s in sizes:
                    parts.append(buf[off:off + s])
                    off += -(-s // 4) * 4
                return parts, buf[off:]
            # the parameter gradients in a buffer of their own (they may live on as .grad); the flowing gradients and scratch in another
            (g_w1, g_b1, g_w2, g_b2, g_w3, g_b3, g_wr, g_br), _ = carve([H1 * D, H1, H2 * H1, H2, Q * H2, Q, Q * C, Q])
            (g_d, gx, g_h1), ws = carve([N * D, 2 * N * C, N * H1], (nws + 3) // 4)
            hp.g_w1, hp.g_b1, hp.g_w2, hp.g_b2 = g_w1.data_ptr(), g_b1.data_ptr(), g_w2.data_ptr(), g_b2.data_ptr()
            hp.g_w3, hp.g_b3, hp.g_wr, hp.g_br = g_w3.data_ptr(), g_b3.data_ptr(), g_wr.data_ptr(), g_br.data_ptr()
            check(lib.fc_echo_head_backward(Fn._p(d), Fn._p(x), Fn._p(h1), Fn._p(h2), Fn._p(g), ctypes.byref(hp), Fn._p(g_d), Fn._p(gx),
                                            Fn._p(g_h1), Fn._p(ws), nws, N, Fn._stream()), 'fc_echo_head_backward')
        return (g_d.view(N, D), torch.view_as_complex(gx.view(N, C, 2)), g_w1.view(H1, D), g_b1, g_w2.view(H2, H1), g_b2, g_w3.view(Q, H2), g_b3,
                g_wr.view(Q, C), g_br)


def echo_block_tail(block, d, x):
    """ECHOBlock's MLP + residual behind the (N, n_des * dS) descriptors as one autograd node, or None when that does not apply"""
    # (an operator-level node like the convolutions': FIELDCONV_BLOCK_CALLS=0, which composes the BLOCKS of per-operator nodes, keeps it)
    if not Fn.on_device(x) or x.dtype != torch.complex64 or d.dtype != torch.float32:
        return None
    if os.environ.get('FIELDCONV_ECHO_TAIL', '1') == '0':          # development: torch's own Linear / ReLU nodes
        return None
    mods = block._modules
    layers = [mods[name] for name in ('lin1', 'lin2', 'lin3', 'res')]
    tens = []
    for lin in layers:
        w, b = lin._parameters.get('weight'), lin._parameters.get('bias')
        if type(lin) is not torch.nn.Linear or w is None or b is None or w.dtype != torch.float32 or not w.is_contiguous():
            return None
        tens += [w, b]
    # FIELDCONV_ECHO_TAIL=aten (development): the tail composed of ATen GEMMs inside one node, as before the native head
    native = os.environ.get('FIELDCONV_ECHO_TAIL', '1') != 'aten' and head_supported(d, x, tens[0], tens[2], tens[4], tens[6])
    nodes = cpp_nodes()
    if nodes is not None:
        return (nodes.echo_head if native and hasattr(nodes, 'echo_head') else nodes.echo_tail)(d, x, *tens)
    return (_EchoHeadFn if native else _EchoTailFn).apply(d, x, *tens)


# --------------------------------------------------------------------------------------------------------------------- LiftBlock
class _LiftBlockFn(torch.autograd.Function):
    """reference nn/lift_block.py:53-55: modReLU(TransField(x)) as fc_lift_block_forward / fc_lift_block_backward"""

    @staticmethod
    def forward(ctx, x, sten, stride, zonal_ang, zonal_mag, phase, bias, ftype, csr):
        lib = _lib.load()
        x = x.contiguous()
        zonal_ang, zonal_mag, phase, bias = zonal_ang.contiguous(), zonal_mag.contiguous(), phase.contiguous(), bias.contiguous()
        N, C_in = x.shape
        C_out, _, R = zonal_ang.shape
        by_t, by_s = Fn._csr(csr.rowptr_t, csr.nbr_t, None), Fn._csr(csr.rowptr_s, csr.nbr_s, None)
        mesh = FcMesh(N, csr.E, R, 0, 0, ctypes.pointer(by_t), ctypes.pointer(by_s), None, None)
        bp = FcLiftBlockParams(C_in, C_out, ftype, zonal_ang.data_ptr(), zonal_mag.data_ptr(), phase.data_ptr(), bias.data_ptr(),
                               None, None, None, None)
        nsaved = lib.fc_lift_block_saved_bytes(ctypes.byref(mesh), ctypes.byref(bp))
        dev = x.device
        with Fn._on(dev):
            out = torch.empty((N, C_out), dtype=torch.complex64, device=dev)
            saved = _u8(nsaved, dev)
            check(lib.fc_lift_block_forward(Fn._p(x), Fn._p(sten), stride, ctypes.byref(mesh), Fn._p(csr.perm_t), ctypes.byref(bp), Fn._p(out),
                                            Fn._p(saved), nsaved, Fn._stream()), 'fc_lift_block_forward')
        ctx.save_for_backward(sten, saved, zonal_ang, zonal_mag, phase, bias)
        ctx.csr, ctx.ftype, ctx.stride, ctx.dims, ctx.nsaved = csr, ftype, stride, (N, C_in, C_out, R), nsaved
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        sten, saved, zonal_ang, zonal_mag, phase, bias = ctx.saved_tensors
        csr, ftype = ctx.csr, ctx.ftype
        N, C_in, C_out, R = ctx.dims
        g_out = g_out.contiguous()
        by_t, by_s = Fn._csr(csr.rowptr_t, csr.nbr_t, None), Fn._csr(csr.rowptr_s, csr.nbr_s, None)
        mesh = FcMesh(N, csr.E, R, 0, 0, ctypes.pointer(by_t), ctypes.pointer(by_s), None, None)
        dev = g_out.device
        shapes = [zonal_ang.shape, zonal_mag.shape, phase.shape if ftype != 0 else None, bias.shape]
        with Fn._on(dev):
            gx = torch.empty((N, C_in), dtype=torch.float32, device=dev)
            g_za, g_zm, g_ph, g_b = _carve_alloc(shapes, dev)
            bp = FcLiftBlockParams(C_in, C_out, ftype, zonal_ang.data_ptr(), zonal_mag.data_ptr(), phase.data_ptr(), bias.data_ptr(),
                                   g_za.data_ptr(), g_zm.data_ptr(), g_ph.data_ptr() if g_ph is not None else None, g_b.data_ptr())
            nws = lib.fc_lift_block_workspace_bytes(ctypes.byref(mesh), ctypes.byref(bp), 1)
            ws = _u8(nws, dev)
            check(lib.fc_lift_block_backward(Fn._p(g_out), Fn._p(sten), ctx.stride, ctypes.byref(mesh), Fn._p(csr.perm_s), ctypes.byref(bp),
                                             Fn._p(saved), ctx.nsaved, Fn._p(gx), Fn._p(ws), nws, Fn._stream()), 'fc_lift_block_backward')
        return gx, None, None, g_za, g_zm, g_ph, g_b, None, None


def lift_block(block, x, supp_edges, lift_sten):
    """LiftBlock.forward through the block-level entry points, or None when they do not apply"""
    if not (enabled() and Fn.on_device(x)) or x.dim() != 2 or lift_sten.dim() != 3:
        return None
    field = block.field
    O, Cin, R = field.zonalAng.shape
    if x.shape[1] != Cin or lift_sten.shape[1] != R or lift_sten.shape[2] < 2 or not Fn.trans_field_specialised(x, lift_sten, field.zonalAng):
        return None                     # (wide / float64 shapes: the run-time kernels through the composed path)
    if lift_sten.shape[2] > 2:
        lift_sten = lift_sten[..., :2]
    from .graph import get_edge_csr
    csr = get_edge_csr(supp_edges, x.shape[0])
    sten, stride = Fn._TransFieldFn._stencil(lift_sten)         # (FCPrecomp's stand-in: the factor table, stride 0)
    nodes = cpp_nodes()
    if nodes is not None:
        arrays = csr._plans.get('lift_arrays')
        if arrays is None:
            arrays = csr._plans['lift_arrays'] = [csr.rowptr_t, csr.nbr_t, csr.perm_t, csr.rowptr_s, csr.nbr_s, csr.perm_s]
        fp, bias = field._parameters, block._modules['nonlin']._parameters['bias']
        phase = fp.get('phase')
        return nodes.lift_block(x, sten, fp['zonalAng'], fp['zonalMag'], phase if phase is not None else field._buffers['phase'], bias,
                                arrays, int(stride), int(field.ftype), int(csr.E))
    return _LiftBlockFn.apply(x, sten, stride, field.zonalAng, field.zonalMag, field.phase, block.nonlin.bias, int(field.ftype), csr)
