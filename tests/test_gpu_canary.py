"""Guard-band canaries over the C ABI (race detection / sanitizers -- GPU ASan and XNACK are not available on this pool,
hence plain canaries).

Every buffer the host layer hands to libfieldconv_hip.so -- outputs, packed filter images, workspaces, saved buffers, gradient
buffers -- is allocated at exactly the size the library's `*_bytes` / `*_floats` queries return.  Here those allocations are
carved out of larger ones whose margins hold a byte pattern; after a forward + backward pass of each entry-point family the margins
must be untouched: a kernel that writes one element past `[0, *_bytes)` of ANY buffer fails the test.  The record arrays are the one
place where the kernels deliberately READ past the last record (reference-side: transforms/fc_precomp.py pads them); the second
test puts NaNs right behind the documented padding and requires bit-identical, finite results."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GUARD = 4096
PATTERN = 0xA5


class GuardedTorch:
    """Stands in for the `torch` module inside fieldconv_amd's host layer: `empty` / `empty_like` on the GPU return views into
    allocations with GUARD bytes of PATTERN on both sides (and the payload pre-filled with the pattern too)."""

    def __init__(self):
        self.blocks = []

    def __getattr__(self, name):
        return getattr(torch, name)

    def _guarded(self, shape, dtype, device):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        n = int(np.prod(shape)) if len(shape) else 1
        item = torch.empty((), dtype=dtype).element_size()
        nbytes = n * item
        buf = torch.full((GUARD + nbytes + GUARD,), PATTERN, dtype=torch.uint8, device=device)
        self.blocks.append((buf, nbytes))
        if nbytes == 0:
            return torch.empty(shape, dtype=dtype, device=device)
        return buf[GUARD:GUARD + nbytes].view(dtype).view(shape)

    def empty(self, *size, dtype=None, device=None, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        dev = torch.device(device) if device is not None else torch.device('cpu')
        if dev.type != 'cuda' or kw:
            return torch.empty(*size, dtype=dtype, device=device, **kw)
        return self._guarded(size, dtype or torch.float32, dev)

    def empty_like(self, t, **kw):
        if t.device.type != 'cuda' or kw or not t.is_contiguous():
            return torch.empty_like(t, **kw)
        return self._guarded(tuple(t.shape), t.dtype, t.device)

    def check(self, what):
        torch.cuda.synchronize()
        bad = []
        for idx, (buf, nbytes) in enumerate(self.blocks):
            lo, hi = buf[:GUARD], buf[GUARD + nbytes:]
            if not bool((lo == PATTERN).all()) or not bool((hi == PATTERN).all()):
                first = int((hi != PATTERN).nonzero()[0]) if not bool((hi == PATTERN).all()) else -int((lo != PATTERN).nonzero()[-1]) - 1
                bad.append((idx, nbytes, first))
        assert not bad, f'{what}: writes outside a buffer (allocation index, payload bytes, first offending byte past the end / before the start): {bad[:5]}'
        n = len(self.blocks)
        self.blocks = []
        return n


@pytest.fixture()
def guarded(monkeypatch):
    import fieldconv_amd.blocks as blocks
    import fieldconv_amd.functional as Fn
    import fieldconv_amd.graph as graph
    import fieldconv_amd.transforms.fc_precomp as pre
    g = GuardedTorch()
    for mod in (Fn, blocks, graph, pre):
        monkeypatch.setattr(mod, 'torch', g)
    # the host-side nodes in Python: the C++ nodes allocate inside torch's C++ API, where this stand-in does not reach
    monkeypatch.setenv('FIELDCONV_CPP_NODES', '0')
    return g


def _mesh(N, k, B, R, dev, seed=0):
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.transforms import FCPrecomp
    data = sphere_support(N, k=k, seed=seed, support='p95').to(dev)
    pre = FCPrecomp(B, R, data.epsilon)
    edges, sten, ln, wxp = pre(data)
    return data, edges, sten, ln, wxp


def _cplx(shape, gen, dev):
    return torch.complex(torch.randn(shape, generator=gen), torch.randn(shape, generator=gen)).to(dev)


# configs 2, 3 and 5 at their channel counts / band limits (config 2 at its size: the H-streaming backward), and three ragged shapes
SHAPES = [
    pytest.param(20000, 32, 48, 48, 2, 6, id='config2'),
    pytest.param(1024, 128, 48, 48, 2, 6, id='config3'),
    pytest.param(4999, 24, 64, 64, 3, 6, id='config5'),
    pytest.param(8203, 9, 16, 32, 1, 4, id='ragged-stream-sized'),
    pytest.param(777, 11, 5, 7, 2, 3, id='ragged-small'),
    pytest.param(3001, 7, 40, 24, 3, 5, id='ragged-mid'),
]


@pytest.mark.parametrize('N,k,I,O,B,R', SHAPES)
def test_no_write_outside_any_buffer_fieldconv(guarded, N, k, I, O, B, R):
    """FCPrecomp + graph build, filter packing, forward (geometric records, factored records, dense rows), the whole backward pass with
    parameter gradients and with an explicit filter, TangentLin, TangentNonLin."""
    from fieldconv_amd import functional as Fn
    from fieldconv_amd.graph import SupportGraph, get_graph
    from fieldconv_amd.nn import FieldConv
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(N)
    data, edges, sten, ln, wxp = _mesh(N, k, B, R, dev)
    n = guarded.check('FCPrecomp + graph build')
    x = _cplx((N, I), gen, dev).requires_grad_(True)
    gy = _cplx((N, O), gen, dev)
    for ftype in (0, 1, 2):
        conv = FieldConv(I, O, band_limit=B, n_rings=R, ftype=ftype).to(dev)
        y = conv(x, edges, sten)
        torch.autograd.grad(y, [x] + list(conv.parameters()), grad_outputs=gy)
        n += guarded.check(f'FieldConv ftype {ftype} (records)')
    # explicit filter; dense rows and 64-byte factored records as well
    W = (_cplx((O, I, R, 2 * B + 1), gen, dev) * 0.1).requires_grad_(True)
    dense = sten.materialize() if hasattr(sten, 'materialize') else sten
    graphs = {'records': get_graph(edges, sten, N)}
    if N <= 5000:
        graphs['dense rows'] = SupportGraph(edges, dense, N, allow_factored=False)
    for name, g in graphs.items():
        y = Fn.field_conv(x, W, g)
        torch.autograd.grad(y, [x, W], grad_outputs=gy)
        n += guarded.check(f'field_conv explicit filter ({name})')
    re_w, im_w = torch.randn(O, I, device=dev, requires_grad=True), torch.randn(O, I, device=dev, requires_grad=True)
    y = Fn.tangent_lin(x, re_w, im_w)
    torch.autograd.grad(y, [x, re_w, im_w], grad_outputs=gy)
    bias = torch.randn(I, device=dev, requires_grad=True)
    y = Fn.tangent_nonlin(x, bias)
    torch.autograd.grad(y, [x, bias], grad_outputs=_cplx((N, I), gen, dev))
    n += guarded.check('TangentLin / TangentNonLin')
    assert n > 20


@pytest.mark.parametrize('N,k,C,B,R', [pytest.param(1024, 128, 48, 2, 6, id='config3'), pytest.param(4999, 24, 64, 3, 6, id='config5'),
                                       pytest.param(2050, 10, 24, 1, 4, id='ragged')])
def test_no_write_outside_any_buffer_blocks(guarded, N, k, C, B, R):
    """FCResNetBlock, ECHOBlock (descriptor splat + native head) and LiftBlock through the block-level entry points (saved buffers and
    workspaces carved by the library from caller-owned allocations of exactly the queried size)."""
    from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(N + 1)
    data, edges, sten, ln, wxp = _mesh(N, k, B, R, dev, seed=1)
    guarded.check('FCPrecomp + graph build')
    x = _cplx((N, C), gen, dev).requires_grad_(True)
    blk = FCResNetBlock(C, C, band_limit=B, n_rings=R, ftype=1).to(dev)
    y = blk(x, edges, sten)
    torch.autograd.grad(y, [x] + list(blk.parameters()), grad_outputs=_cplx(tuple(y.shape), gen, dev))
    n = guarded.check('FCResNetBlock')
    echo = ECHOBlock(C, 8, band_limit=B, n_rings=R, ftype=1).to(dev)
    d = echo(x, edges, sten, ln, wxp)
    torch.autograd.grad(d, [x] + list(echo.parameters()), grad_outputs=torch.randn(d.shape, generator=gen).to(dev), allow_unused=True)
    n += guarded.check('ECHOBlock')
    lift = LiftBlock(3, C, n_rings=R, ftype=1).to(dev)
    pos = torch.randn(N, 3, generator=gen).to(dev).requires_grad_(True)
    z = lift(pos, edges, sten[..., B:B + 2])
    torch.autograd.grad(z, [pos] + list(lift.parameters()), grad_outputs=_cplx(tuple(z.shape), gen, dev), allow_unused=True)
    n += guarded.check('LiftBlock')
    assert n > 10


@pytest.mark.parametrize('N,k,C,B,R', [pytest.param(20000, 32, 48, 2, 6, id='config2'), pytest.param(1500, 40, 24, 2, 6, id='small')])
def test_record_read_ahead_stays_inside_the_documented_padding(N, k, C, B, R):
    """The kernels stream record chunks past a vertex's (and the array's) last record; FCPrecomp pads the arrays for that
    (transforms/fc_precomp.py: pad_rec, pad_geo).  With the arrays moved to the END of an allocation whose bytes behind the documented
    padding are NaNs instead of zeros, the results must be finite and bit-identical: nothing behind the padding is read into a result."""
    from fieldconv_amd import functional as Fn
    from fieldconv_amd.graph import get_graph
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(5)
    data, edges, sten, ln, wxp = _mesh(N, k, B, R, dev, seed=2)
    g = get_graph(edges, sten, N)
    assert g.factored
    x = _cplx((N, C), gen, dev).requires_grad_(True)
    gy = _cplx((N, C), gen, dev)
    W = (_cplx((C, C, R, 2 * B + 1), gen, dev) * 0.1).requires_grad_(True)

    def run():
        y = Fn.field_conv(x, W, g)
        gx, gw = torch.autograd.grad(y, [x, W], grad_outputs=gy)
        return y.detach().clone(), gx.clone(), gw.clone()

    ref = run()
    originals = {}
    for name in ('rec_t', 'rec_s', 'geo_t'):
        t = getattr(g, name, None)
        if t is None:
            continue
        originals[name] = t
        tail = 1 << 20
        buf = torch.empty(t.numel() * 4 + tail, dtype=torch.uint8, device=dev)
        buf[t.numel() * 4:].view(torch.float32).fill_(float('nan'))
        moved = buf[:t.numel() * 4].view(torch.float32).view(t.shape)
        moved.copy_(t)
        setattr(g, name, moved)
    g._plans.clear()
    try:
        out = run()
    finally:
        for name, t in originals.items():
            setattr(g, name, t)
        g._plans.clear()
    for a, b in zip(ref, out):
        assert bool(torch.isfinite(torch.view_as_real(b)).all())
        assert torch.equal(a, b)
