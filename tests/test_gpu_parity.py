"""Parity of the HIP path (through the C ABI) with the reference: golden vectors captured from the
reference modules, the CPU oracle on seeded inputs, and size-independent properties at the
benchmark size.  Tolerance: max|delta| <= 1e-5 * max|ref| in fp32 (the
reference's own fp32-vs-fp64 error is ~2e-7)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import fieldconv_oracle as orc

pytestmark = pytest.mark.gpu
# FC_MFMA=f16 runs the contractions on single halves (reduced precision, reported separately): the same suite
# then checks that mode against its own, looser gate
REDUCED = os.environ.get('FC_MFMA') == 'f16'
TOL = 5e-3 if REDUCED else 1e-5


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


def D(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def H(t):
    return t.detach().cpu().numpy()


def load_params(module, c, prefix='p_'):
    sd = {k[len(prefix):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in c.items() if k.startswith(prefix)}
    module.load_state_dict(sd)
    return module


FC = {k: v for k, v in load_golden('fieldconv.npz').items() if k.endswith('f32')}


@pytest.mark.parametrize('tag', sorted(FC))
def test_fieldconv_golden(tag, dev):
    from fieldconv_amd.nn import FieldConv
    c = FC[tag]
    ftype, B, R = int(c['ftype']), int(c['B']), int(c['R'])
    conv = FieldConv(c['x'].shape[1], c['y'].shape[1], band_limit=B, n_rings=R, ftype=ftype)
    conv.load_state_dict({'zonal': torch.from_numpy(c['zonal']), 'spherical': torch.from_numpy(c['spherical']),
                          'phase': torch.from_numpy(c['phase'])})
    conv = conv.to(dev)
    x = D(c['x'], dev).requires_grad_(True)
    y = conv(x, D(c['edges'], dev), D(c['sten'], dev))
    assert rel_err(H(y), c['y']) < TOL
    params = dict(conv.named_parameters())
    grads = torch.autograd.grad(y, [x] + list(params.values()), grad_outputs=D(c['gy'], dev))
    assert rel_err(H(grads[0]), c['gx']) < TOL
    for (name, _), g in zip(params.items(), grads[1:]):
        assert rel_err(H(g), c['g_' + name]) < TOL, name


def test_pointwise_golden(dev):
    from fieldconv_amd.nn import TangentLin, TangentNonLin
    c = load_golden('pointwise.npz')['tangent_lin']
    lin = TangentLin(c['Re'].shape[1], c['Re'].shape[0])
    lin.load_state_dict({'Re': torch.from_numpy(c['Re']), 'Im': torch.from_numpy(c['Im'])})
    lin = lin.to(dev)
    x = D(c['x'], dev).requires_grad_(True)
    y = lin(x)
    assert rel_err(H(y), c['y']) < TOL
    gx, gRe, gIm = torch.autograd.grad(y, [x, lin.Re, lin.Im], grad_outputs=D(c['gy'], dev))
    assert rel_err(H(gx), c['gx']) < TOL
    assert rel_err(H(gRe), c['gRe']) < TOL
    assert rel_err(H(gIm), c['gIm']) < TOL

    c = load_golden('pointwise.npz')['tangent_nonlin']
    nl = TangentNonLin(c['bias'].shape[1])
    nl.load_state_dict({'bias': torch.from_numpy(c['bias'])})
    nl = nl.to(dev)
    x = D(c['x'], dev).requires_grad_(True)
    y = nl(x)
    assert rel_err(H(y), c['y']) < TOL
    gx, gb = torch.autograd.grad(y, [x, nl.bias], grad_outputs=D(c['gy'], dev))
    assert rel_err(H(gx), c['gx']) < TOL
    assert rel_err(H(gb), c['gbias']) < TOL
    # origin-box entries pass through bit-exactly (reference tangent_nonlin.py:26)
    org = orc.is_origin(c['x'])
    assert np.array_equal(H(y)[org], c['x'][org])


@pytest.mark.parametrize('N,I,O,offset', [(37, 5, 7, 0), (1000, 48, 48, 0), (333, 64, 33, 0), (20000, 48, 48, 0), (500, 126, 121, 0),
                                          (130, 16, 16, 1), (5, 1, 3, 0), (4099, 24, 40, 0), (2048, 48, 50, 0), (2049, 20, 17, 0)])
def test_pointwise_vs_oracle(N, I, O, offset, dev):
    """Ragged sizes, channel counts off the 16-wide MFMA tile, and an input that starts 8 bytes off a
    16-byte boundary (offset=1 -> scalar load path)."""
    from fieldconv_amd.functional import tangent_lin, tangent_nonlin
    rng = np.random.default_rng(N * 131 + I)
    cplx = lambda *s: (rng.standard_normal(s) + 1j * rng.standard_normal(s)).astype(np.complex64)
    x, gy = cplx(N, I), cplx(N, O)
    x[rng.random((N, I)) < 0.05] = 0          # origin-box entries
    Re, Im = rng.standard_normal((O, I)).astype(np.float32), rng.standard_normal((O, I)).astype(np.float32)

    def dev_c(a):      # optionally misaligned by one complex element
        flat = torch.empty(a.size + offset, dtype=torch.cfloat, device=dev)
        v = flat[offset:].view(a.shape)
        v.copy_(torch.from_numpy(a))
        return v
    xt = dev_c(x).requires_grad_(True)
    Ret, Imt = D(Re, dev).requires_grad_(True), D(Im, dev).requires_grad_(True)
    y = tangent_lin(xt, Ret, Imt)
    x64 = x.astype(np.complex128)
    assert rel_err(H(y), orc.tangent_lin_forward(x64, Re, Im)) < TOL
    gx, gRe, gIm = torch.autograd.grad(y, [xt, Ret, Imt], grad_outputs=dev_c(gy))
    rgx, rgRe, rgIm = orc.tangent_lin_backward(x64, Re, Im, gy.astype(np.complex128))
    assert rel_err(H(gx), rgx) < TOL
    assert rel_err(H(gRe), rgRe) < TOL
    assert rel_err(H(gIm), rgIm) < TOL

    bias = (rng.standard_normal((1, I)) * 0.5).astype(np.float32)
    bt = D(bias, dev).requires_grad_(True)
    gyn = cplx(N, I)
    yn = tangent_nonlin(xt, bt)
    assert rel_err(H(yn), orc.tangent_nonlin_forward(x, bias)) < TOL
    gxn, gb = torch.autograd.grad(yn, [xt, bt], grad_outputs=dev_c(gyn))
    rgxn, rgb = orc.tangent_nonlin_backward(x64, bias, gyn.astype(np.complex128))
    assert rel_err(H(gxn), rgxn) < TOL
    assert rel_err(H(gb).reshape(-1), np.asarray(rgb).reshape(-1)) < TOL
    # bitwise reproducible reductions
    gxn2, gb2 = torch.autograd.grad(tangent_nonlin(xt, bt), [xt, bt], grad_outputs=dev_c(gyn))
    assert torch.equal(gb, gb2)
    _, gRe2, _ = torch.autograd.grad(tangent_lin(xt, Ret, Imt), [xt, Ret, Imt], grad_outputs=dev_c(gy))
    assert torch.equal(gRe, gRe2)


@pytest.mark.parametrize('tag', sorted(load_golden('blocks.npz')))
def test_fc_resnet_block_golden(tag, dev):
    from fieldconv_amd.nn import FCResNetBlock
    c = load_golden('blocks.npz')[tag]
    blk = FCResNetBlock(int(c['Cin']), int(c['Cout']), band_limit=int(c['B']), n_rings=int(c['R']), ftype=int(c['ftype']),
                        frontload=bool(c['frontload']))
    blk = load_params(blk, c).to(dev)
    x = D(c['x'], dev).requires_grad_(True)
    y = blk(x, D(c['edges'], dev), D(c['sten'], dev))
    assert rel_err(H(y), c['y']) < TOL
    params = dict(blk.named_parameters())
    grads = torch.autograd.grad(y, [x] + list(params.values()), grad_outputs=D(c['gy'], dev))
    assert rel_err(H(grads[0]), c['gx']) < 2 * TOL
    for (name, _), g in zip(params.items(), grads[1:]):
        assert rel_err(H(g), c['g_' + name]) < 2 * TOL, name


def test_echo_descriptors_golden(dev):
    """ECHO descriptor splat and its input gradient (HIP kernels, reference nn/echo.py:94-148) against the reference
    run: zero rows, origin-box entries and the integer-coordinate / out-of-disk quirks are in the fixture's inputs."""
    from fieldconv_amd.nn import ECHO
    c = load_golden('echo_lift.npz')['echo']
    m = ECHO(c['x'].shape[1], int(c['n_bins'])).to(dev)
    x = D(c['x'], dev).requires_grad_(True)
    y = m(x, D(c['edges'], dev), D(c['ln'], dev), D(c['wxp'], dev))
    assert y.shape == c['y'].shape
    assert rel_err(H(y), c['y']) < TOL
    gx, = torch.autograd.grad(y, [x], grad_outputs=D(c['gy'], dev))
    assert rel_err(H(gx), c['gx']) < 2e-4          # histogram votes: piecewise-linear, fp32 floor/ceil sensitive
    # run-to-run reproducible (no atomics)
    y2 = m(x, D(c['edges'], dev), D(c['ln'], dev), D(c['wxp'], dev))
    assert torch.equal(y, y2)


def test_echo_block_and_lift_block_golden(dev):
    from fieldconv_amd.nn import ECHOBlock, LiftBlock
    c = load_golden('echo_lift.npz')['echo_block']
    m = ECHOBlock(int(c['Cin']), int(c['Cout']), n_des=int(c['n_des']), n_bins=int(c['n_bins']), band_limit=int(c['B']),
                  n_rings=int(c['R']), ftype=1)
    m = load_params(m, c).to(dev)
    x = D(c['x'], dev).requires_grad_(True)
    y = m(x, D(c['edges'], dev), D(c['sten'], dev), D(c['ln'], dev), D(c['wxp'], dev))
    assert rel_err(H(y), c['y']) < 5 * TOL
    params = dict(m.named_parameters())
    grads = torch.autograd.grad(y, [x] + list(params.values()), grad_outputs=D(c['gy'], dev), allow_unused=True)
    assert rel_err(H(grads[0]), c['gx']) < max(2e-4, 4 * TOL if REDUCED else 0.0)          # histogram votes: piecewise-linear, fp32 floor/ceil sensitive
    # all twelve parameter gradients of the block (convolution filter + phase, modReLU bias, the three MLP layers, the residual)
    worst = {}
    for (name, _), gval in zip(params.items(), grads[1:]):
        assert gval is not None, name
        worst[name] = rel_err(H(gval), c['g_' + name])
    assert len(worst) == 12 and max(worst.values()) < max(2e-4, 4 * TOL if REDUCED else 0.0), worst       # (reduced-precision mode: its own gate)
    for ft in (0, 1):
        c = load_golden('echo_lift.npz')[f'lift_block_t{ft}']
        m = load_params(LiftBlock(int(c['Cin']), int(c['Cout']), n_rings=int(c['R']), ftype=ft), c).to(dev)
        xs = D(c['x'], dev).requires_grad_(True)
        y = m(xs, D(c['edges'], dev), D(c['lift_sten'], dev))
        assert rel_err(H(y), c['y']) < 5 * TOL
        params = dict(m.named_parameters())
        grads = torch.autograd.grad(y, [xs] + list(params.values()), grad_outputs=D(c['gy'], dev))
        assert rel_err(H(grads[0]), c['gx']) < 1e-4
        for (name, _), gval in zip(params.items(), grads[1:]):
            assert rel_err(H(gval), c['g_' + name]) < 1e-4, name


@pytest.mark.parametrize('N,k,Cin,O,R,ftype', [(700, 9, 3, 32, 6, 1), (257, 5, 4, 64, 8, 1), (130, 12, 1, 5, 2, 0), (64, 0, 3, 16, 6, 1),
                                             (100, 80, 3, 48, 6, 1), (90, 40, 2, 16, 4, 1),      # 4 / 2 wavefronts per vertex
                                             (120, 7, 3, 150, 6, 1),                              # output channels in blocks of 64
                                             (150, 9, 7, 20, 5, 1), (90, 6, 10, 70, 6, 0),        # more than four scalar inputs
                                             (80, 8, 3, 16, 10, 1)])                              # more than eight rings
# (shapes outside the lane-mapped kernels' range -- > 4 inputs, > 64 outputs, > 8 rings -- take the run-time kernels of
#  csrc/fc_lift_echo_generic.hip from ONE native call per pass)
def test_trans_field_kernels_vs_host_composite(dev, N, k, Cin, O, R, ftype):
    """The TransField kernels against the oracle's torch restatement run on the CPU in float64 (pinned to the reference
    fixtures by the CPU suite): ragged in-degrees, isolated vertices, a strided stencil view."""
    from fieldconv_amd.nn import TransField
    from oracle.torch_composites import trans_field as trans_field_ref
    g = torch.Generator().manual_seed(N + k)
    E = N * k
    dst = torch.randint(0, N, (E,), generator=g)
    src = torch.randint(0, N, (E,), generator=g)
    edges = torch.stack((src, dst), dim=1)
    full = torch.complex(torch.randn(E, R, 5, generator=g), torch.randn(E, R, 5, generator=g)) * 0.2
    if E:
        full[::7, 0, 2] = 0           # origin-box stencil entries
    x = torch.randn(N, Cin, generator=g)
    gy = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
    torch.manual_seed(N + k)                # the module's xavier initialisation: the bound below depends on the smallest |A|
    m = TransField(Cin, O, n_rings=R, ftype=ftype)
    pr = [p.detach().double().requires_grad_(True) for p in m.parameters()]            # zonalAng, zonalMag (, phase)
    ph = pr[2] if ftype != 0 else m.phase.double()
    xr = x.double().requires_grad_(True)
    yr = trans_field_ref(xr, edges, full[..., 2:4].to(torch.complex128), pr[0], pr[1], ph, ftype)
    gr = torch.autograd.grad(yr, [xr] + pr, grad_outputs=gy.to(torch.complex128))
    m = m.to(dev)
    xd = x.to(dev).requires_grad_(True)
    yd = m(xd, edges.to(dev), full.to(dev)[..., 2:4])
    # the full stencil in place of the two-column slice: columns 0 and 1 are used (reference classification.ipynb:195)
    assert torch.equal(m(xd, edges.to(dev), full.to(dev)[..., 2:]), yd)
    gd = torch.autograd.grad(yd, [xd] + list(m.parameters()), grad_outputs=gy.to(dev))
    assert rel_err(H(yd), yr.detach().numpy()) < 5 * TOL
    for a, b in zip(gd, gr):
        assert rel_err(H(a), b.numpy()) < 3e-4          # d angle(A) ~ 1/|A|: fp32 rounding of A is amplified where |A| is small


@pytest.mark.skipif(REDUCED, reason='eleven layers deep: checks the fp32-grade path')
def test_segmentation_net_golden(dev):
    """Config-3 topology end to end (reference segmentation.ipynb:165-236): LiftBlock -> 4 FCResNetBlocks
    -> ECHOBlock, loss and every parameter gradient against the reference run captured in net.npz."""
    from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock
    c = load_golden('net.npz')['segmentation_net']
    B, R, nf = int(c['B']), int(c['R']), int(c['nf'])
    mods = torch.nn.ModuleDict(dict(
        lift=LiftBlock(3, nf, n_rings=R, ftype=1),
        resnet1=FCResNetBlock(nf, nf, band_limit=B, n_rings=R, ftype=1),
        resnet2=FCResNetBlock(nf, nf, band_limit=B, n_rings=R, ftype=1),
        resnet3=FCResNetBlock(nf, nf, band_limit=B, n_rings=R, ftype=1),
        resnet4=FCResNetBlock(nf, nf, band_limit=B, n_rings=R, ftype=1),
        echo=ECHOBlock(nf, int(c['n_classes']), n_des=int(c['n_des']), n_bins=int(c['n_bins']), band_limit=B, n_rings=R,
                       ftype=1)))
    mods = load_params(mods, c).to(dev)
    edges, sten = D(c['edges'], dev), D(c['sten'], dev)
    x = mods['lift'](D(c['pos'], dev), edges, sten[..., B:B + 2])          # the strided slice the notebook passes
    for name in ('resnet1', 'resnet2', 'resnet3', 'resnet4'):
        x = mods[name](x, edges, sten)
    logits = mods['echo'](x, edges, sten, D(c['ln'], dev), D(c['wxp'], dev))
    loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), D(c['labels'], dev))
    assert rel_err(H(logits), c['logits']) < TOL
    assert abs(float(loss.detach()) - float(c['loss'])) < 1e-5 * max(1.0, abs(float(c['loss'])))
    params = dict(mods.named_parameters())
    grads = torch.autograd.grad(loss, list(params.values()))
    errs = sorted(((rel_err(H(g), c['g_' + name]), name) for (name, _), g in zip(params.items(), grads)), reverse=True)
    worst = errs[0]
    print('logits err %.2e, worst gradient errs %s' % (rel_err(H(logits), c['logits']), ['%s %.1e' % (n, e) for e, n in errs[:6]] + ['...'] + ['%s %.1e' % (n, e) for e, n in errs[-3:]]))
    assert worst[0] < 5 * TOL, worst        # measured 4e-6 through eleven layers


@pytest.mark.skipif(REDUCED, reason='twenty-five layers deep: checks the fp32-grade path')
def test_correspondence_net_golden(dev):
    """BASELINE configs[4]'s network (reference correspondence.ipynb, class Net) at its width -- 64 channels, band limit 3,
    frontload=True in the last block, TangentPerceptron meta-residuals -- through the PRODUCT path end to end: our FCPrecomp
    on the raw log-map (fused graph build, FactoredStencil), the strided lift slice, eight FCResNetBlocks, ECHOBlock, two
    linear layers, cross-entropy, every parameter gradient.  Reference run captured in net_correspondence.npz
    (tests/golden/make_golden.py; parameters from tests/golden/param_fill.py on both sides).

    Gates.  This network amplifies float32 rounding: the reference's own float32 run is 4.5e-4 away from its float64 run at
    the trunk output, and re-running the reference in float32 with the input positions perturbed by a few ulp (relative 3e-7;
    eight such twins, the largest deviation per tensor kept: the amplification is heavy-tailed -- an ECHO vote flips a cell or it
    does not) moves the trunk output by 9e-4, the logits by 5e-5 and the ECHOBlock's filter gradients by up to 5e-2 (modReLU, angle() and
    the ECHO rasterisation are discontinuous or ill-conditioned where |x| is small / a vote sits on a cell border).  A float32
    implementation that rounds differently cannot be closer to the capture than the capture is to that perturbed twin, so
    every gate is a small multiple of the fixture's own yardsticks (cond_*, gcond_<name>, measured by the generator with
    the reference itself); the kernels' accuracy proper is tested at 1e-5 against the oracle, layer by layer, elsewhere --
    tools/check_ring_net.py does it for every convolution of this very step (gW to 5e-7 given its inputs)."""
    from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock, TangentPerceptron
    from fieldconv_amd.transforms import FCPrecomp
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from param_fill import fill_params, grad_sample
    c = load_golden('net_correspondence.npz')['correspondence_net']
    B, R, nf = int(c['B']), int(c['R']), int(c['nf'])
    kw = dict(band_limit=B, n_rings=R, ftype=1)
    mods = torch.nn.ModuleDict(dict(
        lift=LiftBlock(3, 16, n_rings=R, ftype=1),
        resnet1=FCResNetBlock(16, nf, **kw), resnet2=FCResNetBlock(nf, nf, **kw), resnet3=FCResNetBlock(nf, nf, **kw),
        resnet4=FCResNetBlock(nf, nf, **kw), resnet5=FCResNetBlock(nf, nf, **kw), resnet6=FCResNetBlock(nf, nf, **kw),
        resnet7=FCResNetBlock(nf, nf, **kw), resnet8=FCResNetBlock(nf, 16, frontload=True, **kw),
        echo=ECHOBlock(16, nf, n_des=int(c['n_des']), n_bins=int(c['n_bins']), **kw),
        res1=TangentPerceptron(16, nf), res2=TangentPerceptron(nf, nf), res3=TangentPerceptron(nf, nf),
        res4=TangentPerceptron(nf, 16), lin1=torch.nn.Linear(nf, 256), lin2=torch.nn.Linear(256, int(c['n_classes']))))
    assert sum(p.numel() for p in mods.parameters()) == int(c['n_params'])
    mods = fill_params(mods)
    pr = dict(mods.named_parameters())
    probe = np.concatenate([H(pr[n]).reshape(-1)[:: max(1, pr[n].numel() // 64)][:64] for n in ('resnet2.conv1.spherical', 'res3.lin.Im', 'lin2.weight')])
    assert np.array_equal(probe, c['pfill_probe'])          # same parameter values as the generator's reference modules
    mods = mods.to(dev)

    class Mesh:
        pass
    d = Mesh()
    d.logMag, d.logAng, d.w, d.supp_edges, d.xp = (D(c[k], dev) for k in ('logMag', 'logAng', 'w', 'edges', 'xp'))
    edges, sten, ln, wxp = FCPrecomp(B, R, float(c['eps']))(d)
    assert edges.shape[0] == int(c['kept_edges'])
    conv = (edges, sten)
    x1 = mods['lift'](D(c['pos'], dev), edges, sten[..., B:B + 2])
    x = mods['resnet1'](x1, *conv)
    x2 = mods['resnet2'](x, *conv) + mods['res1'](x1)
    x = mods['resnet3'](x2, *conv)
    x3 = mods['resnet4'](x, *conv) + mods['res2'](x2)
    x = mods['resnet5'](x3, *conv)
    x4 = mods['resnet6'](x, *conv) + mods['res3'](x3)
    x = mods['resnet7'](x4, *conv)
    x = mods['resnet8'](x, *conv) + mods['res4'](x4)
    h = mods['echo'](x, edges, sten, ln, wxp)
    logits = mods['lin2'](torch.relu(mods['lin1'](h)))
    loss = torch.nn.functional.cross_entropy(logits, D(c['labels'], dev))
    ref_own = max(rel_err(c['x_last'], c['x_last64']), float(c['cond_x_last']))     # the reference's float32 run against its float64 run / its perturbed twin
    ours = rel_err(H(x), c['x_last64'])
    e_logits = rel_err(H(logits), c['logits'])
    print('trunk output vs float64: ours %.2e, reference float32 %.2e; logits vs float32 capture %.2e (perturbed twin %.2e)' % (
        ours, ref_own, e_logits, float(c['cond_logits'])))
    assert ours < 3 * ref_own
    assert e_logits < max(1e-5, 8 * float(c['cond_logits']))
    assert abs(float(loss.detach()) - float(c['loss'])) < 1e-4 * max(1.0, abs(float(c['loss'])))
    params = dict(mods.named_parameters())
    grads = torch.autograd.grad(loss, list(params.values()))
    report = []
    for (name, _), g in zip(params.items(), grads):
        sub, stats = grad_sample(H(g))
        e = rel_err(sub, c['g_' + name])
        cond = float(c['gcond_' + name])
        report.append((e / max(cond, 5e-6), e, cond, name))
        assert abs(stats[0] - c['gstat_' + name][0]) < 0.05 * c['gstat_' + name][0] + 1e-12, name          # 2-norm of the whole tensor
    report.sort(reverse=True)
    print('gradient samples, worst by (error / perturbed-twin deviation): ' + ', '.join('%s %.1e / %.1e' % (n, e, cd) for _, e, cd, n in report[:5]))
    for ratio, e, cond, name in report:
        assert e < max(2e-5, 4 * cond), (name, e, cond)


# ---------------------------------------------------------------- seeded inputs vs the oracle
def make_case(seed, N, k, I, O, B, R, sort_by_source=True, zero_frac=0.01):
    g = torch.Generator().manual_seed(seed)
    dst = torch.arange(N).repeat_interleave(k)
    src = torch.randint(0, N, (N * k,), generator=g)
    keep = torch.rand(N * k, generator=g) > 0.1
    keep &= (dst % 37) != 5                        # some vertices without in-edges
    src, dst = src[keep], dst[keep]
    perm = torch.randperm(src.numel(), generator=g)
    src, dst = src[perm], dst[perm]
    if sort_by_source:
        o = torch.argsort(src, stable=True)
        src, dst = src[o], dst[o]
    E = src.numel()
    edges = torch.stack((src, dst), 1)
    F = 2 * B + 1
    sten = torch.complex(torch.randn(E, R, F, generator=g), torch.randn(E, R, F, generator=g)) * (0.5 / k ** 0.5)
    x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
    x[torch.rand(N, I, generator=g) < zero_frac] = 0
    x[0, 0] = complex(5e-8, -2e-8)
    gy = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
    W = torch.complex(torch.randn(O, I, R, F, generator=g), torch.randn(O, I, R, F, generator=g)) * (1.0 / (I * R) ** 0.5)
    return edges, sten, x, gy, W


CASES = [
    # seed, N,   k,  I,  O,  B, R, sorted
    (1, 1000, 20, 48, 48, 2, 6, True),      # the benchmark shape, small mesh
    (2, 777, 12, 48, 48, 2, 6, False),     # unsorted edges, ragged tail tile (777 = 48*16 + 9)
    (3, 500, 16, 64, 64, 3, 6, True),      # FAUST config: C=64, B=3
    (4, 300, 9, 16, 16, 2, 6, True),       # SHREC config: C=16
    (5, 260, 9, 48, 8, 2, 6, True),        # ECHOBlock-like width change
    (6, 333, 7, 3, 48, 1, 6, False),
    (7, 150, 40, 32, 33, 2, 8, True),
    (10, 120, 6, 12, 20, 3, 7, True),      # n_rings = 7, band limit 3
    (11, 90, 5, 24, 8, 2, 2, False),       # n_rings = 2
    (8, 100, 5, 17, 5, 1, 3, True),
    (9, 15, 4, 8, 8, 2, 4, True),          # fewer vertices than one tile
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'N%d_k%d_I%d_O%d_B%d_R%d' % c[1:7])
def test_fieldconv_vs_oracle(case, dev):
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    seed, N, k, I, O, B, R, srt = case
    edges, sten, x, gy, W = make_case(seed, N, k, I, O, B, R, srt)
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    xd = x.to(dev).requires_grad_(True)
    Wd = W.to(dev).requires_grad_(True)
    y = field_conv(xd, Wd, graph)
    gx, gW = torch.autograd.grad(y, [xd, Wd], grad_outputs=gy.to(dev))
    y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy())
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    assert rel_err(H(y), y_ref) < TOL
    assert rel_err(H(gx), gx_ref) < TOL
    assert rel_err(H(gW), gW_ref) < TOL


def rowwise_err(a, ref):
    """max over rows of max|delta_row| / max|ref_row| (rows of zeros in ref must be exactly zero)."""
    a, ref = np.asarray(a), np.asarray(ref)
    a, ref = a.reshape(a.shape[0], -1), ref.reshape(ref.shape[0], -1)
    num = np.abs(a - ref).max(axis=1)
    den = np.abs(ref).max(axis=1)
    assert np.all(num[den == 0] == 0)
    return float((num[den > 0] / den[den > 0]).max())


@pytest.mark.parametrize('global_scale', [1.0, 1e-18, 1e12], ids=['unit', 'tiny', 'huge'])
def test_dynamic_range_rowwise(global_scale, dev):
    """The MFMA contractions run on operands split into two halves with one power-of-two scale per
    vertex and per filter row (csrc/fc_tile.hpp).  Features spanning ten orders of magnitude from
    vertex to vertex, filter rows spanning five, an all-zero vertex, an all-zero filter row and
    inputs near the ends of the fp32 exponent range must still come out to fp32 accuracy ROW BY ROW
    (a global max-norm would hide the small rows)."""
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    N, k, I, O, B, R = 400, 12, 48, 40, 2, 6
    edges, sten, x, gy, W = make_case(77, N, k, I, O, B, R, True, zero_frac=0.0)
    g = torch.Generator().manual_seed(5)
    x = x * (10.0 ** (torch.rand(N, 1, generator=g) * 10 - 6)) * global_scale
    x[7] = 0
    W = W * (10.0 ** (torch.rand(O, 1, 1, 1, generator=g) * 5 - 3))
    W[3] = 0
    gy = gy * (10.0 ** (torch.rand(N, 1, generator=g) * 6 - 3))
    x, W, gy = x.to(torch.complex64), W.to(torch.complex64), gy.to(torch.complex64)
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    xd = x.to(dev).requires_grad_(True)
    Wd = W.to(dev).requires_grad_(True)
    y = field_conv(xd, Wd, graph)
    gx, gW = torch.autograd.grad(y, [xd, Wd], grad_outputs=gy.to(dev))
    c128 = lambda t: t.numpy().astype(np.complex128)
    y_ref = orc.fieldconv_forward(c128(x), edges.numpy(), c128(sten), c128(W))
    gx_ref, gW_ref = orc.fieldconv_backward(c128(x), edges.numpy(), c128(sten), c128(W), c128(gy))
    assert np.all(np.isfinite(H(y))) and np.all(np.isfinite(H(gx))) and np.all(np.isfinite(H(gW)))
    ey, egx, egw = rowwise_err(H(y), y_ref), rowwise_err(H(gx), gx_ref), rowwise_err(H(gW), gW_ref)
    print(f'row-wise errors: y {ey:.2e} gx {egx:.2e} gW {egw:.2e}')
    assert ey < TOL and egx < 5 * TOL and egw < TOL


@pytest.mark.parametrize('shape', [(48, 48, 2, 6), (16, 24, 1, 6), (64, 64, 3, 6), (8, 8, 2, 4), (20, 12, 1, 3)],
                         ids=lambda s: 'I%d_O%d_B%d_R%d' % s)
def test_factored_stencil_path_vs_oracle_and_dense(shape, dev):
    """FCPrecomp-built stencils (rank-1, two adjacent rings) take the factored kernels; the same
    inputs through the dense kernels and through the oracle must agree."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    from oracle.torch_composites import FCPrecomp           # the tests build their stencils on the CPU
    I, O, B, R = shape
    N, k = 700, 40                     # 40 > one 16-record chunk: exercises the ring refill
    data = sphere_support(N, k, seed=4)
    data.epsilon = float(data.logMag.max()) * 1.0001          # use every ring
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    g = torch.Generator().manual_seed(31)
    x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
    x[torch.rand(N, I, generator=g) < 0.02] = 0
    gy = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
    F = 2 * B + 1
    W = torch.complex(torch.randn(O, I, R, F, generator=g), torch.randn(O, I, R, F, generator=g)) * (1.0 / (I * R) ** 0.5)
    gf = SupportGraph(edges.to(dev), sten.to(dev), N)
    gd = SupportGraph(edges.to(dev), sten.to(dev), N, allow_factored=False)
    assert (gf.factored or os.environ.get('FIELDCONV_DENSE') == '1') and not gd.factored      # FIELDCONV_DENSE=1: dense rows everywhere
    y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy())
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    for graph in (gf, gd):
        xd = x.to(dev).requires_grad_(True)
        Wd = W.to(dev).requires_grad_(True)
        y = field_conv(xd, Wd, graph)
        gx, gW = torch.autograd.grad(y, [xd, Wd], grad_outputs=gy.to(dev))
        assert rel_err(H(y), y_ref) < TOL
        assert rel_err(H(gx), gx_ref) < TOL
        assert rel_err(H(gW), gW_ref) < TOL


def _random_shapes(n, seed=2024):
    rng = np.random.default_rng(seed)
    wide = os.environ.get('FC_FUZZ_WIDE', '0') == '1'         # also layers wider than 64 channels and shapes of the run-time path
    shapes = []
    for _ in range(n):
        R = int(rng.integers(2, 12 if wide else 9))
        B = int(rng.integers(1, 5 if wide else 4))
        I, O = int(rng.integers(1, 161 if wide else 65)), int(rng.integers(1, 161 if wide else 65))
        N = int(rng.integers(5, 400))
        k = int(rng.integers(1, 45))
        shapes.append((N, k, I, O, B, R, bool(rng.integers(0, 2))))
    return shapes


@pytest.mark.parametrize('shape', _random_shapes(int(os.environ.get('FC_FUZZ_SHAPES', '40')), int(os.environ.get('FC_FUZZ_SEED', '2024'))), ids=lambda s: 'N%d_k%d_I%d_O%d_B%d_R%d_%s' % (s[:6] + ('geo' if s[6] else 'dense',)))
def test_random_shapes_vs_oracle(shape, dev):
    """Seeded sweep over (n_rings, band limit, channel counts, mesh size, degree): every compiled shape class
    (one or two frequency groups, one or two slab buffers, ragged tiles, channels off the tile sizes), on
    FCPrecomp stencils (factored / geometric records) or random dense ones."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    from oracle.torch_composites import FCPrecomp           # the tests build their stencils on the CPU
    N, k, I, O, B, R, geo = shape
    if geo:
        k = min(k, N - 1) if N > 1 else 1
        data = sphere_support(max(N, k + 2), max(k, 2), seed=N)
        N = data.num_nodes
        data.epsilon = float(data.logMag.max()) * 1.0001
        edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
        g = torch.Generator().manual_seed(N * 7 + I)
        x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
        x[torch.rand(N, I, generator=g) < 0.02] = 0
        gy = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
        W = torch.complex(torch.randn(O, I, R, 2 * B + 1, generator=g), torch.randn(O, I, R, 2 * B + 1, generator=g)) / (I * R) ** 0.5
    else:
        edges, sten, x, gy, W = make_case(N * 3 + O, N, k, I, O, B, R, bool(N % 2))
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    xd = x.to(dev).requires_grad_(True)
    Wd = W.to(dev).requires_grad_(True)
    y = field_conv(xd, Wd, graph)
    gx, gW = torch.autograd.grad(y, [xd, Wd], grad_outputs=gy.to(dev))
    y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy())
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    assert rel_err(H(y), y_ref) < TOL
    assert rel_err(H(gx), gx_ref) < TOL
    assert rel_err(H(gW), gW_ref) < TOL


def test_geometric_and_generic_records_agree(dev, monkeypatch):
    """The forward pass takes the 32-byte geometric-phase records when the phases allow it; the generic
    factored records and the dense rows must give the same answer on the same mesh."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    from oracle.torch_composites import FCPrecomp           # the tests build their stencils on the CPU
    N, k, I, O, B, R = 900, 20, 40, 24, 3, 6
    data = sphere_support(N, k, seed=8)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    g = torch.Generator().manual_seed(2)
    x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
    x[torch.rand(N, I, generator=g) < 0.03] = 0
    W = (torch.complex(torch.randn(O, I, R, 2 * B + 1, generator=g), torch.randn(O, I, R, 2 * B + 1, generator=g)) * 0.05).to(dev)
    geo = SupportGraph(edges.to(dev), sten.to(dev), N)
    monkeypatch.setenv('FIELDCONV_NO_GEO', '1')
    gen = SupportGraph(edges.to(dev), sten.to(dev), N)
    if os.environ.get('FIELDCONV_DENSE') != '1':
        assert geo.geo_t is not None and gen.geo_t is None and gen.factored
    y_geo, y_gen = field_conv(x.to(dev), W, geo), field_conv(x.to(dev), W, gen)
    y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), H(W))
    assert rel_err(H(y_geo), y_ref) < TOL and rel_err(H(y_gen), y_ref) < TOL


def test_empty_graph_and_isolated_vertices(dev):
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    N, I, O, B, R = 40, 8, 8, 1, 3
    g = torch.Generator().manual_seed(0)
    x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g)).to(dev).requires_grad_(True)
    W = torch.complex(torch.randn(O, I, R, 3, generator=g), torch.randn(O, I, R, 3, generator=g)).to(dev)
    graph = SupportGraph(torch.zeros(0, 2, dtype=torch.long, device=dev), torch.zeros(0, R, 3, dtype=torch.cfloat, device=dev), N)
    y = field_conv(x, W, graph)
    assert torch.count_nonzero(y) == 0
    gx, = torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y))
    assert torch.count_nonzero(gx) == 0


@pytest.mark.parametrize('shape', [
    # N,  k,  I,  O, B, R
    (150, 9, 12, 20, 1, 9),        # nine rings
    (120, 8, 16, 8, 4, 6),         # band limit 4
    (90, 6, 7, 5, 4, 2),           # (band limit 0 cannot be constructed in the reference either: xavier on an empty tensor)
    (60, 5, 70, 66, 2, 12),        # wide and many rings: no channel blocks on this path
    (40, 4, 6, 6, 5, 11),
    (30, 4, 180, 9, 5, 12),        # 180 channels x 12 rings x 11 frequencies: the response of a target exceeds a CU's LDS -> channel blocks
], ids=lambda s: 'N%d_k%d_I%d_O%d_B%d_R%d' % s)
def test_any_rings_and_band_limit_run_time_path(shape, dev):
    """(n_rings, band_limit) pairs without specialised kernels -- n_rings > 8 or band_limit > 3 -- take the run-time path
    (csrc/fc_generic.hip: gather and scatter kernels on dense stencil rows, the contractions as complex GEMMs), as the
    reference takes any (nn/field_conv.py:62-98): module output, input gradient and every parameter gradient against the
    oracle, for a dense random stencil and for FCPrecomp's (through our FCPrecomp, which returns literal rows for such shapes);
    and the same through the FCResNetBlock's fused call path."""
    from fieldconv_amd import _lib
    from fieldconv_amd.nn import FCResNetBlock, FieldConv
    N, k, I, O, B, R = shape
    assert _lib.load().fc_shape_compiled(R, B) == 0
    edges, sten, x, gy, W = make_case(N + I, N, k, I, O, B, R)
    for ftype in (1, 2):
        torch.manual_seed(ftype)
        conv = FieldConv(I, O, band_limit=B, n_rings=R, ftype=ftype).to(dev)
        xd = x.to(dev).requires_grad_(True)
        y = conv(xd, edges.to(dev), sten.to(dev))
        params = list(conv.parameters())
        grads = torch.autograd.grad(y, [xd] + params, grad_outputs=gy.to(dev))
        zon, sph, ph = (H(conv.zonal), H(conv.spherical), H(conv.phase))
        Wn = orc.effective_filter(zon, sph, ph, ftype, B)
        y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), Wn)
        gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), Wn, gy.numpy())
        pg_ref = orc.effective_filter_vjp(gW_ref, zon, sph, ph, ftype, B)
        assert rel_err(H(y), y_ref) < TOL and rel_err(H(grads[0]), gx_ref) < TOL
        for g, r in zip(grads[1:], pg_ref):
            assert rel_err(H(g), r) < TOL
    # FCPrecomp for such a shape, and a block on it (modReLU / residual as separate operators on this path)
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.transforms import FCPrecomp
    from oracle.torch_composites import FCPrecomp as FCPrecompRef
    data = sphere_support(200, 10, seed=3, support='p95')
    e1, s1, _, _ = FCPrecomp(B, R, data.epsilon)(data.to(dev))
    e2, s2, _, _ = FCPrecompRef(B, R, data.epsilon)(data)
    assert torch.is_tensor(s1) and torch.equal(e1.cpu(), e2) and rel_err(H(s1), H(s2)) < 2e-6
    torch.manual_seed(5)
    blk = FCResNetBlock(6, 10, band_limit=B, n_rings=R).to(dev)
    g = torch.Generator().manual_seed(9)
    xb = torch.complex(torch.randn(200, 6, generator=g), torch.randn(200, 6, generator=g))
    yb = blk(xb.to(dev), e1, s1)
    p = {k_: H(v) for k_, v in blk.state_dict().items()}
    yb_ref = orc.fc_resnet_block_forward(xb.numpy(), e2.numpy(), s2.numpy(), p, 1, B)
    assert rel_err(H(yb), yb_ref) < 5 * TOL


def test_layers_too_large_for_the_lds_run_in_narrower_blocks(dev):
    """8 rings x 63 channels: slab, partial sums and record ring exceed the CU's LDS in split mode (fc_supported says so);
    the layer then runs as narrower channel blocks.  Found by the seeded shape sweep with FC_FUZZ_SHAPES=160."""
    import ctypes
    from fieldconv_amd import _lib
    from fieldconv_amd._lib import FcDims
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    from oracle.torch_composites import FCPrecomp
    N, k, I, O, B, R = 40, 14, 63, 42, 2, 8
    if not REDUCED and os.environ.get('FC_MFMA') != 'f32':
        assert _lib.load().fc_supported(ctypes.byref(FcDims(N, N * k, I, O, R, B))) == 0
    assert _lib.load().fc_supported(ctypes.byref(FcDims(N, N * k, 32, 32, R, B))) == 1
    data = sphere_support(N, k, seed=5)
    edges, sten, _, _ = FCPrecomp(B, R, float(data.logMag.max()) * 1.0001)(data)
    g = torch.Generator().manual_seed(8)
    x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
    gy = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
    W = torch.complex(torch.randn(O, I, R, 2 * B + 1, generator=g), torch.randn(O, I, R, 2 * B + 1, generator=g)) / (I * R) ** 0.5
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    xd, Wd = x.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True)
    y = field_conv(xd, Wd, graph)
    gx, gW = torch.autograd.grad(y, [xd, Wd], grad_outputs=gy.to(dev))
    assert rel_err(H(y), orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy())) < TOL
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    assert rel_err(H(gx), gx_ref) < TOL and rel_err(H(gW), gW_ref) < TOL


def test_wide_layers_are_split_into_channel_blocks(dev):
    """More than 64 channels: the module splits input / output channels into blocks of <= 64 (the
    operator is linear in the input channels and independent across output channels)."""
    from fieldconv_amd.nn import FieldConv
    N, k, I, O, B, R = 200, 10, 80, 70, 1, 3
    edges, sten, x, gy, _ = make_case(21, N, k, I, O, B, R, True)
    conv = FieldConv(I, O, band_limit=B, n_rings=R, ftype=1).to(dev)
    xd = x.to(dev).requires_grad_(True)
    y = conv(xd, edges.to(dev), sten.to(dev))
    params = dict(conv.named_parameters())
    grads = torch.autograd.grad(y, [xd] + list(params.values()), grad_outputs=gy.to(dev))
    z, s, p = (H(conv.zonal), H(conv.spherical), H(conv.phase))
    W = orc.effective_filter(z, s, p, 1, B)
    y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), W)
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W, gy.numpy())
    gz, gs, gp = orc.effective_filter_vjp(gW_ref, z, s, p, 1, B)
    assert rel_err(H(y), y_ref) < TOL
    assert rel_err(H(grads[0]), gx_ref) < TOL
    assert rel_err(H(grads[1]), gz) < TOL and rel_err(H(grads[2]), gs) < TOL and rel_err(H(grads[3]), gp) < TOL


def test_wide_layers_natively_explicit_filter_three_blocks_and_records(dev):
    """The native channel-block path (csrc/fc_wide.hip) beyond the module case above: an explicit filter (field_conv) with THREE
    input blocks and two output blocks on dense stencil rows; the module on an FCPrecomp stencil (record-driven kernels inside
    the blocks, 130 -> 70 channels); a wide FCResNetBlock (wide TangentLin through fc_cgemm, 100 -> 72 channels) against the
    oracle's block; and the source contains no Python block loop any more."""
    import inspect
    from fieldconv_amd import functional as Fn
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    from fieldconv_amd.nn import FCResNetBlock, FieldConv
    src = inspect.getsource(Fn.field_conv_params) + inspect.getsource(Fn.field_conv) + inspect.getsource(Fn._GenericFieldConvFn)
    assert 'torch.cat' not in src and 'matmul' not in src and ' @ ' not in src
    # (1) explicit filter, dense rows, 150 -> 70 channels
    N, k, I, O, B, R = 150, 8, 150, 70, 1, 3
    edges, sten, x, gy, W = make_case(23, N, k, I, O, B, R, True)
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    xd, Wd = x.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True)
    y = field_conv(xd, Wd, graph)
    gx, gW = torch.autograd.grad(y, [xd, Wd], grad_outputs=gy.to(dev))
    y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy())
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    assert rel_err(H(y), y_ref) < TOL and rel_err(H(gx), gx_ref) < TOL and rel_err(H(gW), gW_ref) < TOL
    # (2) module on an FCPrecomp stencil (records), 130 -> 70 channels, band limit 2
    from fieldconv_amd.data import sphere_support
    from oracle.torch_composites import FCPrecomp
    N, k, I, O, B, R = 700, 10, 130, 70, 2, 6
    data = sphere_support(N, k, seed=4)
    e2, s2, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    g = torch.Generator().manual_seed(8)
    x2 = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
    gy2 = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
    conv = FieldConv(I, O, band_limit=B, n_rings=R, ftype=1).to(dev)
    x2d = x2.to(dev).requires_grad_(True)
    y2 = conv(x2d, e2.to(dev), s2.to(dev))
    grads = torch.autograd.grad(y2, [x2d] + list(conv.parameters()), grad_outputs=gy2.to(dev))
    z, sp, ph = H(conv.zonal), H(conv.spherical), H(conv.phase)
    W2 = orc.effective_filter(z, sp, ph, 1, B)
    y2_ref, gx2_ref, gW2_ref = orc.fieldconv_forward_backward(x2.numpy(), e2.numpy(), s2.numpy(), W2, gy2.numpy())
    gz, gs, gp = orc.effective_filter_vjp(gW2_ref, z, sp, ph, 1, B)
    assert rel_err(H(y2), y2_ref) < TOL and rel_err(H(grads[0]), gx2_ref) < TOL
    assert rel_err(H(grads[1]), gz) < TOL and rel_err(H(grads[2]), gs) < TOL and rel_err(H(grads[3]), gp) < TOL
    # (3) a wide block: output against the oracle's block
    blk = FCResNetBlock(100, 72, band_limit=1, n_rings=3, ftype=1).to(dev)
    N, k = 150, 8
    e3, s3, x3, _, _ = make_case(29, N, k, 100, 72, 1, 3, True)
    y3 = blk(x3.to(dev), e3.to(dev), s3.to(dev))
    p3 = {n_: H(v) for n_, v in blk.state_dict().items()}
    y3_ref = orc.fc_resnet_block_forward(x3.numpy(), e3.numpy(), s3.numpy(), p3, 1, 1)
    assert rel_err(H(y3), y3_ref) < 2 * TOL
    torch.autograd.grad(y3.abs().sum(), list(blk.parameters()))          # the backward pass of every piece runs


# ---------------------------------------------------------------- full benchmark size: properties
def test_full_size_properties(dev):
    """Config 2 of BASELINE.json (20k vertices, k=32, C=48, B=2, R=6): the oracle needs ~32 GB at
    this size, so check (a) a random subset of output rows / input-gradient rows against the oracle
    on the sub-edge-lists that determine them, (b) linearity in the filter, (c) the adjoint identity
    Re<gy, conv(x;V)> = Re<gW, V>, (d) bitwise run-to-run reproducibility."""
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    N, k, I, O, B, R = 20000, 32, 48, 48, 2, 6
    edges, sten, x, gy, W = make_case(11, N, k, I, O, B, R, True)
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    xd = x.to(dev).requires_grad_(True)
    Wd = W.to(dev).requires_grad_(True)
    y = field_conv(xd, Wd, graph)
    gx, gW = torch.autograd.grad(y, [xd, Wd], grad_outputs=gy.to(dev))

    g = torch.Generator().manual_seed(5)
    sub = torch.randperm(N, generator=g)[:150]
    # (a1) output rows depend only on the in-edges of those rows
    mask = torch.isin(edges[:, 1], sub)
    y_ref = orc.fieldconv_forward(x.numpy(), edges[mask].numpy(), sten[mask].numpy(), W.numpy())
    assert rel_err(H(y)[sub.numpy()], y_ref[sub.numpy()]) < TOL
    # (a2) input-gradient rows depend only on the out-edges of those rows
    mask = torch.isin(edges[:, 0], sub)
    gx_ref, _ = orc.fieldconv_backward(x.numpy(), edges[mask].numpy(), sten[mask].numpy(), W.numpy(), gy.numpy())
    assert rel_err(H(gx)[sub.numpy()], gx_ref[sub.numpy()]) < TOL

    # (b) linearity in the filter
    V = torch.complex(torch.randn(W.shape, generator=g), torch.randn(W.shape, generator=g)).to(dev) * 0.1
    with torch.no_grad():
        yv = field_conv(xd, V, graph)
        ysum = field_conv(xd, Wd + V, graph)
    assert rel_err(H(ysum), H(y + yv)) < TOL
    # (c) adjoint identity for the filter gradient
    lhs = torch.sum(torch.conj(gy.to(dev)) * yv).real.item()
    rhs = torch.sum(torch.conj(gW) * V).real.item()
    assert abs(lhs - rhs) <= max(2e-4, TOL) * max(abs(lhs), abs(rhs), 1.0)
    # (d) deterministic: no atomics anywhere on the path
    y2 = field_conv(xd, Wd, graph)
    gx2, gW2 = torch.autograd.grad(y2, [xd, Wd], grad_outputs=gy.to(dev))
    assert torch.equal(torch.view_as_real(y2), torch.view_as_real(y))
    assert torch.equal(torch.view_as_real(gx2), torch.view_as_real(gx))
    assert torch.equal(torch.view_as_real(gW2), torch.view_as_real(gW))


def test_step_graph_replays_a_training_step(dev):
    """A block's forward + loss + backward captured as one HIP graph: the replay is bit-identical to the eager step,
    and follows in-place updates of the inputs and the parameters (static addresses, fresh values)."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock
    from oracle.torch_composites import FCPrecomp           # the tests build their stencils on the CPU
    from fieldconv_amd.utils import StepGraph
    N, k, C, B, R = 300, 24, 16, 2, 6
    data = sphere_support(N, k).to(dev)
    edges, sten, ln, wxp = FCPrecomp(B, R, data.epsilon)(data)
    g = torch.Generator().manual_seed(5)
    torch.manual_seed(5)
    pos = torch.randn(N, 3, generator=g).to(dev)
    labels = torch.randint(0, 4, (N,), generator=g).to(dev)
    mods = torch.nn.ModuleDict(dict(lift=LiftBlock(3, C, n_rings=R, ftype=1), res=FCResNetBlock(C, C, band_limit=B, n_rings=R),
                                    echo=ECHOBlock(C, 4, n_des=C, n_bins=2, band_limit=B, n_rings=R))).to(dev)
    params = list(mods.parameters())

    def step():
        x = mods['lift'](pos, edges, sten[..., B:B + 2])
        x = mods['res'](x, edges, sten)
        logits = mods['echo'](x, edges, sten, ln, wxp)
        loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), labels)
        return (loss.detach(),) + torch.autograd.grad(loss, params)

    eager = [t.clone() for t in step()]
    graphed = StepGraph(step)
    assert all(torch.equal(a, b) for a, b in zip(eager, graphed.replay()))
    with torch.no_grad():                       # an optimizer step and a new input, in place
        for p, gp in zip(params, eager[1:]):
            p.sub_(0.05 * gp)
        pos.mul_(1.1)
    replayed = [t.clone() for t in graphed.replay()]
    fresh = step()
    assert all(torch.equal(a, b) for a, b in zip(fresh, replayed))
    assert not torch.equal(replayed[0], eager[0])


@pytest.mark.parametrize('N,k', [(200, 96), (1024, 128), (48, 40), (33, 32)])
def test_edge_split_on_small_meshes(dev, monkeypatch, N, k):
    """Meshes whose N/16 tiles cannot fill the chip: several workgroups share a tile, each with a share of every
    vertex's edges (ragged degrees here: a random third of the edges is dropped).  Against the oracle, and against the
    unsplit kernels on the same inputs."""
    from fieldconv_amd import _lib
    from fieldconv_amd._lib import FcDims
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    from oracle.torch_composites import FCPrecomp           # the tests build their stencils on the CPU
    import ctypes
    I, O, B, R = 24, 20, 2, 6
    data = sphere_support(N, k, seed=N)
    edges, sten, _, _ = FCPrecomp(B, R, float(data.logMag.max()) * 1.0001)(data)
    g = torch.Generator().manual_seed(N + k)
    keep = torch.rand(edges.shape[0], generator=g) > 0.33
    edges, sten = edges[keep].contiguous(), sten[keep].contiguous()
    x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
    gy = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
    W = torch.complex(torch.randn(O, I, R, 2 * B + 1, generator=g), torch.randn(O, I, R, 2 * B + 1, generator=g)) / (I * R) ** 0.5
    dims = FcDims(N, int(edges.shape[0]), I, O, R, B)
    assert _lib.load().fc_forward_workspace_bytes(ctypes.byref(dims)) > 0            # the split applies to this shape
    y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy())
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    out = {}
    for split in ('1', '0'):
        for no_geo in ('0', '1'):
            monkeypatch.setenv('FIELDCONV_NO_EDGE_SPLIT', '0' if split == '1' else '1')
            monkeypatch.setenv('FIELDCONV_NO_GEO', no_geo)
            graph = SupportGraph(edges.to(dev), sten.to(dev), N)
            assert graph.factored
            xd = x.to(dev).requires_grad_(True)
            Wd = W.to(dev).requires_grad_(True)
            y = field_conv(xd, Wd, graph)
            gx, gW = torch.autograd.grad(y, [xd, Wd], grad_outputs=gy.to(dev))
            assert rel_err(H(y), y_ref) < TOL
            assert rel_err(H(gx), gx_ref) < TOL
            assert rel_err(H(gW), gW_ref) < TOL
            out[split, no_geo] = H(y)
    assert rel_err(out['1', '0'], out['0', '0']) < TOL and rel_err(out['1', '1'], out['0', '1']) < TOL


@pytest.mark.parametrize('N,k,B,R', [(300, 20, 2, 6), (1024, 128, 2, 6), (77, 9, 1, 3), (150, 12, 3, 8)])
def test_native_graph_build_matches_torch_build(dev, N, k, B, R):
    """fc_graph_build (csrc/fc_graph.hip) against the torch build of the same SupportGraph: identical grouping, slot
    order, permutations and ring-run offsets; records equal up to the rounding of the two weight quotients; a dense
    (non-factorable) stencil is detected and takes the dense kernels."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.graph import EdgeCSR, SupportGraph
    from oracle.torch_composites import FCPrecomp           # the tests build their stencils on the CPU
    data = sphere_support(N, k, seed=N)
    edges, sten, _, _ = FCPrecomp(B, R, float(data.logMag.max()) * 1.0001)(data)
    g = torch.Generator().manual_seed(N)
    order = torch.randperm(edges.shape[0], generator=g)             # no particular input order
    edges, sten = edges[order].contiguous().to(dev), sten[order].contiguous().to(dev)
    a = SupportGraph(edges, sten, N, native=True)
    b = SupportGraph(edges, sten, N, native=False)
    assert a.factored and b.factored and (a.geo_t is not None) and (b.geo_t is not None)
    for name in ('rowptr_t', 'nbr_t', 'runs_t', 'perm_t', 'rowptr_s', 'nbr_s', 'runs_s', 'perm_s'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    for name in ('rec_t', 'rec_s', 'geo_t'):
        ra, rb = getattr(a, name), getattr(b, name)
        assert ra.shape == rb.shape, name
        assert torch.equal(ra[:, 0].view(torch.int32), rb[:, 0].view(torch.int32)) and torch.equal(ra[:, 3].view(torch.int32), rb[:, 3].view(torch.int32))
        keep = [c for c in range(ra.shape[1]) if c not in (0, 3)]
        assert rel_err(H(ra[:, keep]), H(rb[:, keep])) < 1e-6, name
    e = EdgeCSR(edges, N)
    for name in ('rowptr_t', 'rowptr_s'):
        assert torch.equal(getattr(e, name), getattr(a, name))
    assert torch.equal(torch.sort(e.nbr_t.view(-1))[0], torch.sort(a.nbr_t)[0])
    assert torch.equal(edges[e.perm_t, 0].to(torch.int32), e.nbr_t) and torch.equal(edges[e.perm_s, 1].to(torch.int32), e.nbr_s)
    # the reference hands over a transposed view of a (2,E) tensor (transforms/support_graph.py:59), and int32 works too
    view = edges.t().contiguous().t()
    assert not view.is_contiguous()
    v = SupportGraph(view, sten, N, native=True)
    w32 = SupportGraph(edges.to(torch.int32), sten, N, native=True)
    for name in ('rowptr_t', 'nbr_t', 'perm_t', 'rowptr_s', 'nbr_s', 'perm_s'):
        assert torch.equal(getattr(v, name), getattr(a, name)) and torch.equal(getattr(w32, name), getattr(a, name)), name
    # a random dense stencil is not factorable: both builds must say so
    dense = torch.complex(torch.randn(sten.shape, generator=g), torch.randn(sten.shape, generator=g)).to(dev)
    c = SupportGraph(edges, dense, N, native=True)
    assert not c.factored and torch.equal(c.sten_t, dense[c.perm_t]) and torch.equal(c.sten_s, dense[c.perm_s])
    # rank-1 but not geometric phases: factored records without the geometric form
    twisted = sten.clone()
    twisted[:, :, 0] *= 1.5
    d = SupportGraph(edges, twisted, N, native=True)
    assert d.factored and d.geo_t is None
    with pytest.raises(IndexError):
        bad = edges.clone()
        bad[3, 0] = N + 5
        SupportGraph(bad, sten, N, native=True)


def test_native_graph_build_hub_vertex(dev):
    """A hub: every vertex sends an edge to vertex 7 and receives one from it, all at the same radius -- two (vertex, ring)
    runs of 3 000 edges, far beyond what one thread orders (csrc/fc_graph.hip: graph_order_long_kernel).  Same slot order as
    the torch build's stable sorts, and the convolution over it matches the oracle."""
    from fieldconv_amd.data.synthetic import SupportData
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    from oracle.torch_composites import FCPrecomp
    N, B, R, C = 3000, 2, 6, 8
    g = torch.Generator().manual_seed(5)
    hub = torch.full((N,), 7)
    ring = torch.arange(N)
    edges = torch.cat((torch.stack((ring, hub), 1), torch.stack((hub, ring), 1)))
    edges = edges[torch.randperm(2 * N, generator=g)]
    E = edges.shape[0]
    data = SupportData(supp_edges=edges, logMag=torch.full((E,), 0.55), logAng=(torch.rand(E, generator=g) * 2 - 1) * 3.14159,
                       xp=torch.polar(torch.ones(E), (torch.rand(E, generator=g) * 2 - 1) * 3.14159), w=torch.ones(N, 1) / N,
                       epsilon=1.0, num_nodes=N)
    e2, sten, _, _ = FCPrecomp(B, R, 1.0)(data)
    assert e2.shape[0] == E
    a = SupportGraph(e2.to(dev), sten.to(dev), N, native=True)
    b = SupportGraph(e2.to(dev), sten.to(dev), N, native=False)
    runs7 = a.runs_t[7].tolist()
    assert a.factored and int(a.rowptr_t[8] - a.rowptr_t[7]) == N + 1 and max(b_ - a_ for a_, b_ in zip(runs7, runs7[1:])) == N + 1
    for name in ('rowptr_t', 'nbr_t', 'runs_t', 'perm_t', 'rowptr_s', 'nbr_s', 'runs_s', 'perm_s'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g))
    W = torch.complex(torch.randn(C, C, R, 2 * B + 1, generator=g), torch.randn(C, C, R, 2 * B + 1, generator=g))
    y = field_conv(x.to(dev), W.to(dev), a)
    y_ref = orc.fieldconv_forward(x.numpy(), e2.numpy(), sten.numpy(), W.numpy())
    assert rel_err(H(y), y_ref) < TOL


@pytest.mark.parametrize('N,k,C,n_bins', [(100, 80, 48, 3), (90, 40, 17, 2), (300, 9, 64, 1), (70, 66, 5, 4), (60, 10, 130, 2),
                                          (80, 30, 48, 6), (50, 24, 30, 8), (64, 20, 64, 5),    # n_bins > 4: channel blocks by LDS
                                          (40, 12, 9, 10)])                                     # n_bins > 8: the run-time kernels
def test_echo_kernels_vs_host_composite(dev, N, k, C, n_bins):
    """ECHO descriptor kernels against the oracle's torch restatement run on the CPU in float64 (pinned to the reference
    fixtures by the CPU suite), on supports wide enough that 2 or 4 wavefronts share a vertex, with ragged degrees and
    zero features."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import ECHO
    from oracle.torch_composites import echo_descriptors as echo_ref
    from oracle.torch_composites import FCPrecomp           # the tests build their stencils on the CPU
    data = sphere_support(N, k, seed=N)
    edges, _, ln, wxp = FCPrecomp(1, 3, float(data.logMag.max()) * 1.0001)(data)
    g = torch.Generator().manual_seed(N + C)
    keep = torch.rand(edges.shape[0], generator=g) > 0.2
    edges, ln, wxp = edges[keep].contiguous(), ln[keep].contiguous(), wxp[keep].contiguous()
    # self edges: radius exactly 0 as in real data (the synthetic geodesic distance leaves 1e-9; a point a hair off the
    # raster's centre votes with full weight into a sign-dependent cell in the reference's formula)
    ln = torch.where(ln.abs() < 1e-6, torch.zeros_like(ln), ln)
    x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g))
    x[torch.rand(N, C, generator=g) < 0.05] = 0
    m = ECHO(C, n_bins)
    xr = x.to(torch.complex128).requires_grad_(True)
    dr = echo_ref(xr, edges, ln.to(torch.complex128) * 0.999, wxp.to(torch.complex128), n_bins)
    gd = torch.randn(dr.shape, generator=g)
    gr, = torch.autograd.grad(dr, [xr], grad_outputs=gd.double())
    xd = x.to(dev).requires_grad_(True)
    dd = m.to(dev)(xd, edges.to(dev), (ln * 0.999).to(dev), wxp.to(dev))
    gg, = torch.autograd.grad(dd, [xd], grad_outputs=gd.to(dev))
    # votes are piecewise linear in the rotated point (an fp32 floor/ceil flip moves a vanishing vote to the neighbouring
    # cell); ln is scaled by 0.999 so that no point sits exactly on the raster's rim
    # The reference's votes vanish at exactly integer raster coordinates (ceil == floor, nn/echo.py:30-61): a coordinate that
    # rounds to an integer in fp32 but sits 1e-7 beside it in float64 drops a whole vote (about once per 1e6 (edge, channel)
    # pairs, tools/fuzz/fuzz_components.py).  Entries fed by such a vote are not comparable and are left out.
    frame = torch.conj(torch.polar(torch.ones(N, C, dtype=torch.float64), torch.angle(x.to(torch.complex128))))
    qq = torch.view_as_real((ln * 0.999).to(torch.complex128)[:, None] * frame[edges[:, 0]] * n_bins)
    near = ((qq - torch.round(qq)).abs() < 1e-5).any(dim=2) & (ln.abs() > 0)[:, None]
    cols = torch.arange(C)[None, :].expand(edges.shape[0], -1)
    fragile = torch.zeros(N, C, dtype=torch.bool)
    fragile[edges[:, 1][:, None].expand(-1, C)[near], cols[near]] = True
    tainted = torch.zeros(N, dtype=torch.bool)
    tainted[edges[:, 0][fragile.any(dim=1)[edges[:, 1]]]] = True
    ok, okg = (~fragile)[..., None].numpy(), (~tainted)[:, None].numpy()
    assert float(fragile.float().mean()) < 1e-2
    assert rel_err(H(dd) * ok, dr.detach().numpy() * ok) < 5e-6             # measured 1-3e-7
    assert rel_err(H(gg) * okg, gr.numpy() * okg) < 5e-5                    # measured 1-3e-6


@pytest.mark.parametrize('N,k,B,R,shrink', [(300, 20, 2, 6, 1.0), (1024, 128, 2, 6, 0.8), (77, 9, 1, 3, 0.5), (150, 12, 3, 8, 1.0)])
def test_native_fc_precomp_matches_torch(dev, N, k, B, R, shrink):
    """FCPrecomp (fc_precomp_mark / fc_precomp_build, csrc/fc_precomp.hip) against the oracle's torch restatement (pinned
    to the reference fixtures by the CPU suite) on the same inputs: the same edges are kept, in the same order, and
    stencil, ln and wxp agree to fp32 rounding (the area sums are float atomics on the device)."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.transforms import FCPrecomp
    from oracle.torch_composites import fc_precomp
    data = sphere_support(N, k, seed=N)
    eps = float(data.logMag.max()) * shrink                               # shrink < 1: part of the edges falls outside the radius
    e1, s1, l1, w1 = (t.cpu() for t in FCPrecomp(B, R, eps)(data.to(dev)))
    e2, s2, l2, w2 = fc_precomp(data.logMag, data.logAng, data.w, data.supp_edges, data.xp, B, R, eps)
    assert e1.shape[0] > 0 and (shrink == 1.0 or e1.shape[0] < data.supp_edges.shape[0])
    assert torch.equal(e1, e2) and e1.dtype == e2.dtype
    assert s1.shape == s2.shape and s1.dtype == s2.dtype
    assert rel_err(H(s1), H(s2)) < 2e-6 and rel_err(H(l1), H(l2)) < 2e-6 and rel_err(H(w1), H(w2)) < 2e-6
    # the same two rings carry the weight (torch divides by epsilon through its reciprocal, the kernel divides: an edge
    # within one ulp of a knot may get a 1e-7 weight on the neighbouring ring in one of the two)
    tiny = 1e-5 * float(s2.abs().max())
    assert torch.equal(s1.abs() > tiny, s2.abs() > tiny)


@pytest.mark.parametrize('set_to_none', [False, True])
def test_fused_adam_matches_torch_adam(dev, set_to_none):
    """FusedAdam (one flat buffer, fc_adam_step) against torch.optim.Adam on the same block over several steps, with
    weight decay; also as part of a captured HIP graph (device-side step counter).  set_to_none: gradients assigned by
    autograd and packed into the flat buffer by step() instead of accumulated into zeroed views."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import FCResNetBlock
    from fieldconv_amd.optim import FusedAdam
    from oracle.torch_composites import FCPrecomp           # the tests build their stencils on the CPU
    from fieldconv_amd.utils import StepGraph
    import copy
    N, k, C, B, R = 200, 12, 8, 1, 4
    data = sphere_support(N, k).to(dev)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    g = torch.Generator().manual_seed(3)
    torch.manual_seed(3)
    x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev)
    ma = FCResNetBlock(C, C, band_limit=B, n_rings=R).to(dev)
    mb = copy.deepcopy(ma)
    mc = copy.deepcopy(ma)
    kw = dict(lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
    oa = torch.optim.Adam(ma.parameters(), **kw)
    ob = FusedAdam(mb.parameters(), **kw)
    oc = FusedAdam(mc.parameters(), **kw)

    def train_step(m, o):
        o.zero_grad(set_to_none=set_to_none)
        loss = m(x, edges, sten).abs().square().mean()
        loss.backward()
        o.step()
        return loss.detach()
    graphed = StepGraph(lambda: train_step(mc, oc), warmup=2)         # 2 warm-up steps are taken; the capture only records
    for _ in range(2):
        la = train_step(ma, oa)
        lb = train_step(mb, ob)
    for _ in range(4):
        la = train_step(ma, oa)
        lb = train_step(mb, ob)
        lc = graphed.replay().clone()
    assert abs(float(la) - float(lb)) < 1e-5 * abs(float(la)) and abs(float(la) - float(lc)) < 1e-5 * abs(float(la))
    for (name, pa), pb, pc in zip(ma.named_parameters(), mb.parameters(), mc.parameters()):
        assert rel_err(H(pb), H(pa)) < 2e-5, name
        assert rel_err(H(pc), H(pa)) < 2e-5, name


def test_fused_adam_checkpoint_round_trip(dev):
    """FusedAdam.state_dict() carries the moments and the step in torch.optim.Adam's layout: a run that is saved,
    restored into a FRESH FusedAdam (and into a torch.optim.Adam) and continued matches the uninterrupted run; a later
    add_param_group is refused (the flat buffers are laid out at construction)."""
    import copy
    from fieldconv_amd.nn import TangentPerceptron
    from fieldconv_amd.optim import FusedAdam
    torch.manual_seed(4)
    g = torch.Generator().manual_seed(4)
    x = torch.complex(torch.randn(300, 12, generator=g), torch.randn(300, 12, generator=g)).to(dev)
    ma = TangentPerceptron(12, 9).to(dev)
    mb, mc = copy.deepcopy(ma), copy.deepcopy(ma)
    kw = dict(lr=2e-3, betas=(0.8, 0.95), eps=1e-8, weight_decay=1e-3)

    def train(m, o, n):
        for _ in range(n):
            o.zero_grad()
            m(x).abs().square().mean().backward()
            o.step()
    oa = FusedAdam(ma.parameters(), **kw)
    train(ma, oa, 7)                                           # the uninterrupted run
    ob = FusedAdam(mb.parameters(), **kw)
    train(mb, ob, 3)
    sd = copy.deepcopy(ob.state_dict())
    assert float(sd['state'][0]['step']) == 3 and float(sd['state'][0]['exp_avg'].abs().max()) > 0
    weights = copy.deepcopy(mb.state_dict())
    mb2 = TangentPerceptron(12, 9).to(dev)
    mb2.load_state_dict(weights)
    ob2 = FusedAdam(mb2.parameters(), lr=1.0)                  # hyper-parameters come back with the checkpoint
    ob2.load_state_dict(sd)
    train(mb2, ob2, 4)
    mc.load_state_dict(weights)
    oc = torch.optim.Adam(mc.parameters(), **kw)
    oc.load_state_dict(sd)                                     # the same checkpoint resumes under torch's Adam
    train(mc, oc, 4)
    for (name, pa), pb, pc in zip(ma.named_parameters(), mb2.parameters(), mc.parameters()):
        assert rel_err(H(pb), H(pa)) < 1e-6, name
        assert rel_err(H(pc), H(pa)) < 2e-5, name
    with pytest.raises(RuntimeError):
        ob2.add_param_group({'params': [torch.nn.Parameter(torch.zeros(4, device=dev))]})


@pytest.mark.parametrize('N,k', [(700, 14), (9000, 8), (200, 96)])
def test_conv_epilogue_fusion_matches_separate_operators(dev, monkeypatch, N, k):
    """FCResNetBlock runs the residual add and both modReLUs in the epilogues of its convolution kernels
    (fc_epilogue); with FIELDCONV_NO_FUSED_EPILOGUE=1 it composes the separate operators as the reference does
    (nn/fc_resnet_block.py:84-88).  Same arithmetic: outputs and every gradient agree bit for bit -- on the frequency-major
    kernels, the ring-major ones (9000 vertices) and through the edge split of small meshes (200 vertices, 96 neighbours)."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import FCResNetBlock
    from fieldconv_amd.transforms import FCPrecomp
    C, B, R = 16, 2, 6
    data = sphere_support(N, k, seed=N).to(dev)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    torch.manual_seed(N)
    blk = FCResNetBlock(C, C, band_limit=B, n_rings=R).to(dev)
    with torch.no_grad():
        blk.nonlin1.bias.uniform_(-0.4, 0.1)
        blk.nonlin2.bias.uniform_(-0.4, 0.1)
    g = torch.Generator().manual_seed(N)
    x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g))
    x[torch.rand(N, C, generator=g) < 0.02] = 0
    gy = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev)
    params = list(blk.parameters())
    out = {}
    for fused in (True, False):
        monkeypatch.setenv('FIELDCONV_NO_FUSED_EPILOGUE', '0' if fused else '1')
        xd = x.to(dev).requires_grad_(True)
        y = blk(xd, edges, sten)
        out[fused] = (y.detach(),) + torch.autograd.grad(y, [xd] + params, grad_outputs=gy)
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('N,k,cin,cout,frontload,B', [(700, 14, 16, 24, False, 2), (9000, 8, 24, 16, True, 2), (200, 96, 48, 48, False, 2),
                                                      (1024, 128, 48, 48, False, 2), (500, 20, 64, 64, False, 3), (300, 12, 8, 8, True, 1)])
def test_block_level_entry_points_match_the_per_operator_path(dev, monkeypatch, N, k, cin, cout, frontload, B):
    """FCResNetBlock / ECHOBlock / LiftBlock through ONE native call per pass (csrc/fc_blocks.hip: fc_resnet_block_*, fc_echo_block_*,
    fc_lift_block_*; reference nn/fc_resnet_block.py:84-88, nn/echo_block.py:93-103, nn/lift_block.py:53-55) against the same modules
    composed of per-operator autograd nodes (FIELDCONV_BLOCK_CALLS=0): the same kernels in the same order, so outputs and EVERY gradient
    agree bit for bit -- frequency-major and ring-major (9000 vertices) forward kernels, the edge split of small meshes (200 / 1024
    vertices with wide supports), band limits 1-3, both `frontload` settings."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock
    from fieldconv_amd.transforms import FCPrecomp
    R = 6
    data = sphere_support(N, k, seed=N).to(dev)
    edges, sten, ln, wxp = FCPrecomp(B, R, data.epsilon)(data)
    torch.manual_seed(N)
    n_des = min(cin, 12 if B == 3 else cin)
    mods = {'resnet': FCResNetBlock(cin, cout, band_limit=B, n_rings=R, frontload=frontload).to(dev),
            'echo': ECHOBlock(cin, 7, n_des=n_des, n_bins=2 if B == 3 else 3, band_limit=B, n_rings=R).to(dev),
            'lift': LiftBlock(3, cout, n_rings=R, ftype=1).to(dev)}
    with torch.no_grad():
        mods['resnet'].nonlin1.bias.uniform_(-0.4, 0.1)
        mods['resnet'].nonlin2.bias.uniform_(-0.4, 0.1)
        mods['echo'].nonlin.bias.uniform_(-0.4, 0.1)
        mods['lift'].nonlin.bias.uniform_(-0.4, 0.1)
    g = torch.Generator().manual_seed(N)
    x = torch.complex(torch.randn(N, cin, generator=g), torch.randn(N, cin, generator=g))
    x[torch.rand(N, cin, generator=g) < 0.02] = 0
    pos = torch.randn(N, 3, generator=g)
    inputs = {'resnet': lambda: (x.to(dev).requires_grad_(True), edges, sten),
              'echo': lambda: (x.to(dev).requires_grad_(True), edges, sten, ln, wxp),
              'lift': lambda: (pos.to(dev).requires_grad_(True), edges, sten[..., B:B + 2])}
    node = {'resnet': 'ResnetBlockFn', 'echo': 'EchoBlockFn', 'lift': 'LiftBlockFn'}
    from fieldconv_amd.blocks import cpp_nodes
    assert cpp_nodes() is not None, 'fc_torch_nodes.so (the C++ autograd nodes) is not built'
    for name, mod in mods.items():
        params = list(mod.parameters())
        out = {}
        # the block-level node in C++ (fc_torch_nodes.so), the same node in Python (fieldconv_amd/blocks.py), per-operator composition
        for native in ('cpp', 'python', False):
            monkeypatch.setenv('FIELDCONV_BLOCK_CALLS', '1' if native else '0')
            monkeypatch.setenv('FIELDCONV_CPP_NODES', '1' if native == 'cpp' else '0')
            # per-operator leg of the ECHOBlock: its dense tail through torch's own Linear / ReLU nodes (FIELDCONV_ECHO_TAIL=0), so that
            # the comparison below really is native head against the dense layers
            if not native and name == 'echo':
                monkeypatch.setenv('FIELDCONV_ECHO_TAIL', '0')
            else:
                monkeypatch.delenv('FIELDCONV_ECHO_TAIL', raising=False)
            args = inputs[name]()
            y = mod(*args)
            seen, todo, found = set(), [y.grad_fn], None
            while todo:                         # the block-level node is (not) in the autograd graph
                fn = todo.pop()
                if fn is None or fn in seen:
                    continue
                seen.add(fn)
                if node[name] in fn.name():
                    found = 'cpp' if 'CppNode' in fn.name() else 'python'
                todo += [nf for nf, _ in fn.next_functions]
            assert found == (native or None), (name, native, found)
            gen = torch.Generator().manual_seed(1)
            gy = torch.randn(y.shape, generator=gen)
            if y.is_complex():
                gy = torch.complex(gy, torch.randn(y.shape, generator=gen))
            out[native] = (y.detach(),) + torch.autograd.grad(y, [args[0]] + params, grad_outputs=gy.to(dev), allow_unused=True)
        for a, b in zip(out['cpp'], out['python']):                 # the two bindings of the same node: the same calls, the same bits
            assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), name
        for a, b in zip(out['cpp'], out[False]):
            assert (a is None) == (b is None)
            if a is None:
                continue
            if name == 'echo':
                # ECHOBlock's dense tail (three Linear layers + the residual on |x|) runs through the library's own kernels on the block-level
                # path (fc_echo_head_*: fp32 matrix-pipe products, sums in another order) and -- on this leg, FIELDCONV_ECHO_TAIL=0 -- through
                # torch's Linear / ReLU nodes: the same arithmetic width, not the same bits (test_echo_head_matches_the_dense_layers_in_double_precision pins it to fp64)
                assert rel_err(H(a), H(b)) < 1e-5, name
            else:
                assert torch.equal(a, b), name
    # ... while its native half -- convolution + modReLU + descriptor splat -- is bit-identical to the per-operator composition
    from fieldconv_amd.blocks import echo_block_descriptors
    from fieldconv_amd.graph import get_graph
    em = mods['echo']
    monkeypatch.setenv('FIELDCONV_BLOCK_CALLS', '1')
    monkeypatch.setenv('FIELDCONV_CPP_NODES', '1')
    xa = x.to(dev).requires_grad_(True)
    da = echo_block_descriptors(em, xa, get_graph(edges, sten, N), ln, wxp)
    xb = x.to(dev).requires_grad_(True)
    db = em.echo(em.conv.forward_act(xb, edges, sten, em.nonlin.bias[:, : em.n_des]), edges, ln, wxp)
    gd = torch.randn(da.shape, generator=torch.Generator().manual_seed(2)).to(dev)
    conv_params = [em.conv.zonal, em.conv.spherical, em.conv.phase, em.nonlin.bias]
    ga = torch.autograd.grad(da, [xa] + conv_params, grad_outputs=gd)
    gb = torch.autograd.grad(db, [xb] + conv_params, grad_outputs=gd)
    assert torch.equal(da, db) and all(torch.equal(p_, q_) for p_, q_ in zip(ga, gb))


def test_soft_abs_kernel_matches_the_torch_formulation(dev):
    """softAbs (reference utils/field.py:29-37; ECHOBlock's residual branch) as one kernel per pass against the package's branch-free torch
    formulation of it (pinned to the reference by the ECHOBlock fixture): origin-box entries, box-edge entries (strict <), gradients."""
    from fieldconv_amd.functional import soft_abs
    from fieldconv_amd.utils import softAbs
    g = torch.Generator().manual_seed(3)
    x = torch.complex(torch.randn(777, 13, generator=g), torch.randn(777, 13, generator=g))
    x[::5, 0] = 0
    x[1, 1] = complex(3e-8, -5e-8)
    x[2, 2] = complex(9.9e-8, 2e-7)
    x[3, 3] = complex(-1e-7, 0.0)
    gy = torch.randn(777, 13, generator=g).to(dev)
    xa = x.to(dev).requires_grad_(True)
    xb = x.to(dev).requires_grad_(True)
    ya, yb = soft_abs(xa), softAbs(xb)
    assert type(ya.grad_fn).__name__ == '_SoftAbsFnBackward'
    assert rel_err(H(ya), H(yb)) < 1e-6 and bool((ya[::5, 0] == 0).all())
    ga, = torch.autograd.grad(ya, [xa], grad_outputs=gy)
    gb, = torch.autograd.grad(yb, [xb], grad_outputs=gy)
    assert rel_err(H(ga), H(gb)) < 1e-6 and bool((ga[::5, 0] == 0).all())


@pytest.mark.parametrize('N,k,B,R', [(300, 20, 2, 6), (1024, 128, 2, 6), (90, 9, 1, 3), (150, 12, 3, 8)])
def test_lift_block_reads_the_factor_table(dev, N, k, B, R):
    """`supp_sten[..., B:B+2]` of FCPrecomp's stencil stand-in (what the notebooks hand to LiftBlock, reference segmentation.ipynb:204) is a
    stand-in too: the TransField kernels form the two columns w_r c, w_r c e^{i theta} from the (E,8) factor table (sten_stride 0) and
    no (E,R,2) array is built.  Same output and gradients as with the materialised columns; nothing gets materialised."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.graph import LiftColumns
    from fieldconv_amd.nn import LiftBlock
    from fieldconv_amd.transforms import FCPrecomp
    data = sphere_support(N, k, seed=N).to(dev)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    lift_sten = sten[..., B:B + 2]
    assert isinstance(lift_sten, LiftColumns) and tuple(lift_sten.shape) == (edges.shape[0], R, 2) and lift_sten.dtype == torch.complex64
    torch.manual_seed(N)
    mod = LiftBlock(3, 24, n_rings=R, ftype=1).to(dev)
    with torch.no_grad():
        mod.nonlin.bias.uniform_(-0.3, 0.1)
    g = torch.Generator().manual_seed(N)
    pos = torch.randn(N, 3, generator=g).to(dev)
    gy = torch.complex(torch.randn(N, 24, generator=g), torch.randn(N, 24, generator=g)).to(dev)
    params = list(mod.parameters())
    out = []
    for use_table in (True, False):
        x = pos.clone().requires_grad_(True)
        arg = lift_sten if use_table else sten.columns(0, 2)          # the dense (E,R,2) tensor the stand-in stands for
        y = mod(x, edges, arg)
        out.append((y.detach(),) + torch.autograd.grad(y, [x] + params, grad_outputs=gy))
    assert lift_sten._dense is None and sten._dense is None             # nothing was materialised on the way
    # (float32 rounding of the on-the-fly columns against torch's.  The gradients go through angle() and amplify it: 1.4e-5 ... 3.4e-5 over
    #  repeated runs of the SAME case -- FCPrecomp's area sums use float atomics, as the reference's index_add does on a GPU
    #  (csrc/fc_precomp.hip), so the stencil itself differs by an ulp from run to run and the amplified comparison moves with it;
    #  tools/uninit_probe.py shows the spread and that no uninitialised memory is involved)
    for n_, (a, b) in enumerate(zip(*out)):
        err = rel_err(H(a), H(b))
        assert err < (5e-6 if n_ == 0 else 1e-4), (n_, err)
    # and the stand-in still behaves like the tensor when something else asks
    assert torch.equal(lift_sten[:5], sten.columns(0, 2)[:5]) and lift_sten.abs().shape == (edges.shape[0], R, 2)


@pytest.mark.parametrize('ftype', [0, 1])
def test_lift_block_golden_f64(ftype, dev):
    """LiftBlock(...).double() -- the reference's TransField / LiftBlock run in double precision (nn/trans_field.py:78-113,
    nn/lift_block.py:53-55) -- against the reference's own float64 run (fixtures lift_block_t*_f64: 5 scalar inputs, 7 output channels):
    output, input gradient and every parameter gradient to 1e-12 (csrc/fc_lift_echo_generic.hip + fc_tangent_nonlin_*_f64)."""
    from fieldconv_amd.nn import LiftBlock
    c = load_golden('echo_lift.npz')[f'lift_block_t{ftype}_f64']
    m = load_params(LiftBlock(int(c['Cin']), int(c['Cout']), n_rings=int(c['R']), ftype=ftype).double(), c).to(dev)
    xs = D(c['x'], dev).requires_grad_(True)
    y = m(xs, D(c['edges'], dev), D(c['lift_sten'], dev))
    assert y.dtype == torch.complex128 and rel_err(H(y), c['y']) < 1e-12
    params = dict(m.named_parameters())
    grads = torch.autograd.grad(y, [xs] + list(params.values()), grad_outputs=D(c['gy'], dev))
    assert grads[0].dtype == torch.float64 and rel_err(H(grads[0]), c['gx']) < 1e-12
    for (name, _), gval in zip(params.items(), grads[1:]):
        assert rel_err(H(gval), c['g_' + name]) < 1e-12, name


def test_echo_descriptors_in_double_precision(dev):
    """ECHO descriptors of complex128 features (run-time kernels, csrc/fc_lift_echo_generic.hip) against the oracle's torch restatement in
    float64 -- the reference's own ECHO raises for double inputs, so there is no reference run to capture; the restatement is pinned to the
    reference's float32 fixtures by the CPU suite.  In double precision no vote flips a raster cell: 1e-12 on every entry."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import ECHO
    from oracle.torch_composites import FCPrecomp, echo_descriptors as echo_ref
    N, k, C = 70, 14, 6
    data = sphere_support(N, k, seed=7)
    edges, _, ln, wxp = FCPrecomp(1, 3, float(data.logMag.max()) * 1.0001)(data)
    g = torch.Generator().manual_seed(7)
    ln = torch.where(ln.abs() < 1e-6, torch.zeros_like(ln), ln).to(torch.complex128) * 0.999
    wxp = wxp.to(torch.complex128)
    x = torch.complex(torch.randn(N, C, generator=g, dtype=torch.float64), torch.randn(N, C, generator=g, dtype=torch.float64))
    x[torch.rand(N, C, generator=g) < 0.05] = 0
    for n_bins in (2, 11):
        xr = x.clone().requires_grad_(True)
        dr = echo_ref(xr, edges, ln, wxp, n_bins)
        gd = torch.randn(dr.shape, generator=g, dtype=torch.float64)
        gr, = torch.autograd.grad(dr, [xr], grad_outputs=gd)
        xd = x.to(dev).requires_grad_(True)
        dd = ECHO(C, n_bins).to(dev)(xd, edges.to(dev), ln.to(dev), wxp.to(dev))
        gg, = torch.autograd.grad(dd, [xd], grad_outputs=gd.to(dev))
        assert dd.dtype == torch.float64 and dd.shape == dr.shape
        assert rel_err(H(dd), dr.detach().numpy()) < 1e-12 and rel_err(H(gg), gr.numpy()) < 1e-11


# ------------------------------------------------------------------ double precision (the reference's modules run under .double())
FC64 = {k: v for k, v in load_golden('fieldconv.npz').items() if k.endswith('f64')}


@pytest.mark.parametrize('tag', sorted(FC64))
def test_fieldconv_golden_f64(tag, dev):
    """FieldConv(...).double() on complex128 features and a complex128 stencil: the run-time kernels in double precision
    (csrc/fc_generic.hip, fc_cgemm) against the reference's own float64 run (fixtures *_f64: the three ftypes)."""
    from fieldconv_amd.nn import FieldConv
    c = FC64[tag]
    assert c['x'].dtype == np.complex128 and c['sten'].dtype == np.complex128
    ftype, B, R = int(c['ftype']), int(c['B']), int(c['R'])
    conv = FieldConv(c['x'].shape[1], c['y'].shape[1], band_limit=B, n_rings=R, ftype=ftype).double()
    conv.load_state_dict({'zonal': torch.from_numpy(c['zonal']), 'spherical': torch.from_numpy(c['spherical']),
                          'phase': torch.from_numpy(c['phase'])})
    conv = conv.to(dev)
    x = D(c['x'], dev).requires_grad_(True)
    y = conv(x, D(c['edges'], dev), D(c['sten'], dev))
    assert y.dtype == torch.complex128
    assert rel_err(H(y), c['y']) < 1e-12
    params = dict(conv.named_parameters())
    grads = torch.autograd.grad(y, [x] + list(params.values()), grad_outputs=D(c['gy'], dev))
    assert rel_err(H(grads[0]), c['gx']) < 1e-12
    for (name, _), g in zip(params.items(), grads[1:]):
        assert g.dtype == torch.float64 and rel_err(H(g), c['g_' + name]) < 1e-12, name


@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128])
def test_cgemm_three_layouts(dtype, dev):
    """fc_cgemm (MFMA f32 / f64) in the three forms the run-time path uses -- A . B^T, A . conj(B), A^T . conj(B) -- on sizes
    that are no multiples of the 64 x 64 x 16 tiling, against torch.matmul in double precision."""
    from fieldconv_amd import _lib
    from fieldconv_amd.functional import _cgemm
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)

    def rnd(*shape):
        return torch.complex(torch.randn(*shape, generator=g, dtype=torch.float64), torch.randn(*shape, generator=g, dtype=torch.float64))
    tol = 1e-13 if dtype == torch.complex128 else 2e-6
    # (the last two: weight-gradient products with a long contraction -- 128 x 128 outputs over 20 000 / 5 003 vertices -- which
    #  fc_cgemm splits along k over many workgroups: fc_cgemm_workspace_bytes > 0)
    assert lib.fc_cgemm_workspace_bytes(128, 128, 20000, 0) > 0 and lib.fc_cgemm_workspace_bytes(20000, 128, 128, 0) == 0
    for (n, O, K) in ((200, 7, 90), (130, 70, 333), (65, 64, 16), (1, 1, 1), (20000, 128, 128), (5003, 70, 40)):
        A, W, G = rnd(n, K), rnd(O, K), rnd(n, O)
        Ad, Wd, Gd = A.to(dtype).to(dev), W.to(dtype).to(dev), G.to(dtype).to(dev)
        y = torch.empty((n, O), dtype=dtype, device=dev)
        _cgemm(lib, Ad, Wd, y, n, O, K, K, 1, 1, K, False, 0.5)
        assert rel_err(H(y).astype(np.complex128), (A @ W.T * 0.5).numpy()) < tol
        gc = torch.empty((n, K), dtype=dtype, device=dev)
        _cgemm(lib, Gd, Wd, gc, n, K, O, O, 1, K, 1, True, 1.0)
        assert rel_err(H(gc).astype(np.complex128), (G @ W.conj()).numpy()) < tol
        gw = torch.empty((O, K), dtype=dtype, device=dev)
        _cgemm(lib, Gd, Ad, gw, O, K, n, 1, O, K, 1, True, 2.0)
        assert rel_err(H(gw).astype(np.complex128), (G.T @ A.conj() * 2.0).numpy()) < tol


def test_fc_resnet_block_in_double_precision(dev):
    """FCResNetBlock(...).double(): FieldConv, TangentLin and TangentNonLin in double precision.  Output against the oracle
    evaluated in complex128; every gradient through a central difference of the loss along a random direction (double
    precision makes that a 1e-7 check)."""
    from fieldconv_amd.nn import FCResNetBlock
    g = torch.Generator().manual_seed(17)
    N, k, Cin, Cout, B, R = 150, 9, 5, 7, 2, 4
    F = 2 * B + 1
    dst = torch.arange(N).repeat_interleave(k)
    src = torch.randint(0, N, (N * k,), generator=g)
    order = torch.argsort(src, stable=True)
    edges = torch.stack((src[order], dst[order]), 1)
    sten = torch.complex(torch.randn(N * k, R, F, generator=g, dtype=torch.float64), torch.randn(N * k, R, F, generator=g, dtype=torch.float64)) * 0.3
    x = torch.complex(torch.randn(N, Cin, generator=g, dtype=torch.float64), torch.randn(N, Cin, generator=g, dtype=torch.float64))
    # (no exact zeros here: the operator is not differentiable at the origin box -- the reference assigns such entries a zero
    #  angle gradient by convention, which a central difference across the box cannot reproduce; the fixtures cover them)
    blk = FCResNetBlock(Cin, Cout, band_limit=B, n_rings=R, ftype=1).double().to(dev)
    xd = x.to(dev).requires_grad_(True)
    ed, sd = edges.to(dev), sten.to(dev)
    y = blk(xd, ed, sd)
    p = {n_: H(v) for n_, v in blk.state_dict().items()}
    y_ref = orc.fc_resnet_block_forward(x.numpy(), edges.numpy(), sten.numpy(), p, 1, B)
    assert y.dtype == torch.complex128 and rel_err(H(y), y_ref) < 1e-12
    gy = torch.complex(torch.randn(N, Cout, generator=g, dtype=torch.float64), torch.randn(N, Cout, generator=g, dtype=torch.float64)).to(dev)
    params = list(blk.parameters())
    grads = torch.autograd.grad(y, [xd] + params, grad_outputs=gy)

    def loss():
        with torch.no_grad():
            return float(torch.sum(torch.real(torch.conj(gy) * blk(xd, ed, sd))))
    eps = 1e-6
    for t, gt in zip([xd] + params, grads):
        v = torch.randn(t.shape, generator=g, dtype=torch.float64).to(dev)
        if t.is_complex():
            v = torch.complex(v, torch.randn(t.shape, generator=g, dtype=torch.float64).to(dev))
        with torch.no_grad():
            t.add_(eps * v)
            lp = loss()
            t.sub_(2 * eps * v)
            lm = loss()
            t.add_(eps * v)
        fd = (lp - lm) / (2 * eps)
        an = float(torch.sum(torch.real(torch.conj(gt) * v))) if t.is_complex() else float(torch.sum(gt * v))
        assert abs(fd - an) < 1e-6 * max(1.0, abs(an)), (tuple(t.shape), fd, an)


@pytest.mark.gpu
@pytest.mark.parametrize('N,D,C,Q,H1,H2', [(1024, 1392, 48, 8, 128, 64), (4999, 156, 16, 64, 128, 64), (37, 29, 5, 3, 128, 64),
                                           (333, 260, 64, 64, 128, 64), (16, 4, 1, 1, 32, 16), (2050, 75, 7, 10, 96, 48), (1, 3, 2, 2, 50, 30),
                                           (129, 513, 33, 17, 127, 63)])
def test_echo_head_matches_the_dense_layers_in_double_precision(dev, N, D, C, Q, H1, H2):
    """ECHOBlock's tail lin3(relu(lin2(relu(lin1(d))))) + res(softAbs(x)) (reference nn/echo_block.py:95-103) through fc_echo_head_forward /
    fc_echo_head_backward -- fp32 matrix-pipe products, k-split weight gradients, fused masks and bias sums -- against the same layers
    composed in float64 torch: output and all ten gradients (descriptors, input, four weights, four biases) within 1e-5 of the largest
    reference entry (the fp32 ATen composition itself sits at ~3e-7 ... 2e-6 on these sizes).  Shapes: config 3's and config 5's heads,
    ragged sizes (N, D, C, Q not multiples of anything), a single tile, non-reference hidden widths."""
    from fieldconv_amd.blocks import _EchoHeadFn, head_supported
    g = torch.Generator().manual_seed(N + D)
    d = torch.rand(N, D, generator=g).to(dev)
    x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev)
    x[::7, 0] = 0                                      # origins: softAbs and its gradient are 0 there (reference utils/field.py:40-48)
    def lin(o, i):
        return ((torch.rand(o, i, generator=g) * 2 - 1) / i ** 0.5).to(dev), ((torch.rand(o, generator=g) * 2 - 1) / i ** 0.5).to(dev)
    params = [t for pair in (lin(H1, D), lin(H2, H1), lin(Q, H2), lin(Q, C)) for t in pair]
    assert head_supported(d, x, params[0], params[2], params[4], params[6])
    gy = torch.randn(N, Q, generator=g).to(dev)

    leaves = [d.clone().requires_grad_(True), x.clone().requires_grad_(True)] + [p.clone().requires_grad_(True) for p in params]
    y = _EchoHeadFn.apply(*leaves)
    grads = torch.autograd.grad(y, leaves, gy)

    ref_leaves = [t.detach().to(torch.float64 if not t.is_complex() else torch.complex128).requires_grad_(True) for t in leaves]
    d64, x64, w1, b1, w2, b2, w3, b3, wr, br = ref_leaves
    org = (x64.detach().real.abs() < 1e-7) & (x64.detach().imag.abs() < 1e-7)          # the origin box
    a = torch.where(org, torch.zeros_like(x64.real), x64.abs())
    h = torch.relu(torch.relu(d64 @ w1.t() + b1) @ w2.t() + b2)
    y_ref = h @ w3.t() + b3 + a @ wr.t() + br
    ref_grads = torch.autograd.grad(y_ref, ref_leaves, gy.double())

    def close(got, want, what):
        want = want.to(got.dtype)
        err = (got - want).abs().max().item()
        scale = max(want.abs().max().item(), 1e-30)
        assert err <= 1e-5 * scale, f'{what}: {err:.3e} vs scale {scale:.3e}'
    close(y, y_ref, 'y')
    for name, got, want in zip(('g_d', 'gx', 'g_w1', 'g_b1', 'g_w2', 'g_b2', 'g_w3', 'g_b3', 'g_wr', 'g_br'), grads, ref_grads):
        close(got, want, name)
    # the same bits when run again (fixed summation orders)
    y2 = _EchoHeadFn.apply(*leaves)
    grads2 = torch.autograd.grad(y2, leaves, gy)
    assert torch.equal(y, y2) and all(torch.equal(p, q) for p, q in zip(grads, grads2))


@pytest.mark.gpu
def test_echo_head_limits_fall_back_to_the_dense_layers(dev):
    """The native head takes the reference's hidden widths (<= 128 / 64) and at most 64 channels either side; beyond that the entry points
    say FC_ERR_UNSUPPORTED, a short workspace FC_ERR_WORKSPACE and a missing pointer FC_ERR_BAD_ARGUMENT -- nothing is enqueued -- and
    ECHOBlock composes its tail of ATen layers inside one node (blocks._EchoTailFn), with the same result as torch's own modules."""
    import ctypes
    from fieldconv_amd import _lib, blocks
    from fieldconv_amd.functional import _p, _stream
    lib = _lib.load()
    N, D, C, Q = 64, 40, 8, 5
    g = torch.Generator().manual_seed(3)
    d = torch.rand(N, D, generator=g).to(dev)
    x = torch.complex(torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)).to(dev)
    for H1, H2, rc_expected in ((256, 64, -2), (128, 96, -2), (128, 64, 0)):
        W = [torch.randn(*s, generator=g).to(dev) * 0.1 for s in ((H1, D), (H1,), (H2, H1), (H2,), (Q, H2), (Q,), (Q, C), (Q,))]
        hp = _lib.FcEchoHeadParams(D, H1, H2, C, Q, *[w.data_ptr() for w in W], *([None] * 8))
        h1, h2, y = torch.empty(N, H1, device=dev), torch.empty(N, H2, device=dev), torch.empty(N, Q, device=dev)
        nws = lib.fc_echo_head_forward_workspace_bytes(N, ctypes.byref(hp))
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
        rc = lib.fc_echo_head_forward(_p(d), _p(x), ctypes.byref(hp), _p(h1), _p(h2), _p(y), _p(ws), nws, N, _stream())
        assert rc == rc_expected, (H1, H2, rc)
        assert blocks.head_supported(d, x, W[0], W[2], W[4], W[6]) == (rc_expected == 0)
        if rc_expected == 0:
            assert lib.fc_echo_head_forward(_p(d), None, ctypes.byref(hp), _p(h1), _p(h2), _p(y), _p(ws), nws, N, _stream()) == -1
            nb = lib.fc_echo_head_backward_workspace_bytes(N, ctypes.byref(hp))
            assert nb > 0
            gbuf = [torch.empty_like(w) for w in W]
            hp2 = _lib.FcEchoHeadParams(D, H1, H2, C, Q, *[w.data_ptr() for w in W], *[t.data_ptr() for t in gbuf])
            g_d, gx, g_h1 = torch.empty(N, D, device=dev), torch.empty(N, C, dtype=torch.cfloat, device=dev), torch.empty(N, H1, device=dev)
            short = torch.empty(16, dtype=torch.uint8, device=dev)
            assert lib.fc_echo_head_backward(_p(d), _p(x), _p(h1), _p(h2), _p(y), ctypes.byref(hp2), _p(g_d), _p(gx), _p(g_h1), _p(short), 16, N,
                                             _stream()) == -4
        else:
            # the module path: one node of ATen layers, equal to the modules' own composition up to the order of its sums
            leaves = [d.clone().requires_grad_(True), x.clone().requires_grad_(True)] + [w.clone().requires_grad_(True) for w in W]
            y_node = blocks._EchoTailFn.apply(*leaves)
            a = torch.where((x.real.abs() < 1e-7) & (x.imag.abs() < 1e-7), torch.zeros_like(x.real), x.abs())
            y_ref = torch.relu(torch.relu(d @ W[0].t() + W[1]) @ W[2].t() + W[3]) @ W[4].t() + W[5] + a @ W[6].t() + W[7]
            assert rel_err(H(y_node), H(y_ref)) < 2e-6


@pytest.mark.parametrize('N,k,I,O,B,R', [(3000, 12, 48, 48, 2, 6), (9000, 8, 48, 48, 2, 6), (8203, 9, 16, 32, 1, 4), (9000, 8, 64, 16, 2, 6),
                                         (8500, 8, 24, 48, 1, 8), (3080, 20, 48, 48, 2, 6), (4500, 40, 32, 48, 2, 4), (5003, 11, 48, 16, 3, 4)])
def test_two_arithmetic_modes_in_one_process(dev, N, k, I, O, B, R):
    """The arithmetic mode travels in the dims of every call (fc_dims::mode, ABI 11): convolutions of one process run in different
    modes side by side -- `with fieldconv_amd.arithmetic(...)` -- and the library keeps no state between them.  From 3 072 vertices (9 000 here, and
    the smallest meshes the arrangement takes: 3 080, 4 500, 5 003)
    the default mode's backward pass is the gather / stream / gx arrangement while fp32 runs the data / filter kernel pair: the
    two agree to fp32 rounding (and both with the oracle elsewhere) -- also on other shapes the arrangement takes through its run-time
    sized instantiations (16 -> 32 channels on 4 rings: one gxt wavefront; 64 -> 16 channels: three reduction rounds; 8 rings)."""
    if os.environ.get('FC_MFMA') not in (None, '', 'split'):
        pytest.skip('compares the modes against the default one')
    import fieldconv_amd
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import FieldConv
    from fieldconv_amd.transforms import FCPrecomp
    data = sphere_support(N, k=k, seed=4, support='p95').to(dev)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    conv = FieldConv(I, O, band_limit=B, n_rings=R, ftype=1).to(dev)
    gen = torch.Generator().manual_seed(N)
    x = torch.complex(torch.randn(N, I, generator=gen), torch.randn(N, I, generator=gen)).to(dev).requires_grad_(True)
    gy = torch.complex(torch.randn(N, O, generator=gen), torch.randn(N, O, generator=gen)).to(dev)
    if N >= 3072:
        import ctypes
        from fieldconv_amd import _lib
        d = _lib.FcDims(N, int(edges.shape[0]), I, O, R, B)
        print('H-streaming backward:', bool(_lib.load().fc_backward_streams(ctypes.byref(d), 1)))

    def step():
        y = conv(x, edges, sten)
        return (y.detach(),) + torch.autograd.grad(y, [x] + list(conv.parameters()), grad_outputs=gy)

    first = step()
    with fieldconv_amd.arithmetic('f32'):
        f32 = step()
        with fieldconv_amd.arithmetic('f16'):
            f16 = step()
        f32_again = step()
    again = step()
    H = lambda t: t.detach().cpu().numpy()
    for a, b, c, d, e in zip(first, f32, f16, f32_again, again):
        assert torch.equal(a, e) and torch.equal(b, d)                   # nothing of a mode lingers in the library or the plan cache
        assert rel_err(H(a), H(b)) < 1e-5                               # split halves = fp32 MFMA to fp32 rounding
        assert 1e-7 < rel_err(H(c), H(b)) < 5e-3                        # the reduced-precision mode really ran (and is what it claims)
