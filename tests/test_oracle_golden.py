"""Pin the CPU oracle (oracle/fieldconv_oracle.py) to vectors captured from the reference.

The reference ships no tests or golden vectors; the fixtures under
tests/golden were produced by running the reference modules themselves
(tests/golden/make_golden.py).  fp64 fixtures separate algorithmic from rounding error.
"""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import fieldconv_oracle as orc

FC = load_golden('fieldconv.npz')
TOL32 = 2e-6      # reference fp32 vs its own fp64 is ~2e-7
TOL64 = 1e-12


def _tol(tag):
    return TOL64 if tag.endswith('f64') else TOL32


@pytest.mark.parametrize('tag', sorted(FC))
def test_fieldconv_forward_backward(tag):
    c = FC[tag]
    ftype, B = int(c['ftype']), int(c['B'])
    W = orc.effective_filter(c['zonal'], c['spherical'], c['phase'], ftype, B)
    y = orc.fieldconv_forward(c['x'], c['edges'], c['sten'], W)
    assert rel_err(y, c['y']) < _tol(tag)
    gx, gW = orc.fieldconv_backward(c['x'], c['edges'], c['sten'], W, c['gy'])
    assert rel_err(gx, c['gx']) < _tol(tag) * 5
    gz, gs, gp = orc.effective_filter_vjp(gW, c['zonal'], c['spherical'], c['phase'], ftype, B)
    assert rel_err(gz, c['g_zonal']) < _tol(tag) * 5
    assert rel_err(gs, c['g_spherical']) < _tol(tag) * 5
    if ftype == 1:
        assert rel_err(gp, c['g_phase']) < _tol(tag) * 5


def test_fieldconv_fp64_fixture_is_tighter_than_fp32():
    """The fp32 fixture differs from the fp64 one only by rounding (same seed -> same inputs up to dtype)."""
    a, b = FC['fieldconv_s0_t1_B2_R6_f32'], FC['fieldconv_s0_t1_B2_R6_f64']
    assert a['edges'].shape == b['edges'].shape


@pytest.mark.parametrize('tag', sorted(load_golden('precomp.npz')))
def test_fc_precomp(tag):
    c = load_golden('precomp.npz')[tag]
    e, sten, ln, wxp = orc.fc_precomp(c['logMag'], c['logAng'], c['w'], c['edges'], c['xp'],
                                     int(c['B']), int(c['R']), float(c['eps']))
    assert np.array_equal(e, c['out_edges'])
    assert rel_err(sten, c['out_sten']) < 5e-6
    assert rel_err(ln, c['out_ln']) < 5e-6
    assert rel_err(wxp, c['out_wxp']) < 5e-6
    # structure the fast paths rely on (fc_precomp.py:24-25,95): <=2 non-zero rings, rank-1 in (r,f)
    nzr = (np.abs(c['out_sten']).sum(-1) > 0).sum(-1)
    assert nzr.max() <= 2


def test_tangent_lin():
    c = load_golden('pointwise.npz')['tangent_lin']
    assert rel_err(orc.tangent_lin_forward(c['x'], c['Re'], c['Im']), c['y']) < TOL32
    gx, gRe, gIm = orc.tangent_lin_backward(c['x'], c['Re'], c['Im'], c['gy'])
    assert rel_err(gx, c['gx']) < TOL32
    assert rel_err(gRe, c['gRe']) < TOL32 * 5
    assert rel_err(gIm, c['gIm']) < TOL32 * 5


def test_tangent_nonlin():
    c = load_golden('pointwise.npz')['tangent_nonlin']
    assert rel_err(orc.tangent_nonlin_forward(c['x'], c['bias']), c['y']) < TOL32
    gx, gb = orc.tangent_nonlin_backward(c['x'], c['bias'], c['gy'])
    assert rel_err(gx, c['gx']) < TOL32 * 5
    assert rel_err(gb, c['gbias']) < TOL32 * 5


@pytest.mark.parametrize('tag', sorted(load_golden('blocks.npz')))
def test_fc_resnet_block_forward(tag):
    c = load_golden('blocks.npz')[tag]
    p = {k[2:]: v for k, v in c.items() if k.startswith('p_')}
    y = orc.fc_resnet_block_forward(c['x'], c['edges'], c['sten'], p, int(c['ftype']), int(c['B']))
    assert rel_err(y, c['y']) < 5e-6


@pytest.mark.parametrize('tag', sorted(k for k in FC if k.endswith('f32')))
def test_reference_port_torch(tag):
    """The torch CPU port timed as `cpu_baseline` in bench.py reproduces the reference outputs/grads."""
    import torch
    from oracle import reference_port_torch as port
    c = FC[tag]
    ftype, B = int(c['ftype']), int(c['B'])
    x = torch.from_numpy(c['x']).requires_grad_(True)
    z = torch.from_numpy(c['zonal']).requires_grad_(True)
    s = torch.from_numpy(c['spherical']).requires_grad_(True)
    p = torch.from_numpy(c['phase']).requires_grad_(ftype == 1)
    y = port.field_conv(x, torch.from_numpy(c['edges']), torch.from_numpy(c['sten']), z, s, p, ftype, B)
    assert rel_err(y.detach().numpy(), c['y']) < TOL32
    gx, gz = torch.autograd.grad(y, [x, z], grad_outputs=torch.from_numpy(c['gy']))
    assert rel_err(gx.numpy(), c['gx']) < TOL32 * 5
    assert rel_err(gz.numpy(), c['g_zonal']) < TOL32 * 5


def test_correspondence_fixture_parameter_fill_is_reproducible():
    """net_correspondence.npz keeps no parameters: generator and GPU test fill them from tests/golden/param_fill.py.  The
    fill must give the generator's values bit for bit (probes stored in the fixture), here through our own module classes
    -- which also pins parameter names and shapes of the correspondence topology to the reference's."""
    import os
    import sys
    import torch
    from conftest import GOLDEN, load_golden
    sys.path.insert(0, GOLDEN)
    from param_fill import fill_params
    from fieldconv_amd.nn import FCResNetBlock, TangentPerceptron
    c = load_golden('net_correspondence.npz')['correspondence_net']
    B, R, nf = int(c['B']), int(c['R']), int(c['nf'])
    mods = torch.nn.ModuleDict(dict(resnet2=FCResNetBlock(nf, nf, band_limit=B, n_rings=R, ftype=1), res3=TangentPerceptron(nf, nf),
                                    lin2=torch.nn.Linear(256, int(c['n_classes']))))
    fill_params(mods)
    pr = dict(mods.named_parameters())
    probe = np.concatenate([pr[n].detach().numpy().reshape(-1)[:: max(1, pr[n].numel() // 64)][:64]
                            for n in ('resnet2.conv1.spherical', 'res3.lin.Im', 'lin2.weight')])
    assert np.array_equal(probe, c['pfill_probe'])
    assert np.isfinite(c['logits']).all() and c['logits'].shape == (c['pos'].shape[0], int(c['n_classes']))
    # every parameter of the full topology has a gradient sample in the fixture
    assert sum(1 for k in c if k.startswith('g_')) == sum(1 for k in c if k.startswith('gstat_')) >= 100
