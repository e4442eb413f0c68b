"""CPU tests of the host-side logic of fieldconv_amd (module/state_dict layout, filter assembly, graph preprocessing, the
C-ABI surface: symbols only, no kernel launches without a GPU) and of the oracle's torch restatements of FCPrecomp, ECHO
and TransField against the reference fixtures."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, rel_err
from oracle import fieldconv_oracle as orc

import fieldconv_amd
from fieldconv_amd import _lib
from fieldconv_amd.graph import SupportGraph, get_graph
from fieldconv_amd.nn import ECHO, ECHOBlock, FCResNetBlock, FieldConv, LiftBlock, TangentLin, TangentNonLin, TransField
from fieldconv_amd.nn.field_conv import effective_filter
from fieldconv_amd.utils import isOrigin, softAbs, softAngle
from oracle import torch_composites as tc
from oracle.torch_composites import FCPrecomp        # stencils for the CPU tests (the package's FCPrecomp is device-only)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ------------------------------------------------------------------ C ABI surface
def test_library_builds_and_exports_every_declared_symbol():
    from fieldconv_amd.build import build_native
    path = build_native()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, 'include', 'fieldconv_hip.h')).read()
    declared = set(re.findall(r'\b(fc_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in fieldconv_hip.h but not exported'
    assert declared == set(_lib.SIGNATURES), 'ctypes binding out of sync with the header'
    assert _lib.load(path).fc_abi_version() == 11


def test_product_library_reads_no_environment_switch():
    """The development switches (FC_RING, FC_FILTER2, FC_DEBUG, ...) are compiled into libfieldconv_hip_dev.so only (-DFC_DEV_SWITCHES);
    the product library contains none of their names -- `strings libfieldconv_hip.so | grep -c '^FC_'` is 0 -- and says so."""
    from fieldconv_amd import _env
    from fieldconv_amd.build import build_dev, build_native
    blob = open(build_native(), 'rb').read()
    names = set(m.decode() for m in re.findall(rb'FC_[A-Z][A-Z0-9_]{2,}', blob))
    assert not names, sorted(names)
    assert ctypes.CDLL(build_native()).fc_dev_switches() == 0
    dev = open(build_dev(), 'rb').read()
    dev_names = set(m.decode() for m in re.findall(rb'FC_[A-Z][A-Z0-9_]{2,}', dev))
    assert set(_env.LIBRARY_SWITCHES) <= dev_names, sorted(set(_env.LIBRARY_SWITCHES) - dev_names)
    assert ctypes.CDLL(build_dev()).fc_dev_switches() == 1


def test_cpp_autograd_nodes_build_and_bind():
    """fc_torch_nodes.so -- the block-level autograd nodes in C++ (csrc_torch/fc_torch_nodes.cpp; host-side plumbing, no GPU code) -- builds
    against the installed torch, loads, and binds to the library's entry points through function pointers (no compute without a GPU)."""
    from fieldconv_amd import blocks
    from fieldconv_amd.build import build_native, build_torch_nodes, torch_nodes_needs_build
    build_native()
    path = build_torch_nodes()
    assert os.path.exists(path) and not torch_nodes_needs_build()
    nodes = blocks.cpp_nodes()
    assert nodes is not None
    for name in ('bind', 'GraphRef', 'resnet_block', 'echo_block', 'lift_block', 'echo_tail', 'echo_head'):
        assert hasattr(nodes, name), name
    empty = torch.empty(0, dtype=torch.int32)
    ref = nodes.GraphRef([empty] * 8, 10, 0, 6, 2, 1)
    assert ref is not None
    with pytest.raises(Exception):
        nodes.GraphRef([empty] * 3, 10, 0, 6, 2, 1)


def test_supported_query_and_sizes_need_no_gpu():
    lib = _lib.load()
    d = _lib.FcDims(20000, 640000, 48, 48, 6, 2)
    assert lib.fc_supported(ctypes.byref(d)) == 1
    # split-half image (default mode): 48 row scales + 5 frequencies x 4 half planes x 48 rows x 288 entries
    assert lib.fc_packed_filter_floats_fwd(ctypes.byref(d), 0) == 48 + 5 * 2 * 48 * 288
    assert lib.fc_packed_filter_floats_bwd(ctypes.byref(d), 0) == 48 + 5 * 2 * 48 * 288
    assert lib.fc_backward_workspace_bytes(ctypes.byref(d), 0) > 0
    assert lib.fc_records_flags(ctypes.byref(d), 0) == 0 and lib.fc_records_flags(ctypes.byref(d), 1) == 1
    bad = _lib.FcDims(100, 10, 48, 48, 9, 2)
    assert lib.fc_supported(ctypes.byref(bad)) == 0
    wide = _lib.FcDims(100, 10, 128, 48, 6, 2)
    assert lib.fc_supported(ctypes.byref(wide)) == 0
    # 8 rings x 63 channels: slab + partial sums + record ring exceed the 160 KB of LDS in split mode (callers block it)
    if os.environ.get('FC_MFMA') in (None, '', 'split'):
        assert lib.fc_supported(ctypes.byref(_lib.FcDims(100, 1000, 63, 42, 8, 2))) == 0
    assert lib.fc_supported(ctypes.byref(_lib.FcDims(100, 1000, 32, 32, 8, 2))) == 1
    assert b'unsupported' in lib.fc_status_string(-2)


def test_product_path_refuses_cpu_tensors():
    conv = FieldConv(4, 4, band_limit=1, n_rings=3)
    x = torch.zeros(5, 4, dtype=torch.cfloat)
    edges = torch.zeros(3, 2, dtype=torch.long)
    sten = torch.zeros(3, 3, 3, dtype=torch.cfloat)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        conv(x, edges, sten)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        TangentLin(4, 4)(x)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        TangentNonLin(4)(x)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ECHO(4, 2)(x, edges, torch.zeros(3, dtype=torch.cfloat), torch.zeros(3, dtype=torch.cfloat))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        TransField(3, 4, n_rings=3)(torch.zeros(5, 3), edges, sten[..., :2])
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.transforms import FCPrecomp as DeviceFCPrecomp
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        DeviceFCPrecomp(1, 3, 1.0)(sphere_support(20, 4))


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'fieldconv_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                imports = re.findall(r'^\s*(?:from|import)\s+([\w\.]+)', src, flags=re.M)
                assert not any(m.split('.')[0] == 'oracle' for m in imports), f'{f} imports the oracle'


# ------------------------------------------------------------------ modules / state_dict
def test_state_dict_layout_matches_reference_fixture():
    blocks = load_golden('blocks.npz')
    for tag, c in blocks.items():
        blk = FCResNetBlock(int(c['Cin']), int(c['Cout']), band_limit=int(c['B']), n_rings=int(c['R']),
                            ftype=int(c['ftype']), frontload=bool(c['frontload']))
        sd = blk.state_dict()
        ref = {k[2:]: v for k, v in c.items() if k.startswith('p_')}
        assert list(sd) == list(ref), tag            # same names, same order
        for k in sd:
            assert tuple(sd[k].shape) == ref[k].shape, (tag, k)
        blk.load_state_dict({k: T(v) for k, v in ref.items()})     # a reference checkpoint loads
    eb = load_golden('echo_lift.npz')['echo_block']
    m = ECHOBlock(int(eb['Cin']), int(eb['Cout']), n_des=int(eb['n_des']), n_bins=int(eb['n_bins']),
                  band_limit=int(eb['B']), n_rings=int(eb['R']), ftype=1)
    ref = {k[2:]: v for k, v in eb.items() if k.startswith('p_')}
    assert list(m.state_dict()) == list(ref)
    m.load_state_dict({k: T(v) for k, v in ref.items()})
    for ft in (0, 1):
        lb = load_golden('echo_lift.npz')[f'lift_block_t{ft}']
        m = LiftBlock(int(lb['Cin']), int(lb['Cout']), n_rings=int(lb['R']), ftype=ft)
        ref = {k[2:]: v for k, v in lb.items() if k.startswith('p_')}
        assert list(m.state_dict()) == list(ref)
        m.load_state_dict({k: T(v) for k, v in ref.items()})


@pytest.mark.parametrize('ftype', [0, 1, 2])
def test_parameter_kinds_and_attributes(ftype):
    conv = FieldConv(5, 7, band_limit=2, n_rings=6, ftype=ftype)
    names = dict(conv.named_parameters())
    assert ('phase' in names) == (ftype == 1)                 # buffer for ftype 0/2 (field_conv.py:75,93)
    assert 'phase' in conv.state_dict()
    assert (conv.in_channels, conv.out_channels, conv.R, conv.B, conv.ftype) == (5, 7, 6, 2, ftype)
    assert callable(conv.WR)
    assert torch.count_nonzero(conv.zonal) > 0


@pytest.mark.parametrize('tag', sorted(k for k in load_golden('fieldconv.npz') if k.endswith('f32')))
def test_effective_filter_and_its_autograd(tag):
    c = load_golden('fieldconv.npz')[tag]
    ftype, B = int(c['ftype']), int(c['B'])
    z, s, p = (T(c['zonal']).requires_grad_(True), T(c['spherical']).requires_grad_(True),
               T(c['phase']).requires_grad_(ftype == 1))
    W = effective_filter(z, s, p, ftype, B)
    W_ref = orc.effective_filter(c['zonal'], c['spherical'], c['phase'], ftype, B)
    assert rel_err(W.detach().numpy(), W_ref) < 1e-6
    # y through the (oracle-checked) closed form, from our W
    y = orc.fieldconv_forward(c['x'], c['edges'], c['sten'], W.detach().numpy())
    assert rel_err(y, c['y']) < 2e-6
    # parameter gradients: oracle dL/dW_eff pulled back through torch autograd of effective_filter
    _, gW = orc.fieldconv_backward(c['x'], c['edges'], c['sten'], W_ref, c['gy'])
    ins = [z, s] + ([p] if ftype == 1 else [])
    grads = torch.autograd.grad(W, ins, grad_outputs=T(gW.astype(np.complex64)))
    assert rel_err(grads[0].numpy(), c['g_zonal']) < 1e-5
    assert rel_err(grads[1].numpy(), c['g_spherical']) < 1e-5
    if ftype == 1:
        assert rel_err(grads[2].numpy(), c['g_phase']) < 1e-5
    # WR keeps the reference call signature
    conv = FieldConv(c['x'].shape[1], c['y'].shape[1], band_limit=B, n_rings=int(c['R']), ftype=ftype)
    contrib = orc.fieldconv_contrib(c['x'], c['edges'], c['sten'], B)
    y2 = conv.WR(T(contrib), z.detach(), s.detach(), p.detach(), B)
    assert rel_err(y2.numpy(), c['y']) < 2e-6


# ------------------------------------------------------------------ zero-safe helpers
def test_soft_helpers_match_oracle_and_have_finite_grads():
    z = torch.tensor([0j, 3e-8 - 5e-8j, 9.9e-8 + 2e-7j, -1e-7 + 0j, 2.5j, -1.5 + 0j, 1 + 1j], dtype=torch.cfloat,
                     requires_grad=True)
    assert np.array_equal(isOrigin(z).numpy(), orc.is_origin(z.detach().numpy()))
    assert np.allclose(softAngle(z).detach().numpy(), orc.soft_angle(z.detach().numpy()), atol=1e-7)
    assert np.allclose(softAbs(z).detach().numpy(), orc.soft_abs(z.detach().numpy()), atol=1e-7)
    g, = torch.autograd.grad((softAngle(z) + softAbs(z)).sum(), z)
    assert torch.isfinite(torch.view_as_real(g)).all()
    assert g[0] == 0 and g[1] == 0                      # no gradient inside the origin box


# ------------------------------------------------------------------ stencil assembly
class _D:
    pass


@pytest.mark.parametrize('tag', sorted(load_golden('precomp.npz')))
def test_fc_precomp_matches_reference(tag):
    c = load_golden('precomp.npz')[tag]
    d = _D()
    d.logMag, d.logAng, d.w, d.supp_edges, d.xp = T(c['logMag']), T(c['logAng']), T(c['w']), T(c['edges']), T(c['xp'])
    e, sten, ln, wxp = FCPrecomp(int(c['B']), int(c['R']), float(c['eps']))(d)
    assert torch.equal(e, T(c['out_edges']))
    assert rel_err(sten.numpy(), c['out_sten']) < 5e-6
    assert rel_err(ln.numpy(), c['out_ln']) < 5e-6
    assert rel_err(wxp.numpy(), c['out_wxp']) < 5e-6
    assert sten.dtype == torch.cfloat


# ------------------------------------------------------------------ graph preprocessing
def test_support_graph_groups_edges_both_ways():
    g = torch.Generator().manual_seed(3)
    N, E = 23, 150
    edges = torch.randint(0, N, (E, 2), generator=g)
    edges[edges[:, 1] == 5, 1] = 6                     # vertex 5 has no in-edges
    sten = torch.complex(torch.randn(E, 3, 3, generator=g), torch.randn(E, 3, 3, generator=g))
    gr = SupportGraph(edges, sten, N)
    for rowptr, nbr, st, key, other in ((gr.rowptr_t, gr.nbr_t, gr.sten_t, 1, 0), (gr.rowptr_s, gr.nbr_s, gr.sten_s, 0, 1)):
        assert rowptr[0] == 0 and rowptr[-1] == E and rowptr.dtype == torch.int32 and nbr.dtype == torch.int32
        for v in range(N):
            ids = torch.nonzero(edges[:, key] == v).squeeze(-1)          # stable: original order within a group
            lo, hi = int(rowptr[v]), int(rowptr[v + 1])
            assert hi - lo == ids.numel()
            assert torch.equal(nbr[lo:hi].long(), edges[ids, other])
            assert torch.equal(st[lo:hi], sten[ids])
    assert gr.rowptr_t[6] == gr.rowptr_t[5]
    # source-sorted input (the reference's order): no second stencil copy
    order = torch.argsort(edges[:, 0], stable=True)
    gr2 = SupportGraph(edges[order], sten[order], N)
    assert gr2.sten_s.data_ptr() == sten[order].contiguous().data_ptr() or torch.equal(gr2.sten_s, sten[order])
    # cache: same tensors -> same object; in-place edit -> rebuilt
    a = get_graph(edges, sten, N)
    assert get_graph(edges, sten, N) is a
    sten.mul_(2)
    assert get_graph(edges, sten, N) is not a
    dbl = SupportGraph(edges, sten.to(torch.cdouble), N)             # double precision: dense rows for the run-time kernels, no records
    assert not dbl.factored and dbl.sten_t.dtype == torch.complex128 and torch.equal(dbl.rowptr_t, a.rowptr_t)
    with pytest.raises(ValueError):
        SupportGraph(edges, sten.real, N)
    empty = SupportGraph(torch.zeros(0, 2, dtype=torch.long), torch.zeros(0, 3, 3, dtype=torch.cfloat), 4)
    assert empty.rowptr_t.tolist() == [0] * 5


# ------------------------------------------------------------------ the oracle's torch restatements of the "next" rows
def test_echo_descriptor_matches_reference():
    c = load_golden('echo_lift.npz')['echo']
    m = ECHO(c['x'].shape[1], int(c['n_bins']))
    assert torch.equal(m.dMap, T(c['dMap'])) and torch.equal(tc.disk_map(int(c['n_bins']))[0], T(c['dMap']))
    x = T(c['x']).requires_grad_(True)
    y = tc.echo_descriptors(x, T(c['edges']), T(c['ln']), T(c['wxp']), int(c['n_bins']))
    assert rel_err(y.detach().numpy(), c['y']) < 5e-6
    gx, = torch.autograd.grad(y, x, grad_outputs=T(c['gy']))
    assert rel_err(gx.numpy(), c['gx']) < 5e-5


@pytest.mark.parametrize('ftype', [0, 1])
def test_trans_field_matches_reference(ftype):
    c = load_golden('echo_lift.npz')[f'lift_block_t{ftype}']
    m = TransField(int(c['Cin']), int(c['Cout']), n_rings=int(c['R']), ftype=ftype)
    m.load_state_dict({k[len('p_field.'):]: T(v) for k, v in c.items() if k.startswith('p_field.')})      # the reference's layout loads
    y = tc.trans_field(T(c['x']), T(c['edges']), T(c['lift_sten']), m.zonalAng, m.zonalMag, m.phase, ftype)
    # LiftBlock output = modReLU(TransField); check through the oracle's modReLU
    out = orc.tangent_nonlin_forward(y.detach().numpy(), c['p_nonlin.bias'])
    assert rel_err(out, c['y']) < 5e-6
    # the reference's LiftBlock under .double() (5 scalar inputs, 7 output channels): output and every gradient at double precision
    c = load_golden('echo_lift.npz')[f'lift_block_t{ftype}_f64']
    m = TransField(int(c['Cin']), int(c['Cout']), n_rings=int(c['R']), ftype=ftype).double()
    m.load_state_dict({k[len('p_field.'):]: T(v) for k, v in c.items() if k.startswith('p_field.')})
    bias = T(c['p_nonlin.bias']).requires_grad_(True)
    x = T(c['x']).requires_grad_(True)
    y = tc.tangent_nonlin(tc.trans_field(x, T(c['edges']), T(c['lift_sten']), m.zonalAng, m.zonalMag, m.phase, ftype), bias)
    assert y.dtype == torch.complex128 and rel_err(y.detach().numpy(), c['y']) < 1e-13
    params = dict(m.named_parameters())
    grads = torch.autograd.grad(y, [x, bias] + list(params.values()), grad_outputs=T(c['gy']))
    assert rel_err(grads[0].numpy(), c['gx']) < 1e-12 and rel_err(grads[1].numpy(), c['g_nonlin.bias']) < 1e-12
    for (name, _), gval in zip(params.items(), grads[2:]):
        assert rel_err(gval.numpy(), c['g_field.' + name]) < 1e-12, name


def test_factored_and_geometric_records_reconstruct_the_stencil():
    """FCPrecomp stencils are rank-1, 2-sparse in the ring and geometric in the frequency: the records the
    kernels read must reproduce every stencil entry; stencils without that structure are refused."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.graph import factor_stencil, geometric_phases
    for B, R in ((1, 3), (2, 6), (3, 5)):
        data = sphere_support(300, 9, seed=B)
        edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
        F = 2 * B + 1
        rec = factor_stencil(sten)
        assert rec is not None
        E = sten.shape[0]
        q = rec[:, 0].view(torch.int32).long()
        ph = torch.view_as_complex(rec[:, 4:4 + 2 * F].reshape(E, F, 2).contiguous())
        w = torch.zeros(E, R)
        w[torch.arange(E), q] = rec[:, 1]
        w[torch.arange(E), q + 1] = rec[:, 2]
        assert (w[:, :, None] * ph[:, None, :] - sten).abs().max() <= 3e-6 * sten.abs().max()
        geo = geometric_phases(rec, F)
        assert geo is not None and geo.shape == (E, 8)
        c, g = torch.complex(geo[:, 4], geo[:, 5]), torch.complex(geo[:, 6], geo[:, 7])
        m = torch.arange(-B, B + 1)
        ang = torch.angle(g)[:, None] * m[None, :]
        assert (c[:, None] * torch.polar(torch.ones_like(ang), ang) - ph).abs().max() <= 3e-6 * ph.abs().max()
        assert torch.equal(geo[:, :3], rec[:, :3])
        # a perturbed phase breaks the geometric form but not the rank-1 form
        bad = rec.clone()
        bad[7, 4] += 0.05 * float(ph.abs().max())
        assert geometric_phases(bad, F) is None
    g = torch.Generator().manual_seed(0)
    dense = torch.complex(torch.randn(50, 4, 5, generator=g), torch.randn(50, 4, 5, generator=g))
    assert factor_stencil(dense) is None


def test_factored_stencil_stands_in_for_the_tensor():
    """FactoredStencil (what the fused FCPrecomp returns in place of supp_sten): dense rows, the two LiftBlock columns
    and torch functions from the (E,8) factor table, against the oracle's FCPrecomp stencil."""
    import torch
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.graph import FactoredStencil, factor_stencil, geometric_phases
    from oracle.torch_composites import FCPrecomp
    B, R = 2, 6
    data = sphere_support(120, 9, seed=2, support='p95')
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    fac = geometric_phases(factor_stencil(sten), 2 * B + 1)          # [q, w_q, w_{q+1}, 0, c, g]: the factor table's layout
    assert fac is not None
    s = FactoredStencil(fac, R, 2 * B + 1, graph=None)
    assert tuple(s.shape) == tuple(sten.shape) and s.dim() == 3 and s.dtype == torch.complex64 and len(s) == sten.shape[0]
    lift = s[..., B:B + 2]
    assert s._dense is None and float((lift - sten[..., B:B + 2]).abs().max()) < 2e-6 * float(sten.abs().max())
    assert float((s.materialize() - sten).abs().max()) < 2e-6 * float(sten.abs().max())
    assert torch.allclose(torch.abs(s), sten.abs(), atol=1e-6) and torch.allclose(s.abs(), sten.abs(), atol=1e-6)
    assert tuple(s[3:7].shape) == (4, R, 2 * B + 1) and s.contiguous().is_contiguous()


def test_bench_refuses_debug_switches_and_explains_missing_gpus():
    """bench.py never reports a line taken with a library debug switch set, and `--gpus N` from a plain shell is handled by
    the script itself (here: no GPU, so it says so and exits non-zero without touching a device)."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, 'bench.py')
    env = {k: v for k, v in os.environ.items() if not k.startswith('FC_') and k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    res = subprocess.run([sys.executable, bench], env=dict(env, FC_DEBUG_BWD='2'), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert res.returncode != 0 and 'FC_DEBUG_BWD' in res.stderr and res.stdout.strip() == ''
    # a name with one of the package's prefixes that nothing reads (a typo, the switch of a removed kernel) is refused as well
    res = subprocess.run([sys.executable, bench], env=dict(env, FC_BWD_RING='1'), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert res.returncode != 0 and 'unknown switch' in res.stderr and 'FC_BWD_RING' in res.stderr and res.stdout.strip() == ''
    import torch
    if torch.cuda.device_count() == 0:
        res = subprocess.run([sys.executable, bench, '--gpus', '2'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                             timeout=300)
        assert res.returncode == 2 and 'GPU(s) visible' in res.stderr


def test_every_environment_switch_is_registered():
    """fieldconv_amd/_env.py lists every switch the library (getenv in csrc/), the package and bench.py read: bench.py records
    the ones that are set and refuses names it does not know, so an unlisted switch would be refused or go unreported."""
    from fieldconv_amd import _env
    found, lib_switches = set(), set()
    for d, _, files in os.walk(os.path.join(ROOT, 'fieldconv_amd')):
        for f in files:
            if f.endswith(('.hip', '.hpp', '.py')) and f != '_env.py':
                text = open(os.path.join(d, f), errors='ignore').read()
                found |= set(re.findall(r'(?:getenv|dev_env)\("([A-Z][A-Z_0-9]+)"\)', text))
                lib_switches |= set(re.findall(r'dev_env\("([A-Z][A-Z_0-9]+)"\)', text))
                found |= set(re.findall(r"environ(?:\.get)?[\(\[]\s*'((?:FC|FIELDCONV|BENCH)_[A-Z_0-9]+)'", text))
    text = open(os.path.join(ROOT, 'bench.py')).read()
    found |= set(re.findall(r"environ(?:\.get)?[\(\[]\s*'((?:FC|FIELDCONV|BENCH)_[A-Z_0-9]+)'", text))
    assert found, 'no switches parsed'
    assert found <= set(_env.SWITCHES), sorted(found - set(_env.SWITCHES))
    assert set(_env.SWITCHES) <= found, sorted(set(_env.SWITCHES) - found)          # nothing listed that nobody reads any more
    assert lib_switches == set(_env.LIBRARY_SWITCHES), sorted(lib_switches ^ set(_env.LIBRARY_SWITCHES))
    # the library sources call getenv in ONE place: dev_env under -DFC_DEV_SWITCHES (csrc/fc_common.hpp)
    for f in os.listdir(os.path.join(ROOT, 'fieldconv_amd', 'csrc')):
        text = open(os.path.join(ROOT, 'fieldconv_amd', 'csrc', f)).read()
        assert text.count('getenv(') == (1 if f == 'fc_common.hpp' else 0), f
    assert _env.unknown({'FC_MFMA': 'f32', 'FC_TYPO': '1', 'PATH': 'x'}) == ['FC_TYPO']
    assert _env.active({'FC_MFMA': 'f32', 'HOME': 'x'}) == {'FC_MFMA': 'f32'}


def test_kernel_description_and_gpu_count_need_no_gpu():
    """fc_describe_kernels names the kernel families a launch selects (bench.py's config.kernels); bench.py counts devices for
    its self-launch from sysfs, without a HIP call."""
    lib = _lib.load()
    buf = ctypes.create_string_buffer(1024)
    d = _lib.FcDims(20000, 607843, 48, 48, 6, 2)
    assert lib.fc_describe_kernels(ctypes.byref(d), 2, buf, len(buf)) == 0
    text = buf.value.decode()
    if os.environ.get('FC_MFMA') in (None, '') and os.environ.get('FC_RING') in (None, ''):
        assert 'fc_forward_ring_kernel<geometric records,split-f16>' in text, text
        if os.environ.get('FC_BWD_STREAM') in (None, ''):         # config 2 runs the H-streaming arrangement
            assert 'fc_backward_gather_kernel' in text and 'fc_backward_stream_kernel' in text and 'fc_backward_gx_kernel' in text, text
        # config 3's mesh (64 tiles) keeps the data / filter kernel pair; a FAUST-sized one (313 tiles) streams, also at config 5's layer
        # (64 channels, band limit 3: two walks per vertex, the k range in two halves: 14 slices)
        assert lib.fc_describe_kernels(ctypes.byref(_lib.FcDims(1024, 131072, 48, 48, 6, 2)), 2, buf, len(buf)) == 0
        assert b'fc_backward_filter_half2_kernel' in buf.value, buf.value
        if os.environ.get('FC_BWD_STREAM') in (None, ''):
            for dims in (_lib.FcDims(4999, 150000, 48, 48, 6, 2), _lib.FcDims(4999, 132150, 64, 64, 6, 3)):
                assert lib.fc_describe_kernels(ctypes.byref(dims), 2, buf, len(buf)) == 0
                assert b'fc_backward_stream_kernel' in buf.value, buf.value
            assert b'x14;' in buf.value, buf.value
    small = _lib.FcDims(1024, 131072, 48, 48, 6, 2)
    assert lib.fc_describe_kernels(ctypes.byref(small), 1, buf, len(buf)) == 0 and b'frequency-major' in buf.value
    assert lib.fc_describe_kernels(ctypes.byref(_lib.FcDims(100, 10, 48, 48, 9, 2)), 1, buf, len(buf)) == -2
    # (the split decisions follow the device's CU count, which the description carries: pinned for the MI355X's 256 -- the value the
    #  library also assumes when no device can be queried -- and not asserted on a part with another count)
    assert re.search(rb'cus=\d+', buf.value), buf.value
    if os.environ.get('FC_MFMA') in (None, '') and os.environ.get('FC_GROUP_SPLIT') in (None, '') and b'cus=256' in buf.value:
        # the backward data kernel runs a tile's two frequency groups as separate work items where that fills the CUs' rounds better:
        # a FAUST-sized mesh at band limit 3 (313 tiles), the reference's 1 024-vertex mesh (instead of the edge split's last doubling:
        # 2 parts, not 4) -- and not config 2 (1 250 tiles: five rounds either way)
        groups = b'frequency groups of a tile as separate work items'
        assert lib.fc_describe_kernels(ctypes.byref(d), 2, buf, len(buf)) == 0 and groups not in buf.value
        # (in fp32 mode: the default mode streams such a mesh)
        assert lib.fc_describe_kernels(ctypes.byref(_lib.FcDims(4999, 132257, 64, 64, 6, 3, mode=1)), 2, buf, len(buf)) == 0 and groups in buf.value
        assert b'tiles=313 parts=1' in buf.value, buf.value
        assert lib.fc_describe_kernels(ctypes.byref(small), 1, buf, len(buf)) == 0 and groups in buf.value and b'parts=2' in buf.value, buf.value
        assert lib.fc_describe_kernels(ctypes.byref(_lib.FcDims(7500, 200000, 64, 64, 6, 3, mode=1)), 2, buf, len(buf)) == 0 and groups not in buf.value
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    saved = {k: os.environ.pop(k, None) for k in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES')}
    try:
        n = bench.visible_gpus()
        assert n >= 0
        os.environ['HIP_VISIBLE_DEVICES'] = '0'
        assert bench.visible_gpus() == min(n, 1)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_graph_views_keep_the_cached_graph_clean():
    """get_graph hands every user of a mesh the same cached SupportGraph: target restrictions and exchange hooks go on a view
    (own hooks and plans, shared arrays), which reaches the modules through FactoredStencil.wrap."""
    from fieldconv_amd.graph import FactoredStencil, SupportGraph, get_graph
    g = torch.Generator().manual_seed(0)
    N, E, R, F = 12, 40, 3, 3
    edges = torch.stack((torch.randint(0, N, (E,), generator=g), torch.randint(0, 8, (E,), generator=g)), 1)   # targets 0..7 only
    sten = torch.complex(torch.randn(E, R, F, generator=g), torch.randn(E, R, F, generator=g))
    cached = get_graph(edges, sten, N)
    assert get_graph(edges, sten, N) is cached
    with pytest.raises(ValueError):
        cached.restrict_targets(8)                       # would change what every other user of the mesh gets
    view = cached.view()
    view.restrict_targets(8)
    view.on_gx = lambda gx: None
    assert view.n_targets == 8 and cached.n_targets == N and cached.on_gx is None and cached.forward_split is None
    assert view.rowptr_t is cached.rowptr_t and view._plans is not cached._plans
    bound = FactoredStencil.wrap(sten, view)
    assert get_graph(edges, bound, N) is view and tuple(bound.shape) == (E, R, F) and torch.equal(bound.materialize(), sten)
    own = SupportGraph(edges, sten, N)                   # a graph of one's own may be changed directly
    assert own.restrict_targets(8).n_targets == 8


# ------------------------------------------------------------------ foreign calls per network step
class _CountingLibrary:
    """The real library with every ENQUEUEING entry point (the ones that take a stream) replaced by a recorder that returns
    FC_OK: the host side of a network step -- module dispatch, autograd nodes, buffer carving, struct filling -- runs
    unchanged on CPU tensors, nothing is launched, and every crossing into the library is counted.  Pure host queries
    (sizes, fc_supported, ...) pass through to the real functions and are counted as well."""

    def __init__(self, real, header):
        self.real, self.calls = real, []
        self.enqueue = set(re.findall(r'\b(fc_[a-z0-9_]+)\s*\([^;]*?void\*\s*stream\)', header, flags=re.S))

    def __getattr__(self, name):
        fn = getattr(self.real, name)
        if name in self.enqueue:
            def rec(*args):
                self.calls.append(name)
                return 0
            return rec

        def passthrough(*args):
            self.calls.append(name)
            return fn(*args)
        return passthrough


def test_config3_step_takes_at_most_40_foreign_calls(monkeypatch):
    """BASELINE configs[2]'s network (LiftBlock, four FCResNetBlocks, ECHOBlock: nine convolutions; reference
    segmentation.ipynb:196-236) forward + loss + backward: how often does the binding cross into the library?  One call per block and
    pass = 12 (+ 2 for ECHOBlock's dense tail, fc_echo_head_*), where the per-operator path took ~80 (16 per FCResNetBlock).  The reference trains with batch size 1 on a different
    ~1k-vertex mesh every step (segmentation.ipynb:120,137), so the host's per-step cost is what a training run sees."""
    from fieldconv_amd import blocks, functional
    header = open(os.path.join(ROOT, 'include', 'fieldconv_hip.h')).read()
    counting = _CountingLibrary(_lib.load(), header)
    assert {'fc_resnet_block_forward', 'fc_resnet_block_backward', 'fc_forward_params', 'fc_backward_all'} <= counting.enqueue
    assert 'fc_supported' not in counting.enqueue and 'fc_resnet_block_saved_bytes' not in counting.enqueue
    monkeypatch.setattr(_lib, 'load', lambda path=None: counting)
    monkeypatch.setattr(functional, 'on_device', lambda t: True)
    monkeypatch.setattr(functional, '_on', lambda device: functional._NO_GUARD)
    monkeypatch.setattr(functional, '_stream', lambda: ctypes.c_void_p(0))
    for name in ('FIELDCONV_BLOCK_CALLS', 'FIELDCONV_NO_FUSED_EPILOGUE', 'FIELDCONV_NO_EDGE_SPLIT'):
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setenv('FIELDCONV_CPP_NODES', '0')      # the Python binding of the block-level nodes: its calls can be counted (the C++ nodes
                                                        # reach the library through function pointers: the same calls, one per block and pass)

    from fieldconv_amd.data import sphere_support
    N, k, nf, B, R, n_cls = 300, 24, 48, 2, 6, 8
    data = sphere_support(N, k)
    edges, sten, ln, wxp = FCPrecomp(B, R, data.epsilon)(data)
    torch.manual_seed(0)
    net = torch.nn.ModuleDict(dict(
        lift=LiftBlock(3, nf, n_rings=R, ftype=1), r1=FCResNetBlock(nf, nf, band_limit=B, n_rings=R),
        r2=FCResNetBlock(nf, nf, band_limit=B, n_rings=R), r3=FCResNetBlock(nf, nf, band_limit=B, n_rings=R),
        r4=FCResNetBlock(nf, nf, band_limit=B, n_rings=R), echo=ECHOBlock(nf, n_cls, n_des=48, n_bins=3, band_limit=B, n_rings=R)))
    params = list(net.parameters())
    pos = torch.randn(N, 3)
    labels = torch.randint(0, n_cls, (N,))

    def step():
        x = net['lift'](pos, edges, sten[..., B:B + 2])
        for name in ('r1', 'r2', 'r3', 'r4'):
            x = net[name](x, edges, sten)
        logits = net['echo'](x, edges, sten, ln, wxp)
        loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), labels)
        return torch.autograd.grad(loss, params, allow_unused=True)

    step()                                   # first step on a mesh: plans, sizes and slot orders are computed and cached
    first = list(counting.calls)
    counting.calls.clear()
    grads = step()
    steady = list(counting.calls)
    assert all(g is not None for g in grads)
    enq = [c for c in steady if c in counting.enqueue]
    assert sorted(enq) == sorted(['fc_lift_block_forward', 'fc_lift_block_backward', 'fc_echo_block_forward', 'fc_echo_block_backward',
                                  'fc_echo_head_forward', 'fc_echo_head_backward']        # (ECHOBlock's dense tail behind its descriptors)
                                 + ['fc_resnet_block_forward', 'fc_resnet_block_backward'] * 4), enq
    assert len(steady) <= 40, (len(steady), steady)
    assert len(first) <= 80, (len(first), first)
    # the per-operator composition of the same step, for the record: several times as many crossings
    monkeypatch.setenv('FIELDCONV_BLOCK_CALLS', '0')
    monkeypatch.setattr(functional, '_require_device', lambda t, what: None)
    counting.calls.clear()
    step()
    assert len(counting.calls) > 2 * len(steady), (len(counting.calls), len(steady))
