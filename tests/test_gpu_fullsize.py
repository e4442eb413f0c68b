"""Parity of the kernels the benchmark times, at the sizes it times them.

Meshes with more than 4096 vertices make every persistent workgroup of the record-driven (factored / geometric)
kernels walk several tiles: next-tile prefetch of the record ring, slab / scale buffer toggling, XCD-aware tile
order.  The tests below run those paths on FCPrecomp stencils (rank-1, two adjacent rings) and compare

  * whole outputs with the oracle at sizes the oracle finishes in seconds (including shapes whose contraction
    index is padded, where stale LDS contents would show),
  * at BASELINE configs[1] (20k vertices, k=32, C=48, B=2, R=6) random rows of y / gx with the oracle on the
    sub-edge-lists that determine them, gW through the adjoint identity, and the record-driven kernels with the
    dense-stencil kernels on the same inputs,
  * one FCResNetBlock at that size: the loss <gy, y> with gy supported on a few rows depends on a two-hop
    sub-mesh only, so outputs and EVERY gradient can be compared with the reference-structured torch port.

Tolerance: max|delta| <= 1e-5 max|ref|."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import fieldconv_oracle as orc

pytestmark = pytest.mark.gpu
REDUCED = os.environ.get('FC_MFMA') == 'f16'
TOL = 5e-3 if REDUCED else 1e-5


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


def H(t):
    return t.detach().cpu().numpy()


def precomp_case(N, k, I, O, B, R, seed, support='p95'):
    """Sphere mesh + FCPrecomp stencil (built by the oracle on the CPU), features with exact zeros, filter."""
    from fieldconv_amd.data import sphere_support
    from oracle.torch_composites import FCPrecomp
    data = sphere_support(N, k, seed=seed, support=support)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    g = torch.Generator().manual_seed(seed * 31 + I)
    x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
    x[torch.rand(N, I, generator=g) < 0.01] = 0
    x[5, 0] = complex(5e-8, -2e-8)                 # inside the origin box
    gy = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
    F = 2 * B + 1
    W = torch.complex(torch.randn(O, I, R, F, generator=g), torch.randn(O, I, R, F, generator=g)) / (I * R) ** 0.5
    return edges, sten, x, gy, W


def run_conv(graph, x, W, gy, dev):
    from fieldconv_amd.functional import field_conv
    xd = x.to(dev).requires_grad_(True)
    Wd = W.to(dev).requires_grad_(True)
    y = field_conv(xd, Wd, graph)
    gx, gW = torch.autograd.grad(y, [xd, Wd], grad_outputs=gy.to(dev))
    return y.detach(), gx, gW


@pytest.mark.parametrize('shape', [
    # N,    k,  I,  O, B, R
    (6000, 8, 12, 12, 2, 6),       # 375 tiles on 256 workgroups
    (4400, 6, 40, 40, 2, 6),       # k = 6*40 = 240 padded to 256: stale bytes in the padding would poison later tiles
    (4400, 6, 48, 48, 2, 5),       # k = 5*48 = 240 padded to 256
    (4400, 6, 8, 16, 1, 3),        # narrow layer: k = 24 padded to 32
    (5000, 7, 64, 64, 3, 6),       # FAUST shape (C=64, B=3), 313 tiles: the H-streaming backward with two walks per vertex (4 + 3 frequencies) and the k range in two halves (14 slices); in fp32 mode the data kernel's two frequency groups as separate work items
    (9000, 6, 16, 24, 2, 8),       # eight rings at band limit 2: groups of 3 + 2 frequencies, 563 tiles (separate work items again)
    (4200, 9, 24, 56, 1, 8),       # eight rings
    (8990, 7, 12, 16, 2, 6),       # 562 tiles: more than two workgroups per CU can hold at once (FC_RING=1: half tiles in the last round)
    (4800, 7, 48, 48, 2, 6),       # the default layer at 300 tiles: the H-streaming backward near the smallest mesh it takes (192 tiles), 5.9 records per workgroup
    (3100, 8, 16, 32, 1, 4),       # 194 tiles, one gxt wavefront, one gW tile per wavefront: the streaming arrangement's smallest geometry
], ids=lambda s: 'N%d_k%d_I%d_O%d_B%d_R%d' % s)
def test_multi_tile_workgroups_whole_tensors(shape, dev, monkeypatch):
    """Every output of the record-driven kernels against the oracle on meshes where a workgroup walks more than one
    tile; geometric records, generic factored records and dense rows."""
    from fieldconv_amd.graph import SupportGraph
    N, k, I, O, B, R = shape
    edges, sten, x, gy, W = precomp_case(N, k, I, O, B, R, seed=N + I)
    y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy())
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    ed, sd = edges.to(dev), sten.to(dev)
    no_geo_run = os.environ.get('FIELDCONV_NO_GEO') == '1'           # a mode run of tests/test_gpu_modes.py
    graphs = {'geometric': SupportGraph(ed, sd, N)}
    monkeypatch.setenv('FIELDCONV_NO_GEO', '1')
    graphs['factored'] = SupportGraph(ed, sd, N)
    monkeypatch.delenv('FIELDCONV_NO_GEO')
    graphs['dense'] = SupportGraph(ed, sd, N, allow_factored=False)
    if os.environ.get('FIELDCONV_DENSE', '0') != '1':
        assert graphs['geometric'].geo_t is not None or no_geo_run
        assert graphs['factored'].factored and graphs['factored'].geo_t is None and not graphs['dense'].factored
    for name, graph in graphs.items():
        y, gx, gW = run_conv(graph, x, W, gy, dev)
        assert torch.isfinite(torch.view_as_real(y)).all(), name
        assert rel_err(H(y), y_ref) < TOL, name
        assert rel_err(H(gx), gx_ref) < TOL, name
        assert rel_err(H(gW), gW_ref) < TOL, name


@pytest.mark.parametrize('scale', [1e-5, 1.0, 1e4])
def test_small_cotangent_and_sources_without_edges(scale, dev):
    """Filter gradient with a cotangent of any magnitude on a mesh where every tenth vertex has no out-edges.  The split-halves
    filter-gradient kernels scale their second operand x~[j][i] / s_j by the column maximum over a tile; a source row whose H
    is all zero used to enter that maximum with scale 1 and -- when the other rows' scales are large, i.e. the cotangent is
    small -- pushed them into the half-precision denormals (round 3: 4e-3 instead of 4e-7 on the segmentation-net step,
    where the cotangents are 1e-4).  Such rows now carry the inverse scale 0."""
    from fieldconv_amd.graph import SupportGraph
    N, k, I, O, B, R = 1500, 10, 24, 16, 2, 6
    edges, sten, x, gy, W = precomp_case(N, k, I, O, B, R, seed=77)
    keep = (edges[:, 0] % 10) != 3                       # vertices 3, 13, 23, ... are sources of nothing
    edges, sten = edges[keep].contiguous(), sten[keep].contiguous()
    gy = gy * scale
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    assert graph.factored or os.environ.get('FIELDCONV_DENSE', '0') == '1'
    y, gx, gW = run_conv(graph, x, W, gy, dev)
    assert rel_err(H(gx), gx_ref) < TOL
    assert rel_err(H(gW), gW_ref) < TOL


def test_config2_record_kernels_vs_oracle_rows_and_dense(dev, monkeypatch):
    """BASELINE configs[1] on the benchmark's own mesh (sphere, k = 32, 95-percentile support radius, FCPrecomp
    stencil): the geometric and the generic factored kernels, ~5 tiles per workgroup."""
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    N, k, I, O, B, R = 20000, 32, 48, 48, 2, 6
    edges, sten, x, gy, W = precomp_case(N, k, I, O, B, R, seed=0)
    assert 0.93 * N * k < edges.shape[0] < 0.97 * N * k          # the radius filter drops ~5 % of the k-NN edges
    ed, sd = edges.to(dev), sten.to(dev)
    geo = SupportGraph(ed, sd, N)
    monkeypatch.setenv('FIELDCONV_NO_GEO', '1')
    gen = SupportGraph(ed, sd, N)
    monkeypatch.delenv('FIELDCONV_NO_GEO')
    dense = SupportGraph(ed, sd, N, allow_factored=False)
    res = {name: run_conv(graph, x, W, gy, dev) for name, graph in (('geometric', geo), ('factored', gen), ('dense', dense))}

    g = torch.Generator().manual_seed(5)
    sub = torch.randperm(N, generator=g)[:150]
    idx = sub.numpy()
    # output rows depend only on the in-edges of those rows; input-gradient rows only on their out-edges
    m_in = torch.isin(edges[:, 1], sub)
    y_ref = orc.fieldconv_forward(x.numpy(), edges[m_in].numpy(), sten[m_in].numpy(), W.numpy())[idx]
    m_out = torch.isin(edges[:, 0], sub)
    gx_ref = orc.fieldconv_backward(x.numpy(), edges[m_out].numpy(), sten[m_out].numpy(), W.numpy(), gy.numpy())[0][idx]
    V = (torch.complex(torch.randn(W.shape, generator=g), torch.randn(W.shape, generator=g)) * 0.1).to(dev)
    for name, graph in (('geometric', geo), ('factored', gen), ('dense', dense)):
        y, gx, gW = res[name]
        assert rel_err(H(y)[idx], y_ref) < TOL, name
        assert rel_err(H(gx)[idx], gx_ref) < TOL, name
        # filter gradient: adjoint identity Re<gy, conv(x; V)> = Re<gW, V>
        with torch.no_grad():
            yv = field_conv(x.to(dev), V, graph)
        lhs = torch.sum(torch.conj(gy.to(dev)) * yv).real.item()
        rhs = torch.sum(torch.conj(gW) * V).real.item()
        assert abs(lhs - rhs) <= max(2e-4, TOL) * max(abs(lhs), abs(rhs), 1.0), name
    # the three kernel families agree on every entry (dense rows were oracle-checked at this size in round 1)
    for name in ('geometric', 'factored'):
        for a, b, what in zip(res[name], res['dense'], ('y', 'gx', 'gW')):
            assert rel_err(H(a), H(b)) < 2 * TOL, (name, what)
    # deterministic: bitwise equal on a second run
    y2, gx2, gW2 = run_conv(geo, x, W, gy, dev)
    for a, b in zip(res['geometric'], (y2, gx2, gW2)):
        assert torch.equal(torch.view_as_real(a), torch.view_as_real(b))


@pytest.mark.skipif(REDUCED, reason='two convolutions deep with modReLU in between: checks the fp32-grade path')
def test_config2_fc_resnet_block_rows(dev):
    """FCResNetBlock (reference nn/fc_resnet_block.py:84-88) at 20k vertices.  With a cotangent gy supported on a few
    rows, L = Re<gy, y> depends on the two-hop in-neighbourhood of those rows only: the reference-structured torch
    port evaluates the same function on that sub-edge-list, so y on the rows, gx and every parameter gradient are
    comparable in full."""
    from fieldconv_amd.nn import FCResNetBlock
    from oracle import reference_port_torch as port
    from oracle.torch_composites import tangent_lin, tangent_nonlin
    N, k, C, B, R = 20000, 32, 48, 2, 6
    edges, sten, x, _, _ = precomp_case(N, k, C, C, B, R, seed=0)
    g = torch.Generator().manual_seed(9)
    rows = torch.randperm(N, generator=g)[:8]
    gy = torch.zeros(N, C, dtype=torch.cfloat)
    gy[rows] = torch.complex(torch.randn(8, C, generator=g), torch.randn(8, C, generator=g))
    torch.manual_seed(3)
    blk = FCResNetBlock(C, C, band_limit=B, n_rings=R, ftype=1)
    with torch.no_grad():
        blk.nonlin1.bias.uniform_(-0.3, 0.1)           # some channels clipped by the modReLU, most not
        blk.nonlin2.bias.uniform_(-0.3, 0.1)
    names = [n for n, _ in blk.named_parameters()]

    # ---- reference: the same block on the sub-mesh (hop 1: in-edges of the rows; hop 2: in-edges of their sources),
    # vertices renumbered compactly (the port materialises (N,O,I,R,F) like the reference does)
    hop1 = torch.isin(edges[:, 1], rows)
    s1 = torch.unique(torch.cat((edges[hop1, 0], rows)))
    mask = torch.isin(edges[:, 1], s1)
    verts = torch.unique(torch.cat((edges[mask, 0], s1)))
    local = torch.full((N,), -1, dtype=torch.long)
    local[verts] = torch.arange(verts.numel())
    # The yardstick in float64, and once more in float32 to measure what fp32 rounding alone does to this function: conv1's
    # parameter gradients pass through modReLU's tangential term (f(r)/r) g_t, which amplifies the rounding of the first
    # convolution's output wherever a channel's bias is positive and its radius small -- ANY fp32 evaluation sits 1e-5 ... 3e-5 from
    # the float64 one there (conv2's gradients, which do not: 5e-7 ... 2e-6)
    e_sub = local[edges[mask]]

    def reference(rdtype, cdtype):
        s_sub = sten[mask].to(cdtype)
        ref_p = {n: p.detach().to(rdtype).clone().requires_grad_(True) for n, p in blk.named_parameters()}
        xr = x[verts].to(cdtype).clone().requires_grad_(True)

        def conv(xin, pre):
            return port.field_conv(xin, e_sub, s_sub, ref_p[pre + '.zonal'], ref_p[pre + '.spherical'], ref_p[pre + '.phase'], 1, B)
        h = tangent_nonlin(conv(xr, 'conv1'), ref_p['nonlin1.bias'])
        h = conv(h, 'conv2') + tangent_lin(xr, ref_p['res.Re'], ref_p['res.Im'])
        yr = tangent_nonlin(h, ref_p['nonlin2.bias'])
        return yr, torch.autograd.grad(yr, [xr] + [ref_p[n] for n in names], grad_outputs=gy[verts].to(cdtype))

    yr, gr = reference(torch.float64, torch.cdouble)
    _, gr32 = reference(torch.float32, torch.cfloat)
    cond = {name: rel_err(a.numpy(), b.numpy()) for name, a, b in zip(names, gr32[1:], gr[1:])}

    blk = blk.to(dev)
    xd = x.to(dev).requires_grad_(True)
    yd = blk(xd, edges.to(dev), sten.to(dev))
    gd = torch.autograd.grad(yd, [xd] + [p for _, p in blk.named_parameters()], grad_outputs=gy.to(dev))
    assert rel_err(H(yd)[rows.numpy()], yr.detach().numpy()[local[rows].numpy()]) < TOL
    gxd = H(gd[0])
    outside = np.ones(N, dtype=bool)
    outside[verts.numpy()] = False
    assert not gxd[outside].any()                       # nothing outside the two-hop neighbourhood receives gradient
    assert rel_err(gxd[verts.numpy()], gr[0].numpy()) < 2 * TOL
    errs = {name: rel_err(H(a), b.numpy()) for name, a, b in zip(names, gd[1:], gr[1:])}
    print('FCResNetBlock at config 2, parameter gradients against float64:', {n: '%.1e' % e for n, e in errs.items()})
    print('the reference-structured port in float32 against itself in float64:', {n: '%.1e' % e for n, e in cond.items()})
    for name, e in errs.items():
        assert e < max(2 * TOL, 4 * cond[name]), (name, e, cond[name])


@pytest.mark.parametrize('N,k,B,R', [(6000, 9, 2, 6), (300, 20, 3, 8), (900, 12, 1, 3)])
def test_fused_precomp_graph(dev, monkeypatch, N, k, B, R):
    """FCPrecomp's default path (fc_precomp_mark + fc_precomp_graph) goes from the reference's inputs straight to the
    support graph and the per-edge records and returns a FactoredStencil in place of the (E,R,F) tensor.  Against the
    literal outputs (fc_precomp_build), the oracle's FCPrecomp, and the convolution on either: same edges / ln / wxp, the
    dense rows on demand, the two columns LiftBlock takes without materialising the rest, equal convolution results."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import FactoredStencil, SupportGraph, get_edge_csr, get_graph
    from fieldconv_amd.transforms import FCPrecomp
    from oracle.torch_composites import FCPrecomp as FCPrecompRef
    if os.environ.get('FIELDCONV_DENSE') == '1' or os.environ.get('FIELDCONV_EAGER_STENCIL') == '1':
        pytest.skip('the fused build is switched off in this mode')
    data = sphere_support(N, k, seed=N, support='p95')
    dd = data.to(dev)
    e1, s1, l1, w1 = FCPrecomp(B, R, data.epsilon)(dd)
    assert isinstance(s1, FactoredStencil) and tuple(s1.shape) == (e1.shape[0], R, 2 * B + 1) and s1._dense is None
    monkeypatch.setenv('FIELDCONV_EAGER_STENCIL', '1')
    e2, s2, l2, w2 = FCPrecomp(B, R, data.epsilon)(dd)
    monkeypatch.delenv('FIELDCONV_EAGER_STENCIL')
    assert torch.is_tensor(s2) and torch.equal(e1, e2) and torch.equal(torch.view_as_real(l1), torch.view_as_real(l2))
    assert rel_err(H(w1), H(w2)) < 1e-6                          # the area sums are float atomics: order-dependent rounding
    e3, s3, _, _ = FCPrecompRef(B, R, data.epsilon)(data)
    assert torch.equal(e1.cpu(), e3)

    I, O = 24, 16
    g = torch.Generator().manual_seed(N)
    x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
    gy = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
    W = torch.complex(torch.randn(O, I, R, 2 * B + 1, generator=g), torch.randn(O, I, R, 2 * B + 1, generator=g)) / (I * R) ** 0.5
    graph = get_graph(e1, s1, N)
    assert graph is s1.graph and graph.factored and get_edge_csr(e1, N).rowptr_t is graph.rowptr_t
    ya, gxa, gWa = run_conv(graph, x, W, gy, dev)
    yb, gxb, gWb = run_conv(SupportGraph(e2, s2, N), x, W, gy, dev)
    y_ref = orc.fieldconv_forward(x.numpy(), e3.numpy(), s3.numpy(), W.numpy())
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), e3.numpy(), s3.numpy(), W.numpy(), gy.numpy())
    for a, b, r in ((ya, yb, y_ref), (gxa, gxb, gx_ref), (gWa, gWb, gW_ref)):
        assert rel_err(H(a), r) < TOL and rel_err(H(a), H(b)) < TOL
    if B >= 1:
        lift = s1[..., B:B + 2]                          # what the notebooks hand to LiftBlock
        assert s1._dense is None and tuple(lift.shape) == (e1.shape[0], R, 2)
        assert rel_err(H(lift), H(s2[..., B:B + 2])) < 2e-6
    assert rel_err(H(s1), H(s2)) < 2e-6 and s1._dense is not None          # any other use: the dense rows, once
    assert rel_err(H(torch.abs(s1)), H(s2.abs())) < 2e-6 and tuple(s1[5].shape) == (R, 2 * B + 1)
    bad = dd.supp_edges.clone()
    bad[int(torch.argmin(dd.logMag)), 0] = N + 3          # an edge inside the support radius (dropped edges are never read)
    dd.supp_edges = bad
    with pytest.raises(IndexError):
        FCPrecomp(B, R, data.epsilon)(dd)


def test_config2_on_the_graph_the_benchmark_builds(dev):
    """BASELINE configs[1] on the support graph bench.py itself convolves over: the one fieldconv_amd.transforms.FCPrecomp
    returns at 20 000 vertices (fc_precomp_mark + fc_precomp_graph: selection, both groupings and the per-edge records
    straight from (logMag, logAng, w, supp_edges, xp) -- no dense stencil anywhere).  Checked against the oracle, whose
    stencil comes from its own FCPrecomp on the CPU: kept edges / ln / wxp, 150 random rows of y and gx on the sub-edge-lists
    that determine them, the filter gradient through the adjoint identity with the ORACLE evaluating the left-hand side
    (linear in the filter: a few probe filters on the sub-mesh), and the grouping arrays against a stable sort."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import FactoredStencil, get_graph
    from fieldconv_amd.transforms import FCPrecomp
    from oracle.torch_composites import FCPrecomp as FCPrecompRef
    if os.environ.get('FIELDCONV_DENSE') == '1' or os.environ.get('FIELDCONV_EAGER_STENCIL') == '1':
        pytest.skip('the fused build is switched off in this mode')
    N, k, I, O, B, R = 20000, 32, 48, 48, 2, 6
    data = sphere_support(N, k, seed=0, support='p95')
    e_dev, s_dev, ln_dev, wxp_dev = FCPrecomp(B, R, data.epsilon)(data.to(dev))
    assert isinstance(s_dev, FactoredStencil) and s_dev._dense is None
    graph = get_graph(e_dev, s_dev, N)
    assert graph is s_dev.graph and graph.factored and graph.geo_t is not None
    edges, sten, ln, wxp = FCPrecompRef(B, R, data.epsilon)(data)
    assert torch.equal(e_dev.cpu(), edges)
    assert rel_err(H(ln_dev), H(ln)) < 1e-6 and rel_err(H(wxp_dev), H(wxp)) < 1e-6
    # groupings: CSR by target / by source of the kept edges, slots ordered by (vertex, lower ring, edge id)
    E = edges.shape[0]
    for col, rowptr, perm in ((1, graph.rowptr_t, graph.perm_t), (0, graph.rowptr_s, graph.perm_s)):
        deg = torch.bincount(edges[:, col], minlength=N)
        assert torch.equal(rowptr.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(deg, 0)]))
        pc = perm.cpu().long()
        assert torch.equal(torch.sort(pc).values, torch.arange(E))           # a permutation of the kept edges
        assert torch.equal(edges[pc, col], torch.repeat_interleave(torch.arange(N), deg))   # grouped by that endpoint

    g = torch.Generator().manual_seed(11)
    x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
    x[torch.rand(N, I, generator=g) < 0.01] = 0
    gy = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
    W = torch.complex(torch.randn(O, I, R, 2 * B + 1, generator=g), torch.randn(O, I, R, 2 * B + 1, generator=g)) / (I * R) ** 0.5
    y, gx, gW = run_conv(graph, x, W, gy, dev)
    sub = torch.randperm(N, generator=g)[:150]
    idx = sub.numpy()
    m_in = torch.isin(edges[:, 1], sub)
    y_ref = orc.fieldconv_forward(x.numpy(), edges[m_in].numpy(), sten[m_in].numpy(), W.numpy())[idx]
    m_out = torch.isin(edges[:, 0], sub)
    gx_ref = orc.fieldconv_backward(x.numpy(), edges[m_out].numpy(), sten[m_out].numpy(), W.numpy(), gy.numpy())[0][idx]
    assert rel_err(H(y)[idx], y_ref) < TOL
    assert rel_err(H(gx)[idx], gx_ref) < TOL
    # filter gradient: Re<gy, conv(x; V)> = Re<gW, V> with gy supported on `sub`, so that the ORACLE can evaluate the left-hand
    # side on the sub-edge-list; gW of that cotangent comes from the kernels on the whole mesh
    gy_sub = torch.zeros_like(gy)
    gy_sub[sub] = gy[sub]
    _, _, gW_sub = run_conv(graph, x, W, gy_sub, dev)
    for trial in range(3):
        V = torch.complex(torch.randn(W.shape, generator=g), torch.randn(W.shape, generator=g)) * 0.1
        yv = orc.fieldconv_forward(x.numpy(), edges[m_in].numpy(), sten[m_in].numpy(), V.numpy())
        lhs = float(np.sum(np.conj(gy_sub.numpy()) * yv).real)
        rhs = float(torch.sum(torch.conj(gW_sub.cpu()) * V).real)
        assert abs(lhs - rhs) <= max(2e-4, TOL) * max(abs(lhs), abs(rhs), 1.0), trial
    # and on the whole mesh against the kernels' own forward pass
    Vd = (torch.complex(torch.randn(W.shape, generator=g), torch.randn(W.shape, generator=g)) * 0.1).to(dev)
    with torch.no_grad():
        yv = field_conv(x.to(dev), Vd, graph)
    lhs = torch.sum(torch.conj(gy.to(dev)) * yv).real.item()
    rhs = torch.sum(torch.conj(gW) * Vd).real.item()
    assert abs(lhs - rhs) <= max(2e-4, TOL) * max(abs(lhs), abs(rhs), 1.0)


def test_precomp_cache_round_trip(dev, tmp_path):
    """Cached per-mesh preprocessing (transforms/precomp_cache.py): what the fused FCPrecomp produced is written to a file
    and read back; the loaded (supp_edges, supp_sten, ln, wxp) drive a FCResNetBlock and an ECHOBlock to bit-identical
    results without any preprocessing kernel."""
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.graph import FactoredStencil
    from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock
    from fieldconv_amd.transforms import FCPrecomp, load_precomp, save_precomp
    if os.environ.get('FIELDCONV_DENSE') == '1' or os.environ.get('FIELDCONV_EAGER_STENCIL') == '1':
        pytest.skip('the fused build is switched off in this mode')
    N, k, C, B, R = 700, 18, 12, 2, 6
    data = sphere_support(N, k, seed=3, support='p95').to(dev)
    out = FCPrecomp(B, R, data.epsilon)(data)
    path = str(tmp_path / 'mesh.fcp')
    save_precomp(path, out, B, data.epsilon)
    e2, s2, l2, w2 = load_precomp(path, dev)
    assert isinstance(s2, FactoredStencil) and torch.equal(e2, out[0]) and torch.equal(s2.factors, out[1].factors)
    for name in ('rowptr_t', 'nbr_t', 'runs_s', 'perm_s', 'rec_t', 'rec_s', 'geo_t'):
        assert torch.equal(getattr(s2.graph, name), getattr(out[1].graph, name)), name
    torch.manual_seed(0)
    mods = torch.nn.ModuleDict(dict(lift=LiftBlock(3, C, n_rings=R), res=FCResNetBlock(C, C, band_limit=B, n_rings=R),
                                    echo=ECHOBlock(C, 5, n_des=C, n_bins=2, band_limit=B, n_rings=R))).to(dev)
    pos = torch.randn(N, 3, device=dev)

    def run(edges, sten, ln, wxp):
        x = mods['lift'](pos, edges, sten[..., B:B + 2])
        x = mods['res'](x, edges, sten)
        return mods['echo'](x, edges, sten, ln, wxp)
    ya, yb = run(*out), run(e2, s2, l2, w2)
    assert torch.equal(ya, yb)
    with pytest.raises(ValueError):
        torch.save({'format': 'something else'}, path)
        load_precomp(path, dev)


@pytest.mark.parametrize('N,k,I,O,B,R,n_first', [
    (20000, 32, 48, 48, 2, 6, 17000),      # ring-major kernel on the first range, frequency-major on the second
    (9000, 16, 24, 32, 2, 6, 8200),        # both ranges on one kernel family; ranges not multiples of the 16-vertex tile
    (700, 12, 16, 16, 1, 3, 5),            # a first range smaller than one tile
])
def test_forward_in_two_row_ranges_is_the_same_forward(dev, N, k, I, O, B, R, n_first):
    """dist.overlap_forward launches the targets [0, n_first) and [n_first, N) separately (the halo exchange completes in
    between).  Same records, same filter; the kernel variant follows the size of each launch (ring-major from 257 tiles
    up), so the outputs agree to fp32 rounding rather than bit for bit; with and without the fused residual + modReLU
    epilogue; the backward pass does not change at all."""
    from fieldconv_amd.functional import field_conv_act
    from fieldconv_amd.graph import SupportGraph
    from fieldconv_amd.nn import FieldConv
    edges, sten, x, gy, W = precomp_case(N, k, I, O, B, R, seed=4)
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    assert graph.factored
    calls = []
    y0, gx0, gW0 = run_conv(graph, x, W, gy, dev)
    graph.forward_split = (n_first, lambda: calls.append(1))
    y1, gx1, gW1 = run_conv(graph, x, W, gy, dev)
    assert calls == [1]
    assert rel_err(H(y1), H(y0)) < 3e-6
    assert np.array_equal(H(gx0), H(gx1)) and np.array_equal(H(gW0), H(gW1))

    torch.manual_seed(3)
    conv = FieldConv(I, O, band_limit=B, n_rings=R, ftype=1).to(dev)
    bias = (0.05 * torch.randn(1, O)).to(dev)
    addend = gy.to(dev)
    xd = x.to(dev)
    out = []
    for split in (None, (n_first, lambda: None)):
        graph.forward_split = split
        out.append(field_conv_act(xd, conv.zonal, conv.spherical, conv.phase, conv.ftype, B, graph, bias, addend=addend))
    assert rel_err(H(out[1]), H(out[0])) < 3e-6


def test_restricted_targets_of_a_partitioned_mesh(dev):
    """One patch of a two-way partition: the halo vertices (the last rows) are sources only.  With
    SupportGraph.restrict_targets the forward pass returns the owned rows only and the backward pass takes the owned rows'
    gradient; same numbers as the unrestricted graph with the halo rows of gy set to zero -- bit for bit in the backward
    pass, to rounding in the forward pass (the kernel variant follows the number of rows launched)."""
    from fieldconv_amd.data import sphere_partition
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import SupportGraph
    from oracle.torch_composites import FCPrecomp
    B, R, C, O = 2, 6, 24, 16
    data, n_owned, halo, _ = sphere_partition(9000, 2, 1, k=12, seed=2, interior_first=True)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    n_local = data.num_nodes
    assert n_local > n_owned
    g = torch.Generator().manual_seed(8)
    x = torch.complex(torch.randn(n_local, C, generator=g), torch.randn(n_local, C, generator=g))
    gy = torch.complex(torch.randn(n_local, O, generator=g), torch.randn(n_local, O, generator=g))
    gy[n_owned:] = 0
    W = torch.complex(torch.randn(O, C, R, 2 * B + 1, generator=g), torch.randn(O, C, R, 2 * B + 1, generator=g)) / (C * R) ** 0.5
    full = SupportGraph(edges.to(dev), sten.to(dev), n_local)
    y0, gx0, gW0 = run_conv(full, x, W, gy, dev)
    assert float(y0[n_owned:].abs().max()) == 0.0
    part = SupportGraph(edges.to(dev), sten.to(dev), n_local).restrict_targets(n_owned)
    with pytest.raises(ValueError):
        SupportGraph(edges.to(dev), sten.to(dev), n_local).restrict_targets(n_owned - 1)
    for split in (None, (data.n_interior, lambda: None)):
        part.forward_split = split
        xd, Wd = x.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True)
        y = field_conv(xd, Wd, part)
        assert tuple(y.shape) == (n_owned, O)
        gx, gW = torch.autograd.grad(y, [xd, Wd], grad_outputs=gy[:n_owned].to(dev))
        assert rel_err(H(y), H(y0[:n_owned])) < 3e-6
        assert np.array_equal(H(gx), H(gx0)) and np.array_equal(H(gW), H(gW0))
    y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy())
    assert rel_err(H(y), y_ref[:n_owned]) < TOL


@pytest.mark.parametrize('R', [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize('B', [1, 2, 3])
def test_every_compiled_shape_on_a_two_round_mesh(dev, R, B):
    """Every (n_rings, band_limit) pair the library compiles (csrc/fc_kernels.hpp: FC_FOR_EACH_SHAPE), on a mesh of 8 208
    vertices = 513 tiles: the ring-major forward kernel of that shape runs (two rounds of its persistent grid and a half
    tile), the backward kernels walk several tiles per workgroup.  Whole outputs against the oracle."""
    from fieldconv_amd.graph import SupportGraph
    N, k, I, O = 8208, 5, 8 + 4 * (R % 3), 16 - 4 * (B % 2)
    edges, sten, x, gy, W = precomp_case(N, k, I, O, B, R, seed=10 * R + B)
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    assert graph.factored
    y, gx, gW = run_conv(graph, x, W, gy, dev)
    y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy())
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    assert rel_err(H(y), y_ref) < TOL and rel_err(H(gx), gx_ref) < TOL and rel_err(H(gW), gW_ref) < TOL


def test_ring_major_forward_with_empty_outer_rings(dev):
    """Round 1's benchmark mesh: the support radius lies above every k-NN distance, so FCPrecomp drops no edge and the two
    outermost rings of the radial interpolant stay empty (`--support all`).  The ring-major forward then flushes and
    contracts slabs nobody gathered into; whole outputs against the oracle on a two-round mesh."""
    from fieldconv_amd.graph import SupportGraph
    N, k, I, O, B, R = 8208, 6, 24, 24, 2, 6
    edges, sten, x, gy, W = precomp_case(N, k, I, O, B, R, seed=3, support='all')
    assert edges.shape[0] == N * k
    assert float(sten[:, R - 1].abs().max()) == 0.0                 # nothing reaches the outermost ring
    graph = SupportGraph(edges.to(dev), sten.to(dev), N)
    y, gx, gW = run_conv(graph, x, W, gy, dev)
    y_ref = orc.fieldconv_forward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy())
    gx_ref, gW_ref = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.numpy(), gy.numpy())
    assert rel_err(H(y), y_ref) < TOL and rel_err(H(gx), gx_ref) < TOL and rel_err(H(gW), gW_ref) < TOL


@pytest.mark.skipif(REDUCED, reason='checks the fp32-grade path')
def test_config3_full_size_every_convolution_against_the_oracle(dev, monkeypatch):
    """BASELINE configs[2] at ITS size (reference segmentation.ipynb:165-236): LiftBlock(3 -> 48), four
    FCResNetBlocks, ECHOBlock(48 -> 8) forward + loss + backward on a 1 024-vertex mesh with ~128 neighbours per vertex
    (131 000 edges: the small-mesh regime -- 64 tiles, edge split over several workgroups per tile, frequency-major forward).
    The golden network fixture pins the topology end to end at N = 96; here every one of the nine FieldConv launches of the
    full-size step is checked on its own: the inputs it actually received (spied at the launch wrappers) go through the
    oracle, which must reproduce its output, its input gradient and its filter gradient to 1e-5."""
    from fieldconv_amd import functional as Fn
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import ECHOBlock, FCResNetBlock, LiftBlock
    from fieldconv_amd.transforms import FCPrecomp
    from oracle.torch_composites import FCPrecomp as OracleFCPrecomp
    N, k, nf, B, R, n_classes = 1024, 128, 48, 2, 6, 8
    data = sphere_support(N, k, seed=3)
    torch.manual_seed(11)
    mods = torch.nn.ModuleDict(dict(
        lift=LiftBlock(3, nf, n_rings=R, ftype=1),
        r1=FCResNetBlock(nf, nf, band_limit=B, n_rings=R), r2=FCResNetBlock(nf, nf, band_limit=B, n_rings=R),
        r3=FCResNetBlock(nf, nf, band_limit=B, n_rings=R), r4=FCResNetBlock(nf, nf, band_limit=B, n_rings=R),
        echo=ECHOBlock(nf, n_classes, n_des=48, n_bins=3, band_limit=B, n_rings=R))).to(dev)
    convs = [c for name in ('r1', 'r2', 'r3', 'r4') for c in (mods[name].conv1, mods[name].conv2)] + [mods['echo'].conv]
    g = torch.Generator().manual_seed(5)
    pos = torch.randn(N, 3, generator=g).to(dev)
    labels = torch.randint(0, n_classes, (N,), generator=g).to(dev)

    fwd_calls, bwd_calls = [], {}
    orig_f, orig_b = Fn._run_forward, Fn._launch_backward

    def spy_f(lib, x, graph, plan, O, st, pack, addend=None, bias=None, params=None):
        out = orig_f(lib, x, graph, plan, O, st, pack, addend=addend, bias=bias, params=params)
        res = out[0]
        y = res[0] if isinstance(res, tuple) else res            # the pre-activation: convolution + the block's residual (epilogue)
        fwd_calls.append((x.data_ptr(), H(x), H(y), H(addend) if addend is not None else None))
        return out

    def spy_b(lib, x, gy, graph, wpk_b, plan, wshape, st, params=None, **kw):
        res = orig_b(lib, x, gy, graph, wpk_b, plan, wshape, st, params=params, **kw)
        bwd_calls[x.data_ptr()] = (H(gy), H(res[0]), H(res[1]))
        return res

    dd = data.to(dev)
    edges, sten, ln, wxp = FCPrecomp(B, R, data.epsilon)(dd)

    def run_step():
        x = mods['lift'](pos, edges, sten[..., B:B + 2])
        for name in ('r1', 'r2', 'r3', 'r4'):
            x = mods[name](x, edges, sten)
        logits = mods['echo'](x, edges, sten, ln, wxp)
        loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), labels)
        grads = torch.autograd.grad(loss, list(mods.parameters()))
        torch.cuda.synchronize()
        return logits.detach(), loss.detach(), grads

    # (1) the step as the modules run it: ONE native call per block and pass (csrc/fc_blocks.hip) -- no launch wrapper to spy on;
    # (2) the same step composed of per-operator calls, spied per convolution.  The block-level entry points enqueue the same kernels in
    # the same order, so (1) must equal (2) BIT FOR BIT: what the oracle confirms for (2) below holds for the block-level path as well.
    monkeypatch.delenv('FIELDCONV_BLOCK_CALLS', raising=False)
    logits_blk, loss_blk, grads_blk = run_step()
    assert not fwd_calls
    monkeypatch.setenv('FIELDCONV_BLOCK_CALLS', '0')
    monkeypatch.setattr(Fn, 'KEEP_GW_EFF', True)       # the spies compare gW_eff, which a module's backward pass no longer materialises
    Fn._run_forward, Fn._launch_backward = spy_f, spy_b
    try:
        logits, loss, grads = run_step()
    finally:
        Fn._run_forward, Fn._launch_backward = orig_f, orig_b
        monkeypatch.delenv('FIELDCONV_BLOCK_CALLS', raising=False)
    assert torch.equal(logits_blk, logits) and torch.equal(loss_blk, loss)
    assert all(torch.equal(a, b) for a, b in zip(grads_blk, grads))
    assert len(fwd_calls) == len(convs) == 9 and len(bwd_calls) == 9, (len(fwd_calls), len(bwd_calls))
    assert all(torch.isfinite(g_).all() for g_ in grads) and bool(torch.isfinite(loss))

    e_ref, s_ref, _, _ = OracleFCPrecomp(B, R, data.epsilon)(data)          # the oracle's own stencil of the same mesh
    e_ref, s_ref = e_ref.numpy(), s_ref.numpy()
    assert e_ref.shape[0] > 100 * N                                          # ~128 neighbours per vertex survive the support radius
    worst = {}
    for n, (conv, (ptr, xin, yout, addend)) in enumerate(zip(convs, fwd_calls)):
        W = orc.effective_filter(H(conv.zonal), H(conv.spherical), H(conv.phase), 1, B)
        gy, gx, gw = bwd_calls[ptr]
        y_ref, gx_ref, gW_ref = orc.fieldconv_forward_backward(xin, e_ref, s_ref, W, gy)
        if addend is not None:
            y_ref = y_ref + addend          # what the kernel's epilogue writes (the residual dwarfs a freshly initialised convolution)
        errs = (rel_err(yout, y_ref), rel_err(gx, gx_ref), rel_err(gw, gW_ref))
        worst[n] = errs
        assert max(errs) < TOL, (n, errs)
    print('config 3, per convolution (y, gx, gW) errors:', {n: tuple('%.1e' % e for e in v) for n, v in worst.items()})


def _stream_shape_classes(N):
    """One (I, O) per class of run-time geometry the H-streaming backward takes: rings, band limit, channel tiles, k steps, gxt wavefronts,
    gW tiles per wavefront, ragged channel tiles (fc_backward_stream.hpp: plan_stream)."""
    import ctypes
    import itertools
    from fieldconv_amd import _lib
    lib = _lib.load()
    classes = {}
    for R, B in itertools.product((2, 4, 6, 8), (1, 2, 3)):
        for I in (2, 6, 10, 16, 18, 24, 30, 32, 34, 40, 48, 50, 56, 64):
            for O in range(1, 65):
                d = _lib.FcDims(N, N * 8, I, O, R, B)
                if not lib.fc_backward_streams(ctypes.byref(d), 1):
                    continue
                # (plan_stream's geometry: walks per vertex, the k range whole or in two halves, units, wavefronts per role, tiles per gW wavefront)
                KP, NMT, NG = R * O, (I + 15) // 16, ((2 * B + 1) * R + 31) // 32
                key = None
                for KS in (1, 2):
                    KPS = KP // KS
                    KST = KPS // 32
                    G = (NMT * KST + 3) // 4
                    NW = 16 - G
                    if KST < 3 or NW < NMT or NW < 8:
                        continue
                    T = -(-(KPS // 16) // (NW // NMT))
                    if T <= 6:
                        key = (R, B, NG, KS, NMT, KST, G, T, I % 16 == 0, O % 16 == 0)
                        break
                assert key is not None, (R, B, I, O)
                classes.setdefault(key, (I, O))
    return sorted(classes.items())


def test_every_shape_class_of_the_streaming_backward(dev):
    """The gather / stream / gx arrangement against the fp32 data / filter kernel pair of the same process (fieldconv_amd.arithmetic),
    over every class of geometry its plan accepts -- rings x band limit (one or two walks per vertex) x the k range whole or in halves x
    channel tiles x k steps x wavefronts per role x tiles per gW wavefront x ragged tiles (several hundred at this size; every third one
    here plus the first and last of each (rings, band limit) pair, all of them with FC_FULL_MODES=1): y, gx and the three parameter gradients agree to fp32 rounding."""
    if os.environ.get('FC_MFMA') not in (None, '', 'split'):
        pytest.skip('the arrangement exists in the default arithmetic mode only')
    import fieldconv_amd
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import FieldConv
    from fieldconv_amd.transforms import FCPrecomp
    N = 8203
    classes = _stream_shape_classes(N)
    assert len(classes) > 200
    if os.environ.get('FC_FULL_MODES') != '1':
        ends = {}
        for n, (key, _) in enumerate(classes):
            ends.setdefault(key[:2], [n, n])[1] = n
        keep = {n for lo_hi in ends.values() for n in lo_hi} | set(range(0, len(classes), 3))
        classes = [c for n, c in enumerate(classes) if n in keep]
    meshes, worst = {}, 0.0
    for n, (key, (I, O)) in enumerate(classes):
        R, B = key[:2]
        if (R, B) not in meshes:
            data = sphere_support(N, k=8, seed=10 * R + B, support='p95').to(dev)
            meshes[(R, B)] = FCPrecomp(B, R, data.epsilon)(data)[:2]
        edges, sten = meshes[(R, B)]
        conv = FieldConv(I, O, band_limit=B, n_rings=R, ftype=1).to(dev)
        gen = torch.Generator().manual_seed(n)
        x = torch.complex(torch.randn(N, I, generator=gen), torch.randn(N, I, generator=gen)).to(dev).requires_grad_(True)
        gy = torch.complex(torch.randn(N, O, generator=gen), torch.randn(N, O, generator=gen)).to(dev)

        def step():
            y = conv(x, edges, sten)
            return (y.detach(),) + torch.autograd.grad(y, [x] + list(conv.parameters()), grad_outputs=gy)

        got = step()
        with fieldconv_amd.arithmetic('f32'):
            ref = step()
        errs = [rel_err(H(a), H(b)) for a, b in zip(got, ref)]
        assert all(np.isfinite(e) for e in errs) and max(errs) < 1e-5, (key, I, O, errs)
        worst = max(worst, max(errs))
    print('streaming backward: %d shape classes, worst relative difference to the fp32 kernel pair %.1e' % (len(classes), worst))


@pytest.mark.parametrize('N,k,mode', [pytest.param(170_003, 12, None, id='170k_2GB_of_H'), pytest.param(400_003, 10, None, id='400k_5GB_of_H'),
                                       pytest.param(400_003, 10, 'f32', id='400k_kernel_pair_5GB_workspace')])
def test_meshes_beyond_two_and_four_gigabytes_of_workspace(dev, N, k, mode):
    """One GPU holds meshes far larger than config 2 (config 4's 160 000 vertices unpartitioned, and beyond): at 170 003 vertices the
    H-streaming backward writes and reads 2 GB of records (byte offsets past 2^31), at 400 003 vertices 4.9 GB (past 2^32); in fp32
    mode the same mesh takes the data / filter kernel pair with a workspace past 4 GiB.  Rows of y and gx at the start, the END and
    the middle of the tensors against the oracle (its own FCPrecomp on the CPU, sub-edge-lists), the filter gradient through the
    adjoint identity on a cotangent supported on those rows (oracle-evaluated) and on the whole mesh (the kernels' own forward pass)."""
    import contextlib
    import fieldconv_amd
    if mode is not None and os.environ.get('FC_MFMA') not in (None, '', 'split'):
        pytest.skip('selects its own arithmetic mode')
    with (fieldconv_amd.arithmetic(mode) if mode else contextlib.nullcontext()):
        _big_mesh_case(dev, N, k)


def _big_mesh_case(dev, N, k):
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.functional import field_conv
    from fieldconv_amd.graph import get_graph
    from fieldconv_amd.transforms import FCPrecomp
    from oracle.torch_composites import FCPrecomp as FCPrecompRef
    if REDUCED or os.environ.get('FIELDCONV_DENSE') == '1' or os.environ.get('FIELDCONV_EAGER_STENCIL') == '1':
        pytest.skip('record-driven kernels in the fp32-grade modes')
    I, O, B, R = 48, 48, 2, 6
    data = sphere_support(N, k, seed=3, support='p95')
    e_dev, s_dev, _, _ = FCPrecomp(B, R, data.epsilon)(data.to(dev))
    graph = get_graph(e_dev, s_dev, N)
    edges, sten, _, _ = FCPrecompRef(B, R, data.epsilon)(data)
    assert torch.equal(e_dev.cpu(), edges)
    g = torch.Generator().manual_seed(N)
    x = torch.complex(torch.randn(N, I, generator=g), torch.randn(N, I, generator=g))
    gy = torch.complex(torch.randn(N, O, generator=g), torch.randn(N, O, generator=g))
    W = torch.complex(torch.randn(O, I, R, 2 * B + 1, generator=g), torch.randn(O, I, R, 2 * B + 1, generator=g)) / (I * R) ** 0.5
    torch.cuda.reset_peak_memory_stats()
    y, gx, gW = run_conv(graph, x, W, gy, dev)
    print('N = %d, E = %d: peak device memory of the step %.2f GB' % (N, edges.shape[0], torch.cuda.max_memory_allocated() / 2 ** 30))
    assert bool(torch.isfinite(torch.view_as_real(y)).all()) and bool(torch.isfinite(torch.view_as_real(gx)).all())
    sub = torch.cat([torch.arange(0, 10), torch.arange(N - 40, N), torch.randperm(N - 50, generator=g)[:60] + 10])
    idx = sub.numpy()
    m_in = torch.isin(edges[:, 1], sub)
    y_ref = orc.fieldconv_forward(x.numpy(), edges[m_in].numpy(), sten[m_in].numpy(), W.numpy())[idx]
    m_out = torch.isin(edges[:, 0], sub)
    gx_ref = orc.fieldconv_backward(x.numpy(), edges[m_out].numpy(), sten[m_out].numpy(), W.numpy(), gy.numpy())[0][idx]
    assert rel_err(H(y)[idx], y_ref) < TOL
    assert rel_err(H(gx)[idx], gx_ref) < TOL
    gy_sub = torch.zeros_like(gy)
    gy_sub[sub] = gy[sub]
    _, _, gW_sub = run_conv(graph, x, W, gy_sub, dev)
    V = torch.complex(torch.randn(W.shape, generator=g), torch.randn(W.shape, generator=g)) * 0.1
    yv = orc.fieldconv_forward(x.numpy(), edges[m_in].numpy(), sten[m_in].numpy(), V.numpy())
    lhs = float(np.sum(np.conj(gy_sub.numpy()) * yv).real)
    rhs = float(torch.sum(torch.conj(gW_sub.cpu()) * V).real)
    assert abs(lhs - rhs) <= max(2e-4, TOL) * max(abs(lhs), abs(rhs), 1.0)
    Vd = V.to(dev)
    with torch.no_grad():
        yv = field_conv(x.to(dev), Vd, graph)
    lhs = torch.sum(torch.conj(gy.to(dev)) * yv).real.item()
    rhs = torch.sum(torch.conj(gW) * Vd).real.item()
    assert abs(lhs - rhs) <= max(2e-4, TOL) * max(abs(lhs), abs(rhs), 1.0)
