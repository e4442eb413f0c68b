"""CPU oracle for the field-convolution hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement, in our own words, of the arithmetic the
reference (twmitchel/FieldConv) performs on its hot path.  It exists so that the
HIP kernels in ``fieldconv_amd/csrc`` can be checked for parity.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it; the shipped package ``fieldconv_amd`` never does (it fails loudly when the HIP
library is missing instead of falling back to anything in here).

Parity pinning: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so every function below is pinned against fixtures under
``tests/golden/*.npz`` that were produced by importing the *reference modules
themselves* in the build container (``tests/golden/make_golden.py``); see
``tests/test_oracle_golden.py``.

Each function cites the reference lines it follows (paths relative to
/root/reference).  All functions are dtype-generic: pass complex128 inputs to get
an fp64 evaluation, complex64 for an fp32 one.
"""
from __future__ import annotations

import numpy as np

EPS = 1e-7  # utils/field.py:8


# --------------------------------------------------------------------------- #
# zero-safe polar helpers                                   utils/field.py:10-48
# --------------------------------------------------------------------------- #
def is_origin(z, eps=EPS):
    """|re|<eps and |im|<eps, strict, component-wise (utils/field.py:10-16)."""
    z = np.asarray(z)
    return (np.abs(z.real) < eps) & (np.abs(z.imag) < eps)


def soft_angle(z, eps=EPS):
    """angle(z) away from the origin box, 0 inside it (utils/field.py:40-48)."""
    z = np.asarray(z)
    out = np.zeros(z.shape, dtype=z.real.dtype)
    nz = ~is_origin(z, eps)
    out[nz] = np.angle(z[nz])
    return out


def soft_abs(z, eps=EPS):
    """|z| away from the origin box, 0 inside it (utils/field.py:29-37)."""
    z = np.asarray(z)
    out = np.zeros(z.shape, dtype=z.real.dtype)
    nz = ~is_origin(z, eps)
    out[nz] = np.abs(z[nz])
    return out


# --------------------------------------------------------------------------- #
# filter assembly                                     nn/field_conv.py:10-33
# --------------------------------------------------------------------------- #
def effective_filter(zonal, spherical, phase, ftype, B):
    """W_eff[o,i,r,f] such that y = contrib . W_eff / (2B+1).

    ftype 0 (nn/field_conv.py:12): cat(flip(conj(sph)), zonal, sph) along f.
    ftype 1 (nn/field_conv.py:18,23,25): the same coefficients times
            exp(i*phase[o,i,|m|]) (the phases are summed *after* the ring sum in
            the reference, which is the same thing because they do not depend on r).
    ftype 2 (nn/field_conv.py:31): cat(sph[..., :B], zonal_c, sph[..., B:]).
    `spherical` and complex `zonal` arrive as trailing-2 real arrays (view_as_complex).
    """
    zonal = np.asarray(zonal)
    spherical = np.asarray(spherical)
    sph = spherical[..., 0] + 1j * spherical[..., 1]            # (O,I,R,B) or (O,I,R,2B)
    if ftype in (0, 1):
        coeff = np.concatenate(
            (np.conj(sph)[..., ::-1], zonal[..., None].astype(sph.dtype), sph), axis=3)
        if ftype == 1:
            phase = np.asarray(phase)
            ph = np.concatenate((phase[:, :, :0:-1], phase), axis=-1)   # index by |m|
            coeff = coeff * np.exp(1j * ph)[:, :, None, :]
        return coeff
    zc = zonal[..., 0] + 1j * zonal[..., 1]
    return np.concatenate((sph[..., :B], zc[..., None], sph[..., B:]), axis=3)


def effective_filter_vjp(gW, zonal, spherical, phase, ftype, B):
    """Pull a gradient on W_eff back to (zonal, spherical, phase).

    Uses the torch convention for a real loss L: g = dL/dRe + i dL/dIm for every
    complex quantity, so for w = f(p) with real p:  g_p = Re(conj(dw/dp) * g_w).
    Returns (g_zonal, g_spherical, g_phase-or-None) shaped like the parameters.
    """
    zonal = np.asarray(zonal)
    spherical = np.asarray(spherical)
    sph = spherical[..., 0] + 1j * spherical[..., 1]
    gW = np.asarray(gW)
    if ftype == 2:
        g_z = gW[..., B]
        g_s = np.concatenate((gW[..., :B], gW[..., B + 1:]), axis=3)
        return (np.stack((g_z.real, g_z.imag), -1),
                np.stack((g_s.real, g_s.imag), -1), None)
    g_phase = None
    g_coeff = gW
    if ftype == 1:
        phase = np.asarray(phase)
        ph = np.concatenate((phase[:, :, :0:-1], phase), axis=-1)
        P = np.exp(1j * ph)[:, :, None, :]
        coeff = np.concatenate(
            (np.conj(sph)[..., ::-1], zonal[..., None].astype(sph.dtype), sph), axis=3)
        g_coeff = gW * np.conj(P)
        gP = np.sum(gW * np.conj(coeff), axis=2)                      # (O,I,F)
        g_ph_full = np.real(np.conj(1j * P[:, :, 0, :]) * gP)       # d/dphi exp(i phi) = i exp(i phi)
        g_phase = np.zeros(phase.shape, dtype=phase.dtype)
        g_phase[..., 0] = g_ph_full[..., B]
        for q in range(1, B + 1):
            g_phase[..., q] = g_ph_full[..., B + q] + g_ph_full[..., B - q]
    g_zonal = g_coeff[..., B].real
    # sph[b] appears at f=B+1+b directly and at f=B-1-b conjugated
    g_sph = g_coeff[..., B + 1:] + np.conj(g_coeff[..., :B][..., ::-1])
    return g_zonal, np.stack((g_sph.real, g_sph.imag), -1), g_phase


# --------------------------------------------------------------------------- #
# the operator                                       nn/field_conv.py:104-137
# --------------------------------------------------------------------------- #
def rotated_features(x, B):
    """xt[n,c,f] = x * exp(-i m phi), m = f-B, phi = soft_angle(x)  (field_conv.py:128-130)."""
    x = np.asarray(x)
    phi = soft_angle(x)
    m = np.arange(-B, B + 1)
    return x[..., None] * np.exp(-1j * m[None, None, :] * phi[..., None]).astype(x.dtype)


def _segment_sum(values, index, n):
    """out[k] = sum of values[e] over index[e] == k, k < n: the reference's scatter_add / index_add (field_conv.py:134),
    evaluated as a stable sort by index followed by one np.add.reduceat -- the same sums in the same edge order as a sequential
    add, without np.add.at's per-element Python-level dispatch (a 130 000-edge mesh takes seconds instead of minutes)."""
    values = np.asarray(values)
    index = np.asarray(index)
    out = np.zeros((n,) + values.shape[1:], dtype=values.dtype)
    if index.size == 0:
        return out
    order = np.argsort(index, kind='stable')
    sorted_idx = index[order]
    starts = np.flatnonzero(np.r_[True, sorted_idx[1:] != sorted_idx[:-1]])
    out[sorted_idx[starts]] = np.add.reduceat(values[order], starts, axis=0)
    return out


def fieldconv_contrib(x, supp_edges, supp_sten, B):
    """contrib[n,c,r,f] = sum_{e: dst_e = n} xt[src_e,c,f] * S[e,r,f]  (field_conv.py:130,134)."""
    x = np.asarray(x)
    S = np.asarray(supp_sten)
    src, dst = np.asarray(supp_edges)[:, 0], np.asarray(supp_edges)[:, 1]
    N, C = x.shape
    xt = rotated_features(x, B)
    T = xt[src][:, :, None, :] * S[:, None, :, :]
    return _segment_sum(T, dst, N)


def fieldconv_forward(x, supp_edges, supp_sten, W_eff):
    """y[n,o] = 1/F sum_{i,r,f} contrib[n,i,r,f] W_eff[o,i,r,f]  (field_conv.py:137 -> 10-33)."""
    F = W_eff.shape[-1]
    B = (F - 1) // 2
    contrib = fieldconv_contrib(x, supp_edges, supp_sten, B)
    N = contrib.shape[0]
    O = W_eff.shape[0]
    return (contrib.reshape(N, -1) @ W_eff.reshape(O, -1).T) / F


def fieldconv_backward(x, supp_edges, supp_sten, W_eff, gy, _contrib=None):
    """Closed-form VJP of fieldconv_forward w.r.t. (x, W_eff) (torch conj convention).

    Not in the reference as code (it relies on autograd through field_conv.py:128-137);
    pinned against autograd outputs captured in the golden fixtures.
      gC      = gy . conj(W) / F
      gW      = gy^T . conj(contrib) / F
      gxt[j]  = sum_{e: src_e = j} sum_r gC[dst_e,:,r,:] conj(S[e,r,:])
      gx      = sum_f gxt_f conj(u_f) + [x not origin] (i x/|x|^2) sum_f m_f Im(conj(gxt_f) xt_f)
    """
    x = np.asarray(x)
    S = np.asarray(supp_sten)
    W = np.asarray(W_eff)
    gy = np.asarray(gy)
    src, dst = np.asarray(supp_edges)[:, 0], np.asarray(supp_edges)[:, 1]
    N, C = x.shape
    O, _, R, F = W.shape
    B = (F - 1) // 2
    contrib = fieldconv_contrib(x, supp_edges, supp_sten, B) if _contrib is None else _contrib
    gC = (gy @ np.conj(W).reshape(O, -1)).reshape(N, C, R, F) / F
    gW = (gy.T @ np.conj(contrib).reshape(N, -1)).reshape(O, C, R, F) / F
    ge = np.sum(gC[dst] * np.conj(S)[:, None, :, :], axis=2)          # (E,C,F)
    gxt = _segment_sum(ge, src, N)
    phi = soft_angle(x)
    m = np.arange(-B, B + 1)
    u = np.exp(-1j * m[None, None, :] * phi[..., None])
    xt = x[..., None] * u
    gx = np.sum(gxt * np.conj(u), axis=-1)
    nz = ~is_origin(x)
    q = np.sum(m[None, None, :] * np.imag(np.conj(gxt) * xt), axis=-1)
    ang = np.zeros_like(gx)
    ang[nz] = 1j * x[nz] / (np.abs(x[nz]) ** 2) * q[nz]
    return (gx + ang).astype(x.dtype), gW.astype(W.dtype)


def fieldconv_forward_backward(x, supp_edges, supp_sten, W_eff, gy):
    """(y, gx, gW) of one layer with the gathered response shared between fieldconv_forward and fieldconv_backward (the
    full-size config-3 test evaluates nine layers of 130 000 edges each)."""
    W = np.asarray(W_eff)
    F = W.shape[-1]
    contrib = fieldconv_contrib(x, supp_edges, supp_sten, (F - 1) // 2)
    y = (contrib.reshape(contrib.shape[0], -1) @ W.reshape(W.shape[0], -1).T) / F
    gx, gW = fieldconv_backward(x, supp_edges, supp_sten, W_eff, gy, _contrib=contrib)
    return y.astype(np.asarray(x).dtype), gx, gW


# --------------------------------------------------------------------------- #
# pointwise tangent ops           nn/tangent_lin.py:27-29, nn/tangent_nonlin.py:24-35
# --------------------------------------------------------------------------- #
def tangent_lin_forward(x, Re, Im):
    """y[n,o] = sum_i x[n,i] (Re+iIm)[o,i]  (nn/tangent_lin.py:29)."""
    Wc = np.asarray(Re) + 1j * np.asarray(Im)
    return (np.asarray(x) @ Wc.T).astype(np.asarray(x).dtype)


def tangent_lin_backward(x, Re, Im, gy):
    Wc = np.asarray(Re) + 1j * np.asarray(Im)
    gx = np.asarray(gy) @ np.conj(Wc)
    gWc = np.asarray(gy).T @ np.conj(np.asarray(x))
    return gx, gWc.real, gWc.imag


def tangent_nonlin_forward(x, bias):
    """modReLU; origin-box entries pass through untouched (nn/tangent_nonlin.py:24-35)."""
    x = np.asarray(x)
    b = np.asarray(bias).reshape(1, -1)
    out = x.copy()
    nz = ~is_origin(x)
    r = np.abs(x)
    theta = np.angle(x)
    mod = np.maximum(r + b, 0.0) * np.exp(1j * theta)
    out[nz] = mod[nz].astype(x.dtype)
    return out


def tangent_nonlin_backward(x, bias, gy):
    """VJP of modReLU: radial part gated by relu, tangential part scaled by f(r)/r."""
    x = np.asarray(x)
    gy = np.asarray(gy)
    b = np.broadcast_to(np.asarray(bias).reshape(1, -1), x.shape)
    nz = ~is_origin(x)
    r = np.where(nz, np.abs(x), 1.0)
    e = np.where(nz, x / r, 1.0)
    gr = np.real(gy * np.conj(e))
    gt = np.imag(gy * np.conj(e))
    act = (r + b) > 0
    f = np.where(act, r + b, 0.0)
    gx = np.where(nz, e * (np.where(act, gr, 0.0) + 1j * (f / r) * gt), gy)
    gb = np.sum(np.where(nz & act, gr, 0.0), axis=0).reshape(np.asarray(bias).shape)
    return gx.astype(x.dtype), gb


# --------------------------------------------------------------------------- #
# run-time stencil assembly                     transforms/fc_precomp.py:10-97
# --------------------------------------------------------------------------- #
def radial_interpolant(r, n_rings):
    """Linear interpolation weights on sqrt-spaced knots (transforms/fc_precomp.py:10-27).

    Knots s_q = sqrt(q/(R-1)).  The upper knot is the first one >= r (ties pick the
    knot itself, min over non-negative differences), forced to >= 1; the lower knot
    is the one below it.  Exactly two non-zeros per row.
    """
    r = np.asarray(r)
    q = np.arange(n_rings, dtype=r.dtype)
    knots = np.sqrt(q / np.asarray(n_rings - 1, dtype=r.dtype)).astype(r.dtype)
    diff = knots[None, :] - r[:, None]
    diff = np.where(diff < 0, np.asarray(1e8, dtype=r.dtype), diff)
    c = np.argmin(diff, axis=1)
    c = np.where(c == 0, 1, c)
    f = c - 1
    w = np.zeros((r.shape[0], n_rings), dtype=r.dtype)
    rows = np.arange(r.shape[0])
    wc = (r - knots[f]) / (knots[c] - knots[f])
    w[rows, c] = wc
    w[rows, f] = 1 - wc
    return w


def fc_precomp(logMag, logAng, w, supp_edges, xp, band_limit, n_rings, epsilon):
    """(supp_edges', supp_sten, ln, wxp) as FCPrecomp.__call__ (transforms/fc_precomp.py:53-97)."""
    logMag = np.asarray(logMag)
    rdt = logMag.dtype
    r = logMag / np.asarray(epsilon, dtype=rdt)
    keep = np.nonzero(r <= 1.0)[0]
    r = r[keep]
    theta = np.asarray(logAng)[keep]
    edges = np.asarray(supp_edges)[keep]
    xp = np.asarray(xp)[keep]
    cdt = np.result_type(rdt, np.complex64)
    ln = (r * np.exp(1j * theta)).astype(cdt)
    rs = radial_interpolant(r, n_rings)
    m = np.arange(-band_limit, band_limit + 1).astype(rdt)
    fs = np.exp(1j * (m[None, :] * theta[:, None])).astype(cdt)
    wv = np.asarray(w)[:, 0]
    N = wv.shape[0]
    ws = wv[edges[:, 0]]
    tot = np.zeros(N, dtype=rdt)
    np.add.at(tot, edges[:, 1], ws)
    # the reference sizes the scatter output by max(dst)+1; indexing by dst is identical
    wsc = ws / (np.asarray(1e-12, dtype=rdt) + tot[edges[:, 1]])
    wxp = (wsc * xp).astype(cdt)
    sten = (rs[:, :, None] * fs[:, None, :] * wxp[:, None, None]).astype(cdt)
    return edges, sten, ln, wxp


# --------------------------------------------------------------------------- #
# composite block                                 nn/fc_resnet_block.py:65-88
# --------------------------------------------------------------------------- #
def fc_resnet_block_forward(x, supp_edges, supp_sten, p, ftype, B):
    """nonlin2(res(x) + conv2(nonlin1(conv1(x)))); `p` is a dict of numpy parameters
    keyed like the reference state_dict (conv1.zonal, ..., res.Re, nonlin2.bias)."""
    W1 = effective_filter(p['conv1.zonal'], p['conv1.spherical'], p['conv1.phase'], ftype, B)
    W2 = effective_filter(p['conv2.zonal'], p['conv2.spherical'], p['conv2.phase'], ftype, B)
    h = fieldconv_forward(x, supp_edges, supp_sten, W1)
    h = tangent_nonlin_forward(h, p['nonlin1.bias'])
    h = fieldconv_forward(h, supp_edges, supp_sten, W2)
    h = tangent_lin_forward(x, p['res.Re'], p['res.Im']) + h
    return tangent_nonlin_forward(h, p['nonlin2.bias'])
