"""Run-time stencil assembly (same call contract as reference transforms/fc_precomp.py:30-97).  Device tensors go through
the library (csrc/fc_precomp.hip: three launches); the torch code below is what CPU tensors take and what the CPU suite pins
to the reference fixtures."""
import ctypes
import os

import torch


def radialInterpolant(r, n_rings):
    """(E,R) linear-interpolation weights on the equal-area knots sqrt(q/(R-1)); exactly two
    non-zeros per row (reference transforms/fc_precomp.py:10-27).  The upper knot is the first
    knot >= r, never knot 0."""
    knots = torch.sqrt(torch.arange(n_rings, device=r.device) / (n_rings - 1))
    gap = knots[None, :] - r[:, None]
    gap = torch.where(gap < 0, torch.full_like(gap, 1e8), gap)
    hi = torch.argmin(gap, dim=1).clamp_min(1)
    lo = hi - 1
    w_hi = (r - knots[lo]) / (knots[hi] - knots[lo])
    w = torch.zeros(r.shape[0], n_rings, device=r.device, dtype=torch.float32)
    w.scatter_(1, hi[:, None], w_hi[:, None].float())
    w.scatter_(1, lo[:, None], (1 - w_hi)[:, None].float())
    return w


class FCPrecomp(object):
    """Organises per-edge log-map / transport data into the convolution stencil
    (equations (6)-(7) of the paper).

    data needs: logMag (E), logAng (E), w (N,1), supp_edges (E,2), xp (E complex).
    Returns (supp_edges, supp_sten (E',R,2B+1), ln (E'), wxp (E')) restricted to r <= epsilon.
    """

    def __init__(self, band_limit, n_rings, epsilon):
        self.B = band_limit
        self.R = n_rings
        self.max_r = epsilon
        self._memo = None           # (key of the last inputs, outputs)

    @staticmethod
    def _key(t):
        return (t.data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t._version, str(t.device), t.dtype)

    def __call__(self, data):
        r, theta, w, supp_edges, xp = data.logMag, data.logAng, data.w, data.supp_edges, data.xp
        # The reference runs this in every Net.forward (segmentation.ipynb:202).  The result is a pure function of the
        # five input tensors: when the same (unmodified) tensors come back, e.g. the same mesh in the next epoch, the
        # very same output tensors are returned, so that the support-graph cache keyed on them (graph.get_graph)
        # hits as well instead of re-sorting the edges.
        key = (self.B, self.R, float(self.max_r)) + tuple(self._key(t) for t in (r, theta, w, supp_edges, xp))
        if self._memo is not None and self._memo[0] == key:
            return self._memo[1]
        out = self._compute(r, theta, w, supp_edges, xp)
        self._memo = (key, out, (r, theta, w, supp_edges, xp))      # the inputs are kept alive: their addresses are the key
        return out

    def _compute(self, r, theta, w, supp_edges, xp):
        if (r.is_cuda and os.environ.get('FIELDCONV_TORCH_PRECOMP', '0') != '1' and r.dtype == torch.float32
                and theta.dtype == torch.float32 and w.dtype == torch.float32 and xp.dtype == torch.complex64
                and 2 <= self.R <= 8 and 2 * self.B + 1 <= 7 and r.numel() > 0):
            return self._compute_native(r, theta, w, supp_edges, xp)
        return self._compute_torch(r, theta, w, supp_edges, xp)

    def _compute_native(self, r, theta, w, supp_edges, xp):
        from .. import _lib
        lib = _lib.load()
        dev = r.device
        E, N, R, F = int(r.numel()), int(w.shape[0]), self.R, 2 * self.B + 1
        r, theta, xp = r.contiguous(), theta.contiguous(), xp.contiguous()
        wv = w.reshape(-1).contiguous()
        edges = supp_edges.to(torch.int64).contiguous()
        eps = float(self.max_r)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            nbytes = lib.fc_precomp_workspace_bytes(N, E)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.fc_precomp_mark(p(r), eps, N, E, p(ws), nbytes, st), 'fc_precomp_mark')
            off = lib.fc_precomp_kept_count_ptr(p(ws), E) - ws.data_ptr()
            kept = int(ws[off:off + 4].view(torch.int32).item())          # the one synchronisation (the reference's `nonzero`)
            edges_out = torch.empty((kept, 2), dtype=torch.int64, device=dev)
            sten = torch.empty((kept, R, F), dtype=torch.complex64, device=dev)
            ln = torch.empty(kept, dtype=torch.complex64, device=dev)
            wxp = torch.empty(kept, dtype=torch.complex64, device=dev)
            _lib.check(lib.fc_precomp_build(p(r), p(theta), p(xp), p(wv), p(edges), eps, N, E, R, F, p(edges_out), p(sten), p(ln),
                                            p(wxp), p(ws), nbytes, st), 'fc_precomp_build')
        return edges_out.to(supp_edges.dtype), sten, ln, wxp

    def _compute_torch(self, r, theta, w, supp_edges, xp):
        B, R = self.B, self.R
        r = r / self.max_r
        keep = torch.nonzero(r <= 1.0).squeeze(-1)
        r, theta, supp_edges, xp = r[keep], theta[keep], supp_edges[keep, :], xp[keep]
        ln = torch.polar(r, theta)
        ring = radialInterpolant(r, R)
        m = torch.arange(-B, B + 1, device=theta.device)
        ang = m[None, :] * theta[:, None]
        freq = torch.polar(torch.ones_like(ang), ang)
        src, dst = supp_edges[:, 0], supp_edges[:, 1]
        ws = w[src, 0]
        total = torch.zeros(w.shape[0], dtype=ws.dtype, device=ws.device).index_add(0, dst, ws)
        wxp = (ws / (1e-12 + total[dst])) * xp
        supp_sten = ring[:, :, None] * freq[:, None, :] * wxp[:, None, None]
        return supp_edges, supp_sten, ln, wxp

    def __repr__(self):
        return '{}(n_rings={}, epsilon={})'.format(self.__class__.__name__, self.R, self.max_r)
