"""Run-time stencil assembly (same call contract as reference transforms/fc_precomp.py:30-97): three launches of
csrc/fc_precomp.hip (fc_precomp_mark / fc_precomp_build).  Device tensors only, like every operator of this package."""
import ctypes

import torch


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
        if not r.is_cuda:
            raise RuntimeError(f'FCPrecomp: fieldconv_amd runs on a ROCm device only (got {r.device} tensors); there is no '
                               'CPU fallback')
        if (r.dtype != torch.float32 or theta.dtype != torch.float32 or w.dtype != torch.float32
                or xp.dtype != torch.complex64):
            raise ValueError('FCPrecomp expects float32 logMag / logAng / w and complex64 xp')
        if not (2 <= self.R <= 8 and 1 <= 2 * self.B + 1 <= 7):
            raise ValueError('FCPrecomp supports n_rings 2..8 and band limits 0..3')
        if r.numel() == 0:
            F = 2 * self.B + 1
            return (supp_edges[:0], torch.zeros((0, self.R, F), dtype=torch.complex64, device=r.device),
                    torch.zeros(0, dtype=torch.complex64, device=r.device), torch.zeros(0, dtype=torch.complex64, device=r.device))
        return self._compute_native(r, theta, w, supp_edges, xp)

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

    def __repr__(self):
        return '{}(n_rings={}, epsilon={})'.format(self.__class__.__name__, self.R, self.max_r)
