"""Support-graph preprocessing for the HIP kernels.

The operator boundary hands over `supp_edges (E,2) int64` (col 0 = source j, col 1 = target i,
in no particular target order; real data arrives grouped by source) and
`supp_sten (E,R,F) complex64` (reference nn/field_conv.py:104-121).  The kernels want the edges
grouped by target (forward) and by source (backward) with int32 indices, and stream the stencil
rows in slot order, so this module builds, once per (supp_edges, supp_sten) pair:

    by target : rowptr_t (N+1), src_t (E), sten_t = supp_sten[perm_t]
    by source : rowptr_s (N+1), dst_s (E), sten_s = supp_sten[perm_s]   (alias of supp_sten when the
                input is already grouped by source)

Every FieldConv in a network receives the same pair (reference segmentation.ipynb:205), so the
result is cached and the cost is amortised over all convolutions of a forward+backward.
Device tensors are processed by the library's own build (csrc/fc_graph.hip through fc_graph_build: one analysis
kernel, two scans, a counting sort, two placement kernels, one host synchronisation for the verdict); the torch
version below (sort / bincount / cumsum / gather, ~100 launches) is what CPU tensors take and what the tests compare
the native build against.
"""
import collections
import os

import torch

_CACHE_SIZE = 16        # meshes whose preprocessing is kept (a graph is ~100 B per edge)
_cache = collections.OrderedDict()


def factor_stencil(sten, tol=2e-6):
    """Rank-1, ring-2-sparse factorisation of a dense stencil, or None if it does not have that
    structure.  sten (E,R,F) complex64 -> records (E, RECF) float32 as documented in
    include/fieldconv_hip.h (fc_forward_factored): [q bits, w_q, w_{q+1}, 0, Re/Im ph_f ...].

    FCPrecomp builds supp_sten[e,r,f] = rSten[e,r] * fSten[e,f] * wxp[e] with exactly two adjacent
    non-zero interpolation weights (reference transforms/fc_precomp.py:24-25,95); the check below
    accepts any stencil of that shape (every row is reconstructed and compared), so hand-made dense
    stencils simply stay on the dense kernels.  One host synchronisation (the verdict)."""
    E, R, F = sten.shape
    B = (F - 1) // 2
    mag = sten.abs().amax(dim=2)                                   # (E,R)
    nz = mag > 0
    first = torch.argmax(nz.to(torch.int8), dim=1)                 # first non-zero ring, 0 if the row is empty
    q = torch.clamp(first, max=R - 2)
    ring = torch.arange(R, device=sten.device)[None, :]
    outside = nz & ((ring < q[:, None]) | (ring > q[:, None] + 1))
    rows = torch.arange(E, device=sten.device)
    s0 = sten[rows, q]                                             # (E,F)
    s1 = sten[rows, q + 1]
    ph = s0 + s1                                                   # = ph * (w_q + w_{q+1}); weights are rescaled below
    den = (ph.real ** 2 + ph.imag ** 2).sum(1)
    safe = torch.where(den > 0, den, torch.ones_like(den))
    w0 = (s0 * ph.conj()).sum(1).real / safe
    w1 = (s1 * ph.conj()).sum(1).real / safe
    w0 = torch.where(den > 0, w0, torch.zeros_like(w0))
    w1 = torch.where(den > 0, w1, torch.zeros_like(w1))
    err = torch.maximum((s0 - w0[:, None] * ph).abs().amax(1), (s1 - w1[:, None] * ph).abs().amax(1))
    scale = mag.amax(1)
    bad = outside.any() | (err > tol * scale).any()
    if bool(bad):
        return None
    recf = (4 + 2 * F + 3) // 4 * 4
    rec = torch.zeros((E, recf), dtype=torch.float32, device=sten.device)
    rec[:, 0] = q.to(torch.int32).view(torch.float32)
    rec[:, 1] = w0
    rec[:, 2] = w1
    rec[:, 4:4 + 2 * F] = torch.view_as_real(ph.contiguous()).reshape(E, 2 * F)
    return rec


def geometric_phases(rec, F, tol=2e-6):
    """Geometric-phase form of factored records, or None: ph[e,f] = c[e] * g[e]^(f-B) with |g| = 1 -- what
    FCPrecomp produces (fSten = exp(i m theta) times a per-edge weight, reference transforms/fc_precomp.py:88-95).
    rec (E, RECF) as built by factor_stencil -> (E, 8) float32 [q bits, w_q, w_{q+1}, 0, Re c, Im c, Re g, Im g].
    Every phase is reconstructed and compared; one host synchronisation (the verdict)."""
    E = rec.shape[0]
    B = (F - 1) // 2
    if B < 1:
        return None
    ph = torch.view_as_complex(rec[:, 4:4 + 2 * F].reshape(E, F, 2).contiguous())
    c = ph[:, B]
    cmag = c.abs()
    live = cmag > 0
    one = torch.ones_like(c)
    g = torch.where(live, ph[:, B + 1] / torch.where(live, c, one), one)
    scale = ph.abs().amax(1)
    err = (g.abs() - 1).abs() * cmag                       # |g| = 1 (relative to the phases' size)
    p, pc = c, c
    for m in range(1, B + 1):
        p, pc = p * g, pc * g.conj()
        err = torch.maximum(err, torch.maximum((ph[:, B + m] - p).abs(), (ph[:, B - m] - pc).abs()))
    dead = ~live & (scale > 0)                             # c = 0 but other phases are not: not geometric
    if bool((err > tol * scale).any() | dead.any()):
        return None
    geo = torch.zeros((E, 8), dtype=torch.float32, device=rec.device)
    geo[:, 0:3] = rec[:, 0:3]
    geo[:, 4] = c.real
    geo[:, 5] = c.imag
    geo[:, 6] = g.real
    geo[:, 7] = g.imag
    return geo


def _native_build(supp_edges, sten, N, R, F, want_rec, want_geo):
    """fc_graph_build (csrc/fc_graph.hip) on device tensors: both edge groupings, ring-run offsets, slot -> edge
    permutations and -- with a stencil -- the factored / geometric records, plus the verdict flags (the one host
    synchronisation).  Returns a dict of tensors and 'flags' (int)."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    dev, E = supp_edges.device, int(supp_edges.shape[0])
    edges = supp_edges.to(torch.int64).contiguous()
    recf = (4 + 2 * F + 3) // 4 * 4
    i32 = dict(dtype=torch.int32, device=dev)
    out = {}
    with torch.cuda.device(dev):
        for side in ('t', 's'):
            out['rowptr_' + side] = torch.empty(N + 1, **i32)
            out['nbr_' + side] = torch.empty(E, **i32)
            out['runs_' + side] = torch.empty((N, 8), **i32)
            out['perm_' + side] = torch.empty(E, dtype=torch.int64, device=dev)
        out['rec_t'] = out['rec_s'] = out['geo_t'] = None
        if want_rec:            # zeroed: the kernels stream up to 1 KiB past the last record
            out['rec_t'] = torch.zeros((E + 1024 // (recf * 4) + 16, recf), dtype=torch.float32, device=dev)
            out['rec_s'] = torch.zeros_like(out['rec_t'])
            if want_geo:
                out['geo_t'] = torch.zeros((E + 1024 // 32 + 16, 8), dtype=torch.float32, device=dev)
        flags = torch.empty(1, **i32)
        nbytes = lib.fc_graph_workspace_bytes(N, E, R, F, 1 if want_rec else 0)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(lib.fc_graph_build(p(edges), p(sten) if want_rec else None, N, E, R, F, p(out['rowptr_t']), p(out['nbr_t']),
                                      p(out['runs_t']), p(out['perm_t']), p(out['rowptr_s']), p(out['nbr_s']), p(out['runs_s']),
                                      p(out['perm_s']), p(out['rec_t']), p(out['rec_s']), p(out['geo_t']), p(flags), p(ws), nbytes,
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'fc_graph_build')
        out['flags'] = int(flags.item())
    if out['flags'] & 4:
        raise IndexError(f'supp_edges refers to a vertex outside [0, {N})')
    return out



class FactoredStencil:
    """FCPrecomp's `supp_sten` without the (E,R,F) tensor (reference transforms/fc_precomp.py:95).

    The fused build (fc_precomp_graph) goes from FCPrecomp's inputs straight to the support graph and the per-edge records
    the convolutions consume; what the reference returns as `supp_sten` is then only a name for that graph.  This object
    stands in for the tensor: it has its shape / dtype / device, FieldConv takes the attached SupportGraph from it,
    `sten[..., B:B+2]` (what the notebooks hand to LiftBlock, segmentation.ipynb:204) builds just those two columns, and any
    other use -- indexing, torch functions, tensor methods -- materialises the dense rows from the (E,8) factor table
    [q, w_q, w_{q+1}, 0, wxp, e^{i theta}] once and then behaves like the tensor."""

    def __init__(self, factors, R, F, graph):
        self.factors = factors
        self.R, self.F = int(R), int(F)
        self.graph = graph
        self.shape = torch.Size((int(factors.shape[0]), self.R, self.F))
        self.dtype = torch.complex64
        self.device = factors.device
        self.is_cuda = factors.is_cuda
        self.requires_grad = False
        self._dense = None
        self._lift = None

    def bound_to(self, graph):
        """The same stencil (factor table, materialised columns shared) carrying another SupportGraph -- a per-use VIEW of the
        mesh's cached graph with its own target restriction / exchange hooks (SupportGraph.view, dist/halo.py)."""
        other = FactoredStencil.__new__(FactoredStencil)
        other.__dict__.update(self.__dict__)
        other.graph = graph
        return other

    @classmethod
    def wrap(cls, supp_sten, graph):
        """A stand-in for `supp_sten` (a FactoredStencil or the dense (E,R,F) tensor) that carries `graph`: what a caller hands
        to the modules in place of the stencil when the convolution is to use a view of the mesh's graph."""
        if isinstance(supp_sten, FactoredStencil):
            return supp_sten.bound_to(graph)
        other = FactoredStencil.__new__(FactoredStencil)
        other.factors = None
        other.R, other.F = int(supp_sten.shape[1]), int(supp_sten.shape[2])
        other.graph = graph
        other.shape = supp_sten.shape
        other.dtype, other.device, other.is_cuda = supp_sten.dtype, supp_sten.device, supp_sten.is_cuda
        other.requires_grad = False
        other._dense = supp_sten
        other._lift = None
        return other

    def dim(self):
        return 3

    ndim = property(lambda self: 3)

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def numel(self):
        return self.shape[0] * self.R * self.F

    def is_complex(self):
        return True

    def _parts(self):
        fac = self.factors
        q = fac[:, 0].view(torch.int32).to(torch.int64)
        ring = torch.zeros((fac.shape[0], self.R), dtype=torch.float32, device=fac.device)
        ring.scatter_(1, q[:, None], fac[:, 1:2])
        ring.scatter_(1, q[:, None] + 1, fac[:, 2:3])
        c = torch.view_as_complex(fac[:, 4:6].contiguous())
        g = torch.view_as_complex(fac[:, 6:8].contiguous())
        return ring, c, g

    def columns(self, m_lo, m_hi):
        """(E, R, m_hi - m_lo) complex64: the stencil columns of the angular frequencies m_lo <= m < m_hi."""
        ring, c, g = self._parts()
        cols = []
        for m in range(m_lo, m_hi):
            p = c
            for _ in range(abs(m)):
                p = p * (g if m > 0 else g.conj())
            cols.append(ring.to(torch.complex64) * p[:, None])
        return torch.stack(cols, dim=2)

    def materialize(self):
        if self._dense is None:
            B = (self.F - 1) // 2
            self._dense = self.columns(-B, B + 1)
        return self._dense

    def __getitem__(self, idx):
        B = (self.F - 1) // 2
        if (self._dense is None and isinstance(idx, tuple) and len(idx) == 2 and idx[0] is Ellipsis and isinstance(idx[1], slice)
                and idx[1].step in (None, 1) and idx[1].start == B and idx[1].stop == B + 2 and B >= 1):
            if self._lift is None:          # what the notebooks hand to LiftBlock: a stand-in as well (the kernels read the factor table)
                self._lift = LiftColumns(self)
            return self._lift
        return self.materialize()[idx]

    def __len__(self):
        return self.shape[0]

    def __getattr__(self, name):            # anything else a tensor has: the materialised tensor's
        if name.startswith('__') or name in ('factors', 'graph', '_dense', '_lift'):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        unwrap = lambda a: a.materialize() if isinstance(a, FactoredStencil) else a
        args = tuple(unwrap(a) for a in args)
        kwargs = {k: unwrap(v) for k, v in kwargs.items()}
        return func(*args, **kwargs)

    def __repr__(self):
        return f'FactoredStencil(shape={tuple(self.shape)}, device={self.device}, dense={self._dense is not None})'


class LiftColumns:
    """`supp_sten[..., B:B+2]` of a FactoredStencil -- the stencil columns m = 0, 1 that the notebooks hand to LiftBlock (reference
    segmentation.ipynb:204) -- without the (E,R,2) tensor: the TransField kernels form both columns from the (E,8) factor table on the
    fly (fc_trans_field_forward with sten_stride = 0).  Shape / dtype / device of the tensor it stands for; any other use (indexing, torch
    functions, tensor methods) materialises the two columns once and then behaves like that tensor."""

    def __init__(self, stencil):
        self.factors = stencil.factors
        self._stencil = stencil
        self.shape = torch.Size((int(stencil.shape[0]), stencil.R, 2))
        self.dtype = torch.complex64
        self.device = stencil.device
        self.is_cuda = stencil.is_cuda
        self.requires_grad = False
        self._dense = None

    def dim(self):
        return 3

    ndim = property(lambda self: 3)

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def is_complex(self):
        return True

    def materialize(self):
        if self._dense is None:
            self._dense = self._stencil.columns(0, 2)
        return self._dense

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __len__(self):
        return self.shape[0]

    def __getattr__(self, name):
        if name.startswith('__') or name in ('factors', '_stencil', '_dense'):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        unwrap = lambda a: a.materialize() if isinstance(a, (LiftColumns, FactoredStencil)) else a
        return func(*tuple(unwrap(a) for a in args), **{k: unwrap(v) for k, v in kwargs.items()})

    def __repr__(self):
        return f'LiftColumns(shape={tuple(self.shape)}, device={self.device}, dense={self._dense is not None})'


def _shape_compiled(R, B):
    from . import _lib
    return bool(_lib.load().fc_shape_compiled(int(R), int(B)))


class SupportGraph:
    # `_plans`: launch plans and sizes cached per user key -- one dictionary per arithmetic mode of the binding (the packed-image
    # sizes, workspaces and kernel families of a plan follow fc_dims::mode), see the property below
    __slots__ = ('N', 'E', 'R', 'F', 'rowptr_t', 'nbr_t', 'sten_t', 'rowptr_s', 'nbr_s', 'sten_s', 'factored', 'rec_t',
                 'rec_s', 'runs_t', 'runs_s', 'geo_t', 'perm_t', 'perm_s', '_keep', '_plans_by_mode', 'on_gx', 'forward_split', 'n_targets', '_is_view')

    @property
    def _plans(self):
        from . import _lib
        return self._plans_by_mode.setdefault(_lib.current_mode(), {})

    @_plans.setter
    def _plans(self, value):
        from . import _lib
        if not hasattr(self, '_plans_by_mode') or not value:
            self._plans_by_mode = {}
        self._plans_by_mode[_lib.current_mode()] = value

    def __init__(self, supp_edges, supp_sten, N, allow_factored=True, native=None):
        if supp_edges.dim() != 2 or supp_edges.shape[1] != 2:
            raise ValueError('supp_edges must have shape (E, 2)')
        if supp_sten.dim() != 3 or supp_sten.shape[0] != supp_edges.shape[0]:
            raise ValueError('supp_sten must have shape (E, R, 2B+1) with the same E as supp_edges')
        if supp_sten.dtype not in (torch.complex64, torch.complex128):
            raise ValueError('supp_sten must be complex64 (torch.cfloat) or complex128')
        if supp_sten.dtype == torch.complex128:
            # double precision (the reference's modules run under .double()): dense stencil rows in slot order for the
            # run-time kernels (csrc/fc_generic.hip); the record-driven kernels and the native graph build are float32
            allow_factored, native = False, False
        if supp_edges.dtype not in (torch.int64, torch.int32):
            raise ValueError('supp_edges must be an integer tensor')
        dev = supp_sten.device
        E = supp_edges.shape[0]
        self.N, self.E = int(N), int(E)
        self.R, self.F = int(supp_sten.shape[1]), int(supp_sten.shape[2])
        src = supp_edges[:, 0].to(torch.int64)
        dst = supp_edges[:, 1].to(torch.int64)

        def group(key, other, minor=None):
            """CSR of `key` (stable; ties ordered by `minor` if given): rowptr, other[perm] as int32, perm."""
            if E == 0:
                z = torch.zeros(self.N + 1, dtype=torch.int32, device=dev)
                e = torch.zeros(0, dtype=torch.int32, device=dev)
                return z, e, None
            if minor is None:
                sorted_key, perm = torch.sort(key, stable=True)
            else:
                _, perm = torch.sort(key * self.R + minor, stable=True)
                sorted_key = key[perm]
            counts = torch.bincount(sorted_key, minlength=self.N)[: self.N]
            rowptr = torch.zeros(self.N + 1, dtype=torch.int32, device=dev)
            rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
            return rowptr, other[perm].to(torch.int32).contiguous(), perm

        sten = supp_sten.contiguous()
        self._keep = (supp_edges, supp_sten)      # pins the storages the cache key refers to
        self.factored = False
        self.rec_t = self.rec_s = self.sten_t = self.sten_s = self.runs_t = self.runs_s = self.geo_t = None
        self.perm_t = self.perm_s = None
        self._plans = {}           # launch plans per (in, out, band limit): functional._conv_plan
        self.on_gx = None          # optional callback(gx) between the data and filter kernels of a backward pass (dist/halo.py)
        self.forward_split = None  # optional (n_first, callback): forward launches targets [0, n_first), calls back, then the rest
        self.n_targets = self.N    # rows of the convolution's output (restrict_targets)
        self._is_view = False
        if self.R > 8 or self.F > 7 or self.F % 2 == 0 or not _shape_compiled(self.R, (self.F - 1) // 2):
            allow_factored = False      # no specialised kernels (or outside the record tables: 8 ring runs, 7 phases): dense rows, the run-time kernels
        if native is None:
            native = sten.is_cuda and os.environ.get('FIELDCONV_TORCH_GRAPH', '0') != '1'
        if native and E > 0 and 2 <= self.R <= 8 and self.F <= 7 and self.F % 2 == 1:
            self._build_native(supp_edges, sten, allow_factored)
            return
        # factored fast path: FCPrecomp's stencil is w[e,r] * ph[e,f] with two adjacent non-zero rings
        rec = None
        if allow_factored and E > 0 and self.R >= 2 and os.environ.get('FIELDCONV_DENSE', '0') != '1':
            rec = factor_stencil(sten)
        if rec is not None:
            # records grouped by vertex and, inside a vertex, sorted by ring index q: the kernels walk
            # R-1 runs with statically indexed accumulators; the other endpoint rides in the record
            q = rec[:, 0].view(torch.int32).to(torch.int64)
            pad = torch.zeros((1024 // (rec.shape[1] * 4) + 16, rec.shape[1]), dtype=rec.dtype, device=dev)
            self.rowptr_t, self.nbr_t, perm_t = group(dst, src, q)
            self.rowptr_s, self.nbr_s, perm_s = group(src, dst, q)
            self.perm_t, self.perm_s = perm_t, perm_s
            rt = rec.index_select(0, perm_t)
            rt[:, 3] = self.nbr_t.view(torch.float32)
            rs = rec.index_select(0, perm_s)
            rs[:, 3] = self.nbr_s.view(torch.float32)
            self.rec_t = torch.cat((rt, pad), 0)
            self.rec_s = torch.cat((rs, pad), 0)
            # forward pass: the shorter geometric-phase records when the phases allow it
            geo = geometric_phases(rec, self.F) if os.environ.get('FIELDCONV_NO_GEO', '0') != '1' else None
            if geo is not None:
                gt = geo.index_select(0, perm_t)
                gt[:, 3] = self.nbr_t.view(torch.float32)
                self.geo_t = torch.cat((gt, torch.zeros((1024 // 32 + 16, 8), dtype=gt.dtype, device=dev)), 0)

            def run_offsets(key):
                # runs[v, q] = number of v's slots with ring index < q  (N x 8 int32, see fc_csr::runs)
                hist = torch.bincount(key * 8 + q, minlength=self.N * 8)[: self.N * 8].reshape(self.N, 8)
                excl = torch.cumsum(hist, 1) - hist
                return excl.to(torch.int32).contiguous()
            self.runs_t = run_offsets(dst)
            self.runs_s = run_offsets(src)
            self.factored = True
        else:
            self.rowptr_t, self.nbr_t, perm_t = group(dst, src)
            self.rowptr_s, self.nbr_s, perm_s = group(src, dst)
            self.perm_t, self.perm_s = perm_t, perm_s
            if E == 0:
                self.sten_t = self.sten_s = sten
            else:
                self.sten_t = sten.index_select(0, perm_t)
                # input already grouped by source (the reference's own order): a stable sort is then the
                # identity permutation and the stencil needs no second copy (one host sync per build)
                by_source = bool((src[1:] >= src[:-1]).all()) if E > 1 else True
                self.sten_s = sten if by_source else sten.index_select(0, perm_s)

    @classmethod
    def from_precomp(cls, supp_edges, N, R, F, built, geo_ok):
        """The graph fc_precomp_graph built (transforms/fc_precomp.py): records straight from FCPrecomp's inputs."""
        g = cls.__new__(cls)
        g.N, g.E, g.R, g.F = int(N), int(supp_edges.shape[0]), int(R), int(F)
        g._keep = (supp_edges,)
        g._plans = {}
        g.on_gx = None
        g.forward_split = None
        g.n_targets = g.N
        g._is_view = False
        g.sten_t = g.sten_s = None
        for name in ('rowptr_t', 'nbr_t', 'perm_t', 'rowptr_s', 'nbr_s', 'perm_s', 'rec_t', 'rec_s', 'runs_t', 'runs_s'):
            setattr(g, name, built[name])
        g.factored = True
        g.geo_t = built['geo_t'] if geo_ok else None
        return g

    def view(self):
        """A per-use view of this graph: every array is shared, the target restriction, the exchange hooks and the launch
        plans are its own.  get_graph hands the SAME cached object to every caller of a mesh (training and evaluation, two
        partition plans): restrictions and hooks therefore go on a view, which the caller passes to the modules through
        FactoredStencil.wrap(supp_sten, view) in place of the stencil."""
        v = SupportGraph.__new__(SupportGraph)
        for name in SupportGraph.__slots__:
            if hasattr(self, name):
                setattr(v, name, getattr(self, name))
        v._plans = {}
        v.on_gx = None
        v.forward_split = None
        v.n_targets = self.N
        v._is_view = True
        return v

    def _own(self, what):
        if not getattr(self, '_is_view', False) and self in _cache.values():
            raise ValueError(f'{what} would change the graph every user of this mesh gets from the cache: take graph.view() and hand '
                             'the modules FactoredStencil.wrap(supp_sten, view)')

    def restrict_targets(self, n):
        """The vertices from `n` on are sources only (the halo of a partitioned mesh: no edge points at them): the forward
        pass then computes and returns the first n rows, the backward pass takes an (n, O) output gradient; features and
        their gradient keep all N rows.  On a view (or a graph of one's own), not on the cached graph of a mesh."""
        self._own('restrict_targets')
        n = int(n)
        if not 0 < n <= self.N:
            raise ValueError(f'n_targets must lie in (0, {self.N}]')
        if int(self.rowptr_t[n]) != self.E:
            raise ValueError(f'vertices from {n} on still have in-edges')
        self.n_targets = n
        for mode, plans in list(self._plans_by_mode.items()):
            self._plans_by_mode[mode] = {k: v for k, v in plans.items() if not (isinstance(k[0], str) and k[0] == 'rows')}
        return self

    def _build_native(self, supp_edges, sten, allow_factored):
        """csrc/fc_graph.hip: everything the torch code above does, in ~12 launches."""
        want_rec = allow_factored and os.environ.get('FIELDCONV_DENSE', '0') != '1'
        want_geo = want_rec and os.environ.get('FIELDCONV_NO_GEO', '0') != '1' and self.F >= 3
        g = _native_build(supp_edges, sten, self.N, self.R, self.F, want_rec, want_geo)
        for name in ('rowptr_t', 'nbr_t', 'perm_t', 'rowptr_s', 'nbr_s', 'perm_s'):
            setattr(self, name, g[name])
        if want_rec and not (g['flags'] & 1):
            self.factored = True
            self.rec_t, self.rec_s, self.runs_t, self.runs_s = g['rec_t'], g['rec_s'], g['runs_t'], g['runs_s']
            self.geo_t = g['geo_t'] if (want_geo and not (g['flags'] & 2)) else None
        else:                   # dense kernels: stencil rows in slot order (any order inside a vertex is fine)
            self.sten_t = sten.index_select(0, self.perm_t)
            self.sten_s = sten.index_select(0, self.perm_s)

    def check_indices(self):
        """Debug helper: host-synchronising range check of the edge list."""
        if self.E:
            lo = min(int(self.nbr_t.min()), int(self.nbr_s.min()))
            hi = max(int(self.nbr_t.max()), int(self.nbr_s.max()))
            if lo < 0 or hi >= self.N:
                raise IndexError(f'supp_edges refers to vertex {lo if lo < 0 else hi}, but x has {self.N} rows')


def _key(t):
    return (t.data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t._version, str(t.device), t.dtype)


def register_graph(supp_edges, stencil, N, graph):
    """Make a graph that was built together with its stencil stand-in (FactoredStencil) findable by get_graph /
    get_edge_csr under the tensors the modules will be called with."""
    key = (_key(supp_edges), ('factored', id(stencil)), int(N))
    _cache[key] = graph
    while len(_cache) > _CACHE_SIZE:
        _cache.popitem(last=False)


def get_graph(supp_edges, supp_sten, N):
    if isinstance(supp_sten, FactoredStencil):
        if supp_sten.graph.N != int(N):
            raise ValueError(f'x has {int(N)} rows but the stencil was built for {supp_sten.graph.N} vertices')
        return supp_sten.graph
    key = (_key(supp_edges), _key(supp_sten), int(N))
    g = _cache.get(key)
    if g is not None:
        _cache.move_to_end(key)
        return g
    g = SupportGraph(supp_edges, supp_sten, N)
    _cache[key] = g
    while len(_cache) > _CACHE_SIZE:
        _cache.popitem(last=False)
    return g


class EdgeCSR:
    """The edge list grouped by target and by source, with the permutations that bring per-edge data (ln, wxp) into
    slot order.  Used by the ECHO descriptor kernels, which see the edges but no stencil (reference nn/echo.py:94)."""
    __slots__ = ('N', 'E', 'rowptr_t', 'nbr_t', 'perm_t', 'rowptr_s', 'nbr_s', 'perm_s', '_keep', '_plans')

    def __init__(self, supp_edges, N):
        if supp_edges.dim() != 2 or supp_edges.shape[1] != 2:
            raise ValueError('supp_edges must have shape (E, 2)')
        dev = supp_edges.device
        self.N, self.E = int(N), int(supp_edges.shape[0])
        self._keep = supp_edges
        self._plans = {}           # per-mesh derived data of the users (slot-ordered ln / wxp of the ECHO kernels)
        src = supp_edges[:, 0].to(torch.int64)
        dst = supp_edges[:, 1].to(torch.int64)

        def group(key, other):
            if self.E == 0:
                return (torch.zeros(self.N + 1, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev),
                        torch.zeros(0, dtype=torch.int64, device=dev))
            sorted_key, perm = torch.sort(key, stable=True)
            counts = torch.bincount(sorted_key, minlength=self.N)[: self.N]
            rowptr = torch.zeros(self.N + 1, dtype=torch.int32, device=dev)
            rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
            return rowptr, other[perm].to(torch.int32).contiguous(), perm
        if supp_edges.is_cuda and self.E > 0 and os.environ.get('FIELDCONV_TORCH_GRAPH', '0') != '1':
            self._native(supp_edges)
            return
        self.rowptr_t, self.nbr_t, self.perm_t = group(dst, src)
        self.rowptr_s, self.nbr_s, self.perm_s = group(src, dst)

    def _native(self, supp_edges):
        """edge grouping only (fc_graph_build without a stencil)"""
        g = _native_build(supp_edges, None, self.N, 2, 1, False, False)
        for name in ('rowptr_t', 'nbr_t', 'perm_t', 'rowptr_s', 'nbr_s', 'perm_s'):
            setattr(self, name, g[name])

    @classmethod
    def from_support_graph(cls, sg, supp_edges):
        g = cls.__new__(cls)
        g.N, g.E, g._keep = sg.N, sg.E, supp_edges
        g._plans = {}
        g.rowptr_t, g.nbr_t, g.perm_t = sg.rowptr_t, sg.nbr_t, sg.perm_t
        g.rowptr_s, g.nbr_s, g.perm_s = sg.rowptr_s, sg.nbr_s, sg.perm_s
        return g


_edge_cache = collections.OrderedDict()


def get_edge_csr(supp_edges, N):
    key = (_key(supp_edges), int(N))
    g = _edge_cache.get(key)
    if g is not None:
        _edge_cache.move_to_end(key)
        return g
    # the FieldConvs of the same network already grouped these edges (any order inside a vertex serves)
    for (ek, _, n), sg in reversed(_cache.items()):
        if ek == key[0] and n == key[1] and sg.perm_t is not None:
            g = EdgeCSR.from_support_graph(sg, supp_edges)
            break
    else:
        g = EdgeCSR(supp_edges, N)
    _edge_cache[key] = g
    while len(_edge_cache) > _CACHE_SIZE:
        _edge_cache.popitem(last=False)
    return g


def clear_cache():
    _cache.clear()
    _edge_cache.clear()
