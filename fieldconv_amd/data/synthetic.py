"""Seeded synthetic support graphs with the reference's data contract.

The reference's real inputs come from an offline geometry pipeline (geodesic log maps and
parallel transport via the Vector Heat Method, reference transforms/compute_log_xport.py:21-53)
that cannot be built here; these generators produce the same per-edge / per-vertex fields
(supp_edges, logMag, logAng, xp, w) analytically so that FCPrecomp and the operators see inputs
of the right shape, ordering (grouped by source) and statistics.  CPU / numpy only (set-up code).
"""
import math

import numpy as np
import torch


class SupportData:
    """Field container matching what FCPrecomp reads (reference transforms/fc_precomp.py:61)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        return SupportData(**{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})


def random_support(N, k, seed=0):
    """G-rand: every vertex has exactly k in-neighbours drawn uniformly (worst-case locality);
    edge list grouped by source like the reference's files."""
    g = torch.Generator().manual_seed(seed)
    dst = torch.arange(N).repeat_interleave(k)
    src = torch.randint(0, N, (N * k,), generator=g)
    order = torch.argsort(src, stable=True)
    edges = torch.stack((src[order], dst[order]), 1)
    E = edges.shape[0]
    eps = 1.0
    logMag = torch.rand(E, generator=g) * eps
    logAng = (torch.rand(E, generator=g) * 2 - 1) * math.pi
    xp = torch.polar(torch.ones(E), (torch.rand(E, generator=g) * 2 - 1) * math.pi)
    w = (1.0 / N) * (1 + 0.1 * torch.rand(N, 1, generator=g))
    return SupportData(supp_edges=edges, logMag=logMag, logAng=logAng, xp=xp, w=w, epsilon=eps, num_nodes=N)


def _fibonacci_sphere(n_total, lo, hi, seed):
    """Points lo..hi-1 of an n_total-point Fibonacci lattice with a small seeded jitter; the index
    order is monotone in z, so contiguous index ranges are latitude bands."""
    i = np.arange(lo, hi, dtype=np.float64)
    golden = math.pi * (3.0 - math.sqrt(5.0))
    z = 1.0 - 2.0 * (i + 0.5) / n_total
    rad = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    lon = golden * i
    p = np.stack((rad * np.cos(lon), rad * np.sin(lon), z), 1)
    rng = np.random.default_rng(seed)
    # jitter depends on the global index only, so every rank generates identical points
    jit = np.stack([np.random.default_rng([seed, c]).standard_normal(n_total)[lo:hi] for c in range(3)], 1)
    p = p + (0.15 / math.sqrt(n_total)) * jit
    del rng
    return p / np.linalg.norm(p, axis=1, keepdims=True)


def _frames(p):
    zaxis = np.array([0.0, 0.0, 1.0])
    xaxis = np.array([1.0, 0.0, 0.0])
    ref = np.where(np.abs(p[:, 2:3]) < 0.95, zaxis[None], xaxis[None])
    e1 = np.cross(ref, p)
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.cross(p, e1)
    return e1, e2


def _edge_fields(ps, es1, es2, pt, et1, et2):
    """Geodesic log map of target t in the frame of source s, and the unit complex number that
    carries s's frame to t's frame along the great circle (closed form on the unit sphere)."""
    c = np.clip(np.sum(ps * pt, 1), -1.0, 1.0)
    dist = np.arccos(c)
    d_s = pt - c[:, None] * ps                       # direction of travel at s
    n_s = np.linalg.norm(d_s, axis=1, keepdims=True)
    d_s = np.where(n_s > 1e-12, d_s / np.maximum(n_s, 1e-12), es1)
    d_t = c[:, None] * pt - ps                       # same geodesic's direction at t
    n_t = np.linalg.norm(d_t, axis=1, keepdims=True)
    d_t = np.where(n_t > 1e-12, d_t / np.maximum(n_t, 1e-12), et1)
    ang_s = np.arctan2(np.sum(d_s * es2, 1), np.sum(d_s * es1, 1))
    ang_t = np.arctan2(np.sum(d_t * et2, 1), np.sum(d_t * et1, 1))
    same = n_s[:, 0] <= 1e-12
    ang_s = np.where(same, 0.0, ang_s)
    xp_ang = np.where(same, 0.0, ang_t - ang_s)
    return dist, ang_s, xp_ang


def _rcb_order(pts, parts):
    """Recursive coordinate bisection: a permutation that makes every part a contiguous index range of
    (almost) equal size and a compact patch (split along the longest axis of the current point set)."""
    order = np.arange(pts.shape[0])

    def split(idx, nparts):
        if nparts == 1:
            return [idx]
        ext = pts[idx].max(0) - pts[idx].min(0)
        axis = int(np.argmax(ext))
        left_parts = nparts // 2
        cut = (idx.size * left_parts) // nparts
        o = idx[np.argsort(pts[idx, axis], kind='stable')]
        return split(o[:cut], left_parts) + split(o[cut:], nparts - left_parts)
    pieces = split(order, parts)
    bounds = np.cumsum([0] + [p.size for p in pieces]).astype(np.int64)
    return np.concatenate(pieces), bounds


def sphere_partition(n_total, parts, rank, k=32, seed=0, support='all', interior_first=False):
    """G-geo: `n_total` jittered Fibonacci points on the unit sphere, k nearest neighbours
    (self included) as in-neighbours of every vertex, exact geodesic log map / transport, area
    weights.  The vertices are renumbered so that each of the `parts` compact patches (recursive
    coordinate bisection) is a contiguous range of global ids.  Returns the patch owned by `rank`,
    in LOCAL indices:

        data           SupportData over n_owned + n_halo local vertices (owned first); every edge's
                       target is owned by this rank; edges grouped by (local) source
        n_owned        vertices owned by this rank
        halo_global    (n_halo,) global ids of the remote sources, grouped by owning rank
        owner_bounds   (parts+1,) global index ranges of the patches

    support: 'all' -- the filter radius epsilon lies above every k-NN distance (no edge dropped by FCPrecomp, the outer
    rings stay empty); 'p95' -- epsilon is the 95-percentile of the k-NN distances (SURVEY 8(d) G-geo): FCPrecomp drops
    the longest 5 % of the edges and every ring of the radial interpolant is populated.

    interior_first: inside every patch the vertices whose k neighbours all lie in the patch come first;
    `data.n_interior` counts them for this rank (dist.overlap_forward convolves them while the halo rows travel).
    """
    from scipy.spatial import cKDTree
    pts = _fibonacci_sphere(n_total, 0, n_total, seed)
    if parts > 1:
        perm, bounds = _rcb_order(pts, parts)
        pts = pts[perm]
    else:
        bounds = np.array([0, n_total], dtype=np.int64)
    n_interior = None
    if interior_first and parts > 1:
        _, nbr_all = cKDTree(pts).query(pts, k=k)
        owner = np.searchsorted(bounds[1:], np.arange(n_total), side='right')
        boundary = (owner[nbr_all] != owner[:, None]).any(1)
        pts = pts[np.argsort(owner * 2 + boundary, kind='stable')]          # patches stay contiguous ranges
        n_interior = int(np.count_nonzero(~boundary[bounds[rank]:bounds[rank + 1]]))
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    e1, e2 = _frames(pts)
    # neighbours of the owned points: a tree over the points near the patch only (its bounding box grown by three expected
    # k-NN radii), checked against the distances found -- a rank of a large partitioned mesh never indexes the whole mesh
    margin = 3.0 * math.sqrt(4.0 * k / n_total)
    box_lo, box_hi = pts[lo:hi].min(0) - margin, pts[lo:hi].max(0) + margin
    cand = np.nonzero(((pts >= box_lo) & (pts <= box_hi)).all(1))[0]
    dist_k, nbr = cKDTree(pts[cand]).query(pts[lo:hi], k=k)     # (n_owned, k) positions in cand
    if parts > 1 and float(dist_k.max()) >= margin:             # (never on these samplings: a neighbour could lie outside the box)
        _, nbr = cKDTree(pts).query(pts[lo:hi], k=k)
    else:
        nbr = cand[nbr]                                         # global source ids
    n_owned = hi - lo
    dst_g = np.repeat(np.arange(lo, hi), k)
    src_g = nbr.reshape(-1)
    remote = np.unique(src_g[(src_g < lo) | (src_g >= hi)])      # sorted -> grouped by owner range
    local_of = np.full(n_total, -1, dtype=np.int64)
    local_of[lo:hi] = np.arange(n_owned)
    local_of[remote] = n_owned + np.arange(remote.size)
    src_l, dst_l = local_of[src_g], local_of[dst_g]
    dist, ang, xp_ang = _edge_fields(pts[src_g], e1[src_g], e2[src_g], pts[dst_g], e1[dst_g], e2[dst_g])
    order = np.argsort(src_l, kind='stable')
    edges = torch.from_numpy(np.stack((src_l[order], dst_l[order]), 1))
    wrng = np.random.default_rng([seed, 7]).random(n_total)
    w_all = (4 * math.pi / n_total) * (1 + 0.1 * wrng)
    w_local = np.concatenate((w_all[lo:hi], w_all[remote]))
    # filter radius: 1.5 x the expected k-NN radius sqrt(4k/n) of a uniform sphere sampling; the same
    # on every rank and comfortably above the largest k-NN distance, so no edge is dropped (E = n*k)
    eps = float(1.5 * math.sqrt(4.0 * k / n_total))
    if support == 'p95':
        # r^2 of the k nearest neighbours is uniform up to ~4k/n on a uniformly sampled unit sphere; 0.987 of that radius
        # keeps 95.0 % of the edges of the jittered Fibonacci sampling (measured at n = 20 000, k = 32).  A closed form,
        # not a quantile of this rank's edges, so that every rank of a partitioned mesh filters with the same radius.
        eps = float(0.987 * math.sqrt(4.0 * k / n_total))
    elif support != 'all':
        raise ValueError("support must be 'all' or 'p95'")
    data = SupportData(
        supp_edges=edges,
        logMag=torch.from_numpy(dist[order]).float(),
        logAng=torch.from_numpy(ang[order]).float(),
        xp=torch.polar(torch.ones(order.size), torch.from_numpy(xp_ang[order]).float()),
        w=torch.from_numpy(w_local).float()[:, None],
        epsilon=eps, num_nodes=n_owned + int(remote.size), n_interior=n_owned if n_interior is None else n_interior)
    return data, n_owned, torch.from_numpy(remote), torch.from_numpy(bounds)


def sphere_support(N, k=32, seed=0, support='all'):
    """Unpartitioned G-geo mesh (parts=1)."""
    data, _, _, _ = sphere_partition(N, 1, 0, k=k, seed=seed, support=support)
    return data
