"""Wire format for cached per-mesh preprocessing.

The reference caches the OFFLINE part of its pipeline per mesh (`processed/*.pt` with the fields supp_edges, xp, w,
logMag, logAng, sample_idx, reference transforms/compute_log_xport.py:36-50) and redoes the run-time part -- FCPrecomp,
and here the support-graph build -- in every forward.  This module stores what that run-time part produces, once per
(mesh, band_limit, n_rings, epsilon): the kept edges, ln, wxp, the (E,8) factor table that stands for the stencil, both
edge groupings, the ring-run offsets and the per-edge records.  Loading puts the tensors on the device and hands back
exactly what FCPrecomp.__call__ returns, (supp_edges, supp_sten, ln, wxp), with the support graph already attached to the
stencil: no kernel runs before the first convolution.

File: one `torch.save` dict {'format': 'fieldconv_amd.precomp', 'version': 1, 'meta': {...}, 'tensors': {...}};
all index tensors int32 / int64 as the kernels take them, records float32.  ~110 bytes per kept edge."""
import torch

FORMAT, VERSION = 'fieldconv_amd.precomp', 1
_GRAPH_FIELDS = ('rowptr_t', 'nbr_t', 'runs_t', 'perm_t', 'rowptr_s', 'nbr_s', 'runs_s', 'perm_s', 'rec_t', 'rec_s', 'geo_t')


def save_precomp(path, outputs, band_limit, epsilon):
    """outputs: what FCPrecomp(band_limit, n_rings, epsilon)(data) returned on the device (the fused build)."""
    from ..graph import FactoredStencil
    supp_edges, sten, ln, wxp = outputs
    if not isinstance(sten, FactoredStencil):
        raise TypeError('save_precomp stores the fused build (a FactoredStencil); FIELDCONV_EAGER_STENCIL=1 returns dense tensors')
    g = sten.graph
    tensors = {'supp_edges': supp_edges, 'ln': ln, 'wxp': wxp, 'factors': sten.factors}
    for name in _GRAPH_FIELDS:
        t = getattr(g, name)
        if t is not None:
            tensors[name] = t
    meta = {'N': g.N, 'E': g.E, 'R': g.R, 'F': g.F, 'band_limit': int(band_limit), 'epsilon': float(epsilon)}
    torch.save({'format': FORMAT, 'version': VERSION, 'meta': meta, 'tensors': {k: v.detach().cpu() for k, v in tensors.items()}}, path)


def load_precomp(path, device):
    """-> (supp_edges, supp_sten, ln, wxp) on `device`, ready for FieldConv / LiftBlock / ECHOBlock."""
    from ..graph import FactoredStencil, SupportGraph, register_graph
    blob = torch.load(path, map_location='cpu', weights_only=True)       # tensors and plain containers only: nothing is unpickled
    if blob.get('format') != FORMAT or blob.get('version') != VERSION:
        raise ValueError(f'{path}: not a {FORMAT} v{VERSION} file')
    meta = blob['meta']
    t = {k: v.to(device) for k, v in blob['tensors'].items()}
    for name in ('rowptr_t', 'rowptr_s', 'nbr_t', 'nbr_s', 'rec_t', 'rec_s', 'factors', 'supp_edges', 'ln', 'wxp'):
        if name not in t:
            raise ValueError(f'{path}: field {name} is missing')
    E, N = int(meta['E']), int(meta['N'])
    R, F = int(meta['R']), int(meta['F'])
    recf = (4 + 2 * F + 3) // 4 * 4
    if t['supp_edges'].shape != (E, 2) or t['rowptr_t'].numel() != N + 1 or t['rowptr_s'].numel() != N + 1 or t['factors'].shape != (E, 8):
        raise ValueError(f'{path}: tensor shapes do not match the header')
    # the record-streaming kernels read what these arrays say without further checks: a truncated or stale file (another band
    # limit, another padding) must fail here, not as an out-of-bounds device read
    for name in ('rec_t', 'rec_s'):
        if t[name].dim() != 2 or t[name].shape[1] != recf or t[name].shape[0] < E + 1024 // (recf * 4):
            raise ValueError(f'{path}: {name} has shape {tuple(t[name].shape)}, expected (>= E + padding, {recf}) for 2B+1 = {F}')
    if t.get('geo_t') is not None and (t['geo_t'].dim() != 2 or t['geo_t'].shape[1] != 8 or t['geo_t'].shape[0] < E + 32):
        raise ValueError(f'{path}: geo_t has shape {tuple(t["geo_t"].shape)}')
    for name in ('runs_t', 'runs_s'):
        if name not in t or tuple(t[name].shape) != (N, 8):
            raise ValueError(f'{path}: {name} must have shape ({N}, 8)')
    for name in ('nbr_t', 'nbr_s', 'perm_t', 'perm_s'):
        if name not in t or t[name].numel() != E:
            raise ValueError(f'{path}: {name} must have {E} entries')
    if E:
        ends = torch.stack((t['rowptr_t'][-1], t['rowptr_s'][-1], t['rowptr_t'][0], t['rowptr_s'][0])).tolist()
        lo = min(int(t['nbr_t'].min()), int(t['nbr_s'].min()), int(t['supp_edges'].min()))
        hi = max(int(t['nbr_t'].max()), int(t['nbr_s'].max()), int(t['supp_edges'].max()))
        if ends != [E, E, 0, 0] or lo < 0 or hi >= N:
            raise ValueError(f'{path}: row pointers or vertex ids are out of range')
    built = {name: t.get(name) for name in _GRAPH_FIELDS}
    graph = SupportGraph.from_precomp(t['supp_edges'], N, R, F, built, built['geo_t'] is not None)
    sten = FactoredStencil(t['factors'], R, F, graph)
    register_graph(t['supp_edges'], sten, N, graph)
    return t['supp_edges'], sten, t['ln'], t['wxp']
