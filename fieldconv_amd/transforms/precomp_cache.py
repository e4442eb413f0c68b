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
    blob = torch.load(path, map_location='cpu')
    if blob.get('format') != FORMAT or blob.get('version') != VERSION:
        raise ValueError(f'{path}: not a {FORMAT} v{VERSION} file')
    meta = blob['meta']
    t = {k: v.to(device) for k, v in blob['tensors'].items()}
    for name in ('rowptr_t', 'rowptr_s', 'nbr_t', 'nbr_s', 'rec_t', 'rec_s', 'factors', 'supp_edges', 'ln', 'wxp'):
        if name not in t:
            raise ValueError(f'{path}: field {name} is missing')
    E, N = int(meta['E']), int(meta['N'])
    if t['supp_edges'].shape != (E, 2) or t['rowptr_t'].numel() != N + 1 or t['factors'].shape != (E, 8):
        raise ValueError(f'{path}: tensor shapes do not match the header')
    built = {name: t.get(name) for name in _GRAPH_FIELDS}
    graph = SupportGraph.from_precomp(t['supp_edges'], N, int(meta['R']), int(meta['F']), built, built['geo_t'] is not None)
    sten = FactoredStencil(t['factors'], int(meta['R']), int(meta['F']), graph)
    register_graph(t['supp_edges'], sten, N, graph)
    return t['supp_edges'], sten, t['ln'], t['wxp']
