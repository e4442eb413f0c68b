"""Minimal stand-in for the third-party `torch_scatter` package (absent from this image).

Only used by tests/golden/make_golden.py so that the *reference* modules import in the
build container.  The reference calls scatter_add(src, index, dim=0, dim_size=N)
(nn/field_conv.py:134) and scatter_add(src, index) (transforms/fc_precomp.py:87); both are a
plain index-add along dim 0, which is what this does.
"""
import torch


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return res.index_add(0, index, src)


def scatter_min(*args, **kwargs):  # imported by utils/field.py:5, never called on the hot path
    raise NotImplementedError
