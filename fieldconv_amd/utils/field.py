"""Zero-safe polar helpers (same names and semantics as reference utils/field.py:8-58),
written branch-free: no `nonzero` compaction, hence no host synchronisation on a GPU."""
import torch

EPS = 1e-7


def isZero(x, eps=EPS):
    return (x < eps) & (x > -eps)


def isOrigin(z, eps=EPS):
    return isZero(z.real, eps) & isZero(z.imag, eps)


def _safe(z, mask):
    # keep autograd finite at masked entries: angle/abs are evaluated at 1 there, then discarded
    return torch.where(mask, torch.ones_like(z), z)


def softAbs(z, eps=EPS):
    mask = isOrigin(z, eps)
    return torch.where(mask, torch.zeros_like(z.real), torch.abs(_safe(z, mask)))


def softAngle(z, eps=EPS):
    mask = isOrigin(z, eps)
    return torch.where(mask, torch.zeros_like(z.real), torch.angle(_safe(z, mask)))


def softAbsolute(x):
    # the reference flips negative entries in place (utils/field.py:18-26); this is the
    # out-of-place equivalent with the same values and the same (sign) gradient
    return torch.where(x < 0, -x, x)


def softSqrt(x, eps=EPS):
    mask = isZero(x, eps)
    return torch.where(mask, torch.zeros_like(x), torch.sqrt(torch.where(mask, torch.ones_like(x), x)))
