"""Deterministic parameter values for network-sized fixtures.

A correspondence-net replica at 64 channels has 2.9 M parameters: storing them (and their gradients) in a fixture would
be tens of megabytes.  Instead the generator (make_golden.py, reference modules) and the GPU test (our modules) both fill
the parameters from this closed formula -- same names, same shapes by the state_dict contract -- and
the fixture keeps inputs, outputs and a fixed subsample of every gradient.  Test infrastructure only."""
import numpy as np
import torch


def fill_value(name, shape):
    """float32 array for parameter `name`: magnitudes like the reference's xavier_uniform for that shape, values a smooth
    incommensurate function of the flat index and of a per-name phase (no RNG: identical wherever it runs)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = sum((i + 1) * ord(ch) for i, ch in enumerate(name)) % 9973
    if len(shape) >= 2:
        recept = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[1] * recept, shape[0] * recept
        bound = float(np.sqrt(6.0 / (fan_in + fan_out)))
    else:
        bound = 0.1
    i = np.arange(n, dtype=np.float64)
    v = np.sin(0.7390851 * i + 0.137 * seed) * np.cos(0.0123 * i + 1.7 * seed) + 0.31 * np.sin(2.399963 * i + seed)
    return (bound * v / 1.31).astype(np.float32).reshape(shape)


def fill_params(module):
    """Overwrite every parameter of `module` (in place) with fill_value(name, shape).  Buffers are left alone."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(torch.from_numpy(fill_value(name, tuple(p.shape))).to(p.dtype))
    return module


GRAD_STRIDE = 37


def grad_sample(g):
    """What the fixture keeps of one gradient tensor: every 37th entry of the flattened tensor (all of it when it has at
    most 4096 entries), its 2-norm and its sum."""
    flat = np.asarray(g).reshape(-1)
    sub = flat if flat.size <= 4096 else flat[::GRAD_STRIDE]
    return sub.copy(), np.array([np.linalg.norm(flat.astype(np.float64)), flat.astype(np.float64).sum()])
