import torch


def degree(index, num_nodes=None, dtype=None):  # transforms/fc_precomp.py:7 imports the name only
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    return torch.zeros(n, dtype=dtype or torch.float).index_add(0, index, torch.ones_like(index, dtype=dtype or torch.float))
