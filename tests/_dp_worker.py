"""Worker for the data-parallel tests (BASELINE config 5: one mesh per rank, replicated parameters, one all-reduce of
the parameter gradients per step).  world_size-2 gloo.  On the CPU the local convolution is the oracle (a test of the
replication / all-reduce logic); with FC_DIST_TEST_DEVICE=cuda both ranks share cuda:0 and run the HIP kernels through
the FieldConv module, the oracle is only the checker."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fieldconv_amd.data import sphere_support                       # noqa: E402
from fieldconv_amd.nn import FieldConv                               # noqa: E402
from fieldconv_amd.nn.field_conv import effective_filter             # noqa: E402
from oracle import fieldconv_oracle as orc                          # noqa: E402
from oracle.torch_composites import FCPrecomp                       # noqa: E402


CH = (64, 64)        # BASELINE configs[4]: C = 64, band limit 3


def mesh(rank_like, B, R):
    """every rank has its own mesh: different size, degree and features"""
    n, k = (180, 7) if rank_like == 0 else (140, 9)
    data = sphere_support(n, k, seed=11 + rank_like)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    g = torch.Generator().manual_seed(50 + rank_like)
    C, O = CH
    x = torch.complex(torch.randn(n, C, generator=g), torch.randn(n, C, generator=g))
    gy = torch.complex(torch.randn(n, O, generator=g), torch.randn(n, O, generator=g))
    return edges, sten, x, gy


def local_grads(conv, edges, sten, x, gy, B, on_gpu, dev):
    params = [conv.zonal, conv.spherical, conv.phase]
    if on_gpu:
        y = conv(x.to(dev), edges.to(dev), sten.to(dev))
        return [t.cpu() for t in torch.autograd.grad(y, params, grad_outputs=gy.to(dev))]
    W = effective_filter(conv.zonal, conv.spherical, conv.phase, conv.ftype, B)         # torch, differentiable
    _, gW = orc.fieldconv_backward(x.numpy(), edges.numpy(), sten.numpy(), W.detach().numpy(), gy.numpy())
    return list(torch.autograd.grad(W, params, grad_outputs=torch.from_numpy(gW.astype(np.complex64))))


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    on_gpu = os.environ.get('FC_DIST_TEST_DEVICE', 'cpu') == 'cuda'
    dev = torch.device('cuda', 0) if on_gpu else torch.device('cpu')
    from fieldconv_amd.dist import GradientBuckets
    B, R = 3, 6
    torch.manual_seed(77)                                # identical replicas
    conv = FieldConv(*CH, band_limit=B, n_rings=R, ftype=1)
    if on_gpu:
        conv = conv.to(dev)
    params = [conv.zonal, conv.spherical, conv.phase]
    buckets = GradientBuckets(params, bucket_bytes=256 << 10)        # several buckets: 188 416 floats = 736 KB
    assert len(buckets.buckets) > 1 and all(p.grad.data_ptr() >= buckets.flat.data_ptr() for p in params)
    grads = local_grads(conv, *mesh(rank, B, R), B, on_gpu, dev)
    for p, g_ in zip(params, grads):
        p.grad.copy_(g_)
    packed = buckets.flat.clone()
    # the begin() / collect() path: gradients assigned by autograd (here: by hand) and packed with one multi-tensor copy
    buckets.begin()
    assert all(p.grad is None for p in params)
    grads = [g_.to(p.device) for p, g_ in zip(params, grads)]
    for p, g_ in zip(params, grads):
        p.grad = g_.clone()
    buckets.collect()
    assert torch.equal(buckets.flat, packed) and all(p.grad.data_ptr() >= buckets.flat.data_ptr() for p in params)
    buckets.begin()
    params[0].grad = grads[0].clone()                    # a parameter without a gradient reads as zero; accumulate adds
    buckets.collect()
    assert float(params[1].grad.abs().max()) == 0.0
    buckets.begin()
    for p, g_ in zip(params[1:], grads[1:]):
        p.grad = g_.clone()
    buckets.collect(accumulate=True)
    assert torch.equal(buckets.flat, packed)
    buckets.all_reduce()                                 # the one (bucketed) collective of a data-parallel step
    flat = torch.cat([p.grad.reshape(-1).cpu() for p in params])
    # single-process answer: the sum over all meshes, from the oracle
    ref_conv = FieldConv(*CH, band_limit=B, n_rings=R, ftype=1)
    ref_conv.load_state_dict({k_: v.cpu() for k_, v in conv.state_dict().items()})
    total = None
    for r in range(world):
        gr = local_grads(ref_conv, *mesh(r, B, R), B, False, torch.device('cpu'))
        fr = torch.cat([g_.reshape(-1) for g_ in gr])
        total = fr if total is None else total + fr
    err = float((flat - total).abs().max() / total.abs().max())
    print(f'rank {rank}: dp err gparams={err:.2e}', flush=True)
    assert err < (5e-3 if (on_gpu and os.environ.get('FC_MFMA') == 'f16') else 1e-5)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
