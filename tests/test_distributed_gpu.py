"""Two ranks on ONE GPU (gloo with host staging; RCCL refuses two ranks per device): the vertex-partitioned
path with the HIP kernels -- halo exchange forward, transposed exchange backward, filter-gradient
all-reduce -- against the oracle on the unpartitioned mesh.  The N > 1 runs of bench.py use the same
HaloPlan / halo_exchange code with the "nccl" backend."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, free_port

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('overlap', ['1', '0'])
def test_partitioned_fieldconv_two_ranks_one_gpu(overlap):
    """overlap=1: interior targets are convolved before the halo rows are waited for (dist.overlap_forward) and the gradient
    exchange starts between the two backward kernels (dist.overlap_backward); overlap=0: exchange, then convolve."""
    env = dict(os.environ, OMP_NUM_THREADS='4', MASTER_ADDR='127.0.0.1', FC_DIST_TEST_DEVICE='cuda', FC_DIST_OVERLAP=overlap)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', '_dist_worker.py')]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count('err y=') == 2, res.stdout[-3000:]


def test_partitioned_fieldconv_two_ranks_at_config4_per_rank_size():
    """BASELINE configs[3]'s per-rank problem -- 20 000 owned vertices per rank, k = 32, 95-percentile support, 48 channels, band limit 2
    -- on two ranks (one GPU, gloo), overlapped exchanges: sampled rows against the oracle on the union mesh, every row and the all-reduced
    filter gradient against the single-process kernels on the union mesh (tests/_dist_worker.py: check_at_config4_size)."""
    env = dict(os.environ, OMP_NUM_THREADS='4', MASTER_ADDR='127.0.0.1', FC_DIST_TEST_DEVICE='cuda', FC_DIST_OVERLAP='1', FC_DIST_CONFIG4='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', '_dist_worker.py')]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count('config-4 size n_owned=20000') == 2, res.stdout[-3000:]


def test_partitioned_fieldconv_eight_ranks_at_config4_size():
    """BASELINE configs[3] at its partition count: 160 000 vertices in EIGHT parts of 20 000 (one GPU, gloo, host-staged exchanges) through
    the HIP kernels -- parts with several neighbours, peers with zero counts, the one-hop halo (a few per cent of the owned rows), both
    exchanges overlapped, the gather / stream / gx arrangement of the backward pass on every rank.  Sampled rows against the oracle on the
    union mesh; every owned row and the all-reduced filter gradient against ONE process running the kernels on the 160 000-vertex union
    mesh (tests/_dist_worker.py: check_at_config4_size)."""
    env = dict(os.environ, OMP_NUM_THREADS='2', MASTER_ADDR='127.0.0.1', FC_DIST_TEST_DEVICE='cuda', FC_DIST_OVERLAP='1', FC_DIST_CONFIG4='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', '_dist_worker.py')]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count('config-4 size n_owned=20000') == 8, res.stdout[-3000:]
    print('\n'.join(ln for ln in res.stdout.splitlines() if 'config-4 size' in ln))


def test_data_parallel_meshes_two_ranks_one_gpu():
    """BASELINE config 5 in miniature: one mesh per rank through the FieldConv module, parameter gradients all-reduced."""
    env = dict(os.environ, OMP_NUM_THREADS='4', MASTER_ADDR='127.0.0.1', FC_DIST_TEST_DEVICE='cuda')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', '_dp_worker.py')]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count('dp err gparams=') == 2, res.stdout[-3000:]
