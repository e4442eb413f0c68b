"""world_size-2 (and 3) gloo tests: the vertex-partitioned path (halo exchange forward, transposed exchange backward,
filter-gradient all-reduce) against the unpartitioned single-process answer, and the data-parallel path (one mesh per
rank, one all-reduce of the parameter gradients; BASELINE config 5) against the sum over the meshes."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, free_port


@pytest.mark.parametrize('world', [2, 3, 8])
def test_partitioned_fieldconv_gloo(world):
    """world = 8: BASELINE configs[3]'s partition count -- parts with three to five neighbours and peers that exchange nothing
    (zero counts in the all-to-all-v) -- on a small mesh."""
    env = dict(os.environ, OMP_NUM_THREADS='1' if world > 4 else '2', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', '_dist_worker.py')]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count('err y=') == world, res.stdout[-3000:]


def test_halo_plan_at_config4_per_rank_size_gloo():
    """Two ranks at BASELINE configs[3]'s per-rank size (20 000 owned vertices each, k = 32, C = 48): partition, plan sizes,
    both exchanges -- what `bench.py --gpus N` sets up on every rank before its first step."""
    env = dict(os.environ, OMP_NUM_THREADS='4', MASTER_ADDR='127.0.0.1', FC_DIST_PLAN_ONLY='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', '_dist_worker.py')]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count('config-4 plan n_owned=20000') == 2, res.stdout[-3000:]


def test_data_parallel_meshes_two_ranks_gloo():
    env = dict(os.environ, OMP_NUM_THREADS='2', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', '_dp_worker.py')]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count('dp err gparams=') == 2, res.stdout[-3000:]
