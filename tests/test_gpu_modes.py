"""The kernel families that only run under a development switch, and the RCCL code path with one rank.

The library reads its switches once per process (FC_MFMA, FC_FILTER2, ...) and the support-graph build reads
FIELDCONV_DENSE / FIELDCONV_NO_GEO, so each mode gets a FRESH child process that runs a compact parity subset of
this suite (golden FieldConv vectors, seeded oracle shapes incl. a multi-tile mesh, the FCResNetBlock fixtures).
The children are ordinary `python -m pytest` runs started with subprocess (never an exec of this process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SUBSET = ('test_fused_precomp_graph or test_fieldconv_golden or test_segmentation_net_golden or test_small_cotangent or test_fc_resnet_block_golden or N1000_k20 or N777_k12 or N500_k16 or '
          'test_factored_stencil_path_vs_oracle_and_dense or N6000_k8 or N4400_k6_I40 or N8990 or N5000_k7 or N9000_k6 or '
          'test_config2_record_kernels')

MODES = {
    'fp32_mfma': {'FC_MFMA': 'f32'},                    # v_mfma_f32_16x16x4_f32 contractions, fp32 filter-gradient kernel
    'reduced_f16': {'FC_MFMA': 'f16'},                  # single halves; the suite applies its own 5e-3 gate in this mode
    'dense_rows': {'FIELDCONV_DENSE': '1'},             # FCPrecomp stencils through the dense-stencil kernels
    'generic_records': {'FIELDCONV_NO_GEO': '1'},       # 64-byte factored records in the forward pass
    'lds_staged_filter_kernel': {'FC_FILTER2': '0'},
    'data_filter_kernel_pair_on_large_meshes': {'FC_BWD_STREAM': '0'},   # the pair instead of the gather / stream / gx arrangement
    'torch_graph_build': {'FIELDCONV_TORCH_GRAPH': '1'},
    'no_edge_split': {'FIELDCONV_NO_EDGE_SPLIT': '1'},
    'eager_stencil': {'FIELDCONV_EAGER_STENCIL': '1'},   # FCPrecomp returns the dense (E,R,F) tensor; graph built from it
    'separate_pointwise_operators': {'FIELDCONV_NO_FUSED_EPILOGUE': '1'},   # no residual / modReLU epilogue in the convolutions
    'frequency_major_forward': {'FC_RING': '0'},         # the 16-wavefront forward kernels instead of the ring-major ones
    'ring_without_half_tiles': {'FC_RING_HALVES': '0'},
    'ring_without_compact_lds_plans': {'FC_RING_COMPACT': '0'},   # 64 channels at band limit 3 back on the frequency-major forward kernel
    'frequency_groups_inside_one_work_item': {'FC_GROUP_SPLIT': '0'},   # the backward data kernel walks both groups of a tile itself
    'ring_major_any_size': {'FC_RING': '2'},             # ring-major forward kernels also on meshes of up to 4096 vertices
    'separate_finish_kernels': {'FC_SPLIT_FINISH': '1'},   # fc_backward_finish + fc_filter_param_grads instead of the fused launch
    'no_half_tiles': {'FC_HALF_TILES': '0'},
    'half_tiles_in_the_backward_pass_too': {'FC_HALF_TILES': '2'},
    'one_call_per_kernel': {'FIELDCONV_SEPARATE_CALLS': '1'},   # the per-kernel entry points instead of fc_forward_params / fc_backward_all
    'blocks_from_per_operator_nodes': {'FIELDCONV_BLOCK_CALLS': '0'},   # FCResNetBlock / ECHOBlock / LiftBlock composed of per-operator autograd nodes
    'block_nodes_in_python': {'FIELDCONV_CPP_NODES': '0'},      # the block-level nodes of fieldconv_amd/blocks.py instead of fc_torch_nodes.so
    'echo_tail_from_torch_nodes': {'FIELDCONV_ECHO_TAIL': '0'},   # ECHOBlock's dense tail as torch's own Linear / ReLU nodes
    'echo_tail_from_aten_gemms': {'FIELDCONV_ECHO_TAIL': 'aten'},   # ... as one node composed of ATen GEMMs instead of fc_echo_head_*
}


def _clean_env(extra):
    from fieldconv_amd._env import LIBRARY_SWITCHES
    env = {k: v for k, v in os.environ.items() if not (k.startswith('FC_') or k.startswith('FIELDCONV_'))}
    env.update(extra)
    if any(k in LIBRARY_SWITCHES for k in extra):
        env['FIELDCONV_DEV'] = '1'          # the library's switches exist in the development build only, and it is loaded on request only
    env.pop('PYTEST_CURRENT_TEST', None)
    return env


# The default run covers the modes that select another kernel FAMILY or another arithmetic; FC_FULL_MODES=1 sweeps every switch (each
# mode is a fresh child process of ~16 s: the full sweep is two thirds of the suite's time)
DEFAULT_MODES = ('fp32_mfma', 'reduced_f16', 'dense_rows', 'generic_records', 'data_filter_kernel_pair_on_large_meshes',
                 'frequency_major_forward', 'one_call_per_kernel', 'blocks_from_per_operator_nodes')
SWEEP = sorted(MODES) if os.environ.get('FC_FULL_MODES') == '1' else sorted(DEFAULT_MODES)


@pytest.mark.parametrize('mode', SWEEP)
def test_parity_subset_in_mode(mode):
    cmd = [sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_parity.py'),
           os.path.join(ROOT, 'tests', 'test_gpu_fullsize.py'), '-m', 'gpu', '-x', '-q', '-p', 'no:cacheprovider', '-k', SUBSET]
    res = subprocess.run(cmd, cwd=ROOT, env=_clean_env(MODES[mode]), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=1500)
    tail = '\n'.join(res.stdout.strip().splitlines()[-25:])
    assert res.returncode == 0, f'mode {mode} ({MODES[mode]}):\n{tail}'
    assert ' passed' in tail and 'failed' not in tail, tail


def test_bench_partitioned_step_as_one_hip_graph():
    """BENCH_GRAPH_STEP=1: halo exchanges, convolution kernels and the bucketed all-reduce of the partitioned step captured in
    one HIP graph (single rank over RCCL) -- the line is produced and the step is not slower than the eager one."""
    base = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '20', '--warmup', '5', '--verts', '6000', '--no-cpu-baseline',
            '--no-extras', '--cold']
    out = {}
    for flag in ('0', '1'):
        env = _clean_env({'BENCH_FORCE_DIST': '1', 'BENCH_GRAPH_STEP': flag, 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '2953' + str(5 + int(flag)),
                          'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'})
        res = subprocess.run(base, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
        out[flag] = json.loads(res.stdout.strip().splitlines()[-1])
    assert 'HIP graph' in out['1']['config']['step_launch'] and out['0']['config']['step_launch'] == 'eager'
    assert out['1']['value'] > 0.9 * out['0']['value']
    assert out['1']['roofline'] is not None


def test_bench_single_rank_over_rccl():
    """bench.py with BENCH_FORCE_DIST=1: one rank, but the partition / halo-exchange / all-reduce code of
    fieldconv_amd.dist runs over the nccl (= RCCL) backend exactly as it does with N > 1."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2', '--verts', '3000', '--no-cpu-baseline',
           '--no-extras']
    env = _clean_env({'BENCH_FORCE_DIST': '1', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29531', 'RANK': '0', 'WORLD_SIZE': '1',
                      'LOCAL_RANK': '0'})
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['steps'] == 3


@pytest.mark.parametrize('fwd_overlap,mode', [('0', 'layer'), ('1', 'layer'), ('0', 'dp'), ('0', 'net')])
def test_bench_two_ranks_on_one_gpu(fwd_overlap, mode):
    """The driver's multi-GPU command line (torch.distributed.run, --gpus 2) with both ranks on this box's one GPU: gloo
    instead of RCCL (which refuses two ranks per device), otherwise the code bench.py runs with N > 1 -- partition, halo
    plan, restricted targets on the fused-FCPrecomp graph, halo exchange forward / backward, bucketed all-reduce, the
    max-over-ranks timing -- with and without the forward overlap; and the data-parallel mode (one mesh and one replica
    of the correspondence network per rank)."""
    from conftest import free_port
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--verts', '3000', '--dp-verts', '1500', '--net-verts', '512', '--net-k', '40', '--mode', mode, '--no-cpu-baseline', '--no-extras']
    env = _clean_env({'BENCH_BACKEND': 'gloo', 'BENCH_FORWARD_OVERLAP': fwd_overlap, 'OMP_NUM_THREADS': '4'})
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-1500:]                     # rank 0 prints, nobody else
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['scaling'] == 'weak'
    if mode == 'layer':
        assert line['config']['halo_rows_rank0'] > 0 and line['config']['edges_total'] > 2 * 3000 * 25
    else:
        assert 'data-parallel x2' in line['config']['parallelism']


def test_bench_starts_its_own_ranks_from_the_plain_command_line():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the parent never touches the GPU, starts the two ranks
    through torch.distributed.run itself and relays rank 0's JSON line as its last stdout line (both ranks on this box's one
    GPU, hence gloo)."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--verts', '3000',
           '--no-cpu-baseline', '--no-extras']
    env = _clean_env({'BENCH_BACKEND': 'gloo', 'OMP_NUM_THREADS': '4'})
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['warmup'] == 1 and line['steps'] == 3
    assert line['config']['halo_rows_rank0'] > 0


def test_bench_line_is_the_literal_protocol():
    """value / ms_per_step come from the first warmup + steps of the process; the sustained-clock repeat is the `settled` extra."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '2', '--verts', '3000', '--no-cpu-baseline',
           '--no-extras']
    res = subprocess.run(cmd, cwd=ROOT, env=_clean_env({}), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line['warmup'] == 2 and line['steps'] == 4 and line['value'] > 0
    assert line['settled']['extra_untimed_steps'] >= 2 + 4 + 1 and line['settled']['value'] > 0
    assert line['roofline']['frac'] > 0 and line['roofline']['avg_launch_ms'] > 0


def test_bench_data_parallel_mode_single_rank():
    """bench.py --mode dp (BASELINE configs[4]: one mesh per rank, replicated net, one bucketed all-reduce) with a
    single rank over RCCL."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--mode', 'dp', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
           '--no-extras']
    env = _clean_env({'BENCH_FORCE_DIST': '1', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29532', 'RANK': '0', 'WORLD_SIZE': '1',
                      'LOCAL_RANK': '0'})
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and 'data-parallel' in line['config']['parallelism']
    assert line['roofline'] is not None and 0 < line['roofline']['frac'] < 1 and line['roofline']['kernel'].startswith('fc_')


def test_bench_segmentation_net_mode():
    """bench.py --mode net (BASELINE configs[2]: the segmentation network's topology on a 1 024-vertex mesh, launched eagerly through the
    block-level entry points): the line carries roofline, the replayed-graph figure, the host's enqueue time and the per-operator
    comparison, and the eager step is not host-bound by more than a third."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--mode', 'net', '--steps', '10', '--warmup', '5', '--no-cpu-baseline']
    res = subprocess.run(cmd, cwd=ROOT, env=_clean_env({}), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['config']['convs_per_step'] == 9 and 'eager' in line['config']['step_launch']
    assert line['roofline'] is not None and 0 < line['roofline']['frac'] < 1 and line['roofline']['kernel'].startswith('fc_')
    assert line['graph_replay']['ms_per_step'] > 0 and line['new_mesh_every_step']['ms_per_step'] > 0
    assert line['settled']['ms_per_step'] < 1.35 * line['graph_replay']['ms_per_step'], line['settled']
    assert line['host_enqueue_ms_per_step']['block_level_calls'] < line['host_enqueue_ms_per_step']['per_operator_calls']
    assert 'product build' in line['config']['library']
