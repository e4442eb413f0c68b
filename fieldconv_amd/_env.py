"""Registry of the environment switches the package and its library read.

Every switch is a development aid: it selects an older kernel family for an A/B run, switches a fusion off, or
instruments a kernel.  None is needed for normal use.  The registry exists so that (a) a benchmark line can say which
switches were set when it was taken and refuse names it does not know (a typo such as FC_MFMA_MODE=f32 must not silently
benchmark the default), and (b) tests/test_host_logic.py can check that no switch is read anywhere without being listed here.
"""
import os

SWITCHES = {
    # ---- read by the binding and put into the dims of every call it makes (fc_dims::mode; fieldconv_amd.arithmetic(...) overrides it per block)
    'FC_MFMA': 'f32: fp32-MFMA contractions throughout; f16: single halves (reduced precision); default: split halves',
    # ---- read by libfieldconv_hip_dev.so ONLY (the product library reads no variable; fieldconv_amd/_lib.py loads the development build
    #      when one of these is set): once per process
    'FC_RING': '0: frequency-major forward kernels for every mesh size; 2: ring-major ones for every size',
    'FC_GROUP_SPLIT': '0: the backward data kernel never runs the two frequency groups of a tile as separate work items; 2 (development): wherever legal',
    'FC_RING_COMPACT': '0: no compact LDS plans (aliased partials, half-size record chunks) for the ring-major forward kernel',
    'FC_RING_HALVES': '0: no half tiles in the last round of the ring-major forward kernel',
    'FC_HALF_TILES': '0: no half tiles in the frequency-major kernels; 2: half tiles in the backward kernels too',
    'FC_BWD_STREAM': '0: the data / filter kernel pair on large meshes too, instead of the H-streaming arrangement (gather + stream + gx kernels)',
    'FC_FILTER2': '0: the LDS-staged half-precision filter-gradient kernel instead of the register-fed one',
    'FC_SPLIT_FINISH': '1: partial sums and parameter-gradient chain as two launches',
    'FC_EDGE_PARTS_MAX': 'cap (log2) on the number of workgroups that share a tile on small meshes',
    'FC_ECHO_WPV': 'wavefronts per vertex in the ECHO descriptor kernels (1, 2 or 4)',
    'FC_STAMP_KERNEL': 'data | filter | stream: which backward kernel writes in-kernel time stamps (tools/stamps.py)',
    'FC_DEBUG': 'forward kernels: skip phases (WRONG RESULTS; refused by bench.py)',
    'FC_DEBUG_BWD': 'backward kernels: skip phases (bit 0 walk, bit 1 gxt product, bit 2 gW product, bit 3 H stores) (WRONG RESULTS; refused by bench.py)',
    'FC_LIN_DIRECT': '0: TangentLin on small meshes through the LDS-staged kernel instead of the direct one',
    'FC_DEBUG_RG': "ECHOBlock head's grouped GEMM: 1 no products, 2 no loads, 4 no staging (WRONG RESULTS; refused by bench.py)",
    'FC_DEBUG_RP': 'finishing launch: 1 no partial loads, 2 no parameter-gradient chain (WRONG RESULTS; refused by bench.py)',
    # ---- read by the Python package
    'FIELDCONV_DEV': '1: load libfieldconv_hip_dev.so (the library switches above exist there only; one of them set without this raises)',
    'FIELDCONV_HIP_LIB': 'path of a prebuilt libfieldconv_hip.so (development variants, fieldconv_amd.build.build_variant)',
    'FIELDCONV_DENSE': '1: FCPrecomp stencils through the dense-stencil kernels',
    'FIELDCONV_NO_GEO': '1: 64-byte factored records in the forward pass instead of geometric ones',
    'FIELDCONV_TORCH_GRAPH': '1: support graph built with torch ops instead of fc_graph_build',
    'FIELDCONV_NO_EDGE_SPLIT': '1: no edge split on small meshes',
    'FIELDCONV_EAGER_STENCIL': '1: FCPrecomp returns the dense (E,R,F) tensor',
    'FIELDCONV_NO_FUSED_EPILOGUE': '1: residual add and modReLU as separate operators',
    'FIELDCONV_SEPARATE_CALLS': '1: one foreign call per kernel instead of fc_forward_params / fc_backward_all',
    'FIELDCONV_CPP_NODES': '0: the block-level autograd nodes in Python (fieldconv_amd/blocks.py) instead of the C++ ones (fc_torch_nodes.so)',
    'FIELDCONV_ECHO_TAIL': "0: ECHOBlock's dense tail as torch's own Linear / ReLU autograd nodes instead of one node; aten: one node composed "
                           "of ATen GEMMs instead of the native head (fc_echo_head_*)",
    'FIELDCONV_BLOCK_CALLS': '0: FCResNetBlock / ECHOBlock / LiftBlock composed of per-operator autograd nodes instead of the block-level entry points',
    # ---- read by bench.py only
    'BENCH_BACKEND': 'gloo: several ranks share a device (test rigs); default nccl (= RCCL)',
    'BENCH_FORCE_DIST': '1: run the partitioned / data-parallel code path with one rank',
    'BENCH_FORWARD_OVERLAP': '1: interior targets under the forward halo exchange',
    'BENCH_NO_OVERLAP': '1: gradient halo exchange not overlapped with the filter-gradient kernel',
    'BENCH_GRAPH_STEP': '1: the partitioned step captured in one HIP graph',
}

# read by the test suite only (tests/test_gpu_parity.py); known names, so that a shell that still has them set can run bench.py
TEST_SWITCHES = {
    'FC_FULL_MODES': '1: tests/test_gpu_modes.py sweeps every development switch instead of the eight kernel-family / arithmetic modes',
    'FC_FUZZ_SHAPES': 'number of shapes in the seeded sweep against the oracle (default 40)',
    'FC_FUZZ_SEED': 'seed of that sweep',
    'FC_FUZZ_WIDE': '1: the sweep also draws layers wider than 64 channels and (n_rings, band_limit) pairs outside the compiled set',
    'FC_DIST_TEST_DEVICE': 'cuda: the distributed test worker runs the HIP kernels (two gloo ranks on one GPU)',
    'FC_DIST_OVERLAP': '0: the distributed test worker without forward / backward overlap',
    'FC_DIST_PLAN_ONLY': '1: the distributed test worker checks partition and halo plan at config-4 size only',
    'FC_DIST_CONFIG4': '1: the distributed GPU test worker runs config 4\'s per-rank size (20 000 owned vertices per rank)',
}

# the switches that exist in the development build of the library only (csrc: dev_env under -DFC_DEV_SWITCHES)
LIBRARY_SWITCHES = ('FC_BWD_STREAM', 'FC_RING', 'FC_GROUP_SPLIT', 'FC_RING_COMPACT', 'FC_RING_HALVES', 'FC_HALF_TILES', 'FC_FILTER2', 'FC_SPLIT_FINISH',
                    'FC_EDGE_PARTS_MAX', 'FC_ECHO_WPV', 'FC_STAMP_KERNEL', 'FC_DEBUG', 'FC_DEBUG_BWD', 'FC_DEBUG_RP', 'FC_DEBUG_RG', 'FC_LIN_DIRECT')

PREFIXES = ('FC_', 'FIELDCONV_', 'BENCH_')
WRONG_RESULTS = ('FC_DEBUG', 'FC_DEBUG_BWD', 'FC_DEBUG_RP', 'FC_DEBUG_RG')


def active(environ=None):
    """{name: value} of the registered switches that are set."""
    environ = os.environ if environ is None else environ
    return {k: environ[k] for k in sorted(SWITCHES) if k in environ}


def unknown(environ=None):
    """Names with one of our prefixes that nothing reads (typos, switches of removed kernels)."""
    environ = os.environ if environ is None else environ
    return sorted(k for k in environ if k.startswith(PREFIXES) and k not in SWITCHES and k not in TEST_SWITCHES)
