"""Build libfieldconv_hip.so (gfx950 only) in-tree with hipcc.

`python -m fieldconv_amd.build` or `fieldconv_amd.build.build_native()`.  The shared object
lands in fieldconv_amd/_native/ (git-ignored, but it travels to the GPU box with the repo
snapshot) and exposes only the C ABI declared in include/fieldconv_hip.h.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
OUT_DIR = os.path.join(PKG, '_native')
LIB_PATH = os.path.join(OUT_DIR, 'libfieldconv_hip.so')
SOURCES = ['fc_api.hip', 'fc_pack.hip', 'fc_forward.hip', 'fc_forward_split.hip', 'fc_forward_ring.hip', 'fc_backward.hip', 'fc_backward_split.hip', 'fc_generic.hip', 'fc_cgemm.hip', 'fc_wide.hip', 'fc_blocks.hip', 'fc_pointwise.hip', 'fc_pointwise_f64.hip', 'fc_echo.hip', 'fc_trans_field.hip', 'fc_lift_echo_generic.hip', 'fc_head.hip', 'fc_graph.hip', 'fc_precomp.hip', 'fc_optim.hip']
HEADERS = ['fc_common.hpp', 'fc_kernels.hpp', 'fc_tile.hpp', 'fc_forward_kernels.hpp', 'fc_forward_ring.hpp', 'fc_backward_kernels.hpp', 'fc_backward_stream.hpp', os.path.join('..', '..', 'include', 'fieldconv_hip.h')]
# -fno-slp-vectorize: keep the stencil FMAs as v_fma_f32 with a direct SGPR operand; packed
# v_pk_fma_f32 needs SGPR pairs built with s_mov and saturates the CU's single scalar ALU.
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-fno-slp-vectorize',
         '-fno-gpu-rdc', '-Wno-unused-result']


def _hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: libfieldconv_hip.so cannot be built')


def _source_digest():
    """Content hash of everything the library is built from (sources, headers, this script)."""
    import hashlib
    h = hashlib.sha256()
    for d in [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]:
        with open(d, 'rb') as f:
            h.update(os.path.basename(d).encode() + b'\0' + f.read())
    return h.hexdigest()


DIGEST_PATH = LIB_PATH + '.src.sha256'
# The same sources compiled with -DFC_DEV_SWITCHES: the FC_* development variables (fieldconv_amd/_env.py: LIBRARY_SWITCHES) exist in
# this library only.  The binding loads it instead of the product library when one of them is set (fieldconv_amd/_lib.py).
DEV_LIB_PATH = os.path.join(OUT_DIR, 'libfieldconv_hip_dev.so')
DEV_DIGEST_PATH = DEV_LIB_PATH + '.src.sha256'
DEV_FLAGS = ['-DFC_DEV_SWITCHES']


def needs_build():
    """True when the library is missing or was built from different sources.  Decided on content,
    not on time stamps: a copied tree (the GPU box receives a snapshot) must not rebuild."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(DIGEST_PATH):
        return True
    with open(DIGEST_PATH) as f:
        return f.read().strip() != _source_digest()


def dev_needs_build():
    if not os.path.exists(DEV_LIB_PATH) or not os.path.exists(DEV_DIGEST_PATH):
        return True
    with open(DEV_DIGEST_PATH) as f:
        return f.read().strip() != _source_digest()


def build_dev(force=False, verbose=False):
    """libfieldconv_hip_dev.so: the product sources with the development switches compiled in (tests/test_gpu_modes.py, tools/)."""
    if not force and not dev_needs_build():
        return DEV_LIB_PATH
    path = build_native(force=True, verbose=verbose, _variant=(DEV_LIB_PATH, list(DEV_FLAGS)))
    with open(DEV_DIGEST_PATH + '.tmp.%d' % os.getpid(), 'w') as f:
        f.write(_source_digest())
    os.replace(DEV_DIGEST_PATH + '.tmp.%d' % os.getpid(), DEV_DIGEST_PATH)
    return path


# ---- C++ autograd nodes (a torch extension: host-side plumbing above the C ABI, csrc_torch/fc_torch_nodes.cpp) ----------------------------
TORCH_NODES_SRC = os.path.join(PKG, 'csrc_torch', 'fc_torch_nodes.cpp')
TORCH_NODES_PATH = os.path.join(OUT_DIR, 'fc_torch_nodes.so')
TORCH_NODES_DIGEST = TORCH_NODES_PATH + '.src.sha256'


def _torch_nodes_digest():
    import hashlib
    import torch
    h = hashlib.sha256()
    for d in (TORCH_NODES_SRC, os.path.join(PKG, '..', 'include', 'fieldconv_hip.h')):
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update(torch.__version__.encode())            # the extension is built against this torch's headers and ABI
    return h.hexdigest()


def torch_nodes_needs_build():
    if not os.path.exists(TORCH_NODES_PATH) or not os.path.exists(TORCH_NODES_DIGEST):
        return True
    with open(TORCH_NODES_DIGEST) as f:
        return f.read().strip() != _torch_nodes_digest()


def build_torch_nodes(force=False, verbose=False):
    """g++ against the installed torch's headers -> _native/fc_torch_nodes.so (module `fc_torch_nodes`).  No GPU code: the nodes call
    libfieldconv_hip.so through function pointers the binding hands them."""
    if not force and not torch_nodes_needs_build():
        return TORCH_NODES_PATH
    if under_profiler():
        raise RuntimeError('fc_torch_nodes.so is missing or stale and this process runs under a profiler: build first')
    import sysconfig
    import torch
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    tmp = TORCH_NODES_PATH + '.tmp.%d' % os.getpid()
    cmd = (['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-DTORCH_EXTENSION_NAME=fc_torch_nodes', '-DTORCH_API_INCLUDE_EXTENSION_H',
            '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1', '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI),
            '-I' + os.path.join(rocm, 'include')]
           + ['-isystem' + p for p in cpp_extension.include_paths()] + ['-isystem' + sysconfig.get_paths()['include'],
           TORCH_NODES_SRC, '-o', tmp, '-L' + tlib, '-Wl,-rpath,' + tlib, '-lc10', '-lc10_hip', '-ltorch_cpu', '-ltorch', '-ltorch_python',
           '-L' + os.path.join(rocm, 'lib'), '-lamdhip64'])
    if verbose:
        print(' '.join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError('building fc_torch_nodes.so failed:\n' + res.stdout)
    os.replace(tmp, TORCH_NODES_PATH)
    with open(TORCH_NODES_DIGEST + '.tmp.%d' % os.getpid(), 'w') as f:
        f.write(_torch_nodes_digest())
    os.replace(TORCH_NODES_DIGEST + '.tmp.%d' % os.getpid(), TORCH_NODES_DIGEST)
    return TORCH_NODES_PATH


def under_profiler():
    """rocprofv3's preloaded tool library has initialised the GPU before python starts; compiling from such a process
    means hipcc children that exec clang -- the exec hop the GPU pool forbids (it takes the machine down)."""
    env = os.environ
    return bool(env.get('ROCP_TOOL_LIBRARIES')) or 'rocprof' in env.get('LD_PRELOAD', '')


def build_variant(name, extra_flags, verbose=False):
    """Development: the same sources with extra compiler flags (-DFC_EXP_...) as _native/lib<name>.so, selected at run
    time with FIELDCONV_HIP_LIB; the product library and its digest are untouched."""
    return build_native(force=True, verbose=verbose, _variant=(os.path.join(OUT_DIR, 'lib%s.so' % name), list(extra_flags)))


def build_native(force=False, verbose=False, _variant=None):
    """Compile every HIP source for gfx950 (one hipcc per source, in parallel) and link them into one
    shared library; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    if under_profiler():
        raise RuntimeError("libfieldconv_hip.so is missing or stale and this process runs under a profiler: build first "
                           "(python3 -c 'import __graft_entry__; __graft_entry__.build()'), then profile")
    os.makedirs(OUT_DIR, exist_ok=True)
    digest = _source_digest()
    hipcc = _hipcc()
    compile_flags = [f for f in FLAGS if f != '-shared'] + (_variant[1] if _variant else [])
    lib_path = _variant[0] if _variant else LIB_PATH
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(OUT_DIR, src.replace('.hip', '.%d.o' % os.getpid()))
        cmd = [hipcc] + compile_flags + ['-c', '-o', obj, os.path.join(CSRC, src)]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    errors = []
    for src, proc in procs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            errors.append(f'{src}:\n{out}')
    tmp = lib_path + '.tmp.%d' % os.getpid()
    try:
        if errors:
            raise RuntimeError('hipcc failed:\n' + '\n'.join(errors))
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-fno-gpu-rdc', '-o', tmp] + objs
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError('hipcc link failed:\n' + res.stdout)
        os.replace(tmp, lib_path)
        if not _variant:
            with open(DIGEST_PATH + '.tmp.%d' % os.getpid(), 'w') as f:
                f.write(digest)
            os.replace(DIGEST_PATH + '.tmp.%d' % os.getpid(), DIGEST_PATH)
    finally:
        for f in objs + [tmp]:
            if os.path.exists(f):
                os.remove(f)
    return lib_path


if __name__ == '__main__':
    print(build_native(force='--force' in sys.argv, verbose=True))
    if '--dev' in sys.argv:
        print(build_dev(force='--force' in sys.argv, verbose=True))
    if '--nodes' in sys.argv:
        print(build_torch_nodes(force='--force' in sys.argv, verbose=True))
