"""Shader clock and kernel durations of the first steps of the config-2 layer after a synchronisation (development).

A one-wavefront probe kernel (compiled here into /tmp, not part of the library) writes s_memtime (shader cycles) and
s_memrealtime (100 MHz) at four points of every step: start, after the forward pass, between the two backward kernels
(SupportGraph.on_gx), end.  Differences of the 100 MHz stamps are durations, shader cycles over them the clock.

    python tools/clock_trace.py [steps]"""
import ctypes
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

SRC = r'''
#include <hip/hip_runtime.h>
__global__ void probe(unsigned long long* out, int i) {
    if (threadIdx.x == 0) { out[2 * i] = __builtin_amdgcn_s_memtime(); out[2 * i + 1] = __builtin_amdgcn_s_memrealtime(); }
}
extern "C" void clock_probe(unsigned long long* out, int i, void* stream) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out, i);
}
'''


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    __graft_entry__.build()
    os.makedirs('/tmp/clock_probe', exist_ok=True)
    open('/tmp/clock_probe/p.hip', 'w').write(SRC)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', '-o', '/tmp/clock_probe/p.so', '/tmp/clock_probe/p.hip'])
    lib = ctypes.CDLL('/tmp/clock_probe/p.so')
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.graph import get_graph
    from fieldconv_amd.nn import FieldConv
    from fieldconv_amd.transforms import FCPrecomp
    dev = torch.device('cuda:0')
    N, k, C, B, R = 20000, 32, 48, 2, 6
    data = sphere_support(N, k, support='p95').to(dev)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    graph = get_graph(edges, sten, N)
    conv = FieldConv(C, C, band_limit=B, n_rings=R).to(dev)
    x = torch.randn(N, C, dtype=torch.cfloat, device=dev).requires_grad_(True)
    gy = torch.randn(N, C, dtype=torch.cfloat, device=dev)
    params = list(conv.parameters())
    buf = torch.zeros(2 * 4 * (steps + 1), dtype=torch.int64, device=dev)
    slot = [0]

    def probe():
        lib.clock_probe(ctypes.c_void_p(buf.data_ptr()), slot[0], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        slot[0] += 1
    graph.on_gx = lambda gx: probe()

    def step():
        probe()
        y = conv(x, edges, sten)
        probe()
        torch.autograd.grad(y, [x] + params, grad_outputs=gy)
        probe()

    for warm, pause in ((5, 0.0), (5, 0.5), (200, 0.0)):
        buf.zero_()
        slot[0] = 0
        graph.on_gx = None
        for _ in range(warm):
            y = conv(x, edges, sten)
            torch.autograd.grad(y, [x] + params, grad_outputs=gy)
        graph.on_gx = lambda gx: probe()
        torch.cuda.synchronize()
        time.sleep(pause)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        st = buf.cpu().view(-1, 2)[: 4 * steps].view(steps, 4, 2).double()
        print(f'--- {warm} warm-up steps, synchronise, sleep {pause} s, {steps} steps: wall {wall:.4f} ms per step')
        print('step   fwd_us  bwd_data_us  filter+rest_us  step_us   gap_to_next_us   clock_GHz')
        for i in range(steps):
            rt = st[i, :, 1]
            us = lambda a, b: float(b - a) / 100.0          # noqa: E731
            clk = float(st[i, 3, 0] - st[i, 0, 0]) / (float(rt[3] - rt[0]) * 10.0)
            gap = us(rt[3], st[i + 1, 0, 1]) if i + 1 < steps else 0.0
            print(f'{i:4d} {us(rt[0], rt[1]):8.1f} {us(rt[1], rt[2]):10.1f} {us(rt[2], rt[3]):12.1f} {us(rt[0], rt[3]):10.1f} {gap:12.1f} {clk:12.3f}')


if __name__ == '__main__':
    main()
