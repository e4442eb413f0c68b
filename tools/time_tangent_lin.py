import ctypes, os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from fieldconv_amd import _lib
from fieldconv_amd.functional import _p, _stream
lib = _lib.load()
dev = torch.device('cuda:0')
for N, C in ((1024, 48), (4999, 64), (20000, 48), (32000, 64), (40000, 64), (100000, 48), (200000, 64), (500000, 32)):
    x = torch.randn(N, C, dtype=torch.cfloat, device=dev)
    re, im = torch.randn(C, C, device=dev), torch.randn(C, C, device=dev)
    y = torch.empty_like(x)
    def f():
        assert lib.fc_tangent_lin_forward(_p(x), _p(re), _p(im), _p(y), N, C, C, _stream()) == 0
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): f()
    torch.cuda.synchronize()
    print(N, C, 'tangent_lin forward: %.1f us per call (back to back)' % ((time.perf_counter() - t0) / 2000 * 1e6))
