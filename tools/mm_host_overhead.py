import time, torch
dev='cuda'
for lib in ('default','cublas','cublaslt'):
    if lib!='default':
        try: torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e: print(lib, 'ERR', e); continue
    for (M,K,N) in ((1024,1776,256),(1024,256,256),(1024,256,8)):
        a=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev); b=torch.randn(N,device=dev)
        for _ in range(20): torch.nn.functional.linear(a,w,b)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(200): torch.nn.functional.linear(a,w,b)
        t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        print(lib,(M,K,N),'host %.1f us'%((t1-t0)/200*1e6),'wall %.1f us'%((t2-t0)/200*1e6))
