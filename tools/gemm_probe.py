import torch, time
dev='cuda'
def t(fn,n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for (M,K,N) in [(4999,156,128),(4999,160,128),(5008,156,128),(5008,160,128),(4999,128,64),(4999,64,64),(4999,64,256),(4999,256,64),(1024,1392,128),(1024,1408,128),(1024,128,64)]:
    a=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev); b=torch.randn(N,device=dev); g=torch.randn(M,N,device=dev)
    f=t(lambda: torch.addmm(b,a,w.t()))
    dg=t(lambda: g@w)
    wg=t(lambda: g.t()@a)
    print(f'M={M} K={K} N={N}: fwd {f:.1f} us  dgrad {dg:.1f}  wgrad {wg:.1f}')
