"""Library fp32 GEMM (torch.mm -> hipBLASLt/rocBLAS) timings at the GEMM shapes of the small and mid layers (reference
point for the implicit-GEMM kernels, not used by the product path; the tiny shapes are bounded by the eager launch)."""
import torch
dev="cuda"
def t(fn,n=50):
    for _ in range(5): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for (m,k,n) in [(64,5120,320),(64,320,128),(64,128,320),(64,320,320),(64,640,128),(64,128,640),(1024,4096,320),(1024,256,256),(4096,3072,256),(8192,128,3517)]:
    a=torch.randn(m,k,device=dev); b=torch.randn(k,n,device=dev); c=torch.empty(m,n,device=dev)
    us=t(lambda: torch.mm(a,b,out=c))
    # wgrad-like: [k x m] x [m x n]
    at=a.t().contiguous(); g=torch.randn(m,n,device=dev); w=torch.empty(k,n,device=dev)
    us2=t(lambda: torch.mm(at,g,out=w))
    print(f"M={m:5d} K={k:5d} N={n:5d}: mm {us:7.1f} us ({2*m*k*n/us/1e6:6.1f} TF/s)   wgrad-shaped {us2:7.1f} us")
