import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib as L, fused_mlp
DEV="cuda:0"; lib=L.lib()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/iters*1e3
out=[]
for (P,K,N) in [(400000,128,256),(524288,64,128),(524288,64,64),(131072,128,256),(131072,128,128),(32768,128,256),(8192,512,128),(8192,128,512),(4096,1024,256),(4096,256,1024),(4096,256,256),(2048,1024,256),(1024,1024,256)]:
    X=torch.randn(P,K,device=DEV); dY=torch.randn(P,N,device=DEV); dW=torch.zeros(N,K,device=DEV)
    s=fused_mlp._s(X)
    t=timeit(lambda: lib.gb_gemm_wgrad(L.ptr(dY),L.ptr(X),None,L.ptr(dW),P,K,N,s))
    out.append("%dx%dx%d %.1fus %.1fTF"%(P,K,N,t,2.0*P*K*N/t/1e6))
print(os.environ.get("GB_WGRAD_BLOCKS","1024"), os.environ.get("GB_WGRAD_SCRATCH_MB","0"), " | ".join(out))
