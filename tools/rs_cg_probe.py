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
for (P,K,N) in [(400000,64,128),(400000,128,128),(131072,128,128),(400000,128,256),(131072,128,256),(524288,64,64)]:
    X=torch.randn(P,K,device=DEV); W=torch.randn(N,K,device=DEV)/K**0.5; dY=torch.randn(P,N,device=DEV); Y=torch.empty(P,N,device=DEV); dX=torch.empty(P,K,device=DEV)
    aff=torch.cat([torch.rand(K,device=DEV)+0.5, torch.randn(K,device=DEV)*0.1]).contiguous()
    ab=torch.cat([torch.rand(K,device=DEV)+0.5, torch.randn(K,device=DEV)*0.1, torch.zeros(K,device=DEV), torch.ones(K,device=DEV)]).contiguous()
    st=torch.zeros(32*2*N,dtype=torch.float64,device=DEV); dst=torch.zeros(33*2*K,dtype=torch.float64,device=DEV)
    s=fused_mlp._s(X)
    t1=timeit(lambda: lib.gb_gemm_fwd(L.ptr(X),L.ptr(W),L.ptr(aff),L.ptr(Y),L.ptr(st),32,P,K,N,None,s))
    t2=timeit(lambda: lib.gb_gemm_dgrad(L.ptr(dY),L.ptr(W),L.ptr(dX),L.ptr(X),L.ptr(ab),L.ptr(dst),32,P,K,N,None,None,None,s))
    t3=timeit(lambda: lib.gb_gemm_fwd(L.ptr(X),L.ptr(W),None,L.ptr(Y),None,1,P,K,N,None,s))
    fl=2.0*P*K*N
    out.append("%dx%dx%d fwd+st %.0fus %.0fTF dgrad+bn %.0fus %.0fTF fwd %.0fus %.0fTF"%(P,K,N,t1,fl/t1/1e6,t2,fl/t2/1e6,t3,fl/t3/1e6))
print("CG="+os.environ.get("GB_RS_CG","2")); print("\n".join(out))
