import torch, sys
sys.path.insert(0, '.')
from distributedpytorch_amd.ops import _ext
C = _ext.load()
dev='cuda'
def rb(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
def check(name, fn, n=6):
    outs = [fn() for _ in range(n)]
    torch.cuda.synchronize()
    bad = sum(int(not torch.equal(outs[0], o)) for o in outs[1:])
    print(f"{name:50s} nondeterministic repeats: {bad}/{n-1}", flush=True)
for (N,H,Cin,K,R,s,p) in [(8,16,64,16,1,1,0),(8,16,16,16,3,2,1),(8,16,64,64,1,2,0),(8,16,16,64,1,1,0),(8,32,8,64,7,2,3),(64,56,64,256,1,1,0),(64,56,128,128,3,2,1)]:
    Ho=(H+2*p-R)//s+1
    x=rb(N,H,H,Cin); w=rb(K,R,R,Cin)*0.1; dy=rb(N,Ho,Ho,K); wt=C.transpose_filter(w.to(torch.bfloat16)); add=rb(N,H,H,Cin)
    w=w.to(torch.bfloat16)
    check(f"fprop {N,H,Cin,K,R,s}", lambda: C.conv_fprop(x,w,None,s,p,0,False,None)[0])
    check(f"dgrad {N,H,Cin,K,R,s}", lambda: C.conv_dgrad(dy,wt,H,H,s,p,None))
    check(f"dgrad+add {N,H,Cin,K,R,s}", lambda: C.conv_dgrad(dy,wt,H,H,s,p,add))
    g=torch.rand(K,device=dev)+0.5; m=torch.randn(K,device=dev)*0.1; iv=torch.rand(K,device=dev)+0.5
    y=rb(N,Ho,Ho,K); xx=rb(N,Ho,Ho,K)
    check(f"bn_bwd dx {K}", lambda: C.bn_backward(dy,y,xx,g,m,iv,True,True,None,None)[0])
    check(f"bn_bwd dres {K}", lambda: C.bn_backward(dy,y,xx,g,m,iv,True,True,None,None)[1])
    check(f"bn_bwd dgamma {K}", lambda: C.bn_backward(dy,y,xx,g,m,iv,True,True,None,None)[2])
