import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from pyvbmc_amd import VariationalPosterior, _lib, entmc_vbmc, synthetic
ctx=_lib.Context(0); _lib.set_default_context(ctx)
for cfg,D,K,N,nsk in [(3,10,50,400,4096),(5,20,100,800,22),(5,20,100,800,256),(3,10,50,400,28),(3,10,50,400,1024)]:
    wl=synthetic.make_workload(cfg,S=1,D=D,K=K,N=N,Ns_total=nsk*K)
    vp=VariationalPosterior(wl.D,wl.K)
    vp.mu,vp.sigma,vp.lambd=wl.mu.copy(),wl.sigma.reshape(1,-1),wl.lambd.reshape(-1,1)
    vp.w,vp.eta=wl.w.reshape(1,-1),wl.eta.reshape(1,-1)
    eps=np.random.default_rng(1).standard_normal((wl.K,wl.NsK//2,wl.D))
    H,dH=entmc_vbmc(vp,wl.NsK,(True,)*4,True,eps_half=eps)
    ctx.set_timing(True); ms=[]
    for _ in range(20):
        entmc_vbmc(vp,wl.NsK,(True,)*4,True,eps_half=eps); ms.append(ctx.last_kernel_ms(0))
    ctx.set_timing(False)
    print(f"D={D} K={K} NsK={nsk}: entropy kernel alone {1e3*np.median(ms):.2f} us (min {1e3*min(ms):.2f}) plan={ctx.last_entmc_plan()}",flush=True)
