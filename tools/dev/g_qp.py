import numpy as np, sys, time
sys.path.insert(0,'.')
from oracle import slsqp_np as S
from opengoddard_amd import _sqp_native as Q
rng=np.random.default_rng(0)
worst=0
for trial in range(40):
    n=int(rng.integers(3,120)); meq=int(rng.integers(0,n//2+1)); mg=int(rng.integers(0,2*n))
    Zr=rng.normal(size=(n,n))/np.sqrt(n)+np.eye(n)
    g=rng.normal(size=n); C=rng.normal(size=(meq,n)); 
    xf=rng.normal(size=n)*0.3
    c=-C@xf
    G=rng.normal(size=(mg,n)); h=-G@xf+rng.uniform(0,1,mg)*(rng.uniform(size=mg)<0.7)
    lb=np.where(rng.uniform(size=n)<0.5,xf-rng.uniform(0,0.5,n),-np.inf); ub=np.where(rng.uniform(size=n)<0.5,xf+rng.uniform(0,0.5,n),np.inf)
    d,lam,mu,mode,Zn,info=S.qp_solve(Zr,g,C,c,G,h,lb,ub)
    core=Q.QpCore(n,meq,mg)
    core.set_factor(Zr)
    A=np.vstack([C,G]); cc=np.concatenate([c,h])
    t=time.time()
    dd,mult,bm,status,iters=core.solve(A,g,cc,lb,ub)
    dt=time.time()-t
    Zg=core.get_factor()
    H=Zr@Zr.T
    e=np.abs(d-dd).max() if status==1 and mode==1 else np.nan
    em=np.abs(np.concatenate([lam,mu])-mult).max() if status==1 and mode==1 else np.nan
    print(trial,n,meq,mg,'mode',mode,status,'iters',info['ldp_iterations'],iters,'|dd|=%.1e'%e,'mult %.1e'%em,'bm %.1e'%(np.abs(bm-info.get('bound_multipliers',0)).max() if mode==1 else np.nan),'ZZ %.1e'%np.abs(Zg@Zg.T-H).max(),'%.1fms'%(dt*1e3))
    if mode==1: worst=max(worst,e)
    core.close()
print('worst',worst)
