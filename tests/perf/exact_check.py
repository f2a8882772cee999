"""usage: python tests/perf/exact_check.py <problem> [...]   (GPU box)
Exact-Jacobian kernel against the CPU twin (bit for bit) and against the FD sweep, with its host-API time."""
import numpy as np, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import np_path, twin
from opengoddard_amd import problems, _native
from opengoddard_amd.engine import HipEngine
for name in sys.argv[1:]:
    prob,obj=problems.build(name)
    eng=HipEngine(prob,obj)
    tw=twin.Twin(prob,obj,program=eng.program,header=eng.header)
    lb,ub=np_path.bounds_arrays(prob)
    x=np.clip(prob.p,lb,ub)
    F0,JE=eng.exact_stacked(x)
    t=time.time(); 
    for _ in range(5): eng.exact_stacked(x)
    dt=(time.time()-t)/5
    F0t,JT=tw.exact(x)
    F0f,JF=eng.sweep_stacked(x,_native.fd_step(x,lb,ub))
    sc=np.maximum(1.0,np.abs(JT).max(axis=0))[None,:]
    print(name,'n',eng.n,'F0 bit-equal',np.array_equal(F0,F0t),'J bit-equal',np.array_equal(JE,JT),'max|d| %.2e'%np.nanmax(np.abs(JE-JT)),'vs FD rel %.2e'%np.nanmax(np.abs(JE-JF)/sc),'%.2f ms (host API)'%(dt*1e3))
    eng.close()
