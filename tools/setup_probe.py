"""usage (GPU box): python tools/setup_probe.py - what the one-time set-up of the SQP leg consists of at C3: the engine,
the device Jacobian buffers, the first QP handle (loads libogsqp.so) and a second one (82 allocations, the mailboxes, a stream)."""
import time, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0=time.perf_counter()
from opengoddard_amd import problems, sqp, _sqp_native
from opengoddard_amd.engine import HipEngine
from opengoddard_amd import codegen
prob, obj = problems.build("polar_tsto")
P = codegen.trace_problem(prob, obj)
eng = HipEngine(prob, obj, program=P)
import torch; torch.cuda.synchronize()
t1=time.perf_counter()
dj = sqp.DeviceJacobian(eng); torch.cuda.synchronize()
t2=time.perf_counter()
core = _sqp_native.QpCore(eng.n, eng.m_eq, eng.m_ineq, device=eng.device); torch.cuda.synchronize()
t3=time.perf_counter()
core2 = _sqp_native.QpCore(eng.n, eng.m_eq, eng.m_ineq, device=eng.device); torch.cuda.synchronize()
t4=time.perf_counter()
print("engine %.1f ms  DeviceJacobian %.2f ms  QpCore (first: loads libogsqp) %.2f ms  QpCore (second) %.2f ms"%(1e3*(t1-t0),1e3*(t2-t1),1e3*(t3-t2),1e3*(t4-t3)))
