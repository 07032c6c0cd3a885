import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from simpleicp_amd import _lib
N=10_000_000; Q=1000
Xf,Xm,_=bench.synthetic_pair(N)
ctx=_lib.Context(0); ctx.upload(_lib.FIX,Xf); ctx.upload(_lib.MOV,Xm)
sel=np.unique(np.round(np.linspace(0,N-1,Q)).astype(np.int64))
nv,pl=ctx.estimate_normals(_lib.FIX,sel,10); ctx.icp_setup(sel,nv,pl)
obs=np.zeros(6)
x=obs.copy()
for i in range(12):
    t0=time.perf_counter(); R=ctx.icp_iterate(x,obs,obs,0.3,1.0); t1=time.perf_counter(); x=np.array(R.x[:]); t2=time.perf_counter()
    print(f"[py] iterate call {1e6*(t1-t0):.1f} us, np.array {1e6*(t2-t1):.1f} us", file=sys.stderr)
