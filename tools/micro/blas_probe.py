"""What the host BLAS behind numpy / torch is and how its sgemm rate moves with the thread count (the CPU baseline picks the best)."""
import time, os, numpy as np, torch
from threadpoolctl import threadpool_info, threadpool_limits
print([ (d.get('internal_api'), d.get('num_threads'), d.get('filepath','')[-40:]) for d in threadpool_info()])
print(torch.__config__.parallel_info()[:300])
Q=np.random.default_rng(0).standard_normal((2048,2048),dtype=np.float32); D=np.random.default_rng(1).standard_normal((16384,2048),dtype=np.float32)
def t(fn,n=3):
    fn(); t0=time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter()-t0)/n
fl=2*2048*16384*2048
for nt in (32,64,128,256):
    with threadpool_limits(limits=nt):
        dt=t(lambda: Q@D.T)
    print('numpy threads',nt, round(fl/dt/1e9),'GFLOP/s')
Qt,Dt=torch.from_numpy(Q),torch.from_numpy(D)
for nt in (32,64,128,256):
    torch.set_num_threads(nt)
    dt=t(lambda: torch.mm(Qt,Dt.T))
    print('torch threads',nt, round(fl/dt/1e9),'GFLOP/s')
