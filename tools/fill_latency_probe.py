"""Host only: latency from gm_fill_submit to the job being retired by the native fill worker, after the worker has been
idle for a while (the cold start of a timed run), against the same draws replayed inline by the caller."""
import ctypes, os, time, torch, numpy as np, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import _lib
from generative_models_amd._lib import DrawOp
L=_lib.load()
L.gm_fill_submit.restype=ctypes.c_int64
L.gm_fill_completed.restype=ctypes.c_int64
state=torch.get_rng_state()
buf=torch.empty(4, 256*20)
ops=(DrawOp*2)()
for i in range(2):
    ops[i].kind=1; ops[i].n=256*20; ops[i].a=0; ops[i].dst=buf[i].data_ptr(); ops[i].iter_stride=0
gate=np.zeros(2,dtype=np.int64)
def once(sleep):
    time.sleep(sleep)
    t0=time.perf_counter()
    j=L.gm_fill_submit(ctypes.c_void_p(state.data_ptr()), state.numel(), ops, 2, 1, ctypes.c_void_p(gate.ctypes.data), 1)
    t1=time.perf_counter()
    while L.gm_fill_completed()<j: pass
    t2=time.perf_counter()
    return (t1-t0)*1e6,(t2-t0)*1e6
for sl in (0.0, 0.00002, 0.001, 0.01):
    r=[once(sl) for _ in range(30)]
    print(sl, "submit %.1f  done %.1f (median) max %.1f"%(np.median([a for a,b in r]), np.median([b for a,b in r]), max(b for a,b in r)))

def inline():
    t0=time.perf_counter()
    L.gm_host_replay(ctypes.c_void_p(state.data_ptr()), state.numel(), ops, 2, 1)
    return (time.perf_counter()-t0)*1e6
print("inline replay of the same two draws: %.1f us (median)" % np.median([inline() for _ in range(30)]))
