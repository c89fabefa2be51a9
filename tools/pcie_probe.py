import time, numpy as np, torch
n = 512**3
a = np.ones(n, dtype=np.uint32)
d = torch.empty(n, dtype=torch.int32, device="cuda")
out = np.empty(n, dtype=np.float32)
df = torch.empty(n, dtype=torch.float32, device="cuda")
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); best=1e9
    for _ in range(reps):
        t0=time.perf_counter(); fn(); torch.cuda.synchronize(); best=min(best,time.perf_counter()-t0)
    return best*1e3
ta = torch.from_numpy(a.view(np.int32)); to = torch.from_numpy(out)
print("pageable H2D 512MB: %.1f ms" % t(lambda: d.copy_(ta)))
print("pageable D2H 512MB: %.1f ms" % t(lambda: to.copy_(df)))
pa = ta.pin_memory(); po = torch.empty(n, dtype=torch.float32).pin_memory()
print("pinned H2D 512MB: %.1f ms" % t(lambda: d.copy_(pa, non_blocking=True)))
print("pinned D2H 512MB: %.1f ms" % t(lambda: po.copy_(df, non_blocking=True)))
t0=time.perf_counter(); x = torch.empty(n, dtype=torch.float32).pin_memory(); print("pin 512MB alloc: %.1f ms" % ((time.perf_counter()-t0)*1e3))
t0=time.perf_counter(); torch.cuda.cudart().cudaHostRegister(a.ctypes.data, a.nbytes, 0); print("hostRegister 512MB: %.1f ms" % ((time.perf_counter()-t0)*1e3))
ta2 = torch.from_numpy(a.view(np.int32))
print("registered H2D 512MB: %.1f ms" % t(lambda: d.copy_(ta2, non_blocking=True)))
t0=time.perf_counter(); torch.cuda.cudart().cudaHostUnregister(a.ctypes.data); print("hostUnregister: %.1f ms" % ((time.perf_counter()-t0)*1e3))
import ctypes
t0=time.perf_counter(); b = a.copy(); print("host memcpy 512MB (1 thread): %.1f ms" % ((time.perf_counter()-t0)*1e3))
