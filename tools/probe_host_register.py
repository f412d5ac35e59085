import time, numpy as np, torch
torch.cuda.init()
rt = torch.cuda.cudart()
for mb in (64, 256, 1024):
    a = np.ones(mb << 20, np.uint8)
    a[::4096] = 2                     # touch
    for rep in range(3):
        t0 = time.perf_counter(); rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); t1 = time.perf_counter()
        d = torch.empty(a.nbytes, dtype=torch.uint8, device='cuda')
        torch.cuda.synchronize(); t2 = time.perf_counter()
        d.copy_(torch.from_numpy(a), non_blocking=True); torch.cuda.synchronize(); t3 = time.perf_counter()
        rc2 = rt.cudaHostUnregister(a.ctypes.data); t4 = time.perf_counter()
        print('%5d MB: register %.2f ms (%.1f GB/s), copy %.2f ms (%.1f GB/s), unregister %.2f ms, rc %s %s' % (mb, (t1-t0)*1e3, a.nbytes/(t1-t0)/1e9, (t3-t2)*1e3, a.nbytes/(t3-t2)/1e9, (t4-t3)*1e3, rc, rc2))
