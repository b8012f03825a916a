import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
n = 8294400 * 3
d = torch.empty(n, dtype=torch.float64, device="cuda")
p = torch.empty(n, dtype=torch.float64).pin_memory()
for _ in range(2): p.copy_(d, non_blocking=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): p.copy_(d, non_blocking=True); torch.cuda.synchronize()
print("D2H 199 MB into pinned: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
# sliced with events
s = torch.cuda.current_stream()
sl = 16 << 20
nb = n * 8
pb = p.view(torch.uint8); db = d.view(torch.uint8)
for rep in range(3):
    t0 = time.perf_counter()
    for b in range(0, nb, sl):
        pb[b:b + sl].copy_(db[b:b + sl], non_blocking=True)
    torch.cuda.synchronize()
    print("  sliced 16 MiB: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
# host memcpy pinned -> fresh pageable, 1 thread and 4 threads
import threading
for threads in (1, 4, 8):
    f = np.empty(n, dtype=np.float64)
    src = p.numpy()
    def mv(k):
        a = k * n // threads; b = (k + 1) * n // threads
        f[a:b] = src[a:b]
    t0 = time.perf_counter()
    th = [threading.Thread(target=mv, args=(k,)) for k in range(threads)]
    [t.start() for t in th]; [t.join() for t in th]
    print("host memcpy pinned -> fresh pageable, %d threads: %.2f ms" % (threads, (time.perf_counter() - t0) * 1e3))
    t0 = time.perf_counter()
    th = [threading.Thread(target=mv, args=(k,)) for k in range(threads)]
    [t.start() for t in th]; [t.join() for t in th]
    print("   ... into the same (touched) buffer: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
