"""Developer tool: device-to-host bandwidth of a 4K canvas into pinned, touched pageable and fresh pageable memory."""
import torch, time
n=8294400*3
d=torch.empty(n,dtype=torch.float64,device="cuda")
p=torch.empty(n,dtype=torch.float64).pin_memory()
h=torch.empty(n,dtype=torch.float64)
for name,t in (("pinned",p),("pageable (touched)",h)):
    for _ in range(2): t.copy_(d); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(5): t.copy_(d); torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/5
    print(name, "%.2f ms  %.1f GB/s"%(dt*1e3, n*8/dt/1e9))
t0=time.perf_counter(); f=torch.empty(n,dtype=torch.float64); f.copy_(d); torch.cuda.synchronize(); print("fresh pageable %.2f ms"%((time.perf_counter()-t0)*1e3))
