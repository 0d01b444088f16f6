"""Raw host-to-device bandwidth of the box (page-locked over 1 / 2 / 4 streams, pageable): the ceiling of bench.py pcie_inclusive.
    python tools/h2d_bandwidth.py        (MI355X box of round 4: 2 streams 48.8 GB/s, 4 streams 40.8 GB/s, pageable 22.2 GB/s)"""
import torch, time
x = torch.empty(512*1024*1024, dtype=torch.uint8).pin_memory()
y = torch.empty_like(x, device="cuda")
for n in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    t=time.perf_counter()
    for rep in range(4):
        ch = x.numel()//n
        for i,s in enumerate(streams):
            with torch.cuda.stream(s):
                y[i*ch:(i+1)*ch].copy_(x[i*ch:(i+1)*ch], non_blocking=True)
    torch.cuda.synchronize()
    dt=time.perf_counter()-t
    print(n, "streams: %.1f GB/s" % (4*x.numel()/dt/1e9))
xp = torch.empty(512*1024*1024, dtype=torch.uint8)
torch.cuda.synchronize(); t=time.perf_counter()
for rep in range(4): y.copy_(xp)
torch.cuda.synchronize(); print("pageable: %.1f GB/s" % (4*x.numel()/(time.perf_counter()-t)/1e9))
