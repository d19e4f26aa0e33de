"""How many HIP streams of one process run kernels concurrently (HSA queue limit)? spin kernels on N streams."""
import os, sys, time, torch
n_list = [1, 2, 4, 6, 8, 12, 16, 24]
torch.cuda.init()
cyc = 2_000_000
for n in n_list:
    streams = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    for rep in range(2):
        t = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) * 1e3
    print("GPU_MAX_HW_QUEUES=%s streams=%d: %.2f ms" % (os.environ.get("GPU_MAX_HW_QUEUES"), n, dt), flush=True)
