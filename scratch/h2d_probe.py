"""Host-to-device copy rates from pinned memory on this box: one stream, several streams, sizes from one batch (4 MB) up;
host time per enqueue. Run once plain and once with HSA_ENABLE_SDMA=0 (blit kernels instead of the DMA engines)."""
import os, sys, time
import torch
dev = 'cuda:0'
print('HSA_ENABLE_SDMA =', os.environ.get('HSA_ENABLE_SDMA'), flush=True)
for mb in (4, 16, 42, 256):
    n = mb * 1024 * 1024 // 4
    host = torch.empty(n, dtype=torch.float32).pin_memory()
    host.uniform_()
    for streams in (1, 2, 4):
        ss = [torch.cuda.Stream() for _ in range(streams)]
        dst = [torch.empty(n // streams, device=dev) for _ in range(streams)]
        src = [host[i * (n // streams):(i + 1) * (n // streams)] for i in range(streams)]
        reps = 20
        for warm in (True, False):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            enq = 0.0
            for _ in range(reps):
                for s, d, h in zip(ss, dst, src):
                    with torch.cuda.stream(s):
                        e0 = time.perf_counter()
                        d.copy_(h, non_blocking=True)
                        enq += time.perf_counter() - e0
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        print('%4d MB x %d stream(s): %.1f GB/s, host enqueue %.1f us per call' % (mb, streams, reps * mb / 1024 / el * 1.048576, 1e6 * enq / (reps * streams)), flush=True)
