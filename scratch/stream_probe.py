"""How many sampler launches (16 clouds x 16384 points = 16 workgroups on 16 CUs, ~0.7 ms) run side by side on N HIP streams?
   GPU_MAX_HW_QUEUES=<q> python scratch/stream_probe.py [prio]   (prio: the launching thread's current stream is a
   high-priority one, as in bench.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepclr_amd import ops, synthetic
dev = 'cuda:0'
if len(sys.argv) > 1 and sys.argv[1] == 'prio':
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
x = torch.from_numpy(synthetic.make_batch('kitti', 8, 16384)).to(dev)
ops.fps_clouds_grouped(x, 1024); torch.cuda.synchronize()
t0 = time.perf_counter(); ops.fps_clouds_grouped(x, 1024); torch.cuda.synchronize(); single = time.perf_counter() - t0
print('queues=%s single launch %.0f us' % (os.environ.get('GPU_MAX_HW_QUEUES', 'default'), single * 1e6))
for n in (1, 2, 3, 4, 6, 8):
    streams = [torch.cuda.Stream() for _ in range(n)]
    for s in streams:                      # every stream's allocator pool holds the launch's buffers before the clock starts
        with torch.cuda.stream(s):
            keep = [ops.fps_clouds_grouped(x, 1024) for _ in range(6)]
    del keep
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(6):
        for s in streams:
            with torch.cuda.stream(s):
                ops.fps_clouds_grouped(x, 1024)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print('  %d streams x 6 launches: %.0f us -> %.2f launches in flight' % (n, el * 1e6, n * 6 * single / el))
