"""Workspace sampler: launch time against the number of samples (setup = sort + group build is the intercept)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepclr_amd import ops, synthetic
dev = 'cuda:0'
for pairs, n in ((8, 65536), (32, 65536), (32, 16384)):
    x = torch.from_numpy(synthetic.make_batch('kitti', pairs, n)).to(dev)
    for npoint in (2, 128, 256, 512, 1024):
        ops.fps_clouds_grouped(x, npoint); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); idx, gp, gb = ops.fps_clouds_grouped(x, npoint); t.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(t) * 1e3)
        rounds = gb[:, 0, 6].cpu().numpy().mean()
        print('%4d clouds x %5d pts -> %4d samples: median %8.1f us; rounds %.1f' % (2 * pairs, n, npoint, float(np.median(ts)), rounds), flush=True)
