"""Workspace sampler alone (16384 < n <= 65536): time per launch and samples per barrier round."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepclr_amd import ops, synthetic
dev = 'cuda:0'
for pairs, n, npoint in ((8, 65536, 2), (8, 65536, 1024), (32, 65536, 1024), (80, 65536, 1024), (8, 32768, 1024)):
    x = torch.from_numpy(synthetic.make_batch('kitti', pairs, n)).to(dev)
    idx, gp, gb = ops.fps_clouds_grouped(x, npoint)[:3]
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); idx, gp, gb = ops.fps_clouds_grouped(x, npoint)[:3]; t.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(t) * 1e3)
    rounds = gb[:, 0, 6].cpu().numpy()
    print('%4d clouds x %5d pts -> %4d samples: median %8.1f us  min %8.1f us; rounds per cloud mean %.1f (%.2f samples/round), checksum %d'
          % (2 * pairs, n, npoint, float(np.median(ts)), min(ts), rounds.mean(), (npoint - 1) / max(rounds.mean(), 1e-9), int(idx.long().sum())), flush=True)
