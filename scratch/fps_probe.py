"""Sampler alone: time per launch and samples per barrier round (group_box[:, 0, 6]) on the bench clouds.
Modes through the environment: default = per-group candidates, DCLR_FPS_WAVECAND=1, DCLR_FPS_SINGLE=1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepclr_amd import ops, synthetic
dev = 'cuda:0'
for kind, pairs, n, npoint in (('kitti', 32, 16384, 1024), ('kitti', 8, 16384, 1024), ('kitti', 1, 16384, 1024), ('modelnet', 256, 2048, 512), ('kitti', 8, 4096, 1024)):
    x = torch.from_numpy(synthetic.make_batch(kind, pairs, n)).to(dev)
    idx, gp, gb = ops.fps_clouds_grouped(x, npoint)[:3]
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); idx, gp, gb = ops.fps_clouds_grouped(x, npoint)[:3]; t.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(t) * 1e3)
    rounds = gb[:, 0, 6].cpu().numpy()
    print('%-8s %4d clouds x %5d pts -> %4d samples: median %8.1f us  min %8.1f us; barrier rounds per cloud mean %.1f (%.2f samples/round), checksum %d'
          % (kind, 2 * pairs, n, npoint, float(np.median(ts)), min(ts), rounds.mean(), (npoint - 1) / max(rounds.mean(), 1e-9), int(idx.long().sum())), flush=True)
