import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from deepclr_amd import ops, synthetic
dev='cuda:0'
x = torch.from_numpy(synthetic.make_batch('kitti', 80, 16384)).to(dev)
for m in (2, 1024):
    ops.fps_clouds_grouped(x, m); torch.cuda.synchronize(); ts=[]
    for _ in range(8):
        s,t=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); ops.fps_clouds_grouped(x, m); t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t)*1e3)
    print('160 clouds x 16384, m=%d: median %.1f us' % (m, float(np.median(ts))))
