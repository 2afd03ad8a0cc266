"""kNN on feature rows alone: time per launch at the bench sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes
from deepclr_amd import ops, synthetic, lib
dev = 'cuda:0'
for kind, pairs, npoint, k in (('kitti', 32, 1024, 20), ('modelnet', 256, 512, 30), ('kitti', 8, 1024, 20)):
    rng = np.random.default_rng(0)
    rows = torch.zeros(2 * pairs * npoint, 68, device=dev)
    pts = synthetic.make_batch(kind, pairs, npoint)[:, :, :3]
    rows[:, 64:67] = torch.from_numpy(pts).reshape(-1, 3).to(dev)
    out = torch.empty(pairs, npoint, k, dtype=torch.int32, device=dev)
    call = lambda: lib.check(lib.load().dclr_knn_rows(pairs, npoint, k, rows.data_ptr(), out.data_ptr(), lib.stream_ptr()), 'knn')
    call(); torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); call(); t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t) * 1e3)
    print('%-8s %3d pairs x %4d queries, k = %d: median %7.1f us  min %7.1f us  checksum %d' % (kind, pairs, npoint, k, float(np.median(ts)), min(ts), int(out.long().sum())), flush=True)
