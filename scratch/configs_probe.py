"""Forward time of the other BASELINE configurations (diagnostic): C4 ModelNet 2048 pts x 256 pairs, C5 65536 pts x 4 pairs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deepclr_amd import ops, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model

dev = 'cuda:0'
for kind, n, pairs in (('modelnet', 2048, 256), ('kitti', 65536, 4), ('modelnet', 1024, 1)):
    cfg = synthetic.model_cfg(kind)
    model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, seed=0))
    model = model.to(dev).eval()
    x = torch.from_numpy(synthetic.make_batch(kind, pairs, n)).to(dev)
    with torch.no_grad():
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            model(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        t = bench.LaunchTimer(sample_every=1); ops.TIMER = t
        model(x)
        torch.cuda.synchronize(); ops.TIMER = None
    print(kind, n, pairs, 'forward %.2f ms -> %.0f pairs/s (no pipelining)' % (dt * 1e3, pairs / dt),
          {k: round(v['avg_us'], 1) for k, v in sorted(t.summary().items())})
