"""Pipelined throughput of the other BASELINE configurations (diagnostic; bench.py measures configs[1] only)."""
import sys, os, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepclr_amd import synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
from deepclr_amd.pipeline import PipelinedForward

dev = 'cuda:0'
for kind, n, pairs, group, steps in (('modelnet', 2048, 256, 1, 40), ('kitti', 65536, 4, 2, 40), ('kitti', 16384, 8, 4, 200)):
    cfg = synthetic.model_cfg(kind)
    model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, seed=0))
    model = model.to(dev).eval()
    x = torch.from_numpy(synthetic.make_batch(kind, pairs, n)).to(dev)
    r = PipelinedForward(model, depth=3, ahead='knn', group=group)
    for _ in range(3 * group):
        r.prefetch(x, flush=False)
    for _ in range(5):
        y = r.step(x, [x])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = r.step(x, [x])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    with torch.no_grad():
        y_ref, _, _ = model(x)
    print(kind, n, pairs, 'pipelined %.3f ms/step -> %.0f pairs/s; equal to plain forward: %s' % (dt * 1e3, pairs / dt, torch.equal(y, y_ref)))
