"""Host time of one sampling chain + one dense call of the pipelined runner (c2 shapes)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepclr_amd import synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
from deepclr_amd.pipeline import PipelinedForward
dev = torch.device('cuda', 0)
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, seed=0)); model = model.to(dev).eval()
x = torch.from_numpy(synthetic.make_batch('kitti', 8, 16384)).to(dev)
r = PipelinedForward(model, depth=3, ahead='knn', group=10, dense_group=True, inputs_ready=True)
for _ in range(30): r.prefetch(x, flush=False)
for _ in range(40): r.step(x, upcoming=[x])
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(100): r.step(x, upcoming=[x])
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host time per step: %.1f us (100 steps = 10 sampling chains + 10 dense calls)' % ((t1 - t0) * 1e4))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
