import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from deepclr_amd import synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
from deepclr_amd.pipeline import PipelinedForward
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, 0)); model = model.cuda().eval()
x = torch.from_numpy(synthetic.make_batch('kitti', 8, 16384)).cuda()
r = PipelinedForward(model, depth=3)
for _ in range(3): r.prefetch(x)
for _ in range(5): y = r.step(x, [x])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): y = r.step(x, [x])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue per step %.1f us, total per step %.1f us' % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
