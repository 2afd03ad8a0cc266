"""Host time of the group-boundary step of the pipelined runner, split: dense call (merge_rows) / sampling chain (_launch) / rest,
and the pieces of the sampling chain (c2 shapes, steady state, no timers attached)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepclr_amd import synthetic, ops
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
from deepclr_amd.pipeline import PipelinedForward
dev = torch.device('cuda', 0)
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, seed=0)); model = model.to(dev).eval()
x = torch.from_numpy(synthetic.make_batch('kitti', 8, 16384)).to(dev)
r = PipelinedForward(model, depth=3, ahead='knn', group=10, dense_group=True, inputs_ready=True)
acc = {}
def timed(obj, name, label):
    fn = getattr(obj, name)
    def wrap(*a, **k):
        t = time.perf_counter(); out = fn(*a, **k); acc.setdefault(label, []).append(time.perf_counter() - t); return out
    setattr(obj, name, wrap)
timed(model, 'merge_rows', 'dense call (merge_rows)')
timed(r, '_launch', 'sampling chain (_launch)')
timed(model, 'sample', '  chain: sample')
timed(model, 'cloud_feature_rows', '  chain: cloud_feature_rows')
timed(model, 'merge_prep', '  chain: merge_prep')
_cat = torch.cat
def cat(*a, **k):
    t = time.perf_counter(); out = _cat(*a, **k); acc.setdefault('  chain: torch.cat', []).append(time.perf_counter() - t); return out
torch.cat = cat
for _ in range(30): r.prefetch(x, flush=False)
for _ in range(40): r.step(x, upcoming=[x])
torch.cuda.synchronize(); acc.clear()
steps = []
for _ in range(100):
    t = time.perf_counter(); r.step(x, upcoming=[x]); steps.append(time.perf_counter() - t)
torch.cuda.synchronize()
steps.sort()
print('step host time: median slice step %.1f us, group-boundary steps %.0f us (mean of the 10 longest)' % (1e6 * steps[50], 1e6 * sum(steps[-10:]) / 10))
for k, v in acc.items():
    print('%-32s %3d calls, mean %.1f us' % (k, len(v), 1e6 * sum(v) / len(v)))
