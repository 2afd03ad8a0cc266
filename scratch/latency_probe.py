"""Single-pair latency (what the reference's scripts/timing.py measures): helper.predict(source, template)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deepclr_amd import ops, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model, ModelInferenceHelper

dev = 'cuda:0'
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, seed=0))
model = model.to(dev).eval()
for seq in (False, True):
    helper = ModelInferenceHelper(model, is_sequential=seq)
    x = torch.from_numpy(synthetic.make_batch('kitti', 1, 16384)).to(dev)
    tm = []
    for i in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if seq:
            if not helper.has_state():
                helper.predict(x[0])
            helper.predict(x[1])
        else:
            helper.predict(x[1], x[0])
        b.record(); torch.cuda.synchronize()
        tm.append(a.elapsed_time(b))
    tm = sorted(tm[5:])
    print('sequential' if seq else 'pairwise', 'median ms', tm[len(tm)//2], 'min', tm[0])
t = bench.LaunchTimer(sample_every=1); ops.TIMER = t
helper = ModelInferenceHelper(model, is_sequential=False)
for i in range(5):
    helper.predict(x[1], x[0])
torch.cuda.synchronize(); ops.TIMER = None
for k, v in sorted(t.summary().items()):
    print(k, round(v['avg_us'], 1), v['launches'])
