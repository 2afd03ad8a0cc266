import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from deepclr_amd import ops, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
dev='cuda:0'
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, 0)); model = model.to(dev).eval()
head = model._merge_layers[1]
layers = head._packed_f16()
for pairs in (8, 80):
    e = torch.randn(pairs * 1024, ops.E_STRIDE, device=dev).abs()
    fn = lambda: ops.head_conv_fused_f16(e, ops.E_STRIDE, layers, pairs)
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(10):
        s,t=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); fn(); fn(); t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t)*500)
    print('head %d pairs: median %.1f us min %.1f' % (pairs, float(np.median(ts)), min(ts)))
