import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, time
import oracle
from deepclr_amd import synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
cfg = synthetic.model_cfg('kitti'); sd = synthetic.random_state_dict(cfg, 0)
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(sd); model = model.to('cuda:0').eval()
for n in (70000, 100000):
    x = torch.from_numpy(synthetic.make_batch('kitti', 1, n))
    try:
        with torch.no_grad():
            t0 = time.time(); y, _, _ = model(x.to('cuda:0')); torch.cuda.synchronize(); dt = time.time() - t0
        y_ref = oracle.build_oracle_model(cfg, sd)(x)
        print(n, 'ok %.1f ms' % (dt * 1e3), 'max |y - oracle| = %.2e' % float((y.cpu() - y_ref).abs().max()))
    except Exception as e:
        print(n, 'raised', type(e).__name__, str(e)[:200])
