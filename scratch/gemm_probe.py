import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from deepclr_amd import ops, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, 0)); model = model.cuda().eval()
x = torch.from_numpy(synthetic.make_batch('kitti', 8, 16384)).cuda()
with torch.no_grad():
    f = model.cloud_feature_rows(x)
    for _ in range(int(os.environ.get('REPS', '10'))):
        y = model.merge_rows(f, 8)
torch.cuda.synchronize()
print('done')
