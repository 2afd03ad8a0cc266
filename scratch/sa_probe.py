import os, sys, torch, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from deepclr_amd import ops, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, 0)); model = model.cuda().eval()
x = torch.from_numpy(synthetic.make_batch('kitti', 8, 16384)).cuda()
sam = model._cloud_layers[0]._sa0
fps = ops.fps_clouds(x, 1024)
mlps = sam.packed_mlps()
def t(tag):
    for _ in range(3): ops.sa_msg_fused(x, fps, sam.radii, sam.nsamples, mlps)
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.sa_msg_fused(x, fps, sam.radii, sam.nsamples, mlps)
    b.record(); torch.cuda.synchronize(); print(tag, a.elapsed_time(b) / 10 * 1e3, 'us')
for dbg in ('0',):
    os.environ['DCLR_SA_DEBUG'] = dbg
    t('debug=' + dbg)
_, counts = ops.sa_msg_fused(x, fps, sam.radii, sam.nsamples, mlps, want_counts=True)
os.environ['DCLR_SA_DEBUG'] = '0'
_, counts = ops.sa_msg_fused(x, fps, sam.radii, sam.nsamples, mlps, want_counts=True)
print('counts mean', counts.float().mean(dim=(0,1)), 'max', counts.amax(dim=(0,1)))
