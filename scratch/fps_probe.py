import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from deepclr_amd import ops, synthetic
x = torch.from_numpy(synthetic.make_batch('kitti', 8, 16384)).cuda()
def t(tag, fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); print(tag, round(a.elapsed_time(b) / it * 1e3, 1), 'us', flush=True)
t('fps 16 clouds x 16384 -> 1024 ' + ('plain' if os.environ.get('DCLR_FPS_PLAIN') else 'pruned'), lambda: ops.fps_clouds(x, 1024))
t('fps 16 clouds x 16384 -> 64', lambda: ops.fps_clouds(x, 64))
t('fps 16 clouds x 16384 -> 2', lambda: ops.fps_clouds(x, 2))
xm = torch.from_numpy(synthetic.make_batch('modelnet', 8, 2048)).cuda()
t('fps 16 clouds x 2048 -> 512', lambda: ops.fps_clouds(xm, 512))
