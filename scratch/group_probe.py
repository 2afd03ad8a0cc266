import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from deepclr_amd import ops, synthetic
x = torch.from_numpy(synthetic.make_batch('kitti', 8, 16384)).cuda()
idx, gpts, gbox = ops.fps_clouds_grouped(x, 1024)
cent = torch.gather(x[:, :, :3], 1, idx.long()[:, :, None].expand(-1, -1, 3))      # (16,1024,3)
lo, hi = gbox[:, None, :, 0:3], gbox[:, None, :, 3:6]                               # (16,1,64,3)
c = cent[:, :, None, :]
d = torch.clamp(torch.maximum(lo - c, c - hi), min=0)
lb = (d * d).sum(-1)                                                                # (16,1024,64)
for r in (0.5, 1.0):
    n = (lb < r * r).sum(-1).float()
    print('radius', r, 'groups reached per centroid: mean %.2f  median %.0f  max %.0f' % (n.mean(), n.median(), n.max()))
ext = (gbox[..., 3:6] - gbox[..., 0:3])
print('group box extent mean xyz', ext.mean(dim=(0, 1)).tolist(), 'median', ext.median(dim=1).values.mean(0).tolist())
