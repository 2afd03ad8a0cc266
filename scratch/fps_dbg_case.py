import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from deepclr_amd import ops
from test_gpu_ops import _cloud
for kind, b, n, m in (('kitti', 1, 16385, 60), ('kitti', 1, 16386, 60), ('kitti', 1, 16448, 60), ('kitti', 1, 16449, 60), ('kitti', 1, 16500, 60), ('kitti', 1, 16639, 60), ('kitti', 1, 32769, 60), ('kitti', 1, 33000, 60)):
    xyz = _cloud(kind, b, n, seed=16385 + 200)
    want = oracle.furthest_point_sample(xyz, m)
    x = xyz.to('cuda:0')
    got, gp, gb = ops.fps_clouds_grouped(x, m)
    got = got.cpu()
    bad = (got != want).nonzero()
    print(kind, n, m, 'ok' if len(bad) == 0 else 'first mismatch %s got %d want %d' % (bad[0].tolist(), got[tuple(bad[0])], want[tuple(bad[0])]), flush=True)
    if len(bad):
        k = gp[0, :, 3].contiguous().view(torch.int32).cpu()
        for name, v in (('got', int(got[tuple(bad[0])])), ('want', int(want[tuple(bad[0])]))):
            pos = int((k == v).nonzero()[0])
            gid = pos // 256
            print('   %s idx %d at sorted pos %d: gid %d (wave %d, g %d), lane %d slot %d' % (name, v, pos, gid, gid % 16, gid // 16, pos % 64, (pos % 256) // 64))
        real = (k.view(-1, 256) >= 0).sum(1)
        print('   real points per gid:', real.tolist()[60:70], '... nonempty groups', int((real > 0).sum()))
