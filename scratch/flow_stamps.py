"""Cycle stamps inside flow32_kernel (library built with -DDCLR_FLOW_STAMPS, selected with DCLR_LIB): per phase, the median / p90
over the sampled workgroups of one launch, in shader cycles (s_memtime)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepclr_amd import ops, synthetic, lib
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
dev = 'cuda:0'
torch.manual_seed(0)
NST = 10
names = ["phase A", "wait bar1", "L2 mfma", "wait bar2", "L2 epilogue", "wait bar3", "L3 mfma a", "L3 epi a + mfma b", "L3 epi b + tail"]
for kind, pairs, npoint in (('kitti', 80, 1024), ('modelnet', 256, 512)):
    cfg = synthetic.model_cfg(kind)
    model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, 0)); model = model.to(dev).eval()
    flow = model._merge_layers[0]._embedding
    p = flow._packed()
    rows = 2 * pairs * npoint
    f = torch.zeros(rows, ops.F_STRIDE, device=dev); f[:, :64] = torch.randn(rows, 64, device=dev).abs(); f[:, 64:67] = torch.randn(rows, 3, device=dev) * (20 if kind == 'kitti' else 0.5)
    half = pairs * npoint
    pt = ops.linear(f[:half], p['wt'], None, 128, 64, relu=False); ps = ops.linear(f[half:], p['ws'], None, 128, 64, relu=False)
    idx = ops.knn_rows(f, pairs, npoint, flow._k)
    fn = lambda: ops.flow_embedding_fused_f16(f, idx, pt, ps, p['w1a'], p['b1'], p['w2h'], p['b2'], p['w3h'], p['b3'], flow._radius)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); t.record(); torch.cuda.synchronize()
    blocks = min(1024, (pairs * npoint // 4) // 64)
    buf = np.zeros((blocks, 4, NST), dtype=np.uint64)
    fnc = lib.load().dclr_debug_flow_stamps
    fnc.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = fnc(buf.ctypes.data, blocks); assert rc == 0, rc
    st = buf.astype(np.int64)
    d = np.diff(st, axis=2)                                   # (blocks, waves, 9)
    total = st[:, :, 9] - st[:, :, 0]
    span = (st[:, :, 9].max() - st[:, :, 0].min())
    print('%s %d pairs k=%d: launch %.1f us; sampled %d workgroups; first start .. last end = %d cycles -> %.2f GHz if that is the launch'
          % (kind, pairs, flow._k, s.elapsed_time(t) * 1e3, blocks, span, span / (s.elapsed_time(t) * 1e3) / 1e3))
    print('  workgroup lifetime (per wave): median %d  p10 %d  p90 %d cycles' % (np.median(total), np.percentile(total, 10), np.percentile(total, 90)))
    for i, n in enumerate(names):
        v = d[:, :, i].reshape(-1)
        print('  %-20s median %6d  p10 %6d  p90 %6d   (%.1f %%)' % (n, np.median(v), np.percentile(v, 10), np.percentile(v, 90), 100.0 * v.mean() / total.mean()))
