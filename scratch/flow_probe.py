"""Flow-embedding kernel alone (k = 20, KITTI sizes; k = 30 ModelNet), HIP events; DCLR_FLOW_ABL selects timing-only ablations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepclr_amd import ops, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
dev = 'cuda:0'
torch.manual_seed(0)
for kind, pairs, npoint in (('kitti', 8, 1024), ('kitti', 80, 1024), ('modelnet', 256, 512)):
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
    out = fn(); torch.cuda.synchronize()
    import hashlib; digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]      # bit-identity across builds
    ts = []
    for _ in range(10):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); fn(); t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t) * 500)
    flop = 2.0 * pairs * npoint * flow._k * (128 * 128 + 128 * 256 + 128 * 5)
    med = float(np.median(ts))
    print('%-8s %3d pairs k=%2d: median %8.1f us min %8.1f  -> %.0f TFLOP/s f32-equivalent, frac %.3f  sha1 %s' % (kind, pairs, flow._k, med, min(ts), flop / med / 1e6, flop / med / 1e6 / 838.9, digest), flush=True)
