"""Flow-embedding kernel alone at a given k (256 pairs x 512 points, ModelNet sizes): A/B of the 16- and 32-wide forms via DCLR_LIB."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepclr_amd import ops, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
dev = 'cuda:0'
torch.manual_seed(0)
for k in [int(v) for v in sys.argv[1:]] or [24, 25, 26, 28, 30, 32]:
    cfg = synthetic.model_cfg('modelnet'); cfg['params']['merge']['params']['k'] = k
    model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, 0)); model = model.to(dev).eval()
    flow = model._merge_layers[0]._embedding
    p = flow._packed()
    pairs, npoint = 256, 512
    rows = 2 * pairs * npoint
    f = torch.zeros(rows, ops.F_STRIDE, device=dev); f[:, :64] = torch.randn(rows, 64, device=dev).abs(); f[:, 64:67] = torch.randn(rows, 3, device=dev) * 0.5
    half = pairs * npoint
    pt = ops.linear(f[:half], p['wt'], None, 128, 64, relu=False); ps = ops.linear(f[half:], p['ws'], None, 128, 64, relu=False)
    idx = ops.knn_rows(f, pairs, npoint, k)
    fn = lambda: ops.flow_embedding_fused_f16(f, idx, pt, ps, p['w1a'], p['b1'], p['w2h'], p['b2'], p['w3h'], p['b3'], flow._radius)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); fn(); t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t) * 500)
    print('k=%2d tile %d: median %8.1f us min %8.1f' % (k, ops.flow_f16_tile(k), float(np.median(ts)), min(ts)), flush=True)
