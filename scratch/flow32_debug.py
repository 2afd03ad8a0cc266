"""Debug: flow kernels at every k on hand-made lists (from tests/test_gpu_model.py), printing the error per k and variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepclr_amd import ops, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
DEV = 'cuda:0'
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, 13)); model = model.to(DEV).eval()
me = model._merge_layers[0]._embedding
p = me._packed()
(w1, b1), (w2, b2), (w3, b3) = me._conv.affine_params()
w64 = [(w.detach().double().reshape(w.shape[0], -1), b.detach().double()) for w, b in ((w1, b1), (w2, b2), (w3, b3))]
for pairs, npoint in ((3, 37), (8, 16)):
    rng = np.random.default_rng(100 * pairs + npoint)
    rows = 2 * pairs * npoint
    f = np.zeros((rows, ops.F_STRIDE), dtype=np.float32)
    f[:, :64] = np.abs(rng.normal(size=(rows, 64))); f[:, 64:67] = rng.normal(scale=1.5, size=(rows, 3))
    f_rows = torch.from_numpy(f).to(DEV); half = pairs * npoint
    w1f = w1.detach().double().reshape(w1.shape[0], -1)
    pt = (f_rows[:half, :64].double() @ w1f[:, 3:67].t()).float().contiguous()
    ps = (f_rows[half:, :64].double() @ w1f[:, 67:131].t()).float().contiguous()
    f64 = f_rows.double(); tmpl, src = f64[:half].view(pairs, npoint, -1), f64[half:].view(pairs, npoint, -1)
    for variant in ('plain', 'unfilled', 'radius'):
        for k in (20, 25, 26, 27, 28, 29, 30, 31, 32):
            idx = rng.integers(0, npoint, size=(pairs, npoint, k)).astype(np.int32)
            radius = 1e9
            if variant == 'unfilled':
                idx[rng.random(idx.shape) < 0.15] = -1
            if variant == 'radius':
                radius = 2.0
            idx_t = torch.from_numpy(idx).to(DEV)
            tile = ops.flow_f16_tile(k)
            w2h, w3h = ops.pack_weight_f16(w2, 128, tile), ops.pack_weight_f16(w3, 128, tile)
            e16 = ops.flow_embedding_fused_f16(f_rows, idx_t, pt, ps, p['w1a'], p['b1'], w2h, p['b2'], w3h, p['b3'], radius)
            li = torch.from_numpy(idx.clip(0)).long().to(DEV)
            nb = torch.gather(src.unsqueeze(1).expand(-1, npoint, -1, -1), 2, li.unsqueeze(-1).expand(-1, -1, -1, f64.shape[1]))
            diff = nb[..., 64:67] - tmpl[:, :, None, 64:67]
            h = torch.cat((diff, tmpl[:, :, None, :64].expand(-1, -1, k, -1), nb[..., :64]), dim=-1)
            for w, b in w64:
                h = torch.relu(h @ w.t() + b)
            dead = (diff.norm(dim=-1, keepdim=True) >= radius) | (idx_t < 0).unsqueeze(-1)
            want = torch.where(dead, torch.zeros_like(h), h).max(dim=2).values.view(half, 256)
            err = (e16[:, :256].double() - want).abs()
            bad_rows = (err.max(dim=1).values > 1e-4).nonzero().flatten().tolist()
            print('%d x %d %-9s k=%2d tile %d: max err %.3e  bad points %s' % (pairs, npoint, variant, k, tile, err.max().item(), bad_rows[:12]), flush=True)
