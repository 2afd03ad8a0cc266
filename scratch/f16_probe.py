"""Split-fp16 vs f32 matrix path: agreement and kernel times (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deepclr_amd import ops, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model

dev = 'cuda:0'
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, seed=0))
model = model.to(dev).eval()
x = torch.from_numpy(synthetic.make_batch('kitti', 8, 16384)).to(dev)
out = {}
with torch.no_grad():
    f_rows = model.cloud_feature_rows(x)
    for prec in ('f32', 'f16x2'):
        ops.PRECISION = prec
        e = model._merge_layers[0].forward_rows(f_rows, 8, 1024)
        y = model._merge_layers[1].forward_rows(e, 8)
        out[prec] = (e.clone(), y.clone())
        t = bench.LaunchTimer(sample_every=1); ops.TIMER = t
        for _ in range(10):
            e = model._merge_layers[0].forward_rows(f_rows, 8, 1024)
            y = model._merge_layers[1].forward_rows(e, 8)
        torch.cuda.synchronize(); ops.TIMER = None
        print(prec, {k: round(v['avg_us'], 1) for k, v in sorted(t.summary().items())})
e32, y32 = out['f32']; e16, y16 = out['f16x2']
print('E rows  max abs diff', (e32 - e16).abs().max().item(), 'max |E|', e32.abs().max().item())
print('pose y  max abs diff', (y32 - y16).abs().max().item())
print(y32[0].tolist()); print(y16[0].tolist())
# head alone on identical input, against float64
head = model._merge_layers[1]
want = e32[:, :259].double()
want = torch.cat((want[:, 256:259], want[:, :256]), dim=1)
for w, b in head.conv.affine_params():
    want = torch.relu(want @ w.detach().double().reshape(w.shape[0], -1).t() + b.detach().double())
want = want.view(8, 1024, -1).max(dim=1).values
for prec in ('f32', 'f16x2'):
    ops.PRECISION = prec
    if prec == 'f32':
        g = ops.head_conv_fused(e32, head._packed(), 8)
    else:
        g = ops.head_conv_fused_f16(e32, ops.E_STRIDE, head._packed_f16(), 8)
    err = (g.double() - want).abs()
    print(prec, 'head vs fp64: max abs', err.max().item(), 'max rel', (err / want.abs().clamp(min=1e-3)).max().item())
