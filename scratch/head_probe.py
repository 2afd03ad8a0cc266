"""A/B timing of the pose-head conv chain alone: LDS-resident split-f16 kernel (head16_kernel) vs the register-resident
one (head_reg_kernel), interleaved rounds in one process, HIP events on the launch stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ.setdefault('DCLR_HEAD_REG', '1')
from deepclr_amd import ops, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model

dev = 'cuda:0'
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, 0)); model = model.to(dev).eval()
head = model._merge_layers[1]
packed, bias = head._packed_reg()
l16 = head._packed_f16()
FLOP_ROW = 2.0 * sum(a * b for a, b in zip([264, 256, 256, 512, 512], [256, 256, 512, 512, 1024]))
for rows in (8192, 16384, 32768, 131072):
    pairs = rows // 1024
    e = torch.zeros(rows, ops.E_STRIDE, device=dev)
    e[:, :259] = torch.randn(rows, 259, device=dev)
    fns = {'lds16': lambda: ops.head_conv_fused_f16(e, ops.E_STRIDE, l16, pairs),
           'reg': lambda: ops.head_conv_reg_f16(e, ops.E_STRIDE, packed, bias, pairs)}
    a, b = fns['lds16'](), fns['reg']()
    torch.cuda.synchronize()
    print('rows %6d  max|lds16 - reg| = %.3g (scale %.3g)' % (rows, (a - b).abs().max().item(), a.abs().max().item()), flush=True)
    times = {k: [] for k in fns}
    for rnd in range(12):
        for k, fn in fns.items():
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); 
            for _ in range(4): fn()
            t.record(); torch.cuda.synchronize()
            times[k].append(s.elapsed_time(t) / 4 * 1e3)
    for k, v in times.items():
        med = float(np.median(v[2:]))
        print('   %-6s median %8.1f us  min %8.1f us   %.1f TFLOP/s (f32-equivalent)  frac of 838.9: %.3f' %
              (k, med, min(v), FLOP_ROW * rows / med / 1e6, FLOP_ROW * rows / med / 1e6 / 838.9), flush=True)
