"""Stamped (diagnostic) build of head_reg_kernel: where a stage's cycles go. DCLR_HR_ABL=4 (with DMA) or 5 (without)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault('DCLR_HEAD_REG', '1')
from deepclr_amd import ops, synthetic, lib
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
dev = 'cuda:0'
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, 0)); model = model.to(dev).eval()
head = model._merge_layers[1]
packed, bias = head._packed_reg()
rows = 8192
e = torch.zeros(rows, ops.E_STRIDE, device=dev); e[:, :259] = torch.randn(rows, 259, device=dev)
for _ in range(5):
    ops.head_conv_reg_f16(e, ops.E_STRIDE, packed, bias, rows // 1024)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
l = lib.load()
l.dclr_head_reg_debug.argtypes = [ctypes.c_void_p]
assert l.dclr_head_reg_debug(out) == 0
for w, base in ((0, 0), (3, 8)):
    tot, a, d, b = out[base], out[base + 1], out[base + 2], out[base + 3]
    print('wave %d: kernel %d cycles (s_memtime ticks); per stage (258): half A + wait + barrier %.0f, DMA issue %.0f, half B %.0f; stages %.0f %% of kernel'
          % (w, tot, a / 258, d / 258, b / 258, 100.0 * (a + d + b) / max(tot, 1)))
