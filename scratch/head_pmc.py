"""Few launches of the two head kernels for a rocprofv3 --pmc pass (scratch/head_pmc.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault('DCLR_HEAD_REG', '1')
from deepclr_amd import ops, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
dev = 'cuda:0'
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, 0)); model = model.to(dev).eval()
head = model._merge_layers[1]
packed, bias = head._packed_reg(); l16 = head._packed_f16()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
e = torch.zeros(rows, ops.E_STRIDE, device=dev); e[:, :259] = torch.randn(rows, 259, device=dev)
for _ in range(6):
    ops.head_conv_fused_f16(e, ops.E_STRIDE, l16, rows // 1024)
    ops.head_conv_reg_f16(e, ops.E_STRIDE, packed, bias, rows // 1024)
torch.cuda.synchronize()
