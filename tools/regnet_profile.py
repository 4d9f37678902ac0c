"""One RegNet forward on the fp16 engine under `rocprofv3 --kernel-trace --stats` (batch from argv): which kernels the time
goes to.    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rn -o rn -- python3 tools/regnet_profile.py 16"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from empanada_napari_amd import synth  # noqa: E402
from empanada_napari_amd.engines import HipPanopticDeepLab  # noqa: E402
from test_regnet import regnet_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tag = sys.argv[2] if len(sys.argv) > 2 else 'x'
cfg, P = regnet_model(tag)
m = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
x = torch.from_numpy(synth.em_tiles(B, 1024, seed=3))[:, None].cuda()
for _ in range(4):
    m(x, 2, False, sub=0.57571 * 255, mul=1 / (0.12765 * 255))
torch.cuda.synchronize()
