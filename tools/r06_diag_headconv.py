#!/usr/bin/env python
"""Diagnosis for tests/test_determinism.py::test_staged_two_stream_step_is_bit_reproducible: which operand of the fg/bg head's last weight gradient differs
between two staged steps -- dy, x, or only the result?  Records clones of every native.head_conv3x3_wgrad call over repeated steps, for the step variants."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

from helpers import make_batch  # noqa: E402
from pcaccumulation_amd import distributed as pdist, native  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402
from pcaccumulation_amd.loss import FuseLoss  # noqa: E402
from pcaccumulation_amd.motionnet import MotionNet  # noqa: E402
from pcaccumulation_amd.synthetic import fill_state_dict_  # noqa: E402

dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = 'mixed'
torch.manual_seed(0)
model = MotionNet(cfg)
fill_state_dict_(model)
model = model.to(dev).train().channels_last_()
inp = make_batch(cfg, [21, 22], 5, 30000, mode='lidar')
inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
opt = torch.optim.SGD(model.parameters(), lr=0.0)
rec = []
orig = native.head_conv3x3_wgrad


def spy(gy, x, want_bias=True):
    out = orig(gy, x, want_bias=want_bias)
    if os.environ.get('PCACC_DIAG_SYNC'):
        torch.cuda.current_stream().synchronize()
    rec.append((gy.detach().clone(), x.detach().clone(), out[0].detach().clone(), int(torch.cuda.current_stream().cuda_stream)))
    return out


if not os.environ.get('PCACC_DIAG_NOSPY'):
    native.head_conv3x3_wgrad = spy
for kw in (dict(two_streams=True, early_thread=True), dict(two_streams=True, early_thread=True), dict(two_streams=True, early_thread=False), dict(two_streams=False, early_thread=False)):
    step = pdist.DataParallelStep(model, opt, FuseLoss(cfg['loss']), iter_size=1, grad_clip=None, catch=False, pipelined=True, **kw)
    rec.clear()
    grads = []
    allg = []
    for r in range(12):
        torch.manual_seed(5)
        step(dict(inp))
        torch.cuda.synchronize()
        grads.append(model.semseg_head.seg_head[3].weight.grad.detach().clone())
        allg.append({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    bad = {}
    for r in range(2, 12):
        for k in allg[1]:
            if not torch.equal(allg[1][k], allg[r][k]):
                bad.setdefault(k, []).append(r)
    print(kw, 'parameters whose gradient differs from run 1 in some later run:', bad)
    n = len(rec) // 12
    for r in range(1, 12 if n else 0):
        for c in range(n):
            a, b = rec[c], rec[r * n + c]
            print('  run %d call %d: dy equal %s (%d differ)  x equal %s (%d differ)  dw equal %s   | final grad equal %s' % (
                r, c, torch.equal(a[0], b[0]), int((a[0] != b[0]).sum()), torch.equal(a[1], b[1]), int((a[1] != b[1]).sum()), torch.equal(a[2], b[2]),
                torch.equal(grads[0], grads[r])))
