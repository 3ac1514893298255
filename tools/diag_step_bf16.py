"""Which gradient tensors differ between the staged (pipelined) and the plain step in bf16, against the plain step's own run-to-run scatter."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import make_batch
from pcaccumulation_amd.synthetic import fill_state_dict_
from pcaccumulation_amd import distributed as pdist
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.loss import FuseLoss
from pcaccumulation_amd.motionnet import MotionNet
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=16)
cfg['misc']['compute_dtype'] = os.environ.get('PCACC_DTYPE', 'bf16')
torch.manual_seed(0)
model = MotionNet(cfg)
fill_state_dict_(model)
with torch.no_grad():
    model.semseg_head.seg_head[3].bias += torch.tensor([1e4, 0.0])
model = model.to(dev).train().channels_last_()
inp = make_batch(cfg, [11, 12], 3, 6000)
inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
loss_fn = FuseLoss(cfg['loss'])
opt = torch.optim.SGD(model.parameters(), lr=0.0)
def run(step):
    torch.manual_seed(5)
    stats = step(inp)
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
plain = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=None, catch=False, pipelined=False)
a, b, c = run(plain), run(plain), run(plain)
st = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=None, catch=False, pipelined=True, two_streams=False)
run(st); s = run(st)
rel = lambda x, y: float((x - y).abs().max()) / (float(y.abs().max()) + 1e-12)
rows = sorted(((rel(s[k], a[k]), max(rel(b[k], a[k]), rel(c[k], a[k])), k, tuple(a[k].shape), float(a[k].abs().max())) for k in a), reverse=True)
for r in rows[:14]:
    print('staged-vs-plain %.4f  plain-vs-plain %.4f  %-50s %s max|g| %.3g' % r)
