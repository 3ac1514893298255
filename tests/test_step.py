"""distributed.DataParallelStep on one GPU: the staged step (early backward of the ego / fb / perm terms in the middle of the forward,
motion heads + TubeNet + their backward on a second HIP stream beside it) against the plain step (one backward at the end, one
stream) on the same weights, scene and seed: same loss statistics, same gradients."""
import pytest
import torch

from helpers import make_batch
from pcaccumulation_amd import distributed as pdist
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.loss import FuseLoss
from pcaccumulation_amd.motionnet import MotionNet
from pcaccumulation_amd.synthetic import fill_state_dict_

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('compute_dtype', ['fp32', 'bf16'])
def test_two_stream_pipelined_step_matches_plain_step(compute_dtype):
    dev = torch.device('cuda:0')
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=16)
    cfg['misc']['compute_dtype'] = compute_dtype
    torch.manual_seed(0)
    model = MotionNet(cfg)
    fill_state_dict_(model)
    model = model.to(dev).train().channels_last_()
    inp = make_batch(cfg, [11, 12], 3, 6000)
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    loss_fn = FuseLoss(cfg['loss'])
    opt = torch.optim.SGD(model.parameters(), lr=0.0)                       # the step runs, the weights stay: gradients are the output
    def run(step):
        torch.manual_seed(5)
        stats = step(inp)
        torch.cuda.synchronize()
        assert step.skipped == 0
        return (float(stats['loss']), {k: float(stats[k]) for k in ('ego_l1_loss', 'fb_loss', 'mos_loss', 'obj_loss') if k in stats},
                {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})

    def rel(a, b):
        """largest |difference| of two gradient sets relative to the largest entry of the tensor it occurs in; loss difference"""
        worst = max(float((a[2][k] - b[2][k]).abs().max()) / (float(b[2][k].abs().max()) + 1e-12) for k in b[2])
        return worst, abs(a[0] - b[0]) / abs(b[0])

    plain = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=None, catch=False, pipelined=False)
    assert plain.side is None
    ref, again = run(plain), run(plain)
    # the step is not bit-reproducible (atomic row sums, library convolutions; in bf16 a rounding can flip a foreground decision and
    # with it a key-point draw): the plain step against itself sets the scale for "the same"
    noise_g, noise_l = rel(again, ref)
    tol_g = max(4 * noise_g, 2e-2 if compute_dtype == 'bf16' else 5e-3)
    tol_l = max(4 * noise_l, 1e-4)
    for kw in (dict(two_streams=True), dict(two_streams=False)):
        step = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=None, catch=False, pipelined=True, **kw)
        assert (step.side is not None) == kw['two_streams']
        run(step)                                                             # twice: the second call reuses cached blocks of both streams
        got = run(step)
        assert got[1].keys() == ref[1].keys() and got[2].keys() == ref[2].keys()
        d_g, d_l = rel(got, ref)
        assert d_l <= tol_l and d_g <= tol_g, (kw, d_l, tol_l, d_g, tol_g)
