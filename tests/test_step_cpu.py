"""distributed.DataParallelStep on the oracle-backed CPU backend (test infrastructure): the staged step on a batch whose motion heads
and TubeNet are both skipped (0 < n_fb <= MIN_POINTS) -- the second-pass loss is then constants only and has no graph; the early
gradients (ego / fb / perm terms) are the step's gradients and the optimizer step must still be taken, as the unstaged path and the
reference (libs/trainer.py:176-179) do.  ADVICE round 2."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import dist_worker  # noqa: E402
from oracle import cpu_backend  # noqa: E402
from pcaccumulation_amd import distributed as pdist  # noqa: E402
from pcaccumulation_amd import motionnet as mn  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402
from pcaccumulation_amd.loss import FuseLoss  # noqa: E402


def test_pipelined_step_with_few_foreground_points_still_steps(monkeypatch):
    cpu_backend.install(monkeypatch)
    if True:
        cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=8)
        inp = dist_worker.motionnet_batch(cfg, 0)
        monkeypatch.setattr(mn, 'MIN_POINTS', 10 ** 9)                  # every foreground count is "few": STPN heads and TubeNet skipped
        got = {}
        for pipelined in (False, True):
            model = dist_worker.motionnet_model(cfg)
            opt = torch.optim.SGD(model.parameters(), lr=0.0)
            step = pdist.DataParallelStep(model, opt, FuseLoss(cfg['loss']), iter_size=1, grad_clip=None, catch=False, pipelined=pipelined)
            assert step.pipelined == pipelined
            torch.manual_seed(100)
            stats = step(inp)
            assert step.ok and step.skipped == 0, step.last_error
            got[pipelined] = (float(stats['loss']), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        assert got[True][1].keys() == got[False][1].keys() and len(got[True][1]) > 0
        assert abs(got[True][0] - got[False][0]) <= 1e-5 * abs(got[False][0])
        for k, g in got[False][1].items():
            assert torch.allclose(got[True][1][k], g, rtol=1e-4, atol=1e-6 * float(g.abs().max()) + 1e-12), k


def test_step_hooks_are_called_once_each_and_cleared(monkeypatch):
    """before_sync (inside the forward, in front of its one host sync), after_forward, after_backward: each exactly once per micro-step, in that
    order, and the model keeps no hook of the step afterwards -- also when the forward raises."""
    cpu_backend.install(monkeypatch)
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=8)
    inp = dist_worker.motionnet_batch(cfg, 0)
    model = dist_worker.motionnet_model(cfg)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    step = pdist.DataParallelStep(model, opt, FuseLoss(cfg['loss']), iter_size=1, grad_clip=None, catch=False, pipelined=True)
    assert not step._early_thread and not step._tuning                  # the helper thread is a GPU matter
    calls = []
    stats = step(inp, before_sync=lambda: calls.append('sync'), after_forward=lambda: calls.append('fwd'), after_backward=lambda: calls.append('bwd'))
    assert calls == ['sync', 'fwd', 'bwd'] and stats is not None
    assert model.before_sync is None and model.after_ego is None and model.side_stream is None

    def boom():
        raise RuntimeError('data pipeline')
    try:
        step(inp, before_sync=boom)
        raise AssertionError('the hook error must surface')
    except RuntimeError as e:
        assert 'data pipeline' in str(e)
    assert model.before_sync is None and model.after_ego is None


def test_helper_thread_is_joined_whatever_fails_after_the_forward(monkeypatch):
    """the early backward issued from a helper thread must be joined -- and the interpreter's switch interval restored -- when
    something between the forward and the second backward raises (a caller's after_forward hook here), with catch=True (the reference's
    swallow-and-continue loop, libs/trainer.py:234-235) as well as catch=False; a malformed PCACC_SWITCH_INTERVAL must not raise mid-step."""
    import threading
    cpu_backend.install(monkeypatch)
    monkeypatch.setenv('PCACC_SWITCH_INTERVAL', 'not-a-number')
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=8)
    inp = dist_worker.motionnet_batch(cfg, 0)
    model = dist_worker.motionnet_model(cfg)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    before = sys.getswitchinterval()

    def boom():
        raise RuntimeError('prefetch failed')
    for catch in (True, False):
        step = pdist.DataParallelStep(model, opt, FuseLoss(cfg['loss']), iter_size=1, grad_clip=None, catch=catch, pipelined=True)
        step._early_thread = True                                          # the GPU-only choice, forced: the mechanism is host code
        assert step._helper_switch_interval == 2e-4
        if catch:
            assert step(inp, after_forward=boom) is None and not step.ok and 'prefetch failed' in str(step.last_error)
        else:
            try:
                step(inp, after_forward=boom)
                raise AssertionError('the hook error must surface')
            except RuntimeError as e:
                assert 'prefetch failed' in str(e)
        assert not [t for t in threading.enumerate() if t.name == 'pcacc-early-backward']
        assert sys.getswitchinterval() == before
    # and the threaded step itself completes on this backend
    step = pdist.DataParallelStep(model, opt, FuseLoss(cfg['loss']), iter_size=1, grad_clip=None, catch=False, pipelined=True)
    step._early_thread = True
    assert step(inp) is not None and step.ok
    assert sys.getswitchinterval() == before
