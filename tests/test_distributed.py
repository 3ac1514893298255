"""N > 1 path on CPU: two gloo ranks (world_size 2), each a plain subprocess.

Checks that the gradient reduction (round 1's blocking flat all-reduce and the bucketed, hook-driven reducer with iter_size
accumulation) yields the gradient of the two ranks' mean loss, also when a parameter has NO gradient on one rank (data-dependent
branches of MotionNet, models/motionnet.py:222,243) or on any rank; that a rank whose step raises does not hang the others and
the step is skipped everywhere; and -- on the tiny MotionNet scene with one rank lacking any foreground point -- that the
all-reduced gradient equals the single-process gradient of the two scenes."""
import os
import socket
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(mode, tmp_path, timeout, world=2):
    port = str(_free_port())
    outs = [str(tmp_path / ('%s_rank%d.pt' % (mode, r))) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker.py'), mode, str(r), str(world), port, outs[r]]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=timeout) == 0
    return [torch.load(o) for o in outs]


def test_gradient_reduction_two_gloo_ranks(tmp_path):
    sys.path.insert(0, HERE)
    from dist_worker import Net, toy_input
    got = _launch('toy', tmp_path, 180)

    def reference(micros):
        torch.manual_seed(0)
        net = Net()
        total = 0
        for rank in range(2):
            for micro in range(micros):
                total = total + net(toy_input(rank, micro), use_b=(rank == 0)) / micros
        (total / 2).backward()
        return {k: p.grad for k, p in net.named_parameters()}

    ref1, ref2 = reference(1), reference(2)
    for rank in range(2):
        g = got[rank]
        assert g['ok'] is False and g['mx'] == 11.0
        for k in ref1:
            if ref1[k] is None:                               # Net.c: round 1's path hands the optimizer zeros there (ADVICE)
                assert float(g['flat'][k].abs().sum()) == 0.0
            else:
                assert torch.allclose(g['flat'][k], ref1[k], atol=1e-6), (rank, k)
        for k in ref2:
            want = ref2[k] if ref2[k] is not None else torch.zeros_like(g['bucketed'][k])
            assert torch.allclose(g['bucketed'][k], want, atol=1e-6), (rank, k)
        # a parameter no rank produced a gradient for is None for the optimizer (single-process semantics), a view again afterwards
        assert g['none_inside'] == ['c.weight', 'c.bias'] and g['none_after'] == [] and g['ok2'] is True
        # rank 1 raised in its forward: nobody hangs, nobody steps; the next (healthy) step is taken by both
        assert g['skipped'] == 1 and g['unchanged'] and g['stepped'] and g['plain_by_default']
    assert torch.equal(got[0]['a_after'], got[1]['a_after'])
    # two-pass micro-step: `d`'s bucket(s) lead the sequence and are out after the first backward, nothing else is
    def two_pass_reference():
        grads = []
        for rank in range(2):
            torch.manual_seed(0)
            net = Net()
            y = net.a(toy_input(rank, 0))
            (net.d(y.detach().repeat(1, 100)).sum() * 1e-3 + (net.b(y).sum() if rank == 0 else y.sum())).backward()
            grads.append({k: p.grad for k, p in net.named_parameters()})
        return {k: sum((g[k] if g[k] is not None else 0) for g in grads) / 2 for k in grads[0]}
    ref3 = two_pass_reference()
    for rank, g in enumerate(got):
        assert g['n_early'] >= 1 and g['seq'] == got[0]['seq'] and g['refused']
        early = set(g['seq'][:g['n_early']])
        assert [b for b, out in enumerate(g['out_after_first']) if out] == sorted(early)
        for k in ref3:
            want = ref3[k] if torch.is_tensor(ref3[k]) else torch.zeros_like(g['two_pass'][k])
            assert torch.allclose(g['two_pass'][k], want, atol=1e-6), (rank, k)


def test_motionnet_data_dependent_branches_two_gloo_ranks(tmp_path):
    """Tiny MotionNet scene per rank; rank 1's scene has no foreground, so the STPN and the TubeNet never run there."""
    sys.path.insert(0, HERE)
    from dist_worker import motionnet_batch, motionnet_model
    from oracle import cpu_backend
    from pcaccumulation_amd.config import default_config
    from pcaccumulation_amd.loss import FuseLoss
    got = _launch('motionnet', tmp_path, 600)
    assert got[0]['n_buckets'] >= 4 and got[0]['skipped'] == got[1]['skipped'] == 0
    # the early backward's buckets (pillar encoder ... ego head) lead the launch sequence, identically on both ranks
    assert got[0]['n_early'] >= 2 and got[0]['seq'] == got[1]['seq'] and got[0]['seq'][0] > 0
    assert got[0]['touched']['motionhead.final_proj.0.weight'] and not got[1]['touched']['motionhead.final_proj.0.weight']
    assert got[0]['touched']['reconstructor.alignment.regressor.0.weight'] and not got[1]['touched']['reconstructor.alignment.regressor.0.weight']

    # single process: the two scenes through the same weights (BatchNorm statistics per scene, as per GPU), mean of the losses
    import pytest
    mp = pytest.MonkeyPatch()
    try:
        cpu_backend.install(mp)
        cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=8)
        model = motionnet_model(cfg)
        loss_fn = FuseLoss(cfg['loss'])
        total, losses = 0, []
        for rank in range(2):
            torch.manual_seed(100 + rank)
            inp = motionnet_batch(cfg, rank)
            stats = loss_fn(model(inp), inp)
            losses.append(float(stats['loss']))
            total = total + stats['loss']
        (total / 2).backward()
    finally:
        mp.undo()
    for rank in range(2):
        assert abs(got[rank]['loss'] - losses[rank]) <= 1e-5 * abs(losses[rank])
        for k, p in model.named_parameters():
            want = p.grad if p.grad is not None else torch.zeros_like(p)
            g = got[rank]['grads'][k]
            # layers upstream of the STPN's max over frames / max-pools: near-ties in empty regions pick another winner when the
            # convolutions sum in another order (2 threads per rank here, all cores in this process) -- see test_model_parity
            loose = k.startswith(('motionhead.init_conv', 'motionhead.down_convs', 'motionhead.up_convs'))
            atol = (2e-2 if loose else 1e-5) * float(want.abs().max()) + 1e-6
            assert torch.allclose(g, want, rtol=1e-4, atol=atol), (rank, k, float((g - want).abs().max()), float(want.abs().max()))


def test_four_gloo_ranks_with_rank_dependent_branches(tmp_path):
    """world_size 4: every rank skips a different head (rank 3 two of them), in one micro-step nobody runs head 2 -- buckets whose
    gradients are missing on some / all ranks must still go out in the same order everywhere; two optimizer steps with iter_size 2
    against one process that sums the four ranks' losses."""
    sys.path.insert(0, HERE)
    from dist_worker import Branchy, branchy_input
    got = _launch('branchy', tmp_path, 240, world=4)
    torch.manual_seed(0)
    net = Branchy()
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    for s in range(2):
        opt.zero_grad(set_to_none=True)
        total = 0
        for rank in range(4):
            for micro in range(2):
                total = total + net(branchy_input(rank, 2 * s + micro)) / 2
        (total / 4).backward()
        for rank in range(4):
            for k, p in net.named_parameters():
                g = got[rank]['grads'][s][k]
                want = p.grad if p.grad is not None else torch.zeros_like(p)
                assert g is not None and torch.allclose(g, want, atol=1e-6), (s, rank, k)
        opt.step()
    for rank in range(4):
        assert got[rank]['skipped'] == 0 and got[rank]['n_buckets'] >= 4
        for k, p in net.named_parameters():
            assert torch.allclose(got[rank]['params'][k], p.detach(), atol=1e-6), (rank, k)


def test_forced_process_group_at_world_size_one(tmp_path, monkeypatch):
    """The code path of tests/test_bench_multirank.py::test_one_rccl_rank_runs_the_production_step on CPU: one gloo rank with
    PCACC_FORCE_PROCESS_GROUP=1 initialises a process group (distributed.init_from_env), the reducer goes through its flat buffer and issues one
    all_reduce per bucket plus the agreement reduce, `.grad` become views of the flat buffer, and the gradient is the single-process gradient."""
    sys.path.insert(0, HERE)
    from dist_worker import Net, toy_input
    monkeypatch.setenv('PCACC_FORCE_PROCESS_GROUP', '1')
    got = _launch('forced', tmp_path, 120, world=1)[0]
    torch.manual_seed(0)
    net = Net()
    net(toy_input(0, 0), use_b=True).backward()
    assert got['n_buckets'] >= 3 and got['collectives'] == got['n_buckets'] + 1 and got['flag'] == 1 and got['views']
    assert got['none_inside'] == ['c.weight', 'c.bias'] and got['plain_active'] is False
    for k, p in net.named_parameters():
        if p.grad is None:
            assert float(got['grads'][k].abs().sum()) == 0.0          # outside sparse_grads() an untouched parameter shows its zeroed view
        else:
            assert torch.allclose(got['grads'][k], p.grad, atol=1e-6), k
