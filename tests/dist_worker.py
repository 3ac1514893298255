"""One gloo rank of tests/test_distributed.py (launched as a plain subprocess: `python dist_worker.py mode rank world port out.pt`)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(4, 3)
        self.b = torch.nn.Linear(3, 2)          # only used when the input says so (data-dependent branch)
        self.c = torch.nn.Linear(3, 5)          # never used: no gradient on any rank
        self.d = torch.nn.Linear(300, 300)      # big enough to close a bucket (bucket_bytes is small in the test)

    def forward(self, x, use_b):
        y = self.a(x)
        z = self.d(y.repeat(1, 100)).sum() * 1e-3
        return (self.b(y).sum() if use_b else y.sum()) + z


def toy_input(rank, micro):
    return torch.arange(8, dtype=torch.float32).view(2, 4) * 0.1 + rank + 0.5 * micro


def toy(rank, world, out):
    from pcaccumulation_amd import distributed as pdist
    # (1) round 1's blocking flat all-reduce
    torch.manual_seed(0)
    net = Net()
    net(toy_input(rank, 0), use_b=(rank == 0)).backward()
    assert (net.b.weight.grad is None) == (rank == 1)
    pdist.FlatGradAllReduce(net.parameters())()
    res = {'flat': {k: p.grad.clone() for k, p in net.named_parameters()}}
    res['ok'] = pdist.all_ok(rank == 0, torch.device('cpu'))
    res['mx'] = pdist.max_over_ranks(10.0 + rank, torch.device('cpu'))
    # (2) bucketed reducer, iter_size = 2: accumulate micro-step 0 locally, all-reduce (from the hooks) in micro-step 1
    torch.manual_seed(0)
    net = Net()
    red = pdist.BucketedGradReducer(net.parameters(), bucket_bytes=64 * 1024)
    assert len(red.buckets) >= 3
    red.zero()
    for micro in range(2):
        loss = net(toy_input(rank, micro), use_b=(rank == 0)) / 2
        red.prepare(loss, sync=(micro == 1))
        loss.backward()
    red.finish()
    ok = bool(red.agree(True).item())
    res['bucketed'] = {k: p.grad.clone() for k, p in net.named_parameters()}
    with red.sparse_grads():
        res['none_inside'] = [k for k, p in net.named_parameters() if p.grad is None]
    res['none_after'] = [k for k, p in net.named_parameters() if p.grad is None]
    res['ok2'] = ok
    # (3) a rank that fails still issues its collectives, and every rank skips the step
    opt = torch.optim.SGD(net.parameters(), lr=0.1)

    class Boom(torch.nn.Module):
        def forward(self, inp):
            if inp['fail']:
                raise RuntimeError('injected')
            return net(inp['x'], True)
    step = pdist.DataParallelStep(Boom(), opt, lambda o, i: {'loss': o}, iter_size=1, grad_clip=1.0, reducer=red)
    res['plain_by_default'] = (not step.pipelined) and step.side is None          # N > 1: one stream, one backward unless asked
    before = net.a.weight.detach().clone()
    step({'x': toy_input(rank, 0), 'fail': rank == 1})
    res['skipped'] = step.skipped
    res['unchanged'] = bool(torch.equal(before, net.a.weight.detach()))
    step({'x': toy_input(rank, 0), 'fail': False})
    res['stepped'] = not bool(torch.equal(before, net.a.weight.detach()))
    res['a_after'] = net.a.weight.detach().clone()
    # (4) two backward passes per micro-step: the buckets of set_early() go out during the first, the others during the second
    torch.manual_seed(0)
    net = Net()
    red = pdist.BucketedGradReducer(net.parameters(), bucket_bytes=64 * 1024)
    res['n_early'] = red.set_early(list(net.d.parameters()))
    res['seq'] = list(red._seq)
    red.zero()
    red.begin(sync=True)
    y = net.a(toy_input(rank, 0))
    loss_a = net.d(y.detach().repeat(1, 100)).sum() * 1e-3
    red.prepare(loss_a, part='early')
    loss_a.backward()
    res['out_after_first'] = list(red._launched)
    loss_b = net.b(y).sum() if rank == 0 else y.sum()
    red.prepare(loss_b, part='rest')
    loss_b.backward()
    red.finish()
    res['two_pass'] = {k: p.grad.clone() for k, p in net.named_parameters()}
    # a second loss that reaches an already reduced bucket is refused (its gradient would be lost)
    red.zero()
    red.begin(sync=True)
    loss_a = net.d(toy_input(rank, 0).repeat(1, 75)).sum()
    red.prepare(loss_a, part='early')
    loss_a.backward()
    try:
        red.prepare(net.d(toy_input(rank, 0).repeat(1, 75)).sum(), part='rest')
        res['refused'] = False
    except RuntimeError:
        res['refused'] = True
    red.finish()
    torch.save(res, out)


class Branchy(torch.nn.Module):
    """Four heads over a shared trunk; which heads run depends on the input (a rank-dependent set), like MotionNet's STPN / TubeNet
    branches (models/motionnet.py:222,243)."""

    def __init__(self):
        super().__init__()
        self.trunk = torch.nn.Linear(6, 16)
        self.heads = torch.nn.ModuleList([torch.nn.Linear(16, 120) for _ in range(4)])      # 120 x 16 floats: a bucket each at 8 KB buckets

    def forward(self, inp):
        y = torch.tanh(self.trunk(inp['x']))
        out = y.sum() * 0.1
        for i, h in enumerate(self.heads):
            if inp['use'][i]:
                out = out + h(y).pow(2).mean() * (i + 1)
        return out


def branchy_input(rank, step):
    g = torch.Generator().manual_seed(1000 + 17 * rank + step)
    # rank r skips head r (and rank 3 additionally head 0); on step 1 nobody runs head 2 at all
    use = [i != rank and not (rank == 3 and i == 0) and not (step == 1 and i == 2) for i in range(4)]
    return {'x': torch.randn(5, 6, generator=g), 'use': use}


def branchy(rank, world, out):
    """DataParallelStep over `world` ranks with a rank-dependent set of skipped branches: two optimizer steps, iter_size 2."""
    from pcaccumulation_amd import distributed as pdist
    torch.manual_seed(0)
    net = Branchy()
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    red = pdist.BucketedGradReducer(net.parameters(), bucket_bytes=8 * 1024)
    step = pdist.DataParallelStep(net, opt, lambda o, i: {'loss': o}, iter_size=2, grad_clip=None, reducer=red, catch=False)
    grads = []
    for s in range(2):
        for micro in range(2):
            step(branchy_input(rank, 2 * s + micro))
        grads.append({k: (p.grad.clone() if p.grad is not None else None) for k, p in net.named_parameters()})
    torch.save({'params': {k: p.detach().clone() for k, p in net.named_parameters()}, 'grads': grads, 'skipped': step.skipped,
                'n_buckets': len(red.buckets)}, out)


def motionnet_batch(cfg, rank):
    """rank 0: an ordinary tiny scene; rank 1: a scene without any foreground point (STPN and TubeNet are skipped there)."""
    from helpers import oracle_voxeliser
    from pcaccumulation_amd.dataloader import collate_fn
    from pcaccumulation_amd.synthetic import make_sequence, attach_voxels
    vox = oracle_voxeliser(cfg)
    if rank == 0:
        s = make_sequence(70, 3, 1400, cfg)
    else:
        s = make_sequence(71, 3, 1100, cfg, n_inst=0)
    return collate_fn([attach_voxels(s, vox)])


def motionnet_model(cfg):
    from pcaccumulation_amd.motionnet import MotionNet
    from pcaccumulation_amd.synthetic import fill_state_dict_
    torch.manual_seed(0)
    model = MotionNet(cfg)
    fill_state_dict_(model)
    with torch.no_grad():
        model.semseg_head.seg_head[3].bias += torch.tensor([1e4, 0.0])        # every pillar predicted background
    return model.train()


def motionnet(rank, world, out):
    """The tiny MotionNet scene through DataParallelStep on the oracle-backed CPU backend (test infrastructure)."""
    from oracle import cpu_backend
    from pcaccumulation_amd import distributed as pdist
    from pcaccumulation_amd.config import default_config
    from pcaccumulation_amd.loss import FuseLoss
    cpu_backend.install()
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=8)
    model = motionnet_model(cfg)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)                         # the step runs, the weights stay: gradients are the output
    step = pdist.DataParallelStep(model, opt, FuseLoss(cfg['loss']), iter_size=1, grad_clip=None, catch=False, pipelined=True)   # staged: still supported with N > 1
    torch.manual_seed(100 + rank)
    stats = step(motionnet_batch(cfg, rank))
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    touched = {k: t for (k, _), t in zip(model.named_parameters(), step.reducer.touched)}
    torch.save({'grads': grads, 'touched': touched, 'loss': float(stats['loss']), 'skipped': step.skipped,
                'n_buckets': len(step.reducer.buckets), 'n_early': step.reducer._n_early, 'seq': list(step.reducer._seq)}, out)


def forced(rank, world, out):
    """World size 1 with PCACC_FORCE_PROCESS_GROUP=1 (set by the test): a process group exists, the reducer is ACTIVE -- gradients travel through the flat
    buffer and every bucket through all_reduce -- and the result is the plain single-process gradient; without the switch one rank reduces nothing."""
    from pcaccumulation_amd import distributed as pdist
    assert world == 1 and torch.distributed.is_initialized()
    torch.manual_seed(0)
    net = Net()
    red = pdist.BucketedGradReducer(net.parameters(), bucket_bytes=64 * 1024)
    assert red.active and red.world == 1 and red.flat.numel() == red.numel
    red.zero()
    loss = net(toy_input(0, 0), use_b=True)
    red.prepare(loss, sync=True)
    loss.backward()
    red.finish()
    flag = red.agree(True)
    res = {'grads': {k: (p.grad.clone() if p.grad is not None else None) for k, p in net.named_parameters()}, 'collectives': red.collectives,
           'n_buckets': len(red.buckets), 'flag': int(flag.item()), 'views': all(p.grad is None or p.grad.data_ptr() == v.data_ptr() for p, v in zip(red.params, red.views))}
    with red.sparse_grads():
        res['none_inside'] = [k for k, p in net.named_parameters() if p.grad is None]
    os.environ.pop('PCACC_FORCE_PROCESS_GROUP')
    plain = pdist.BucketedGradReducer(Net().parameters())
    res['plain_active'] = plain.active
    torch.save(res, out)


def main():
    mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
    torch.set_num_threads(2)
    from pcaccumulation_amd import distributed as pdist
    r, w, _ = pdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    {'toy': toy, 'motionnet': motionnet, 'branchy': branchy, 'forced': forced}[mode](rank, world, out)
    pdist.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
