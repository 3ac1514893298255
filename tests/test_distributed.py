"""N > 1 path on CPU: two gloo ranks (world_size 2) launched with torch.multiprocessing.

Checks that the flat-buffer all-reduce yields the gradient of the concatenated batch (mean over ranks), also when
a parameter has NO gradient on one rank (data-dependent branches of MotionNet, models/motionnet.py:222,243), and
that the "all ranks ok" flag and the max-over-ranks timing reduce correctly."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(4, 3)
        self.b = torch.nn.Linear(3, 2)          # only used when the input says so (data-dependent branch)

    def forward(self, x, use_b):
        y = self.a(x)
        return self.b(y).sum() if use_b else y.sum()


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from pcaccumulation_amd import distributed as pdist
    r, w, _ = pdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    net = _Net()
    x = torch.arange(8, dtype=torch.float32).view(2, 4) + rank
    loss = net(x, use_b=(rank == 0))
    loss.backward()
    assert (net.b.weight.grad is None) == (rank == 1)
    pdist.FlatGradAllReduce(net.parameters())()
    ok = pdist.all_ok(rank == 0, torch.device('cpu'))
    mx = pdist.max_over_ranks(10.0 + rank, torch.device('cpu'))
    q.put((rank, {k: p.grad.clone() for k, p in net.named_parameters()}, ok, mx))
    pdist.barrier()
    torch.distributed.destroy_process_group()


def test_flat_allreduce_two_gloo_ranks():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: mean over the two ranks' losses
    torch.manual_seed(0)
    net = _Net()
    total = 0
    for rank in range(2):
        x = torch.arange(8, dtype=torch.float32).view(2, 4) + rank
        total = total + net(x, use_b=(rank == 0))
    (total / 2).backward()
    ref = {k: p.grad for k, p in net.named_parameters()}
    for rank, grads, ok, mx in got:
        assert ok is False and mx == 11.0
        for k in ref:
            assert torch.allclose(grads[k], ref[k], atol=1e-6), (rank, k)
