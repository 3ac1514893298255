"""N > 1 path on CPU: two gloo ranks (world_size 2), each a plain subprocess.

Checks that the flat-buffer all-reduce yields the gradient of the concatenated batch (mean over ranks), also when
a parameter has NO gradient on one rank (data-dependent branches of MotionNet, models/motionnet.py:222,243), and
that the "all ranks ok" flag and the max-over-ranks timing reduce correctly."""
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


def test_flat_allreduce_two_gloo_ranks(tmp_path):
    sys.path.insert(0, HERE)
    from dist_worker import Net
    port = str(_free_port())
    outs = [str(tmp_path / ('rank%d.pt' % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker.py'), str(r), '2', port, outs[r]]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    # single-process reference: mean over the two ranks' losses
    torch.manual_seed(0)
    net = Net()
    total = 0
    for rank in range(2):
        x = torch.arange(8, dtype=torch.float32).view(2, 4) + rank
        total = total + net(x, use_b=(rank == 0))
    (total / 2).backward()
    ref = {k: p.grad for k, p in net.named_parameters()}
    for rank in range(2):
        got = torch.load(outs[rank])
        assert got['ok'] is False and got['mx'] == 11.0
        for k in ref:
            assert torch.allclose(got['grads'][k], ref[k], atol=1e-6), (rank, k)
