"""One gloo rank of tests/test_distributed.py (launched as a plain subprocess: `python dist_worker.py rank world port out.pt`)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(4, 3)
        self.b = torch.nn.Linear(3, 2)          # only used when the input says so (data-dependent branch)

    def forward(self, x, use_b):
        y = self.a(x)
        return self.b(y).sum() if use_b else y.sum()


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
    from pcaccumulation_amd import distributed as pdist
    r, w, _ = pdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    net = Net()
    x = torch.arange(8, dtype=torch.float32).view(2, 4) + rank
    net(x, use_b=(rank == 0)).backward()
    assert (net.b.weight.grad is None) == (rank == 1)
    pdist.FlatGradAllReduce(net.parameters())()
    ok = pdist.all_ok(rank == 0, torch.device('cpu'))
    mx = pdist.max_over_ranks(10.0 + rank, torch.device('cpu'))
    torch.save({'grads': {k: p.grad.clone() for k, p in net.named_parameters()}, 'ok': ok, 'mx': mx}, out)
    pdist.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
