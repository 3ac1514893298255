"""Chamfer golden vectors from the reference's own C++ CPU path (oracle/_ref, built by `make -C oracle ref`)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_chamfer  # noqa: E402


def gen_chamfer(save):
    rng = np.random.RandomState(12)
    b, n, m = 2, 700, 450
    x1 = rng.uniform(-5, 5, (b, n, 3)).astype(np.float32)
    x2 = rng.uniform(-5, 5, (b, m, 3)).astype(np.float32)
    x2[:, 9] = x2[:, 4]                 # duplicated target: the lower index must win
    x2[:, m - 1] = x2[:, 4]
    x1[:, 0] = x2[:, 4]                 # exact hit (distance 0) on the duplicated target
    x1[:, 11] = x1[:, 3]                # duplicated query
    t1, t2 = torch.from_numpy(x1), torch.from_numpy(x2)
    d1, d2, i1, i2 = ref_chamfer.forward(t1, t2)
    gd1 = torch.from_numpy(rng.randn(b, n).astype(np.float32))
    gd2 = torch.from_numpy(rng.randn(b, m).astype(np.float32))
    g1, g2 = ref_chamfer.backward(t1, t2, gd1, gd2, i1, i2)
    save('chamfer', xyz1=x1, xyz2=x2, dist1=d1.numpy(), dist2=d2.numpy(), idx1=i1.numpy(), idx2=i2.numpy(),
         grad_dist1=gd1.numpy(), grad_dist2=gd2.numpy(), grad_xyz1=g1.numpy(), grad_xyz2=g2.numpy())
