"""Test-mode instance clustering (models/cluster.py:15-111: sparse_quantize voxel-downsample + CPU DBSCAN).

NOT BUILT: it is SURVEY.md section 8f rank 1 ("next" row) -- a device->host->device round trip that only runs
in misc.mode == 'test' (models/motionnet.py:237-241).  The class exists so that MotionNet(cfg) constructs with
the reference's attribute names; calling it raises instead of silently doing something else."""
import torch.nn as nn


class Cluster(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        c = cfg['cluster']
        self.min_p_cluster = c['min_p_cluster']
        self.eps = c['eps_dbscan']
        self.min_samples = c['min_samples_dbscan']
        self.voxel_size = c['voxel_size']

    def forward(self, *args, **kwargs):
        raise NotImplementedError("misc.mode='test' needs the DBSCAN clustering step (models/cluster.py), which is "
                                  "outside the hot path built so far (SURVEY.md 8f rank 1); use mode 'train' or 'val'")
