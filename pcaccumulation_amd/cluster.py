"""Test-mode instance clustering: host mirror of models/cluster.py (Cluster) over the HIP path of include/pcacc.h C1.

The reference moves the points predicted moving to the host, voxel-down-samples them (torchsparse sparse_quantize),
runs scikit-learn DBSCAN in the horizontal plane sample by sample and copies the labels back (cluster.py:52-111).
Here the whole batch is clustered on the device in one call with the same labels (csrc/cluster.hip explains why the
result does not depend on DBSCAN's visiting order); nothing is read back, so mode='test' keeps the single host sync
of the forward pass plus one for the number of reconstructed points."""
import torch
import torch.nn as nn

from . import native


class Cluster(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        c = cfg['cluster']
        self.min_p_cluster = c['min_p_cluster']
        self.voxel_size = c['voxel_size']                  # kept like the reference; the down-sampling sizes are fixed below
        self.eps = c['eps_dbscan']
        self.min_samples = c['min_samples_dbscan']
        metric = c.get('cluster_metric', 'euclidean')
        if metric != 'euclidean':
            raise NotImplementedError("cluster.cluster_metric=%r: only 'euclidean' (configs/*.yaml) is built" % metric)

    def forward(self, transformed_points, mos, offset, time_indice, results, fb_labels=None, use_offset=True):
        """cluster.py:86-111.  transformed_points [N,3], mos [N] (1 = moving), offset [N,2], time_indice [N,2];
        writes results['inst_labels_est'] [N] i64 (0 = background / ignored)."""
        n_batches = results.get('_n_batches')
        if n_batches is None:
            n_batches = int(time_indice[:, 0].max()) + 1 if time_indice.size(0) else 1
        sel = ((fb_labels if fb_labels is not None else mos) == 1).reshape(-1).to(torch.uint8).contiguous()
        batch = time_indice[:, 0].to(torch.int32).contiguous()
        pts = transformed_points.detach().float().contiguous()
        off = offset.detach().float().contiguous() if use_offset else None
        results['inst_labels_est'] = native.cluster(pts, off, sel, batch, n_batches, 0.05 if use_offset else 0.15,
                                                    self.eps, self.min_samples, self.min_p_cluster)
