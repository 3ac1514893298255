"""Voxelization: host mirror of libs/voxel_generator.py:117-154 (class Voxelization), running on the GPU.

`Voxelization(cfg)(points)` takes the same numpy [N,4] (x,y,z,t) float32 array and returns the same dict of
numpy arrays (`coordinates` [M,4] int32 (z,y,x,t), `num_voxels` [1] int64, `shape` [4] int64 (nx,ny,nz,nt),
`point_to_voxel_map` [N,1] int32), bit-identical to the numba kernel.  `voxelize_device` is the same
computation for points that already live in HBM (no PCIe round trip; what bench.py uses).
"""
import numpy as np
import torch

from . import native


class Voxelization(object):
    def __init__(self, cfg):
        self.voxel_size = np.array(cfg['voxel_size'], dtype=np.float32)
        self.point_cloud_range = np.array(cfg['range'], dtype=np.float32)
        self.n_sweeps = cfg['n_sweeps']
        grid_size = (self.point_cloud_range[3:] - self.point_cloud_range[:3]) / self.voxel_size
        self.grid_size = np.round(grid_size).astype(np.int64)
        self.max_voxels = int(self.grid_size[0] * self.grid_size[1] * self.grid_size[2] * self.n_sweeps)
        self.device = torch.device('cuda')

    def voxelize_launch(self, points):
        """Asynchronous part of voxelize_device: (coords [max,4] i32, p2v [N] i32, num_voxels [1] i32 on device)."""
        return native.voxelize(points.contiguous(), self.voxel_size.tolist(), self.point_cloud_range.tolist(),
                               self.grid_size.tolist(), self.n_sweeps, self.max_voxels)

    def voxelize_device(self, points):
        """points [N,4] f32 cuda -> (coordinates [M,4] i32, point_to_voxel_map [N] i32, num_voxels int), on device."""
        coords, p2v, num = native.voxelize(points.contiguous(), self.voxel_size.tolist(), self.point_cloud_range.tolist(),
                                           self.grid_size.tolist(), self.n_sweeps, self.max_voxels)
        m = int(num.item())
        return coords[:m], p2v, m

    def __call__(self, points):
        pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(self.device)
        coords, p2v, m = self.voxelize_device(pts)
        return {
            'coordinates': coords.cpu().numpy(),
            'num_voxels': np.array([m], dtype=np.int64),
            'shape': np.hstack((self.grid_size, np.array([self.n_sweeps]))),
            'point_to_voxel_map': p2v.cpu().numpy()[:, None],
        }
