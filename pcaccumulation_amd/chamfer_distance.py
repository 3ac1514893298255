"""ChamferDistance module: host mirror of chamfer_distance/chamfer_distance.py:9-57.

Same class names and call convention (`ChamferDistance()(xyz1, xyz2) -> dist1, dist2`, squared distances,
gradients w.r.t. both clouds).  The reference JIT-builds a CUDA extension at import, allocates its outputs
on the host and copies them over on every call (chamfer_distance.py:16-28); here the outputs are allocated
on the device and the HIP kernels are called through the C ABI.
"""
import torch

from . import native


class ChamferDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1 = xyz1.contiguous().float()
        xyz2 = xyz2.contiguous().float()
        dist1, dist2, idx1, idx2 = native.chamfer_forward(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, _g3, _g4):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        return native.chamfer_backward(xyz1, xyz2, graddist1.contiguous().float(), graddist2.contiguous().float(), idx1, idx2)


class ChamferDistance(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        dist1, dist2, _, _ = ChamferDistanceFunction.apply(xyz1, xyz2)
        return dist1, dist2

    def forward_with_indices(self, xyz1, xyz2):
        """dist1, dist2, idx1, idx2 (the reference keeps the indices private in ctx)."""
        return ChamferDistanceFunction.apply(xyz1, xyz2)
