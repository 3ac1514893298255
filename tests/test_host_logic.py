"""Host-side pieces that replace a third-party call of the reference, pinned against that call (CPU)."""
import numpy as np
import torch

from pcaccumulation_amd.tpointnet import mat2quat


def test_mat2quat_is_scipy():
    """models/tpointnet.py:62-66 calls scipy's Rotation.from_matrix(...).as_quat() on the host; the device restatement must
    give the same quaternion (same branch, same sign) for random rotations and the 180-degree / identity edge cases."""
    from scipy.spatial.transform import Rotation as R
    mats = np.concatenate([R.random(3000, random_state=7).as_matrix(),
                           R.from_euler('z', [179.999, 180, -180, 0, 1e-9, 90, -90], degrees=True).as_matrix(),
                           R.from_euler('xyz', [[180, 0, 0], [0, 180, 0], [179.9, 0.1, 0], [0, 0, 0]], degrees=True).as_matrix()])
    want = R.from_matrix(mats).as_quat()
    got = mat2quat(torch.from_numpy(mats)).numpy()
    assert got.dtype == np.float64 and np.abs(got - want).max() < 1e-15
    got32 = mat2quat(torch.from_numpy(mats.astype(np.float32))).numpy()            # float32 poses in, float64 out
    want32 = R.from_matrix(mats.astype(np.float32).astype(np.float64)).as_quat()
    # scipy re-orthogonalises matrices that are only orthogonal to float32 precision: agreement to that precision
    assert np.abs(got32 - want32).max() < 1e-7
