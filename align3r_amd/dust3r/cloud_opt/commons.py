"""Small helpers of the global aligner (dust3r/cloud_opt/commons.py semantics)."""
import numpy as np
import torch

from ...aligner import cosine_schedule, linear_schedule  # noqa: F401  (commons.py:123-130)


def edge_str(i, j):
    return f'{i}_{j}'


def get_conf_trf(mode):          # commons.py:42-55
    if mode == 'log':
        return lambda x: x.log()
    if mode == 'sqrt':
        return lambda x: x.sqrt()
    if mode == 'm1':
        return lambda x: x - 1
    if mode in ('id', 'none'):
        return lambda x: x
    raise ValueError(f'bad mode for {mode=}')


def signed_log1p(x):             # commons.py:113-115
    return torch.sign(x) * torch.log1p(torch.abs(x))


def signed_expm1(x):             # commons.py:118-120
    return torch.sign(x) * torch.expm1(torch.abs(x))


def get_imshapes(edges, pred_i, pred_j):     # commons.py:27-39
    n_imgs = max(max(e) for e in edges) + 1
    imshapes = [None] * n_imgs
    for e, (i, j) in enumerate(edges):
        shape_i, shape_j = tuple(pred_i[e].shape[0:2]), tuple(pred_j[e].shape[0:2])
        if imshapes[i]:
            assert imshapes[i] == shape_i, f'incorrect shape for image {i}'
        if imshapes[j]:
            assert imshapes[j] == shape_j, f'incorrect shape for image {j}'
        imshapes[i], imshapes[j] = shape_i, shape_j
    return imshapes


def unitquat_to_rotmat(q):
    """XYZW unit quaternion -> rotation matrix (closed form used where the reference calls roma, base_opt.py:188)."""
    x, y, z, w = q.unbind(-1)
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    rows = [1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)]
    return torch.stack(rows, -1).reshape(q.shape[:-1] + (3, 3))


def rotmat_to_unitquat(R):
    """Rotation matrix -> XYZW unit quaternion (stands in for roma.rotmat_to_unitquat, base_opt.py:203):
    the numerically stable largest-component branch selection."""
    R = torch.as_tensor(R, dtype=torch.float64)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = R.reshape(9).tolist()
    tr = m00 + m11 + m22
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = ((m21 - m12) / s, (m02 - m20) / s, (m10 - m01) / s, 0.25 * s)
    elif m00 > m11 and m00 > m22:
        s = np.sqrt(1.0 + m00 - m11 - m22) * 2
        q = (0.25 * s, (m01 + m10) / s, (m02 + m20) / s, (m21 - m12) / s)
    elif m11 > m22:
        s = np.sqrt(1.0 + m11 - m00 - m22) * 2
        q = ((m01 + m10) / s, 0.25 * s, (m12 + m21) / s, (m02 - m20) / s)
    else:
        s = np.sqrt(1.0 + m22 - m00 - m11) * 2
        q = ((m02 + m20) / s, (m12 + m21) / s, 0.25 * s, (m10 - m01) / s)
    return torch.tensor(q, dtype=torch.float32)
