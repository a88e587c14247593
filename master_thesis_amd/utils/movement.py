"""Drop-in for the reference's utils/movement.py: synthetic camera movement, on the GPU.

`MovementSimulator.simulate_movement` (utils/movement.py:23-61) turns one frame into an n-frame sequence by warping it
with n stacked affine transforms and returns the inverse movement as a sampling grid.  The 3x3 matrix chain stays on the
host (tiny float32 LAPACK inverses and matmuls in the reference's order, batched over the sequences; the thetas agree
with the reference's to the last bit or two, the elementwise theta conversion exactly); the two image-sized steps run
through the C ABI:

  * `F.grid_sample(x.repeat(n), affine_grid(theta))`  -> tcl_warp_fwd(theta=..., align_corners=1) reading the ONE source
    frame through a zero frame stride (no n-fold copy, no grid in HBM);
  * `F.affine_grid(theta_inv)`                         -> tcl_affine_grid.

`simulate_movement_batch` is the data-parallel form: B independent frames (one per training sequence) in one launch.

GPU-only contract: the frame `x` must already be on the HIP device (the reference calls simulate_movement on CPU tensors
inside its dataset; `install_dropin()` therefore does NOT replace the reference's `utils.movement`, and a CPU frame raises a
TclError that says so).  Do not call this from forked DataLoader workers: a forked child cannot use the parent's HIP context;
simulate the movement in the training process after the batch has been moved to the device (simulate_movement_batch).
The matrix chain differs from the reference in one documented way: the inverses come from one batched float32
`torch.linalg.inv` instead of per-matrix `np.linalg.inv` (same LAPACK routine; `tests/golden/movement.npz` pins the result).
skimage is not needed: `AffineTransform(translation, scale, rotation).params` is restated from its published formula
(the library is absent from this image, so that one matrix constructor is pinned by the formula, not by skimage).
"""
import math

import numpy as np
import torch

from .. import _lib, _ops


class MovementSimulator:

    def __init__(self, max_displacement, max_scaling, max_rotation):
        self.max_displacement = max_displacement
        self.max_scaling = max_scaling
        self.max_rotation = max_rotation

    def random_affine(self):
        """One random (3,3) float32 transform; consumes numpy's global RNG in the reference's order (utils/movement.py:15-21)."""
        d = self.max_displacement
        tx, ty = np.random.randint(low=-d, high=d, size=2) if d > 0 else (0, 0)
        sx, sy = np.random.uniform(low=1 - self.max_scaling, high=1 + self.max_scaling, size=2)
        rot = np.random.uniform(low=-self.max_rotation, high=self.max_rotation)
        return torch.from_numpy(affine_params((tx, ty), (sx, sy), rot)).float()

    def simulate_movement(self, x, n, affine_matrices=None):
        """x (C,H,W) on the GPU -> (y (C,n,H,W), flow_inv (n,H,W,2), affine_matrices).  utils/movement.py:23-61."""
        c, h, w = x.size()
        y, flow_inv, mats = self.simulate_movement_batch(x.unsqueeze(0), n, None if affine_matrices is None else [affine_matrices])
        return y[0], flow_inv[0], mats[0]

    def simulate_movement_batch(self, x, n, affine_matrices=None):
        """x (B,C,H,W) -> (y (B,C,n,H,W), flow_inv (B,n,H,W,2), [B lists of n (3,3) matrices])."""
        if not x.is_cuda:
            raise _lib.TclError("MovementSimulator runs on the GPU only: move the frame to the HIP device first (x.to('cuda')); for CPU "
                                "tensors inside a DataLoader worker keep the reference's own utils.movement")
        b, c, h, w = x.size()
        if affine_matrices is None:
            affine_matrices = []
            for _ in range(b):
                mats = [self.random_affine() for _ in range(n - 1)]
                affine_matrices.append(mats[:n // 2] + [MovementSimulator.identity_affine()] + mats[n // 2:])
        mats = torch.stack([torch.stack(list(seq)) for seq in affine_matrices]).float()        # (B,n,3,3), host
        theta = affines2theta(stack_transformations_batch(mats, n // 2).reshape(b * n, 3, 3), h, w).to(x.device)
        theta_inv = affines2theta(stack_transformations_batch(torch.linalg.inv(mats), n // 2).reshape(b * n, 3, 3), h, w).to(x.device)
        # every frame of a sequence samples the same source frame: frame stride 0
        y, _ = _ops.WarpFn.apply(x.unsqueeze(2).expand(b, c, n, h, w), None, None, theta, True, _lib.VMODE_NONE)
        flow_inv = _ops.AffineGridFn.apply(theta_inv, h, w, True).reshape(b, n, h, w, 2)
        return y, flow_inv, affine_matrices

    @staticmethod
    def identity_affine():
        return torch.from_numpy(np.linalg.inv(affine_params((0, 0), (1, 1), 0))).float()

    @staticmethod
    def identity_affine_theta(h, w):
        return MovementSimulator.affine2theta(MovementSimulator.identity_affine(), h, w)

    @staticmethod
    def affine_inverse(affine):
        return torch.from_numpy(np.linalg.inv(np.asarray(affine)))

    @staticmethod
    def transform_single(image, flow):
        """image (C,H,W) warped by one (3,3) transform with torch's default align_corners=False.  utils/movement.py:76-81."""
        c, h, w = image.size()
        theta = MovementSimulator.affine2theta(flow, h, w).unsqueeze(0).to(image.device)
        y, _ = _ops.WarpFn.apply(image.reshape(1, c, 1, h, w), None, None, theta, False, _lib.VMODE_NONE)
        return y[0, :, 0]

    @staticmethod
    def stack_transformations(affine_matrices, t):
        """Per-frame transforms (n,3,3) -> transforms relative to frame t: frames before t chain inverses backwards,
        frames after t chain forwards.  utils/movement.py:83-96."""
        return stack_transformations_batch(affine_matrices.float().unsqueeze(0), t)[0]

    @staticmethod
    def affine2theta(param, h, w):
        """Pixel-space (3,3) transform -> the (2,3) theta of affine_grid's [-1,1] frame.  utils/movement.py:98-107."""
        return affines2theta(torch.as_tensor(param).unsqueeze(0), h, w)[0]


def stack_transformations_batch(mats, t):
    """(B,n,3,3) float32 on the host -> the same chain for B sequences at once (2n batched 3x3 ops instead of 2n per
    sequence: the per-sequence form costs ~0.2 ms of host time per sequence, more than the device work of a whole batch)."""
    n = mats.size(1)
    stacked = torch.zeros_like(mats)
    stacked[:, t] = mats[:, t]
    inv = torch.linalg.inv(mats[:, :t]) if t > 0 else None
    for i in range(t - 1, -1, -1):
        stacked[:, i] = torch.matmul(inv[:, i], stacked[:, i + 1])
    for i in range(t + 1, n):
        stacked[:, i] = torch.matmul(mats[:, i], stacked[:, i - 1])
    return stacked


def affine_params(translation, scale, rotation):
    """skimage.transform.AffineTransform(translation=, scale=, rotation=).params (shear 0), float64."""
    (tx, ty), (sx, sy) = translation, scale
    return np.array([[sx * math.cos(rotation), -sy * math.sin(rotation), tx],
                     [sx * math.sin(rotation), sy * math.cos(rotation), ty],
                     [0, 0, 1]], dtype=np.float64)


def affines2theta(params, h, w):
    """(n,3,3) float32 -> (n,2,3) float32, elementwise in the reference's operation order (float32 scalar arithmetic,
    widened to float64 only for storage, so rounding back to float32 is the identity)."""
    p = params.float()
    theta = torch.zeros((p.size(0), 2, 3), dtype=torch.float32)
    theta[:, 0, 0] = p[:, 0, 0]
    theta[:, 0, 1] = p[:, 0, 1] * h / w
    theta[:, 0, 2] = p[:, 0, 2] * 2 / w + p[:, 0, 0] + p[:, 0, 1] - 1
    theta[:, 1, 0] = p[:, 1, 0] * w / h
    theta[:, 1, 1] = p[:, 1, 1]
    theta[:, 1, 2] = p[:, 1, 2] * 2 / h + p[:, 1, 0] + p[:, 1, 1] - 1
    return theta
