import torch, sys, os
sys.path.insert(0, os.getcwd())
from master_thesis_amd import _lib, _ops
from bench_ops import timed
DEV='cuda:0'
def run(B, R, H, W, rot, name):
    g = torch.Generator().manual_seed(1)
    xr = torch.rand(B, 3, R, H, W, generator=g).to(DEV)
    vr = (torch.rand(B, 1, R, H, W, generator=g) < 0.85).float().to(DEV)
    noise = torch.randn(B * R, 2, 3, generator=g)
    noise[:, :, :2] *= rot
    noise[:, :, 2] *= 0.05
    theta = (torch.tensor([[1.0, 0, 0], [0, 1.0, 0]]) + noise).to(DEV)
    npix = B * R * H * W
    ms = timed(lambda: _ops.WarpFn.apply(xr, vr, None, theta, False, _lib.VMODE_BILINEAR_THRESHOLD), 20)
    print('%-40s %.4f ms %.0f GB/s' % (name, ms, npix * 32 / ms / 1e6))
    ms = timed(lambda: _ops.WarpFn.apply(xr, vr, None, theta, False, _lib.VMODE_NEAREST), 20)
    print('%-40s %.4f ms %.0f GB/s' % (name + ' (v nearest)', ms, npix * 32 / ms / 1e6))
run(32, 10, 240, 424, 0.05, '240x424 rot 0.05')
run(32, 10, 240, 424, 0.0, '240x424 rot 0')
run(32, 10, 256, 512, 0.05, '256x512 rot 0.05')
run(32, 10, 256, 512, 0.0, '256x512 rot 0')
run(32, 10, 240, 424, 0.01, '240x424 rot 0.01')
