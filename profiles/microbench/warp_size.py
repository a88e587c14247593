import sys, torch
sys.path.insert(0, '.')
from master_thesis_amd import _lib, _ops
DEV = 'cuda:0'
def timed(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / iters
R, H, W = 4, 256, 256
ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
ident = torch.stack([xs, ys], -1)
for B in (8, 16, 32, 64, 128, 256):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 3, R, H, W, generator=g).to(DEV)
    v = (torch.rand(B, 1, R, H, W, generator=g) < 0.85).float().to(DEV)
    theta = torch.randn(B, R, 1, 1, 2, generator=g) * 0.15
    f2 = (ident[None, None] * (1 + 0.1 * torch.randn(B, R, 1, 1, 1, generator=g)) + theta).to(DEV).reshape(-1, H, W, 2)
    npix = B * R * H * W
    ms = timed(lambda: _ops.WarpFn.apply(x, v, f2, None, True, _lib.VMODE_NEAREST))
    ms2 = timed(lambda: _ops.WarpFn.apply(x, None, f2, None, True, _lib.VMODE_NONE))
    print('B=%3d  x+v: %6.1f MB %.4f ms %.2f TB/s | x only: %6.1f MB %.4f ms %.2f TB/s' % (B, npix * 40 / 1e6, ms, npix * 40 / ms / 1e9, npix * 32 / 1e6, ms2, npix * 32 / ms2 / 1e9))
