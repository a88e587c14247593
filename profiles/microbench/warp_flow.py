"""Affine forward (theta source, bilinear visibility, 240x424 x 320) against the amount of rotation / shear in theta.
usage: warp_flow.py [sigma ...]   (sigma = std of the noise on the 2x2 part; the translation always gets N(0, 0.05))"""
import sys, ctypes, torch
sys.path.insert(0, '.')
from master_thesis_amd import _lib
lib = _lib.lib()
p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
dev = 'cuda:0'
B, R, H, W = 32, 10, 240, 424
def run(sigma, nsets=6, reps=6):
    g = torch.Generator().manual_seed(1)
    sets = []
    for _ in range(nsets):
        x = torch.rand(B, 3, R, H, W, generator=g).to(dev); v = (torch.rand(B, 1, R, H, W, generator=g) < 0.8).float().to(dev)
        n = torch.randn(B * R, 2, 3, generator=g)
        n[:, :, :2] *= sigma; n[:, :, 2] *= 0.05
        th = (torch.tensor([[1.0, 0, 0], [0, 1.0, 0]]) + n).to(dev)
        xo = torch.empty(B, R, 3, H, W, device=dev); vo = torch.empty(B, R, 1, H, W, device=dev)
        sets.append((x, v, th, xo, vo))
    def call(s):
        x, v, th, xo, vo = s
        _lib.call("tcl_warp_fwd", p(x), x.stride(0), x.stride(2), x.stride(1), p(v), v.stride(0), v.stride(2), None, p(th),
                  p(xo), xo.stride(0), xo.stride(1), xo.stride(2), p(vo), vo.stride(0), vo.stride(1), B, R, 3, H, W, 0, 2, None)
    for s in sets: call(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for s in sets: call(s)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / (reps * nsets)
    print("sigma %.3f : %.4f ms  %.0f GB/s" % (sigma, ms, B * R * H * W * 32 / ms / 1e6), flush=True)
for s in ([float(a) for a in sys.argv[1:]] or [0.0, 0.01, 0.02, 0.05, 0.1]):
    run(s)
