import sys, ctypes, torch
sys.path.insert(0, '.')
from master_thesis_amd import _lib
lib = _lib.lib()
p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
dev = 'cuda:0'
def run(B, R, H, W, theta_src, vmode, nsets=6):
    g = torch.Generator().manual_seed(1)
    sets = []
    for _ in range(nsets):
        x = torch.rand(B, 3, R, H, W, generator=g).to(dev); v = (torch.rand(B, 1, R, H, W, generator=g) < 0.8).float().to(dev)
        th = (torch.tensor([[1.0, 0, 0], [0, 1.0, 0]]) + 0.05 * torch.randn(B * R, 2, 3, generator=g)).to(dev)
        grid = torch.nn.functional.affine_grid(th, (B * R, 1, H, W), align_corners=False).contiguous()
        xo = torch.empty(B, R, 3, H, W, device=dev); vo = torch.empty(B, R, 1, H, W, device=dev)
        sets.append((x, v, th, grid, xo, vo))
    def call(s):
        x, v, th, grid, xo, vo = s
        _lib.call("tcl_warp_fwd", p(x), x.stride(0), x.stride(2), x.stride(1), p(v), v.stride(0), v.stride(2), None if theta_src else p(grid), p(th) if theta_src else None,
                  p(xo), xo.stride(0), xo.stride(1), xo.stride(2), p(vo), vo.stride(0), vo.stride(1), B, R, 3, H, W, 0, vmode, None)
    for s in sets: call(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        for s in sets: call(s)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / (6 * nsets)
    npix = B * R * H * W
    nbytes = npix * (32 + (0 if theta_src else 8))
    print("%3dx%-3d x%4d  %s  vmode %d : %.4f ms  %.0f GB/s" % (H, W, B * R, "theta" if theta_src else "grid ", vmode, ms, nbytes / ms / 1e6), flush=True)
for (B, R, H, W) in ((32, 10, 240, 424), (32, 10, 256, 256), (32, 10, 256, 512), (32, 4, 256, 256), (32, 10, 240, 416)):
    for th in (True, False):
        for vm in (2, 1):
            run(B, R, H, W, th, vm)
