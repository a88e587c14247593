import ctypes, torch, sys, os
sys.path.insert(0, '.')
from master_thesis_amd import _lib
lib = _lib.lib(); dev = 'cuda:0'
precs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '0').split(',')]
def ev(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it
B, C, F, S = 16, 256, 1, 64; P = S * S
xt = torch.randn(B, C, P, device=dev); xr = torch.randn(B, C, F, P, device=dev)
G = torch.randn(B, F, P, P, device=dev)
out = torch.empty(B, F, P, P, device=dev); gt = torch.empty_like(xt); gr = torch.empty_like(xr)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for prec in precs:
    ws = torch.empty(lib.tcl_corr4d_fwd_workspace_bytes(B, C, F, P, prec), dtype=torch.uint8, device=dev)
    ws2 = torch.empty(lib.tcl_corr4d_bwd_workspace_bytes(B, C, F, P, prec), dtype=torch.uint8, device=dev)
    _lib.call("tcl_corr4d_fwd", p(xt), None, p(xr), None, p(out), B, C, F, P, prec, p(ws), ws.numel(), None)
    tf = ev(lambda: _lib.call("tcl_corr4d_contract", p(ws), p(out), B, C, F, P, prec, None))
    ta = ev(lambda: _lib.call("tcl_corr4d_bwd", p(G), p(xt), None, p(xr), None, p(ws), p(gt), None, B, C, F, P, prec, p(ws2), ws2.numel(), None))
    tr = ev(lambda: _lib.call("tcl_corr4d_bwd", p(G), p(xt), None, p(xr), None, p(ws), None, p(gr), B, C, F, P, prec, p(ws2), ws2.numel(), None))
    fl = 2.0 * P * P * C * B * F / 1e9
    print('%-28s prec %d: fwd gemm %.3f ms (%.0f TF)  bwd d_a(+norm) %.3f  bwd d_r(+norm) %.3f' % (os.path.basename(_lib.LIB_PATH), prec, tf, fl / tf, ta, tr))
