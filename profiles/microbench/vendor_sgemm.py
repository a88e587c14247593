"""Reference point for the fp32 contraction: torch.bmm (rocBLAS / hipBLASLt SGEMM) on the C2 shapes, same box, same run as
the library's own kernels (profiles/microbench/gemm_ab.py).  Not part of the product: the product never calls a vendor GEMM."""
import torch
dev = 'cuda:0'
def ev(fn, it=20):
    for _ in range(it): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it
B, C, P = 16, 256, 4096
a = torch.randn(B, C, P, device=dev); r = torch.randn(B, C, P, device=dev); G = torch.randn(B, P, P, device=dev)
out = torch.empty(B, P, P, device=dev); d = torch.empty(B, C, P, device=dev)
fl = 2.0 * P * P * C * B / 1e9
for name, fn in [("fwd  a^T r      (P x P, K=C)", lambda: torch.bmm(a.transpose(1, 2), r, out=out)),
                 ("bwd  r G^T      (C x P, K=P)", lambda: torch.bmm(r, G.transpose(1, 2), out=d)),
                 ("bwd  a G        (C x P, K=P)", lambda: torch.bmm(a, G, out=d))]:
    ms = ev(fn)
    print('torch.bmm fp32 %s: %.3f ms  %.1f TFLOP/s  (%.3f of 157.3)' % (name, ms, fl / ms, fl / ms / 157.3), flush=True)
