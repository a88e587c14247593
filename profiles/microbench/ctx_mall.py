#!/usr/bin/env python3
"""How much does the 256 MB Infinity Cache help the second pass over c_feats?  tcl_ctx_fwd on ONE buffer set, back to back,
for batches whose c_feats is 72 ... 287 MB: per-target time when the whole batch stays cache-resident vs when it does not."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from master_thesis_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
C, R, h, w, H, W = 128, 10, 30, 53, 240, 424
p = lambda t: ctypes.c_void_p(t.data_ptr())
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
flush = torch.zeros(1 << 28, dtype=torch.float32, device=dev)
for B in (4, 8, 16, 24, 28, 32, 48, 64):
    g = torch.Generator().manual_seed(B)
    cf = (0.05 * torch.randn(B, C, R + 1, h, w, generator=g)).to(dev)
    vt = (torch.rand(B, H // 8, W // 8, generator=g) < 0.6).float().repeat_interleave(8, 1).repeat_interleave(8, 2).to(dev)
    va = (torch.rand(B, R, H // 8, W // 8, generator=g) < 0.7).float().repeat_interleave(8, 2).repeat_interleave(8, 3).to(dev)
    p_in, c_mask, c_match = torch.empty(B, 2 * C + 1, h, w, device=dev), torch.empty(B, h, w, device=dev), torch.empty(B, R, h, w, device=dev)
    ws = torch.empty(lib.tcl_ctx_workspace_bytes(B, C, R, h, w), dtype=torch.uint8, device=dev)
    call = lambda: _lib.call("tcl_ctx_fwd", p(cf), p(vt), p(va), p(p_in), p(c_mask), p(c_match), B, C, R, h, w, H, W, 0, 0, p(ws), ws.numel(), sp)
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    e1.synchronize()
    warm = e0.elapsed_time(e1) / 20
    tot = 0.0
    for _ in range(10):
        flush.sum()
        e0.record()
        call()
        e1.record()
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    cold = tot / 10
    print("B=%2d c_feats %5.1f MB: back to back %.4f ms (%.2f us/target), after a 1 GiB read %.4f ms (%.2f us/target)" %
          (B, cf.numel() * 4 / 1e6, warm, warm / B * 1e3, cold, cold / B * 1e3), flush=True)
