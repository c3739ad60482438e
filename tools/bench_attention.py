"""Microbenchmark (GPU): attention core at the BASELINE shapes, 10 launches per hipGraph, HIP events.  Run once as is and once with
SBGM_NO_LDS_ATTENTION=1 to compare the LDS-staged kernel with the register kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbgm_danra_amd import _native as N
L = N.lib()
for (B, S, C_, h) in ((32, 256, 128, 4), (16, 1024, 128, 4), (16, 256, 256, 4), (32, 64, 256, 4), (8, 256, 128, 4)):
    qkv = torch.randn(B, S, 3 * C_, device="cuda"); out = torch.empty(B, S, C_, device="cuda")
    f = lambda: N.check(L.sbgm_mha_core_fwd(qkv.data_ptr(), out.data_ptr(), B, S, C_, h, N.stream()))
    f()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(10): f()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"B={B:2d} S={S:4d} C={C_:3d} heads={h}: {us:7.1f} us  {4.0 * B * S * S * C_ / us / 1e6:6.1f} TFLOP/s", flush=True)
