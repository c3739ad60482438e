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


def graph_time(f, n=10):
    f()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): f()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


# the per-token halves of the block: sbgm_attn_qkv_fwd (LN1 + in_proj) and sbgm_attn_tail_fwd (out_proj + LN2 + FF)
for (M, C_) in ((8192, 128), (2048, 256), (16384, 128), (4096, 256), (2048, 128)):
    x, att, out = torch.randn(M, C_, device="cuda"), torch.randn(M, C_, device="cuda"), torch.empty(M, C_, device="cuda")
    qkv = torch.empty(M, 3 * C_, device="cuda")
    vec = lambda n: torch.randn(n, device="cuda")
    def packed(co):
        w = torch.randn(co, C_, device="cuda") / C_ ** 0.5
        p = torch.empty(L.sbgm_conv_packed_numel(co, 1, 1, C_), device="cuda")
        N.check(L.sbgm_conv_pack_weight(w.data_ptr(), p.data_ptr(), co, C_, 1, 1, C_, N.stream()))
        return p
    g1, b1, win, bin_ = vec(C_), vec(C_), packed(3 * C_), vec(3 * C_)
    wo, bo, g2, b2, w1, bb1, w2, bb2 = packed(C_), vec(C_), vec(C_), vec(C_), packed(C_), vec(C_), packed(C_), vec(C_)
    t_in = graph_time(lambda: N.check(L.sbgm_attn_qkv_fwd(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), win.data_ptr(), bin_.data_ptr(),
                                                           qkv.data_ptr(), M, C_, 1e-5, N.stream())))
    t_out = graph_time(lambda: N.check(L.sbgm_attn_tail_fwd(att.data_ptr(), x.data_ptr(), wo.data_ptr(), bo.data_ptr(), g2.data_ptr(),
                                                            b2.data_ptr(), w1.data_ptr(), bb1.data_ptr(), w2.data_ptr(), bb2.data_ptr(),
                                                            out.data_ptr(), M, C_, 1e-5, N.stream())))
    fl = 6.0 * M * C_ * C_
    print(f"tokens={M:5d} C={C_:3d}: qkv {t_in:6.1f} us {fl / t_in / 1e6:5.1f} TF   tail {t_out:6.1f} us {fl / t_out / 1e6:5.1f} TF", flush=True)
