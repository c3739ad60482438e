"""times the attention-sized linears: pipelined implicit GEMM tiles vs the preloaded-linear kernel (winograd bit 5), warm and after a 64 MiB fill"""
import ctypes as C, os, sys, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sbgm_danra_amd import _native as N
lib = N.lib()
dev = torch.device("cuda", 0)
junk = torch.empty(16 << 20, device=dev)
def bench(M, Cin, Cout, tile, wpt, bits, cold):
    x = torch.randn(1, 1, M, Cin, device=dev); w = torch.randn(Cout, Cin, 1, 1, device=dev) / math.sqrt(Cin)
    packed = torch.empty(lib.sbgm_conv_packed_numel(Cout, 1, 1, Cin), device=dev)
    N.check(lib.sbgm_conv_pack_weight(w.data_ptr(), packed.data_ptr(), Cout, Cin, 1, 1, Cin, N.stream()))
    out = torch.empty(1, 1, M, Cout, device=dev)
    a = N.ConvArgs(x.data_ptr(), packed.data_ptr(), out.data_ptr(), None, None, None, None, 1, 1, M, Cin, Cout, 1, 1, 1, 0, N.NONE, 0,
                   tile[0], tile[1], 0, wpt, bits, 0, 0, 0, None, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(30):
        if cold: junk.zero_()
        e0.record()
        for _ in range(1 if cold else 20): N.check(lib.sbgm_conv2d_fwd(C.byref(a), N.stream()))
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / (1 if cold else 20) * 1e3)
    ts.sort()
    return ts[len(ts) // 4]
for (M, Cin, Cout) in ((2048, 256, 256), (2048, 256, 768), (512, 512, 512), (512, 512, 1536), (1024, 512, 512), (512, 256, 256)):
    print(f"== M={M} {Cin}->{Cout}")
    for label, tile, wpt, bits in (("igemm 2x1 ws2", (2, 1), 2, 0), ("igemm 2x1 ws4", (2, 1), 4, 0), ("igemm 4x1 ws2", (4, 1), 2, 0), ("igemm 2x2 ws4", (2, 2), 4, 0),
                                   ("pre 2x1 ws4", (2, 1), 4, 32), ("pre 2x2 ws4", (2, 2), 4, 32), ("pre 4x1 ws4", (4, 1), 4, 32), ("pre 1x1 ws4", (1, 1), 4, 32), ("pre 1x2 ws4", (1, 2), 4, 32),
                                   ("pre 2x1 ws8", (2, 1), 8, 32), ("pre 2x2 ws8", (2, 2), 8, 32), ("pre 4x1 ws8", (4, 1), 8, 32), ("pre 1x2 ws8", (1, 2), 8, 32)):
        try:
            print(f"  {label:14s} warm {bench(M, Cin, Cout, tile, wpt, bits, False):6.2f} us   cold {bench(M, Cin, Cout, tile, wpt, bits, True):6.2f} us")
        except N.NativeError as e:
            print(f"  {label:14s} refused")
