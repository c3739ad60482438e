"""Microbenchmark (GPU): the decoder's conv_up layers at the C2 shape, plain LDS-Winograd kernel on a pre-upsampled input vs the
fused upsample-on-load mode (in_mode 2) and the affine-on-load mode (in_mode 1), every tile, HIP events around back-to-back launches."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbgm_danra_amd import _native as N
L = N.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
VERBOSE = len(sys.argv) > 2
def run(Cin, Cout, H, reps=10):
    h = H // 2
    xlo = torch.randn(B, h, h, Cin, device="cuda"); xhi = torch.randn(B, H, H, Cin, device="cuda")
    skip = torch.randn(B, h, h, Cin, device="cuda"); aff_lo = torch.randn(B * Cin * 2, device="cuda"); 
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; bias = torch.randn(Cout, device="cuda")
    packed = torch.empty(L.sbgm_conv_wino_packed_numel(Cout, Cin), device="cuda")
    N.check(L.sbgm_conv_wino_pack_weight(w.data_ptr(), packed.data_ptr(), Cout, Cin, Cin, N.stream()))
    packed2 = torch.empty(L.sbgm_conv_wino2d_packed_numel(Cout, Cin), device="cuda")
    N.check(L.sbgm_conv_wino2d_pack_weight(w.data_ptr(), packed2.data_ptr(), Cout, Cin, Cin, N.stream()))
    out = torch.empty(B, H, H, Cout, device="cuda")
    fl = 2.0 * B * H * H * Cin * Cout * 9
    res = {}
    for mode, x, aff, sk, act in ((0, xhi, None, None, 0), (1, xhi, aff_lo, None, 0), (2, xlo, None, None, 0), (2, xlo, aff_lo, skip, N.SILU)):
        best = None
        best2 = None
        for tile in ((4, 1), (4, 2), (2, 1), (2, 2), (1, 1), (1, 2), ("2d", 1, 1), ("2d", 2, 1), ("2d", 2, 2)):
            w2d = tile[0] == "2d"
            if Cout % (16 * (tile[1] if w2d else tile[0])): continue
            for db in (0, 4):
                if w2d:
                    a = N.ConvArgs(x.data_ptr(), packed2.data_ptr(), out.data_ptr(), None, bias.data_ptr(), None, None, B, H, H, Cin, Cout, 3, 3, 1, 1,
                                   0, 0, tile[1], 0, 0, tile[2], 8 | db, 0, 0, 0, None, 0, mode, N.ptr(aff), N.ptr(sk), act)
                else:
                    a = N.ConvArgs(x.data_ptr(), packed.data_ptr(), out.data_ptr(), None, bias.data_ptr(), None, None, B, H, H, Cin, Cout, 3, 3, 1, 1,
                                   0, 0, tile[0], tile[1], 0, 0, 3 | db, 0, 0, 0, None, 0, mode, N.ptr(aff), N.ptr(sk), act)
                try:
                    for _ in range(2): N.check(L.sbgm_conv2d_fwd(C.byref(a), N.stream()))
                except N.NativeError:
                    continue
                e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(reps): N.check(L.sbgm_conv2d_fwd(C.byref(a), N.stream()))
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / reps * 1e3
                if VERBOSE: print(f"    mode{mode} {tile} db={db}: {us:7.1f} us {fl / us / 1e6:6.1f} TF", flush=True)
                if w2d:
                    if best2 is None or us < best2[0]: best2 = (us, tile, db)
                elif best is None or us < best[0]: best = (us, tile, db)
        res[(mode, sk is not None)] = (best, best2)
    fmt = lambda b: f"{b[0]:6.1f}us {fl / b[0] / 1e6:5.1f}TF {b[1]}{'db' if b[2] else ''}"
    print(f"{Cin:4d}->{Cout:4d} @{H:3d}^2:", flush=True)
    for (m, s), (b, b2) in res.items():
        print(f"   mode{m}{'+pre' if s else '    '}: 1-D {fmt(b)}   2-D {fmt(b2)}   ratio {b[0] / b2[0]:.2f}", flush=True)
for cfg in ((64, 64, 128), (64, 64, 64), (64, 64, 32), (128, 128, 32), (128, 128, 16), (256, 256, 16), (128, 64, 32), (256, 128, 16)):
    run(*cfg)
