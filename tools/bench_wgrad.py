"""Per-layer microbenchmark of sbgm_conv2d_wgrad_bias (wgrad + unpack + the two memsets of the non-pooled path)."""
import sys, torch
from sbgm_danra_amd import _native as N
L = N.lib()
def run(B, H, C, cout, k=3, stride=1, reps=20):
    pad = k // 2 if k != 8 else 3
    OH = (H + 2 * pad - k) // stride + 1
    dy = torch.randn(B, OH, OH, cout, device="cuda"); x = torch.randn(B, H, H, C, device="cuda")
    dw = torch.zeros(cout, C, k, k, device="cuda"); db = torch.zeros(cout, device="cuda"); ws = torch.zeros(k * k * cout * C, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def f():
        N.check(L.sbgm_conv2d_wgrad_bias(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, H, H, C, C, cout, k, k, stride, pad, st))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2 * B * OH * OH * C * cout * k * k
    print(f"B{B} H{H} C{C}->{cout} k{k}s{stride}: {us:.1f} us  {fl/us/1e6:.1f} TF", flush=True)
cfgs = [(8, 64, 64, 64), (8, 32, 128, 128), (8, 16, 256, 256), (8, 64, 128, 64), (8, 32, 256, 128), (8, 16, 512, 256), (8, 128, 64, 64), (8, 64, 128, 128), (32, 64, 64, 64), (32, 16, 256, 256)]
if len(sys.argv) > 1 and sys.argv[1] == "general":
    cfgs = [(8, 8, 512, 512), (8, 8, 1024, 512), (8, 16, 256, 512, 3, 2), (8, 32, 128, 256, 3, 2), (8, 64, 64, 128, 3, 2), (8, 16, 256, 512, 1, 2),
            (8, 16, 256, 768, 1, 1), (8, 16, 256, 256, 1, 1), (8, 8, 512, 1536, 1, 1), (8, 32, 256, 128, 1, 1), (8, 64, 128, 64, 1, 1)]
elif len(sys.argv) > 1:
    cfgs = cfgs[:int(sys.argv[1])]
for a in cfgs:
    run(*a)
