import torch
def bench(f, reps=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, cin, cout in [(2048, 256, 768), (2048, 256, 256), (512, 512, 1536), (512, 512, 512), (8192, 256, 128), (32768, 128, 64), (2048, 256, 512), (8192,128,256)]:
    x = torch.randn(M, cin, device="cuda"); dy = torch.randn(M, cout, device="cuda"); out = torch.empty(cout, cin, device="cuda")
    us = bench(lambda: torch.mm(dy.t(), x, out=out))
    print(f"M{M} {cin}->{cout}: mm {us:.1f} us  {2*M*cin*cout/us/1e6:.1f} TF", flush=True)
