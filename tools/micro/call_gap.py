"""where does the fixed cost of a 20-step sampler call go?  host time inside the call, wall to completion, after idle gaps of different lengths"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
import sbgm_danra_amd as S
dev = torch.device("cuda", 0)
net = bench.build_model(dev); net.eval()
B, HW = 32, 128
cond = torch.randn(B, 1, HW, HW, device=dev)
net.autotune(B, HW, HW, cond_channels=(0, 0, 1), cache="profiles/r03_c2_tiles.txt")
def run(n):
    return S.Euler_Maruyama_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=B, num_steps=n, device=dev, img_size=HW, cond_img=cond, seed=3)
for _ in range(3): run(5)
torch.cuda.synchronize()
for idle in (0.0, 0.001, 0.01, 0.1, 0.5):
    hs, ws = [], []
    for _ in range(8):
        torch.cuda.synchronize(); time.sleep(idle)
        t0 = time.perf_counter(); run(20); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        hs.append((t1 - t0) * 1e3); ws.append((t2 - t0) * 1e3)
    hs.sort(); ws.sort()
    print(f"idle {idle*1e3:6.1f} ms before the call: host time in call {hs[len(hs)//2]:6.3f} ms, wall to completion {ws[len(ws)//2]:6.3f} ms = {ws[len(ws)//2]/20:.4f} ms/step (min {ws[0]/20:.4f})")
for n in (20, 40, 100):
    ws = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); ws.append((time.perf_counter() - t0) * 1e3)
    ws.sort(); print(f"{n} steps: {ws[len(ws)//2]:.3f} ms = {ws[len(ws)//2]/n:.4f} ms/step")
