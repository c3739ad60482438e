"""per-call wall time of a sequence of sampler calls with varying step counts (is the first call after a change of num_steps slower?)"""
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
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S.Euler_Maruyama_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=B, num_steps=n, device=dev, img_size=HW, cond_img=cond, seed=1234, use_graph=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for n in (5, 20, 20, 20, 7, 20, 20, 200, 20, 20, 20, 20, 5, 5, 20):
    ms = run(n)
    print(f"num_steps {n:4d}: {ms:8.3f} ms = {ms / n:.4f} ms/step")
