"""Fixed cost of one sampler call (Python wrapper + native setup) next to its per-step time: wall time of calls with 2, 20, 100 steps
and a cProfile of the 2-step call."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cProfile, pstats
import bench
import sbgm_danra_amd as S
dev = torch.device("cuda", 0)
net = bench.build_model(dev)
net.eval()
B, HW = 32, 128
cond = torch.randn(B, 1, HW, HW, device=dev)
net.autotune(B, HW, HW, cond_channels=(0, 0, 1))
def run(n):
    return S.Euler_Maruyama_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=B, num_steps=n, device=dev, img_size=HW, cond_img=cond, seed=3)
for _ in range(3): run(5)
torch.cuda.synchronize()
for n in (2, 20, 100):
    t0 = time.perf_counter()
    for _ in range(10): run(n)
    torch.cuda.synchronize()
    print(f"num_steps={n}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): run(2)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
