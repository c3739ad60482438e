"""one process, one GPU: a batch of 32 sampled as ONE chain of B = 32 vs as TWO concurrent chains of B = 16 (two model handles, each replaying its
step graph on its own private stream; a sampler call only enqueues, so one host thread feeds both)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
import sbgm_danra_amd as S
dev = torch.device("cuda", 0)
HW, N = 128, 60
def mk(B):
    net = bench.build_model(dev); net.eval()
    cond = torch.randn(B, 1, HW, HW, device=dev)
    net.autotune(B, HW, HW, cond_channels=(0, 0, 1))
    return net, cond
def run(net, cond, n=N):
    return S.Euler_Maruyama_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=cond.shape[0], num_steps=n, device=dev, img_size=HW,
                                    cond_img=cond, seed=3, use_graph=True)
a32 = mk(32)
a16, b16 = mk(16), mk(16)
a8 = [mk(8) for _ in range(4)]
for _ in range(2):
    run(*a32, 100); run(*a16, 100); run(*b16, 100)
    for m in a8: run(*m, 30)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(4)]
def timed(chains, reps=5):
    # every chain is called from its OWN caller stream: the call fences its private graph stream against the caller's stream on both
    # sides, so chains issued from one stream would run one after the other
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for c, st in zip(chains, streams):
            with torch.cuda.stream(st):
                run(*c)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best / N * 1e3
print(f"1 chain  x B=32: {timed([a32]):.4f} ms per step of 32 samples")
print(f"2 chains x B=16: {timed([a16, b16]):.4f} ms per step of 32 samples")
print(f"4 chains x B=8 : {timed(a8):.4f} ms per step of 32 samples")
print(f"1 chain  x B=16: {timed([a16]):.4f} ms per step of 16 samples")
