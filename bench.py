#!/usr/bin/env python3
"""bench.py — reverse-SDE sampling throughput of the native score-UNet path on MI355X.

Workload (BASELINE.json configs[1]): 128x128 target, 1 LR condition (C_in = 2), batch 32 per GPU, VE-SDE
Euler-Maruyama sampling; one "step" = one SDE step = one network evaluation + the fused update over the batch.
Metric: denoising steps / s = batch x SDE steps / wall-second, whole job (sum over ranks; each rank owns whole
independent batches, no data-path collective => "weak" scaling).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     : the implicit-GEMM convolution kernel family (dominant: ~95 % of evaluation time), algorithmic
                 FLOPs (2*M*Cout*K) / summed launch durations measured with HIP events on the launch
                 stream, against the fp32 matrix/vector peak (157.3 TFLOP/s, MI355X_MICROARCH.md).
  cpu_baseline : the CPU oracle (oracle/torch_ref.py, kind "port") timed on this box's host cores on a bounded
                 sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3          # MI355X fp32 vector == fp32 MFMA peak (guide: chip-level parameters)
PEAK_HBM_GBS = 8000.0
FLOP_PER_SAMPLE_128 = 5.146e9     # SURVEY.md §8d, 128x128, C_in = 2
CLOCK_WARM_S = 0.15               # continuous sampling the device needs before its clocks are steady (measured, tools/micro/call_seq.py)
BYTES_PER_EVAL = lambda b, hw: 76.2e6 + 29.2e6 * b * (hw / 128.0) ** 2   # noqa: E731  layer-fused model, SURVEY §8d


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a torchrun environment: start N rank processes (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set as torch.distributed.run would) BEFORE this process touches the GPU, relay rank 0's JSON line and
    exit with the worst child status.  The children are fresh interpreters started as child processes — nothing execs over a
    process that has initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def kernel_source_hash():
    """sha256 over the kernel sources + the C-ABI header (16 hex digits).  The PMC traffic files under profiles/ record the hash
    of the tree they were measured on; a file measured on other kernels is refused (the GPU box has no .git to ask for HEAD)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "sbgm_danra_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".cpp")))
    for f in files + [os.path.join(ROOT, "include", "sbgm_hip.h")]:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def build_model(dev, n_cond=1):
    import sbgm_danra_amd as S
    torch.manual_seed(42)
    enc = S.Encoder(n_cond, 256, block_layers=[2, 2, 2, 2], n_heads=4)
    dec = S.Decoder(512, 1, 256, n_heads=4, norm="group", gn_groups=8, activation=nn.SiLU)
    net = S.ScoreNet(S.marginal_prob_std_fn, enc, dec, device=dev, debug_pre_sigma_div=False)
    with torch.no_grad():                                  # reference training init (training.py:188-201)
        for m in net.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    m.bias.fill_(0.01)
    net.eval()
    return net


def usable_cores():
    """Every core this process can actually run on: the affinity mask, capped by the cgroup CPU quota when there is one (a GPU box
    shows all 256 logical CPUs of the host in the mask but schedules a one-GPU job on a 16-core share: 256 threads on that share
    ran the oracle 100x slower than 16).  Returns (threads, how they were counted)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    how = f"affinity mask {n}"
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except Exception:
            pass
    if quota is not None and quota < n:
        import math
        n = max(1, int(math.ceil(quota)))
        how += f", cgroup CPU quota {quota:.1f} -> {n}"
    return n, how


def cpu_baseline(batch, hw, budget_s=20.0, sampler="em", n_cond=1, max_steps=50):
    """Oracle on the host cores: 1 warm-up + as many SDE steps as fit the budget (at least 2).  sampler "pc" = 2 evaluations per step
    (corrector + predictor, reference score_sampling.py:136-230), "em" = 1."""
    from oracle import torch_ref as O
    cores, how = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    ora = O.build_scorenet(n_cond).eval()
    x = torch.randn(batch, 1, hw, hw)
    c = torch.randn(batch, n_cond, hw, hw) if n_cond else None
    t = torch.full((batch,), 0.5)
    evals = 2 if sampler == "pc" else 1
    with torch.no_grad():
        ora(x, t, cond_img=c)
        n, t0 = 0, time.perf_counter()
        while n < 2 or (time.perf_counter() - t0) < budget_s * 0.6:
            for _ in range(evals):
                s = ora(x, t, cond_img=c)
                x = x + 1e-3 * s + 0.03 * torch.randn_like(x)
            n += 1
            if n >= max_steps:
                break
        dt = time.perf_counter() - t0
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    name = "Euler-Maruyama" if sampler == "em" else "predictor-corrector"
    return {"value": batch * n / dt, "unit": "denoising steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} {name} steps of the CPU oracle at batch {batch}, {hw}x{hw}, after 1 warm-up "
                      f"({dt:.1f} s, PyTorch-CPU {torch.__version__}, {model}; threads = {how})"}


def sampling_secondary(dev, B, HW, sampler_key, steps, warmup, n_cond=1, tune=True):
    """another BASELINE sampling configuration beside the headline (C4: 256x256, batch 16, predictor-corrector; C1: 64x64, one
    sample, no condition): same timing rules as the headline, its own autotune and its own roofline block.  tune=False: the engine's
    static kernel choice, i.e. what a caller of the public samplers gets without ever calling ScoreNet.autotune"""
    import sbgm_danra_amd as S
    net = build_model(dev, n_cond=n_cond)
    g = torch.Generator().manual_seed(42)
    cond = torch.randn(B, n_cond, HW, HW, generator=g).to(dev) if n_cond else None
    if tune:
        net.autotune(B, HW, HW, cond_channels=(0, 0, n_cond))
    sampler = S.Euler_Maruyama_sampler if sampler_key == "em" else S.pc_sampler
    evals = 1 if sampler_key == "em" else 2
    kw = dict(batch_size=B, device=dev, img_size=HW, cond_img=cond, use_graph=True, seed=1234)
    tw = time.perf_counter()
    sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=max(2, warmup), **kw)
    torch.cuda.synchronize()
    warm_s = time.perf_counter() - tw                    # untimed steps up to CLOCK_WARM_S of sampling, as for the headline
    if warm_s < CLOCK_WARM_S:
        sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=max(2, int((CLOCK_WARM_S - warm_s) / max(warm_s / max(2, warmup), 1e-4)) + 1), **kw)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=steps, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    roof = conv_roofline(net, cond, B, HW, sampler_key, dt / (steps * evals) * 1e3, dev) if n_cond and tune else None
    return {"value": B * steps / dt, "unit": "denoising steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "network_evals_per_s": B * steps * evals / dt,
            "workload": f"{HW}x{HW}, C_in={1 + n_cond}, batch {B}, {'Euler-Maruyama' if sampler_key == 'em' else 'predictor-corrector'} "
                        f"({evals} network eval/step), hipGraph on" + ("" if tune else ", NO autotune (static kernel choice)"), "roofline": roof}


def domain_secondary(dev, steps, warmup):
    """BASELINE configs[4] on one GPU: a 589x789 field as 12 tiles of 256x256 (halo 32), predictor-corrector, stitched"""
    import sbgm_danra_amd as S
    from sbgm_danra_amd.tiling import FullDomainTiler
    net = build_model(dev)
    tiler = FullDomainTiler((589, 789), 256, 32, device=dev)
    net.autotune(len(tiler), 256, 256, cond_channels=(0, 0, 1))
    g = torch.Generator().manual_seed(42)
    cond = torch.randn(1, 589, 789, generator=g).to(dev)
    run = lambda n: tiler.sample(net, S.pc_sampler, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=n, cond_img=cond, seed=7)   # noqa: E731
    run(max(2, warmup))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert out.shape == (1, 589, 789) and torch.isfinite(out).all()
    flops_step = 2 * FLOP_PER_SAMPLE_128 * len(tiler) * 4.0                     # 2 evaluations of 12 tiles of 256x256 per SDE step
    return {"value": len(tiler) * steps / dt, "unit": "tile denoising steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "warmup": warmup, "tiles": len(tiler), "seconds_per_500_step_field": dt / steps * 500,
            "workload": "589x789 domain, 12 tiles of 256x256 (halo 32), predictor-corrector, tile extraction + stitch included",
            "whole_step": {"gflop": flops_step * 1e-9, "fp32_frac": flops_step / (dt / steps) / (PEAK_FP32_TFLOPS * 1e12)}}


def conv_roofline(net, cond, B, HW, sampler_key, ms_eval_in, dev, profile_csv=None):
    """Roofline of the dominant convolution kernel family of one network evaluation at (B, HW): every convolution launch bracketed
    by HIP events on the launch stream (sbgm_model_profile_forward), grouped by kernel function as `rocprofv3 --stats` does; HBM
    traffic from the committed PMC passes of the same workload when they were measured on these kernel sources."""
    from sbgm_danra_amd import _native as N
    lib = N.lib()
    x = torch.randn(B, 1, HW, HW, device=dev) * 10
    t = torch.full((B,), 0.5, device=dev)
    o = torch.empty_like(x)
    eng = net._engine(None, None, cond)
    prof = N.Profile()
    best = None
    import csv
    import tempfile
    csv_path = profile_csv or os.path.join(tempfile.gettempdir(), f"sbgm_conv_profile_{os.getpid()}.csv")
    for _ in range(5):
        N.check(lib.sbgm_model_profile_forward(eng.h, x.data_ptr(), t.data_ptr(), None, cond.data_ptr(), None, None,
                                               o.data_ptr(), B, HW, HW, C.byref(prof), csv_path.encode(), N.stream()))
        if best is None or prof.ms_conv < best[0]:
            with open(csv_path) as f:
                rows = list(csv.DictReader(f))
            best = (prof.ms_conv, prof.flops_conv, prof.n_conv, prof.ms_total_with_events, prof.ms_conv_max,
                    prof.flops_conv_max, rows)
    ms_conv, fl_conv, n_conv, ms_tot, ms_max, fl_max, rows = best
    if not profile_csv:
        os.unlink(csv_path)
    # group the launches by kernel instantiation, the way `rocprofv3 --stats` does (profiles/r01_bench_kernel_stats.csv)
    per = {}
    for r in rows:
        k = per.setdefault(r["kernel"].replace(";", ","), {"launches": 0, "ms": 0.0, "gflop": 0.0, "bytes": 0.0})
        k["launches"] += 1
        k["ms"] += float(r["ms"])
        k["gflop"] += float(r["gflop"])
        # algorithmic bytes of a launch: input once + output once + weights once (fp32)
        cin, cout, kk = int(r["Cin_pad"]), int(r["Cout"]), int(r["kh"]) * int(r["kw"])
        k["bytes"] += 4.0 * (int(r["B"]) * int(r["H"]) * int(r["W"]) * cin + int(r["M"]) * cout + kk * cin * cout)
    kernels = [{"kernel": n, "launches": v["launches"], "avg_us": 1e3 * v["ms"] / v["launches"],
                "gflop_per_launch": v["gflop"] / v["launches"], "tflops": v["gflop"] / v["ms"],
                "alg_bytes_per_launch": v["bytes"] / v["launches"]}
               for n, v in sorted(per.items(), key=lambda kv: -kv[1]["ms"])]
    for k in kernels:        # Winograd instantiations execute 2/3 (F(2,3) along rows) or 4/9 (F(2x2,3x3)) of the algorithmic (direct-convolution) FLOPs as MFMAs
        wino = "wino_kernel" in k["kernel"] or ("lds_kernel" in k["kernel"] and k["kernel"].split(",")[2].strip() == "true")
        k["executed_flop_ratio"] = 4.0 / 9.0 if "w2d" in k["kernel"] else (2.0 / 3.0 if wino else 1.0)      # F(2x2,3x3): 4 of 9
    # The dominant kernel = the kernel FUNCTION (template) with the largest share of an evaluation; the autotuner spreads its
    # launches over several instantiations (tile / Winograd / double-buffer arguments), listed under "instantiations" with
    # the names `rocprofv3 --stats` prints, so the two can be compared row by row.
    fam = {}
    for k in kernels:
        f = fam.setdefault(k["kernel"].split("<")[0], {"launches": 0, "us": 0.0, "gflop": 0.0, "bytes": 0.0, "exec": 0.0, "inst": []})
        f["launches"] += k["launches"]
        f["us"] += k["avg_us"] * k["launches"]
        f["gflop"] += k["gflop_per_launch"] * k["launches"]
        f["bytes"] += k["alg_bytes_per_launch"] * k["launches"]
        f["exec"] += k["gflop_per_launch"] * k["launches"] * k["executed_flop_ratio"]
        f["inst"].append(k)
    dname, df = max(fam.items(), key=lambda kv: kv[1]["us"])
    dom = {"kernel": dname + "<...>", "launches": df["launches"], "avg_us": df["us"] / df["launches"],
           "gflop_per_launch": df["gflop"] / df["launches"], "tflops": df["gflop"] / df["us"] * 1e3,
           "alg_bytes_per_launch": df["bytes"] / df["launches"], "executed_flop_ratio": df["exec"] / df["gflop"],
           "instantiations": df["inst"]}
    # HBM-side traffic of that kernel: PMC counters cannot be read from inside this process; they come from the separately
    # collected rocprofv3 passes of THIS workload (tools/collect_profiles.sh -> profiles/<round>_pmc_traffic_<workload>.json).
    # A file measured on other kernel sources (source_hash) is refused: traffic = null with the reason.
    traffic, traffic_src = None, None
    wl = f"b{B}_{HW}_{sampler_key}"
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        tp = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_{wl}.json")
        if not os.path.exists(tp):
            continue
        with open(tp) as f:
            tj = json.load(f)
        if tj.get("source_hash") != kernel_source_hash():
            traffic_src = f"refused profiles/{tag}_pmc_traffic_{wl}.json: measured on kernel sources {tj.get('source_hash')}, this tree is {kernel_source_hash()}"
            break
        ents = [(v["bytes_per_launch"], v["launches"]) for n, v in tj["kernels"].items() if n.split("<")[0] == dname]
        if ents:
            traffic = sum(b * n for b, n in ents) / sum(n for _, n in ents)
            traffic_src = f"profiles/{tag}_pmc_traffic_{wl}.json (2*FETCH_SIZE + WRITE_SIZE per launch, launch-weighted over the instantiations; same kernel sources)"
        break
    if traffic_src is None:
        traffic_src = f"no profiles/*_pmc_traffic_{wl}.json for this workload"
    ach = fl_conv / (ms_conv * 1e-3) * 1e-12
    ms_eval = ms_eval_in
    flops_eval = FLOP_PER_SAMPLE_128 * B * (HW / 128.0) ** 2
    # top level = the single kernel instantiation with the largest share of an evaluation; `conv_family` = all convolution
    # launches of one evaluation (3 kernel templates, fp32 v_mfma_f32_16x16x4_f32); `kernels` = every instantiation.
    roof = {"bound": "mfma", "kernel": dom["kernel"], "launches_per_eval": dom["launches"], "avg_launch_us": dom["avg_us"],
            "gflop_per_launch": dom["gflop_per_launch"],
            "achieved": dom["tflops"], "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK_FP32_TFLOPS,
            # `achieved` counts the algorithmic (direct-convolution) FLOPs of SURVEY 8d; a Winograd F(2,3) kernel issues 2/3 of
            # them, so the matrix pipe itself is busy for about frac * executed_flop_ratio of the time
            "executed_flop_ratio": dom["executed_flop_ratio"],
            "mfma_pipe_frac": dom["tflops"] * dom["executed_flop_ratio"] / PEAK_FP32_TFLOPS,
            "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
            "alg_bytes_per_launch": dom["alg_bytes_per_launch"], "instantiations": dom["instantiations"],
            "conv_family": {"launches": n_conv, "ms_per_eval": ms_conv, "gflop_per_eval": fl_conv * 1e-9, "achieved": ach,
                            "frac": ach / PEAK_FP32_TFLOPS},
            "slowest_launch": {"ms": ms_max, "tflops": fl_max / (ms_max * 1e-3) * 1e-12},
            "kernels": kernels,
            "whole_eval": {"ms": ms_eval, "gflop": flops_eval * 1e-9,
                           "fp32_frac": flops_eval / (ms_eval * 1e-3) / (PEAK_FP32_TFLOPS * 1e12),
                           "hbm_frac_layer_fused_bytes": BYTES_PER_EVAL(B, HW) / (ms_eval * 1e-3) / (PEAK_HBM_GBS * 1e9)}}
    return roof


def train_secondary(dev, steps=10, warmup=3):
    """C3 per-GPU training step (128x128, 4 conditions, batch 8): loss_fn forward + native backward replayed as one hipGraph + the
    native Adam step, timed like the headline; plus the roofline of the weight-gradient kernel family — every convolution /
    linear weight gradient of the step re-launched in isolation (10 launches in one graph, HIP events) on the step's own shapes."""
    import sbgm_danra_amd as S
    from sbgm_danra_amd import _native as N
    from sbgm_danra_amd import train_graph as T
    net = build_model(dev, n_cond=4)
    net.train()
    opt = S.optim.Adam(net.parameters(), lr=5e-4, weight_decay=1e-6)
    B, HW = 8, 128
    g = torch.Generator().manual_seed(42)
    x, cond = torch.randn(B, 1, HW, HW, generator=g).to(dev), torch.randn(B, 4, HW, HW, generator=g).to(dev)

    def fwd_bwd():
        loss = S.loss_fn(net, x, S.marginal_prob_std_fn, cond_img=cond)
        loss.backward()
        return loss
    T._WGRAD_LOG[0] = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            opt.zero_grad(set_to_none=True)
            fwd_bwd()
            opt.step()
            if i == 0:
                geoms, T._WGRAD_LOG[0] = T._WGRAD_LOG[0], None
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss = fwd_bwd()
    tw = time.perf_counter()
    for _ in range(warmup):
        graph.replay()
        opt.step()
    torch.cuda.synchronize()
    warm_s = time.perf_counter() - tw
    for _ in range(int((CLOCK_WARM_S - warm_s) / max(warm_s / max(1, warmup), 1e-4)) + 1 if warm_s < CLOCK_WARM_S else 0):
        graph.replay()                                   # untimed: steady device clocks, as for the headline
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
        opt.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # ---- weight-gradient family: each layer launched alone, and the whole family the way the step runs it (queued during the backward
    # sweep, three batched launches + one layout launch at its end: sbgm_wgrad_defer bits 0 and 1) --------------------------------
    lib, tot_us, tot_fl, rows, ops = N.lib(), 0.0, 0.0, [], []
    for (Bq, H, W, cs, cin, cout, k, stride, pad) in geoms:
        oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        dy, xx = torch.randn(Bq, oh, ow, cout, device=dev), torch.randn(Bq, H, W, cs, device=dev)
        dw, ws = torch.empty(cout, cin, k, k, device=dev), torch.zeros(k * k * cout * cs, device=dev)

        def launch(dy=dy, xx=xx, dw=dw, ws=ws, g=(Bq, H, W, cs, cin, cout, k, stride, pad)):
            N.check(lib.sbgm_conv2d_wgrad(dy.data_ptr(), xx.data_ptr(), dw.data_ptr(), ws.data_ptr(), g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[6],
                                          g[7], g[8], N.stream()))
        ops.append((launch, ws))
        launch()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg):
            for _ in range(10):
                launch()
        gg.replay()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        torch.cuda.synchronize()
        e0.record()
        gg.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        fl = 2.0 * Bq * oh * ow * cin * cout * k * k
        tot_us += us
        tot_fl += fl
        rows.append({"B": Bq, "H": H, "W": W, "Cin": cin, "Cout": cout, "k": k, "stride": stride, "us": us, "tflops": fl / us * 1e-6})
    iso_us = tot_us

    def family():                                             # zero the slabs (the step's pool arrives zeroed), queue every layer, flush
        for _, ws in ops:
            ws.zero_()
        prev = lib.sbgm_wgrad_defer(3)
        try:
            for launch, _ in ops:
                launch()
        finally:
            lib.sbgm_wgrad_defer(prev)
        N.check(lib.sbgm_wgrad_flush(N.stream()))
    family()
    gz, gf = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gz):
        for _, ws in ops:
            ws.zero_()
    with torch.cuda.graph(gf):
        family()
    best = None
    for _ in range(5):
        ev = [torch.cuda.Event(True) for _ in range(3)]
        torch.cuda.synchronize()
        ev[0].record(); gz.replay(); ev[1].record(); gf.replay(); ev[2].record()
        torch.cuda.synchronize()
        us = (ev[1].elapsed_time(ev[2]) - ev[0].elapsed_time(ev[1])) * 1e3          # the zero-fills are the pool's, not the family's
        best = us if best is None else min(best, us)
    tot_us = best
    ach = tot_fl / tot_us * 1e-6
    rows.sort(key=lambda r: -r["us"])
    return {"metric": "training samples/sec at 128x128 (loss_fn forward + backward as one hipGraph + native Adam), 1 GPU",
            "value": B * steps / dt, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "workload": "128x128 4-cond->1-target (C_in=5), batch 8 per GPU (BASELINE configs[2] per-GPU shape)",
            "final_loss": float(loss.detach()), "fp32_frac_of_step": 3 * 5.247e9 * B / (dt / steps) / (PEAK_FP32_TFLOPS * 1e12),
            "wgrad_roofline": {"bound": "mfma", "kernel": "conv weight-gradient family (conv3x3_wgrad_batched<16|8> / conv_tap_wgrad_batched / conv_wgrad + layout pass)",
                               "launches_per_step": len(geoms), "sum_us": tot_us, "gflop": tot_fl * 1e-9, "achieved": ach,
                               "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                               "isolated_sum_us": iso_us,
                               "note": "sum_us / achieved: the step's weight gradients queued and flushed as the step runs them (batched launches per kernel "
                                       "family + one OIHW layout launch; HIP events around a graph replay, zero-fills subtracted); isolated_sum_us / slowest: "
                                       "each layer launched alone (10 per graph), including its zero-fill and layout pass",
                               "slowest": rows[:5]}}


def train_bench(a):
    """BASELINE configs[2]: 128x128, 4 ERA5 conditions -> 1 target, batch 8 per GPU (global 64 on 8), one optimizer step =
    loss_fn forward + native backward + RCCL gradient all-reduce (world > 1) + Adam.  Secondary line: not the headline."""
    import torch.distributed as dist
    import sbgm_danra_amd as S
    from sbgm_danra_amd import parallel
    rank, world, local = parallel.init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    net = build_model(dev, n_cond=4)
    net.train()
    parallel.broadcast_parameters(net)
    torch.manual_seed(1000 + rank)                 # identical weights, but every replica draws its own loss noise (t, z)
    # the reference's default foreach Adam issues ~330 tiny per-parameter kernels per step
    opt = S.optim.Adam(net.parameters(), lr=5e-4, weight_decay=1e-6)      # torch.optim.Adam with a one-launch native step()
    # the model's flat gradient arena IS the bucket: backward writes into it, one RCCL all-reduce, Adam reads it (no copies)
    bucket = parallel.GradientBucket(net) if world > 1 else None
    if a.sync_bn:
        from sbgm_danra_amd.train_graph import set_sync_batchnorm
        set_sync_batchnorm(True)
    B, HW = (a.batch if a.batch != 32 else 8), a.size
    g = torch.Generator().manual_seed(42 + rank)
    x, cond = torch.randn(B, 1, HW, HW, generator=g).to(dev), torch.randn(B, 4, HW, HW, generator=g).to(dev)

    graphed = not a.no_graph and not (a.sync_bn and world > 1)     # SyncBatchNorm's collectives run eagerly inside the step
    def fwd_bwd():
        loss = S.loss_fn(net, x, S.marginal_prob_std_fn, cond_img=cond)
        loss.backward()
        return loss

    if bucket is not None:
        opt.grad_scale = 1.0 / world               # the averaging rides on the Adam launch: the bucket keeps the SUM
        if not graphed:                            # eager steps: the decoder slice is all-reduced from inside backward (overlap)
            from sbgm_danra_amd.train_graph import set_overlap_bucket
            set_overlap_bucket(bucket)

    def finish():
        if bucket is not None:
            bucket.all_reduce_(average=False)
        opt.step()

    if graphed:
        # The step's launch sequence is static (shapes, tiles and the RNG draw sites are fixed), so forward + backward are
        # captured once into a hipGraph (torch.cuda.graphs) and replayed: one host call per step instead of ~900 launches
        # through Python.  The gradient all-reduce and the optimizer run eagerly after the replay.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                fwd_bwd()
                finish()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            static_loss = fwd_bwd()

        def step():
            graph.replay()
            finish()
            return static_loss
    else:
        def step():
            opt.zero_grad()
            loss = fwd_bwd()
            finish()
            return loss

    for _ in range(max(1, a.warmup)):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    if rank == 0:
        print(json.dumps({"metric": "training samples/sec at 128x128 (forward + backward + Adam)", "value": B * world * a.steps / dt,
                          "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"{HW}x{HW} 4-cond->1-target (C_in=5), batch {B}/GPU, loss_fn + backward + Adam"
                                                 + (", RCCL gradient all-reduce (76 MB flat arena, no copies)" if world > 1 else "")
                                                 + (", SyncBatchNorm" if (a.sync_bn and world > 1) else (", per-replica BatchNorm statistics" if world > 1 else ""))
                                                 + (", forward+backward replayed as one hipGraph" if graphed else ""),
                                     "global_batch": B * world, "final_loss": float(loss.detach())}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def domain_bench(a):
    """BASELINE configs[4]: one 589x789 field as 12 overlapping 256x256 tiles (halo 32), predictor-corrector sampling, tiles
    sharded over the ranks (one all-reduce of the finished tiles before the blend).  Secondary line: not the headline."""
    import torch.distributed as dist
    import sbgm_danra_amd as S
    from sbgm_danra_amd import parallel
    from sbgm_danra_amd.tiling import FullDomainTiler
    rank, world, local = parallel.init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    net = build_model(dev)
    tiler = FullDomainTiler((589, 789), 256, 32, device=dev)
    mine = len(range(len(tiler))[rank::world])
    if not a.no_autotune:
        net.autotune(mine, 256, 256, cond_channels=(0, 0, 1), cache=a.tune_cache)
    g = torch.Generator().manual_seed(42)
    cond = torch.randn(1, 589, 789, generator=g).to(dev)
    run = lambda n: tiler.sample(net, S.pc_sampler, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=n, cond_img=cond, seed=7)   # noqa: E731
    run(max(2, a.warmup))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(a.steps)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert out.shape == (1, 589, 789) and torch.isfinite(out).all()
    if world > 1:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    if rank == 0:
        print(json.dumps({"metric": "denoising steps/sec on the full 589x789 domain (tiles x SDE-steps/s)", "value": len(tiler) * a.steps / dt,
                          "unit": "tile denoising steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"589x789 domain, {len(tiler)} tiles of 256x256 (halo 32), predictor-corrector, "
                                                 f"{a.steps} steps, tiles sharded over {world} rank(s), domain-keyed noise, normalised ramp blend",
                                     "seconds_per_field": dt, "tiles": len(tiler)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--sampler", choices=["em", "pc"], default="em")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-clock-warm", action="store_true", help="do not run extra untimed steps when the warm-up is shorter than 0.15 s of sampling")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 training-step measurement appended to the default line")
    ap.add_argument("--tune-cache", default=None, help="tile-table file: loaded when present, else written after autotuning "
                                                       "(lets the rocprofv3 passes replay exactly the benchmarked kernels)")
    ap.add_argument("--profile-csv", default=None, help="write the per-convolution event timings here")
    ap.add_argument("--sync-bn", action="store_true", help="train mode: SyncBatchNorm (statistics summed over the ranks), eager launches")
    ap.add_argument("--mode", choices=["sample", "train", "domain"], default="sample",
                    help="sample = BASELINE configs[1] (the headline metric); train = configs[2] optimizer steps; domain = "
                         "configs[4] full-domain tiled sampling (secondary lines)")
    a = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:      # plain `python bench.py --gpus N`: be our own launcher
        sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))
    if os.environ.get("SBGM_BENCH_DRYRUN"):                # launcher self-test (tests/test_cpu_host.py): no GPU work
        if int(os.environ.get("RANK", 0)) == 0:
            print(json.dumps({"dryrun": True, "n_gpus": int(os.environ.get("WORLD_SIZE", 1)), "mode": a.mode,
                              "master": os.environ.get("MASTER_ADDR", "") + ":" + os.environ.get("MASTER_PORT", "")}), flush=True)
        return 0
    if a.mode == "train":
        return train_bench(a)
    if a.mode == "domain":
        return domain_bench(a)

    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}, or without a "
                 f"torchrun environment (bench.py then starts the ranks itself)")
    import torch.distributed as dist
    from sbgm_danra_amd import parallel
    rank, world, local = parallel.init_distributed()       # RCCL ("nccl") over xGMI: only the barrier and the max-over-ranks of the time
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import sbgm_danra_amd as S
    from sbgm_danra_amd import _native as N
    net = build_model(dev)
    B, HW = a.batch, a.size
    g = torch.Generator().manual_seed(42 + rank)
    cond = torch.randn(B, 1, HW, HW, generator=g).to(dev)
    if not a.no_autotune:
        net.autotune(B, HW, HW, cond_channels=(0, 0, 1), cache=a.tune_cache)
    sampler = S.Euler_Maruyama_sampler if a.sampler == "em" else S.pc_sampler
    evals_per_step = 1 if a.sampler == "em" else 2
    kw = dict(batch_size=B, device=dev, img_size=HW, cond_img=cond, use_graph=not a.no_graph, seed=1234 + rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up (untimed): graph capture / instantiate paths, caches, clocks; the finite-check below loads its torch kernels here, not
    # between two timed loops
    assert torch.isfinite(cond).all()
    tw = time.perf_counter()
    sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=max(2, a.warmup), **kw)
    barrier()
    # Device clocks: after the tuning phase (short launches, host-side gaps) the part needs ~0.1-0.15 s of continuous load to reach its
    # steady clock (tools/micro/call_seq.py, call_gap.py: a 20-step call right after 10 / 100 / 500 ms of idle is 0.6 / 1.2 / 1.6 ms slower).
    # W = 5 warm-up steps are 8 ms.  When the requested warm-up is shorter than CLOCK_WARM_S of sampling, more UNTIMED steps of the same
    # workload follow it (reported as config.clock_warm_steps); the timed region is unchanged: exactly K steps between two barriers.
    clock_warm_steps = 0
    warm_s = time.perf_counter() - tw
    per_step = warm_s / max(2, a.warmup)
    if warm_s < CLOCK_WARM_S and not a.no_clock_warm:
        clock_warm_steps = max(2, int((CLOCK_WARM_S - warm_s) / max(per_step, 1e-4)) + 1)
        sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=clock_warm_steps, **kw)
    barrier()
    t0 = time.perf_counter()
    out = sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=a.steps, **kw)
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    if world > 1:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    # the same K-step loop twice more (reported beside the headline, which stays the FIRST timed loop): at K = 20 the timed region is
    # ~35 ms, so one number alone says little about its own spread
    repeats = [dt]
    for _ in range(2):
        barrier()
        t1 = time.perf_counter()
        sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=a.steps, **kw)
        barrier()
        repeats.append(time.perf_counter() - t1)

    roof, cpu = None, None
    if rank == 0:
        roof = conv_roofline(net, cond, B, HW, a.sampler, dt / (a.steps * evals_per_step) * 1e3, dev, a.profile_csv)
    secondary = None
    if rank == 0 and world == 1 and not a.no_secondary and B == 32 and HW == 128:
        # the other BASELINE configurations beside the headline, driver-timed too (each in a try: never lose the headline line):
        # configs[2] per-GPU training step, configs[3] 256x256 predictor-corrector, configs[4] full-domain tiles, configs[0] 64x64
        del net, out, cond
        torch.cuda.empty_cache()
        secondary = {}
        for name, fn in (("train_c3", lambda: train_secondary(dev)),
                         ("c4_pc", lambda: sampling_secondary(dev, 16, 256, "pc", 10, 3)),
                         ("c5_domain", lambda: domain_secondary(dev, 6, 2)),
                         ("c1_em", lambda: sampling_secondary(dev, 1, 64, "em", 50, 5, n_cond=0)),
                         ("c2_untuned", lambda: sampling_secondary(dev, 32, 128, "em", a.steps, a.warmup, tune=False))):
            try:
                secondary[name] = fn()
            except Exception as e:
                secondary[name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
    if rank == 0 and not a.no_cpu_baseline and world == 1:         # the CPU reference leg is reported at N=1 only; it runs LAST: its
        cpu = cpu_baseline(B, HW)                                  # OpenMP workers keep spinning on the host cores the GPU legs launch from
        if secondary is not None:                                   # CPU evals/s for the other configurations as well
            for name, args in (("c1_em", dict(batch=1, hw=64, budget_s=4.0, sampler="em", n_cond=0, max_steps=50)),
                               ("c4_pc", dict(batch=16, hw=256, budget_s=10.0, sampler="pc", max_steps=5))):
                try:
                    if isinstance(secondary.get(name), dict):
                        secondary[name]["cpu_baseline"] = cpu_baseline(**args)
                except Exception as e:
                    secondary[name]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        value = B * a.steps * world / dt
        line = {"metric": f"denoising steps/sec (batch x SDE-steps/s) at {HW}x{HW}", "value": value, "unit": "denoising steps/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{HW}x{HW} 1-cond->1-target (C_in=2), batch {B}/GPU, VE-SDE "
                                       f"{'Euler-Maruyama' if a.sampler == 'em' else 'predictor-corrector'} sampling, "
                                       f"{evals_per_step} network eval/step, hipGraph={'off' if a.no_graph else 'on'}",
                           "global_batch": B * world, "sampler": a.sampler, "network_evals_per_s": value * evals_per_step,
                           "clock_warm_steps": clock_warm_steps,
                           "repeat_ms_per_step": {"runs": [r / a.steps * 1e3 for r in repeats], "min": min(repeats) / a.steps * 1e3,
                                                  "median": sorted(repeats)[1] / a.steps * 1e3,
                                                  "note": "3 consecutive timed loops of the same K steps; ms_per_step / value are the first"},
                           "parallelism": f"dp{world} (independent batches, no collective)"},
                "roofline": roof, "cpu_baseline": cpu, "secondary": secondary}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
