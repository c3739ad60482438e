"""CPU suite, part 2: host logic of the drop-in boundary — config schema loader, naming, batch extraction, model
factory / state_dict compatibility, the C-ABI library's symbol table, loud failure without a device, and the
one-process-per-GPU layer exercised with world_size-2 gloo groups."""
import ctypes
import json
import os
import re
import socket

import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

import sbgm_danra_amd as S
from sbgm_danra_amd import _native as N
from sbgm_danra_amd import parallel
from sbgm_danra_amd.config_loader import load_config, to_config
from sbgm_danra_amd.synthetic_data import synthetic_loader
from sbgm_danra_amd.training_utils import get_model, get_optimizer, infer_in_channels
from sbgm_danra_amd.utils import extract_samples, get_model_string, report_precip_extremes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "sbgm_danra_amd", "config", "default_config.yaml")


@pytest.fixture()
def cfg(tmp_path, monkeypatch):
    for k in ("DATA_DIR", "CKPT_DIR", "SAMPLE_DIR", "STATS_LOAD_DIR"):
        monkeypatch.setenv(k, str(tmp_path / k.lower()))
    monkeypatch.setenv("SLURM_CPUS_PER_TASK", "3")
    return load_config(CFG)


def test_config_env_interpolation_and_both_access_styles(cfg, tmp_path):
    assert cfg.paths.checkpoint_dir == cfg["paths"]["checkpoint_dir"] == str(tmp_path / "ckpt_dir")
    assert cfg.data_handling.num_workers == 3 and cfg["sampler"]["block_layers"] == [2, 2, 2, 2]
    assert cfg.get("model", {}).get("decoder_norm") == "group"
    with pytest.raises(FileNotFoundError):
        load_config(str(tmp_path / "nope.yaml"))
    c = to_config({"a": "${env:SBGM_UNSET_VAR_X}", "b": "x/${env:SBGM_UNSET_VAR_X}/y"})
    assert c.a is None and c.b == "x//y"


def test_model_string_and_channel_inference(cfg):
    assert get_model_string(cfg) == "sbgm_mi355x__HR_prcp_DANRA__SIZE_128x128__LR_prcp_ERA5__LOSS_sdfweighted__HEADS_4__TIMESTEPS_1000"
    assert infer_in_channels(cfg) == 1
    cfg.stationary_conditions.geographic_conditions.sample_w_geo = True
    cfg.lowres.condition_variables = ["temp", "prcp"]
    assert infer_in_channels(cfg) == 2 + 4


def test_extract_samples_layout(cfg):
    cfg.stationary_conditions.geographic_conditions.sample_w_geo = True
    cfg.stationary_conditions.geographic_conditions.sample_w_sdf = True
    cfg.stationary_conditions.seasonal_conditions.sample_w_cond_season = True
    cfg.lowres.condition_variables = ["temp", "prcp"]
    cfg.highres.data_size = [32, 32]
    batch = next(iter(synthetic_loader(cfg, 3)))
    x, cls, lr, lsm_hr, lsm, sdf, topo, hp, lp = extract_samples(batch, "cpu")
    assert x.shape == (3, 1, 32, 32) and lr.shape == (3, 2, 32, 32) and lsm.shape == topo.shape == (3, 2, 32, 32)
    assert sdf.shape == (3, 1, 32, 32) and cls.shape == (3,) and cls.dtype == torch.int64 and hp is None and lp is None
    assert torch.equal(lsm[:, 1], torch.ones(3, 32, 32))                      # value||mask convention
    assert torch.equal(lr, torch.cat([batch["prcp_lr"], batch["temp_lr"]], 1))  # sorted *_lr keys
    with pytest.raises(ValueError):
        extract_samples({"foo": torch.zeros(1)}, "cpu")


def test_precip_sentinel_is_device_only():
    """the sentinel statistics run on the device (csrc/postproc.hip); a CPU tensor is refused, not silently handled"""
    from sbgm_danra_amd._native import NativeError
    with pytest.raises(NativeError):
        report_precip_extremes(torch.rand(3, 1, 64, 64), "t", 500.0, logger=lambda *_: None)


def test_get_model_matches_reference_state_dict(cfg, golden_dir):
    """keys / shapes identical to what the reference's get_model builds (manifest captured from the reference)"""
    cfg.training.device = "cpu"
    man = json.load(open(os.path.join(golden_dir, "state_manifest.json")))
    model, ckpt_dir, ckpt_name = get_model(cfg)
    sd = model.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == {k: v[0] for k, v in man["ncond1_cls0"].items()}
    assert ckpt_name.endswith(".pth.tar") and len(sd) == 231
    assert sum(p.numel() for p in model.parameters()) == 19_062_082           # SURVEY.md §4 [probe]
    cfg.stationary_conditions.geographic_conditions.sample_w_geo = True
    cfg.stationary_conditions.seasonal_conditions.sample_w_cond_season = True
    cfg.lowres.condition_variables = ["temp", "prcp"]
    model2, _, _ = get_model(cfg)
    assert {k: list(v.shape) for k, v in model2.state_dict().items()} == {k: v[0] for k, v in man["ncond6_cls4"].items()}
    assert float(model2.encoder.label_emb.weight[0].abs().sum()) == 0.0       # null class row
    assert isinstance(get_optimizer(cfg, model2), torch.optim.Adam)
    # a checkpoint written from the oracle (= reference layout) loads into the native module
    from oracle import torch_ref as O
    model2.load_state_dict(O.synth_state_dict(O.build_scorenet(6, num_classes=4)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sbgm_hip.h")).read()
    declared = set(re.findall(r"\b(sbgm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sbgm_model", "sbgm_model_config", "sbgm_sampler_args", "sbgm_conv_args", "sbgm_profile"}
    assert len(declared) >= 35
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    lib = N.lib()                                # dlopen + resolve all of them (no compute, no GPU needed)
    assert lib.sbgm_abi_version() == N.ABI_VERSION
    raw = ctypes.CDLL(N.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.sbgm_conv_packed_numel(64, 3, 3, 64) == 9 * 4 * 64 * 16


def test_no_cpu_fallback():
    enc = S.Encoder(1, 256)
    dec = S.Decoder(512, 1, 256, norm="group", activation=nn.SiLU)
    net = S.ScoreNet(S.marginal_prob_std_fn, enc, dec, device=torch.device("cpu")).eval()
    with pytest.raises(N.NativeError):
        with torch.no_grad():
            net(torch.randn(1, 1, 32, 32), torch.rand(1), cond_img=torch.randn(1, 1, 32, 32))
    with pytest.raises(N.NativeError):                       # stand-alone sub-module calls are native too
        net.encoder(torch.randn(1, 1, 32, 32), torch.rand(1), cond_img=torch.randn(1, 1, 32, 32))
    with pytest.raises(NotImplementedError):                 # leaf containers have no forward of their own
        net.encoder.layer1[0](torch.randn(1, 64, 8, 8))
    with pytest.raises(N.NativeError):
        S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=1, num_steps=2, device="cpu", img_size=32,
                     cond_img=torch.randn(1, 1, 32, 32))
    assert "oracle" not in "".join(open(os.path.join(ROOT, "sbgm_danra_amd", f)).read()
                                   for f in os.listdir(os.path.join(ROOT, "sbgm_danra_amd")) if f.endswith(".py")).replace(
        "oracle (", "").replace("the oracle", "")


def test_schedule_functions_match_oracle():
    from oracle import torch_ref as O
    t = torch.tensor([1e-5, 1e-3, 0.3, 1.0])
    assert torch.equal(S.marginal_prob_std_fn(t), O.marginal_prob_std_fn(t))
    assert torch.equal(S.diffusion_coeff_fn(t), O.diffusion_coeff_fn(t))


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 13, 64):
        for w in (1, 2, 3, 8):
            parts = [list(parallel.shard_range(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


# ---- world_size-2 gloo tests ------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    parallel.init_distributed("gloo")
    try:
        fn(rank, world, out_dir)
    finally:
        torch.distributed.destroy_process_group()


def _spawn(fn, tmp_path, world=2):
    mp.spawn(_worker, args=(world, _free_port(), fn, str(tmp_path)), nprocs=world, join=True)


def _grad_bucket_case(rank, world, out_dir):
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(2, 4, 3, padding=1), nn.GroupNorm(2, 4), nn.Conv2d(4, 1, 1))
    parallel.broadcast_parameters(net)
    full = torch.randn(8, 2, 6, 6, generator=torch.Generator().manual_seed(1))
    x = full[list(parallel.shard_range(8, rank, world))]
    (net(x) ** 2).mean().backward()
    parallel.GradientBucket(net.parameters()).all_reduce_()
    torch.save([p.grad.clone() for p in net.parameters()], os.path.join(out_dir, f"g{rank}.pt"))
    if rank == 0:                                   # single-process reference over the whole batch
        ref = nn.Sequential(nn.Conv2d(2, 4, 3, padding=1), nn.GroupNorm(2, 4), nn.Conv2d(4, 1, 1))
        ref.load_state_dict(net.state_dict())
        (ref(full) ** 2).mean().backward()
        torch.save([p.grad.clone() for p in ref.parameters()], os.path.join(out_dir, "gref.pt"))


def test_gradient_all_reduce_equals_full_batch_gradient(tmp_path):
    _spawn(_grad_bucket_case, tmp_path)
    g0, g1, gref = (torch.load(tmp_path / f, weights_only=True) for f in ("g0.pt", "g1.pt", "gref.pt"))
    for a, b, r in zip(g0, g1, gref):
        assert torch.equal(a, b)                                              # replicas agree bit-for-bit
        assert torch.allclose(a, r, rtol=1e-5, atol=1e-7)                     # mean of shard grads == full-batch grad


def _sharded_sampling_case(rank, world, out_dir):
    from oracle import torch_ref as O                 # checker model stands in for the score network on CPU
    m = O.build_scorenet(1).eval()
    m.load_state_dict(O.synth_state_dict(m))

    def kwargs(i):
        g = torch.Generator().manual_seed(100 + i)
        noise = [torch.randn(1, 1, 32, 32, generator=g) for _ in range(5)]
        return dict(score_model=m, marginal_prob_std=O.marginal_prob_std_fn, diffusion_coeff=O.diffusion_coeff_fn, batch_size=1,
                    num_steps=2, img_size=32, cond_img=torch.randn(1, 1, 32, 32, generator=g), noise=iter(noise))
    res = parallel.sample_sharded(O.pc_sampler, 3, kwargs, gather=True)
    if rank == 0:
        assert sorted(res) == [0, 1, 2]
        solo = {i: O.pc_sampler(**kwargs(i)) for i in range(3)}
        assert all(torch.equal(res[i], solo[i]) for i in range(3))           # sharding changes nothing: no collective
        torch.save(True, os.path.join(out_dir, "ok.pt"))
    else:
        assert sorted(res) == list(parallel.shard_range(3, rank, world))


def test_sampling_shards_without_collectives(tmp_path):
    _spawn(_sharded_sampling_case, tmp_path)
    assert (tmp_path / "ok.pt").exists()


# ---- the C ABI's structs: header == ctypes binding == documented binding ------------------------------------------------
def _header_structs():
    """{struct name: [field names in order]} for every `typedef struct ... { } name;` of include/sbgm_hip.h"""
    hdr = open(os.path.join(ROOT, "include", "sbgm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", hdr, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for d in decl.split(","):
                m = re.search(r"(\w+)\s*(\[[^\]]*\])?\s*$", d.strip())
                fields.append(m.group(1))
        out[name] = fields
    return out


def test_header_structs_match_the_bindings(tmp_path):
    """field count / order of every struct in the header == the ctypes classes of _native.py == the snippet in INTEGRATION.md,
    and sizeof() agrees with a gcc-compiled probe of the header (a binding copied from stale documentation once made
    sbgm_model_create read past the caller's struct)"""
    import subprocess
    structs = _header_structs()
    pairs = {"sbgm_model_config": N.ModelConfig, "sbgm_sampler_args": N.SamplerArgs, "sbgm_conv_args": N.ConvArgs,
             "sbgm_pack_desc": N.PackDesc, "sbgm_adam_desc": N.AdamDesc, "sbgm_profile": N.Profile, "sbgm_assemble_args": N.AssembleArgs}
    assert set(structs) == set(pairs), set(structs) ^ set(pairs)
    for cname, cls in pairs.items():
        assert structs[cname] == [f[0] for f in cls._fields_], cname
    assert structs["sbgm_model_config"][0] == "struct_size"
    # sizeof from the C compiler
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include "sbgm_hip.h"\nint main(void){' +
                   "".join(f'printf("{c} %zu\\n", sizeof({c}));' for c in pairs) + "return 0;}\n")
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(sizes[cname]) == ctypes.sizeof(cls), (cname, sizes[cname], ctypes.sizeof(cls))
    assert N.lib().sbgm_model_config_size() == ctypes.sizeof(N.ModelConfig)
    # the documented binding
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("class ModelConfig(C.Structure)"):]
    block = block[:block.index("def check")]
    assert re.findall(r'\("(\w+)",', block) == structs["sbgm_model_config"]
    ctor = re.search(r"cfg = ModelConfig\((.*)\)\n", doc).group(1)
    assert ctor.startswith("C.sizeof(ModelConfig)")


def test_model_create_rejects_a_config_from_another_header():
    """no GPU needed: the size check precedes every allocation"""
    lib = N.lib()
    h = ctypes.c_void_p()
    good = ctypes.sizeof(N.ModelConfig)
    for bad in (0, 2, good - 4, good + 4):          # 0 / 2 = what an old binding (first field n_lsm_channels) would put there
        cfg = N.ModelConfig(bad, 0, 0, 1, 256, (ctypes.c_int * 4)(2, 2, 2, 2), 4, 0, 512, 1, 8, 2, 25.0, 0)
        assert lib.sbgm_model_create(ctypes.byref(cfg), ctypes.byref(h)) != 0 and not h.value
        assert b"struct_size" in lib.sbgm_last_error()


# ---- bench.py as its own multi-GPU launcher ------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["sample", "train", "domain"])
def test_bench_spawns_its_own_ranks(mode):
    """`python bench.py --gpus N` (how the driver calls it) must start N rank processes itself; dry-run: the ranks only report"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SBGM_BENCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1", "--mode", mode],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0's line only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["mode"] == mode and out["master"].startswith("127.0.0.1:")


# ---- gradient arena / bucket ----------------------------------------------------------------------------------------------
def _bucket_views_case(rank, world, out_dir):
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(2, 4, 3, padding=1), nn.GroupNorm(2, 4), nn.Conv2d(4, 1, 1), nn.Linear(6, 6))
    net[3].weight.requires_grad_(False)                          # frozen parameter: not part of the bucket
    parallel.broadcast_parameters(net)
    bucket = parallel.GradientBucket(net)
    x = torch.randn(4, 2, 6, 6, generator=torch.Generator().manual_seed(rank))
    for step in range(2):
        net.zero_grad(set_to_none=True)
        (net(x) ** 2).mean().backward()
        local = [p.grad.clone() for p in net.parameters() if p.requires_grad]
        bucket.all_reduce_()
        flat, views, _offs = bucket._layout()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, views))
        torch.save(local, os.path.join(out_dir, f"local{rank}_{step}.pt"))
        torch.save([p.grad.clone() for p in bucket.params], os.path.join(out_dir, f"avg{rank}_{step}.pt"))


def test_gradient_bucket_on_a_model_repoints_grads_into_the_flat_tensor(tmp_path):
    _spawn(_bucket_views_case, tmp_path)
    for step in range(2):
        l0, l1, a0, a1 = (torch.load(tmp_path / f"{n}_{step}.pt", weights_only=True) for n in ("local0", "local1", "avg0", "avg1"))
        for x0, x1, y0, y1 in zip(l0, l1, a0, a1):
            assert torch.equal(y0, y1) and torch.allclose(y0, (x0 + x1) / 2, rtol=1e-6, atol=1e-8)


def _bucket_overlap_case(rank, world, out_dir):
    """two-part network ("encoder" | "decoder"): the decoder slice is all-reduced asynchronously from INSIDE backward (a hook on
    the boundary activation, as train_graph._BucketBoundary does on the bottleneck feature map), the rest afterwards.  Must equal the
    single all-reduce bit for bit; with average=False the flat tensor holds the SUM (the optimizer launch applies 1 / world)."""
    torch.manual_seed(0)
    enc = nn.Sequential(nn.Conv2d(2, 4, 3, padding=1), nn.GroupNorm(2, 4))
    dec = nn.Sequential(nn.Conv2d(4, 4, 3, padding=1), nn.Conv2d(4, 1, 1))
    net = nn.Sequential(enc, dec)
    extra = nn.Parameter(torch.zeros(3))                         # trainable, never used: no local gradient on any rank
    dec.register_parameter("unused", extra)
    parallel.broadcast_parameters(net)
    x = torch.randn(4, 2, 6, 6, generator=torch.Generator().manual_seed(10 + rank))
    results = {}
    for mode in ("single", "overlap", "sum"):
        bucket = parallel.GradientBucket(net)
        net.zero_grad(set_to_none=True)
        h = enc(x)
        if mode != "single":
            h.register_hook(lambda g, b=bucket: (b.begin_early(list(dec.parameters())), g)[1])
        (dec(h) ** 2).mean().backward()
        assert extra.grad is None
        bucket.all_reduce_(average=mode != "sum")
        assert bucket.early == (0 if mode == "single" else 1)
        assert extra.grad is None                                    # unused on every rank: stays without a gradient (no weight decay)
        flat = bucket._layout()[0]
        results[mode] = flat.clone()
    assert torch.equal(results["single"], results["overlap"])
    assert torch.equal(results["sum"] / world, results["single"])
    # a parameter only rank 0 uses: rank 1 must receive the averaged gradient (replicas would diverge otherwise)
    lone = nn.Parameter(torch.ones(5))
    dec.register_parameter("lone", lone)
    bucket = parallel.GradientBucket(net)
    net.zero_grad(set_to_none=True)
    out = (dec(enc(x)) ** 2).mean() + (lone.sum() * 3.0 if rank == 0 else 0.0)
    out.backward()
    assert (lone.grad is None) == (rank != 0)
    bucket.all_reduce_()
    assert lone.grad is not None and torch.allclose(lone.grad, torch.full((5,), 3.0 / world))
    torch.save(results["single"], os.path.join(out_dir, f"flat{rank}.pt"))


def test_overlapped_bucket_equals_single_all_reduce(tmp_path):
    _spawn(_bucket_overlap_case, tmp_path)
    assert torch.equal(torch.load(tmp_path / "flat0.pt", weights_only=True), torch.load(tmp_path / "flat1.pt", weights_only=True))


def test_sync_batchnorm_refuses_captured_steps(monkeypatch):
    """SyncBatchNorm needs host-driven collectives between kernel halves: the pipeline must refuse use_hip_graph with it before any
    warm-up / capture, not fail inside torch.cuda.graph"""
    from sbgm_danra_amd import train_graph, training
    pipe = training.TrainingPipeline_general.__new__(training.TrainingPipeline_general)
    pipe.cfg = {"training": {"use_hip_graph": True}}
    pipe.device = "cuda"
    pipe.model = nn.Linear(2, 2)
    pipe._bucket = None
    pipe.optimizer = torch.optim.SGD(pipe.model.parameters(), lr=0.1)
    monkeypatch.setattr(train_graph, "_sync_world", lambda: object())
    with pytest.raises(ValueError, match="sync_batchnorm"):
        pipe.train_batches([], epochs=1, verbose=False)


# ---- full-domain tiles over ranks (stub sampler) ---------------------------------------------------------------------------
def _stub_tiles(idx):                                # a tile's result depends on its index only (as with per-tile Langevin norms)
    return torch.stack([torch.full((1, 4, 4), float(i + 1)) * torch.arange(16.).view(1, 4, 4) for i in idx])


def _tile_shard_case(rank, world, out_dir):
    from sbgm_danra_amd.tiling import sample_tiles_sharded
    want = _stub_tiles(list(range(7)))
    calls = []

    def run(idx):
        calls.append(list(idx))
        return _stub_tiles(idx)
    for per in (None, 1, 2):
        calls.clear()
        got = sample_tiles_sharded(7, run, (1, 4, 4), "cpu", per)
        assert torch.equal(got, want)
        assert sorted(sum(calls, [])) == list(range(7))[rank::world]
        assert all(len(c) <= (per or 7) for c in calls)
    torch.save(True, os.path.join(out_dir, f"ok{rank}.pt"))


def test_tile_sharding_is_world_size_invariant(tmp_path):
    """rank r runs tiles r::world, one all-reduce merges them: the merged tile set equals the single-process result, for ragged
    counts, for tiles_per_batch smaller than a rank's share, and no collective runs while tiles are being sampled"""
    from sbgm_danra_amd.tiling import sample_tiles_sharded
    _spawn(_tile_shard_case, tmp_path)
    assert (tmp_path / "ok0.pt").exists() and (tmp_path / "ok1.pt").exists()
    assert torch.equal(sample_tiles_sharded(7, _stub_tiles, (1, 4, 4), "cpu", 3), _stub_tiles(list(range(7))))      # world size 1


def test_graft_entry_build_hook():
    """The driver's build hook and the first command of INTEGRATION.md: make (a no-op when current) + dlopen + ABI check."""
    import __graft_entry__ as G
    G.build()


def test_attention_dropout_is_accepted_like_the_reference_signature():
    """ImageSelfAttention(..., dropout=p) (reference score_unet.py:118-127) constructs with p > 0 — the identity in eval mode, which the
    sampling path uses; train mode runs the masked attention core on the autograd path (tests/test_gpu_backward.py)"""
    import sbgm_danra_amd as S
    a = S.ImageSelfAttention(64, 4, dropout=0.1)
    assert a.dropout == 0.1 and a.mha.dropout == 0.1
    assert set(a.state_dict()) == set(S.ImageSelfAttention(64, 4).state_dict())
    with pytest.raises(ValueError):
        S.ImageSelfAttention(30, 4)
