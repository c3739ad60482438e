"""Callers either side of the path, on the GPU: the CLI `generate` flow (checkpoint in the reference's format ->
SampleGenerator -> pc_sampler -> npz files) and the validation half of TrainingPipeline_general."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def cfg_path(tmp_path, monkeypatch):
    for k in ("DATA_DIR", "CKPT_DIR", "SAMPLE_DIR", "STATS_LOAD_DIR"):
        monkeypatch.setenv(k, str(tmp_path / k.lower()))
    monkeypatch.setenv("SLURM_CPUS_PER_TASK", "2")
    raw = yaml.safe_load(open(os.path.join(ROOT, "sbgm_danra_amd", "config", "default_config.yaml")))
    raw["highres"]["data_size"] = [64, 64]
    raw["lowres"]["data_size"] = [64, 64]
    raw["lowres"]["condition_variables"] = ["temp", "prcp"]
    raw["stationary_conditions"]["geographic_conditions"]["sample_w_geo"] = True
    raw["stationary_conditions"]["seasonal_conditions"]["sample_w_cond_season"] = True
    raw["sampler"]["n_timesteps"] = 4
    raw["evaluation"].update(batch_size=3, gen_type=["multiple", "single", "repeated"], n_repeats=2)
    raw["training"]["batch_size"] = 2
    p = tmp_path / "run.yaml"
    p.write_text(yaml.safe_dump(raw))
    return str(p)


def test_cli_generate_from_reference_format_checkpoint(cfg_path):
    from oracle import torch_ref as O
    from sbgm.cli import main_app                      # reference module path, resolves to the native package
    from sbgm.utils import get_model_string, load_config
    cfg = load_config(cfg_path)
    # a checkpoint as the reference writes it: {'network_params': state_dict, 'optimizer_params': ...}
    ora = O.build_scorenet(6, num_classes=4)
    ckpt_dir = os.path.join(cfg.paths.path_save, cfg.paths.checkpoint_dir)
    os.makedirs(ckpt_dir, exist_ok=True)
    torch.save({"network_params": O.synth_state_dict(ora), "optimizer_params": {}}, os.path.join(ckpt_dir, get_model_string(cfg) + ".pth.tar"))
    main_app.main(["--config_path", cfg_path, "--mode", "generate"])
    out = os.path.join(cfg.paths.sample_dir, "generation", get_model_string(cfg), "generated_samples")
    files = sorted(os.listdir(out))
    assert {"gen_samples_multi_n_3.npz", "gen_samples_single.npz", "gen_samples_repeated_n_2.npz", "eval_samples_multi_n_3.npz",
            "seasons_multi_n_3.npz"} <= set(files)
    g = np.load(os.path.join(out, "gen_samples_multi_n_3.npz"))["arr_0"]
    assert g.shape == (3, 64, 64) and np.isfinite(g).all()
    assert np.load(os.path.join(out, "gen_samples_single.npz"))["arr_0"].shape == (1, 64, 64)
    with pytest.raises(RuntimeError):                   # reference main_app.py:65-66
        os.remove(os.path.join(ckpt_dir, get_model_string(cfg) + ".pth.tar"))
        main_app.main(["--config_path", cfg_path, "--mode", "generate"])


def test_pipeline_validation_and_checkpoint_roundtrip(cfg_path):
    from sbgm.score_unet import diffusion_coeff_fn, loss_fn, marginal_prob_std_fn
    from sbgm.training import TrainingPipeline_general
    from sbgm.training_utils import get_dataloader, get_model, get_optimizer
    from sbgm.utils import load_config
    cfg = load_config(cfg_path)
    torch.manual_seed(0)
    model, _, _ = get_model(cfg)
    pipe = TrainingPipeline_general(model, loss_fn, marginal_prob_std_fn, diffusion_coeff_fn, get_optimizer(cfg, model),
                                    torch.device("cuda"), None, cfg)
    _, val_dl, gen_dl = get_dataloader(cfg)
    v = pipe.validate_batches(val_dl, verbose=False)
    assert np.isfinite(v) and v > 0
    pipe.save_model(pipe.checkpoint_dir, pipe.checkpoint_name)
    before = {k: t.clone() for k, t in model.state_dict().items()}
    with torch.no_grad():
        model.decoder.final_layer.conv.weight.mul_(0)
    pipe.load_checkpoint(pipe.checkpoint_path)
    assert all(torch.equal(before[k], t) for k, t in model.state_dict().items())
    ck = torch.load(pipe.checkpoint_path, weights_only=True)
    assert set(ck) == {"network_params", "optimizer_params"}
    gen = pipe.generate_and_plot_samples(gen_dl, cfg=cfg, epoch=1)
    assert gen.shape[1:] == (1, 64, 64) and torch.isfinite(gen).all()


def test_cli_train_one_epoch_then_generate(cfg_path):
    """`--mode full_pipeline` end to end on synthetic batches: native forward+backward, Adam, best-val checkpoint in the
    reference's format, then generation from that checkpoint."""
    from sbgm.cli import main_app
    from sbgm.utils import get_model_string, load_config
    cfg = load_config(cfg_path)
    main_app.main(["--config_path", cfg_path, "--mode", "full_pipeline", "--skip_evaluation"])
    ckpt = os.path.join(cfg.paths.path_save, cfg.paths.checkpoint_dir, get_model_string(cfg) + ".pth.tar")
    # checkpoint_dir is absolute in this config, so path_save/checkpoint_dir == checkpoint_dir
    assert os.path.exists(ckpt)
    ck = torch.load(ckpt, weights_only=True)
    assert set(ck) == {"network_params", "optimizer_params"} and len(ck["network_params"]) == 232
    assert all(torch.isfinite(v).all() for v in ck["network_params"].values() if v.dtype.is_floating_point)
    out = os.path.join(cfg.paths.sample_dir, "generation", get_model_string(cfg), "generated_samples")
    assert np.isfinite(np.load(os.path.join(out, "gen_samples_multi_n_3.npz"))["arr_0"]).all()
    losses = os.path.join(cfg.paths.path_save, "samples", get_model_string(cfg), "losses", f"losses_{get_model_string(cfg)}.pkl")
    assert os.path.exists(losses)


def test_generation_back_transform_on_device(cfg_path, tmp_path):
    """with evaluation.transform_back the saved samples are in physical units; the back-transform (built
    from the saved global statistics like the reference's generation_main.py:93-108) runs on the device before the one
    device->host copy.  Checked against the oracle's transform of the raw samples of an identical run."""
    import json
    import yaml as _yaml
    from oracle import torch_ref as O
    from oracle import transforms_ref as OT
    from sbgm.evaluate_sbgm.generation_main import generation_main
    from sbgm.utils import get_model_string, load_config
    raw = _yaml.safe_load(open(cfg_path))
    raw["evaluation"].update(gen_type=["multiple"], transform_back=True)
    raw["lowres"]["scaling_methods"] = ["zscore", "log_zscore"]
    p2 = tmp_path / "run_bt.yaml"
    p2.write_text(_yaml.safe_dump(raw))
    cfg = load_config(str(p2))
    stats = dict(mean=281.5, std=9.25, min=0.0, max=120.0, log_mean=-1.1, log_std=1.9, log_min=-4.6, log_max=5.5)
    for model, var in (("DANRA", "prcp"), ("ERA5", "temp"), ("ERA5", "prcp")):
        d = os.path.join(cfg.paths.stats_load_dir, model, var, "all")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"global_stats__{model}__589x789__crop__170_350_340_520__{var}__all.json"), "w") as f:
            json.dump(stats, f)
    ora = O.build_scorenet(6, num_classes=4)
    ckpt_dir = os.path.join(cfg.paths.path_save, cfg.paths.checkpoint_dir)
    os.makedirs(ckpt_dir, exist_ok=True)
    torch.save({"network_params": O.synth_state_dict(ora), "optimizer_params": {}}, os.path.join(ckpt_dir, get_model_string(cfg) + ".pth.tar"))
    out = os.path.join(cfg.paths.sample_dir, "generation", get_model_string(cfg), "generated_samples")
    bt = generation_main(cfg)["multiple"]
    cond_t = np.load(os.path.join(out, "cond_samples_temp_multi_n_3.npz"))["arr_0"]
    cfg.evaluation.transform_back = False
    plain = generation_main(cfg)["multiple"]                                   # same seed -> same raw samples
    want = OT.PrcpLogBackTransform(scale_type="log_zscore", glob_mean_log=-1.1, glob_std_log=1.9, glob_min_log=-4.6,
                                   glob_max_log=5.5, buffer_frac=0.5, clamp_log_min=-4.6, clamp_log_max=5.5)(plain)
    assert bt.shape == (3, 64, 64) and (bt > 0).all()
    assert float(((bt - want).abs() / want.abs().clamp_min(1e-6)).max()) <= 2e-6
    assert cond_t.shape == (3, 64, 64) and abs(float(cond_t.mean()) - 281.5) < 5.0      # z-scored N(0,1) field -> Kelvin-like


def test_training_with_device_side_condition_dropout(cfg_path):
    """SURVEY 8f rank 2 in the training loop: raw batches (1-channel geo fields) + classifier-free-guidance dropout are
    assembled on the device; the loss equals the loss on a batch that the oracle's per-sample dataset logic prepared."""
    from oracle import data_ref as OD
    from sbgm.score_unet import diffusion_coeff_fn, loss_fn, marginal_prob_std_fn
    from sbgm.training import TrainingPipeline_general
    from sbgm.training_utils import get_model, get_optimizer
    from sbgm.utils import load_config
    from sbgm_danra_amd.synthetic_data import synthetic_loader
    cfg = load_config(cfg_path)
    cfg.classifier_free_guidance.enabled = True
    cfg.monitoring.extreme_prcp.enabled = False
    torch.manual_seed(0)
    model, _, _ = get_model(cfg)
    pipe = TrainingPipeline_general(model, loss_fn, marginal_prob_std_fn, diffusion_coeff_fn, get_optimizer(cfg, model),
                                    torch.device("cuda"), None, cfg)
    batch = next(iter(synthetic_loader(cfg, 8, raw_geo=True)))
    assert batch["lsm"].shape[1] == 1
    # find a seed whose 8 draws drop at least one sample, then replay it for both paths
    seed = next(s for s in range(100) if (torch.manual_seed(s), sum(float(torch.rand(())) < 0.1 for _ in range(8)))[1] > 0)
    torch.manual_seed(seed)
    draws = [float(torch.rand(())) for _ in range(8)]
    t_z_state = torch.get_rng_state()
    torch.manual_seed(seed)                               # CPU generator: the 8 dropout draws (also reseeds the device ...)
    torch.cuda.manual_seed(5)                             # ... so the (t, z) generator is set afterwards
    model.eval()                                          # same BatchNorm statistics on both sides
    with torch.no_grad():
        _, got = pipe._loss(batch, "train")
        # oracle-prepared batch, same (t, z) draws: replay the RNG state right after the 8 dropout draws
        items = [{k: v[b].clone() for k, v in batch.items()} for b in range(8)]
        done = [OD.finish_sample(it, "train", {"enabled": True, "drop_prob": 0.2}, draws[b])[0] for b, it in enumerate(items)]
        prepared = {k: torch.stack([d[k] for d in done]) for k in done[0]}
        torch.set_rng_state(t_z_state)
        torch.cuda.manual_seed(5)
        x, seasons, cond, _h, lsm, _s, topo, _a, _b = __import__("sbgm_danra_amd.utils", fromlist=["x"]).extract_samples(prepared, "cuda")
        want = loss_fn(model, x, marginal_prob_std_fn, y=seasons, cond_img=cond, lsm_cond=lsm, topo_cond=topo)
    assert any(d < 0.1 for d in draws)
    assert torch.isfinite(got) and abs(float(got) - float(want)) <= 1e-5 * abs(float(want))
    model.train()
    assert np.isfinite(pipe.train_batches(synthetic_loader(cfg, 2, n_items=4, raw_geo=True), epochs=1, verbose=False))


def test_training_step_as_hip_graph(cfg_path):
    """training.use_hip_graph: forward + backward replayed as one graph; parameters move, the loss stays finite, BatchNorm
    counters advance by exactly one per step (the capture warm-up is rolled back), and a second epoch reuses the graph."""
    from sbgm.score_unet import diffusion_coeff_fn, loss_fn, marginal_prob_std_fn
    from sbgm.training import TrainingPipeline_general
    from sbgm.training_utils import get_model, get_optimizer
    from sbgm.utils import load_config
    from sbgm_danra_amd.synthetic_data import synthetic_loader
    cfg = load_config(cfg_path)
    cfg.training.use_hip_graph = True
    cfg.monitoring.extreme_prcp.enabled = False
    torch.manual_seed(0)
    model, _, _ = get_model(cfg)
    pipe = TrainingPipeline_general(model, loss_fn, marginal_prob_std_fn, diffusion_coeff_fn, get_optimizer(cfg, model),
                                    torch.device("cuda"), None, cfg)
    before = model.encoder.conv2.weight.detach().clone()
    dl = synthetic_loader(cfg, 2, n_items=6)
    a = pipe.train_batches(dl, epochs=2, current_epoch=1, verbose=False)
    assert np.isfinite(a) and int(model.encoder.bn1.num_batches_tracked) == 3
    assert not torch.equal(before, model.encoder.conv2.weight)
    b = pipe.train_batches(dl, epochs=2, current_epoch=2, verbose=False)
    assert np.isfinite(b) and int(model.encoder.bn1.num_batches_tracked) == 6 and len(pipe._graphs) == 1
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) < 1e8 for k, p in model.named_parameters()
               if not k.startswith("decoder.final_layer.time_"))


def test_hip_graph_is_the_default_and_falls_back_when_a_step_cannot_be_captured(cfg_path):
    """training.use_hip_graph defaults to `auto` (the reference's YAML files do not have the key): the step is captured without being
    asked for; a loss function with a host synchronisation inside (not capturable) makes the pipeline fall back to eager steps —
    with the model as it was before the attempt — instead of failing."""
    from sbgm.score_unet import diffusion_coeff_fn, loss_fn, marginal_prob_std_fn
    from sbgm.training import TrainingPipeline_general
    from sbgm.training_utils import get_model, get_optimizer
    from sbgm.utils import load_config
    from sbgm_danra_amd.synthetic_data import synthetic_loader
    cfg = load_config(cfg_path)
    del cfg["training"]["use_hip_graph"]
    cfg.monitoring.extreme_prcp.enabled = False
    torch.manual_seed(0)
    model, _, _ = get_model(cfg)
    pipe = TrainingPipeline_general(model, loss_fn, marginal_prob_std_fn, diffusion_coeff_fn, get_optimizer(cfg, model),
                                    torch.device("cuda"), None, cfg)
    dl = synthetic_loader(cfg, 2, n_items=4)
    a = pipe.train_batches(dl, epochs=1, current_epoch=1, verbose=False)
    assert np.isfinite(a) and len(pipe._graphs) == 1 and not getattr(pipe, "_graph_failed", False)

    def syncing_loss(model_, x, mps, **kw):
        out = loss_fn(model_, x, mps, **kw)
        float(out.detach().sum())                       # a host read inside the step: illegal under stream capture
        return out
    torch.manual_seed(0)
    model2, _, _ = get_model(cfg)
    pipe2 = TrainingPipeline_general(model2, syncing_loss, marginal_prob_std_fn, diffusion_coeff_fn, get_optimizer(cfg, model2),
                                     torch.device("cuda"), None, cfg)
    before = model2.encoder.conv2.weight.detach().clone()
    b = pipe2.train_batches(dl, epochs=1, current_epoch=1, verbose=False)
    assert np.isfinite(b) and pipe2._graph_failed and not getattr(pipe2, "_graphs", {})
    assert int(model2.encoder.bn1.num_batches_tracked) == 2 and not torch.equal(before, model2.encoder.conv2.weight)
    c = pipe2.train_batches(dl, epochs=2, current_epoch=2, verbose=False)          # stays eager, keeps training
    assert np.isfinite(c) and int(model2.encoder.bn1.num_batches_tracked) == 4
    # train-mode attention dropout draws a fresh seed per call: auto runs such a model eagerly from the start
    torch.manual_seed(0)
    model4, _, _ = get_model(cfg)
    att = [m for m in model4.modules() if type(m).__name__ == "ImageSelfAttention"]
    att[0].dropout = att[-1].dropout = 0.2
    pipe4 = TrainingPipeline_general(model4, loss_fn, marginal_prob_std_fn, diffusion_coeff_fn, get_optimizer(cfg, model4),
                                     torch.device("cuda"), None, cfg)
    d = pipe4.train_batches(dl, epochs=1, current_epoch=1, verbose=False)
    assert np.isfinite(d) and not getattr(pipe4, "_graphs", {}) and not getattr(pipe4, "_graph_failed", False)
    # an explicit `true` does not hide the problem
    cfg.training.use_hip_graph = True
    pipe3 = TrainingPipeline_general(model2, syncing_loss, marginal_prob_std_fn, diffusion_coeff_fn, get_optimizer(cfg, model2),
                                     torch.device("cuda"), None, cfg)
    with pytest.raises(Exception):
        pipe3.train_batches(dl, epochs=1, current_epoch=1, verbose=False)
    torch.cuda.synchronize()


def _graph_pipe(cfg_path, **training_overrides):
    from sbgm.score_unet import diffusion_coeff_fn, loss_fn, marginal_prob_std_fn
    from sbgm.training import TrainingPipeline_general
    from sbgm.training_utils import get_model, get_optimizer
    from sbgm.utils import load_config
    cfg = load_config(cfg_path)
    cfg.training.use_hip_graph = True
    cfg.monitoring.extreme_prcp.enabled = False
    for k, v in training_overrides.items():
        setattr(cfg.training, k, v)
    torch.manual_seed(0)
    model, _, _ = get_model(cfg)
    pipe = TrainingPipeline_general(model, loss_fn, marginal_prob_std_fn, diffusion_coeff_fn, get_optimizer(cfg, model),
                                    torch.device("cuda"), None, cfg)
    return cfg, model, pipe


def test_evaluation_after_graph_training_sees_the_trained_weights(cfg_path):
    """hipGraph replays and the native Adam step change weights and BatchNorm buffers without touching torch's version counters;
    the inference engine must still re-upload them: after graph-mode training, model.eval() must equal a FRESH ScoreNet loaded
    from model.state_dict() (validation and best-checkpoint selection depend on it)."""
    from sbgm.training_utils import get_model
    from sbgm_danra_amd.synthetic_data import synthetic_loader
    from sbgm_danra_amd.utils import extract_samples
    cfg, model, pipe = _graph_pipe(cfg_path)
    dl = synthetic_loader(cfg, 2, n_items=4)
    batch = next(iter(dl))
    x, seasons, cond, _h, lsm, _s, topo, _a, _b = extract_samples(batch, "cuda")
    t = torch.tensor([0.3, 0.7], device="cuda")
    model.eval()
    with torch.no_grad():
        before = model(x, t, seasons, cond, lsm, topo).clone()           # the engine uploads the initial weights here
    pipe.train_batches(dl, epochs=2, current_epoch=1, verbose=False)      # capture + replays + native Adam steps
    pipe.train_batches(dl, epochs=2, current_epoch=2, verbose=False)      # replays only
    model.eval()
    with torch.no_grad():
        after = model(x, t, seasons, cond, lsm, topo)
    fresh, _, _ = get_model(cfg)
    fresh.load_state_dict(model.state_dict())
    fresh.eval()
    with torch.no_grad():
        want = fresh(x, t, seasons, cond, lsm, topo)
    assert not torch.equal(before, after)
    assert torch.equal(after, want)
    # and the pipeline's own validation runs on the current weights
    v1 = pipe.validate_batches(dl, verbose=False)
    pipe.train_batches(dl, epochs=3, current_epoch=3, verbose=False)
    v2 = pipe.validate_batches(dl, verbose=False)
    assert np.isfinite(v1) and np.isfinite(v2)


def test_graph_steps_with_two_batch_shapes_use_their_own_gradients(cfg_path):
    """a ragged last batch captures a second graph; replaying the FIRST graph afterwards must hand the optimizer the gradients
    that replay wrote (each capture allocates its own .grad tensors): interleaved graph steps == eager steps"""
    from sbgm_danra_amd.synthetic_data import synthetic_loader
    from sbgm_danra_amd.utils import extract_samples
    cfg, model, pipe = _graph_pipe(cfg_path)
    model.train()
    big = next(iter(synthetic_loader(cfg, 3, n_items=3, seed=5)))
    small = next(iter(synthetic_loader(cfg, 1, n_items=1, seed=6)))

    def eager_grads(batch, noise):
        for p in model.parameters():
            p.grad = None
        x, seasons, cond, _h, lsm, sdf, topo, _a, _b = extract_samples(batch, "cuda")
        pipe.loss_fn(model, x, pipe.marginal_prob_std_fn, y=seasons, cond_img=cond, lsm_cond=lsm, topo_cond=topo,
                     sdf_cond=sdf if pipe.sdf_weighted_loss else None, noise=noise).backward()
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    # graph steps draw their noise on the device; read it back through a wrapper model so the eager run can reuse it
    import sbgm_danra_amd.score_unet as SU
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    seq = [big, small, big, small, big]
    for i, batch in enumerate(seq):
        model.load_state_dict(sd0)                      # same weights for every comparison (no optimizer step in between)
        _, loss = pipe._graph_step(batch)
        torch.cuda.synchronize()
        got = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        assert torch.isfinite(loss)
        # recompute the same step eagerly: reproduce the graph's draw (seed, offset - 1 after the loss advanced it)
        st = SU._loss_rng_state(torch.device("cuda", torch.cuda.current_device()))
        seed, off = int(st[0]), int(st[1]) - 1
        x = extract_samples(batch, "cuda")[0]
        B, per = x.shape[0], x[0].numel()
        lib = SU.N.lib()
        state = torch.tensor([seed, off], dtype=torch.int64, device="cuda")
        xp, z, tt, sd_ = torch.empty_like(x), torch.empty_like(x), torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
        SU.N.check(lib.sbgm_dsm_perturb(x.data_ptr(), None, None, state.data_ptr(), 0, 1e-3, 25.0, xp.data_ptr(), z.data_ptr(), tt.data_ptr(),
                                        sd_.data_ptr(), B, per, SU.N.stream()))
        model.load_state_dict(sd0)
        want = eager_grads(batch, (tt, z))
        assert got.keys() == want.keys()
        worst = max(float((got[k] - want[k]).abs().max() / want[k].abs().max().clamp_min(1e-30)) for k in want)
        assert worst < 1e-4, (i, worst)
    assert len(pipe._graphs) == 2


def test_cli_generation_on_two_ranks_writes_rank_owned_batches(cfg_path):
    """`--mode generate` under two ranks (gloo rehearsal on one GPU): every rank samples its own conditioning batch with its own noise and
    writes *_rank<r> files — no two ranks write one path, the two batches differ, the repeats are split between the ranks"""
    import socket
    import subprocess
    import sys
    from oracle import torch_ref as O
    from sbgm.utils import get_model_string, load_config
    cfg = load_config(cfg_path)
    ora = O.build_scorenet(6, num_classes=4)
    ckpt_dir = os.path.join(cfg.paths.path_save, cfg.paths.checkpoint_dir)
    os.makedirs(ckpt_dir, exist_ok=True)
    torch.save({"network_params": O.synth_state_dict(ora), "optimizer_params": {}}, os.path.join(ckpt_dir, get_model_string(cfg) + ".pth.tar"))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = "import sys; sys.path.insert(0, %r); from sbgm.cli import main_app; main_app.main(['--config_path', %r, '--mode', 'generate'])" % (ROOT, cfg_path)
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    base["SBGM_DIST_BACKEND"] = "gloo"
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                                                                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    out = os.path.join(cfg.paths.sample_dir, "generation", get_model_string(cfg), "generated_samples")
    files = set(os.listdir(out))
    want = {f"gen_samples_multi_n_3_rank{r}.npz" for r in range(2)} | {f"gen_samples_single_rank{r}.npz" for r in range(2)} \
        | {f"gen_samples_repeated_n_2_rank{r}.npz" for r in range(2)}
    assert want <= files and "gen_samples_multi_n_3.npz" not in files, sorted(files)
    a, b = (np.load(os.path.join(out, f"gen_samples_multi_n_3_rank{r}.npz"))["arr_0"] for r in range(2))
    assert a.shape == b.shape == (3, 64, 64) and np.isfinite(a).all() and np.isfinite(b).all() and not np.array_equal(a, b)
    ea, eb = (np.load(os.path.join(out, f"eval_samples_multi_n_3_rank{r}.npz"))["arr_0"] for r in range(2))
    assert not np.array_equal(ea, eb)                                  # different conditioning batches
    reps = [np.load(os.path.join(out, f"gen_samples_repeated_n_2_rank{r}.npz"))["arr_0"] for r in range(2)]
    assert sum(r_.shape[0] if r_.ndim == 3 else 1 for r_ in reps) == 2


def test_cli_training_on_two_ranks_keeps_the_replicas_identical(cfg_path, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m sbgm.cli.main_app --mode train` rehearsed on this box's one GPU over gloo:
    the pipeline's default (captured step, gradient all-reduce on the arena after each replay, 1/world folded into Adam, rank-sharded
    loader, rank-distinct noise) must leave both replicas with bit-identical weights."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = ("import sys, torch; sys.path.insert(0, %r); from sbgm.cli import main_app; from sbgm.utils import load_config; import os\n"
            "cfg = load_config(%r)\n"
            "from sbgm_danra_amd.training_main import train_main\n"
            "cfg.training.epochs = 1; cfg.monitoring.extreme_prcp.enabled = False\n"
            "import sbgm_danra_amd.training as TR\n"
            "orig = TR.TrainingPipeline_general.train_batches\n"
            "def spy(self, *a, **k):\n"
            "    r = orig(self, *a, **k)\n"
            "    torch.save({'sd': {k_: v.cpu() for k_, v in self.model.state_dict().items()}, 'graphs': len(getattr(self, '_graphs', {}))},\n"
            "               os.path.join(%r, 'rank%%s.pt' %% os.environ.get('RANK', '0')))\n"
            "    return r\n"
            "TR.TrainingPipeline_general.train_batches = spy\n"
            "train_main(cfg)\n") % (ROOT, cfg_path, str(tmp_path))
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    base["SBGM_DIST_BACKEND"] = "gloo"
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                                                                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    a, b = (torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=True) for r in range(2))
    assert a["graphs"] >= 1 and b["graphs"] >= 1                      # the captured step was the default on both ranks
    for k in a["sd"]:
        if "running_" in k or "num_batches" in k:                      # per-replica BatchNorm statistics (DistributedDataParallel's default)
            continue
        assert torch.equal(a["sd"][k], b["sd"][k]), k
    assert all(torch.isfinite(v).all() for v in a["sd"].values() if v.dtype.is_floating_point)


@pytest.mark.parametrize("mode,extra", [("sample", ["--batch", "2", "--size", "64"]), ("train", ["--batch", "2", "--size", "64"]),
                                        ("train", ["--batch", "2", "--size", "64", "--sync-bn"]), ("domain", [])])
def test_bench_multi_rank_rehearsal_on_one_gpu(mode, extra):
    """`python bench.py --gpus 2` as the driver calls it (no torchrun environment): bench.py starts the 2 rank processes itself; here
    they share this box's one GPU and talk over gloo (SBGM_DIST_BACKEND) — every multi-rank code path except RCCL itself: rank
    spawn, process group, per-rank seeds, barrier + max-over-ranks timing, gradient all-reduce on the arena, SyncBatchNorm,
    tile sharding + merge, rank-0-only JSON line"""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SBGM_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--mode", mode, "--no-cpu-baseline",
           "--no-autotune", "--no-secondary"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0 and np.isfinite(out["value"])
    if mode == "sample":
        assert out["config"]["global_batch"] == 4 and out["scaling"] == "weak"
    if mode == "train":
        assert np.isfinite(out["config"]["final_loss"]) and ("SyncBatchNorm" in out["config"]["workload"]) == ("--sync-bn" in extra)
