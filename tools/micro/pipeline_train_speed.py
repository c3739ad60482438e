"""samples/s of TrainingPipeline_general.train_batches at the C3 per-GPU shape (128x128, 4 conditions, batch 8), hipGraph step on and
off, next to bench.py's secondary.train_c3 (which replays the step on one resident batch)."""
import sys, os, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, yaml
tmp = tempfile.mkdtemp()
for k in ("DATA_DIR", "CKPT_DIR", "SAMPLE_DIR", "STATS_LOAD_DIR"):
    os.environ[k] = os.path.join(tmp, k.lower())
raw = yaml.safe_load(open(os.path.join(ROOT, "sbgm_danra_amd", "config", "default_config.yaml")))
raw["highres"]["data_size"] = [128, 128]; raw["lowres"]["data_size"] = [128, 128]
raw["lowres"]["condition_variables"] = ["temp", "prcp", "ewvf", "nwvf"]
raw["stationary_conditions"]["geographic_conditions"]["sample_w_geo"] = False
raw["stationary_conditions"]["seasonal_conditions"]["sample_w_cond_season"] = False
raw["training"]["batch_size"] = 8
p = os.path.join(tmp, "run.yaml"); open(p, "w").write(yaml.safe_dump(raw))
from sbgm.score_unet import diffusion_coeff_fn, loss_fn, marginal_prob_std_fn
from sbgm.training import TrainingPipeline_general
from sbgm.training_utils import get_model, get_optimizer
from sbgm.utils import load_config
from sbgm_danra_amd.synthetic_data import synthetic_loader
for graph in (True, False):
    cfg = load_config(p)
    cfg.training.use_hip_graph = graph
    cfg.monitoring.extreme_prcp.enabled = False
    torch.manual_seed(0)
    model, _, _ = get_model(cfg)
    pipe = TrainingPipeline_general(model, loss_fn, marginal_prob_std_fn, diffusion_coeff_fn, get_optimizer(cfg, model), torch.device("cuda"), None, cfg)
    n_items = 8 * 60
    dl = synthetic_loader(cfg, 8, n_items=n_items)
    t0 = time.perf_counter(); coll = list(dl); print(f"loader alone: {(time.perf_counter() - t0) / len(coll) * 1e3:.2f} ms per batch (host randn)", flush=True)
    pipe.train_batches(dl, epochs=3, current_epoch=1, verbose=False)          # tuning, capture
    for ep, (name, src) in enumerate((("live loader", dl), ("live loader", dl), ("collated batches", coll), ("collated batches", coll)), 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = pipe.train_batches(src, epochs=5, current_epoch=ep, verbose=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"use_hip_graph={graph} {name}: {n_items / dt:.0f} samples/s, {dt / (n_items / 8) * 1e3:.2f} ms per step, loss {loss:.4g}", flush=True)
