"""`python -m sbgm.cli.main_app --config_path X.yaml --mode {train,generate,full_pipeline}` — the reference's
dispatcher (sbgm/cli/main_app.py:42-90) for the modes on the hot path.  `evaluate` / `data_splits` belong to
out-of-scope subsystems and exit with a message."""
import argparse
import os

from ..utils import get_model_string, load_config
from . import launch_generation, launch_sbgm


def check_model_exists(cfg) -> bool:
    d = os.path.join(cfg["paths"]["path_save"], cfg["paths"]["checkpoint_dir"])
    return os.path.exists(os.path.join(d, get_model_string(cfg) + ".pth.tar"))


def main(argv=None):
    ap = argparse.ArgumentParser(description="SBGM full pipeline launcher (MI355X-native hot path)")
    ap.add_argument("--config_path", required=True)
    ap.add_argument("--mode", choices=["train", "generate", "evaluate", "full_pipeline", "data_splits"], default="full_pipeline")
    ap.add_argument("--skip_train", action="store_true")
    ap.add_argument("--skip_generation", action="store_true")
    ap.add_argument("--skip_evaluation", action="store_true")
    a = ap.parse_args(argv)
    cfg = load_config(a.config_path)
    if a.mode in ("evaluate", "data_splits"):
        raise SystemExit(f"mode '{a.mode}' is outside the accelerated hot path (see DESIGN.md, out of scope)")
    if a.mode == "train" or (a.mode == "full_pipeline" and not a.skip_train):
        launch_sbgm.run(cfg)
    if a.mode == "generate" or (a.mode == "full_pipeline" and not a.skip_generation):
        if not check_model_exists(cfg):
            raise RuntimeError("Cannot generate: model checkpoint not found")
        launch_generation.run(cfg)
    print("\nPipeline finished successfully.")


if __name__ == "__main__":
    main()
