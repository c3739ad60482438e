"""Import-path alias: `sbgm.score_unet`, `sbgm.score_sampling`, `sbgm.training`, `sbgm.training_utils`,
`sbgm.training_main`, `sbgm.utils`, `sbgm.special_transforms`, `sbgm.cli.{main_app,launch_sbgm,launch_generation}`,
`sbgm.evaluate_sbgm.{generation,generation_main}` resolve to the MI355X-native implementations in
`sbgm_danra_amd`, so code written against the reference's module paths runs unchanged."""
import importlib
import sys

_ALIASES = ["score_unet", "score_sampling", "training", "training_utils", "training_main", "utils", "special_transforms", "cli",
            "cli.main_app", "cli.launch_sbgm", "cli.launch_generation", "evaluate_sbgm", "evaluate_sbgm.generation",
            "evaluate_sbgm.generation_main"]
for _name in _ALIASES:
    sys.modules[f"sbgm.{_name}"] = importlib.import_module(f"sbgm_danra_amd.{_name}")
    if "." not in _name:
        globals()[_name] = sys.modules[f"sbgm.{_name}"]
