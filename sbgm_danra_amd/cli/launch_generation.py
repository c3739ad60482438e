"""reference sbgm/cli/launch_generation.py:4-7"""
from ..evaluate_sbgm.generation_main import generation_main


def run(cfg):
    return generation_main(cfg)
