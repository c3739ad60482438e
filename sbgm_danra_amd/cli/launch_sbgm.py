"""reference sbgm/cli/launch_sbgm.py:4-7"""
from ..training_main import train_main


def run(cfg):
    return train_main(cfg)
