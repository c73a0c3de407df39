#!/usr/bin/env python3
"""python classification/train.py -c train_configs/<config>.json   (reference: classification/train.py, same flag and JSON schema)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "myrtle-vision_amd"))

from myrtle_vision.engine import launch_training  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", type=str, help="JSON file for configuration")
    args = parser.parse_args()
    with open(args.config) as f:
        config = json.loads(f.read())
    launch_training(config, "classification")
