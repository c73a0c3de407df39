#!/usr/bin/env python3
"""python classification/test_quantize.py -c <config>.json [--calib_steps N] [--quantized_ckpt]
(reference: classification/test_quantize.py:145-156: prepare_qat -> calibrate -> convert -> evaluate)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "myrtle-vision_amd"))

from myrtle_vision.engine import evaluate  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", type=str, help="JSON file for configuration")
    parser.add_argument("--calib_steps", type=int, default=10, help="calibration forward passes before convert()")
    parser.add_argument("--quantized_ckpt", action="store_true", help="the checkpoint was saved from a prepared model")
    args = parser.parse_args()
    with open(args.config) as f:
        config = json.loads(f.read())
    evaluate(config, "classification", quantize=True, calib_steps=args.calib_steps, quantized_ckpt=args.quantized_ckpt)
