#!/usr/bin/env python3
"""HBM roofline of the calibration kernels on a test-split sized batch (one volume is launch-bound,
the eval script processes 160 subjects): python tools/calib_bench.py [volumes]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == '__main__':
    v = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    print(json.dumps(bench.calibration_kernels(torch.device('cuda'), volumes=v), indent=1))
