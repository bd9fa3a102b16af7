#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_dict.py -x -q -m gpu 2>&1 | tail -8
