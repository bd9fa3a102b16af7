#!/bin/bash
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 1024 --grid-y 128 --no-cpu-baseline 2>&1 | grep -E "resident stamps" 
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 256 --pc jacobi --no-cpu-baseline 2>&1 | grep -E "resident stamps"
