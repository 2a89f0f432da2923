#!/bin/bash
for s in 2048 4096 8192 32768; do
  UCF_GRID_SLOTS=$s timeout -k 10 300 python bench.py --steps 3 --warmup 1 --mode fast --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slots=$s', round(d['value']), d['roofline']['kernel_ms'])"
done
