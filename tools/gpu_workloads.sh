#!/bin/bash
mkdir -p gpurun_out
for w in c2 c2pp c3 c4 c5; do
  extra=""
  [ $w = c3 ] && extra="--nt 512 --nr 256"
  [ $w = c4 ] && extra="--nt 512 --nr 256"
  timeout -k 10 600 python bench.py --steps 2 --warmup 1 --mode ${MODE:-fast} --no-cpu --workload $w $extra > gpurun_out/wl_$w.log 2>&1; rc=$?
  echo "[$w] rc=$rc $(tail -1 gpurun_out/wl_$w.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'pts/s', round(d['roofline']['kernel_ms'],1),'ms', 'frac', round(d['roofline']['frac'],3))" 2>&1 | tail -1)"
  [ $rc -ge 124 ] && exit $rc
done
