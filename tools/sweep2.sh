#!/bin/bash
for st in 1 2 3; do for mb in 0 64; do
  r=$(timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams $st --micro-batch $mb 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "streams $st mb $mb : img/s ms/step = $r"
done; done
