#!/bin/bash
for C in 4 8 16 64; do
  echo -n "chunks=$C "
  TKMK_MSM_CHUNKS=$C timeout -k 10 300 python bench.py --logn 24 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print(round(d['ms_per_step'],2), k['hist'], k['scan'], k['scatter'], k['accumulate'])"
done
