#!/bin/bash
# timing-only ablations of the coarse kernel (results are wrong for ABL != 0)
for abl in 0 1; do
  RCN_COARSE_ABL=$abl timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ABL=$abl step_ms=%.2f coarse_ms=%.3f TF=%.0f frac=%.3f rerank_ms=%.2f'%(d['ms_per_step'], r['launch_ms'], r['achieved'], r['frac'], r['rerank_ms']))"
done
