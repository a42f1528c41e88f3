#!/bin/bash
for kr in 1 0; do for b in 1024 32 16 64; do
  if [ $kr = 1 ]; then export TZ_GS_KROWS=1; else unset TZ_GS_KROWS; fi
  TZ_LIB=tzddpc_amd/lib/ab/k1g.so timeout -k 10 300 python bench.py --config genstack_dim5_k1 --batch $b --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('appended_K_rows=$kr batch $b', 'ms', round(l['ms_per_step'],4), r['bound'], round(r['frac'],3), r.get('achieved'))"
done; done
