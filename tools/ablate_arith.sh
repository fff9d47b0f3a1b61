#!/bin/bash
# Ablation of the eight-wide light kernel on the arithmetic script (4 streams in, 4 out, 64 Mi instances): the debug kernel with
# parts of a tile's work switched off (VVM_DEBUG_SKIP: 2 = no stream loads, 8 = no output stores, 32 = no look-back, 1 = no tile
# program at all, 128 = in-kernel clocks), kernel time per configuration.
OPS=${1:-1,16}
for skip in 0 128 2 8 10 32 42 1; do
  echo "VVM_DEBUG_SKIP=$skip"
  VVM_DEBUG_SKIP=$skip VVM_DEBUG_STATS=1 timeout -k 10 120 python tools/bench_configs.py --which arith --arith-ops $OPS --steps 10 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        if 'kernel_ms' in d: print('   ', d['config'][:22], round(d['kernel_ms'], 4))
    elif l.startswith('vvm stats'): last = l
try: print('   ', last)
except NameError: pass
" || exit 1
done
