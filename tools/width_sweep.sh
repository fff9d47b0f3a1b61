#!/bin/bash
# usage: tools/width_sweep.sh   (on the GPU box)  kernel time of the fountain script at the four- and the eight-wide kernel over sizes
for n in 131072 262144 524288 1048576 2097152 4194304 16777216; do
	for w in 4 8; do
		timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --particles $n --width $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, $w, '%.4f ms' % d['roofline']['kernel_ms'])" || exit 1
	done
done
