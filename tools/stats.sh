# usage: [SKIP=n] tools/stats.sh [bench args]   -- per-tile averages from the kernel's profiling counters (debug bit 7)
VVM_DEBUG_STATS=1 VVM_DEBUG_SKIP=$(( ${SKIP:-0} | 128 )) timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2> gpurun_out/stats.err > gpurun_out/stats.out || exit 1
python - <<'PY'
import re, json
rows=[list(map(int,re.findall(r'\d+', l.split('stats (cumulative):')[1]))) for l in open('gpurun_out/stats.err') if 'vvm stats' in l]
a,b=rows[-2],rows[-1]
d=[y-x for x,y in zip(a,b)]
# fields: lookbacks rounds polls clocks/16 | ticket load interp tiles  (the '16' literal is parsed too)
lb, rounds, polls, _16, lbclk, ticket, load, interp, tiles = d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8]
print("kernel_ms", json.loads(open('gpurun_out/stats.out').read())['roofline']['kernel_ms'])
print("per look-back: rounds %.2f polls %.2f clocks %.0f" % (rounds/lb, polls/lb, 16*lbclk/lb))
print("per tile clocks: ticket %.0f load-phase %.0f interp %.0f  (tiles %d)" % (16*ticket/tiles, 16*load/tiles, 16*interp/tiles, tiles))
PY
