VVM_DEBUG_STATS=1 VVM_DEBUG_SKIP=$(( ${SKIP:-0} | 128 )) timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2> gpurun_out/stats.err > gpurun_out/stats.out || exit 1
python - <<'PY'
import re
rows=[list(map(int,re.findall(r'\d+', l.split('stats (cumulative):')[1]))) for l in open('gpurun_out/stats.err') if 'vvm stats' in l]
a,b=rows[-2],rows[-1]
d=[y-x for x,y in zip(a,b)]
lb=max(d[0],1); tiles=65536*20
print("lb clocks %.0f polls %.2f | per tile: ticket %.0f  issue(A) %.0f  resolve(A-B) %.0f  regwrite(B-C) %.0f" % (16*d[4]/lb, d[2]/lb, 16*d[5]/tiles, 16*d[6]/tiles, 16*d[7]/tiles, 16*d[8]/tiles))
PY
