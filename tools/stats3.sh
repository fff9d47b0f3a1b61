VVM_DEBUG_STATS=1 VVM_DEBUG_SKIP=128 timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2> gpurun_out/stats.err > gpurun_out/stats.out || exit 1
python - <<'PY'
import re, json
rows=[list(map(int,re.findall(r'\d+', l.split('stats (cumulative):')[1]))) for l in open('gpurun_out/stats.err') if 'vvm stats' in l]
a,b=rows[-2],rows[-1]
d=[y-x for x,y in zip(a,b)]
out=json.loads(open('gpurun_out/stats.out').read().strip().splitlines()[-1])
kms=out['roofline']['kernel_ms']
names=["lookbacks","rounds","polls","lb_clk16","ticket_clk16","load_clk16","interp_clk16","tiles"]
print(dict(zip(names,d)), "kernel_ms", kms)
tiles=d[7]
blocks=512
tot=(d[4]+d[5]+d[6])*16
print("sum clocks per block %.0f -> clock MHz if == kernel time: %.1f" % (tot/blocks, tot/blocks/(kms*1e3)))
print("per tile clocks: ticket %.0f load %.0f interp(+acquire+resolve+store+stage) %.0f ; lookback per lb %.0f (inside interp)" % (16*d[4]/tiles, 16*d[5]/tiles, 16*d[6]/tiles, 16*d[3]/max(d[0],1)))
PY
