#!/bin/bash
# Diagnostic: starts two rank processes of tests/sharded_rank_process.py by hand (no pytest parent), with the HIP runtime's and the
# kernel driver interface's logging on, and keeps their output under gpurun_out/.
mkdir -p gpurun_out
export HSAKMT_DEBUG_LEVEL=${HSAKMT_DEBUG_LEVEL:-7} AMD_LOG_LEVEL=${AMD_LOG_LEVEL:-3}
PORT=29517
timeout -k 10 120 python -u tests/sharded_rank_process.py 0 2 $PORT 3 /tmp/diag_r0.json > gpurun_out/diag_r0.log 2>&1 &
P0=$!
timeout -k 10 120 python -u tests/sharded_rank_process.py 1 2 $PORT 3 /tmp/diag_r1.json > gpurun_out/diag_r1.log 2>&1 &
P1=$!
wait $P0; R0=$?
wait $P1; R1=$?
echo "rank 0 exit $R0, rank 1 exit $R1"
tail -c 3000 gpurun_out/diag_r0.log
exit $(( R0 | R1 ))
