#!/bin/bash
# Real per-item cost of k_quotient's work items (19 gates in gate-list order, PoseidonGate also by its three parts, then the
# permutation argument of each challenge): a calibration build
#   NLX_BUILD_VARIANT=qcal NLX_EXTRA_FLAGS=-DNLX_QUOTIENT_CALIBRATE python near-light-client_amd/build.py
# lets every wave evaluate ONE item; the quotient kernel's time then ranks the items (2^16 rows = 2^19 points: the unit of
# prover.hip gate_eval_cost).  gpurun -- 'bash tools/quotient_calibrate.sh'  -> gpurun_out/r03/quotient_calibrate.txt
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03/quotient_calibrate.txt
: > $OUT
for k in $(seq 0 20) 65554 131090 262162; do
  NLX_Q_ONLY=$k NLX_BUILD_VARIANT=qcal python bench.py --workload outer --log-n 16 --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('item %6d  k_quotient %.3f ms  (stage %.3f)' % ($k, d['kernel_ms_per_proof']['quotient'], d['stage_ms_last_proof']['quotient_eval']))" >> $OUT
done
cat $OUT
