#!/bin/bash
# Regenerates the files under profiles/ on an MI355X box (run from the repo root through gpurun, e.g.
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh v7').
# Kernel stats and PMC counters are collected in SEPARATE rocprofv3 runs (never --pmc together with trace domains
# other than --kernel-trace); the program follows "--" directly as `python3 script`.
set -e
TAG=${1:-vX}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/profiles_$TAG
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$OUT"
cd "$ROOT"
python bench.py --steps 24 --warmup 3 > "$OUT/r01_bench_default_$TAG.json"
python bench.py --steps 12 --warmup 3 --inflight 1 --no-cpu-baseline > "$OUT/r01_bench_single_stream_$TAG.json"
python bench.py --steps 24 --warmup 3 --gate-mix basic --no-cpu-baseline > "$OUT/r01_bench_basic_mix_$TAG.json"
python bench.py --workload verify128 --steps 3 --warmup 1 > "$OUT/r01_verify128_1gpu_$TAG.json"
python bench.py --workload stark --steps 18 --warmup 2 --no-cpu-baseline > "$OUT/r01_stark_256x2p16_$TAG.json"
python bench.py --workload sha256 --log-blocks 14 --steps 4 --warmup 1 > "$OUT/r01_sha256_2p14_blocks_$TAG.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats_default" -o s --output-format csv -- python3 "$ROOT/bench.py" --steps 18 --warmup 2 --no-cpu-baseline > /dev/null
rocprofv3 --kernel-trace --stats -d "$OUT/stats_single" -o s --output-format csv -- python3 "$ROOT/bench.py" --steps 8 --warmup 2 --inflight 1 --no-cpu-baseline > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace -d "$OUT/pmc_insts" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null
cp "$OUT/stats_default/s_kernel_stats.csv" "$OUT/r01_bench_default_kernel_stats_$TAG.csv"
cp "$OUT/stats_single/s_kernel_stats.csv" "$OUT/r01_bench_single_stream_kernel_stats_$TAG.csv"
ls "$OUT"
