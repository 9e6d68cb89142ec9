#!/bin/bash
# Round-2 profile collection on an MI355X box (run from the repo root through gpurun:
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles_r02.sh v1').
# Kernel stats and PMC counters come from SEPARATE rocprofv3 runs (never --pmc together with trace domains other than
# --kernel-trace); the program follows "--" directly as `python3 script`.  Results land in gpurun_out/profiles_r02_<tag>/;
# the summaries to be judged are copied into profiles/ by hand (tracked).
set -e
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/profiles_r02_$TAG
mkdir -p "$OUT"
cd "$ROOT"
echo "== ubench"; ./tools/ubench/pub > "$OUT/r02_valu_ubench_$TAG.txt" 2>&1 || true
echo "== bench sync"; python bench.py --steps 10 --warmup 2 > "$OUT/r02_bench_sync_$TAG.json"
echo "== bench outer 2^16"; python bench.py --workload outer --steps 24 --warmup 3 --no-cpu-baseline > "$OUT/r02_bench_outer_2p16_$TAG.json"
python bench.py --workload outer --steps 12 --warmup 3 --inflight 1 --no-cpu-baseline > "$OUT/r02_bench_outer_2p16_single_stream_$TAG.json"
for ln in 13 18 20; do python bench.py --workload outer --log-n $ln --steps 8 --warmup 2 --inflight 1 --no-cpu-baseline > "$OUT/r02_bench_outer_2p${ln}_single_stream_$TAG.json"; done
echo "== verify128"; python bench.py --workload verify128 --steps 2 --warmup 1 > "$OUT/r02_verify128_1gpu_$TAG.json"
python bench.py --workload verify128 --map-log-n 15 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/r02_verify128_1gpu_map2p15_$TAG.json"
echo "== ntt24"; python bench.py --workload ntt24 --steps 5 --warmup 1 > "$OUT/r02_ntt24_$TAG.json"
python bench.py --workload ntt24 --ntt-field bn254 --steps 3 --warmup 1 > "$OUT/r02_ntt24_bn254_2p24_$TAG.json"
echo "== msm24"; python bench.py --workload msm24 --steps 5 --warmup 1 > "$OUT/r02_msm24_$TAG.json"
python bench.py --workload msm24 --ntt-log-n 20 --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/r02_msm_2p20_$TAG.json"
echo "== starks"; python bench.py --workload ed25519 --log-slots 10 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/r02_ed25519_2p10_slots_$TAG.json"
python bench.py --workload sha256 --log-blocks 14 --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/r02_sha256_2p14_blocks_$TAG.json"
python bench.py --workload sha512 --log-blocks 14 --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/r02_sha512_2p14_blocks_$TAG.json"
cd /tmp && export TMPDIR=/tmp
echo "== rocprof sync"
rocprofv3 --kernel-trace --stats -d "$OUT/stats_sync" -o s --output-format csv -- python3 "$ROOT/bench.py" --steps 6 --warmup 1 --no-cpu-baseline --no-extra > /dev/null
echo "== rocprof outer single"
rocprofv3 --kernel-trace --stats -d "$OUT/stats_outer_single" -o s --output-format csv -- python3 "$ROOT/bench.py" --workload outer --steps 8 --warmup 2 --inflight 1 --no-cpu-baseline > /dev/null
echo "== rocprof msm24"
rocprofv3 --kernel-trace --stats -d "$OUT/stats_msm24" -o s --output-format csv -- python3 "$ROOT/bench.py" --workload msm24 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
cp "$OUT/stats_msm24/s_kernel_stats.csv" "$OUT/r02_msm24_kernel_stats_$TAG.csv"
echo "== pmc"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o p --output-format csv -- python3 "$ROOT/bench.py" --workload outer --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o p --output-format csv -- python3 "$ROOT/bench.py" --workload outer --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace -d "$OUT/pmc_insts" -o p --output-format csv -- python3 "$ROOT/bench.py" --workload outer --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null
cp "$OUT/stats_sync/s_kernel_stats.csv" "$OUT/r02_bench_sync_kernel_stats_$TAG.csv"
cp "$OUT/stats_outer_single/s_kernel_stats.csv" "$OUT/r02_bench_outer_2p16_single_stream_kernel_stats_$TAG.csv"
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/r02_pmc_traffic_$TAG.json" || true
ls "$OUT"
