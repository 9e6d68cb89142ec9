#!/bin/bash
# Round-4 profile collection on an MI355X box (run from the repo root through gpurun:
#   gpurun --timeout 1150 -- 'bash tools/collect_profiles_r04.sh v2').
# Kernel stats and PMC counters come from SEPARATE rocprofv3 runs (never --pmc together with trace domains other than
# --kernel-trace); the program follows "--" directly as `python3 script`.  Results land in gpurun_out/profiles_r04_<tag>/;
# the summaries to be judged are copied into profiles/ by hand (tracked).
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/profiles_r04_$TAG
mkdir -p "$OUT"
cd "$ROOT"
echo "== ubench"; ./tools/ubench/pub > "$OUT/r04_poseidon_occupancy_$TAG.txt" 2>&1 || true
echo "== bench sync"; python bench.py --steps 10 --warmup 2 > "$OUT/r04_bench_sync_$TAG.json" 2> "$OUT/bench_sync.err"
echo "== bench outer"; python bench.py --workload outer --steps 24 --warmup 3 --no-cpu-baseline > "$OUT/r04_bench_outer_2p16_$TAG.json" 2>/dev/null
for ln in 13 16 18 20; do python bench.py --workload outer --log-n $ln --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline > "$OUT/r04_bench_outer_2p${ln}_single_stream_$TAG.json" 2>/dev/null; done
echo "== verify128"; python bench.py --workload verify128 --steps 2 --warmup 1 > "$OUT/r04_verify128_1gpu_$TAG.json" 2>/dev/null
echo "== ntt24"; python bench.py --workload ntt24 --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/r04_ntt24_$TAG.json" 2>/dev/null
echo "== sync variants"; python bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline --stark-batch-cols 0 > "$OUT/r04_bench_sync_single_batch_$TAG.json" 2>/dev/null
python bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline --stark-variant grouped-leaves > "$OUT/r04_bench_sync_grouped_leaves_variant_$TAG.json" 2>/dev/null
NLX_AIR_VM=1 python bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline > "$OUT/r04_bench_sync_air_interpreter_$TAG.json" 2>/dev/null
echo "== starks"; python bench.py --workload ed25519 --log-slots 10 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/r04_ed25519_2p10_slots_$TAG.json" 2>/dev/null
python bench.py --workload sha256 --log-blocks 14 --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/r04_sha256_2p14_blocks_$TAG.json" 2>/dev/null
python bench.py --workload sha512 --log-blocks 14 --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/r04_sha512_2p14_blocks_$TAG.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
echo "== rocprof sync"
rocprofv3 --kernel-trace --stats -d "$OUT/stats_sync" -o s --output-format csv -- python3 "$ROOT/bench.py" --steps 6 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
echo "== rocprof outer 2^18 single"
rocprofv3 --kernel-trace --stats -d "$OUT/stats_outer18_single" -o s --output-format csv -- python3 "$ROOT/bench.py" --workload outer --log-n 18 --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
echo "== rocprof ntt24"
rocprofv3 --kernel-trace --stats -d "$OUT/stats_ntt24" -o s --output-format csv -- python3 "$ROOT/bench.py" --workload ntt24 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd "$ROOT"
cp "$OUT/stats_sync/s_kernel_stats.csv" "$OUT/r04_bench_sync_kernel_stats_$TAG.csv"
cp "$OUT/stats_outer18_single/s_kernel_stats.csv" "$OUT/r04_bench_outer_2p18_single_stream_kernel_stats_$TAG.csv"
cp "$OUT/stats_ntt24/s_kernel_stats.csv" "$OUT/r04_ntt24_kernel_stats_$TAG.csv"
python3 tools/trace_summary.py "$OUT/stats_sync/s_kernel_trace.csv" k_hash_lde_leaves > "$OUT/r04_bench_sync_leaf_hash_by_grid_$TAG.txt"
python3 tools/trace_summary.py "$OUT/stats_ntt24/s_kernel_trace.csv" k_ntt > "$OUT/r04_ntt24_passes_by_grid_$TAG.txt"
echo "== pmc outer 2^18"
bash tools/r04_pmc_outer18.sh $TAG > "$OUT/pmc18.log" 2>&1
cp gpurun_out/r04/pmc18_$TAG/r04_pmc_traffic_outer_2p18.json gpurun_out/r04/pmc18_$TAG/r04_pmc_by_kernel_outer_2p18.txt "$OUT/" 2>/dev/null
rm -rf "$OUT"/stats_*/s_kernel_trace.csv "$OUT"/stats_*/s_domain_stats.csv
ls "$OUT"
python3 -c "
import json
d=json.load(open('$OUT/r04_bench_sync_$TAG.json')); print('sync', d['value'], d['ms_per_step'], d['ms_one_proof_at_a_time'])
for ln in (13,16,18,20):
    print('outer', ln, json.load(open('$OUT/r04_bench_outer_2p%d_single_stream_$TAG.json'%ln))['ms_per_step'])
print('outer16 x3', json.load(open('$OUT/r04_bench_outer_2p16_$TAG.json'))['value'])
print('verify128', json.load(open('$OUT/r04_verify128_1gpu_$TAG.json'))['ms_per_step'])
print('ntt24', json.load(open('$OUT/r04_ntt24_$TAG.json'))['ms_per_step'])
for v in ('single_batch','grouped_leaves_variant','air_interpreter'):
    d=json.load(open('$OUT/r04_bench_sync_%s_$TAG.json'%v)); print('sync', v, d['value'], d['ms_per_step'], d['ms_one_proof_at_a_time']['step'])
for w in ('ed25519_2p10_slots','sha256_2p14_blocks','sha512_2p14_blocks'):
    print(w, json.load(open('$OUT/r04_%s_$TAG.json'%w))['ms_per_step'])
"
