mkdir -p gpurun_out/r02
for ln in 16 20 22; do timeout -k 10 300 python bench.py --workload msm24 --ntt-log-n $ln --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02/msm_$ln.json 2> gpurun_out/r02/msm_$ln.err || exit 1; python -c "
import json; d=json.load(open('gpurun_out/r02/msm_$ln.json')); print($ln, d['ms_per_step'], d['config']['device_ms_rank0'])"; done
timeout -k 10 500 python bench.py --workload msm24 --steps 3 --warmup 1 > gpurun_out/r02/msm_24.json 2> gpurun_out/r02/msm_24.err && python -c "
import json; d=json.load(open('gpurun_out/r02/msm_24.json')); print(24, d['ms_per_step'], d['config']['device_ms_rank0'], d['cpu_baseline'])"
