#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out/r03"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_stark.py tests/test_sha256_air.py tests/test_sha512_air.py tests/test_ed25519_air.py tests/test_gpu_headline_parity.py -m gpu -x -q > gpurun_out/r03/tests_opendigest.txt 2>&1
rc=$?
echo "rc=$rc" >> gpurun_out/r03/tests_opendigest.txt
tail -5 gpurun_out/r03/tests_opendigest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 bench.py --no-extra > gpurun_out/r03/sync_opendigest.json 2> gpurun_out/r03/sync_opendigest.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03/sync_opendigest.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"])
print(d["ms_one_proof_at_a_time"])
print(d.get("parity_checked", {}).get("all_bytes_equal"))
PY
