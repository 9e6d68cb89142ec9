"""Determinism and memory stability ("identical outputs for identical inputs"): tools/soak.py proves the same
inputs over and over with every prover on one context; proof bytes must not change and device memory must not grow."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_repeated_proofs_are_identical_and_memory_is_stable():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "5"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:] + r.stdout[-500:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["iterations"] == 5 and len(out["proofs"]) >= 13 and out["in_use_bytes"] > 0
    # the provers (plonky2 with and without lookup tables, the STARKs with and without batches, a generated AIR kernel), the natural-order
    # NTT, and the BN254 NTT, MSM, PLONK quotient chain (plain and blinded) and KZG opening
    for name in ("plonky2_2p13", "plonky2_2p11_lookup_tables", "sha256_2p7_batches_generated_kernel", "ed25519_2p8", "bn254_plonk_quotient_2p10_blinded"):
        assert name in out["proofs"]
