"""include/nlx.h is a plain-C header: a C caller compiles and links against libnlx.so with gcc (CPU), fails
loudly without a GPU, and proves on the GPU."""
import os
import subprocess

import pytest

from conftest import ROOT


def _build(tmp_path):
    exe = str(tmp_path / "prove_example")
    lib_dir = os.path.join(ROOT, "near-light-client_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "prove_example.c"), "-L", lib_dir, "-lnlx", "-Wl,-rpath," + lib_dir, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_c_caller_builds_and_fails_loudly_without_gpu(nlx, tmp_path):
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe, "8"], capture_output=True, text=True)
    assert r.returncode == 1 and "nlx_ctx_create" in r.stderr  # no CPU fallback


@pytest.mark.gpu
def test_c_caller_proves_on_gpu(nlx, tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "11"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("ok: 2^11 rows")
