"""Build the HIP/C-ABI shared library (libnlx.so) for gfx950, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container as well as
on the MI355X box.  Objects are cached under csrc/.obj and rebuilt when a source or header is
newer.  The workload generator (libnlx_synth.so, plain C++) is built by the same entry point.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# Generator pair (include/nlx_field.h): default set 7; NLX_GL_GENERATOR_SET=2021 builds the other candidate pair
# into libnlx_gen2021.so (own object cache) so the parity suites can be run under both.
GEN_SET = os.environ.get("NLX_GL_GENERATOR_SET", "7")
if GEN_SET not in ("7", "2021"):
    raise RuntimeError("NLX_GL_GENERATOR_SET must be 7 or 2021")
_SUFFIX = "" if GEN_SET == "7" else "_gen" + GEN_SET
# kernel-tuning experiments: NLX_BUILD_VARIANT=name NLX_EXTRA_FLAGS="-DNLX_QW=4 ..." builds libnlx_name.so beside the real one
# (own object cache); near-light-client_amd/_lib.py loads it when NLX_BUILD_VARIANT is set.  Never used by tests or the bench.
# Sanitizer build of the HOST side (the device side cannot be: no GPU ASan on the pool) - tools/asan_host.sh:
#   NLX_BUILD_VARIANT=asan NLX_EXTRA_FLAGS="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer"
#   NLX_EXTRA_LDFLAGS="-fsanitize=address,undefined -shared-libsan"  ->  libnlx_asan.so
VARIANT = os.environ.get("NLX_BUILD_VARIANT", "")
if VARIANT:
    _SUFFIX += "_" + VARIANT
OBJ = os.path.join(CSRC, ".obj" + _SUFFIX)
LIB = os.path.join(HERE, "libnlx%s.so" % _SUFFIX)
# workload generation (synthetic circuits / witnesses: inputs for tests, examples and bench.py) is NOT part of the product
# library: csrc/synth.cpp -> libnlx_synth.so (host C++ only; include/nlx_synth.h)
SYNTH_SOURCES = ("synth.cpp",)
SYNTH_LIB = os.path.join(HERE, "libnlx_synth%s.so" % _SUFFIX)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -Xarch_host -mavx2: the host side of the library (transcript hashing, FRI bookkeeping) runs on the GPU node's x86-64 CPU
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Xarch_host", "-mavx2", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(HERE, "..", "include"), "-DNLX_GL_GENERATOR_SET=" + GEN_SET] + os.environ.get("NLX_EXTRA_FLAGS", "").split()


AIRGEN = os.path.join(CSRC, "airgen")


def _generate_air_kernels():
    """csrc/airgen/*.hip: straight-line kernels for the fixed AIR programs (airgen.py), written by a child interpreter (the
    generator stubs the package's ctypes layer - this must not leak into a process that goes on to load the library).  Files are
    rewritten only when their text changes, so an unchanged AIR costs no recompilation."""
    r = subprocess.run([sys.executable, os.path.join(HERE, "airgen.py")], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("airgen.py failed:\n%s\n%s" % (r.stdout, r.stderr))


# The generated AIR kernels (~5 minutes of a clean build) go into the library of record only: the other candidate generator pair's
# build (libnlx_gen2021.so, there for the parity suites) and NLX_NO_AIRGEN=1 builds run every program on the interpreter.
WITH_AIRGEN = GEN_SET == "7" and os.environ.get("NLX_NO_AIRGEN", "") != "1"
if not WITH_AIRGEN:
    FLAGS.append("-DNLX_NO_AIRGEN")


def _sources():
    top = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))
    gen = sorted(os.path.join("airgen", f) for f in os.listdir(AIRGEN) if f.endswith(".hip")) if WITH_AIRGEN and os.path.isdir(AIRGEN) else []
    return top + gen


def _headers_mtime(only=None):
    """newest header; `only`: the names a source is known to include (the generated AIR kernels include air_vm.hpp -> gl.hpp and
    the public headers, nothing else - a change to any other header must not cost their five minutes of compilation)"""
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".hpp", ".h", ".inc")) and (only is None or f in only):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


AIRGEN_HEADERS = ("air_vm.hpp", "gl.hpp", "nlx.h", "nlx_field.h")


def _compile(src, verbose):
    obj = os.path.join(OBJ, src.replace(os.sep, "_") + ".o")
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build_lib(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    if WITH_AIRGEN:
        _generate_air_kernels()
    hm, hm_gen = _headers_mtime(), _headers_mtime(AIRGEN_HEADERS)
    todo, objs, synth_objs = [], [], []
    for src in _sources():
        obj = os.path.join(OBJ, src.replace(os.sep, "_") + ".o")
        (synth_objs if src in SYNTH_SOURCES else objs).append(obj)
        sm = max(os.path.getmtime(os.path.join(CSRC, src)), hm_gen if src.startswith("airgen" + os.sep) else hm)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < sm:
            todo.append(src)
    if todo:
        todo.sort(key=lambda s_: -os.path.getsize(os.path.join(CSRC, s_)))   # the generated AIR kernels take ~a minute each: start them first
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(lambda s: _compile(s, verbose), todo))
    for lib, members in ((LIB, objs), (SYNTH_LIB, synth_objs)):
        if todo or not os.path.exists(lib):
            cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + members + os.environ.get("NLX_EXTRA_LDFLAGS", "").split()
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
