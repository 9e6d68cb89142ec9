"""Where does a pipelined Sync step's time go?  Times (a) the three STARKs concurrently, alone; (b) the outer proof alone;
(c) STARK trio and outer proof at the same time (no dependency); (d) bench.py's pipeline.  Run on the GPU box."""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nlxpkg  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=18)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--no-priority", action="store_true")
ap.add_argument("--stark-cus", type=int, default=0, help="reserve every k-th .. CUs: number of CUs (of 256) given to the STARK contexts")
ap.add_argument("--pattern", default="spread", choices=["spread", "block"])
a = ap.parse_args()
args = argparse.Namespace(log_n=a.log_n, gate_mix="nearx")
nlx = nlxpkg.load()
st = bench.sync_step_setup(args, nlx, torch, 0, 0)
if not a.no_priority and not a.stark_cus:
    for c in st["ctxs"][:3]:
        c.set_priority(True)
if a.stark_cus:
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count   # 256 on MI355X
    if a.pattern == "spread":
        step = n_cu // a.stark_cus
        stark = [i for i in range(n_cu) if i % step == 0][: a.stark_cus]
    else:
        stark = list(range(a.stark_cus))
    rest = [i for i in range(n_cu) if i not in set(stark)]
    for c in st["ctxs"][:3]:
        c.set_cu_mask(stark)
    st["ctxs"][3].set_cu_mask(rest)
    print("STARK contexts on %d CUs (%s), outer on %d" % (len(stark), a.pattern, len(rest)))
jobs = [lambda: st["p256"].prove(st["sha_msgs"]), lambda: st["p512"].prove(st["sig_msgs"]), lambda: st["ped"].prove(st["slot_words"])]
outer = lambda: st["cd"].prove_into(st["wires"], st["pis"].ctypes.data)
pool = ThreadPoolExecutor(8)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def trio():
    for f in [pool.submit(j) for j in jobs]:
        f.result()


def both():
    fs = [pool.submit(j) for j in jobs] + [pool.submit(outer)]
    for f in fs:
        f.result()


for i, j in enumerate(jobs):
    print("stark %d alone          %.2f ms" % (i, timeit(j, a.reps)))
print("trio concurrent        %.2f ms" % timeit(trio, a.reps))
print("outer alone            %.2f ms" % timeit(outer, a.reps))
print("trio + outer together  %.2f ms" % timeit(both, a.reps))
# (a "two outers together" line was dropped: one nlx_circuit is not re-entrant, the line had submitted a single outer proof)
