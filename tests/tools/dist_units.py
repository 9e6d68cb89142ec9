"""The two exchange primitives of the multi-GPU paths, alone, under the backend the driver's N > 1 run uses:
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/tools/dist_units.py nccl
(`nccl` = RCCL, one rank per GPU, every tensor on the rank's own device; `gloo` = the same code on host tensors, which is how
it runs in the CPU suite).  Checks mapreduce.all_gather_blobs - ragged blob sizes, a job count that is not a multiple of the
rank count, a rank that owns nothing - and split_ntt._exchange - a pairwise swap of device tensors.  Prints one line per rank.
nearx/src/verify.rs:69-90 (the map-reduce whose levels all_gather_blobs joins), BASELINE.json configs[3] / [4]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import importlib.util
    import types
    import torch
    import torch.distributed as dist
    backend = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    # the two modules have no GPU dependency of their own: load them without the package's libnlx.so import
    pkg = types.ModuleType("nlx_units")
    pkg.__path__ = [os.path.join(ROOT, "near-light-client_amd")]
    sys.modules["nlx_units"] = pkg

    def load(name, stub_lib=False):
        if stub_lib:
            sys.modules["nlx_units._lib"] = types.SimpleNamespace(dll=None)
        spec = importlib.util.spec_from_file_location("nlx_units." + name, os.path.join(ROOT, "near-light-client_amd", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["nlx_units." + name] = mod
        spec.loader.exec_module(mod)
        return mod
    load("nearx_io")
    load("succinct_io")
    mr = load("mapreduce")
    sn = load("split_ntt", stub_lib=True)
    # ---- all_gather_blobs: 2 world + 1 jobs, job j's proof has 1000 + 37 j bytes, outputs of 0 .. 2 bytes ----
    for n_jobs in (2 * world + 1, 1, world):
        local_blobs = {j: mr.Blob(bytes([j % 251]) * (j % 3), bytes([(7 * j + 1) % 256]) * (1000 + 37 * j))
                       for j in range(n_jobs) if mr.owner(j, world) == rank}
        got = mr.all_gather_blobs(local_blobs, n_jobs, rank, world, dist, device)
        assert len(got) == n_jobs
        for j, b in enumerate(got):
            assert b.output == bytes([j % 251]) * (j % 3) and b.proof == bytes([(7 * j + 1) % 256]) * (1000 + 37 * j), (n_jobs, j)
    # ---- _exchange: every rank swaps a (3, 4096) int64 tensor with rank ^ 1 ----
    if world % 2 == 0:
        mine = (torch.arange(3 * 4096, dtype=torch.int64).reshape(3, 4096) + (rank << 40)).to(device)
        theirs = sn._exchange(dist, mine, rank ^ 1)
        want = torch.arange(3 * 4096, dtype=torch.int64).reshape(3, 4096) + ((rank ^ 1) << 40)
        assert theirs.device == mine.device and torch.equal(theirs.cpu(), want)
    # ---- MAX of a host scalar, the bench's timing reduction ----
    tt = torch.tensor([float(rank)], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert float(tt.item()) == world - 1
    if backend == "nccl":
        torch.cuda.synchronize()
    dist.barrier()
    print("dist_units ok: rank %d of %d over %s" % (rank, world, backend), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
