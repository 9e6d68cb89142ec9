"""One NTT split over the ranks of this launch (torch.distributed.run, all ranks on GPU 0 over gloo: a rehearsal of the
RCCL path on a one-GPU box), compared on rank 0 with the same transform done by one context:
    python -m torch.distributed.run --nproc-per-node 4 ... tests/tools/split_ntt_ranks.py <log_n> <n_cols>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nlxpkg  # noqa: E402


def main():
    log_n, n_cols = int(sys.argv[1]), int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    nlx = nlxpkg.load()
    ctx = nlx.Context(0)
    n, m = 1 << log_n, (1 << log_n) // world
    g = torch.Generator(device="cpu").manual_seed(77)
    host = torch.randint(0, 2 ** 62, (n_cols, n), generator=g, dtype=torch.int64)
    host[0, 0], host[0, n - 1] = -1, 0                       # 2^64 - 1 is not canonical: keep the edge value below p instead
    host[0, 0] = 0x7FFFFFFF00000000
    mine = host[:, rank * m:(rank + 1) * m].contiguous().to("cuda:0")
    S = nlx.split_ntt
    S.split_ntt(ctx, mine, log_n, rank, world, dist)
    got = S.gather_natural(mine, rank, world, dist)
    ok = True
    if rank == 0:
        whole = host.to("cuda:0")
        ctx.check(nlx.lib.dll.nlx_ntt_batch(ctx.handle, whole.data_ptr(), n_cols, log_n, 0, 1))
        ok = np.array_equal(got, whole.cpu().numpy().view(np.uint64))
        print("split over %d ranks equals one transform: %s" % (world, ok))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
