"""CPU test (gloo, world sizes 2 and 4) of the split of ONE NTT over several ranks (near-light-client_amd/split_ntt.py;
BASELINE.json configs[4]): the exchange pattern - rank r swaps slices with r XOR (G >> (level + 1)) -, which
half of each cross-rank level a rank computes and with which twiddles, and the cyclic output distribution.  The two compute
steps are passed in as numpy code over the oracle's field helpers (the GPU kernels behind the defaults are compared with a
single-context transform in tests/test_gpu_primitives.py)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, field_generators

P = 0xFFFFFFFF00000001


def _ntt_natural(col, w):
    """plain recursive NTT of a list of Python ints (natural order in and out) with root w"""
    n = len(col)
    if n == 1:
        return list(col)
    ev, od = _ntt_natural(col[0::2], w * w % P), _ntt_natural(col[1::2], w * w % P)
    out, t, h = [0] * n, 1, n // 2
    for k in range(h):
        x = t * od[k] % P
        out[k], out[k + h] = (ev[k] + x) % P, (ev[k] - x) % P
        t = t * w % P
    return out


def _worker(rank, world, port, q, log_n, n_cols, w_n):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import nlxpkg
    nlx = nlxpkg.load()
    S = nlx.split_ntt
    n, m = 1 << log_n, (1 << log_n) // world
    world_log = world.bit_length() - 1
    rng = np.random.RandomState(5)
    host = rng.randint(0, 1 << 62, size=(n_cols, n), dtype=np.int64)
    mine = torch.from_numpy(host[:, rank * m:(rank + 1) * m].copy())
    log = []

    def level_fn(mine_, theirs_, level):
        # what nlx_ntt_split_level computes, in Python integers: the rank's half of the level's butterflies
        bit = world_log - 1 - level
        upper = (rank >> bit) & 1
        idx0 = (rank & ((1 << bit) - 1)) * m
        log.append((level, rank ^ (world >> (level + 1)), upper))
        a = mine_.numpy().view(np.uint64)
        b = theirs_.numpy().view(np.uint64)
        for c in range(n_cols):
            for i in range(m):
                x, y = int(a[c, i]), int(b[c, i])
                a[c, i] = (y - x) * pow(w_n, (idx0 + i) << level, P) % P if upper else (x + y) % P

    def local_fn(mine_):
        a = mine_.numpy().view(np.uint64)
        w_m = pow(w_n, world, P)
        for c in range(n_cols):
            a[c, :] = _ntt_natural([int(v) for v in a[c]], w_m)

    S.split_ntt(None, mine, log_n, rank, world, dist, level_fn=level_fn, local_fn=local_fn)
    got = S.gather_natural(mine, rank, world, dist)
    q.put((rank, got if rank == 0 else None, log, S.residue_of_rank(rank, world)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_split_exchange_pattern_and_output_distribution(world):
    log_n, n_cols = 6, 2
    w_n = pow(field_generators()[2], 1 << (32 - log_n), P)          # the 2^log_n-th root of unity of the build's generator pair
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, log_n, n_cols, w_n)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    rng = np.random.RandomState(5)
    host = rng.randint(0, 1 << 62, size=(n_cols, 1 << log_n), dtype=np.int64)
    want = np.array([_ntt_natural([int(v) % P for v in col], w_n) for col in host.view(np.uint64)], dtype=np.uint64)
    assert np.array_equal(res[0][1], want)
    world_log = world.bit_length() - 1
    for rank, _, log, residue in res:
        assert [lv for lv, _, _ in log] == list(range(world_log))
        for level, partner, upper in log:
            assert partner == rank ^ (world >> (level + 1)) and upper == (rank >> (world_log - 1 - level)) & 1
        assert residue == int(format(rank, "0%db" % world_log)[::-1], 2)
    assert sorted(r[3] for r in res) == list(range(world))        # every residue class mod G is held by exactly one rank
