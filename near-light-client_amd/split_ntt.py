"""ONE NTT split over the GPUs of a node (BASELINE.json configs[4]; SURVEY.md §8e: the only intra-transform sharding in
scope - "the config-5 2^24 NTT stress exercises it").

Layout: every column's 2^log_n points are dealt out in contiguous slices, rank r holding [r m, (r + 1) m), m = 2^log_n / G.
A decimation in frequency's first log2(G) levels pair element j with j + n / 2^(level + 1), which lives on rank
r XOR (G >> (level + 1)): per level each rank swaps its slice with that partner (send / recv over RCCL - xGMI is
point-to-point, so a pairwise exchange uses exactly one link per rank) and computes its half of the butterflies
(`nlx_ntt_split_level`).  What remains is an independent m-point transform per rank (`nlx_ntt_batch`): no all-to-all, no
transposes, log2(G) exchanges of the slice.  Result: rank r holds X[k] for k = bitrev_G(r) (mod G) at local index k div G
- the cyclic distribution a following pointwise stage can use as it is; `gather_natural` reassembles the natural order on
every rank (tests, small sizes).

Communication volume per rank: log2(G) x slice bytes - at 16 columns x 2^24 points on 8 GPUs 3 x 256 MB, i.e. more time on
the links (7 x ~150 GB/s per GPU, one link per exchange) than the 1.4 ms of arithmetic left per rank: the split is
link-bound by construction, which is why the reference's shape never asks for it outside this stress configuration.
"""
import numpy as np

from ._lib import dll


def _exchange(dist, mine, partner):
    """swap device tensors with `partner`; gloo (rehearsal on one GPU) moves host copies"""
    import torch
    gloo = dist.get_backend() == "gloo"
    send = mine.cpu() if gloo else mine
    recv = torch.empty_like(send)
    if gloo:   # plain send / recv, the lower rank sending first (gloo has no batched point-to-point)
        if dist.get_rank() < partner:
            dist.send(send, partner)
            dist.recv(recv, partner)
        else:
            dist.recv(recv, partner)
            dist.send(send, partner)
    else:
        ops = [dist.P2POp(dist.isend, send, partner), dist.P2POp(dist.irecv, recv, partner)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return recv.to(mine.device) if gloo else recv


def split_ntt(ctx, mine, log_n, rank, world, dist, level_fn=None, local_fn=None):
    """mine: (n_cols, 2^log_n / world) int64 device tensor, this rank's slice of every column (overwritten).  Forward NTT of
    the whole columns; returns `mine`: X[k] for k = bitrev_G(rank) (mod G), local index k div G.
    level_fn(mine, theirs, level) / local_fn(mine): the two compute steps, by default the library's kernels on `ctx`
    (nlx_ntt_split_level, nlx_ntt_batch); the CPU tests of the exchange pattern pass their own."""
    world_log = world.bit_length() - 1
    if world != 1 << world_log:
        raise ValueError("the number of ranks must be a power of two")
    n_cols, m = mine.shape
    if m << world_log != 1 << log_n or not mine.is_contiguous():
        raise ValueError("slice shape")
    if level_fn is None:
        def level_fn(mine_, theirs_, level_):
            import torch
            torch.cuda.current_stream(mine_.device).synchronize()
            ctx.check(dll.nlx_ntt_split_level(ctx.handle, mine_.data_ptr(), theirs_.data_ptr(), n_cols, log_n, world_log, rank, level_))
    if local_fn is None:
        def local_fn(mine_):
            ctx.check(dll.nlx_ntt_batch(ctx.handle, mine_.data_ptr(), n_cols, log_n - world_log, 0, 1))
    for level in range(world_log):
        partner = rank ^ (world >> (level + 1))
        theirs = _exchange(dist, mine, partner)
        level_fn(mine, theirs, level)
    local_fn(mine)
    return mine


def residue_of_rank(rank, world):
    """the residue k mod G of the outputs rank `rank` holds"""
    bits = world.bit_length() - 1
    return int(format(rank, "0%db" % bits)[::-1], 2) if bits else 0


def gather_natural(mine, rank, world, dist):
    """every rank's outputs reassembled in natural order, (n_cols, 2^log_n) uint64 on the host (tests, small sizes)"""
    import torch
    gloo = dist.get_backend() == "gloo"
    local = mine.cpu() if gloo else mine
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local)
    n_cols, m = mine.shape
    out = np.zeros((n_cols, m * world), dtype=np.uint64)
    for r, p in enumerate(parts):
        out[:, residue_of_rank(r, world)::world] = p.cpu().numpy().view(np.uint64)
    return out
