"""The wrap's PLONK quotient chain over the GPUs of a node: the SCHEDULE of DESIGN.md §7 (row f.4) - who transforms which column,
what the all-to-all moves, who evaluates which points, where the quotient's inverse transform runs.  No arithmetic lives here:
the three compute steps are callables.  On GPUs they are the library's kernels (the transforms: nlx_bn254_ntt; a per-slice
pointwise kernel is not split out of nlx_bn254_plonk_quotient yet, so nothing in the product calls this module with device
steps today); the CPU rehearsal (tests/tools/plonk_split_rehearsal.py, gloo) passes the big-integer model's and checks the
result against the unsplit chain.

  (ii)  transforms by COLUMN: column j (in the caller's fixed order) belongs to rank j mod N, which turns its values on H into
        the 4n evaluations on the coset (`transform`);
  (iii) the pointwise pass by POINTS: rank r evaluates points [r 4n/N, (r + 1) 4n/N) and needs every column there, plus
        `halo` more points of the columns the pass reads at x w (z: 4 points on the 4n-point coset) - one all-to-all of slices;
  (iv)  t's evaluations go back to rank 0, which runs the one inverse transform (`finish`).

Values travel as int64 tensors of shape (points, limbs) - four 64-bit limbs of an fr.Element; the schedule never looks inside.
gnark's prover computes the same chain on one device (backend/plonk/bn254/prove.go computeQuotient, named through the platform's
wrapper entry in /root/reference/succinct.json:7-8,17-18; BASELINE.json configs[4])."""
from . import split_ntt as _split_ntt   # the pairwise exchange primitive (RCCL: batched isend / irecv, gloo: ordered send / recv)


def column_owner(j, world):
    return j % world


def point_slice(rank, world, n_points):
    m = n_points // world
    return rank * m, (rank + 1) * m


def split_quotient_chain(names, local_values, n_points, rank, world, dist, transform, pointwise, finish, halo_names=("z",), halo=4, limbs=4,
                         device=None):
    """names: every column, in the same order on every rank; local_values[name]: this rank's columns' values on H (only the
    columns it owns are read).  transform(name, values) -> (n_points, limbs) int64 tensor on the rank's device;
    pointwise(columns: name -> (slice + halo or slice, limbs) tensor, first_point, count) -> (count, limbs) tensor;
    finish(t_evals (n_points, limbs)) -> the result, on rank 0 (None elsewhere)."""
    import torch
    if world & (world - 1) or n_points % world:
        raise ValueError("the number of ranks must be a power of two dividing the number of points")
    m = n_points // world
    # (ii) my columns, whole
    mine = {nm: transform(nm, local_values[nm]) for j, nm in enumerate(names) if column_owner(j, world) == rank}
    if device is None:
        device = next(iter(mine.values())).device if mine else torch.device("cpu")

    def rows_for(nm, r):   # what rank r needs of column nm: its slice, plus the halo (wrapping) where the pass reads ahead
        lo, hi = point_slice(r, world, n_points)
        idx = torch.arange(lo, hi + (halo if nm in halo_names else 0), device=device) % n_points
        return idx

    # (iii) the all-to-all as N - 1 pairwise exchanges (rank ^ s): one link per rank and step on xGMI
    cols = {}
    for nm, ev in mine.items():
        cols[nm] = ev[rows_for(nm, rank)].contiguous()
    for s in range(1, world):
        partner = rank ^ s
        out_names = [nm for j, nm in enumerate(names) if column_owner(j, world) == rank]
        in_names = [nm for j, nm in enumerate(names) if column_owner(j, world) == partner]
        send = torch.cat([mine[nm][rows_for(nm, partner)] for nm in out_names]) if out_names else torch.zeros((0, limbs), dtype=torch.int64, device=device)
        n_in = sum(m + (halo if nm in halo_names else 0) for nm in in_names)
        n_out = send.shape[0]
        # equal-length buffers keep the exchange one primitive: pad to the longer side
        size = max(n_in, n_out, 1)
        buf = torch.zeros((size, limbs), dtype=torch.int64, device=device)
        buf[:n_out] = send
        got = _split_ntt._exchange(dist, buf, partner)
        at = 0
        for nm in in_names:
            k = m + (halo if nm in halo_names else 0)
            cols[nm] = got[at:at + k].contiguous()
            at += k
    # my points
    lo, _ = point_slice(rank, world, n_points)
    t_slice = pointwise(cols, lo, m)
    # (iv) t back to rank 0
    if world == 1:
        return finish(t_slice)
    gloo = dist.get_backend() == "gloo"
    if rank == 0:
        parts = [t_slice]
        for r in range(1, world):
            buf = torch.empty((m, limbs), dtype=torch.int64, device=torch.device("cpu") if gloo else device)
            dist.recv(buf, r)
            parts.append(buf.to(device))
        return finish(torch.cat(parts))
    dist.send(t_slice.cpu() if gloo else t_slice, 0)
    return None
