"""CPU: the generated tables of the device permutation's fused partial rounds (csrc/poseidon_blocks.inc).

The generator (tools/gen_poseidon_blocks.py) carries a Python model of the device schedule - integer byte-plane arithmetic on the
very digit / seed / constant tables it emits - and compares it with the naive permutation and upstream's known answers before it
writes anything.  Here: the committed file is what the generator produces, and the model holds on further states."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_poseidon_blocks as gpb  # noqa: E402


def test_committed_tables_are_the_generators(tmp_path):
    tb, blocks, layer, n, worst = gpb.build()
    out = tmp_path / "poseidon_blocks.inc"
    gpb.emit(tb, blocks, layer, str(out))
    committed = open(os.path.join(ROOT, "near-light-client_amd", "csrc", "poseidon_blocks.inc")).read()
    assert out.read_text() == committed
    assert n >= 44 and worst <= tb["dmax_main"] < 1 << 20


def test_block_schedule_equals_naive_permutation(golden):
    tb, blocks, layer, _, _ = gpb.build()
    rnd = random.Random(7)
    P = gpb.P
    E = 0xFFFFFFFF
    edge = [0, 1, E, E + 1, P - 1, P - 2, 1 << 63, P - E, 2 * E, (1 << 32) + 1, P >> 1, 7]
    states = [list(k["in"]) for k in golden["poseidon_kat"]] + [edge, edge[::-1]] + [[rnd.randrange(P) for _ in range(12)] for _ in range(60)]
    for st in states:
        got, dmax = gpb.model(tb, blocks, layer, st, rnd)
        assert got == gpb.naive(st)
        assert dmax <= tb["dmax_main"]
    for kat in golden["poseidon_kat"]:
        assert gpb.model(tb, blocks, layer, list(kat["in"]), rnd)[0] == list(kat["out"])


def test_digits_and_seeds():
    tb = gpb.tables()
    K = gpb.K
    for r in range(gpb.ROWS):
        for j in range(gpb.SLOTS):
            ds = [tb["digits"][p][r][j] for p in range(K)]
            assert all(-128 <= d <= 127 for d in ds)
            assert sum(d << (8 * p) for p, d in enumerate(ds)) == tb["A"][r][j]
    # slots beyond the state and the block's scalars, and the idle rows, are zero: their B bytes / outputs are undefined
    for p in range(K):
        for r in range(gpb.ROWS):
            assert all(tb["digits"][p][r][j] == 0 for j in range(gpb.T + K - 1, gpb.SLOTS))
        for r in range(gpb.T, gpb.ROWS):
            assert all(d == 0 for d in tb["digits"][p][r])
    gpb.bounds(tb)
