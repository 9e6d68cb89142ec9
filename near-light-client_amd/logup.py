"""Range-check lookups for multi-round AIRs: the log-derivative argument (LogUp) with the challenge in the quadratic
extension (starkyx commits its traces in rounds precisely so that lookup / bus accumulators
can depend on verifier challenges; curta's own lookup constraints are not in the reference, Cargo.lock:6515).

The cells of `cols` (round-0 columns) are shown to lie in {0, .., 2^bits - 1}: with alpha = challenge(0) + challenge(1) X
drawn after round 0 is committed,

    sum over rows and lookups of 1 / (alpha + v)  =  sum over rows of m / (alpha + t),        t(i) = i mod 2^bits,

where m (a round-0 column) holds the multiplicity of value v in row v (with the table spread over several columns:
one m and one table column per part).  Round 1 commits one helper
h = 1/(alpha + v1) + 1/(alpha + v2) per pair of lookups (a single-lookup helper if the count is odd), g = m/(alpha + t)
and the running sum phi; all constraints have degree <= 3 and hold on every row including the wrap (the running sum
telescopes to zero around the cycle, so no boundary constraint is needed).  The table is a periodic column - the
verifier evaluates it itself, nothing is committed for it.

Extension elements are two base columns (a, b) = a + b X, X^2 = 7 (plonky2's QuadraticExtension<GoldilocksField>).
"""
import numpy as np

from ._lib import dll

W = 7  # X^2 = W


def round_cols(n_lookups, table_cols=1):
    """number of round-1 columns: helpers, one g per table column, phi (two base columns each)"""
    return 2 * ((n_lookups + 1) // 2) + 2 * table_cols + 2


class RangeCheck:
    """Adds the lookup constraints to `air` (a two-round Air whose round 0 draws at least two challenges).

    cols: the looked-up round-0 columns; mult_col: the (first) round-0 multiplicity column; table_cols: the table may be
    spread over that many periodic columns (a power of two), so a trace shorter than the table can carry it - then
    mult_col .. mult_col + table_cols - 1 are the multiplicity columns; first: index of the first round-1
    column used (round_cols(len(cols)) consecutive columns); challenge: index of alpha's first base challenge;
    fused=False writes the helper constraints out with the DSL instead of NLX_AIR_EMIT_LOGUP (same constraint values,
    same proof, thirty times the program words: kept for the tests that compare the two)."""

    def __init__(self, air, cols, bits, mult_col, first, challenge=0, fused=True, table_cols=1):
        self.cols, self.bits, self.mult_col, self.first = [int(c) for c in cols], bits, mult_col, first
        self.n_helpers = (len(self.cols) + 1) // 2
        self.table_cols = table_cols
        assert table_cols >= 1 and table_cols & (table_cols - 1) == 0 and table_cols <= (1 << bits)
        self.n_round_cols = round_cols(len(self.cols), table_cols)
        L, N = air.local, air.next  # noqa: N806
        a0, a1 = air.challenge(challenge), air.challenge(challenge + 1)
        period = (1 << bits) // table_cols          # the table spread over table_cols columns: column c holds c P + (i mod P)
        tables = [air.periodic(range(c * period, (c + 1) * period)) for c in range(table_cols)]
        sum0, sum1 = None, None
        for j in range(self.n_helpers):
            h0, h1 = L(first + 2 * j), L(first + 2 * j + 1)
            v2 = self.cols[2 * j + 1] if 2 * j + 1 < len(self.cols) else None
            if fused:
                # h (alpha + v1)(alpha + v2) = (alpha + v1) + (alpha + v2): both coefficients as one VM instruction
                air.constraint_logup(self.cols[2 * j], v2, first + 2 * j, challenge)
            elif v2 is not None:
                v1, v2 = L(self.cols[2 * j]), L(v2)
                u0 = (a0 + v1) * (a0 + v2) + a1 * a1 * W
                u1 = a1 * (a0 * 2 + v1 + v2)
                air.constraint(h0 * u0 + h1 * u1 * W - (a0 * 2 + v1 + v2))
                air.constraint(h0 * u1 + h1 * u0 - a1 * 2)
            else:
                v1 = L(self.cols[2 * j])
                air.constraint(h0 * (a0 + v1) + h1 * a1 * W - 1)
                air.constraint(h0 * a1 + h1 * (a0 + v1))
            sum0 = h0 if sum0 is None else sum0 + h0
            sum1 = h1 if sum1 is None else sum1 + h1
        g = first + 2 * self.n_helpers
        phi = g + 2 * table_cols
        for c, t in enumerate(tables):
            g0, g1 = L(g + 2 * c), L(g + 2 * c + 1)
            air.constraint(g0 * (a0 + t) + g1 * a1 * W - L(mult_col + c))
            air.constraint(g0 * a1 + g1 * (a0 + t))
            sum0, sum1 = sum0 - g0, sum1 - g1
        air.constraint(N(phi) - L(phi) - sum0)
        air.constraint(N(phi + 1) - L(phi + 1) - sum1)

    # ---- witness side (GPU): both take the round-0 trace as a device tensor or host array [n_cols0, n] ----
    def multiplicities(self, ctx, trace):
        """Fills column mult_col of `trace` in place (nlx_logup_multiplicities); raises if a cell is outside the table."""
        n = trace.shape[1]
        cols = np.array(self.cols, dtype=np.uint32)
        base = trace.data_ptr() if hasattr(trace, "data_ptr") else trace.ctypes.data
        ctx.check(dll.nlx_logup_multiplicities(ctx.handle, base, trace.shape[0], n.bit_length() - 1, cols.ctypes.data, cols.size,
                                               self.bits, self.table_cols, self.mult_col))

    def round1(self, ctx, trace, alpha, out):
        """Writes the round-1 columns for challenge alpha = (a0, a1) into `out` ([n_round_cols, n], device or host)."""
        n = trace.shape[1]
        cols = np.array(self.cols, dtype=np.uint32)
        al = np.array([int(alpha[0]), int(alpha[1])], dtype=np.uint64)
        base = trace.data_ptr() if hasattr(trace, "data_ptr") else trace.ctypes.data
        optr = out.data_ptr() if hasattr(out, "data_ptr") else out.ctypes.data
        ctx.check(dll.nlx_logup_round(ctx.handle, base, trace.shape[0], n.bit_length() - 1, cols.ctypes.data, cols.size, self.bits,
                                      self.table_cols, self.mult_col, al.ctypes.data, optr))
        return out
