"""Host-side mirror of the starky STARK interface over the C ABI.

nearx's Ed25519 / SHA-256 gadgets are proved by curta/starkyx STARKs inside plonky2x
(nearx/src/builder.rs: `curta_eddsa_verify`, `curta_sha256`; Cargo.lock:6515 `starkyx`, un-vendored).
Their prover is the starky flow (`starky::prover::prove`, `StarkConfig::standard_fast_config`): commit the
trace, draw alphas, evaluate the AIR over a coset, commit the quotient, open at zeta and g*zeta, FRI.

The AIR is *data*: `Air` records constraints written with ordinary Python operators against
`local(i)`, `next(i)`, `public(i)` and compiles them into the register program the HIP quotient kernel
interprets (include/nlx.h NLX_AIR_*), mirroring starky's
`Stark::eval_packed_generic` + `ConstraintConsumer::{constraint, constraint_transition,
constraint_first_row, constraint_last_row}`.
"""
import ctypes

import numpy as np

from ._lib import dll, synth_dll, ptr, NlxError

P = 0xFFFFFFFF00000001
(AIR_LOCAL, AIR_NEXT, AIR_PUBLIC, AIR_CONST, AIR_ADD, AIR_SUB, AIR_MUL, AIR_EMIT_TRANSITION, AIR_EMIT_FIRST,
 AIR_EMIT_LAST, AIR_EMIT, AIR_PERIODIC, AIR_PACK_LOCAL, AIR_PACK_NEXT, AIR_EMIT_BOOL, AIR_LOADV, AIR_XOR3, AIR_CH,
 AIR_MAJ, AIR_SEGMENT, AIR_EMIT_LOGUP, AIR_MAC) = range(22)
_AIR_BINARY = (AIR_ADD, AIR_SUB, AIR_MUL)
_AIR_TERNARY = (AIR_XOR3, AIR_CH, AIR_MAJ, AIR_MAC)
_AIR_FUSED_EMITS = (AIR_EMIT_BOOL, AIR_EMIT_LOGUP)   # constraints that are one instruction over columns, no expression
AIR_NUM_REGS = 64
AIR_MAX_RESIDENT_LEAVES = 12   # loads kept in registers (LRU) before they are re-loaded
AIR_LOAD_BATCH = 8             # loads issued together (NLX_AIR_LOADV): memory-level parallelism of the VM
AIR_SEGMENT_NODES = 400        # arithmetic nodes per program segment (NLX_AIR_SEGMENT): the GPU runs segments in parallel
AIR_MAX_SEGMENTS = 256


class StarkDesc(ctypes.Structure):
    """nlx_stark_desc (include/nlx.h)."""
    _fields_ = [(k, ctypes.c_uint32) for k in (
        "degree_bits", "n_cols", "num_challenges", "rate_bits", "cap_height", "quotient_degree_factor",
        "fri_pow_bits", "fri_num_queries", "fri_arity_bits", "fri_final_poly_bits", "num_public_inputs",
        "n_words")] + [("program", ctypes.POINTER(ctypes.c_uint64)), ("n_periodic", ctypes.c_uint32),
                       ("period_bits", ctypes.c_uint32), ("periodic", ctypes.POINTER(ctypes.c_uint64)),
                       ("n_rounds", ctypes.c_uint32), ("round_cols", ctypes.c_uint32 * 3),
                       ("round_challenges", ctypes.c_uint32 * 3), ("leaf_group_cols", ctypes.c_uint32),
                       ("round_values", ctypes.c_uint32 * 3), ("openings_group", ctypes.c_uint32), ("batch_cols", ctypes.c_uint32)]


# the opt-in "grouped-leaves" variant (StarkConfig.grouped()) switches grouped leaves on up to this many LDE rows (2^k): at 2^16
# rows one leaf per lane is one wave per SIMD, which hashes at 1.6 G permutations/s where two or more waves reach 2.0
# (profiles/r03_poseidon_occupancy_v2.txt)
LEAF_GROUP_MAX_LOG_ROWS = 16
AUTO = "auto"


class StarkConfig:
    """starky::config::StarkConfig::standard_fast_config().

    The DEFAULT is the reference's protocol: a Merkle leaf is hash_or_noop(row) over the whole row (leaf_group_cols = 0,
    SURVEY 8a row a6) and the challenger observes every opened value (openings_group = 0, starky's observe_openings) - what
    plonky2x's in-circuit STARK verifier re-hashes (/root/reference/nearx/src/builder.rs:152,220,316).  Round 3's protocol
    variant (grouped leaves + an openings digest on wide, short traces) is opt-in: StarkConfig.grouped(), or explicit values;
    bench.py reports it as "stark_variant": "grouped-leaves" and never as the headline."""

    def __init__(self, **kw):
        self.num_challenges = 2
        self.rate_bits = 1
        self.cap_height = 4
        self.fri_pow_bits = 16
        self.fri_num_queries = 84
        self.fri_arity_bits = 4
        self.fri_final_poly_bits = 5
        # Merkle leaves over runs of this many columns (nlx_stark_desc.leaf_group_cols); 0 = whole-row leaves (starky);
        # AUTO = by shape, see leaf_group_for
        self.leaf_group_cols = 0
        # the transcript observes a digest of the openings (runs of this many values) instead of every value
        # (nlx_stark_desc.openings_group); 0 = starky's transcript; AUTO = 64 when the trace has more than 256 columns, else 0
        self.openings_group = 0
        # a commitment round of more than this many columns is committed as several PolynomialBatches of at most this many
        # columns - plain plonky2 batches, each with whole-row hash_or_noop leaves, its own cap and FRI oracle
        # (nlx_stark_desc.batch_cols); 0 = one batch per round
        self.batch_cols = 0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown config field %s" % k)
            setattr(self, k, v)

    @classmethod
    def grouped(cls, **kw):
        """round 3's variant: leaf_group_cols and openings_group chosen by the trace's shape"""
        return cls(leaf_group_cols=AUTO, openings_group=AUTO, **kw)

    @property
    def variant(self):
        return "starky" if self.leaf_group_cols == 0 and self.openings_group == 0 else "grouped-leaves"

    def leaf_group_for(self, degree_bits, widest_commitment):
        """The statement's leaf_group_cols.  AUTO: whole-row leaves (0, starky's tree) unless the trace is WIDE AND SHORT: a
        commitment of more than 256 columns on at most 2^16 LDE rows has no more leaves than the GPU has lanes and hundreds of
        sequential permutations per leaf (the Sync step's SHA-512 trace: 4 745 columns x 2^10 rows = 594 permutations on each
        of 1 024 lanes); runs of 128 columns turn that into 16 permutations on each of 38 x 1 024 lanes.  The value is part of
        the statement (the verifier reads leaf_group_cols from the descriptor, and it is in the AIR digest)."""
        if self.leaf_group_cols != AUTO:
            return int(self.leaf_group_cols)
        return 128 if widest_commitment > 256 and degree_bits + self.rate_bits <= LEAF_GROUP_MAX_LOG_ROWS else 0

    def openings_group_for(self, n_cols):
        if self.openings_group != AUTO:
            return int(self.openings_group)
        return 64 if n_cols > 256 else 0


def _pow2_factor(e):
    """(other, s) if e is MUL(other, 2^s) with 1 <= s < 62, else None."""
    if e.op != AIR_MUL:
        return None
    for c, o in ((e.a, e.b), (e.b, e.a)):
        if c.op == AIR_CONST and c.a > 1 and c.a & (c.a - 1) == 0 and c.a.bit_length() - 1 < 62:
            return o, c.a.bit_length() - 1
    return None


class _Expr:
    """Node of the constraint DAG.  degree = polynomial degree in the trace columns.  ADD / SUB nodes carry
    a shift: a +- b * 2^sh (multiplications by powers of two are folded into the neighbouring sum)."""
    __slots__ = ("air", "op", "a", "b", "c", "degree", "uses", "reg", "sh")

    def __init__(self, air, op, a=None, b=None, degree=0, sh=0, c=None):
        self.air, self.op, self.a, self.b, self.c, self.degree, self.sh = air, op, a, b, c, degree, sh
        self.uses = 0
        self.reg = None

    def operands(self):
        if self.op in _AIR_BINARY:
            return (self.a, self.b)
        if self.op in _AIR_TERNARY:
            return (self.a, self.b, self.c)
        return ()

    def _lift(self, o):
        return o if isinstance(o, _Expr) else self.air.const(o)

    def __add__(self, o):
        o = self._lift(o)
        deg = max(self.degree, o.degree)
        f = _pow2_factor(o)
        if f:
            return _Expr(self.air, AIR_ADD, self, f[0], deg, f[1])
        f = _pow2_factor(self)
        if f:
            return _Expr(self.air, AIR_ADD, o, f[0], deg, f[1])
        # sum + product: one multiply-add instruction (r[dst] = r[c] + r[a] * r[b])
        if o.op == AIR_MUL:
            return _Expr(self.air, AIR_MAC, o.a, o.b, deg, c=self)
        if self.op == AIR_MUL:
            return _Expr(self.air, AIR_MAC, self.a, self.b, deg, c=o)
        return _Expr(self.air, AIR_ADD, self, o, deg)

    __radd__ = __add__

    def __sub__(self, o):
        o = self._lift(o)
        deg = max(self.degree, o.degree)
        f = _pow2_factor(o)
        if f:
            return _Expr(self.air, AIR_SUB, self, f[0], deg, f[1])
        return _Expr(self.air, AIR_SUB, self, o, deg)

    def __rsub__(self, o):
        return self._lift(o) - self

    def __mul__(self, o):
        o = self._lift(o)
        return _Expr(self.air, AIR_MUL, self, o, self.degree + o.degree)

    __rmul__ = __mul__


STEP_TAG_LEN = 4   # field elements of a step tag: extra public inputs that no constraint reads but every transcript absorbs


def step_tag(data):
    """Four field elements naming a job (e.g. one Sync step: SHA-256 of its public input and output bytes, 56 bits per
    element).  The AIRs of one job declare STEP_TAG_LEN extra public inputs and all take the same tag, so the transcripts of
    its proofs open with the same values and a proof made for one step cannot be presented with another's."""
    import hashlib
    h = hashlib.sha256(bytes(data)).digest()
    return [int.from_bytes(h[7 * i:7 * i + 7], "little") for i in range(STEP_TAG_LEN)]


class Air:
    """An AIR over `n_cols` trace columns and `num_public_inputs` public inputs."""

    def __init__(self, n_cols, num_public_inputs=0, rounds=None, round_values=None):
        """rounds: None for a classic single-round AIR, or [(columns, verifier_challenges), ...] - round r commits
        that many columns (column indices run through the rounds in order) and, once its Merkle cap is in the
        transcript, that many base-field challenges are drawn; `challenge(k)` reads them in the order drawn."""
        self.n_cols = n_cols
        self.num_public_inputs = num_public_inputs
        self.rounds = rounds
        # round_values[r]: field elements the prover sends with round r (totals of accumulator columns); readable through
        # round_value(r, i).  The values array is public inputs | values r0 | challenges r0 | values r1 | challenges r1 ..
        self.round_values = list(round_values) if round_values is not None else ([0] * len(rounds) if rounds else [])
        assert len(self.round_values) == (len(rounds) if rounds else 0) and all(0 <= v <= 64 for v in self.round_values)
        if rounds is not None:
            if not 1 <= len(rounds) <= 3 or sum(c for c, _ in rounds) != n_cols or any(c < 1 for c, _ in rounds):
                raise ValueError("rounds must split the columns into 1..3 non-empty groups")
        self._emits = []  # (op, expr)
        self.segment_nodes = AIR_SEGMENT_NODES
        self.max_resident_leaves = AIR_MAX_RESIDENT_LEAVES   # more: fewer re-loads, a larger register file (LDS) per wave
        self._leaf_cache = {}
        self.period_bits = 0
        self._periodic = []  # value arrays, each of length 2^period_bits

    def _leaf(self, op, idx, degree, cnt=None):
        key = (op, idx, cnt)
        if key not in self._leaf_cache:
            self._leaf_cache[key] = _Expr(self, op, idx, cnt, degree)
        return self._leaf_cache[key]

    # three-operand forms for bit-valued columns (one VM instruction each)
    def _lift(self, x):
        return x if isinstance(x, _Expr) else self.const(x)

    def xor3(self, x, y, z):
        """x ^ y ^ z as a polynomial (exact on {0,1}): s = x + y - 2xy, s + z - 2sz."""
        x, y, z = self._lift(x), self._lift(y), self._lift(z)
        return _Expr(self, AIR_XOR3, x, y, x.degree + y.degree + z.degree, c=z)

    def ch(self, e, f, g):
        """g + e (f - g): e ? f : g."""
        e, f, g = self._lift(e), self._lift(f), self._lift(g)
        return _Expr(self, AIR_CH, e, f, e.degree + max(f.degree, g.degree), c=g)

    def maj(self, x, y, z):
        """xy + z (x + y - 2xy): majority of three bits."""
        x, y, z = self._lift(x), self._lift(y), self._lift(z)
        return _Expr(self, AIR_MAJ, x, y, x.degree + y.degree + z.degree, c=z)

    def pack(self, base, nbits, next_row=False):
        """sum_i 2^i * column[base + i] of the local (or next) row as ONE VM instruction: the word behind
        `nbits` bit columns.  Loads like a leaf (re-computable), degree 1."""
        assert 1 <= nbits <= 32 and 0 <= base and base + nbits <= self.n_cols
        return self._leaf(AIR_PACK_NEXT if next_row else AIR_PACK_LOCAL, base, 1, nbits)

    def local(self, i):
        assert 0 <= i < self.n_cols
        return self._leaf(AIR_LOCAL, i, 1)

    def next(self, i):
        assert 0 <= i < self.n_cols
        return self._leaf(AIR_NEXT, i, 1)

    def public(self, i):
        assert 0 <= i < self.num_public_inputs
        return self._leaf(AIR_PUBLIC, i, 0)

    def challenge(self, k):
        """The k-th verifier challenge of a multi-round AIR (a constant for the constraint degree)."""
        assert self.rounds is not None and 0 <= k < sum(n for _, n in self.rounds)
        return self._leaf(AIR_PUBLIC, self.num_public_inputs + self.challenge_offset(k), 0)

    def challenge_offset(self, k):
        """position of the k-th challenge among everything after the public inputs (round values come in between)"""
        off = 0
        for r, (_, n_ch) in enumerate(self.rounds):
            off += self.round_values[r]
            if k < n_ch:
                return off + k
            k -= n_ch
            off += n_ch
        raise IndexError(k)

    def round_value(self, r, i):
        """The i-th value the prover sends with round r (a constant for the constraint degree)."""
        assert self.rounds is not None and 0 <= r < len(self.rounds) and 0 <= i < self.round_values[r]
        off = sum(self.round_values[q] + self.rounds[q][1] for q in range(r))
        return self._leaf(AIR_PUBLIC, self.num_public_inputs + off + i, 0)

    def const(self, v):
        return self._leaf(AIR_CONST, int(v) % P, 0)

    def periodic(self, values):
        """A verifier-computable column repeating `values` (length a power of two) down the trace - round
        constants, round selectors, a lookup table.  Columns of different periods are tiled to the longest one.
        Degree 1."""
        values = np.array([int(v) % P for v in values], dtype=np.uint64)
        bits = (len(values) - 1).bit_length()
        if len(values) != 1 << bits or bits > 16:
            raise ValueError("the period must be a power of two <= 2^16")
        if bits > self.period_bits:
            self.period_bits = bits
        self._periodic = [np.tile(c, (1 << self.period_bits) // len(c)) for c in self._periodic + [values]]
        return self._leaf(AIR_PERIODIC, len(self._periodic) - 1, 1)

    # ConstraintConsumer
    def constraint_transition(self, e):
        self._emits.append((AIR_EMIT_TRANSITION, e, 1))

    def constraint_first_row(self, e):
        self._emits.append((AIR_EMIT_FIRST, e, 1))

    def constraint_last_row(self, e):
        self._emits.append((AIR_EMIT_LAST, e, 1))

    def constraint(self, e):
        self._emits.append((AIR_EMIT, e, 1))

    def constraint_boolean(self, col, count=1):
        """constraint(x * (x - 1)) for x = local(col) .. local(col + count - 1), in column order, as one VM
        instruction (eight loads in flight)."""
        assert 0 <= col and count >= 1 and col + count <= self.n_cols
        self._emits.append((AIR_EMIT_BOOL, self._leaf(AIR_LOCAL, col, 1), count))

    def constraint_logup(self, v1_col, v2_col, h_col, challenge=0):
        """The two base-field constraints of one LogUp helper h = 1/(alpha + v1) + 1/(alpha + v2) in the quadratic
        extension (logup.py), as ONE VM instruction: v1, v2 (None: a single-lookup helper) and (h, h + 1) are local
        columns, alpha = challenge(k) + challenge(k + 1) X.  Same constraint values as writing them out with the
        DSL, a thirtieth of the program words."""
        assert self.rounds is not None and 0 <= challenge and challenge + 1 < sum(n for _, n in self.rounds)
        assert self.challenge_offset(challenge + 1) == self.challenge_offset(challenge) + 1   # drawn in the same round
        challenge = self.challenge_offset(challenge)       # the instruction indexes the values array after the public inputs
        assert challenge < 63
        for c in (v1_col, h_col, h_col + 1) + (() if v2_col is None else (v2_col,)):
            assert 0 <= c < self.n_cols
        self._emits.append((AIR_EMIT_LOGUP, (v1_col, 0xFFFF if v2_col is None else v2_col, h_col, challenge), 2))

    @property
    def num_constraints(self):
        return sum(cnt for _, _, cnt in self._emits)

    @property
    def constraint_degree(self):
        """Stark::constraint_degree(): filters (z_last / lagrange) add one to the expression degree."""
        d = 1
        for op, e, _ in self._emits:
            if op == AIR_EMIT_LOGUP:
                d = max(d, 2 if e[1] == 0xFFFF else 3)
                continue
            d = max(d, 2 if op == AIR_EMIT_BOOL else e.degree + (0 if op == AIR_EMIT else 1))
        return d

    def quotient_degree_factor(self):
        """Stark::quotient_degree_factor() = max(1, constraint_degree - 1), rounded up to a power of two
        (compute_quotient_polys works on the coset of size n << log2_ceil(factor))."""
        q = max(1, self.constraint_degree - 1)
        return 1 << (q - 1).bit_length()

    def compile(self, segment_nodes=None):
        """Flatten the DAG into program words.

        * Shared sub-expressions (ADD / SUB / MUL nodes) and PACK words are computed once and stay in their
          register until the last use anywhere in the program.
        * Plain loads (trace / public / periodic / constant) are rematerialisable: at most
          AIR_MAX_RESIDENT_LEAVES stay resident (LRU), the rest are re-loaded, so the register file - the LDS
          footprint that bounds the quotient kernel's occupancy - stays small.
        * Loads are issued in batches: at a miss the assembler looks ahead in the constraint for the next
          loads it will need and emits them together behind one NLX_AIR_LOADV hint, so the kernel has up to
          AIR_LOAD_BATCH loads in flight per lane instead of one.
        * A long program is cut into segments (NLX_AIR_SEGMENT) of about `segment_nodes` arithmetic nodes
          (default: self.segment_nodes; 0 = one segment): no register is live across a boundary, so the GPU can
          evaluate the segments of one point on different waves.
        Constraints are emitted in declaration order (that order defines the alpha powers for prover and
        verifier alike)."""
        ops = _AIR_BINARY + _AIR_TERNARY
        packs = (AIR_PACK_LOCAL, AIR_PACK_NEXT)

        def is_value(x):
            return x.op in ops or x.op in packs

        # cut the constraint list into segments of about `segment_nodes` arithmetic nodes; a shared
        # sub-expression is recomputed in every segment that uses it
        budget = self.segment_nodes if segment_nodes is None else segment_nodes

        def dag(emit):
            """ids of the arithmetic nodes under a constraint"""
            eop, root, cnt = emit
            out, stack = set(), ([] if eop in _AIR_FUSED_EMITS else [root])
            while stack:
                x = stack.pop()
                if is_value(x) and id(x) not in out:
                    out.add(id(x))
                    stack.extend(x.operands())
            return out

        # A boundary is free where the next constraint shares nothing with the segment so far; otherwise the
        # shared nodes are computed again.  Cut at the first free boundary past the budget, or at twice the
        # budget if none comes.
        segments, cur, cur_cost, seen = [], [], 0, set()
        for emit in self._emits:
            nodes = dag(emit)
            full = len(nodes) if nodes else emit[2]
            cost = len(nodes - seen) if nodes else emit[2]
            over = budget and cur and cur_cost + cost > budget and len(segments) < AIR_MAX_SEGMENTS - 1
            if over and (cost == full or cur_cost + cost > 2 * budget):
                segments.append(cur)
                cur, cur_cost, seen, cost = [], 0, set(), full
            cur.append(emit)
            cur_cost += cost
            seen |= nodes
        segments.append(cur)
        words = []
        for i, seg in enumerate(segments):
            if i:
                words.append(AIR_SEGMENT)
            words += self._compile_segment(seg, is_value, packs)
        return np.array(words, dtype=np.uint64)

    def _compile_segment(self, emits, is_value, packs):
        """Program words of one segment (register allocation starts from an empty register file)."""
        batchable = (AIR_LOCAL, AIR_NEXT, AIR_PUBLIC, AIR_PERIODIC)
        computed, per_emit, all_vals = set(), [], []
        for eop, root, _ in emits:
            if eop in _AIR_FUSED_EMITS:
                per_emit.append([])
                continue
            nodes, seen, stack = [], set(), [(root, False)]
            while stack:
                x, done = stack.pop()
                if done:
                    nodes.append(x)
                    continue
                if not is_value(x) or id(x) in seen or id(x) in computed:
                    continue
                seen.add(id(x))
                stack.append((x, True))
                for y in reversed(x.operands()):
                    stack.append((y, False))
            for x in nodes:
                computed.add(id(x))
                all_vals.append(x)
            per_emit.append(nodes)
        for x in all_vals:
            x.uses, x.reg = 0, None
        for x in all_vals:
            for y in x.operands():
                if is_value(y):
                    y.uses += 1
        for eop, root, _ in emits:
            if eop not in _AIR_FUSED_EMITS and is_value(root):
                root.uses += 1

        words = []
        free = list(range(AIR_NUM_REGS))
        resident = []  # plain loads currently holding a register, least recently used first

        def take_free():
            r = min(free)
            free.remove(r)
            return r

        cur = {"seq": [], "pos": 0}

        def next_use(y):
            seq = cur["seq"]
            for i in range(cur["pos"], len(seq)):
                if seq[i] is y:
                    return i
            return 1 << 30

        def alloc(pinned=(), for_leaf=False, needed_at=None):
            """A free register, or the one of the resident load whose next use is farthest away (Belady).
            With `needed_at` (prefetch) the eviction is refused - returns None - when every candidate is needed
            sooner than the load being prefetched."""
            if free and not (for_leaf and len(resident) >= self.max_resident_leaves):
                return take_free()
            best, best_use = None, -1
            for y in resident:
                if any(y is q for q in pinned):
                    continue
                u = next_use(y)
                if u > best_use:
                    best, best_use = y, u
            if best is not None and (needed_at is None or best_use > needed_at):
                resident[:] = [q for q in resident if q is not best]
                r, best.reg = best.reg, None
                return r
            if needed_at is not None:
                return None
            if free:
                return take_free()
            raise ValueError("AIR needs more than %d live registers" % AIR_NUM_REGS)

        def touch(y):
            resident[:] = [q for q in resident if q is not y]
            resident.append(y)

        def load_word(y):
            return y.op | y.reg << 8 | y.a << 24

        def ensure(y, pinned, upcoming):
            """Register of operand y; `upcoming` = the plain loads this constraint will need next."""
            if is_value(y):
                return y.reg
            if y.reg is not None:
                touch(y)
                return y.reg
            if y.op == AIR_CONST:
                y.reg = alloc(pinned, True)
                words.append(AIR_CONST | y.reg << 8)
                words.append(y.a)
                touch(y)
                return y.reg
            y.reg = alloc(pinned, True)
            touch(y)
            batch = [y]
            for z in upcoming:
                if len(batch) >= AIR_LOAD_BATCH:
                    break
                if z.op in batchable and z.reg is None and not any(z is q for q in batch):
                    r = alloc(tuple(pinned) + tuple(batch), True, needed_at=next_use(z))
                    if r is None:
                        break
                    z.reg = r
                    touch(z)
                    batch.append(z)
            if len(batch) > 1:
                words.append(AIR_LOADV | len(batch) << 8)
            for z in batch:
                words.append(load_word(z))
            return y.reg

        def release(y):
            y.uses -= 1
            if y.uses == 0:
                if y.reg is not None:
                    free.append(y.reg)
                y.reg = None
                if not is_value(y):
                    resident[:] = [q for q in resident if q is not y]

        for (op, root, cnt), nodes in zip(emits, per_emit):
            if op == AIR_EMIT_BOOL:
                words.append(AIR_EMIT_BOOL | root.a << 24 | cnt << 40)
                continue
            if op == AIR_EMIT_LOGUP:
                v1, v2, hcol, k = root
                words.append(AIR_EMIT_LOGUP | v2 << 8 | v1 << 24 | hcol << 40 | k << 56)
                continue
            # plain-load use counts within this constraint, and their order of use
            leaf_seq = []
            for x in nodes:
                for y in x.operands():
                    if not is_value(y):
                        leaf_seq.append(y)
            if not is_value(root):
                leaf_seq.append(root)
            for y in leaf_seq:
                y.uses, y.reg = 0, None
            for y in leaf_seq:
                y.uses += 1
            pos = 0
            cur["seq"], cur["pos"] = leaf_seq, 0
            for x in nodes:
                if x.op in packs:
                    x.reg = alloc()
                    words.append(x.op | x.reg << 8 | x.a << 24 | x.b << 40)
                    continue
                opnds = x.operands()
                n_leaf = sum(1 for y in opnds if not is_value(y))
                regs = []
                for i, y in enumerate(opnds):
                    regs.append(ensure(y, opnds[:i], leaf_seq[pos:pos + 4 * AIR_LOAD_BATCH]))
                pos += n_leaf
                cur["pos"] = pos
                for y in opnds:
                    release(y)
                x.reg = alloc()
                third = regs[2] if len(regs) == 3 else x.sh
                words.append(x.op | x.reg << 8 | regs[0] << 24 | regs[1] << 40 | third << 56)
            words.append(op | ensure(root, (), leaf_seq[pos:]) << 24)
            release(root)
            assert not resident
        return words


class Stark:
    """A compiled AIR + config at a fixed trace length: what starky's `prove(stark, config, trace, pis)`
    takes.  `desc` is the C-ABI descriptor (nlx_stark_desc); `build(ctx)` makes it resident on a GPU."""

    def __init__(self, air, degree_bits, config=None, program=None):
        """program: a caller-assembled register program for the same constraints (default: air.compile()).  The program is
        part of the statement - the transcript opens with a digest of it - so two provers agree on proof bytes only if they
        run the same words."""
        self.air = air
        self.config = config or StarkConfig()
        self.program = air.compile() if program is None else np.ascontiguousarray(program, dtype=np.uint64)
        cfg = self.config
        qdf = air.quotient_degree_factor()
        if qdf > (1 << cfg.rate_bits):
            raise ValueError("constraint degree %d needs rate_bits >= %d" % (air.constraint_degree, qdf.bit_length() - 1))
        u64p = ctypes.POINTER(ctypes.c_uint64)
        self.periodic = (np.concatenate(air._periodic) if air._periodic else np.zeros(1, dtype=np.uint64))
        if air._periodic and air.period_bits > degree_bits:
            raise ValueError("period longer than the trace")
        self.desc = StarkDesc(degree_bits, air.n_cols, cfg.num_challenges, cfg.rate_bits, cfg.cap_height, qdf,
                              cfg.fri_pow_bits, cfg.fri_num_queries, cfg.fri_arity_bits, cfg.fri_final_poly_bits,
                              air.num_public_inputs, len(self.program), self.program.ctypes.data_as(u64p),
                              len(air._periodic), air.period_bits, self.periodic.ctypes.data_as(u64p))
        self.desc.leaf_group_cols = cfg.leaf_group_for(degree_bits, max(c for c, _ in air.rounds) if air.rounds is not None else air.n_cols)
        self.desc.openings_group = cfg.openings_group_for(air.n_cols)
        self.desc.batch_cols = int(cfg.batch_cols)
        if air.rounds is not None:
            self.desc.n_rounds = len(air.rounds)
            for r, (c, k) in enumerate(air.rounds):
                self.desc.round_cols[r] = c
                self.desc.round_challenges[r] = k
                self.desc.round_values[r] = air.round_values[r]
        self.degree_bits = degree_bits

    def build(self, ctx):
        return StarkProver(ctx, self)


_ROUND_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32,
                             ctypes.POINTER(ctypes.c_uint64))


class StarkProver:
    """Device-resident prover for one Stark (nlx_stark_build / nlx_stark_prove)."""

    def __init__(self, ctx, stark):
        self.ctx = ctx
        self.stark = stark
        h = ctypes.c_void_p()
        ctx.check(dll.nlx_stark_build(ctx.handle, ctypes.byref(stark.desc), ctypes.byref(h)))
        self.handle = h
        self._buf = np.zeros(dll.nlx_stark_proof_max_bytes(h), dtype=np.uint8)
        ctx._adopt(self)

    def prove(self, trace, public_inputs=()):
        """starky::prover::prove.  trace: (n_cols, n) uint64, column-major as starky's
        Vec<PolynomialValues> (host array or device tensor).  Returns the proof bytes."""
        air = self.stark.air
        if tuple(trace.shape) != (air.n_cols, 1 << self.stark.degree_bits):
            raise ValueError("trace must be (n_cols, n)")
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        if pis.size != air.num_public_inputs:
            raise ValueError("expected %d public inputs" % air.num_public_inputs)
        ln = ctypes.c_size_t()
        self.ctx.check(dll.nlx_stark_prove(self.handle, ptr(trace), ptr(pis) if pis.size else None,
                                           self._buf.ctypes.data, self._buf.size, ctypes.byref(ln)))
        return self._buf[:ln.value].tobytes()

    def prove_rounds(self, round_fn, public_inputs=()):
        """Multi-round proving (nlx_stark_prove_rounds).  round_fn(round, known: list[int]) returns round r's
        columns - a (round_cols[r], n) uint64 host array or a device tensor - computed from `known`, everything after
        the public inputs in the values array so far (round values and challenges of the earlier rounds); a round
        with round values returns (columns, values)."""
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        if pis.size != self.stark.air.num_public_inputs:
            raise ValueError("expected %d public inputs" % self.stark.air.num_public_inputs)
        desc = self.stark.desc
        keep, errors = [], []

        def cb(_user, rnd, ch_ptr, n_ch, values_out):
            try:
                arr = round_fn(rnd, [int(ch_ptr[i]) for i in range(n_ch)])
                n_rv = desc.round_values[rnd] if desc.n_rounds else 0
                if n_rv:                      # a round with values returns (columns, values)
                    arr, vals = arr
                    if len(vals) != n_rv:
                        raise ValueError("round %d: expected %d round values" % (rnd, n_rv))
                    for i, v in enumerate(vals):
                        values_out[i] = int(v) % P
                want = (desc.round_cols[rnd] if desc.n_rounds else desc.n_cols, 1 << desc.degree_bits)
                if tuple(arr.shape) != want:
                    raise ValueError("round %d: expected columns of shape %r" % (rnd, want))
                if isinstance(arr, np.ndarray):
                    arr = np.ascontiguousarray(arr, dtype=np.uint64)
                keep.append(arr)
                return ptr(arr)
            except Exception as e:  # an exception must not cross the C frame: NULL makes the call fail cleanly
                errors.append(e)
                return None

        fn = _ROUND_FN(cb)
        ln = ctypes.c_size_t()
        rc = dll.nlx_stark_prove_rounds(self.handle, fn, None, ptr(pis) if pis.size else None, self._buf.ctypes.data,
                                        self._buf.size, ctypes.byref(ln))
        if errors:
            raise errors[0]
        self.ctx.check(rc)
        return self._buf[:ln.value].tobytes()

    def prove_into(self, trace, public_inputs_ptr):
        ln = ctypes.c_size_t()
        self.ctx.check(dll.nlx_stark_prove(self.handle, ptr(trace), public_inputs_ptr, self._buf.ctypes.data,
                                           self._buf.size, ctypes.byref(ln)))
        return ln.value

    def stage_times(self):
        n = ctypes.c_uint32()
        names = (ctypes.c_char_p * 24)()
        ms = (ctypes.c_float * 24)()
        dll.nlx_stark_stage_times(self.handle, ctypes.byref(n), names, ms)
        return [(names[i].decode(), ms[i]) for i in range(n.value)]

    def close(self):
        if self.handle and self.ctx.handle:
            dll.nlx_stark_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# Example AIRs (starky's own FibonacciStark, and a wide synthetic AIR shaped like a hash-round table)
# ---------------------------------------------------------------------------------------------
def fibonacci_air():
    """starky::fibonacci_stark::FibonacciStark: columns (x0, x1); public inputs (x0[0], x1[0], x1[n-1])."""
    air = Air(2, 3)
    air.constraint_first_row(air.local(0) - air.public(0))
    air.constraint_first_row(air.local(1) - air.public(1))
    air.constraint_last_row(air.local(1) - air.public(2))
    air.constraint_transition(air.next(0) - air.local(1))
    air.constraint_transition(air.next(1) - air.local(0) - air.local(1))
    return air


def fibonacci_trace(degree_bits, x0=0, x1=1):
    n = 1 << degree_bits
    t = np.zeros((2, n), dtype=np.uint64)
    a, b = x0 % P, x1 % P
    for i in range(n):
        t[0, i], t[1, i] = a, b
        a, b = b, (a + b) % P
    return t, np.array([t[0, 0], t[1, 0], t[1, n - 1]], dtype=np.uint64)


def wide_air(n_cols=64, seed=1):
    """Synthetic degree-3 AIR, n_cols columns in groups of four (a, b, c, d):
         next.a = a*b + c        next.b = b*c + k1      next.c = (a + b + c) * d      d boolean: d*(d-1) = 0, next.d = d
       plus first-row pins on the public inputs.  Column count and multiplicative depth are those of a
       byte-oriented hash-round table (curta's SHA-256 / Ed25519 AIRs are hundreds of such columns)."""
    assert n_cols % 4 == 0
    air = Air(n_cols, 2)
    rng = np.random.default_rng(seed)
    air.k1 = np.array([int(rng.integers(1, P, dtype=np.uint64)) for _ in range(n_cols // 4)], dtype=np.uint64)
    for g in range(n_cols // 4):
        a, b, c, d = (air.local(4 * g + k) for k in range(4))
        na, nb, nc, nd = (air.next(4 * g + k) for k in range(4))
        air.constraint_transition(na - (a * b + c))
        air.constraint_transition(nb - (b * c + int(air.k1[g])))
        air.constraint_transition(nc - (a + b + c) * d)
        air.constraint(d * (d - 1))
        air.constraint_transition(nd - d)
    air.constraint_first_row(air.local(0) - air.public(0))
    air.constraint_first_row(air.local(1) - air.public(1))
    return air


def wide_trace(air, degree_bits, seed=1):
    """Satisfying witness of wide_air (generated by the native workload generator, nlx_synth_stark_trace)."""
    t = np.zeros((air.n_cols, 1 << degree_bits), dtype=np.uint64)
    pis = np.zeros(2, dtype=np.uint64)
    rc = synth_dll.nlx_synth_stark_trace(air.n_cols, degree_bits, seed, ptr(air.k1), ptr(t), ptr(pis))
    if rc != 0:
        raise NlxError(rc, "nlx_synth_stark_trace")
    return t, pis
