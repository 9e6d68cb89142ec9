"""SHA-256 compression as an AIR for the STARK prover.

nearx hashes headers, Merkle nodes and approval messages with `curta_sha256` (nearx/src/variables.rs:71-72,
nearx/src/merkle.rs:49, nearx/src/builder.rs:220,316); curta proves those hashes with a STARK whose AIR is
not in the reference (starkyx is un-vendored).  This is an independent AIR for the same function, written
for the register-program VM of include/nlx.h and laid out for the MI355X: a WIDE trace, sixteen rounds per
row and four rows per 512-bit block, so that every state bit is stored exactly once.  A narrow one-round-per-row
layout has to carry copies of the working variables and of the message window from row to row (302 columns x 64
rows = 19 328 cells per block in the first version of this file); here a row sees the sixteen rounds before it
through the (local, next) window, and a block costs 4 x 1 953 = 7 812 cells - 2.5x less to extend, hash and open.
All constraints have degree <= 3, so the quotient fits two chunks at rate_bits = 1.

Row layout.  Round slot j (0..15) of a row holds, for round t = 16 q + j of its block (q = row within the block):

    A[32], E[32]    bits of the working variables a_t, e_t produced by round t      (LSB first)
    W[32]           bits of the schedule word W_t
    CA[3], CE[3]    carry bits of the two additions of the round;  CW[2] carries of the schedule addition
    SW              the word the schedule recurrence gives for this slot (equal to W_t in rows q >= 1)

followed by

    PA[4][32], PE[4][32]   bits of a_{t0-1..t0-4}, e_{t0-1..t0-4} (t0 = 16 q): the state the row starts from
    HIN[8]                 the block's input chaining value (constant over its four rows)
    CY[8]                  carries of HIN + final state (meaningful in the last row of a block)
    IS_FIRST               1 in the first row of a block that starts a new message (chaining value = IV)

Periodic columns (period 4 rows): K_t for each of the sixteen slots, the first-row and last-row selectors.
The program (15 k instructions) is cut into segments (NLX_AIR_SEGMENT) that the GPU evaluates in parallel.
Every constraint is an all-rows constraint: a block boundary either chains or resets to the IV, so the relations
between consecutive rows also hold across the wrap from the last row to the first.

Soundness notes: a word that only ever enters additions (HIN, SW) may be off by a multiple of 2^32 without
consequence, because every sum is re-derived from boolean bit columns and the carries are bounded; equalities
between boolean vectors are enforced as packed-word equalities.  Public inputs: the eight words of the last
block's output chaining value (the digest of the last message).
"""
import hashlib
import struct

import numpy as np

from .stark import STEP_TAG_LEN, Air

SLOT = 105                      # columns per round slot
oA, oE, oW, oCA, oCE, oCW, oSW = 0, 32, 64, 96, 99, 102, 104
PA = 16 * SLOT                  # 1680
PE = PA + 128                   # 1808
HIN = PE + 128                  # 1936
CY = HIN + 8                    # 1944
IS_FIRST = CY + 8               # 1952
N_COLS = IS_FIRST + 1           # 1953 (round 0)
ACC = N_COLS                    # round 1: the binding accumulator, one element of F_p^2 = two base columns
ROWS_PER_BLOCK = 4

IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def _round_constants():
    """K[t] = first 32 bits of the fractional part of the cube root of the t-th prime (FIPS 180-4 §4.2.2),
    computed with integer arithmetic."""
    primes, c = [], 2
    while len(primes) < 64:
        if all(c % q for q in primes if q * q <= c):
            primes.append(c)
        c += 1

    def icbrt(v):
        lo, hi = 0, 1 << 48
        while lo < hi:
            mid = (lo + hi + 1) >> 1
            if mid * mid * mid <= v:
                lo = mid
            else:
                hi = mid - 1
        return lo

    return [icbrt(q << 96) & 0xFFFFFFFF for q in primes]


K = _round_constants()
assert K[0] == 0x428a2f98 and K[63] == 0xc67178f2


def sha256_air(tagged=False):
    air = Air(N_COLS + 2, 8 + (STEP_TAG_LEN if tagged else 0), rounds=[(N_COLS, 2), (2, 0)], round_values=[0, 2])   # tagged: public inputs 8..11 = the step tag
    L, N = air.local, air.next  # noqa: N806
    two32 = 1 << 32
    k_slot = [air.periodic([K[16 * q + j] for q in range(4)]) for j in range(16)]
    is_q0 = air.periodic([1, 0, 0, 0])
    is_q3 = air.periodic([0, 0, 0, 1])

    def weighted(terms):
        acc = terms[0]
        for i in range(1, 32):
            acc = acc + terms[i] * (1 << i)
        return acc

    # bit vectors of a_t / e_t for t relative to the row start (-4..15), on the local row
    def a_bits(t):
        base = t * SLOT + oA if t >= 0 else PA + 32 * (-t - 1)
        return [L(base + i) for i in range(32)]

    def e_bits(t):
        base = t * SLOT + oE if t >= 0 else PE + 32 * (-t - 1)
        return [L(base + i) for i in range(32)]

    def a_word(t, row_next=False):
        return air.pack(t * SLOT + oA if t >= 0 else PA + 32 * (-t - 1), 32, next_row=row_next)

    def e_word(t, row_next=False):
        return air.pack(t * SLOT + oE if t >= 0 else PE + 32 * (-t - 1), 32, next_row=row_next)

    # schedule words relative to the CURRENT row = `next`; negative indices reach into `local` (the previous row)
    def w_bits_cur(t):
        return [N(t * SLOT + oW + i) for i in range(32)] if t >= 0 else [L((16 + t) * SLOT + oW + i) for i in range(32)]

    def w_word_cur(t):
        return air.pack(t * SLOT + oW, 32, next_row=True) if t >= 0 else air.pack((16 + t) * SLOT + oW, 32)

    # 1. booleanity: all bits of the sixteen slots, the start state, the boundary carries and the flag
    for j in range(16):
        air.constraint_boolean(j * SLOT, 104)
    air.constraint_boolean(PA, 256)
    air.constraint_boolean(CY, 9)

    # 2. the sixteen rounds of the row
    for j in range(16):
        a1, a2, a3 = a_bits(j - 1), a_bits(j - 2), a_bits(j - 3)
        e1, e2, e3 = e_bits(j - 1), e_bits(j - 2), e_bits(j - 3)
        sig1 = weighted([air.xor3(e1[(i + 6) % 32], e1[(i + 11) % 32], e1[(i + 25) % 32]) for i in range(32)])
        ch = weighted([air.ch(e1[i], e2[i], e3[i]) for i in range(32)])
        sig0 = weighted([air.xor3(a1[(i + 2) % 32], a1[(i + 13) % 32], a1[(i + 22) % 32]) for i in range(32)])
        maj = weighted([air.maj(a1[i], a2[i], a3[i]) for i in range(32)])
        t1 = e_word(j - 4) + sig1 + ch + k_slot[j] + air.pack(j * SLOT + oW, 32)
        air.constraint(a_word(j) + air.pack(j * SLOT + oCA, 3) * two32 - (t1 + sig0 + maj))
        air.constraint(e_word(j) + air.pack(j * SLOT + oCE, 3) * two32 - (a_word(j - 4) + t1))
        # in rows q >= 1 the schedule word IS the recurrence's value (row 0 holds the message block)
        air.constraint((1 - is_q0) * (air.pack(j * SLOT + oW, 32) - L(j * SLOT + oSW)))

    # 3. the schedule recurrence, written on the current (= next) row with the previous row behind it:
    #    SW_t + 2^32 cw = s1(W[t-2]) + W[t-7] + s0(W[t-15]) + W[t-16]
    for j in range(16):
        w2, w15 = w_bits_cur(j - 2), w_bits_cur(j - 15)
        s0 = weighted([air.xor3(w15[(i + 7) % 32], w15[(i + 18) % 32], w15[i + 3]) if i + 3 < 32
                       else air.xor3(w15[(i + 7) % 32], w15[(i + 18) % 32], 0) for i in range(32)])
        s1 = weighted([air.xor3(w2[(i + 17) % 32], w2[(i + 19) % 32], w2[i + 10]) if i + 10 < 32
                       else air.xor3(w2[(i + 17) % 32], w2[(i + 19) % 32], 0) for i in range(32)])
        air.constraint(N(j * SLOT + oSW) + air.pack(j * SLOT + oCW, 2, next_row=True) * two32
                       - (s1 + w_word_cur(j - 7) + s0 + w_word_cur(j - 16)))

    # 4. row to row inside a block: the next row starts from this row's last four rounds; HIN is carried along
    nb = 1 - is_q3
    for k in range(4):
        air.constraint(nb * (a_word(-k - 1, True) - a_word(15 - k)))
        air.constraint(nb * (e_word(-k - 1, True) - e_word(15 - k)))
    for k in range(8):
        air.constraint(nb * (N(HIN + k) - L(HIN + k)))

    # 5. block boundary (this row is the last of its block): the next block starts from IV or from HIN + final state
    out = [a_word(15), a_word(14), a_word(13), a_word(12), e_word(15), e_word(14), e_word(13), e_word(12)]
    nxt = [a_word(-1, True), a_word(-2, True), a_word(-3, True), a_word(-4, True),
           e_word(-1, True), e_word(-2, True), e_word(-3, True), e_word(-4, True)]
    ho = [L(HIN + k) + out[k] - L(CY + k) * two32 for k in range(8)]
    for k in range(8):
        air.constraint(is_q3 * (nxt[k] - ho[k] - N(IS_FIRST) * (IV[k] - ho[k])))
        air.constraint(is_q3 * (N(HIN + k) - nxt[k]))

    # 6. the first row starts a message; 7. the last row's output chaining value is the public digest
    cur = [a_word(-1), a_word(-2), a_word(-3), a_word(-4), e_word(-1), e_word(-2), e_word(-3), e_word(-4)]
    air.constraint_first_row(L(IS_FIRST) - 1)
    for k in range(8):
        air.constraint_first_row(cur[k] - IV[k])
    for k in range(8):
        air.constraint_last_row(ho[k] - air.public(k))

    # 8. binding: a challenge gamma in F_p^2 (drawn once the trace is committed) and a round-1 accumulator fold every
    # block's (message-start flag, 16 message words) - on its first row - and 8 output chaining words - on its last row -
    # into one Horner fingerprint; the total is a round value of the proof, which the relying party recomputes from the
    # messages it believes were hashed (fingerprint() below).  Each row holds what was absorbed before it.
    def ext_mul(x, y):
        return x[0] * y[0] + x[1] * y[1] * 7, x[0] * y[1] + x[1] * y[0]
    gamma = (air.challenge(0), air.challenge(1))

    def horner(start, elems):
        """((start gamma + e_0) gamma + e_1) .. gamma + e_last: a chain, so that few values are live at a time"""
        c0, c1 = start
        for e in elems:
            c0, c1 = ext_mul((c0, c1), gamma)
            c0 = c0 + e
        return c0, c1
    acc = (L(ACC), L(ACC + 1))
    after_head = horner(acc, [L(IS_FIRST)] + [air.pack(j * SLOT + oW, 32) for j in range(16)])   # 17 elements, rows q = 0
    after_tail = horner(acc, ho)                                                                  # 8 elements, rows q = 3
    for c in range(2):
        air.constraint_first_row(acc[c])
        air.constraint_transition(N(ACC + c) - (acc[c] + is_q0 * (after_head[c] - acc[c]) + is_q3 * (after_tail[c] - acc[c])))
        air.constraint_last_row(after_tail[c] - air.round_value(1, c))
    return air


# ---------------------------------------------------------------------------------------------
# host helpers: padding, and a plain-Python trace generator (tests only; the product path generates the
# trace on the GPU with nlx_sha256_trace)
# ---------------------------------------------------------------------------------------------
def pad_message(msg):
    """FIPS 180-4 §5.1.1 padding -> list of 16-word blocks."""
    ml = len(msg) * 8
    data = msg + b"\x80" + b"\x00" * ((55 - len(msg)) % 64) + struct.pack(">Q", ml)
    assert len(data) % 64 == 0
    return [list(struct.unpack(">16I", data[i:i + 64])) for i in range(0, len(data), 64)]


def blocks_for_messages(messages, log_blocks=None):
    """The padded blocks of `messages`, preceded by as many empty messages as it takes to fill 2^log_blocks
    blocks - the filler goes FIRST so that the AIR's public output (the last block's chaining value) is the
    digest of the caller's last message.
    Returns (blocks uint32 [n_blocks,16], is_first uint8 [n_blocks], digest of the last message)."""
    blocks, first = [], []
    for m in messages:
        pb = pad_message(m)
        blocks += pb
        first += [1] + [0] * (len(pb) - 1)
    need = max(1, len(blocks))
    lb = (need - 1).bit_length() if log_blocks is None else log_blocks
    if need > (1 << lb):
        raise ValueError("messages need %d blocks > 2^%d" % (need, lb))
    fill = (1 << lb) - len(blocks)
    blocks = pad_message(b"") * fill + blocks
    first = [1] * fill + first
    digest = struct.unpack(">8I", hashlib.sha256(messages[-1] if messages else b"").digest())
    return (np.array(blocks, dtype=np.uint32), np.array(first, dtype=np.uint8),
            np.array(digest, dtype=np.uint64))


def _rotr(x, r):
    return ((x >> r) | (x << (32 - r))) & 0xFFFFFFFF


def _s0(x):
    return _rotr(x, 7) ^ _rotr(x, 18) ^ (x >> 3)


def _s1(x):
    return _rotr(x, 17) ^ _rotr(x, 19) ^ (x >> 10)


def _schedule(block):
    w = [int(x) for x in block]
    for i in range(16, 64):
        w.append((w[i - 16] + _s0(w[i - 15]) + w[i - 7] + _s1(w[i - 2])) & 0xFFFFFFFF)
    return w


def reference_trace(blocks, is_first):
    """(N_COLS, 4 * n_blocks) trace, plain Python.  Mirrors the column semantics documented above."""
    nb = len(blocks)
    n = ROWS_PER_BLOCK * nb
    t = np.zeros((N_COLS, n), dtype=np.uint64)

    def put_bits(base, row, v, cnt=32):
        for i in range(cnt):
            t[base + i, row] = (v >> i) & 1

    scheds = [_schedule(b) for b in blocks]
    h = list(IV)
    for bi in range(nb):
        if is_first[bi] or bi == 0:
            h = list(IV)
        w = scheds[bi]
        wprev = scheds[bi - 1]                       # the recurrence of row 0 looks back into the previous block (cyclic)
        # a[t + 4], e[t + 4] for t = -4..63
        a = [h[3], h[2], h[1], h[0]]
        e = [h[7], h[6], h[5], h[4]]
        ca, ce = [], []
        for r in range(64):
            a1, a2, a3, a4 = a[-1], a[-2], a[-3], a[-4]
            e1, e2, e3, e4 = e[-1], e[-2], e[-3], e[-4]
            s1 = _rotr(e1, 6) ^ _rotr(e1, 11) ^ _rotr(e1, 25)
            ch = (e1 & e2) ^ (~e1 & e3 & 0xFFFFFFFF)
            s0 = _rotr(a1, 2) ^ _rotr(a1, 13) ^ _rotr(a1, 22)
            mj = (a1 & a2) ^ (a1 & a3) ^ (a2 & a3)
            t1 = e4 + s1 + ch + K[r] + w[r]
            sa, se = t1 + s0 + mj, a4 + t1
            a.append(sa & 0xFFFFFFFF)
            e.append(se & 0xFFFFFFFF)
            ca.append(sa >> 32)
            ce.append(se >> 32)
        for q in range(4):
            row = 4 * bi + q
            for j in range(16):
                r = 16 * q + j
                base = j * SLOT
                put_bits(base + oA, row, a[r + 4])
                put_bits(base + oE, row, e[r + 4])
                put_bits(base + oW, row, w[r])
                put_bits(base + oCA, row, ca[r], 3)
                put_bits(base + oCE, row, ce[r], 3)
                ext = (wprev[48:] + w) if q == 0 else None   # W[-16..-1] of row 0 = previous block's W[48..63]
                if q == 0:
                    wm = lambda k: ext[16 + k]               # noqa: E731  k in -16..15
                    sw = _s1(wm(j - 2)) + wm(j - 7) + _s0(wm(j - 15)) + wm(j - 16)
                else:
                    sw = _s1(w[r - 2]) + w[r - 7] + _s0(w[r - 15]) + w[r - 16]
                t[base + oSW, row] = sw & 0xFFFFFFFF
                put_bits(base + oCW, row, sw >> 32, 2)
            for k in range(4):
                put_bits(PA + 32 * k, row, a[16 * q + 4 - 1 - k])
                put_bits(PE + 32 * k, row, e[16 * q + 4 - 1 - k])
            for k in range(8):
                t[HIN + k, row] = h[k]
            if q == 0:
                t[IS_FIRST, row] = 1 if (is_first[bi] or bi == 0) else 0
        fin = [a[67], a[66], a[65], a[64], e[67], e[66], e[65], e[64]]
        for k in range(8):
            t[CY + k, 4 * bi + 3] = (h[k] + fin[k]) >> 32
        h = [(h[k] + fin[k]) & 0xFFFFFFFF for k in range(8)]
    return t, np.array(h, dtype=np.uint64)


def _ext_mul(x, y):
    gl = 0xFFFFFFFF00000001
    return (x[0] * y[0] + 7 * x[1] * y[1]) % gl, (x[0] * y[1] + x[1] * y[0]) % gl


def block_outputs(blocks, is_first):
    """output chaining value of every block (the eight words the AIR absorbs on a block's last row)"""
    outs, h = [], list(IV)
    for bi, blk in enumerate(blocks):
        if is_first[bi] or bi == 0:
            h = list(IV)
        w = _schedule(blk)
        a, b, c, d, e, f, g, hh = h
        for r in range(64):
            t1 = (hh + (_rotr(e, 6) ^ _rotr(e, 11) ^ _rotr(e, 25)) + ((e & f) ^ (~e & g & 0xFFFFFFFF)) + K[r] + w[r]) & 0xFFFFFFFF
            t2 = ((_rotr(a, 2) ^ _rotr(a, 13) ^ _rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))) & 0xFFFFFFFF
            hh, g, f, e, d, c, b, a = g, f, e, (d + t1) & 0xFFFFFFFF, c, b, a, (t1 + t2) & 0xFFFFFFFF
        h = [(x + y) & 0xFFFFFFFF for x, y in zip(h, (a, b, c, d, e, f, g, hh))]
        outs.append(list(h))
    return outs


def fingerprint(blocks, is_first, gamma):
    """What the proof's round value must be for these padded blocks (blocks_for_messages): Horner in F_p^2 over, block by
    block, (message-start flag, 16 message words, 8 output chaining words) - the relying party's side of the binding."""
    gl = 0xFFFFFFFF00000001
    acc = (0, 0)
    for bi, (blk, out) in enumerate(zip(blocks, block_outputs(blocks, is_first))):
        for v in [1 if (is_first[bi] or bi == 0) else 0] + [int(x) for x in blk] + out:
            acc = _ext_mul(acc, gamma)
            acc = ((acc[0] + v) % gl, acc[1])
    return acc


def binding_columns(blocks, is_first, gamma):
    """Round-1 accumulator columns (2, 4 n_blocks) and the total, plain Python (tests and the oracle path)."""
    gl = 0xFFFFFFFF00000001
    n = ROWS_PER_BLOCK * len(blocks)
    out = np.zeros((2, n), dtype=np.uint64)
    acc = (0, 0)
    for bi, (blk, ho) in enumerate(zip(blocks, block_outputs(blocks, is_first))):
        for q in range(4):
            out[0, 4 * bi + q], out[1, 4 * bi + q] = acc
            elems = ([1 if (is_first[bi] or bi == 0) else 0] + [int(x) for x in blk]) if q == 0 else (ho if q == 3 else [])
            for v in elems:
                acc = _ext_mul(acc, gamma)
                acc = ((acc[0] + v) % gl, acc[1])
    return out, acc


def cpu_rounds(blocks, is_first, trace):
    """round function for a CPU prover of this AIR (tests, the bench's cpu_baseline): round 0 = the given trace, round 1 = the
    binding accumulator and its total for the gamma the prover drew"""
    def fn(rnd, known):
        if rnd == 0:
            return trace
        cols, total = binding_columns(blocks, is_first, known[:2])
        return cols, list(total)
    return fn


class Sha256Prover:
    """Proves SHA-256 of a batch of messages on one GPU: trace generation (nlx_sha256_trace) straight into HBM, the
    binding accumulator (nlx_sha256_bind_round) once the prover has drawn gamma, and the two-round STARK
    (nlx_stark_prove_rounds) on the device-resident columns.  2^log_blocks compression blocks per proof."""

    def __init__(self, ctx, log_blocks, config=None, segment_nodes=None, step_tag=None):
        from .stark import Stark
        self.ctx = ctx
        self.step_tag = None if step_tag is None else [int(v) for v in step_tag]   # see stark.step_tag
        self.log_blocks = log_blocks
        if log_blocks < 2:
            raise ValueError("at least four blocks per proof (a block is four trace rows, a STARK at least sixteen)")
        air = sha256_air(tagged=self.step_tag is not None)
        if segment_nodes is not None:
            air.segment_nodes = segment_nodes
        self.stark = Stark(air, log_blocks + 2, config)
        self.prover = self.stark.build(ctx)
        self._trace = self._acc = None
        self.last_total = None

    def generate_trace(self, blocks, is_first):
        """Returns (device round-0 trace tensor [N_COLS, n] int64, digest words uint64[8])."""
        import torch
        from ._lib import dll
        blocks = np.ascontiguousarray(blocks, dtype=np.uint32)
        is_first = np.ascontiguousarray(is_first, dtype=np.uint8)
        if blocks.shape != (1 << self.log_blocks, 16) or is_first.shape != (1 << self.log_blocks,):
            raise ValueError("expected 2^%d blocks" % self.log_blocks)
        n = ROWS_PER_BLOCK << self.log_blocks
        if self._trace is None:
            self._trace = torch.empty((N_COLS, n), dtype=torch.int64, device="cuda:%d" % self.ctx.device)
            self._acc = torch.empty((2, n), dtype=torch.int64, device=self._trace.device)
        digest = np.zeros(8, dtype=np.uint64)
        self.ctx.check(dll.nlx_sha256_trace(self.ctx.handle, blocks.ctypes.data, is_first.ctypes.data, self.log_blocks,
                                            self._trace.data_ptr(), digest.ctypes.data))
        return self._trace, digest

    def round1(self, known):
        """The binding accumulator for gamma = known[0:2] (device columns) and its total, the proof's round value."""
        from ._lib import dll
        gamma = np.array([int(known[0]), int(known[1])], dtype=np.uint64)
        total = np.zeros(2, dtype=np.uint64)
        self.ctx.check(dll.nlx_sha256_bind_round(self.ctx.handle, self._trace.data_ptr(), self.log_blocks, gamma.ctypes.data,
                                                 self._acc.data_ptr(), total.ctypes.data))
        self.last_total = (int(total[0]), int(total[1]))
        return self._acc, [int(total[0]), int(total[1])]

    def prove_trace(self, digest):
        """The proof for the trace generate_trace() left on the device (public inputs: the last digest)."""
        pis = [int(v) for v in digest] + (self.step_tag or [])
        return self.prover.prove_rounds(lambda rnd, known: self._trace if rnd == 0 else self.round1(known), pis)

    def prove(self, messages):
        """Returns (proof bytes, digest words of the last message in the batch)."""
        blocks, first, want = blocks_for_messages(messages, self.log_blocks)
        _, digest = self.generate_trace(blocks, first)
        assert np.array_equal(digest, want)  # the GPU's chaining value is the real SHA-256 digest
        return self.prove_trace(digest), digest

    def close(self):
        self.prover.close()
        self._trace = self._acc = None
