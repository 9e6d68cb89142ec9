"""SHA-512 compression as an AIR for the STARK prover.

Every Ed25519 verification nearx proves (`curta_eddsa_verify_sigs_conditional`, nearx/src/builder.rs:152) hashes
R || A || M with SHA-512 - for NEAR approvals 32 + 32 + 41 = 105 bytes, one 1024-bit block per signature.  curta's
AIR for it is not in the reference (starkyx is un-vendored); this is an independent AIR for the same function, the
64-bit sibling of sha256_air.py: a WIDE trace, twenty rounds per row and four rows per block, every state bit stored
once.  A 64-bit word does not fit a Goldilocks element uniquely, so every word is handled as two 32-bit halves and
every addition mod 2^64 is two additions mod 2^32 chained by the low half's carry.  All constraints have degree <= 3
(quotient factor 2 at rate_bits = 1).

Row layout.  Round slot j (0..19) of a row holds, for round t = 20 q + j of its block (q = row within the block):

    A[64], E[64]      bits of the working variables a_t, e_t produced by round t      (LSB first)
    W[64]             bits of the schedule word W_t
    CA[3+3], CE[3+3]  carry bits (low half, high half) of the two additions of the round
    CW[2+2]           carry bits of the schedule addition
    SW[2]             the halves of the word the schedule recurrence gives for this slot (= W_t in rows q >= 1)

followed by

    PA[4][64], PE[4][64]   bits of a_{t0-1..t0-4}, e_{t0-1..t0-4} (t0 = 20 q): the state the row starts from
    HIN[8][2]              the block's input chaining value as (low, high) halves, constant over its four rows
    CY[8][2]               carries of HIN + final state (meaningful in the last row of a block)
    IS_FIRST               1 in the first row of a block that starts a new message (chaining value = IV)

Periodic columns (period 4 rows): the halves of K_t for each of the twenty slots, the first-row-of-block and
last-row-of-block selectors.  Every constraint is an all-rows constraint (a block boundary either chains or resets
to the IV, so the row-to-row relations also hold across the wrap).  Public inputs: the sixteen halves (low, high per
word) of the last block's output chaining value - the digest of the last message.
"""
import hashlib
import math
import struct

import numpy as np

from .stark import STEP_TAG_LEN, Air

ROUNDS = 80
SLOTS = 20                      # rounds per row
ROWS_PER_BLOCK = 4
SLOT = 210                      # columns per round slot
oA, oE, oW, oCA, oCE, oCW, oSW = 0, 64, 128, 192, 198, 204, 208
PA = SLOTS * SLOT               # 4200
PE = PA + 256                   # 4456
HIN = PE + 256                  # 4712
CY = HIN + 16                   # 4728
IS_FIRST = CY + 16              # 4744
N_COLS = IS_FIRST + 1           # 4745 (round 0)
ACC = N_COLS                    # round 1: the binding accumulator, one element of F_p^2 = two base columns
M64 = (1 << 64) - 1


def _first_primes(count):
    primes, c = [], 2
    while len(primes) < count:
        if all(c % q for q in primes if q * q <= c):
            primes.append(c)
        c += 1
    return primes


def _icbrt(v):
    lo, hi = 0, 1 << ((v.bit_length() + 2) // 3 + 1)
    while lo < hi:
        mid = (lo + hi + 1) >> 1
        if mid * mid * mid <= v:
            lo = mid
        else:
            hi = mid - 1
    return lo


# FIPS 180-4 §4.2.3 / §5.3.5: first 64 bits of the fractional parts of the cube / square roots of the first primes
K = [_icbrt(q << 192) & M64 for q in _first_primes(80)]
IV = [math.isqrt(q << 128) & M64 for q in _first_primes(8)]
assert K[0] == 0x428a2f98d728ae22 and K[79] == 0x6c44198c4a475817 and IV[0] == 0x6a09e667f3bcc908 and IV[7] == 0x5be0cd19137e2179


def _halves(x):
    return x & 0xFFFFFFFF, x >> 32


def sha512_air(tagged=False):
    air = Air(N_COLS + 2, 16 + (STEP_TAG_LEN if tagged else 0), rounds=[(N_COLS, 2), (2, 0)], round_values=[0, 2])   # tagged: public inputs 16..19 = the step tag
    L, N = air.local, air.next  # noqa: N806
    two32 = 1 << 32
    k_slot = [[air.periodic([_halves(K[SLOTS * q + j])[h] for q in range(4)]) for h in range(2)] for j in range(SLOTS)]
    is_q0 = air.periodic([1, 0, 0, 0])
    is_q3 = air.periodic([0, 0, 0, 1])

    def weighted(bits32):
        acc = bits32[0]
        for i in range(1, 32):
            acc = acc + bits32[i] * (1 << i)
        return acc

    def halves_of(bits64):
        return weighted(bits64[:32]), weighted(bits64[32:])

    def a_base(t):
        return t * SLOT + oA if t >= 0 else PA + 64 * (-t - 1)

    def e_base(t):
        return t * SLOT + oE if t >= 0 else PE + 64 * (-t - 1)

    def a_bits(t):
        return [L(a_base(t) + i) for i in range(64)]

    def e_bits(t):
        return [L(e_base(t) + i) for i in range(64)]

    def word(base, row_next=False):
        """(low half, high half) of the 64 bit columns starting at `base`"""
        return air.pack(base, 32, next_row=row_next), air.pack(base + 32, 32, next_row=row_next)

    # schedule words relative to the CURRENT row = `next`; negative indices reach into `local` (the previous row)
    def w_bits_cur(t):
        return [N(t * SLOT + oW + i) for i in range(64)] if t >= 0 else [L((SLOTS + t) * SLOT + oW + i) for i in range(64)]

    def w_word_cur(t):
        return word(t * SLOT + oW, True) if t >= 0 else word((SLOTS + t) * SLOT + oW)

    # 1. booleanity: all bits of the twenty slots, the start state, the boundary carries and the flag
    for j in range(SLOTS):
        air.constraint_boolean(j * SLOT, 208)
    air.constraint_boolean(PA, 512)
    air.constraint_boolean(CY, 17)

    # 2. the twenty rounds of the row
    for j in range(SLOTS):
        a1, a2, a3 = a_bits(j - 1), a_bits(j - 2), a_bits(j - 3)
        e1, e2, e3 = e_bits(j - 1), e_bits(j - 2), e_bits(j - 3)
        sig1 = halves_of([air.xor3(e1[(i + 14) % 64], e1[(i + 18) % 64], e1[(i + 41) % 64]) for i in range(64)])
        ch = halves_of([air.ch(e1[i], e2[i], e3[i]) for i in range(64)])
        sig0 = halves_of([air.xor3(a1[(i + 28) % 64], a1[(i + 34) % 64], a1[(i + 39) % 64]) for i in range(64)])
        maj = halves_of([air.maj(a1[i], a2[i], a3[i]) for i in range(64)])
        e4, a4 = word(e_base(j - 4)), word(a_base(j - 4))
        wj = word(j * SLOT + oW)
        an, en = word(a_base(j)), word(e_base(j))
        ca = (air.pack(j * SLOT + oCA, 3), air.pack(j * SLOT + oCA + 3, 3))
        ce = (air.pack(j * SLOT + oCE, 3), air.pack(j * SLOT + oCE + 3, 3))
        t1 = [e4[h] + sig1[h] + ch[h] + k_slot[j][h] + wj[h] for h in range(2)]
        air.constraint(an[0] + ca[0] * two32 - (t1[0] + sig0[0] + maj[0]))
        air.constraint(an[1] + ca[1] * two32 - (t1[1] + sig0[1] + maj[1] + ca[0]))
        air.constraint(en[0] + ce[0] * two32 - (a4[0] + t1[0]))
        air.constraint(en[1] + ce[1] * two32 - (a4[1] + t1[1] + ce[0]))
        # in rows q >= 1 the schedule word IS the recurrence's value (row 0 holds the message block)
        for h in range(2):
            air.constraint((1 - is_q0) * (wj[h] - L(j * SLOT + oSW + h)))

    # 3. the schedule recurrence, written on the current (= next) row with the previous row behind it:
    #    SW_t + 2^64 cw = s1(W[t-2]) + W[t-7] + s0(W[t-15]) + W[t-16], in halves
    for j in range(SLOTS):
        w2, w15 = w_bits_cur(j - 2), w_bits_cur(j - 15)
        s0 = halves_of([air.xor3(w15[(i + 1) % 64], w15[(i + 8) % 64], w15[i + 7] if i + 7 < 64 else 0) for i in range(64)])
        s1 = halves_of([air.xor3(w2[(i + 19) % 64], w2[(i + 61) % 64], w2[i + 6] if i + 6 < 64 else 0) for i in range(64)])
        w7, w16 = w_word_cur(j - 7), w_word_cur(j - 16)
        cw = (air.pack(j * SLOT + oCW, 2, next_row=True), air.pack(j * SLOT + oCW + 2, 2, next_row=True))
        air.constraint(N(j * SLOT + oSW) + cw[0] * two32 - (s1[0] + w7[0] + s0[0] + w16[0]))
        air.constraint(N(j * SLOT + oSW + 1) + cw[1] * two32 - (s1[1] + w7[1] + s0[1] + w16[1] + cw[0]))

    # 4. row to row inside a block: the next row starts from this row's last four rounds and HIN is carried along;
    # 5. block boundary (this row is the last of its block): the next block starts from IV or from HIN + final state;
    # 6. the first row starts a message; 7. the last row's output chaining value is the public digest.
    # Written word by word so that each word's packed halves are used while they are in registers.
    nb = 1 - is_q3
    air.constraint_first_row(L(IS_FIRST) - 1)
    out_halves = []                                             # the block's output chaining value, (low, high) per word
    for k in range(8):
        base_out = a_base(SLOTS - 1 - k) if k < 4 else e_base(SLOTS - 1 - (k - 4))
        base_start = a_base(-k - 1) if k < 4 else e_base(-(k - 4) - 1)
        out, nxt, cur = word(base_out), word(base_start, True), word(base_start)
        lo = L(HIN + 2 * k) + out[0] - L(CY + 2 * k) * two32
        hi = L(HIN + 2 * k + 1) + out[1] + L(CY + 2 * k) - L(CY + 2 * k + 1) * two32
        out_halves += [lo, hi]
        for h, ho in enumerate((lo, hi)):
            iv = _halves(IV[k])[h]
            air.constraint(nb * (nxt[h] - out[h]))
            air.constraint(nb * (N(HIN + 2 * k + h) - L(HIN + 2 * k + h)))
            air.constraint(is_q3 * (nxt[h] - ho - N(IS_FIRST) * (iv - ho)))
            air.constraint(is_q3 * (N(HIN + 2 * k + h) - nxt[h]))
            air.constraint_first_row(cur[h] - iv)
            air.constraint_last_row(ho - air.public(2 * k + h))

    # 8. binding (as in sha256_air.py): a challenge gamma in F_p^2 and a round-1 accumulator fold every block's
    # (message-start flag, 16 message words as (low, high) halves) - on its first row - and 8 output chaining words as
    # halves - on its last row - into one Horner fingerprint; the total is a round value of the proof.
    def ext_mul(x, y):
        return x[0] * y[0] + x[1] * y[1] * 7, x[0] * y[1] + x[1] * y[0]
    gamma = (air.challenge(0), air.challenge(1))

    def horner(start, elems):
        c0, c1 = start
        for e in elems:
            c0, c1 = ext_mul((c0, c1), gamma)
            c0 = c0 + e
        return c0, c1
    acc = (L(ACC), L(ACC + 1))
    msg = [L(IS_FIRST)]
    for j in range(16):
        msg += list(word(j * SLOT + oW))
    after_head, after_tail = horner(acc, msg), horner(acc, out_halves)      # 33 elements on rows q = 0, 16 on rows q = 3
    for c in range(2):
        air.constraint_first_row(acc[c])
        air.constraint_transition(N(ACC + c) - (acc[c] + is_q0 * (after_head[c] - acc[c]) + is_q3 * (after_tail[c] - acc[c])))
        air.constraint_last_row(after_tail[c] - air.round_value(1, c))
    return air


# ---------------------------------------------------------------------------------------------
# host helpers: padding, and a plain-Python trace generator (tests only; the product path generates the
# trace on the GPU with nlx_sha512_trace)
# ---------------------------------------------------------------------------------------------
def pad_message(msg):
    """FIPS 180-4 §5.1.2 padding -> list of 16-word (64-bit) blocks."""
    ml = len(msg) * 8
    data = msg + b"\x80" + b"\x00" * ((111 - len(msg)) % 128) + struct.pack(">QQ", ml >> 64, ml & M64)
    assert len(data) % 128 == 0
    return [list(struct.unpack(">16Q", data[i:i + 128])) for i in range(0, len(data), 128)]


def digest_halves(words):
    """eight 64-bit words -> the sixteen public inputs (low, high per word)"""
    out = []
    for w in words:
        out += list(_halves(int(w)))
    return np.array(out, dtype=np.uint64)


def blocks_for_messages(messages, log_blocks=None):
    """The padded blocks of `messages`, preceded by as many empty messages as it takes to fill 2^log_blocks blocks
    (the filler goes FIRST so that the AIR's public output is the digest of the caller's last message).
    Returns (blocks uint64 [n_blocks,16], is_first uint8 [n_blocks], digest words uint64[8] of the last message)."""
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
    digest = struct.unpack(">8Q", hashlib.sha512(messages[-1] if messages else b"").digest())
    return (np.array(blocks, dtype=np.uint64), np.array(first, dtype=np.uint8), np.array(digest, dtype=np.uint64))


def _rotr(x, r):
    return ((x >> r) | (x << (64 - r))) & M64


def _s0(x):
    return _rotr(x, 1) ^ _rotr(x, 8) ^ (x >> 7)


def _s1(x):
    return _rotr(x, 19) ^ _rotr(x, 61) ^ (x >> 6)


def _schedule(block):
    w = [int(x) for x in block]
    for i in range(16, ROUNDS):
        w.append((w[i - 16] + _s0(w[i - 15]) + w[i - 7] + _s1(w[i - 2])) & M64)
    return w


def _add_halves(terms):
    """sum of 64-bit terms mod 2^64 as the AIR does it: (value, low-half carry, high-half carry)"""
    lo = sum(t & 0xFFFFFFFF for t in terms)
    c_lo = lo >> 32
    hi = sum(t >> 32 for t in terms) + c_lo
    return ((hi & 0xFFFFFFFF) << 32) | (lo & 0xFFFFFFFF), c_lo, hi >> 32


def reference_trace(blocks, is_first):
    """(N_COLS, 4 * n_blocks) trace, plain Python.  Mirrors the column semantics documented above."""
    nb = len(blocks)
    n = ROWS_PER_BLOCK * nb
    t = np.zeros((N_COLS, n), dtype=np.uint64)

    def put_bits(base, row, v, cnt=64):
        for i in range(cnt):
            t[base + i, row] = (v >> i) & 1

    scheds = [_schedule(b) for b in blocks]
    h = list(IV)
    for bi in range(nb):
        if is_first[bi] or bi == 0:
            h = list(IV)
        w = scheds[bi]
        ext = scheds[bi - 1][ROUNDS - 16:] + w            # ext[16 + t] = W_t; t < 0 reaches the previous block (cyclic)
        a = [h[3], h[2], h[1], h[0]]                      # a[t + 4], e[t + 4] for t = -4..79
        e = [h[7], h[6], h[5], h[4]]
        ca, ce = [], []
        for r in range(ROUNDS):
            a1, a2, a3, a4 = a[-1], a[-2], a[-3], a[-4]
            e1, e2, e3, e4 = e[-1], e[-2], e[-3], e[-4]
            s1 = _rotr(e1, 14) ^ _rotr(e1, 18) ^ _rotr(e1, 41)
            ch = (e1 & e2) ^ (~e1 & e3 & M64)
            s0 = _rotr(a1, 28) ^ _rotr(a1, 34) ^ _rotr(a1, 39)
            mj = (a1 & a2) ^ (a1 & a3) ^ (a2 & a3)
            va, ca_lo, ca_hi = _add_halves([e4, s1, ch, K[r], w[r], s0, mj])
            ve, ce_lo, ce_hi = _add_halves([a4, e4, s1, ch, K[r], w[r]])
            a.append(va)
            e.append(ve)
            ca.append((ca_lo, ca_hi))
            ce.append((ce_lo, ce_hi))
        for q in range(ROWS_PER_BLOCK):
            row = ROWS_PER_BLOCK * bi + q
            for j in range(SLOTS):
                r = SLOTS * q + j
                base = j * SLOT
                put_bits(base + oA, row, a[r + 4])
                put_bits(base + oE, row, e[r + 4])
                put_bits(base + oW, row, w[r])
                put_bits(base + oCA, row, ca[r][0], 3)
                put_bits(base + oCA + 3, row, ca[r][1], 3)
                put_bits(base + oCE, row, ce[r][0], 3)
                put_bits(base + oCE + 3, row, ce[r][1], 3)
                sw, cw_lo, cw_hi = _add_halves([_s1(ext[16 + r - 2]), ext[16 + r - 7], _s0(ext[16 + r - 15]), ext[16 + r - 16]])
                t[base + oSW, row], t[base + oSW + 1, row] = _halves(sw)
                put_bits(base + oCW, row, cw_lo, 2)
                put_bits(base + oCW + 2, row, cw_hi, 2)
            for k in range(4):
                put_bits(PA + 64 * k, row, a[SLOTS * q + 3 - k])
                put_bits(PE + 64 * k, row, e[SLOTS * q + 3 - k])
            for k in range(8):
                t[HIN + 2 * k, row], t[HIN + 2 * k + 1, row] = _halves(h[k])
            if q == 0:
                t[IS_FIRST, row] = 1 if (is_first[bi] or bi == 0) else 0
        fin = [a[83], a[82], a[81], a[80], e[83], e[82], e[81], e[80]]
        for k in range(8):
            v, c_lo, c_hi = _add_halves([h[k], fin[k]])
            t[CY + 2 * k, ROWS_PER_BLOCK * bi + 3], t[CY + 2 * k + 1, ROWS_PER_BLOCK * bi + 3] = c_lo, c_hi
            h[k] = v
    return t, np.array(h, dtype=np.uint64)


def _ext_mul(x, y):
    gl = 0xFFFFFFFF00000001
    return (x[0] * y[0] + 7 * x[1] * y[1]) % gl, (x[0] * y[1] + x[1] * y[0]) % gl


def block_outputs(blocks, is_first):
    """output chaining value (eight 64-bit words) of every block"""
    outs, h = [], list(IV)
    for bi, blk in enumerate(blocks):
        if is_first[bi] or bi == 0:
            h = list(IV)
        w = _schedule(blk)
        a, b, c, d, e, f, g, hh = h
        for r in range(ROUNDS):
            t1 = (hh + (_rotr(e, 14) ^ _rotr(e, 18) ^ _rotr(e, 41)) + ((e & f) ^ (~e & g & M64)) + K[r] + w[r]) & M64
            t2 = ((_rotr(a, 28) ^ _rotr(a, 34) ^ _rotr(a, 39)) + ((a & b) ^ (a & c) ^ (b & c))) & M64
            hh, g, f, e, d, c, b, a = g, f, e, (d + t1) & M64, c, b, a, (t1 + t2) & M64
        h = [(x + y) & M64 for x, y in zip(h, (a, b, c, d, e, f, g, hh))]
        outs.append(list(h))
    return outs


def _block_elements(blk, first, out):
    head = [1 if first else 0]
    for wv in blk:
        head += list(_halves(int(wv)))
    tail = []
    for wv in out:
        tail += list(_halves(int(wv)))
    return head, tail


def fingerprint(blocks, is_first, gamma):
    """What the proof's round value must be for these padded blocks: Horner in F_p^2 over, block by block, the message-start
    flag, the 16 message words and the 8 output chaining words as (low, high) halves."""
    gl = 0xFFFFFFFF00000001
    acc = (0, 0)
    for bi, (blk, out) in enumerate(zip(blocks, block_outputs(blocks, is_first))):
        head, tail = _block_elements(blk, is_first[bi] or bi == 0, out)
        for v in head + tail:
            acc = _ext_mul(acc, gamma)
            acc = ((acc[0] + v) % gl, acc[1])
    return acc


def binding_columns(blocks, is_first, gamma):
    """Round-1 accumulator columns (2, 4 n_blocks) and the total, plain Python (tests and the oracle path)."""
    gl = 0xFFFFFFFF00000001
    out = np.zeros((2, ROWS_PER_BLOCK * len(blocks)), dtype=np.uint64)
    acc = (0, 0)
    for bi, (blk, ho) in enumerate(zip(blocks, block_outputs(blocks, is_first))):
        head, tail = _block_elements(blk, is_first[bi] or bi == 0, ho)
        for q in range(4):
            out[0, 4 * bi + q], out[1, 4 * bi + q] = acc
            for v in (head if q == 0 else (tail if q == 3 else [])):
                acc = _ext_mul(acc, gamma)
                acc = ((acc[0] + v) % gl, acc[1])
    return out, acc


def cpu_rounds(blocks, is_first, trace):
    """round function for a CPU prover of this AIR: round 0 = the given trace, round 1 = the binding accumulator"""
    def fn(rnd, known):
        if rnd == 0:
            return trace
        cols, total = binding_columns(blocks, is_first, known[:2])
        return cols, list(total)
    return fn


class Sha512Prover:
    """Proves SHA-512 of a batch of messages on one GPU: trace generation (nlx_sha512_trace) straight into HBM,
    then nlx_stark_prove on the device-resident trace.  2^log_blocks compression blocks per proof."""

    def __init__(self, ctx, log_blocks, config=None, segment_nodes=None, step_tag=None):
        from .stark import Stark
        self.ctx = ctx
        self.log_blocks = log_blocks
        if log_blocks < 2:
            raise ValueError("at least four blocks per proof (a block is four trace rows, a STARK at least sixteen)")
        self.step_tag = None if step_tag is None else [int(v) for v in step_tag]   # see stark.step_tag
        air = sha512_air(tagged=self.step_tag is not None)
        if segment_nodes is not None:
            air.segment_nodes = segment_nodes
        self.stark = Stark(air, log_blocks + 2, config)
        self.prover = self.stark.build(ctx)
        self._trace = self._acc = None
        self.last_total = None

    def generate_trace(self, blocks, is_first):
        """Returns (device trace tensor [N_COLS, n] int64, digest words uint64[8])."""
        import torch
        from ._lib import dll
        blocks = np.ascontiguousarray(blocks, dtype=np.uint64)
        is_first = np.ascontiguousarray(is_first, dtype=np.uint8)
        if blocks.shape != (1 << self.log_blocks, 16) or is_first.shape != (1 << self.log_blocks,):
            raise ValueError("expected 2^%d blocks" % self.log_blocks)
        n = ROWS_PER_BLOCK << self.log_blocks
        if self._trace is None:
            self._trace = torch.empty((N_COLS, n), dtype=torch.int64, device="cuda:%d" % self.ctx.device)
            self._acc = torch.empty((2, n), dtype=torch.int64, device=self._trace.device)
        digest = np.zeros(8, dtype=np.uint64)
        self.ctx.check(dll.nlx_sha512_trace(self.ctx.handle, blocks.ctypes.data, is_first.ctypes.data, self.log_blocks,
                                            self._trace.data_ptr(), digest.ctypes.data))
        return self._trace, digest

    def prove(self, messages):
        """Returns (proof bytes, digest words of the last message in the batch)."""
        blocks, first, want = blocks_for_messages(messages, self.log_blocks)
        trace, digest = self.generate_trace(blocks, first)
        assert np.array_equal(digest, want)  # the GPU's chaining value is the real SHA-512 digest
        return self.prove_trace(digest_halves(digest)), digest

    def round1(self, known):
        """The binding accumulator for gamma = known[0:2] (device columns) and its total, the proof's round value."""
        from ._lib import dll
        gamma = np.array([int(known[0]), int(known[1])], dtype=np.uint64)
        total = np.zeros(2, dtype=np.uint64)
        self.ctx.check(dll.nlx_sha512_bind_round(self.ctx.handle, self._trace.data_ptr(), self.log_blocks, gamma.ctypes.data,
                                                 self._acc.data_ptr(), total.ctypes.data))
        self.last_total = (int(total[0]), int(total[1]))
        return self._acc, [int(total[0]), int(total[1])]

    def prove_trace(self, public_inputs):
        """The proof for the trace generate_trace() left on the device (public inputs: the last digest's sixteen halves)."""
        pis = [int(v) for v in public_inputs] + (self.step_tag or [])
        return self.prover.prove_rounds(lambda rnd, known: self._trace if rnd == 0 else self.round1(known), pis)

    def close(self):
        self.prover.close()
        self._trace = self._acc = None
