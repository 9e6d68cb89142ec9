"""SHA-256 compression as an AIR for the STARK prover.

nearx hashes headers, Merkle nodes and approval messages with `curta_sha256` (nearx/src/variables.rs:71-72,
nearx/src/merkle.rs:49, nearx/src/builder.rs:220,316); curta proves those hashes with a STARK whose AIR is
not in the reference (starkyx is un-vendored).  This is an independent AIR for the same function, written
for the register-program VM of include/nlx.h: one row per round, 64 rows per 512-bit block, all
constraints of degree <= 3 so the quotient fits two chunks at rate_bits = 1.

Column layout (bits are LSB first).  Row t of a block holds the working state *before* round t:

    A, B, C, E, F, G   32 bit columns each      D, H  one word column each
    HIN[8]             the block's input chaining value (constant over the block)
    WIN[16]            message-schedule window: WIN[j] = W[t + j]
    W1B, W14B          bits of WIN[1] and WIN[14] (for sigma0 / sigma1)
    NEW_A, NEW_E       the round's outputs;  CA[3], CE[3] carry bits of the two additions
    NEW_W              W[t + 16];            CW[2] carry bits
    CY[8]              carries of HIN + state-after-round-63 (meaningful in the last row of a block)
    IS_FIRST           1 in round 0 of a block that starts a new message (chaining value = IV)

Periodic columns (period 64): K[t] and the last-round selector.  Because the round index is periodic and
a block boundary either chains or resets to the IV, every transition also holds across the wrap from the
last row to the first, so all transition-type constraints are plain all-rows constraints (no `x - g^-1`
filter, which would cost a degree).

Soundness notes: every word that is only ever used inside additions (D, H, HIN, WIN[j], NEW_*) may be off
by a multiple of 2^32 without consequence, because each addition's result is re-derived from boolean bit
columns and the carries are bounded; bit-for-bit equalities between boolean vectors are enforced as one
packed-word equality.  Public inputs: the eight words of the last block's output chaining value (the
digest of the last message).
"""
import hashlib
import struct

import numpy as np

from .stark import Air

A, B, C, E, F, G = 0, 32, 64, 96, 128, 160
D, H = 192, 193
HIN = 194
WIN = 202
W1B = 218
W14B = 250
NEW_A, NEW_E, NEW_W = 282, 283, 284
CA, CE, CW = 285, 288, 291
CY = 293
IS_FIRST = 301
N_COLS = 302

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


def sha256_air():
    air = Air(N_COLS, 8)
    L = air.local
    N = air.next  # noqa: N806
    k_t = air.periodic(K)
    is63 = air.periodic([0] * 63 + [1])
    two32 = 1 << 32

    def pk(row, base, nbits=32):
        return air.pack(base, nbits, next_row=(row is N))

    def xor2(x, y):
        return x + y - 2 * (x * y)

    xor3 = air.xor3

    def weighted(terms):
        acc = terms[0]
        for i in range(1, 32):
            acc = acc + terms[i] * (1 << i)
        return acc

    # 1. booleanity
    for base, cnt in ((A, 192), (W1B, 64), (CA, 8), (CY, 8), (IS_FIRST, 1)):
        air.constraint_boolean(base, cnt)
    # 2. the two decomposed schedule words
    air.constraint(pk(L, W1B) - L(WIN + 1))
    air.constraint(pk(L, W14B) - L(WIN + 14))

    # 3. the round
    a = [L(A + i) for i in range(32)]
    b = [L(B + i) for i in range(32)]
    c = [L(C + i) for i in range(32)]
    e = [L(E + i) for i in range(32)]
    f = [L(F + i) for i in range(32)]
    g = [L(G + i) for i in range(32)]
    sig1 = weighted([xor3(e[(i + 6) % 32], e[(i + 11) % 32], e[(i + 25) % 32]) for i in range(32)])
    ch = weighted([air.ch(e[i], f[i], g[i]) for i in range(32)])
    sig0 = weighted([xor3(a[(i + 2) % 32], a[(i + 13) % 32], a[(i + 22) % 32]) for i in range(32)])
    maj = weighted([air.maj(a[i], b[i], c[i]) for i in range(32)])
    t1 = L(H) + sig1 + ch + k_t + L(WIN)
    air.constraint(L(NEW_A) + pk(L, CA, 3) * two32 - (t1 + sig0 + maj))
    air.constraint(L(NEW_E) + pk(L, CE, 3) * two32 - (L(D) + t1))

    # 4. message schedule: W[t+16] = s1(W[t+14]) + W[t+9] + s0(W[t+1]) + W[t]
    w1 = [L(W1B + i) for i in range(32)]
    w14 = [L(W14B + i) for i in range(32)]
    s0 = weighted([xor3(w1[(i + 7) % 32], w1[(i + 18) % 32], w1[i + 3]) if i + 3 < 32
                   else xor2(w1[(i + 7) % 32], w1[(i + 18) % 32]) for i in range(32)])
    s1 = weighted([xor3(w14[(i + 17) % 32], w14[(i + 19) % 32], w14[i + 10]) if i + 10 < 32
                   else xor2(w14[(i + 17) % 32], w14[(i + 19) % 32]) for i in range(32)])
    air.constraint(L(NEW_W) + pk(L, CW, 2) * two32 - (s1 + L(WIN + 9) + s0 + L(WIN)))

    # 5. inside a block (not after round 63): the state and the window shift
    nb = 1 - is63
    pa, pb, pc_, pe, pf, pg = (pk(L, X) for X in (A, B, C, E, F, G))
    npa, npb, npc, npe, npf, npg = (pk(N, X) for X in (A, B, C, E, F, G))
    air.constraint(nb * (npb - pa))
    air.constraint(nb * (npc - pb))
    air.constraint(nb * (N(D) - pc_))
    air.constraint(nb * (npf - pe))
    air.constraint(nb * (npg - pf))
    air.constraint(nb * (N(H) - pg))
    air.constraint(nb * (npa - L(NEW_A)))
    air.constraint(nb * (npe - L(NEW_E)))
    for k in range(8):
        air.constraint(nb * (N(HIN + k) - L(HIN + k)))
    for j in range(15):
        air.constraint(nb * (N(WIN + j) - L(WIN + j + 1)))
    air.constraint(nb * (N(WIN + 15) - L(NEW_W)))

    # 6. block boundary (after round 63): next state = IV if the next block starts a message, else HIN + state
    out = [L(NEW_A), pa, pb, pc_, L(NEW_E), pe, pf, pg]
    nxt = [npa, npb, npc, N(D), npe, npf, npg, N(H)]
    ho = [L(HIN + k) + out[k] - L(CY + k) * two32 for k in range(8)]
    for k in range(8):
        air.constraint(is63 * (nxt[k] - ho[k] - N(IS_FIRST) * (IV[k] - ho[k])))
        air.constraint(is63 * (N(HIN + k) - nxt[k]))

    # 7. first row starts a message; 8. the last row's output chaining value is the public digest
    cur = [pa, pb, pc_, L(D), pe, pf, pg, L(H)]
    air.constraint_first_row(L(IS_FIRST) - 1)
    for k in range(8):
        air.constraint_first_row(cur[k] - IV[k])
    for k in range(8):
        air.constraint_last_row(ho[k] - air.public(k))
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


def reference_trace(blocks, is_first):
    """(N_COLS, 64 * n_blocks) trace, plain Python.  Mirrors the column semantics documented above."""
    nb = len(blocks)
    n = 64 * nb
    t = np.zeros((N_COLS, n), dtype=np.uint64)

    def put_bits(base, row, v, cnt=32):
        for i in range(cnt):
            t[base + i, row] = (v >> i) & 1

    h = list(IV)
    for bi in range(nb):
        if is_first[bi]:
            h = list(IV)
        w = [int(x) for x in blocks[bi]]
        for i in range(16, 80):
            x1, x14 = w[i - 15], w[i - 2]
            s0 = _rotr(x1, 7) ^ _rotr(x1, 18) ^ (x1 >> 3)
            s1 = _rotr(x14, 17) ^ _rotr(x14, 19) ^ (x14 >> 10)
            w.append((w[i - 16] + s0 + w[i - 7] + s1) & 0xFFFFFFFF)
        a, b, c, d, e, f, g, hh = h
        for r in range(64):
            row = 64 * bi + r
            for base, v in ((A, a), (B, b), (C, c), (E, e), (F, f), (G, g)):
                put_bits(base, row, v)
            t[D, row], t[H, row] = d, hh
            for k in range(8):
                t[HIN + k, row] = h[k]
            for j in range(16):
                t[WIN + j, row] = w[r + j]
            put_bits(W1B, row, w[r + 1])
            put_bits(W14B, row, w[r + 14])
            s1 = _rotr(e, 6) ^ _rotr(e, 11) ^ _rotr(e, 25)
            ch = (e & f) ^ (~e & g & 0xFFFFFFFF)
            s0 = _rotr(a, 2) ^ _rotr(a, 13) ^ _rotr(a, 22)
            mj = (a & b) ^ (a & c) ^ (b & c)
            t1 = hh + s1 + ch + K[r] + w[r]
            sa, se = t1 + s0 + mj, d + t1
            na, ne = sa & 0xFFFFFFFF, se & 0xFFFFFFFF
            t[NEW_A, row], t[NEW_E, row] = na, ne
            put_bits(CA, row, sa >> 32, 3)
            put_bits(CE, row, se >> 32, 3)
            x1, x14 = w[r + 1], w[r + 14]
            sw = (_rotr(x14, 17) ^ _rotr(x14, 19) ^ (x14 >> 10)) + w[r + 9] + (_rotr(x1, 7) ^ _rotr(x1, 18) ^ (x1 >> 3)) + w[r]
            t[NEW_W, row] = sw & 0xFFFFFFFF
            put_bits(CW, row, sw >> 32, 2)
            if r == 0:
                t[IS_FIRST, row] = int(is_first[bi])
            if r == 63:
                out = [na, a, b, c, ne, e, f, g]
                for k in range(8):
                    t[CY + k, row] = (h[k] + out[k]) >> 32
                h = [(h[k] + out[k]) & 0xFFFFFFFF for k in range(8)]
            a, b, c, d, e, f, g, hh = na, a, b, c, ne, e, f, g
    return t, np.array(h, dtype=np.uint64)


class Sha256Prover:
    """Proves SHA-256 of a batch of messages on one GPU: trace generation (nlx_sha256_trace) straight into
    HBM, then nlx_stark_prove on the device-resident trace.  2^log_blocks compression blocks per proof."""

    def __init__(self, ctx, log_blocks, config=None):
        from .stark import Stark
        self.ctx = ctx
        self.log_blocks = log_blocks
        self.stark = Stark(sha256_air(), log_blocks + 6, config)
        self.prover = self.stark.build(ctx)
        self._trace = None

    def generate_trace(self, blocks, is_first):
        """Returns (device trace tensor [N_COLS, n] int64, digest words uint64[8])."""
        import torch
        from ._lib import dll
        blocks = np.ascontiguousarray(blocks, dtype=np.uint32)
        is_first = np.ascontiguousarray(is_first, dtype=np.uint8)
        if blocks.shape != (1 << self.log_blocks, 16) or is_first.shape != (1 << self.log_blocks,):
            raise ValueError("expected 2^%d blocks" % self.log_blocks)
        n = 64 << self.log_blocks
        if self._trace is None:
            self._trace = torch.empty((N_COLS, n), dtype=torch.int64, device="cuda:%d" % self.ctx.device)
        digest = np.zeros(8, dtype=np.uint64)
        self.ctx.check(dll.nlx_sha256_trace(self.ctx.handle, blocks.ctypes.data, is_first.ctypes.data, self.log_blocks,
                                            self._trace.data_ptr(), digest.ctypes.data))
        return self._trace, digest

    def prove(self, messages):
        """Returns (proof bytes, digest words of the last message in the batch)."""
        blocks, first, want = blocks_for_messages(messages, self.log_blocks)
        trace, digest = self.generate_trace(blocks, first)
        assert np.array_equal(digest, want)  # the GPU's chaining value is the real SHA-256 digest
        return self.prover.prove(trace, digest), digest

    def close(self):
        self.prover.close()
        self._trace = None
