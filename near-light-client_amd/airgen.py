"""Straight-line HIP evaluators for FIXED AIR programs (row a12: starkyx / curta's AIR constraint evaluation).

`k_air_quotient` (csrc/stark.hip) INTERPRETS a register program: ~71 machine instructions per program word where a field
multiplication is ~20 and an addition 5 - decode on the scalar unit, the register file in LDS, a jump per word.  The AIRs of
a Sync step are fixed when the library is built (the reference fixes them the same way: the curta gadgets behind
/root/reference/nearx/src/builder.rs:152,220,316 are compiled into the circuit), so this module turns the SAME program words
into straight-line HIP, one function per program segment, registers as local variables:

    python near-light-client_amd/airgen.py          (run by build.py when an AIR source is newer than the generated files)

writes csrc/airgen/air_<name>.hip (+ registry.hip).  A generated kernel is found at nlx_stark_build by the FNV-1a hash of the
canonicalised program words, takes the launch parameters of the interpreter (AirParams, csrc/air_vm.hpp) over the same grid
(points x segments) and calls the same field functions in the same order, so its outputs equal the interpreter's word for word;
the interpreter stays the generic path (any other program) and the parity reference (NLX_AIR_VM=1 forces it;
tests/test_gpu_airgen.py compares proof bytes both ways).  The module runs WITHOUT the built library: it stubs `_lib` and only
uses the pure-Python assembler (stark.Air.compile)."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "csrc", "airgen")
P = 0xFFFFFFFF00000001
MASK64 = (1 << 64) - 1

(LOCAL, NEXT, PUBLIC, CONST, ADD, SUB, MUL, EMIT_TRANSITION, EMIT_FIRST, EMIT_LAST, EMIT, PERIODIC, PACK_LOCAL, PACK_NEXT, EMIT_BOOL,
 LOADV, XOR3, CH, MAJ, SEGMENT, EMIT_LOGUP, MAC) = range(22)


def program_hash(words):
    """FNV-1a over the words' bytes, little-endian (csrc/air_vm.hpp airgen_program_hash)"""
    h = 0xcbf29ce484222325
    for b in np.ascontiguousarray(words, dtype="<u8").tobytes():
        h = ((h ^ b) * 0x100000001b3) & MASK64
    return h


def canonical_words(words):
    """the library keeps a copy of the program with the CONST immediates reduced mod p (nlx_stark_build); hash THAT"""
    out, i = [int(w) for w in words], 0
    while i < len(out):
        if (out[i] & 0xFF) == CONST:
            out[i + 1] %= P
            i += 2
        else:
            i += 1
    return np.array(out, dtype=np.uint64)


def _segments(words):
    """[(first word, end word)] of the program's segments, as nlx_stark_build cuts them (the SEGMENT word belongs to neither)"""
    segs, lo, i = [], 0, 0
    while i < len(words):
        op = int(words[i]) & 0xFF
        if op == SEGMENT:
            segs.append((lo, i))
            lo = i + 1
        i += 2 if op == CONST else 1
    segs.append((lo, len(words)))
    return segs


def _emit_constraint(lines, expr):
    lines.append("    c = %s;" % expr)
    lines.append("    acc0 = gl::add(gl::mul(acc0, g.a0), c); acc1 = gl::add(gl::mul(acc1, g.a1), c);")


SCHED_EVERY = 4


def _segment_body(words, lo, hi):
    """C++ statements for words[lo:hi]; every statement mirrors the interpreter's case for its opcode (csrc/stark.hip).
    hipcc schedules a long straight-line block for instruction-level parallelism and keeps everything it hoisted alive: the
    multiplier segments of the Ed25519 AIR (43 live values in the assembler's plan = 86 registers) came out at 214 - 241
    registers, two waves per SIMD.  A barrier after every SCHED_EVERY register-writing words keeps the order the assembler
    planned (its plan bounds the live values); the wave-level parallelism comes from the other waves."""
    L, i, since = [], lo, 0
    R = lambda k: "r%d" % k
    while i < hi:
        if since >= SCHED_EVERY:
            L.append("    __builtin_amdgcn_sched_barrier(0);")
            since = 0
        if (int(words[i]) & 0xFF) in (LOCAL, NEXT, PUBLIC, CONST, ADD, SUB, MUL, PERIODIC, PACK_LOCAL, PACK_NEXT, XOR3, CH, MAJ, MAC):
            since += 1
        w = int(words[i])
        op, dst, a, b, sh = w & 0xFF, (w >> 8) & 0xFFFF, (w >> 24) & 0xFFFF, (w >> 40) & 0xFFFF, (w >> 56) & 0x3F
        i += 1
        if op == LOCAL:
            L.append("    %s = COL(%d, g.row);" % (R(dst), a))
        elif op == NEXT:
            L.append("    %s = COL(%d, g.row_next);" % (R(dst), a))
        elif op == PUBLIC:
            L.append("    %s = PIS(%d);" % (R(dst), a))
        elif op == PERIODIC:
            L.append("    %s = g.per[(size_t)%d * g.per_stride + g.per_off];" % (R(dst), a))
        elif op == CONST:
            L.append("    %s = 0x%016xull;" % (R(dst), int(words[i]) % P))
            i += 1
        elif op == ADD:
            L.append("    %s = gl::add(%s, mul_pow2(%s, %d));" % (R(dst), R(a), R(b), sh) if sh else "    %s = gl::add(%s, %s);" % (R(dst), R(a), R(b)))
        elif op == SUB:
            L.append("    %s = gl::sub(%s, mul_pow2(%s, %d));" % (R(dst), R(a), R(b), sh) if sh else "    %s = gl::sub(%s, %s);" % (R(dst), R(a), R(b)))
        elif op == MUL:
            L.append("    %s = gl::mul(%s, %s);" % (R(dst), R(a), R(b)))
        elif op == MAC:
            L.append("    %s = gl::add(%s, gl::mul(%s, %s));" % (R(dst), R(sh), R(a), R(b)))
        elif op == CH:
            L.append("    %s = gl::add(%s, gl::mul(%s, gl::sub(%s, %s)));" % (R(dst), R(sh), R(a), R(b), R(sh)))
        elif op in (XOR3, MAJ):
            L.append("    { const uint64_t xy = gl::mul(%s, %s), sx = gl::sub(gl::add(%s, %s), gl::add(xy, xy));" % (R(a), R(b), R(a), R(b)))
            if op == XOR3:
                L.append("      const uint64_t sz = gl::mul(sx, %s); %s = gl::sub(gl::add(sx, %s), gl::add(sz, sz)); }" % (R(sh), R(dst), R(sh)))
            else:
                L.append("      %s = gl::add(xy, gl::mul(%s, sx)); }" % (R(dst), R(sh)))
        elif op in (PACK_LOCAL, PACK_NEXT):
            row = "g.row" if op == PACK_LOCAL else "g.row_next"
            L.append("    { uint64_t v[%d];" % b)
            for k in range(b):
                L.append("      v[%d] = COL(%d, %s);" % (k, a + k, row))
            L.append("      uint64_t acc = 0;")
            for k in range(b):   # gl::add(0, v) = v for a canonical v: the interpreter's first step
                L.append("      acc = gl::add(acc, mul_pow2(v[%d], %d));" % (k, k))
            L.append("      %s = acc; }" % R(dst))
        elif op == EMIT_BOOL:
            cnt = b if b else 1
            for i0 in range(0, cnt, 8):
                m = min(8, cnt - i0)
                L.append("    { uint64_t v[%d];" % m)
                for k in range(m):
                    L.append("      v[%d] = COL(%d, g.row);" % (k, a + i0 + k))
                for k in range(m):
                    L.append("      c = gl::mul(v[%d], gl::sub(v[%d], 1)); acc0 = gl::add(gl::mul(acc0, g.a0), c); acc1 = gl::add(gl::mul(acc1, g.a1), c);" % (k, k))
                L.append("    }")
        elif op == EMIT_LOGUP:
            L.append("    { const uint64_t al0 = PIS(g.n_pis + %d), al1 = PIS(g.n_pis + %d);" % (sh, sh + 1))
            L.append("      const uint64_t h0 = COL(%d, g.row), h1 = COL(%d, g.row), v1 = COL(%d, g.row);" % (b, b + 1, a))
            L.append("      uint64_t c0, c1;")
            if dst == 0xFFFF:
                L.append("      const uint64_t d0 = gl::add(al0, v1);")
                L.append("      c0 = gl::sub(gl::add(gl::mul(h0, d0), mul_pow2(gl::mul(h1, al1), 3)), gl::add(gl::mul(h1, al1), 1));")
                L.append("      c1 = gl::add(gl::mul(h0, al1), gl::mul(h1, d0));")
            else:
                L.append("      const uint64_t v2 = COL(%d, g.row);" % dst)
                L.append("      const uint64_t s2 = gl::add(gl::add(al0, al0), gl::add(v1, v2));")
                L.append("      const uint64_t a1sq = gl::mul(al1, al1);")
                L.append("      const uint64_t u0 = gl::add(gl::mul(gl::add(al0, v1), gl::add(al0, v2)), gl::sub(mul_pow2(a1sq, 3), a1sq));")
                L.append("      const uint64_t u1 = gl::mul(al1, s2);")
                L.append("      const uint64_t hu = gl::mul(h1, u1);")
                L.append("      c0 = gl::sub(gl::add(gl::mul(h0, u0), gl::sub(mul_pow2(hu, 3), hu)), s2);")
                L.append("      c1 = gl::sub(gl::add(gl::mul(h0, u1), gl::mul(h1, u0)), gl::add(al1, al1));")
            L.append("      acc0 = gl::add(gl::mul(gl::add(gl::mul(acc0, g.a0), c0), g.a0), c1);")
            L.append("      acc1 = gl::add(gl::mul(gl::add(gl::mul(acc1, g.a1), c0), g.a1), c1); }")
        elif op == LOADV:
            # the interpreter's hint "the next words are independent loads" marks where the assembler wants a batch of loads in
            # flight: a scheduling barrier there keeps hipcc from hoisting EVERY load of the segment to its top (354 registers and
            # scratch without it) - live values stay what the assembler planned, the batch's loads still issue together
            L.append("    __builtin_amdgcn_sched_barrier(0);")
        elif op == EMIT_TRANSITION:
            _emit_constraint(L, "gl::mul(%s, g.z_last)" % R(a))
        elif op == EMIT_FIRST:
            _emit_constraint(L, "gl::mul(%s, g.l_first)" % R(a))
        elif op == EMIT_LAST:
            _emit_constraint(L, "gl::mul(%s, g.l_last)" % R(a))
        elif op == EMIT:
            _emit_constraint(L, R(a))
        else:
            raise ValueError("unknown opcode %d at word %d" % (op, i - 1))
    return L


def _registers_of(words, lo, hi):
    regs, i = set(), lo
    while i < hi:
        w = int(words[i])
        op = w & 0xFF
        if op in (LOCAL, NEXT, PUBLIC, CONST, ADD, SUB, MUL, PERIODIC, PACK_LOCAL, PACK_NEXT, XOR3, CH, MAJ, MAC):
            regs.add((w >> 8) & 0xFFFF)
        i += 2 if op == CONST else 1
    return sorted(regs)


PART_WORDS = 10000   # a translation unit holds the segments of about this many program words (build time ~4 s per 1 000 words)


def generate_sources(name, words):
    """{file name: text}: the segments of `words` as NON-inlined device functions (hipcc needs ~4 s per 1 000 words that way; one
    kernel with every segment inlined did not finish in 40 minutes), in parts of ~PART_WORDS words - a part is a translation unit
    with its own kernel over the whole (points x segments) grid whose blocks leave at once when the segment belongs to another part -
    and the registry entry."""
    words = canonical_words(words)
    h = program_hash(words)
    segs = _segments(words)
    parts, cur, cur_words = [], [], 0
    for lo, hi in segs:
        if cur and cur_words + (hi - lo) > PART_WORDS:
            parts.append(cur)
            cur, cur_words = [], 0
        cur.append((lo, hi))
        cur_words += hi - lo
    parts.append(cur)
    files = {}
    for pi, part in enumerate(parts):
        ns = "airgen_%s_p%d" % (name, pi)
        out = ["// GENERATED by near-light-client_amd/airgen.py from the program words of AIR '%s' (%d words, %d segments, hash 0x%016x):" % (name, len(words), len(segs), h),
               "// part %d of %d, segments starting at words %s." % (pi + 1, len(parts), ", ".join(str(lo) for lo, _ in part)),
               "// Do not edit: the interpreter k_air_quotient (stark.hip) runs the same words and is the parity reference.",
               '#include "../air_vm.hpp"', "", "namespace nlx {", "namespace %s {" % ns, "",
               "// what a segment needs of the point: passed BY VALUE (registers); the wave-uniform members are made scalar again inside",
               "// the function (readfirstlane), so a column's base address is a scalar load and its element ONE vector load",
               "// The column-pointer table and the public inputs are the same for every lane and constant for the launch: read through the",
               "// constant address space they are scalar loads (s_load), and a column's value is then ONE global load with the pointer in",
               "// SGPRs and the row's byte offset in a VGPR.  Through plain pointers hipcc made each a per-lane flat load with a wait",
               "// between the pointer and the value (two memory latencies per column, two more VGPRs per load in flight).",
               "// row / row_next are BYTE offsets (LDE rows * 8 < 2^32: the host selects these kernels for LDEs of <= 2^28 rows only).",
               "typedef const uint64_t __attribute__((address_space(4)))* airgen_kptr;",
               "typedef const char __attribute__((address_space(1)))* airgen_gptr;",
               "#define COL(c, off) (*(const uint64_t __attribute__((address_space(1)))*)((airgen_gptr)((airgen_kptr)g.cols)[c] + (off)))",
               "#define PIS(i) (((airgen_kptr)g.pis)[i])",
               "struct G {", "    uint64_t cols, pis; const uint64_t* per;",
               "    size_t per_stride, per_off; uint64_t a0, a1, z_last, l_first, l_last; uint32_t row, row_next, n_pis;", "};",
               "struct Acc { uint64_t a0, a1; };",
               "template <typename T> __device__ __forceinline__ T* uni(T* p) {",
               "    const uint64_t v = (uint64_t)p;",
               "    return (T*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v));",
               "}",
               "__device__ __forceinline__ uint64_t uni(uint64_t v) {",
               "    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);",
               "}", ""]
        for lo, hi in part:
            regs = _registers_of(words, lo, hi)
            out.append("static __device__ __noinline__ Acc seg_%d(G gv) {" % lo)
            out.append("    G g = gv;")
            out.append("    g.cols = uni(gv.cols); g.pis = uni(gv.pis); g.per = uni(gv.per); g.per_stride = uni((uint64_t)gv.per_stride);")
            out.append("    g.a0 = uni(gv.a0); g.a1 = uni(gv.a1); g.n_pis = (uint32_t)__builtin_amdgcn_readfirstlane((int)gv.n_pis);")
            out.append("    uint64_t c, acc0 = 0, acc1 = 0;")
            if regs:
                out.append("    uint64_t %s;" % ", ".join("r%d" % r for r in regs))
            out += _segment_body(words, lo, hi)
            out.append("    return Acc{acc0, acc1};")
            out.append("}")
            out.append("")
        out += [
            "// the interpreter's prologue and epilogue (csrc/stark.hip k_air_quotient), the program between them as a switch on the segment",
            "__global__ __launch_bounds__(AIRGEN_BLOCK) void kernel(AirParams p) {",
            "    const uint32_t sg = blockIdx.y;",
            "    const uint32_t first = p.n_seg > 1 ? p.seg[2 * sg] : 0u;",
            "    if (%s) return;   // another part's segment" % " && ".join("first != %du" % lo for lo, _ in part),
            "    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;",
            "    const unsigned log_Q = p.log_n + p.qdb;",
            "    if (pos >> log_Q) return;",
            "    const size_t n = (size_t)1 << p.log_n;",
            "    const uint32_t rq = (uint32_t)(pos >> p.log_n), k = (uint32_t)(pos & (n - 1));",
            "    const uint32_t r = rq << (p.rate_bits - p.qdb);",
            "    const size_t qrow_next = ((size_t)rq << p.log_n) + ((k + 1) & (n - 1));",
            "    const uint64_t x = gl::mul(p.coset_base[rq], root_pow_(p.w_n_table, k, (uint32_t)(n >> 1)));",
            "    const uint64_t zh = gl::inv(p.zh_inv[rq]);",
            "    G g;",
            "    g.cols = (uint64_t)p.cols; g.pis = (uint64_t)p.pis; g.per = p.periodic; g.n_pis = p.n_pis;",
            "    g.row = (uint32_t)((((size_t)r << p.log_n) + k) << 3); g.row_next = (uint32_t)((((size_t)r << p.log_n) + ((k + 1) & (n - 1))) << 3);",
            "    g.per_stride = (size_t)1 << (p.qdb + p.period_bits);",
            "    g.per_off = ((size_t)rq << p.period_bits) + (k & ((1u << p.period_bits) - 1));",
            "    g.a0 = p.alphas[0]; g.a1 = p.alphas[1];",
            "    g.z_last = gl::sub(x, p.g_inv); g.l_first = gl::mul(zh, p.l_inv[pos]); g.l_last = gl::mul(zh, p.l_inv[qrow_next]);",
            "    Acc acc{0, 0};",
            "    switch (first) {",
        ]
        for lo, _ in part:
            out.append("        case %d: acc = seg_%d(g); break;" % (lo, lo))
        out += [
            "        default: break;",
            "    }",
            "    const size_t Q = (size_t)1 << log_Q;",
            "    if (p.n_seg > 1) {",
            "        uint64_t* dst = p.part + (((size_t)sg * p.nc) << log_Q) + pos;",
            "        dst[0] = gl::mul(acc.a0, p.seg_mul[2 * sg]);",
            "        dst[Q] = gl::mul(acc.a1, p.seg_mul[2 * sg + 1]);",
            "        return;",
            "    }",
            "    const uint64_t zi = p.zh_inv[rq];",
            "    p.out[pos] = gl::mul(acc.a0, zi);",
            "    p.out[Q + pos] = gl::mul(acc.a1, zi);",
            "}",
            "",
            "}  // namespace %s" % ns,
            "void airgen_launch_%s_p%d(hipStream_t st, unsigned tiles, unsigned n_segments, const AirParams& p) {" % (name, pi),
            "    hipLaunchKernelGGL(%s::kernel, dim3(tiles, n_segments), dim3(AIRGEN_BLOCK), 0, st, p);" % ns,
            "}",
            "}  // namespace nlx", ""]
        files["air_%s_p%d.hip" % (name, pi)] = "\n".join(out)
    main = ["// GENERATED by near-light-client_amd/airgen.py: AIR '%s' (%d words, %d segments in %d parts, hash 0x%016x)." % (name, len(words), len(segs), len(parts), h),
            '#include "../air_vm.hpp"', "", "namespace nlx {"]
    main += ["void airgen_launch_%s_p%d(hipStream_t st, unsigned tiles, unsigned n_segments, const AirParams& p);" % (name, pi) for pi in range(len(parts))]
    main += ["static void airgen_launch_%s(hipStream_t st, unsigned tiles, unsigned n_segments, const AirParams& p) {" % name]
    main += ["    airgen_launch_%s_p%d(st, tiles, n_segments, p);" % (name, pi) for pi in range(len(parts))]
    main += ["}",
             "#if !defined(__HIP_DEVICE_COMPILE__)   // a host object (it holds a host function's address): not for the device pass",
             'extern const AirGenEntry airgen_entry_%s = {0x%016xull, %du, "%s", airgen_launch_%s};' % (name, h, len(words), name, name),
             "#endif",
             "}  // namespace nlx", ""]
    files["air_%s.hip" % name] = "\n".join(main)
    return files


def registry_source(names):
    out = ["// GENERATED by near-light-client_amd/airgen.py: the table nlx_stark_build searches (csrc/air_vm.hpp airgen_find).",
           '#include "../air_vm.hpp"', "", "namespace nlx {"]
    out += ["extern const AirGenEntry airgen_entry_%s;" % n for n in names]
    out += ["", "#if !defined(__HIP_DEVICE_COMPILE__)", "const AirGenEntry* airgen_find(uint64_t program_hash, uint32_t n_words) {",
            "    static const AirGenEntry* const table[] = {%s};" % ", ".join(["&airgen_entry_%s" % n for n in names] + ["nullptr"]),
            "    for (const AirGenEntry* const* e = table; *e; e++)",
            "        if ((*e)->program_hash == program_hash && (*e)->n_words == n_words) return *e;",
            "    return nullptr;", "}", "#endif", "}  // namespace nlx", ""]
    return "\n".join(out)


def _load_air_modules():
    """the AIR definitions, importable without the built library: `_lib` (the ctypes bindings) is stubbed - compiling a program
    is pure Python"""
    if "nlx_amd" in sys.modules and hasattr(sys.modules["nlx_amd"], "stark"):
        return sys.modules["nlx_amd"]
    import importlib
    pkg = types.ModuleType("nlx_amd")
    pkg.__path__ = [HERE]
    sys.modules["nlx_amd"] = pkg
    lib = types.ModuleType("nlx_amd._lib")

    class NlxError(RuntimeError):
        pass
    lib.dll, lib.synth_dll, lib.ptr, lib.NlxError = None, None, (lambda a: None), NlxError
    sys.modules["nlx_amd._lib"] = lib
    for m in ("stark", "logup", "fp25519", "sha256_air", "sha512_air", "ed25519_air"):
        setattr(pkg, m, importlib.import_module("nlx_amd." + m))
    return pkg


# the fixed programs: the three AIRs of a Sync step under its step tag (bench.py sync_step_setup: the headline), the untagged
# SHA-256 AIR (the map jobs' STARKs of the Verify job, the SHA tests) and the untagged Ed25519 AIR at 2^8 slots (one range-table column)
def fixed_programs():
    pkg = _load_air_modules()
    progs = [("sha256_tagged", pkg.sha256_air.sha256_air(tagged=True).compile()),
             ("sha256", pkg.sha256_air.sha256_air(tagged=False).compile()),
             ("sha512_tagged", pkg.sha512_air.sha512_air(tagged=True).compile()),
             ("ed25519_2p7_tagged", pkg.ed25519_air.Ed25519Stark(7, tagged=True).stark.program)]
    return progs


def sources():
    progs = fixed_programs()
    files = {}
    for name, words in progs:
        files.update(generate_sources(name, words))
    files["registry.hip"] = registry_source([name for name, _ in progs])
    return files


def write_all(out_dir=OUT_DIR):
    os.makedirs(out_dir, exist_ok=True)
    files = sources()
    for stale in os.listdir(out_dir):
        if stale.endswith(".hip") and stale not in files:
            os.remove(os.path.join(out_dir, stale))
    for fn, text in files.items():
        path = os.path.join(out_dir, fn)
        if not os.path.exists(path) or open(path).read() != text:
            with open(path, "w") as f:
                f.write(text)
    return sorted(files)


if __name__ == "__main__":
    for fn in write_all():
        print(os.path.join(OUT_DIR, fn))
