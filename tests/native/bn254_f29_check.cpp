// Host build of the 29-bit-limb field arithmetic of the BN254 kernels (csrc/bn254_f29.hpp; base field q: the MSM, scalar field
// r: the NTT): reads "modulus op A B" lines (hexadecimal integers below 2^261), prints the result as a hexadecimal integer
// and whether its limbs 0..7 are below 2^29.
// tests/test_bn254_model.py checks every answer, and the bounds the header states, with Python integers.
#include <cstdio>
#include <cstring>
#include <string>
#include "bn254_f29.hpp"

using namespace nlx::f29;

static Fe parse(const char* hex) {   // integer -> limbs by bit slicing
    unsigned char bits[272] = {0};
    const size_t len = strlen(hex);
    for (size_t i = 0; i < len; i++) {
        const char ch = hex[len - 1 - i];
        const int d = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10;
        for (int b = 0; b < 4; b++)
            if (4 * i + b < 272) bits[4 * i + b] = (d >> b) & 1;
    }
    Fe r = zero();
    for (int i = 0; i < NL; i++)
        for (int b = 0; b < (i == NL - 1 ? 32 : LB); b++)
            if (LB * i + b < 272 && bits[LB * i + b]) r.v[i] |= 1u << b;
    return r;
}
static void print(const Fe& a) {   // sum v[i] 2^(29 i) as hex (limb 8 may carry more than 29 bits)
    unsigned char bits[300] = {0};
    unsigned carry_check = 1;
    for (int i = 0; i < NL; i++) {
        if (i < NL - 1 && a.v[i] > MASK) carry_check = 0;
        for (int b = 0; b < 32; b++) {
            if (!((a.v[i] >> b) & 1)) continue;
            int pos = LB * i + b;   // add 2^pos
            while (bits[pos]) bits[pos++] = 0;
            bits[pos] = 1;
        }
    }
    std::string s;
    for (int nib = 74; nib >= 0; nib--) {
        int d = 0;
        for (int b = 0; b < 4; b++) d |= bits[4 * nib + b] << b;
        s += "0123456789abcdef"[d];
    }
    printf("%s %u\n", s.c_str(), carry_check);
}

template <class M>
static int run(const char* op, const Fe& x, const Fe& y) {
    if (!strcmp(op, "mul")) print(mul<M>(x, y));
    else if (!strcmp(op, "add")) print(add(x, y));
    else if (!strcmp(op, "sub4")) print(sub<4, M>(x, y));
    else if (!strcmp(op, "sub8")) print(sub<8, M>(x, y));
    else if (!strcmp(op, "tighten")) print(tighten<M>(x));
    else if (!strcmp(op, "canonical")) print(canonical<M>(x));
    else if (!strcmp(op, "iszero")) printf("%d 1\n", is_zero_mod<M>(x) ? 1 : 0);
    else if (!strcmp(op, "frommont") || !strcmp(op, "words")) {
        uint32_t w[8];
        to_words256(x, w);   // the operand is below 2^256
        if (!strcmp(op, "words")) print(from_words256(w));   // slicing there and back
        else print(from_mont256<M>(w));
    } else if (!strcmp(op, "tocanon")) {
        uint32_t w[8];
        to_canonical256<M>(x, w);
        std::string s;
        char t[16];
        for (int k = 7; k >= 0; k--) { snprintf(t, sizeof t, "%08x", w[k]); s += t; }
        printf("%s 1\n", s.c_str());
    } else return 2;
    return 0;
}

int main() {   // lines: modulus ("q" | "r") op A B
    char mod[8], op[32], a[128], b[128];
    while (scanf("%7s %31s %127s %127s", mod, op, a, b) == 4) {
        const Fe x = parse(a), y = parse(b);
        const int rc = mod[0] == 'q' ? run<QMod>(op, x, y) : run<RMod>(op, x, y);
        if (rc) return rc;
    }
    return 0;
}
