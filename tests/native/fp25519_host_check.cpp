// Host build of the device witness code (csrc/fp25519.hpp is __host__ __device__): reads "a b" pairs of 64-hex-digit
// numbers, one pair of products per line ("a b" or "a b c d"), prints c, q and the 15 carries.  Used by
// tests/test_fp25519.py to check the arithmetic against Python's big integers without a GPU.
#include <cstdio>
#include <cstring>
#include "fp25519.hpp"

static void parse(const char* hex, uint32_t limbs[16]) {
    for (int i = 0; i < 16; i++) {
        unsigned v = 0;
        sscanf(hex + 4 * (15 - i), "%4x", &v);
        limbs[i] = v;
    }
}

int main() {
    char buf[4][80];
    char line[400];
    while (fgets(line, sizeof line, stdin)) {
        const int k = sscanf(line, "%64s %64s %64s %64s", buf[0], buf[1], buf[2], buf[3]);
        if (k != 2 && k != 4) continue;
        uint64_t prod[32];
        memset(prod, 0, sizeof prod);
        uint32_t a[16], b[16];
        for (int t = 0; t < k; t += 2) {
            parse(buf[t], a);
            parse(buf[t + 1], b);
            nlx::fp::mul_acc(prod, a, b);
        }
        nlx::fp::Unit u;
        nlx::fp::finish(prod, u);
        for (int i = 15; i >= 0; i--) printf("%04x", u.c[i]);
        printf(" ");
        for (int i = 16; i >= 0; i--) printf("%04x", u.q[i]);
        for (int m = 0; m < 15; m++) printf(" %u", u.carry[m]);
        printf("\n");
    }
    return 0;
}
