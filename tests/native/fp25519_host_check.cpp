// Host build of the device witness code (csrc/fp25519.hpp compiles as plain C++): each input line is
//     [f C] (+|-) A B [(+|-) A B ...]
// with 64-hex-digit numbers: the signed products of one unit and optionally the representative C to use as result.
// Prints c, the committed quotient and the 15 carries.  tests/test_fp25519.py compares them with Python's integers.
#include <cstdio>
#include <cstring>
#include "fp25519.hpp"

static void parse(const char* hex, int32_t limbs[16]) {
    for (int i = 0; i < 16; i++) {
        unsigned v = 0;
        sscanf(hex + 4 * (15 - i), "%4x", &v);
        limbs[i] = (int32_t)v;
    }
}

int main() {
    char line[1200];
    while (fgets(line, sizeof line, stdin)) {
        int64_t prod[32];
        memset(prod, 0, sizeof prod);
        uint32_t cfix[16];
        bool fixed = false;
        char* tok = strtok(line, " \n");
        while (tok) {
            if (tok[0] == 'f') {
                int32_t tmp[16];
                parse(strtok(nullptr, " \n"), tmp);
                for (int i = 0; i < 16; i++) cfix[i] = (uint32_t)tmp[i];
                fixed = true;
            } else {
                const int sign = tok[0] == '-' ? -1 : 1;
                int32_t a[16], b[16];
                parse(strtok(nullptr, " \n"), a);
                parse(strtok(nullptr, " \n"), b);
                nlx::fp::mul_acc(prod, a, b, sign);
            }
            tok = strtok(nullptr, " \n");
        }
        nlx::fp::Unit u;
        nlx::fp::finish(prod, u, fixed ? cfix : nullptr);
        for (int i = 15; i >= 0; i--) printf("%04x", u.c[i]);
        printf(" ");
        for (int i = 16; i >= 0; i--) printf("%04x", u.q[i]);
        for (int m = 0; m < 15; m++) printf(" %u", u.carry[m]);
        printf("\n");
    }
    return 0;
}
