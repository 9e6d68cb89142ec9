// Host build of the Ed25519 trace generator's row code (csrc/ed25519_rows.hpp): reads slots as five 64-hex-digit numbers,
// one of 128 digits and a flag (ax ay rx ry s d active) per line and writes the round-0 trace of all slots, column-major
// u64, to the file named in argv[1]; exit code 5 if the row code reports a false statement.
// tests/test_ed25519_air.py compares it with the Python reference trace.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ed25519_rows.hpp"

using namespace nlx;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::vector<ed::Slot> slots;
    char buf[5][80], dbuf[160];
    unsigned active = 1;
    while (scanf("%64s %64s %64s %64s %64s %128s %u", buf[0], buf[1], buf[2], buf[3], buf[4], dbuf, &active) == 7) {
        uint64_t w[ed::SLOT_WORDS] = {0};
        for (int v = 0; v < 5; v++)
            for (int k = 0; k < 4; k++) {
                unsigned long long x = 0;
                sscanf(buf[v] + 16 * (3 - k), "%16llx", &x);
                w[4 * v + k] = x;
            }
        for (int k = 0; k < 8; k++) {
            unsigned long long x = 0;
            sscanf(dbuf + 16 * (7 - k), "%16llx", &x);
            w[20 + k] = x;
        }
        w[28] = active;
        ed::Slot s;
        ed::slot_from_words(w, s);
        slots.push_back(s);
    }
    const size_t ns = slots.size(), n = ns * ed::ROWS;
    std::vector<uint64_t> trace((size_t)ed::N_COLS0 * n);
    // pass 1: the input point of every row, and every slot's final point - on values only (fe25519_fast.hpp), checked
    // against the witness code's row
    std::vector<ed::Point> in(n), fin(ns);
    for (size_t k = 0; k < ns; k++) {
        ed::FastSlot fs;
        ed::fast_slot(slots[k], fs);
        ed::FastPoint fq;
        {
            uint32_t zero[16] = {0}, one[16] = {1};
            fq.x = fe::from_limbs16(zero);
            fq.y = fe::from_limbs16(one);
            fq.z = fe::from_limbs16(one);
        }
        ed::NoSink none;
        for (int r = 0; r < ed::ROWS; r++) {
            ed::Point q;
            ed::fast_store(fq, q);
            in[k * ed::ROWS + r] = q;
            const int bit = ed::ROWS - 1 - r;
            const int sbit = (slots[k].sw[bit >> 4] >> (bit & 15)) & 1, hbit = (slots[k].hw[bit >> 4] >> (bit & 15)) & 1;
            ed::Point o, o2;
            ed::row_main(none, q, sbit, hbit, slots[k], o);
            ed::fast_row(fq, sbit, hbit, fs);
            ed::fast_store(fq, o2);
            for (int i = 0; i < 16; i++)
                if (o.x[i] != o2.x[i] || o.y[i] != o2.y[i] || o.z[i] != o2.z[i]) {
                    fprintf(stderr, "fast row differs from the witness row: slot %zu row %d\n", k, r);
                    return 4;
                }
        }
        ed::fast_store(fq, fin[k]);
    }
    // pass 2: every row on its own
    bool all_ok = true;
    for (size_t k = 0; k < ns; k++)
        for (int r = 0; r < ed::ROWS; r++) {
            const size_t row = k * ed::ROWS + r, prev = (k + ns - 1) % ns;
            auto put = [&](uint32_t col, uint64_t v) { trace[(size_t)col * n + row] = v; };
            ed::Point o;
            all_ok &= ed::emit_row(r, slots[k], in[row], slots[prev].ry, &fin[prev], slots[prev].active != 0, put, o) && slots[k].s_in_range;
        }
    FILE* f = fopen(argv[1], "wb");
    if (!f) return 3;
    fwrite(trace.data(), 8, trace.size(), f);
    fclose(f);
    return all_ok ? 0 : 5;
}
