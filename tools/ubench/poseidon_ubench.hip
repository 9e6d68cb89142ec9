// Development micro-benchmark: instruction issue rates on gfx950 and Poseidon permutation variants.
// Build: hipcc -O3 --offload-arch=gfx950 -I near-light-client_amd/csrc tools/ubench/poseidon_ubench.hip -o /tmp/pub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "poseidon.hpp"
#include "gl32.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ---- issue-rate probes: N dependent-free instructions per lane, 8 independent chains ----
template <int OP>
__global__ void k_rate(uint32_t* out, uint32_t a, uint32_t b, int iters) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    uint64_t y0 = x0, y1 = x1, y2 = x2, y3 = x3, y4 = x4, y5 = x5, y6 = x6, y7 = x7;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) {  // v_mad_u32_u24
            asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n"
                         "v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (OP == 1) {  // v_mul_lo_u32
            asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                         "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 2) {  // v_mul_hi_u32
            asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                         "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 3) {  // v_mad_u64_u32
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(a), "v"(b) : "vcc");
        } else if (OP == 4) {  // v_add_u32 (reference full-rate op)
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 5) {  // v_lshl_add_u64
            asm volatile("v_lshl_add_u64 %0, %0, 1, %0\n v_lshl_add_u64 %1, %1, 1, %1\n v_lshl_add_u64 %2, %2, 1, %2\n v_lshl_add_u64 %3, %3, 1, %3\n"
                         "v_lshl_add_u64 %4, %4, 1, %4\n v_lshl_add_u64 %5, %5, 1, %5\n v_lshl_add_u64 %6, %6, 1, %6\n v_lshl_add_u64 %7, %7, 1, %7\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));
        } else if (OP == 6) {  // v_fma_f64
            double d0 = __longlong_as_double(y0), d1 = __longlong_as_double(y1), d2 = __longlong_as_double(y2), d3 = __longlong_as_double(y3);
            double d4 = __longlong_as_double(y4), d5 = __longlong_as_double(y5), d6 = __longlong_as_double(y6), d7 = __longlong_as_double(y7);
            double c = (double)a;
            asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                         "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(c));
            y0 = __double_as_longlong(d0); y1 = __double_as_longlong(d1); y2 = __double_as_longlong(d2); y3 = __double_as_longlong(d3);
            y4 = __double_as_longlong(d4); y5 = __double_as_longlong(d5); y6 = __double_as_longlong(d6); y7 = __double_as_longlong(d7);
        } else if (OP == 7) {  // v_mul_u32_u24
            asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n"
                         "v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 8) {  // v_mul_hi_u32_u24
            asm volatile("v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_hi_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n"
                         "v_mul_hi_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_hi_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP >= 20 && OP <= 39) {  // round 4: the glue instructions of the matrix-core linear layer
#define NLX_R8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc")
#define I20(k) "v_xor_b32 %" #k ", 0x80808080, %" #k "\n"
#define I21(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define I22(k) "v_lshl_add_u32 %" #k ", %" #k ", 8, %8\n"
#define I23(k) "v_lshlrev_b32 %" #k ", 16, %" #k "\n"
#define I24(k) "v_cndmask_b32_e64 %" #k ", 0, -1, vcc\n"
#define I25(k) "v_mov_b32 %" #k ", %8\n"
#define I26(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define I27(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define I28(k) "v_xor_b32 %" #k ", %8, %" #k "\n"
#define I29(k) "v_bfe_u32 %" #k ", %" #k ", 3, 20\n"
#define I30(k) "v_cndmask_b32_e32 %" #k ", 0, %8, vcc\n"
#define I31(k) "v_add_co_u32 %" #k ", vcc, %" #k ", %8\n"
#define I32(k) "v_and_b32 %" #k ", %8, %" #k "\n"
#define I33(k) "v_sub_u32 %" #k ", %" #k ", %8\n"
#define I34(k) "v_alignbit_b32 %" #k ", %" #k ", %8, 8\n"
#define I35(k) "v_bitop3_b32 %" #k ", %" #k ", %8, %9 bitop3:0x96\n"
#define I36(k) "v_mov_b32_dpp %" #k ", %" #k " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I37(k) "v_add_u32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I38(k) "v_lshlrev_b32_sdwa %" #k ", %9, %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
#define I39(k) "v_or_b32_sdwa %" #k ", %" #k ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n"
            if (OP == 20) NLX_R8(I20); else if (OP == 21) NLX_R8(I21); else if (OP == 22) NLX_R8(I22); else if (OP == 23) NLX_R8(I23);
            else if (OP == 24) NLX_R8(I24); else if (OP == 25) NLX_R8(I25); else if (OP == 26) NLX_R8(I26); else if (OP == 27) NLX_R8(I27);
            else if (OP == 28) NLX_R8(I28); else if (OP == 29) NLX_R8(I29); else if (OP == 30) NLX_R8(I30); else if (OP == 31) NLX_R8(I31);
            else if (OP == 32) NLX_R8(I32); else if (OP == 33) NLX_R8(I33); else if (OP == 34) NLX_R8(I34); else if (OP == 35) NLX_R8(I35);
            else if (OP == 36) NLX_R8(I36); else if (OP == 37) NLX_R8(I37); else if (OP == 38) NLX_R8(I38); else NLX_R8(I39);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ (uint32_t)(y0 ^ y1 ^ y2 ^ y3 ^ y4 ^ y5 ^ y6 ^ y7);
}

// ---- the same probes counted in SHADER CYCLES (s_memtime) instead of wall time: cycles per wave-instruction per SIMD at
// 1, 2, 4 and 8 waves per SIMD, so that the issue cost is separated from the clock the chip holds under the load ----
template <int OP>
__global__ void k_cycles(uint32_t* out, unsigned long long* cyc, uint32_t a, uint32_t b, int iters) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    uint64_t y0 = x0, y1 = x1, y2 = x2, y3 = x3, y4 = x4, y5 = x5, y6 = x6, y7 = x7;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 4) {
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 3) {
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(a), "v"(b) : "vcc");
        } else if (OP == 10) {  // the carry-chain glue of the field code: v_add_co_u32 feeding v_addc_co_u32 through VCC
            asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, 0, %3, vcc\n"
                         "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, 0, %7, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
        } else if (OP == 11) {  // the multiply-accumulate + carry-count pair of the unreduced sums (GateAcc::mac)
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_addc_co_u32 %4, vcc, 0, %4, vcc\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n"
                         "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_addc_co_u32 %6, vcc, 0, %6, vcc\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_addc_co_u32 %7, vcc, 0, %7, vcc\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
        } else if (OP == 12) {  // v_cndmask / v_cmp pair of a canonicalising add
            asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ c0 ^ c1 ^ c2 ^ c3 ^ (uint32_t)(y0 ^ y1 ^ y2 ^ y3 ^ y4 ^ y5 ^ y6 ^ y7);
}

template <int OP>
int cycles(const char* name, uint32_t* d_out, unsigned long long* d_cyc) {
    const int iters = 4096, threads = 256;  // one block of 4 waves = one wave per SIMD of a CU
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        const int blocks = 256 * waves_per_simd;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_cycles<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 12345u, 678u, 16);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_cycles<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 12345u, 678u, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks * 4);
        CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double per_wave = sum / h.size() / (iters * 8.0);       // s_memtime ticks per instruction of one wave
        const double per_simd = per_wave / waves_per_simd;            // waves of a SIMD interleave
        const double ns = ms * 1e6 / ((double)blocks * 4 * iters * 8 / 1024.0);
        // an s_memtime tick is one shader cycle (MI355X_MICROARCH.md, cycle-constants table): ticks per instruction per SIMD
        // slot = the issue cost in cycles; ns / ticks = the clock period the chip held during the launch
        printf("%-34s %d wave(s)/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD; s_memtime %.4f ticks per instruction per wave (%.4f per SIMD slot -> %.2f GHz)\n",
               name, waves_per_simd, ms, ns, per_wave, per_simd, per_simd / ns);
    }
    return 0;
}

template <int OP>
int rate(const char* name, uint32_t* d_out) {
    const int iters = 4096, blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u, 678u, 16);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u, 678u, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double wave_instrs = (double)blocks * (threads / 64) * iters * 8;
    double per_simd = wave_instrs / 1024.0;  // 256 CUs x 4 SIMDs
    printf("%-18s %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles @2.4GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    return 0;
}


// ---- does a matrix instruction overlap the vector pipe?  NV v_mad_u64_u32 + NM v_mfma_i32_32x32x32_i8 per iteration, the
// matrix instructions on NM independent accumulators (each depends only on its own previous result, NV vector instructions away)
typedef int pi32x4_t __attribute__((ext_vector_type(4)));
typedef int pi32x16_t __attribute__((ext_vector_type(16)));
template <int NV, int NM>
__global__ __launch_bounds__(256) void k_mix(uint32_t* out, uint32_t a, uint32_t b, int iters) {
    uint64_t y[8];
    for (int i = 0; i < 8; i++) y[i] = threadIdx.x + i;
    pi32x16_t acc[NM > 0 ? NM : 1];
    for (int m = 0; m < (NM > 0 ? NM : 1); m++)
        for (int i = 0; i < 16; i++) acc[m][i] = 0;
    pi32x4_t fa, fb;
    fa.x = (int)(threadIdx.x * 3u); fa.y = (int)a; fa.z = (int)b; fa.w = 1;
    fb.x = (int)(threadIdx.x * 5u); fb.y = (int)b; fb.z = (int)a; fb.w = 2;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < NM; m++) acc[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc[m], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; i++)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y[i & 7]) : "v"(a), "v"(b) : "vcc");
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) r ^= (uint32_t)y[i] ^ (uint32_t)(y[i] >> 32);
    for (int m = 0; m < NM; m++)
        for (int i = 0; i < 16; i++) r ^= (uint32_t)acc[m][i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NV, int NM>
int mix(uint32_t* d_out, int waves_per_simd) {
    const int iters = 4096, blocks = 256 * waves_per_simd, threads = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mix<NV, NM>), dim3(blocks), dim3(threads), 0, 0, d_out, 12345u, 678u, 16);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<NV, NM>), dim3(blocks), dim3(threads), 0, 0, d_out, 12345u, 678u, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd_iters = (double)blocks * (threads / 64) * iters / 1024.0;
    printf("mix %2d v_mad_u64_u32 + %d v_mfma_i32_32x32x32_i8, %d wave(s)/SIMD: %.3f ms -> %.1f ns per iteration per SIMD\n", NV, NM, waves_per_simd, ms,
           ms * 1e6 / per_simd_iters);
    return 0;
}

// ---- Poseidon variants ----
namespace v1 {
// MDS on three 22-bit limbs with full-rate 24-bit multiply-adds
__device__ __forceinline__ void mds_layer(uint64_t (&s)[12]) {
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t l0[12], l1[12], l2[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        l0[i] = (uint32_t)s[i] & 0x3FFFFFu;
        l1[i] = (uint32_t)(s[i] >> 22) & 0x3FFFFFu;
        l2[i] = (uint32_t)(s[i] >> 44);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint32_t a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            a0 = __umul24(l0[(i + r) % 12], C[i]) + a0;
            a1 = __umul24(l1[(i + r) % 12], C[i]) + a1;
            a2 = __umul24(l2[(i + r) % 12], C[i]) + a2;
        }
        if (r == 0) {
            a0 += l0[0] << 3; a1 += l1[0] << 3; a2 += l2[0] << 3;
        }
        // value = a0 + a1 * 2^22 + a2 * 2^44   (a_k < 2^31)
        uint64_t lo = (uint64_t)a0 + ((uint64_t)a1 << 22);
        uint64_t add = (uint64_t)(a2 & 0xFFFFFu) << 44;
        uint64_t l = lo + add;
        uint64_t h = (uint64_t)(a2 >> 20) + (l < add ? 1u : 0u);
        uint64_t t1 = (h << 32) - h;
        uint64_t res = l + t1;
        if (res < t1) res += gl::EPS;
        s[r] = res;
    }
}
__device__ __forceinline__ void permute_loose(uint64_t (&s)[12]) {
    const uint64_t* rc = poseidon::RC_DEV;
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = poseidon::sbox7(gl::add_loose(s[i], rc[r * 12 + i]));
        mds_layer(s);
    }
#pragma unroll 1
    for (int r = 4; r < 26; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl::add_loose(s[i], rc[r * 12 + i]);
        s[0] = poseidon::sbox7(s[0]);
        mds_layer(s);
    }
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = poseidon::sbox7(gl::add_loose(s[i], rc[r * 12 + i]));
        mds_layer(s);
    }
}
}  // namespace v1

namespace v2 {
__device__ __forceinline__ void permute_loose(uint64_t (&st)[12]) {
    const uint64_t* rc = poseidon::RC_DEV;
    gl32::F s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl32::from_u64(st[i]);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl32::sbox7(gl32::add_const(s[i], rc[r * 12 + i]));
        gl32::mds_layer(s);
    }
#pragma unroll 1
    for (int r = 4; r < 26; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl32::add_const(s[i], rc[r * 12 + i]);
        s[0] = gl32::sbox7(s[0]);
        gl32::mds_layer(s);
    }
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl32::sbox7(gl32::add_const(s[i], rc[r * 12 + i]));
        gl32::mds_layer(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = gl32::to_u64(s[i]);
}
}  // namespace v2

namespace v3 {
// round constants folded into the previous linear layer's accumulators
__device__ __forceinline__ void permute_loose(uint64_t (&st)[12]) {
    const uint64_t* rc = poseidon::RC_DEV;
    gl32::F s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl32::add_const(gl32::from_u64(st[i]), rc[i]);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl32::sbox7(s[i]);
        gl32::mds_layer(s, rc + (r + 1) * 12);
    }
#pragma unroll 1
    for (int r = 4; r < 26; r++) {
        s[0] = gl32::sbox7(s[0]);
        gl32::mds_layer(s, rc + (r + 1) * 12);
    }
#pragma unroll 1
    for (int r = 26; r < 29; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl32::sbox7(s[i]);
        gl32::mds_layer(s, rc + (r + 1) * 12);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl32::sbox7(s[i]);
    gl32::mds_layer(s);
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = gl32::to_u64(s[i]);
}
}  // namespace v3


namespace v4 {
// The linear layer on the MATRIX cores.  out[r] = sum_j M[r][j] s_j with M[r][j] = C[(j - r) mod 12] (+ 8 at [0][0]) and
// 64-bit s_j = sum_b 2^(8b) byte_b(s_j): eight products of the constant 12 x 12 matrix with the state's byte planes, one
// v_mfma_i32_32x32x32_i8 each for all 64 states of a wave.  No lane ever moves data: B[k][col n] is supplied by lanes n
// (k < 16) and n + 32 (k >= 16) - each its OWN state's twelve bytes of the plane - and A is block-placed so that output r of
// the state in lane n + 32 h lands in row (r & 3) + 8 (r >> 2) + 4 h, which the 32 x 32 accumulator layout (col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) hands back to that same lane as register r.  Bytes are signed for the
// instruction, so the planes are biased by 128 (xor 0x80) and the constant 128 * rowsum * (1 + 2^8 + 2^16 + 2^24) rides in
// the round-constant table (RCB) that seeds the recombination.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ i32x4 a_fragment() {
    constexpr int C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint32_t lane = threadIdx.x & 63, rho = lane & 31, h = lane >> 5;
    const uint32_t g = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
    uint32_t w[4] = {0, 0, 0, 0};
    if (h == g && rho < 24) {
#pragma unroll
        for (int j = 0; j < 12; j++) {
            uint32_t m = 0;
#pragma unroll
            for (int rr = 0; rr < 12; rr++)
                if ((uint32_t)rr == r) m = (uint32_t)C[(j - rr + 12) % 12] + ((rr == 0 && j == 0) ? 8u : 0u);
            w[j >> 2] |= m << (8 * (j & 3));
        }
    }
    i32x4 a;
    a.x = (int)w[0]; a.y = (int)w[1]; a.z = (int)w[2]; a.w = (int)w[3];
    return a;
}

// 4 x 4 byte transpose: p[b] = (x0.byte b, x1.byte b, x2.byte b, x3.byte b)
__device__ __forceinline__ void transpose4(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t (&p)[4]) {
    const uint32_t t01l = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t01h = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
    const uint32_t t23l = __builtin_amdgcn_perm(x3, x2, 0x05010400u), t23h = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
    p[0] = __builtin_amdgcn_perm(t23l, t01l, 0x05040100u);
    p[1] = __builtin_amdgcn_perm(t23l, t01l, 0x07060302u);
    p[2] = __builtin_amdgcn_perm(t23h, t01h, 0x05040100u);
    p[3] = __builtin_amdgcn_perm(t23h, t01h, 0x07060302u);
}

__device__ __forceinline__ int opaque_s(int v) {   // a wave-uniform constant the compiler must treat as a register
    asm("" : "+s"(v));                              // (so that d * 2^8k + acc stays ONE v_mad_i64_i32 instead of shifts and adds)
    return v;
}
// rcb: 24 wave-uniform words for this layer: [r] = bias + low half of the next round's constant, [12 + r] = bias + high half
__device__ __forceinline__ void mds_layer(gl32::F (&s)[12], const i32x4 a, const uint64_t* __restrict__ rcb) {
    const int m0 = opaque_s(1), m8 = opaque_s(1 << 8), m16 = opaque_s(1 << 16), m24 = opaque_s(1 << 24);
    i32x16 zero;
#pragma unroll
    for (int i = 0; i < 16; i++) zero[i] = 0;
    uint64_t acc[2][12];
#pragma unroll
    for (int half = 0; half < 2; half++) {   // planes 0 .. 3 come from the low words, 4 .. 7 from the high words
        uint32_t pl[4][3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            uint32_t p[4];
            if (half == 0)
                transpose4(s[4 * q].lo ^ 0x80808080u, s[4 * q + 1].lo ^ 0x80808080u, s[4 * q + 2].lo ^ 0x80808080u, s[4 * q + 3].lo ^ 0x80808080u, p);
            else
                transpose4(s[4 * q].hi ^ 0x80808080u, s[4 * q + 1].hi ^ 0x80808080u, s[4 * q + 2].hi ^ 0x80808080u, s[4 * q + 3].hi ^ 0x80808080u, p);
#pragma unroll
            for (int b = 0; b < 4; b++) pl[b][q] = p[b];
        }
        i32x16 d[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            i32x4 bf;
            bf.x = (int)pl[b][0]; bf.y = (int)pl[b][1]; bf.z = (int)pl[b][2]; bf.w = 0;
            d[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bf, zero, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 12; r++) {
            int64_t t = (int64_t)d[0][r] * m0 + (int64_t)rcb[12 * half + r];
            t = (int64_t)d[1][r] * m8 + t;
            t = (int64_t)d[2][r] * m16 + t;
            t = (int64_t)d[3][r] * m24 + t;
            acc[half][r] = (uint64_t)t;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) s[r] = gl32::fold_acc(acc[0][r], acc[1][r]);
}

__device__ __forceinline__ void permute_loose(uint64_t (&st)[12], const uint64_t* __restrict__ rcb) {
    const uint64_t* rc = poseidon::RC_DEV;
    const i32x4 a = a_fragment();
    gl32::F s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl32::add_const(gl32::from_u64(st[i]), rc[i]);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl32::sbox7(s[i]);
        mds_layer(s, a, rcb + (r + 1) * 24);
    }
#pragma unroll 1
    for (int r = 4; r < 26; r++) {
        s[0] = gl32::sbox7(s[0]);
        mds_layer(s, a, rcb + (r + 1) * 24);
    }
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl32::sbox7(s[i]);
        mds_layer(s, a, rcb + (r + 1) * 24);   // entry 30: the bias alone
    }
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = gl32::to_u64(s[i]);
}
}  // namespace v4

__device__ uint64_t RCB_DEV[31 * 24];

#ifndef PUB_WAVES
#define PUB_WAVES 4
#endif
template <int V>
__global__ __launch_bounds__(256, PUB_WAVES) void k_perm(uint64_t* states, size_t n, int reps) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = states[i * n + t];
    for (int k = 0; k < reps; k++) {
        if (V == 0) poseidon::permute_loose(s);
        else if (V == 1) v1::permute_loose(s);
        else if (V == 2) v2::permute_loose(s);
        else if (V == 3) v3::permute_loose(s);
        else v4::permute_loose(s, RCB_DEV);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) states[i * n + t] = gl::canon(s[i]);
}

template <int V>
int bench_perm(const char* name, uint64_t* d_states, size_t n, std::vector<uint64_t>& out) {
    std::vector<uint64_t> init(12 * n);
    uint64_t x = 88172645463325252ull;
    for (auto& v : init) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x % gl::P; }
    CK(hipMemcpy(d_states, init.data(), init.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 4;
    hipLaunchKernelGGL(k_perm<V>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d_states, n, 1);
    CK(hipMemcpy(d_states, init.data(), init.size() * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_perm<V>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d_states, n, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    out.resize(12 * n);
    CK(hipMemcpy(out.data(), d_states, out.size() * 8, hipMemcpyDeviceToHost));
    printf("%-28s %.3f ms for %zu x %d perms -> %.3f Gperm/s\n", name, ms, n, reps, n * reps / ms / 1e6);
    return 0;
}

int main() {
    uint32_t* d_out;
    CK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    rate<4>("v_add_u32", d_out);
    rate<0>("v_mad_u32_u24", d_out);
    rate<7>("v_mul_u32_u24", d_out);
    rate<8>("v_mul_hi_u32_u24", d_out);
    rate<1>("v_mul_lo_u32", d_out);
    rate<2>("v_mul_hi_u32", d_out);
    rate<3>("v_mad_u64_u32", d_out);
    rate<5>("v_lshl_add_u64", d_out);
    rate<6>("v_fma_f64", d_out);
    rate<20>("v_xor_b32 literal", d_out);
    rate<28>("v_xor_b32 vgpr", d_out);
    rate<21>("v_perm_b32", d_out);
    rate<22>("v_lshl_add_u32", d_out);
    rate<23>("v_lshlrev_b32", d_out);
    rate<24>("v_cndmask_b32_e64", d_out);
    rate<25>("v_mov_b32", d_out);
    rate<26>("v_add3_u32", d_out);
    rate<27>("v_and_or_b32", d_out);
    rate<29>("v_bfe_u32", d_out);
    rate<30>("v_cndmask_b32_e32", d_out);
    rate<31>("v_add_co_u32 alone", d_out);
    rate<32>("v_and_b32", d_out);
    rate<33>("v_sub_u32", d_out);
    rate<34>("v_alignbit_b32", d_out);
    rate<35>("v_bitop3_b32", d_out);
    rate<36>("v_mov_b32_dpp quad", d_out);
    rate<37>("v_add_u32_sdwa", d_out);
    rate<38>("v_lshlrev_b32_sdwa", d_out);
    rate<39>("v_or_b32_sdwa preserve", d_out);
    for (int w : {4, 8}) {
        mix<16, 0>(d_out, w); mix<16, 1>(d_out, w); mix<16, 2>(d_out, w); mix<32, 1>(d_out, w); mix<64, 1>(d_out, w); mix<64, 0>(d_out, w);
        mix<0, 1>(d_out, w); mix<0, 2>(d_out, w);
    }
    unsigned long long* d_cyc;
    CK(hipMalloc(&d_cyc, 256 * 8 * 4 * 8));
    cycles<4>("v_add_u32", d_out, d_cyc);
    cycles<10>("v_add_co_u32 + v_addc_co_u32", d_out, d_cyc);
    cycles<12>("v_cmp_lt_u32 + v_cndmask_b32", d_out, d_cyc);
    cycles<3>("v_mad_u64_u32", d_out, d_cyc);
    cycles<11>("v_mad_u64_u32 + v_addc_co_u32", d_out, d_cyc);
    size_t n = 1 << 21;
    uint64_t* d_states;
    CK(hipMalloc(&d_states, 12 * n * 8));
    std::vector<uint64_t> o0, o1;
    bench_perm<0>("poseidon v0 (mad_u64_u32 MDS)", d_states, n, o0);
    bench_perm<1>("poseidon v1 (u24 limb MDS)", d_states, n, o1);
    printf("v1 == v0: %s\n", o0 == o1 ? "yes" : "NO");
    std::vector<uint64_t> o2;
    bench_perm<2>("poseidon v2 (u32-pair asm)", d_states, n, o2);
    printf("v2 == v0: %s\n", o0 == o2 ? "yes" : "NO");
    std::vector<uint64_t> o3;
    bench_perm<3>("poseidon v3 (rc in accumulators)", d_states, n, o3);
    printf("v3 == v0: %s\n", o0 == o3 ? "yes" : "NO");
    {   // v4's table: per layer l (the constants of round l, l = 30: none) the 12 low and 12 high accumulator seeds
        std::vector<uint64_t> rcb(31 * 24);
        const uint64_t spread = 1ull + (1ull << 8) + (1ull << 16) + (1ull << 24);
        for (int l = 0; l <= 30; l++)
            for (int r = 0; r < 12; r++) {
                const uint64_t bias = 128ull * (r == 0 ? 264 : 256) * spread;
                const uint64_t c = l < 30 ? poseidon::RC_HOST[l * 12 + r] : 0;
                rcb[l * 24 + r] = bias + (uint32_t)c;
                rcb[l * 24 + 12 + r] = bias + (c >> 32);
            }
        CK(hipMemcpyToSymbol(HIP_SYMBOL(RCB_DEV), rcb.data(), rcb.size() * 8));
    }
    std::vector<uint64_t> o4;
    bench_perm<4>("poseidon v4 (MDS on v_mfma_i32_32x32x32_i8)", d_states, n, o4);
    printf("v4 == v0: %s\n", o0 == o4 ? "yes" : "NO");
    if (o0 != o4) {
        size_t bad = 0, first = (size_t)-1;
        for (size_t i = 0; i < o0.size(); i++) if (o0[i] != o4[i]) { bad++; if (first == (size_t)-1) first = i; }
        printf("   %zu of %zu words differ, first at %zu (state %zu word %zu): %016llx vs %016llx\n", bad, o0.size(), first, first % n, first / n,
               (unsigned long long)o0[first], (unsigned long long)o4[first]);
    }
    // occupancy: dynamic LDS per 256-lane block caps the blocks per CU = waves per SIMD (160 KB of LDS per CU)
    for (int V = 0; V <= 1; V++)
        for (int w : {8, 6, 5, 4, 3, 2}) {
            const size_t lds = w == 8 ? 0 : (size_t)(160 * 1024 / w) & ~(size_t)255;
            std::vector<uint64_t> init(12 * n);
            uint64_t x = 88172645463325252ull;
            for (auto& v : init) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x % gl::P; }
            CK(hipMemcpy(d_states, init.data(), init.size() * 8, hipMemcpyHostToDevice));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto kern = V == 0 ? k_perm<0> : k_perm<4>;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL(kern, dim3((unsigned)(n / 256)), dim3(256), lds, 0, d_states, n, 1);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3((unsigned)(n / 256)), dim3(256), lds, 0, d_states, n, 4);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("occupancy sweep %s: at most %d wave(s) per SIMD (dynamic LDS %zu B per block): %.3f ms -> %.3f G permutations/s\n",
                   V == 0 ? "v0 (VALU MDS)" : "v4 (MFMA MDS)", w, lds, ms, (double)n * 4 / (ms * 1e-3) / 1e9);
        }
    return 0;
}
