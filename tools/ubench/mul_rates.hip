// Micro-benchmark: issue cost of the integer multiply flavours on gfx950 (per wave64 instruction, per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mul_rates.hip -o gpurun_out/mul_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_IT 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 12345u, a3 = a1 * 3u;
    uint32_t b0 = 0x114253d5u + seed, b1 = 0x87c37b91u, b2 = 0x2745937fu, b3 = 0x4cf5ad43u;
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(b0)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a1) : "v"(b1));
                           asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a2) : "v"(b2)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a3) : "v"(b3)); }
            if (OP == 1) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a0) : "v"(b0)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a1) : "v"(b1));
                           asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a2) : "v"(b2)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a3) : "v"(b3)); }
            if (OP == 2) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a0) : "v"(b0)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a1) : "v"(b1));
                           asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a2) : "v"(b2)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a3) : "v"(b3)); }
            if (OP == 3) { asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a0) : "v"(b0)); asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a1) : "v"(b1));
                           asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a2) : "v"(b2)); asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a3) : "v"(b3)); }
            if (OP == 4) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b0)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(b1));
                           asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(b2)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(b3)); }
            if (OP == 5) { uint64_t t0, t1; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t0) : "v"(a0), "v"(b0) : "vcc"); a0 = (uint32_t)t0 ^ (uint32_t)(t0 >> 32);
                           asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t1) : "v"(a1), "v"(b1) : "vcc"); a1 = (uint32_t)t1 ^ (uint32_t)(t1 >> 32); }
            if (OP == 6) { asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a0) : "v"(b0)); asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a1) : "v"(b1));
                           asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a2) : "v"(b2)); asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a3) : "v"(b3)); }
            if (OP == 7) { uint64_t t0 = ((uint64_t)a1 << 32) | a0; asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(t0)); a0 = (uint32_t)t0; a1 = (uint32_t)(t0 >> 32);
                           uint64_t t1 = ((uint64_t)a3 << 32) | a2; asm volatile("v_lshlrev_b64 %0, 9, %0" : "+v"(t1)); a2 = (uint32_t)t1; a3 = (uint32_t)(t1 >> 32); }
#define SKX_OP4(N, TXT) if (OP == N) { asm volatile(TXT : "+v"(a0) : "v"(b0), "v"(b1)); asm volatile(TXT : "+v"(a1) : "v"(b1), "v"(b2)); \
                                       asm volatile(TXT : "+v"(a2) : "v"(b2), "v"(b3)); asm volatile(TXT : "+v"(a3) : "v"(b3), "v"(b0)); }
            SKX_OP4(8, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
            SKX_OP4(9, "v_xor_b32 %0, %0, %1")
            SKX_OP4(10, "v_and_or_b32 %0, %0, %1, %2")
            SKX_OP4(11, "v_alignbit_b32 %0, %0, %1, 5")
            SKX_OP4(12, "v_add3_u32 %0, %0, %1, %2")
            SKX_OP4(13, "v_lshl_or_b32 %0, %0, 2, %1")
            SKX_OP4(14, "v_min_u32 %0, %0, %1")
            SKX_OP4(15, "v_perm_b32 %0, %0, %1, %2")
            SKX_OP4(16, "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
            SKX_OP4(17, "v_bfi_b32 %0, %0, %1, %2")
            SKX_OP4(18, "v_xad_u32 %0, %0, %1, %2")
            if (OP == 19) { uint64_t t0 = ((uint64_t)a1 << 32) | a0, t1 = ((uint64_t)a3 << 32) | a2, c0 = ((uint64_t)b1 << 32) | b0;
                            asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(t0) : "v"(c0)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(t1) : "v"(c0));
                            a0 = (uint32_t)t0; a1 = (uint32_t)(t0 >> 32); a2 = (uint32_t)t1; a3 = (uint32_t)(t1 >> 32); }
            if (OP == 20) { asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a0), "+v"(a1) : "v"(b0), "v"(b1) : "vcc");
                            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a2), "+v"(a3) : "v"(b2), "v"(b3) : "vcc"); }
            if (OP == 21) { asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b0), "v"(a1) : "vcc");
                            asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b2), "v"(a3) : "vcc"); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
template <int OP>
static void run(const char* name, int per_iter, uint32_t* d) {
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)blocks * 4 / 1024.0 * N_IT * 8 * per_iter;  // wave instructions per SIMD
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("%-18s %8.3f ms  %6.2f ns per wave-instruction per SIMD (= %.1f cycles at %.2f GHz nominal)\n", name, ms,
           ms * 1e6 / instr_per_simd, ms * 1e-3 * clk * 1e3 / instr_per_simd, clk / 1e6);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<4>("v_add_u32", 4, d); run<0>("v_mul_lo_u32", 4, d); run<1>("v_mul_hi_u32", 4, d); run<2>("v_mul_u32_u24", 4, d);
    run<6>("v_mul_hi_u32_u24", 4, d); run<3>("v_mad_u32_u24", 4, d); run<5>("v_mad_u64_u32(+xor)", 2, d); run<7>("v_lshlrev_b64", 2, d);
    run<8>("v_bitop3_b32", 4, d); run<9>("v_xor_b32", 4, d); run<10>("v_and_or_b32", 4, d); run<11>("v_alignbit_b32", 4, d);
    run<12>("v_add3_u32", 4, d); run<13>("v_lshl_or_b32", 4, d); run<14>("v_min_u32", 4, d); run<15>("v_perm_b32", 4, d);
    run<16>("v_lshlrev_b32_sdwa", 4, d); run<17>("v_bfi_b32", 4, d); run<18>("v_xad_u32", 4, d); run<19>("v_lshl_add_u64", 2, d);
    run<20>("v_add_co + v_addc", 4, d); run<21>("v_cmp + v_cndmask", 4, d);
    return 0;
}
