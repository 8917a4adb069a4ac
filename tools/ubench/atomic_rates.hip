// Micro-benchmark: scattered 32-bit atomic adds from the whole chip -- what bounds gain_sparse_kernel (one add per posting and batch
// into gain[batch][genome]: 2.5 M adds in 0.73 ms = 3.4 G/s alone on the chip).
//   shared  : every workgroup adds into ONE array of n counters (the kernel's pattern)
//   per-XCD : workgroups of XCD x (HW_REG_XCC_ID) add into replica x of the array -- a line is only ever touched by one XCD's L2
//   stride  : counters 64 bytes apart (the kernel's gain_s layout) or 4
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/atomic_rates.hip -o gpurun_out/atomic_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* a, unsigned n, unsigned stride, unsigned per_thread, int per_xcd, unsigned* xcc_seen) {
    const unsigned x = __builtin_amdgcn_s_getreg(6164) & 7u;  // hwreg(HW_REG_XCC_ID, 0, 4)
    if (threadIdx.x == 0) atomicOr(&xcc_seen[blockIdx.x & 7u], 1u << x);
    unsigned* base = per_xcd ? a + (size_t)x * n * stride : a;
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (unsigned i = 0; i < per_thread; ++i) {
        h = h * 1664525u + 1013904223u;
        atomicAdd(&base[(size_t)((h >> 8) % n) * stride], 1u);
    }
}
int main() {
    const unsigned n = 8u * 40960u;  // counters (eight batches x 40 960 genomes)
    unsigned *a = nullptr, *seen = nullptr;
    hipMalloc(&a, (size_t)8 * n * 16 * 4);
    hipMalloc(&seen, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (unsigned blocks : {256u, 2048u})
        for (unsigned stride : {1u, 16u})
            for (int per_xcd : {0, 1}) {
                const unsigned per_thread = 2560000u / (blocks * 256u) * 4u;
                hipMemset(a, 0, (size_t)8 * n * 16 * 4);
                hipMemset(seen, 0, 64);
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(e0, nullptr);
                    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, nullptr, a, n, stride, per_thread, per_xcd, seen);
                    hipEventRecord(e1, nullptr);
                    hipEventSynchronize(e1);
                    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                unsigned s[8]; hipMemcpy(s, seen, 32, hipMemcpyDeviceToHost);
                bool rr = true; for (int i = 0; i < 8; ++i) rr = rr && s[i] == (1u << i);
                const double adds = (double)blocks * 256 * per_thread;
                std::vector<unsigned> host((size_t)8 * n * 16);
                hipMemcpy(host.data(), a, host.size() * 4, hipMemcpyDeviceToHost);
                unsigned long long total = 0; for (unsigned v : host) total += v;
                printf("blocks %4u stride %2u %-7s %.3f ms  %.1f G adds/s  (sum %s, block %% 8 == XCD: %s)\n", blocks, stride, per_xcd ? "per-XCD" : "shared",
                       best, adds / best / 1e6, total == (unsigned long long)adds * 5 ? "ok" : "WRONG", rr ? "yes" : "no");
            }
    return 0;
}
