// Micro-benchmark: how fast does the scan's geometry stream the reference matrix from HBM, by load width?
// One block of 256 threads per 128 KB (a band of 64 rows x a tile of 256 genomes x 8 B, contiguous), as scan_lean_kernel;
//   W8  : every lane loads 8 B per row (its genome's hash), 8 + 8 rows in flight            -- the kernel's pattern
//   W16 : every lane loads 16 B (two genomes' hashes), half of the block per row, 4 + 4 loads in flight (same bytes in flight)
//   NT  : the same with non-temporal loads
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_rates.hip -o /tmp/stream_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
struct alignas(16) U2 { u64 x, y; };
// ST: every block also stores a 5 x 256 x 8 B result slab when it is done (0 none, 1 plain, 2 non-temporal), as scan_lean_kernel
template <int W, bool NT, int U, int ST = 0, int WORDS = 5, int EVERY = 1>
__global__ __launch_bounds__(256) void k(const u64* __restrict__ mat, u64* __restrict__ out, u64* __restrict__ slab) {
    const u64* band = mat + (size_t)blockIdx.x * 64 * 256;
    const unsigned c = threadIdx.x;
    u64 acc = 0;
    if (W == 8) {
        u64 h[U], hn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) h[u] = NT ? __builtin_nontemporal_load(&band[(size_t)u * 256 + c]) : band[(size_t)u * 256 + c];
        for (int i = U; i + U <= 64; i += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) hn[u] = NT ? __builtin_nontemporal_load(&band[(size_t)(i + u) * 256 + c]) : band[(size_t)(i + u) * 256 + c];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= h[u];
#pragma unroll
            for (int u = 0; u < U; ++u) h[u] = hn[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= h[u];
    } else {
        const U2* b2 = reinterpret_cast<const U2*>(band);  // 128 pairs per row
        const unsigned p = c & 127u, par = c >> 7;         // rows par, par + 2, ...
        auto ld = [&](int r) -> U2 {
            const U2* a = &b2[(size_t)(2 * r + par) * 128 + p];
            if (NT) { U2 v; v.x = __builtin_nontemporal_load(&a->x); v.y = __builtin_nontemporal_load(&a->y); return v; }
            return *a;
        };
        U2 h[U], hn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) h[u] = ld(u);
        for (int i = U; i + U <= 32; i += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) hn[u] = ld(i + u);
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= h[u].x ^ h[u].y;
#pragma unroll
            for (int u = 0; u < U; ++u) h[u] = hn[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= h[u].x ^ h[u].y;
    }
    if (acc == 0x123456789ull) out[blockIdx.x] = acc;
    if (ST >= 3) {  // compact M[word][genome]: band b of tile t touches words b .. b + WORDS - 1 (158 tiles x 157 bands at C2)
        const unsigned n_tiles = 158, t = blockIdx.x % n_tiles, b = blockIdx.x / n_tiles;
        u64* o = slab + (size_t)b * (n_tiles * 256) + t * 256 + c;
#pragma unroll
        for (int w = 0; w < WORDS; ++w) { if (ST == 4) atomicOr(&o[(size_t)w * n_tiles * 256], acc + w); else o[(size_t)w * n_tiles * 256] = acc + w; }
    } else if (ST && blockIdx.x % EVERY == EVERY - 1) {
        u64* o = slab + (size_t)(blockIdx.x / EVERY) * WORDS * 256 + c;
#pragma unroll
        for (int w = 0; w < WORDS; ++w) { if (ST == 2) __builtin_nontemporal_store(acc + w, &o[w * 256]); else o[w * 256] = acc + w; }
    }
}
template <int W, bool NT, int U, int ST = 0, int WORDS = 5, int EVERY = 1>
static void run(const char* name, const u64* mat, size_t bytes, u64* out, u64* slab = nullptr) {
    const unsigned blocks = (unsigned)(bytes / (64 * 256 * 8));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<W, NT, U, ST, WORDS, EVERY>), dim3(blocks), dim3(256), 0, 0, mat, out, slab); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a); hipLaunchKernelGGL((k<W, NT, U, ST, WORDS, EVERY>), dim3(blocks), dim3(256), 0, 0, mat, out, slab); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    printf("%-34s %6.1f GB  %7.3f ms  %6.2f TB/s  (%.3f of 8 TB/s)\n", name, bytes / 1e9, best, bytes / best / 1e9, bytes / best / 1e9 / 8.0);
}
int main() {
    for (size_t gb10 : {32, 120}) {
        const size_t bytes = gb10 * 100000000ull / (64 * 256 * 8) * (64 * 256 * 8);
        u64 *mat, *out, *slab; hipMalloc(&mat, bytes); hipMalloc(&out, 1 << 22); hipMemset(mat, 1, bytes);
        hipMalloc(&slab, bytes / (64 * 256 * 8) * 5 * 256 * 8);
        run<8, false, 8>("8 B per lane, 8+8 in flight", mat, bytes, out);
        run<8, true, 8>("8 B per lane, nt", mat, bytes, out);
        run<8, false, 16>("8 B per lane, 16+16 in flight", mat, bytes, out);
        run<8, false, 8, 1>("8 B per lane + slab stores", mat, bytes, out, slab);
        run<8, false, 8, 2>("8 B per lane + nt slab stores", mat, bytes, out, slab);
        run<8, true, 8, 1>("8 B nt loads + slab stores", mat, bytes, out, slab);
        run<8, true, 8, 2>("8 B nt loads + nt slab stores", mat, bytes, out, slab);
        run<8, false, 8, 1, 2, 1>("8 B + 2-word slab per block", mat, bytes, out, slab);
        run<8, false, 8, 1, 1, 1>("8 B + 1-word slab per block", mat, bytes, out, slab);
        run<8, false, 8, 1, 12, 8>("8 B + 12 words per 8 blocks", mat, bytes, out, slab);
        run<8, true, 8, 1, 12, 8>("8 B nt + 12 words per 8 blocks", mat, bytes, out, slab);
        run<8, false, 8, 1, 20, 16>("8 B + 20 words per 16 blocks", mat, bytes, out, slab);
        run<8, true, 8, 1, 20, 16>("8 B nt + 20 words per 16 blocks", mat, bytes, out, slab);
        run<8, true, 8, 2, 20, 16>("8 B nt + 20 nt words per 16 blocks", mat, bytes, out, slab);
        run<8, false, 8, 3, 4, 1>("8 B + stores into compact M (4 words)", mat, bytes, out, slab);
        run<8, true, 8, 3, 4, 1>("8 B nt + stores into compact M", mat, bytes, out, slab);
        run<8, false, 8, 4, 4, 1>("8 B + atomicOr into compact M", mat, bytes, out, slab);
        run<8, true, 8, 4, 4, 1>("8 B nt + atomicOr into compact M", mat, bytes, out, slab);
        run<16, false, 4>("16 B per lane, 4+4 in flight", mat, bytes, out);
        run<16, true, 4>("16 B per lane, 4+4, nt", mat, bytes, out);
        run<16, false, 8>("16 B per lane, 8+8 in flight", mat, bytes, out);
        run<16, true, 8>("16 B per lane, 8+8, nt", mat, bytes, out);
        hipFree(mat); hipFree(out); hipFree(slab);
    }
    return 0;
}
