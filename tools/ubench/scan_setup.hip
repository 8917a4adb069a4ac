// Micro-benchmark: what does scan_lean_kernel's per-block SET-UP cost on top of the pure stream of its band?
// One 256-thread block per 128 KB (64 rows x 256 genomes x 8 B), non-temporal loads, 8 + 8 rows in flight -- the kernel's stream --
// plus, step by step, what the kernel does before it streams: V1 the dependent loads of its window and the window's bounds,
// V2 the slice copied to LDS behind a barrier, V3 the directory (4096 byte stores) behind a second barrier + the result tile zeroed,
// V4 only the LDS footprint (16 KB static: fewer blocks per CU), V5 = V3 with the first rows requested BEFORE the set-up.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/scan_setup.hip -o /tmp/scan_setup ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
template <int V, int PADKB = 0>
__global__ __launch_bounds__(256) void k(const u64* __restrict__ mat, u64* __restrict__ out, const u32* __restrict__ win, const u64* __restrict__ q) {
    __shared__ u64 slice[256];
    __shared__ unsigned char dir[4104];
    __shared__ u64 acc[5][256];
    __shared__ u64 pad[PADKB ? PADKB * 128 : 1];  // (occupancy experiments: more LDS per block, fewer blocks per CU)
    // V >= 10: XCD-aware order -- all bands of a tile on XCD (tile % 8) (observed: block b runs on XCD b % 8; checked per block
    // with HW_REG_XCC_ID, never assumed), so that the tile's columns of M can be OR-ed by atomics executed in that XCD's L2
    // (workgroup-scope form: no sc1) instead of at the memory side; a block that finds itself elsewhere uses the device-scope
    // form on a second copy of M
    unsigned blk = blockIdx.x, xt = 0, xb = 0; bool home = false;
    if (V >= 10) {
        const unsigned nb = 154, x8 = blockIdx.x & 7u, j = blockIdx.x >> 3;
        xt = (j / nb) * 8u + x8; xb = j % nb;
        if (xt >= 158u) return;
        blk = xt * nb + xb;
        home = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) == x8;
    }
    const u64* band = mat + (size_t)blk * 64 * 256;
    const unsigned c = threadIdx.x;
    u64 x = 0;
    if (PADKB) { pad[c] = c; x = pad[(c * 7) & 127]; }
    u64 h[8], hn[8];
    if (V == 5) {
#pragma unroll
        for (int u = 0; u < 8; ++u) h[u] = __builtin_nontemporal_load(&band[(size_t)u * 256 + c]);
    }
    if (V >= 1 && V != 4) {  // (V 6..: the set-up of V3)
        const u32 qa = win[2 * blk], qb = win[2 * blk + 1];
        const u64 lo = q[qa], hi = q[qb - 1];
        x = lo ^ hi;
        if (V >= 2) {
            const u32 n = qb - qa;
            for (u32 i = c; i < n; i += 256) slice[i] = q[qa + i];
            __syncthreads();
            if (V >= 3) {
                const unsigned shift = 64 - __builtin_clzll((hi - lo) | 1ull) > 12 ? 64 - __builtin_clzll((hi - lo) | 1ull) - 12 : 0;
                for (u32 i = c; i <= n; i += 256) {
                    const u32 bj = i < n ? (u32)((slice[i] - lo) >> shift) : 4096u;
                    const u32 bp = i == 0 ? 0xFFFFFFFFu : (u32)((slice[i - 1] - lo) >> shift);
                    for (u32 y = bp + 1u; y <= bj && y < 4100u; ++y) dir[y] = (unsigned char)i;
                }
#pragma unroll
                for (int w = 0; w < 5; ++w) acc[w][c] = 0;
                __syncthreads();
                x ^= dir[(c * 16) & 4095] ^ slice[c & 127] ^ acc[c % 5][c];
            } else x ^= slice[c & 127];
        }
    }
    if (V == 4) { slice[c] = c; dir[c] = (unsigned char)c; acc[0][c] = c; __syncthreads(); x = slice[(c + 1) & 255] ^ dir[(c + 7) & 255] ^ acc[0][(c + 3) & 255]; }
    if (V != 5) {
#pragma unroll
        for (int u = 0; u < 8; ++u) h[u] = __builtin_nontemporal_load(&band[(size_t)u * 256 + c]);
    }
    // V >= 6: the lean probe per element -- bucket from the high word, one byte of the directory, two slice entries, two compares;
    // V >= 7: a hit ORs its bit into the lane's column of the LDS tile (the matrix is filled so that HITP % of the elements hit);
    // V >= 8: the tile's non-zero words go to a compact M by atomicOr when the band is done
    u64* my_acc = &acc[0][c];
    auto probe = [&](u64 hv) {
        if (V < 6 || V == 9 || V == 10) { x ^= hv; return; }
        const u32 bk = min((u32)(hv >> 32) >> 8, 4096u);
        const u32 j = dir[bk & 4095u];
        const u64 e0 = slice[j & 255u], e1 = slice[(j + 1u) & 255u];
        const bool m1 = e1 == hv;
        if (V >= 7) {
            if ((e0 == hv) || m1) {
                const u32 qr = (j & 255u) + (m1 ? 1u : 0u);
                __hip_atomic_fetch_or(&my_acc[(size_t)((qr >> 6) % 5u) * 256], 1ull << (qr & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else x ^= (e0 == hv || m1) ? e0 : e1;
    };
    for (int i = 8; i + 8 <= 64; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) hn[u] = __builtin_nontemporal_load(&band[(size_t)(i + u) * 256 + c]);
#pragma unroll
        for (int u = 0; u < 8; ++u) probe(h[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) h[u] = hn[u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) probe(h[u]);
    if (V >= 10) {
        u64* o = out + (1 << 19) + (size_t)(xb / 3u) * (158 * 256) + xt * 256 + c;   // M[word][genome]: a band reaches ~5 words, the next band 1-2 further
        if (!home) { o += (size_t)64 * 158 * 256; if (c == 0) atomicAdd((unsigned*)out + 8, 1u); }
#pragma unroll
        for (int w = 0; w < 5; ++w) {
            const u64 v = V == 10 ? x + w : my_acc[(size_t)w * 256];
            if (v) { if (home) __hip_atomic_fetch_or(&o[(size_t)w * 158 * 256], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); else atomicOr(&o[(size_t)w * 158 * 256], v); }
        }
    } else if (V >= 8) {
        const unsigned n_tiles = 158, t = blockIdx.x % n_tiles, b = blockIdx.x / n_tiles;
        u64* o = out + (1 << 19) + (size_t)b * (n_tiles * 256) + t * 256 + c;   // M[word b + w][genome]
#pragma unroll
        for (int w = 0; w < 5; ++w) { const u64 v = V == 9 ? x + w : my_acc[(size_t)w * 256]; if (v) atomicOr(&o[(size_t)w * n_tiles * 256], v); }
    }
    if (x == 0x123456789ull) out[blockIdx.x] = x;
}
template <int V, int PADKB = 0>
static void run(const char* name, const u64* mat, size_t bytes, u64* out, const u32* win, const u64* q) {
    const unsigned blocks = V >= 10 ? 160u * 154u : (unsigned)(bytes / (64 * 256 * 8));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<V, PADKB>), dim3(blocks), dim3(256), 0, 0, mat, out, win, q); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a); hipLaunchKernelGGL((k<V, PADKB>), dim3(blocks), dim3(256), 0, 0, mat, out, win, q); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    printf("%-58s %7.3f ms  %6.2f TB/s  (%.3f of 8 TB/s)\n", name, best, bytes / best / 1e9, bytes / best / 1e9 / 8.0);
}
int main() {
    const size_t bytes = 3200000000ull / (64 * 256 * 8) * (64 * 256 * 8);
    const unsigned blocks = (unsigned)(bytes / (64 * 256 * 8));
    u64 *mat, *out, *q; u32* win;
    hipMalloc(&mat, bytes); hipMalloc(&out, (size_t)(1 << 22) + (size_t)170 * 158 * 256 * 8); hipMemset(out, 0, (size_t)(1 << 22) + (size_t)170 * 158 * 256 * 8); hipMemset(mat, 1, bytes);
    const unsigned nq = 10240;
    std::vector<u64> hq(nq); for (unsigned i = 0; i < nq; ++i) hq[i] = (u64)i * 0x000F423F00000ull + (i * 2654435761u % 977);
    std::vector<u32> hw(2 * blocks); for (unsigned b = 0; b < blocks; ++b) { const unsigned qa = (b / 158) * 60 % (nq - 256); hw[2 * b] = qa; hw[2 * b + 1] = qa + 200; }
    hipMalloc(&q, nq * 8); hipMalloc(&win, 2 * blocks * 4);
    hipMemcpy(q, hq.data(), nq * 8, hipMemcpyHostToDevice); hipMemcpy(win, hw.data(), 2 * blocks * 4, hipMemcpyHostToDevice);
    run<0>("V0 pure nt stream", mat, bytes, out, win, q);
    run<4>("V4 + 16 KB of LDS per block (occupancy only)", mat, bytes, out, win, q);
    run<1>("V1 + window and bounds loaded first (dependent loads)", mat, bytes, out, win, q);
    run<2>("V2 + slice of 200 entries copied to LDS, barrier", mat, bytes, out, win, q);
    run<3>("V3 + directory of 4096 bytes built, tile zeroed, barrier", mat, bytes, out, win, q);
    run<5>("V5 = V3 with the first 8 rows requested before the set-up", mat, bytes, out, win, q);
    run<6>("V6 = V3 + the lean probe on every element, no hits", mat, bytes, out, win, q);
    run<7>("V7 = V6 + hits OR-ed into the LDS tile", mat, bytes, out, win, q);
    run<8>("V8 = V7 + the tile's words atomicOr-ed into a compact M", mat, bytes, out, win, q);
    run<9>("V9 = V3 + 5 atomicOr per lane into a compact M, no probe", mat, bytes, out, win, q);
    hipMemset(out, 0, 64);
    run<10>("V10 = V9 in XCD order, atomics in the XCD's L2", mat, bytes, out, win, q);
    run<11>("V11 = V8 in XCD order, atomics in the XCD's L2", mat, bytes, out, win, q);
    { unsigned away = 0; hipMemcpy(&away, (unsigned*)out + 8, 4, hipMemcpyDeviceToHost); printf("blocks that did not run on XCD (block %% 8): %u of %u\n", away, 12 * 158 * 154); }
    run<7, 8>("V7 with 24 KB of LDS in all (6 blocks per CU)", mat, bytes, out, win, q);
    run<7, 16>("V7 with 32 KB (5 per CU)", mat, bytes, out, win, q);
    run<7, 24>("V7 with 40 KB (4 per CU)", mat, bytes, out, win, q);
    run<7, 36>("V7 with 52 KB (3 per CU)", mat, bytes, out, win, q);
    run<7, 46>("V7 with 62 KB (2 per CU)", mat, bytes, out, win, q);
    run<0>("V0 again", mat, bytes, out, win, q);
    return 0;
}
