// skx_common.hpp -- shared constants and device helpers for the gfx950 kernels.
//
// Algorithms follow the reference's dependencies as restated in SURVEY.md 8(a):
//   murmur3 (A6)   murmurhash3 0.0.5 src/mmh3_128.rs, h1 of MurmurHash3_x64_128, u64 seed
//   classify (A4)  needletail 0.4.1 src/sequence.rs normalize(iupac=false)
//   canonical (A4) needletail 0.4.1 src/kmer.rs CanonicalKmers (min(fwd, revcomp) bytewise)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace skx {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u64 kPad = 0xFFFFFFFFFFFFFFFFull;    // matrix padding: never a query (queries <= max_ref < kEmpty)
constexpr u64 kEmpty = 0xFFFFFFFFFFFFFFFEull;  // empty slot of an LDS probe table
constexpr int kTileGenomes = 256;              // genomes per reference tile (one lane per genome)
constexpr int kWave = 64;

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ void wave_sync() {
    // LDS hand-off between lanes of ONE wave: order the DS ops and stop the compiler from
    // caching LDS values across the point.  No s_barrier: waves of a block run independent reads.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ u64 readlane64(u64 v, int l) {
    u32 lo = __builtin_amdgcn_readlane((u32)v, l);
    u32 hi = __builtin_amdgcn_readlane((u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_xor64(u64 v, int m) {
    u32 lo = __shfl_xor((u32)v, m, 64);
    u32 hi = __shfl_xor((u32)(v >> 32), m, 64);
    return ((u64)hi << 32) | lo;
}

// ---------------------------------------------------------------- murmur3 x64_128 (h1)
// a u64 from its two halves as a plain register pair (an `(hi << 32) | lo` expression tends to be folded into the
// neighbouring additions and costs moves + 64-bit adds instead)
__host__ __device__ __forceinline__ u64 make_u64(u32 lo, u32 hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef u32 u32x2_ __attribute__((ext_vector_type(2)));
    u32x2_ v;
    v.x = lo; v.y = hi;
    return __builtin_bit_cast(u64, v);
#else
    return ((u64)hi << 32) | lo;
#endif
}
// rotate left by a compile-time amount; on the device two v_alignbit_b32 (the compiler's shift/or sequences cost 3-4)
template <int R>
__host__ __device__ __forceinline__ u64 rotl64c(u64 x) {
    static_assert(R > 0 && R < 64 && R != 32, "rotation amount");
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    if constexpr (R < 32) {
        // new_hi = (hi << R) | (lo >> (32 - R)) = alignbit(hi, lo, 32 - R); new_lo = alignbit(lo, hi, 32 - R)
        const u32 nh = __builtin_amdgcn_alignbit(hi, lo, 32 - R), nl = __builtin_amdgcn_alignbit(lo, hi, 32 - R);
        return make_u64(nl, nh);
    } else {
        // rotate by 32 (swap the halves), then by R - 32
        const u32 nh = __builtin_amdgcn_alignbit(lo, hi, 64 - R), nl = __builtin_amdgcn_alignbit(hi, lo, 64 - R);
        return make_u64(nl, nh);
    }
#else
    return (x << R) | (x >> (64 - R));
#endif
}
__host__ __device__ __forceinline__ u64 fmix64(u64 k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
// x * 5 as one shift-add (v_lshl_add_u64 on gfx950) instead of two quarter-rate 32-bit multiplies
__host__ __device__ __forceinline__ u64 mul5(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 r;
    asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return x * 5;
#endif
}
// w0..w3: the k key bytes as little-endian u64 words, zero beyond k (k <= 32)
__host__ __device__ __forceinline__ u64 murmur3_h1_words(u64 w0, u64 w1, u64 w2, u64 w3, u32 k, u64 seed) {
    const u64 c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
    u64 h1 = seed, h2 = seed;
#define SKX_MM_BLOCK(K1, K2)                                                      \
    {                                                                             \
        u64 k1 = (K1), k2 = (K2);                                                 \
        k1 *= c1; k1 = rotl64c<31>(k1); k1 *= c2; h1 ^= k1;                       \
        h1 = rotl64c<27>(h1); h1 += h2; h1 = mul5(h1) + 0x52dce729ull;            \
        k2 *= c2; k2 = rotl64c<33>(k2); k2 *= c1; h2 ^= k2;                       \
        h2 = rotl64c<31>(h2); h2 += h1; h2 = mul5(h2) + 0x38495ab5ull;            \
    }
    if (k >= 16) SKX_MM_BLOCK(w0, w1)
    if (k >= 32) SKX_MM_BLOCK(w2, w3)
#undef SKX_MM_BLOCK
    const u32 tail = k & 15u;
    const u64 t1 = (k >= 16) ? w2 : w0, t2 = (k >= 16) ? w3 : w1;
    if (tail > 8) { u64 k2 = t2; k2 *= c2; k2 = rotl64c<33>(k2); k2 *= c1; h2 ^= k2; }
    if (tail > 0) { u64 k1 = t1; k1 *= c1; k1 = rotl64c<31>(k1); k1 *= c2; h1 ^= k1; }
    h1 ^= k; h2 ^= k;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
}

// the same for exactly 16 key bytes (k = 16, one block, no tail); SEED0: the seed is known to be 0 (sketchy's default,
// src/cli.rs:47-48), which folds the seed's xors and one 64-bit add away
template <bool SEED0>
__host__ __device__ __forceinline__ u64 murmur3_h1_16(u64 w0, u64 w1, u64 seed) {
    const u64 c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
    u64 k1 = w0, k2 = w1;
    k1 *= c1; k1 = rotl64c<31>(k1); k1 *= c2;
    k2 *= c2; k2 = rotl64c<33>(k2); k2 *= c1;
    u64 h1, h2;
    if (SEED0) {
        h1 = rotl64c<27>(k1);                      // (0 ^ k1) rotated, + h2 (= 0)
        h1 = mul5(h1) + 0x52dce729ull;
        h2 = rotl64c<31>(k2) + h1;
    } else {
        h1 = rotl64c<27>(seed ^ k1) + seed;
        h1 = mul5(h1) + 0x52dce729ull;
        h2 = rotl64c<31>(seed ^ k2) + h1;
    }
    h2 = mul5(h2) + 0x38495ab5ull;
    h1 ^= 16u; h2 ^= 16u;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    return h1 + h2;
}

// the same from the two FIRST-STAGE PRODUCTS p1 = w0 * c1 and p2 = w1 * c2 (mod 2^64): the k = 16 wave sketcher gets them out
// of an LDS table indexed by the canonical k-mer's 2-bit code (sketch_mul_tables, skx_kernels.hip) instead of multiplying
template <bool SEED0>
__host__ __device__ __forceinline__ u64 murmur3_h1_16_pre(u64 p1, u64 p2, u64 seed) {
    const u64 c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
    u64 k1 = rotl64c<31>(p1) * c2;
    u64 k2 = rotl64c<33>(p2) * c1;
    u64 h1, h2;
    if (SEED0) {
        h1 = rotl64c<27>(k1);
        h1 = mul5(h1) + 0x52dce729ull;
        h2 = rotl64c<31>(k2) + h1;
    } else {
        h1 = rotl64c<27>(seed ^ k1) + seed;
        h1 = mul5(h1) + 0x52dce729ull;
        h2 = rotl64c<31>(seed ^ k2) + h1;
    }
    h2 = mul5(h2) + 0x38495ab5ull;
    h1 ^= 16u; h2 ^= 16u;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    return h1 + h2;
}

// ... stopping one step short: x1, x2 with murmur3_h1_16_pre = f(x1) + f(x2), f(x) = x ^ (x >> 33) -- fmix64's last xor-shift,
// which only touches the LOW word.  The high word of the hash is hi(x1) + hi(x2) (+ a carry), so a sketcher that only keeps
// hashes up to some bound can tell from ONE 32-bit add whether a lane can be in range at all, and finish the hash
// (murmur3_finish_pair) for the few that can.
template <bool SEED0>
__host__ __device__ __forceinline__ void murmur3_h1_16_pre_split(u64 p1, u64 p2, u64 seed, u64& x1, u64& x2) {
    const u64 c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
    u64 k1 = rotl64c<31>(p1) * c2;
    u64 k2 = rotl64c<33>(p2) * c1;
    u64 h1, h2;
    if (SEED0) {
        h1 = rotl64c<27>(k1);
        h1 = mul5(h1) + 0x52dce729ull;
        h2 = rotl64c<31>(k2) + h1;
    } else {
        h1 = rotl64c<27>(seed ^ k1) + seed;
        h1 = mul5(h1) + 0x52dce729ull;
        h2 = rotl64c<31>(seed ^ k2) + h1;
    }
    h2 = mul5(h2) + 0x38495ab5ull;
    h1 ^= 16u; h2 ^= 16u;
    h1 += h2; h2 += h1;
    h1 ^= h1 >> 33; h1 *= 0xff51afd7ed558ccdull; h1 ^= h1 >> 33; h1 *= 0xc4ceb9fe1a85ec53ull;
    h2 ^= h2 >> 33; h2 *= 0xff51afd7ed558ccdull; h2 ^= h2 >> 33; h2 *= 0xc4ceb9fe1a85ec53ull;
    x1 = h1; x2 = h2;
}
__host__ __device__ __forceinline__ u64 murmur3_finish_pair(u64 x1, u64 x2) { return (x1 ^ (x1 >> 33)) + (x2 ^ (x2 >> 33)); }

// ---------------------------------------------------------------- membership filter
// Blocked Bloom filter over the reference's DISTINCT hashes: 64-bit words, word = h >> shift (hashes are uniform up to the
// largest reference hash, so position in that range needs no hash function), four bit positions inside the word from
// the low 24 bits of h (independent of the word index whenever shift >= 24).  No false negatives; at 32 table bits per
// distinct hash a word holds two keys on average: ~12 % of its bits set, false positives ~0.03 % -- where the direct-mapped
// bitmap of rounds 1-2 needed 64 bits per NON-distinct hash (4 GB at C2 instead of 32 MB) for the same dictionary size.
__host__ __device__ __forceinline__ u64 filter_mask(u64 h) {
    return (1ull << (h & 63u)) | (1ull << ((h >> 6) & 63u)) | (1ull << ((h >> 12) & 63u)) | (1ull << ((h >> 18) & 63u));
}
__device__ __forceinline__ bool filter_hit(const u64* __restrict__ words, u32 shift, u64 h) {
    const u64 m = filter_mask(h);
    return (words[h >> shift] & m) == m;
}

// ---------------------------------------------------------------- bases
// 0..3 = A,C,G,T ; 4 = any other retained byte (N, '-', IUPAC -> N) ; 5 = removed (whitespace)
__host__ __device__ __forceinline__ u32 classify_base(u32 c) {
    const u32 u = c & 0xDFu;
    u32 code = 4u;
    code = (u == 'A') ? 0u : code;
    code = (u == 'C') ? 1u : code;
    code = (u == 'G') ? 2u : code;
    code = (u == 'T' || u == 'U') ? 3u : code;
    code = (c == ' ' || c == '\t' || c == '\r' || c == '\n') ? 5u : code;
    return code;
}

// canonical k-mer as 2-bit codes, first base in the most significant position; returns its hash.
// fwd/rc ordering on the packed value == bytewise order of the ASCII strings (A<C<G<T).
// KT > 0: compile-time k (fully unrolled); KT == 0: run-time k (1..32).
template <int KT>
__host__ __device__ __forceinline__ u64 hash_canonical_packed(u64 canon, u32 k_rt, u64 seed) {
    const u32 k = KT > 0 ? (u32)KT : k_rt;
    u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
    for (u32 j = 0; j < (KT > 0 ? (u32)KT : 32u); ++j) {
        if (j < k) {
            const u32 c = (u32)(canon >> (2 * (k - 1 - j))) & 3u;
            const u64 ascii = (u64)((0x54474341u >> (8 * c)) & 0xFFu) << (8 * (j & 7));  // "ACGT"[c]
            if (j < 8) w0 |= ascii; else if (j < 16) w1 |= ascii; else if (j < 24) w2 |= ascii; else w3 |= ascii;
        }
    }
    return murmur3_h1_words(w0, w1, w2, w3, k, seed);
}

}  // namespace skx
