// skx_kernels.hip -- hand-written gfx950 kernels for sketchy's streaming path.
//
// Pipeline of one scoring pass over a batch of B reads (DESIGN.md has the full picture):
//
//   sketch_wave_kernel     one wavefront per read: normalise, canonical k-mers, murmur3; of the hashes that can
//                          match at all (<= largest reference hash) those some genome holds (membership bitmap)
//                          are kept, sorted in LDS, deduplicated, truncated to s       (A3-A6 of SURVEY 8(a))
//                          (reads of any length: chunked through the wave's LDS)
//   sketch_block_kernel    one block per read whose hashes overflow a wave's buffer; full sketches of long sequences
//   count_scan_a/b         exclusive scan of the per-read pair counts
//   dict_* kernels         (read, hash) pairs + Q: the pass's sorted distinct query hashes (hash set + bucket sort)
//   pair_q_kernel          pair -> index of its hash in Q
//   window_kernel          per reference tile-band: the slice [qa,qb) of Q its hashes can meet
//   scan_kernel            THE roofline kernel: streams the resident s x N matrix once, probes an
//                          LDS table of the slice, ORs hit bits into M[word][genome]   (A2)
//   transpose_bits_kernel  M[word][genome] -> Mq[query][genome word] (64x64 bit transposes)
//   seg_sum / chunk_* / seg_prefix / rank_seg_top1 | rank_seg_topk | rank_seg / *_merge
//                          running table and per-read (sum desc, index asc) top-k, pruned      (A1, A7)
//
// Integer work throughout (u64 hash compares, bit counts): no MFMA.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "skx_common.hpp"
#include "skx_kernels.hpp"

#ifndef SKX_FEW_CANDS
#define SKX_FEW_CANDS 8  /* up to eight candidates (sixteen, with twice the LDS: the same) take the count path (rows fetched once, counts in LDS); round 2's form -- one
                            gather loop per candidate -- gained nothing beyond one: 1 -> 77.0 M reads/s, 6 -> 76.2 M, 16 -> 76.8 M */
#endif
#ifndef SKX_SCAN_ROWS
#define SKX_SCAN_ROWS 8
#endif
#ifndef SKX_SCAN_OCC
#define SKX_SCAN_OCC 1
#endif
#ifndef SKX_SCAN_SPLIT_READS
#define SKX_SCAN_SPLIT_READS 1
#endif
#ifndef SKX_WALK_BLOCKS
#define SKX_WALK_BLOCKS 256
#endif
#ifndef SKX_GAIN_SPARSE_STRIDE
#define SKX_GAIN_SPARSE_STRIDE 16
#endif
#ifndef SKX_SEGSUM_PRIO
#define SKX_SEGSUM_PRIO 2
#endif
#ifndef SKX_MQ_NT
#define SKX_MQ_NT 0
#endif
#ifndef SKX_RANK1_PRIO
#define SKX_RANK1_PRIO 3
#endif
namespace skx {

// position of genome word gw of query row q in the group-major bit matrix Mq (nq_rows rows per group)
__host__ __device__ __forceinline__ size_t mq_index(u32 gw, u32 q, u32 nq_rows) {
    return ((size_t)(gw / kRankWords) * nq_rows + q) * kRankWords + gw % kRankWords;
}

// =====================================================================================
// reference upload: genome-major columns -> tiled rank-major matrix
// =====================================================================================
// dst[t][i][c] = (i < eff_len[g]) ? src[g*s + i] : kPad   with g = t*256 + c.
// 32x32 LDS transpose so both sides are coalesced.  grid: (ceil(s/32), n_tile_genomes/32)
// src / eff_len start at the chunk's first genome; that genome sits at padded index pad_base (species are padded to
// whole rank groups; the matrix was filled with kPad beforehand, so only real genomes are written)
__global__ __launch_bounds__(256) void ref_tile_kernel(const u64* __restrict__ src, const u32* __restrict__ eff_len,
                                                       u64* __restrict__ dst, u32 s, u32 pad_base, u32 g_count) {
    __shared__ u64 tile[32][33];
    const u32 tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;  // 32 x 8
    const u32 i0 = blockIdx.x * 32u, gl0 = blockIdx.y * 32u;  // gl: genome index local to this chunk
    for (u32 yy = ty; yy < 32u; yy += 8u) {
        const u32 gl = gl0 + yy, i = i0 + tx;
        u64 v = kPad;
        if (gl < g_count && i < s && i < eff_len[gl]) v = src[(size_t)gl * s + i];
        tile[yy][tx] = v;
    }
    __syncthreads();
    for (u32 yy = ty; yy < 32u; yy += 8u) {
        const u32 i = i0 + yy, gl = gl0 + tx, g = pad_base + gl;
        if (i < s && gl < g_count) {
            const u32 t = g / kTileGenomes, c = g % kTileGenomes;
            dst[((size_t)t * s + i) * kTileGenomes + c] = tile[tx][yy];
        }
    }
}

// per (band b, tile t): smallest and largest real hash in the band's rows.  One block per (b,t).
__global__ __launch_bounds__(256) void band_bounds_kernel(const u64* __restrict__ mat, u32 s, u32 n_tiles, u32 rb,
                                                          u64* __restrict__ lo, u64* __restrict__ hi) {
    const u32 t = blockIdx.x % n_tiles, b = blockIdx.x / n_tiles, c = threadIdx.x;
    const u32 i0 = b * rb, i1 = min(s, i0 + rb);
    const u64* col = mat + (size_t)t * s * kTileGenomes + c;
    u64 mylo = kPad, myhi = 0;
    bool any = false;
    const u64 first = col[(size_t)i0 * kTileGenomes];
    if (first != kPad) {
        any = true; mylo = first; myhi = first;
        for (u32 i = i1; i > i0; --i) {  // last real element of this column inside the band
            const u64 v = col[(size_t)(i - 1) * kTileGenomes];
            if (v != kPad) { myhi = v; break; }
        }
    }
    __shared__ u64 slo[256], shi[256];
    __shared__ int sany[256];
    slo[c] = mylo; shi[c] = myhi; sany[c] = any;
    __syncthreads();
    for (u32 w = 128; w > 0; w >>= 1) {
        if (c < w) {
            if (sany[c + w]) {
                if (!sany[c]) { slo[c] = slo[c + w]; shi[c] = shi[c + w]; sany[c] = 1; }
                else { slo[c] = min(slo[c], slo[c + w]); shi[c] = max(shi[c], shi[c + w]); }
            }
        }
        __syncthreads();
    }
    if (c == 0) {
        if (sany[0]) { lo[blockIdx.x] = slo[0]; hi[blockIdx.x] = shi[0]; }
        else { lo[blockIdx.x] = 1; hi[blockIdx.x] = 0; }  // empty band: lo > hi
    }
}

// =====================================================================================
// read sketching: one wavefront per read, any read length
// =====================================================================================
// LDS per wave: hashes[HCAP] u64, then codes[kSketchCap + 64] bytes.  The read is normalised and hashed in CHUNKS
// of kSketchCap raw bytes; the last k-1 codes of a chunk are carried to the front of the next, so windows across a
// chunk border are seen exactly once and a read of any length needs no scratch outside the wave's LDS.
// INRANGE: keep only hashes <= max_ref before sorting.  Every such hash is smaller than every
// dropped one, so the first min(s, #distinct kept) of them ARE the part of the bottom-s sketch that
// can meet the reference (the only part scoring needs); out_len is then that count, not |sketch|.
// A read whose kept hashes overflow the buffer is handed on through a device-side list (never through the host):
//   HCAP < kSketchCap  -> `retry` (redone by the HCAP = kSketchCap variant; out_len = kSketchRetry meanwhile)
//   HCAP = kSketchCap  -> `big`   (redone by sketch_block_kernel, which holds 16 384 hashes and selects in passes)
// !INRANGE (full sketches: debug outputs, skx_sketch_reads): reads with more than kSketchCap k-mers go to `big` at once.
constexpr u32 kSketchRetry = 0xFFFFFFFFu;
// hash slots per read of the in-range fast variant (production keeps ~8 in-range hashes per 1.5 kb read; a read with more goes
// to the 2048-slot variant through the retry list).  128 since round 4: with the 4 KB of product tables per workgroup 256
// slots cost the kernel a workgroup per CU -- 17.2 KB instead of 23.3 KB: 8 per CU alone, 4 beside the scan's LDS pad.
constexpr int kSketchSmallHashes = 128;
// Row pool (sketch_finish, pool mode): kPoolParts sub-pools, each with its own bump counter on its own cache line behind the 16
// words of chk (chk[16 + 16 i]); a workgroup uses sub-pool blockIdx.x % kPoolParts.  (ONE counter serialised the batch:
// 98 304 same-address atomics at ~10 ns each are 1 ms -- twice the sketch kernel -- measured as 48 -> 37 M reads/s for a
// lone C2 batch.)  chk[11] = the entries the batch asked for, summed by publish_kernel.
constexpr u32 kChkPool = 11;
constexpr u32 kPoolParts = 64;
constexpr u32 kChkWords = 16 + 16 * kPoolParts;  // u32 words of a chk block
// list layout: [0] = number of entries, [1..] = read indices
__device__ __forceinline__ void list_append(u32* __restrict__ list, u32 r) { list[1u + atomicAdd(&list[0], 1u)] = r; }

__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = (u32)__shfl_up((int)v, d, 64);
        if ((int)lane_id() >= d) v += o;
    }
    return v;
}

// normalisation table of the wave sketchers (one per block, in LDS): 0..3 = A C G T/U (either case), 4 = any other retained
// byte, 0x80 = removed (whitespace) -- classify_base() with the "removed" class moved to a flag bit
__device__ __forceinline__ void fill_base_lut(unsigned char* lut) {
    for (u32 i = threadIdx.x; i < 256u; i += blockDim.x) {  // (blocks of 64 or 256 threads)
        const u32 c = classify_base(i);
        lut[i] = (unsigned char)(c == 5u ? 0x80u : c);
    }
}
// First-stage products of murmur3 for k = 16, out of LDS instead of the multiplier (round 4).  The 16 key bytes are ASCII
// bases, so murmur3's k1 * c1 and k2 * c2 are sums over FOUR bases at a time: with W(i) = the little-endian word of the four
// bases whose 2-bit codes form byte i (first base in the two highest bits, as in the rolling windows),
//     k1 * c1 = W(lo) * c1 + ((W(hi) * c1) << 32)    (mod 2^64; lo / hi = bases 0-3 / 4-7, likewise k2 * c2 with bases 8-15)
// i.e. t1[lo] + (low word of t1[hi] << 32) from a 256-entry table of 64-bit products: two LDS reads and one 32-bit add on the
// otherwise idle DS pipe instead of four quarter-rate multiplier instructions (v_mad_u64_u32 + 2 v_mul_lo_u32 + v_add3: ~21 issue
// cycles, profiles/r03f_mul_rates.txt) -- and the index is the CANONICAL 2-bit window the loop rolls anyway, so the rolling
// 16-byte ASCII blocks of both strands (eight v_alignbit / v_perm per k-mer) and the four v_cndmask that picked one are gone
// too: ~100 -> ~78 VALU instructions per 64 k-mers.
#ifndef SKX_SK_LOTAB
#define SKX_SK_LOTAB 1  /* the 32-bit halves of the products from compact tables of their own (conflicts: below) */
#endif
#ifndef SKX_SK_EARLY
#define SKX_SK_EARLY 1  /* in-range test on the hash's high word before its low word is finished (sketch_one_read) */
#endif
template <int KT>
struct SketchTables {
    unsigned char lut[256];
    u64 t1[KT == 16 ? 256 : 1];  // W(i) * c1
    u64 t2[KT == 16 ? 256 : 1];  // W(i) * c2
    // the low words again, packed: a 4-byte read out of the 8-byte entries above only ever touches the even banks (ds_read_b32
    // banks = (a / 4) mod 32: 32 random lanes on 16 banks, ~5 deep; on all 32, ~3.4)
    u32 t1lo[(KT == 16 && SKX_SK_LOTAB) ? 256 : 1];
    u32 t2lo[(KT == 16 && SKX_SK_LOTAB) ? 256 : 1];
};
template <int KT>
__device__ __forceinline__ void fill_sketch_tables(SketchTables<KT>* tb) {
    fill_base_lut(tb->lut);
    if constexpr (KT == 16) {
        for (u32 i = threadIdx.x; i < 256u; i += blockDim.x) {
            const u32 sel = ((i >> 6) & 3u) | (((i >> 4) & 3u) << 8) | (((i >> 2) & 3u) << 16) | ((i & 3u) << 24);
            const u64 w = (u64)__builtin_amdgcn_perm(0u, 0x54474341u, sel);  // "ACGT"[selector byte]
            tb->t1[i] = w * 0x87c37b91114253d5ull;
            tb->t2[i] = w * 0x4cf5ad432745937full;
#if SKX_SK_LOTAB
            tb->t1lo[i] = (u32)(w * 0x87c37b91114253d5ull);
            tb->t2lo[i] = (u32)(w * 0x4cf5ad432745937full);
#endif
        }
    }
}
// byte `BYTE` of x, shifted left by SH (the byte offset of a table entry of 1 << SH bytes): one SDWA instruction
template <int BYTE, u32 SH>
__device__ __forceinline__ u32 byte_shl(u32 x) {
    u32 r;
    if constexpr (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "s"(SH), "v"(x));
    else if constexpr (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "s"(SH), "v"(x));
    else if constexpr (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "s"(SH), "v"(x));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "s"(SH), "v"(x));
    return r;
}
// normalise raw bytes [from, to) of the read into codes[nb ...]; returns the new end.  Four bytes per lane: one dword
// load, four table reads, and -- as long as nothing has been removed and the end is word-aligned (the usual case: reads
// hold no whitespace) -- one word store: ~4 VALU instructions per 64 bases instead of ~22 for the byte-at-a-time
// classification (measured: the sketch kernel is VALU-bound and a third of its instructions were outside the hash loop).
// A group of 256 bytes with a removed byte in it, and everything behind it in this call, is compacted byte by byte.
// `bad` collects (per lane) the invalid-code bits (bit 2 of a byte) of everything normalised: a chunk without any lets the
// hash loop drop its per-window validity tracking (sketch_one_read)
__device__ __forceinline__ u32 wave_normalise4(const uint8_t* __restrict__ rd, u32 from, u32 to, uint8_t* codes, u32 nb,
                                               u32 lane, const unsigned char* lut, u32& bad) {
    bool packed = (nb & 3u) == 0u;  // wave-uniform
    for (u32 base = from; base < to; base += 256u) {
        const u32 idx = base + 4u * lane;
        const u32 have = idx < to ? min(4u, to - idx) : 0u;
        u32 x = 0x41414141u;  // (absent bytes: a plain base, never written)
        if (have == 4u) {
            __builtin_memcpy(&x, rd + idx, 4);
        } else {
            for (u32 b = 0; b < have; ++b) x = (x & ~(0xFFu << (8u * b))) | ((u32)rd[idx + b] << (8u * b));
        }
        const u32 w = (u32)lut[x & 0xFFu] | ((u32)lut[(x >> 8) & 0xFFu] << 8) | ((u32)lut[(x >> 16) & 0xFFu] << 16) |
                      ((u32)lut[x >> 24] << 24);
        bad |= w & 0x04040404u;  // (absent bytes are 'A': code 0; removed bytes are 0x80)
        if (packed && __ballot((w & 0x80808080u) != 0u) == 0ull) {
            uint8_t* dst = codes + nb + 4u * lane;
            if (have == 4u) *reinterpret_cast<u32*>(dst) = w;
            else for (u32 b = 0; b < have; ++b) dst[b] = (uint8_t)(w >> (8u * b));
            nb += min(256u, to - base);
        } else {
            packed = false;
            u32 cnt = 0;
#pragma unroll
            for (u32 b = 0; b < 4u; ++b) cnt += (b < have && !((w >> (8u * b)) & 0x80u)) ? 1u : 0u;
            const u32 incl = wave_incl_scan(cnt);
            u32 pos = nb + incl - cnt;
#pragma unroll
            for (u32 b = 0; b < 4u; ++b)
                if (b < have && !((w >> (8u * b)) & 0x80u)) codes[pos++] = (uint8_t)(w >> (8u * b));
            nb += (u32)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    return nb;
}

// 4-bit packed input (skx_stream_set_packed_input): nibble i of the stream = base i, low nibble of a byte first; 0..3 = A C G T,
// anything else = a retained non-ACGT byte (N).  Whitespace does not exist in this format (skx_pack_bases drops it).
__device__ __forceinline__ u32 packed_code(const uint8_t* __restrict__ bases, u64 nib) {
    const u32 v = ((u32)bases[nib >> 1] >> (4u * (u32)(nib & 1ull))) & 0xFu;
    return v > 3u ? 4u : v;
}
// wave_normalise4 for packed input: bases [nib0 + from, nib0 + to) -> codes[nb ...] (nb a multiple of 4: whole words)
__device__ __forceinline__ u32 wave_normalise_packed(const uint8_t* __restrict__ bases, u64 nib0, u32 from, u32 to, uint8_t* codes,
                                                     u32 nb, u32 lane, u32& bad) {
    for (u32 base = from; base < to; base += 256u) {
        const u32 idx = base + 4u * lane;
        const u32 have = idx < to ? min(4u, to - idx) : 0u;
        u32 w = 0;
        if (have) {
            const u64 p = nib0 + idx;
            u32 x = 0;
            if (idx + 8u <= to) {  // (8 nibbles of the read from here on: one dword covers the 4 wanted at either parity)
                __builtin_memcpy(&x, bases + (p >> 1), 4);
                x >>= 4u * (u32)(p & 1ull);
            } else if (have == 4u) {  // 4 nibbles from bit 4 * (p & 1) of two or three bytes
                const uint8_t* src = bases + (p >> 1);
                x = (u32)src[0] | ((u32)src[1] << 8) | ((p & 1ull) ? ((u32)src[2] << 16) : 0u);
                x >>= 4u * (u32)(p & 1ull);
            } else {
                for (u32 b = 0; b < have; ++b) x |= (((u32)bases[(p + b) >> 1] >> (4u * (u32)((p + b) & 1ull))) & 0xFu) << (4u * b);
            }
            w = (x & 0xFu) | ((x & 0xF0u) << 4) | ((x & 0xF00u) << 8) | ((x & 0xF000u) << 12);
            const u32 inv = ((w >> 1) | w) & 0x04040404u;  // nibble > 3
            w = ((w & 0x03030303u) & ~((inv >> 2) * 3u)) | inv;
            bad |= inv;  // (w holds exactly the `have` nibbles of the read: the rest is zero)
        }
        uint8_t* dst = codes + nb + 4u * lane;
        if (have == 4u) *reinterpret_cast<u32*>(dst) = w;
        else for (u32 b = 0; b < have; ++b) dst[b] = (uint8_t)(w >> (8u * b));
        nb += min(256u, to - base);
    }
    return nb;
}

// ---- long reads: split over waves ---------------------------------------------------------------------------------
// One wavefront per read serialises a long read on one wave: a 50 kb read is 25 chunks in a row next to seven other waves on
// its SIMD (~2 ms at C4, longer than the rest of the batch takes).  In production mode (INRANGE) a read of more than
// kLongSplit raw bytes is therefore cut into SEGMENTS = the chunks of kSketchCap raw bytes the serial loop would walk, one
// wave per segment: the wave finds the k-1 retained codes in front of its chunk itself (scanning backwards over whitespace),
// hashes exactly the windows that END in the chunk -- the same windows, hence the same multiset of hashes, as the serial
// loop -- and leaves its in-range hashes in the segment's slot (kSegSlots entries; production keeps ~7 per chunk).  One wave
// per long read then gathers the slots and finishes like any other read: sort, distinct, truncate to s, membership filter
// (sketch_merge_kernel).  A slot that overflows, or more than kSketchCap hashes in all, sends the read to the block sketcher.
constexpr u32 kLongSplit = 4u * kSketchCap;  // raw bytes; reads up to here stay on one wave (<= 4 chunks)
constexpr u32 kSegSlots = 64;
constexpr u32 kSegOvf = 0xFFFFFFFFu;
// the same test in batch_check_kernel (which lists the long reads), the read waves (which skip them) and the segment waves
__device__ __forceinline__ bool is_split_long(u64 o0, u64 o1, u64 lo, u64 n_bases) {
    return o0 >= lo && o1 >= o0 && o1 - lo <= n_bases && o1 - o0 > (u64)kLongSplit;
}

// ---- k-mer prefilter (struct KmerFilter: skx_kernels.hpp) ------------------------------------------------------------
constexpr u32 kKmerMix = 0x9E3779B1u;  // odd: a bijection of the 32-bit code space
constexpr u32 kPfQueue = 128;          // survivors (canonical codes) a wave parks in LDS before it hashes them, 64 at a time
__device__ __forceinline__ bool kmer_filter_hit(const u32* __restrict__ words, u32 shift, u32 code) {
    const u32 m = code * kKmerMix;
    const u32 w = words[m >> shift];
    const u32 mask = (1u << (m & 31u)) | (1u << ((m >> 5) & 31u));
    return (w & mask) == mask;
}
// the 16 ASCII bytes of a 2-bit packed 16-mer (first base in the two highest bits), as murmur3's two little-endian words
__device__ __forceinline__ void ascii16_from_code(u32 code, u64& w0, u64& w1) {
    u32 a[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const u32 b = code >> (24 - 8 * g);  // bases 4g .. 4g+3 in the low byte, first base in its two highest bits
        const u32 sel = ((b >> 6) & 3u) | (((b >> 4) & 3u) << 8) | (((b >> 2) & 3u) << 16) | ((b & 3u) << 24);
        a[g] = __builtin_amdgcn_perm(0u, 0x54474341u, sel);  // "ACGT"[selector byte]
    }
    w0 = make_u64(a[0], a[1]);
    w1 = make_u64(a[2], a[3]);
}
__device__ __forceinline__ u32 revcomp16(u32 x) {
    u32 r = __brev(~x);  // bit-reversed complement: the 2-bit groups are in reverse order, each with its two bits swapped
    return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}
// One pass over all 2^32 codes; a code is canonical when it does not exceed its reverse complement (palindromes: equal,
// the same bytes either way).  INSERT = false: count the k-mers whose hash passes; true: set their two bits.
template <bool INSERT>
__global__ __launch_bounds__(256) void kmer_filter_build_kernel(u64 seed, u64 max_ref, const u64* __restrict__ filt, u32 filt_shift,
                                                                u32* __restrict__ n_keys, u32* __restrict__ words, u32 shift) {
    constexpr u32 kPerThread = 256;
    const u32 base = (blockIdx.x * 256u + threadIdx.x) * kPerThread;  // (2^32 codes = 65 536 blocks x 256 threads x 256)
    u32 mine = 0;
    for (u32 i = 0; i < kPerThread; ++i) {
        const u32 x = base + i;
        if (x > revcomp16(x)) continue;
        u64 w0, w1;
        ascii16_from_code(x, w0, w1);
        const u64 h = murmur3_h1_16<false>(w0, w1, seed);
        if (h > max_ref) continue;
        if (!filter_hit(filt, filt_shift, h)) continue;
        if (INSERT) {
            const u32 m = x * kKmerMix;
            atomicOr(&words[m >> shift], (1u << (m & 31u)) | (1u << ((m >> 5) & 31u)));
        } else {
            ++mine;
        }
    }
    if (!INSERT) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mine += (u32)__shfl_xor((int)mine, d, 64);
        if (lane_id() == 0 && mine) atomicAdd(n_keys, mine);
    }
}

// the tail of every sketch: `m` hashes in the wave's LDS buffer (any order, m <= HCAP) -> ascending, distinct, truncated to
// s, (production) only those some genome holds -> out_sk row r, out_len[r], out_cnt_in[r]
// POOL MODE (sk_stride == 0; production batches): rows are not max_reads x s entries -- a read keeps a handful of pairs --
// but exact-size reservations out of one pool: out_sk = the pool, out_len[r] receives the row's start in it (the sketch
// length has no consumer in production mode), chk[11] is the bump counter, pool_cap the pool's entries.  A reservation that
// does not fit raises chk[6] |= 4 and writes nothing; chk[11] still sums every request, so the host knows how much to allocate
// before it repeats the batch (rare: a pool holds 16 pairs per read of the largest batch, C2 needs 2.4).
// The pool starts with a FIXED part: kRowFixed entries per read (row r at r * kRowFixed) -- a row that fits takes it without any
// atomic (a returning atomic is a ~2 us round trip next to the other streams' kernels: per read, that was 5 % of the C2 step);
// only longer rows reserve behind it, in the sub-pools.
constexpr u32 kRowFixed = 16;
__device__ __forceinline__ u32 pool_reserve(u32* __restrict__ chk, u32 pool_fixed, u32 pool_cap, u32 want /* wave-uniform */, bool& ok) {
    const u32 part = blockIdx.x % kPoolParts, part_cap = (pool_cap - pool_fixed) / kPoolParts;
    u32 off = 0;
    if (lane_id() == 0 && want) off = atomicAdd(&chk[16u + 16u * part], want);
    off = __builtin_amdgcn_readfirstlane(off);
    ok = off + want <= part_cap && off + want >= off;
    if (!ok && lane_id() == 0) atomicOr(&chk[6], 4u);
    return pool_fixed + part * part_cap + off;
}
__device__ __forceinline__ u32 pool_row(u32* __restrict__ chk, u32 pool_fixed, u32 pool_cap, u32 r, u32 want, bool& ok) {
    if (want <= kRowFixed && (r + 1u) * kRowFixed <= pool_fixed) { ok = true; return r * kRowFixed; }
    return pool_reserve(chk, pool_fixed, pool_cap, want, ok);
}
template <int HCAP, bool INRANGE>
__device__ __forceinline__ void sketch_finish(u64* hashes, u32 m, u32 r, u32 s, u64 max_ref, u64* __restrict__ out_sk,
                                              u32 sk_stride, u32* __restrict__ out_len, u32* __restrict__ out_cnt_in,
                                              const u64* __restrict__ filt, u32 filt_shift, u32* __restrict__ chk = nullptr,
                                              u32 pool_cap = 0, u32 pool_fixed = 0) {
    const u32 lane = lane_id();
    const u64 lt = lanemask_lt();
    // At most one hash per lane (production: a 1.5 kb read keeps ~8): sort by counting.  Lane l holds hash l; for every j the
    // wave sees hash j as a scalar, finds the lanes holding the same value (ballot) -- j counts only if it is the first of
    // them -- and every lane holding something larger moves up one place: ~5 VALU instructions per hash instead of the
    // ~300 of a 64-element bitonic network through LDS.  Leaves the DISTINCT hashes, ascending, at hashes[0 .. m).
    if (m <= 64u) {
        const u64 h = lane < m ? hashes[lane] : kPad;
        u32 rank = 0;
        u64 heads = 0;
        for (u32 j = 0; j < m; ++j) {
            const u64 hj = make_u64((u32)__builtin_amdgcn_readlane((int)(u32)h, (int)j),
                                    (u32)__builtin_amdgcn_readlane((int)(u32)(h >> 32), (int)j));
            const u64 eq = __ballot(h == hj);
            if ((u32)__builtin_ctzll(eq) == j) {  // (lane j itself is in eq: never zero)
                heads |= 1ull << j;
                rank += hj < h ? 1u : 0u;
            }
        }
        wave_sync();
        if ((heads >> lane) & 1ull) hashes[rank] = h;
        m = (u32)__popcll(heads);
        wave_sync();
    }
    // pad to a power of two (>= 64) for the bitonic network
    u32 p2 = 64;
    while (p2 < m) p2 <<= 1;
    if (m > 64u) {
        for (u32 i = m + lane; i < p2; i += 64u) hashes[i] = kPad;
        wave_sync();
    }

    // bitonic sort ascending
    if (m > 64u) {
        for (u32 size = 2; size <= p2; size <<= 1) {
            for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
                for (u32 t = lane; t < (p2 >> 1); t += 64u) {
                    const u32 i = 2u * t - (t & (stride - 1u));
                    const u32 j = i + stride;
                    const bool up = (i & size) == 0u;
                    const u64 a = hashes[i], b = hashes[j];
                    if ((a > b) == up) { hashes[i] = b; hashes[j] = a; }
                }
                wave_sync();
            }
        }
    }

    // distinct, truncate to s, count the part that can meet the reference at all.
    // INRANGE with a membership bitmap (production): of the hashes that made it into the bottom-s only those some genome
    // holds are written -- strictly AFTER the truncation: a hash ranked beyond s is not part of the sketch even if
    // everything before it is dropped (reads with more distinct in-range hashes than s, e.g. small s).
    u32 outn = 0, cin = 0, wrote = 0;
    if (INRANGE && filt != nullptr && sk_stride == 0u) {
        // pool mode: count what the row will hold, reserve exactly that, write it
        u64* out = nullptr;
        u32 off = 0;
        bool ok = true;
        if (m <= 64u) {  // (the usual case: one step, the membership lookups are done once)
            const bool v = lane < m;
            const u64 h = v ? hashes[lane] : 0;
            const bool take = v && lane < s;  // (hashes[0 .. m) are distinct here: the counting sort dropped the duplicates)
            const bool keep = take && filter_hit(filt, filt_shift, h);
            const u64 km = __ballot(keep);
            wrote = (u32)__popcll(km);
            off = pool_row(chk, pool_fixed, pool_cap, r, wrote, ok);
            if (ok && keep) out_sk[(size_t)off + __popcll(km & lt)] = h;
        } else {
            for (int pass = 0; pass < 2; ++pass) {
                outn = 0; wrote = 0;
                for (u32 base = 0; base < m && outn < s; base += 64u) {
                    const u32 i = base + lane;
                    const bool v = i < m;
                    const u64 h = v ? hashes[i] : 0;
                    const bool head = v && (i == 0 || hashes[i - 1] != h);
                    const u64 mask = __ballot(head);
                    const u32 pos = outn + __popcll(mask & lt);
                    const bool keep = head && pos < s && filter_hit(filt, filt_shift, h);
                    const u64 km = __ballot(keep);
                    if (pass == 1 && keep) out[wrote + __popcll(km & lt)] = h;
                    wrote += __popcll(km);
                    outn += __popcll(mask);
                }
                if (pass == 0) {
                    off = pool_row(chk, pool_fixed, pool_cap, r, wrote, ok);
                    if (!ok) break;
                    out = out_sk + (size_t)off;
                }
            }
        }
        if (lane == 0) {
            out_len[r] = off;                 // the row's place in the pool
            out_cnt_in[r] = ok ? wrote : 0u;  // (a batch with a failed reservation is repeated with a larger pool)
        }
        return;
    }
    u64* out = out_sk + (size_t)r * sk_stride;
    for (u32 base = 0; base < m && outn < s; base += 64u) {
        const u32 i = base + lane;
        const bool v = i < m;
        const u64 h = v ? hashes[i] : 0;
        const bool head = v && (i == 0 || hashes[i - 1] != h);
        const u64 mask = __ballot(head);
        const u32 pos = outn + __popcll(mask & lt);
        const bool take = head && pos < s;
        if (INRANGE && filt != nullptr) {
            bool keep = false;
            if (take) keep = filter_hit(filt, filt_shift, h);
            const u64 km = __ballot(keep);
            if (keep) out[wrote + __popcll(km & lt)] = h;
            wrote += __popcll(km);
        } else {
            if (take) out[pos] = h;
            cin += __popcll(__ballot(take && h <= max_ref));
        }
        outn += __popcll(mask);
    }
    if (lane == 0) {
        out_len[r] = min(outn, s);
        out_cnt_in[r] = (INRANGE && filt != nullptr) ? wrote : cin;
    }
}

// SEG: the wave hashes ONE chunk (segment seg_i) of a long read and leaves its in-range hashes in seg_h / seg_cnt.
template <int KT, int HCAP, bool INRANGE, bool SEG = false>
__device__ __forceinline__ void sketch_one_read(unsigned char* smem, u32 r, const uint8_t* __restrict__ bases,
                                                const u64* __restrict__ offsets, u32 k_rt, u64 seed, u32 s, u64 max_ref,
                                                u64* __restrict__ out_sk, u32 sk_stride, u32* __restrict__ out_len,
                                                u32* __restrict__ out_cnt_in, u32* __restrict__ retry, u32* __restrict__ big,
                                                const u64* __restrict__ filt, u32 filt_shift, u64 n_bases,
                                                u32* __restrict__ chk, const SketchTables<KT>* tb, bool packed,
                                                bool split_long = false, u32 seg_i = 0, u64* __restrict__ seg_h = nullptr,
                                                u32* __restrict__ seg_cnt = nullptr, KmerFilter kf = KmerFilter{nullptr, 0u},
                                                u32 pool_cap = 0, u32 pool_fixed = 0) {
    static_assert(!SEG || INRANGE, "segments exist in production mode only");
    constexpr u32 CAP = kSketchCap;
    constexpr u32 kPerWave = HCAP * 8 + CAP + 128;  // (hash slots; 32: carry of k-1 codes, ending word-aligned; a chunk; 64 codes of padding)
    constexpr u32 kChunkAt = 32;                    // a chunk's codes start here; the carried k-1 end here
    const u32 wv = threadIdx.x >> 6, lane = lane_id();
    u64* hashes = reinterpret_cast<u64*>(smem + (size_t)wv * kPerWave);
    uint8_t* codes = smem + (size_t)wv * kPerWave + HCAP * 8;
    // the prefilter's survivors: behind all waves' buffers, and only there when the launch carries a k-mer filter (launch_sketch)
    u32* pq = reinterpret_cast<u32*>(smem + (size_t)(blockDim.x >> 6) * kPerWave + (size_t)wv * kPfQueue * 4);
    const u32 k = KT > 0 ? (u32)KT : k_rt;
    const u64 o0 = offsets[r], o1 = offsets[r + 1];
    // the caller vouches for n_bases bytes from offsets[0] on: a read reaching outside is never touched (flagged in
    // chk[6]; the push then fails with SKX_ERR_INVALID instead of faulting)
    const u64 lo = offsets[0];
    if (o0 < lo || o1 < o0 || o1 - lo > n_bases) {
        if constexpr (SEG) {
            if (lane == 0) *seg_cnt = 0;  // (cannot happen: batch_check_kernel lists only reads inside the batch)
            return;
        }
        if (lane == 0) {
            out_len[r] = 0; out_cnt_in[r] = 0;
            if (chk) atomicOr(&chk[6], 1u);
        }
        return;
    }
    // a long read of a production batch belongs to the segment waves and sketch_merge_kernel
    if (!SEG && INRANGE && split_long && o1 - o0 > (u64)kLongSplit) return;
    const u32 lraw = (u32)(o1 - o0);
    if (!INRANGE && lraw > CAP + k - 1u) {  // a full sketch of more k-mers than the buffer holds: the block sketcher
        if (lane == 0) { out_len[r] = kSketchRetry; out_cnt_in[r] = 0; list_append(big, r); }
        return;
    }
    const uint8_t* rd = bases + o0;
    const u64 lt = lanemask_lt();
    const unsigned char* lut = tb->lut;
    // (reads beyond kLongSplit only get here when they are not split: full sketches, the 2048-slot retry variant)
    if (lraw > 4u * CAP) __builtin_amdgcn_s_setprio(2);

    // k-mer prefilter (production, k = 16): a read with at most s windows cannot be truncated, so only windows whose k-mer
    // the reference can hold at all need their hash (KmerFilter, skx_kernels.hpp); longer reads take the plain loop
    const bool use_pf = KT == 16 && INRANGE && kf.words != nullptr && filt != nullptr && (u64)lraw <= (u64)s + 15u;
    u32 m = 0;    // hashes collected so far (any order: they are sorted afterwards); wave-uniform (kept scalar)
    u32 ovf = 0;  // the hash buffer overflowed: the read is handed on after the loop (wave-uniform, scalar)
    // (in production mode 99.5 % of the hashes are out of range: three iterations in four keep nothing at all and skip
    // everything below the ballot)
    auto append = [&](bool valid, u64 h) {
        if (INRANGE) valid = valid && (h <= max_ref);
        const u64 mask = __builtin_amdgcn_ballot_w64(valid);
        if (mask) {
            const u32 cnt = (u32)__popcll(mask);
            const u32 over = __builtin_amdgcn_readfirstlane(m + cnt > (u32)HCAP ? 1u : 0u);
            ovf |= over;
            if (!over && valid) hashes[m + __popcll(mask & lt)] = h;
            m = __builtin_amdgcn_readfirstlane(m + cnt);
        }
    };

    // the same with the validity as a wave mask (production: the in-range test is the only per-lane compare left)
    auto append_masked = [&](u64 vmask, u64 h) {
        const u64 mask = INRANGE ? (vmask & __builtin_amdgcn_ballot_w64(h <= max_ref)) : vmask;
        if (mask) {
            const u32 cnt = (u32)__popcll(mask);
            const u32 over = __builtin_amdgcn_readfirstlane(m + cnt > (u32)HCAP ? 1u : 0u);
            ovf |= over;
            if (!over && ((mask >> lane) & 1ull)) hashes[m + __popcll(mask & lt)] = h;
            m = __builtin_amdgcn_readfirstlane(m + cnt);
        }
    };

    u32 carry = 0;  // codes kept from the previous chunk at codes[kChunkAt - carry .. kChunkAt)
    bool carry_bad = false;  // ... one of them is not a base (wave-uniform)
    u32 cbase0 = 0;
    if constexpr (SEG) {
        cbase0 = seg_i * CAP;
        if (cbase0 >= lraw) {  // (never: the segments of a read cover exactly its chunks)
            if (lane == 0) *seg_cnt = 0;
            return;
        }
        if (cbase0 > 0u) {
            // the last k-1 RETAINED codes in front of the chunk (what the serial loop would have carried to here)
            const u32 want = k - 1u;
            if (packed) {  // (no whitespace in this format: the k-1 nibbles in front)
                const u32 pc = lane < want ? packed_code(bases, o0 + cbase0 - want + lane) : 0u;
                if (lane < want) codes[kChunkAt - want + lane] = (uint8_t)pc;
                carry_bad = __builtin_amdgcn_ballot_w64(pc > 3u) != 0ull;
                carry = want;  // (cbase0 >= CAP > k - 1)
            } else {
                u32 got = 0;
                for (u32 end = cbase0; end > 0u && got < want;) {
                    const u32 beg = end > 64u ? end - 64u : 0u;
                    const u32 idx = beg + lane;
                    const u32 c = idx < end ? (u32)lut[rd[idx]] : 0x80u;
                    const bool kept = !(c & 0x80u);
                    const u64 km = __ballot(kept);
                    const u64 above = lane == 63u ? 0ull : (~0ull << (lane + 1u));
                    const u32 rk = got + (u32)__popcll(km & above);  // retained codes between this byte and the chunk
                    if (kept && rk < want) codes[kChunkAt - 1u - rk] = (uint8_t)c;
                    carry_bad = carry_bad || __builtin_amdgcn_ballot_w64(kept && rk < want && c > 3u) != 0ull;
                    got = __builtin_amdgcn_readfirstlane(min(want, got + (u32)__popcll(km)));
                    end = beg;
                }
                carry = got;
            }
            wave_sync();
        }
    }
    for (u32 cbase = cbase0;; cbase += CAP) {
        const u32 cend = min(lraw, cbase + CAP);
        // 1. normalise the chunk behind the carried codes
        const u32 cs = kChunkAt - carry;  // first code of the buffer
        u32 bad_bits = 0;
        const u32 ne = __builtin_amdgcn_readfirstlane(packed ? wave_normalise_packed(bases, o0, cbase, cend, codes, kChunkAt, lane, bad_bits)
                                                             : wave_normalise4(rd, cbase, cend, codes, kChunkAt, lane, lut, bad_bits));  // one past the last
        // (no N, no IUPAC code anywhere in the buffer -- the usual case: every window inside it is valid)
#ifdef SKX_SK_FORCE_CLEAN  /* diagnostic builds only (instruction counts of the specialisation on reads known to be clean) */
        const bool all_bases = true;
#else
        const bool all_bases = !carry_bad && __builtin_amdgcn_ballot_w64(bad_bits != 0u) == 0ull;
#endif
        codes[ne + lane] = 4;  // 64 invalid codes behind the chunk: lanes past the last window read them unconditionally
        wave_sync();
        // 2. canonical k-mer hashes of the windows that END in this chunk, compacted
        const u32 nb = ne - cs;
        const u32 nk = nb >= k ? nb - k + 1u : 0u;
        if constexpr (KT == 16) {
            // Each lane walks a contiguous run of positions with ROLLING windows: the 2-bit forward / reverse-
            // complement codes (for the canonical choice) and their 16 ASCII bytes as two little-endian words each
            // (the murmur3 block) -- one LDS byte read and a few shifts per k-mer instead of rebuilding all 16 bases.
            const u32 run = __builtin_amdgcn_readfirstlane((nk + 63u) / 64u);
            const u32 p0 = cs + lane * run;
            u32 fwd = 0, rc = 0;            // 16 bases x 2 bits: forward (first base highest) / reverse complement
            u32 clean = 0;                  // consecutive valid bases ending at the newest one
            // (rounds 1-3 also rolled the 16 ASCII bytes of both strands -- the murmur3 block; the first-stage products now
            // come out of the tables indexed by the canonical 2-bit window: SketchTables)
            const unsigned char* t1b = reinterpret_cast<const unsigned char*>(tb->t1);
            const unsigned char* t2b = reinterpret_cast<const unsigned char*>(tb->t2);
#if SKX_SK_LOTAB
            const unsigned char* t1l = reinterpret_cast<const unsigned char*>(tb->t1lo);
            const unsigned char* t2l = reinterpret_cast<const unsigned char*>(tb->t2lo);
            constexpr u32 kLoShift = 2;
#else
            const unsigned char *t1l = t1b, *t2l = t2b;
            constexpr u32 kLoShift = 3;
#endif
            if (nk) {
                // (a window that reaches past the chunk takes in padding codes, which reset `clean`: no bounds tests --
                // positions are clamped into the padded buffer, p0 + t + 15 <= nb + 63 whenever the window can be valid)
                const u32 lim = ne + 63u;
                // Window state after the first 15 codes of the run, straight from the code bytes (15 rolling steps cost
                // ~150 instructions per chunk, this ~45): five aligned words funnel-shifted to the run's start; the 2-bit
                // packings and the validity bits by multiplications that gather one field per byte into the top byte
                // (fields never overlap: no carries).  A run that starts in the padding is clamped into it (every window
                // it sees is invalid either way).
                {
                    const u32 al = min(p0 & ~3u, (ne + 44u) & ~3u);
                    const u32* cw = reinterpret_cast<const u32*>(codes + al);
                    const u32 w0 = cw[0], w1 = cw[1], w2 = cw[2], w3 = cw[3], w4 = cw[4];
                    const u32 sh = p0 & 3u;  // (clamped lanes: any shift reads padding)
                    const u32 b0 = __builtin_amdgcn_alignbyte(w1, w0, sh), b1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
                    const u32 b2 = __builtin_amdgcn_alignbyte(w3, w2, sh), b3 = __builtin_amdgcn_alignbyte(w4, w3, sh);
                    const u32 x0 = b0 & 0x03030303u, x1 = b1 & 0x03030303u, x2 = b2 & 0x03030303u, x3 = b3 & 0x03030303u;
                    // 2-bit packings: fwd = sum c_j << 2 (14 - j), rc = sum (3 - c_j) << (2 + 2 j), j = 0..14
                    constexpr u32 kF = 0x40100401u, kR = 0x01041040u;
                    const u32 fw16 = (((x0 * kF) >> 24) << 24) | (((x1 * kF) >> 24) << 16) | (((x2 * kF) >> 24) << 8) | ((x3 * kF) >> 24);
                    fwd = fw16 >> 2;
                    const u32 rc16 = (((x0 ^ 0x03030303u) * kR) >> 24) | ((((x1 ^ 0x03030303u) * kR) >> 24) << 8) |
                                     ((((x2 ^ 0x03030303u) * kR) >> 24) << 16) | ((((x3 ^ 0x03030303u) * kR) >> 24) << 24);
                    rc = rc16 << 2;
                    // clean = valid codes in a row ending at c_14 (an invalid code has bit 2 set)
                    constexpr u32 kV = 0x04081020u;
                    const u32 inv = (((b0 & 0x04040404u) * kV) >> 28) | ((((b1 & 0x04040404u) * kV) >> 28) << 4) |
                                    ((((b2 & 0x04040404u) * kV) >> 28) << 8) | ((((b3 & 0x04040404u) * kV) >> 28) << 12);
                    clean = (u32)__builtin_clz(((inv & 0x7FFFu) << 17) | 0x10000u);
                }
                u32 cnext = codes[min(p0 + 15u, lim)];  // (the next code is requested one iteration ahead of its use)
                if (use_pf) {
                    // 2-bit windows only; the table word of window t is requested one iteration before it is tested
                    u32 qn = 0;  // survivors parked in pq (wave-uniform)
                    auto drain = [&]() {
                        wave_sync();
                        for (u32 i0 = 0; i0 < qn; i0 += 64u) {
                            const bool v = i0 + lane < qn;
                            const u32 code = v ? pq[i0 + lane] : 0u;
                            u64 a0, a1;
                            ascii16_from_code(code, a0, a1);
                            const u64 h = seed == 0 ? murmur3_h1_16<true>(a0, a1, 0) : murmur3_h1_16<false>(a0, a1, seed);
                            append(v, h);
                        }
                        qn = 0;
                        wave_sync();
                    };
                    auto test = [&](bool valid, u32 word, u32 mix, u32 canon) {
                        const u32 mask = (1u << (mix & 31u)) | (1u << ((mix >> 5) & 31u));
                        const bool hit = valid && (word & mask) == mask;
                        const u64 hm = __ballot(hit);
                        if (hm) {
                            if (hit) pq[qn + (u32)__popcll(hm & lt)] = canon;
                            qn = __builtin_amdgcn_readfirstlane(qn + (u32)__popcll(hm));
                            if (qn > kPfQueue - 64u) drain();
                        }
                    };
                    bool pvalid = false;
                    u32 pword = 0, pmix = 0, pcanon = 0;
                    for (u32 t = 0; t < run; ++t) {
                        const u32 ccur = cnext;
                        cnext = codes[min(p0 + t + 16u, lim)];
                        clean = (ccur >> 2) ? 0u : clean + 1u;
                        const u32 c2 = ccur & 3u;
                        fwd = (fwd << 2) | c2;
                        rc = __builtin_amdgcn_alignbit(c2 ^ 3u, rc, 2);
                        const u32 canon = min(fwd, rc);
                        const u32 mix = canon * kKmerMix;
                        const u32 word = kf.words[mix >> kf.shift];
                        test(pvalid, pword, pmix, pcanon);
                        pvalid = clean >= 16u; pword = word; pmix = mix; pcanon = canon;
                    }
                    test(pvalid, pword, pmix, pcanon);
                    if (qn) drain();
                } else {
                    // One k-mer per lane and iteration, software-pipelined by hand: the table reads of window t + 1 are issued
                    // before window t goes through murmur3 (their LDS latency hides under its ~60 instructions).  Two
                    // specialisations, both wave-uniform: the seed (0 = sketchy's default folds two xors and an add away)
                    // and `all_bases` (no invalid code in the whole buffer: a window is valid iff it lies inside it -- one
                    // compare against the lane's window count instead of the four-instruction run-length counter).
                    const unsigned char* cp = codes + p0 + 16u;  // (p0 + t + 16 <= ne + 63 for every lane and t < run: inside the padding)
                    // (hash <= max_ref needs hi(hash) <= hi(max_ref); hi(hash) = hi(x1) + hi(x2) + carry: in range only if that sum
                    // + 1 lies in [0, hi(max_ref) + 1] modulo 2^32 -- with hi(max_ref) all ones or all ones but one every lane passes)
                    const u32 max_hi = (u32)(max_ref >> 32), max_hi1 = max_hi == 0xFFFFFFFFu ? max_hi : max_hi + 1u;
                    const u32 mine = nk > lane * run ? nk - lane * run : 0u;  // windows of this lane's run that lie inside the buffer
                    auto hash_run = [&](auto seed0_c, auto clean_c) {
                        constexpr bool SEED0 = decltype(seed0_c)::value, CLEAN = decltype(clean_c)::value;
                        u64 a = 0, b = 0;
                        u32 ah = 0, bh = 0;
                        u64 vnext = 0;  // lanes whose window is valid, as a wave mask (kept scalar: it meets the in-range ballot with one s_and)
                        auto advance = [&](u32 t) {  // window t: roll it in, request its products
                            const u32 ccur = cnext;
                            cnext = cp[t];
                            if constexpr (CLEAN) {
                                // (the only codes above 3 are the padding behind the buffer: they spoil windows that are
                                // invalid by position, in this lane's run and nobody else's -- no masking)
                                vnext = __builtin_amdgcn_ballot_w64(t < mine);
                                fwd = (fwd << 2) | ccur;
                            } else {
                                clean = (ccur >> 2) ? 0u : clean + 1u;
                                vnext = __builtin_amdgcn_ballot_w64(clean >= 16u);
                                fwd = (fwd << 2) | (ccur & 3u);
                            }
                            rc = __builtin_amdgcn_alignbit(~ccur, rc, 2);  // (rc >> 2) | (complement << 30): the low two bits of ~code
                            // canonical = bytewise min(forward, reverse complement) = the smaller 2-bit packing (A < C < G < T;
                            // a palindrome: the same bytes either way); its four bytes index the product tables
                            const u32 canon = min(fwd, rc);
                            a = *reinterpret_cast<const u64*>(t1b + byte_shl<3, 3>(canon));
                            ah = *reinterpret_cast<const u32*>(t1l + byte_shl<2, kLoShift>(canon));
                            b = *reinterpret_cast<const u64*>(t2b + byte_shl<1, 3>(canon));
                            bh = *reinterpret_cast<const u32*>(t2l + byte_shl<0, kLoShift>(canon));
                        };
                        advance(0u);
                        auto finish = [&](u32 t) {  // window t through murmur3 while window t + 1 is on its way
                            const u64 p1 = make_u64((u32)a, (u32)(a >> 32) + ah), p2 = make_u64((u32)b, (u32)(b >> 32) + bh);
                            const u64 vcur = vnext;
                            advance(t + 1u);  // (one window past the run at the end: harmless -- padding codes, any table index is inside the table)
                            if constexpr (INRANGE && SKX_SK_EARLY) {
                                // production keeps hashes <= max_ref only -- 0.4 % of them: stop in front of fmix64's last
                                // xor-shift (it touches the low word only), bound the hash's high word with one add, and
                                // finish the low word for the lanes that can be in range (one iteration in five has any)
                                u64 x1, x2;
                                murmur3_h1_16_pre_split<SEED0>(p1, p2, seed, x1, x2);
                                const u32 hs1 = (u32)(x1 >> 32) + (u32)(x2 >> 32) + 1u;  // high word of the sum, + 1: the carry out of the low words is 0 or 1
                                const u64 maybe = vcur & __builtin_amdgcn_ballot_w64(hs1 <= max_hi1);
                                if (maybe) append_masked(maybe, murmur3_finish_pair(x1, x2));
                            } else {
                                append_masked(vcur, murmur3_h1_16_pre<SEED0>(p1, p2, seed));
                            }
                        };
                        u32 t = 0;
                        for (; t + 1u < run; t += 2u) { finish(t); finish(t + 1u); }  // (two per trip: the loop-carried products change registers instead of being copied)
                        if (t < run) finish(t);
                    };
                    if (seed == 0) {
                        if (all_bases) hash_run(std::true_type{}, std::true_type{}); else hash_run(std::true_type{}, std::false_type{});
                    } else {
                        if (all_bases) hash_run(std::false_type{}, std::true_type{}); else hash_run(std::false_type{}, std::false_type{});
                    }
                }
            }
        } else {
            for (u32 base = 0; base < nk; base += 64u) {
                const u32 p = base + lane;
                bool valid = p < nk;
                u64 fwd = 0, rc = 0;
                u32 bad = 0;
                if (valid) {
#pragma unroll
                    for (u32 j = 0; j < 32u; ++j) {
                        if (j < k) {
                            u32 c = codes[cs + p + j];
                            bad |= c >> 2;
                            c &= 3u;
                            fwd = (fwd << 2) | c;
                            rc |= (u64)(3u - c) << (2 * j);
                        }
                    }
                }
                valid = valid && (bad == 0);
                const u64 h = hash_canonical_packed<KT>(fwd < rc ? fwd : rc, k, seed);
                append(valid, h);
            }
        }
        if constexpr (SEG) break;  // one chunk per segment wave
        if (ovf) {  // hand the read on (fast variant -> 2048-slot variant -> block sketcher)
            if (lane == 0) {
                out_len[r] = kSketchRetry; out_cnt_in[r] = 0;
                list_append(HCAP < (int)CAP ? retry : big, r);
            }
            __builtin_amdgcn_s_setprio(0);
            return;
        }
        if (cend >= lraw) break;
        // 3. the last k-1 codes open the next chunk
        const u32 keep = min(nb, k - 1u);
        const u32 cv = lane < keep ? (u32)codes[ne - keep + lane] : 0u;
        wave_sync();
        if (lane < keep) codes[kChunkAt - keep + lane] = (uint8_t)cv;
        carry = keep;
        carry_bad = __builtin_amdgcn_ballot_w64(cv > 3u) != 0ull;
    }
    if constexpr (SEG) {
        // the segment's in-range hashes (unsorted, duplicates included) go to its slot; the merge wave does the rest
        wave_sync();
        if (ovf || m > kSegSlots) {
            if (lane == 0) *seg_cnt = kSegOvf;
        } else {
            if (lane < m) seg_h[lane] = hashes[lane];
            if (lane == 0) *seg_cnt = m;
        }
        return;
    }
    sketch_finish<HCAP, INRANGE>(hashes, m, r, s, max_ref, out_sk, sk_stride, out_len, out_cnt_in, filt, filt_shift, chk, pool_cap, pool_fixed);
    __builtin_amdgcn_s_setprio(0);
}
// from_list = 0: wave w of the grid sketches read w.  from_list = 1: a small fixed grid walks the reads an earlier
// variant appended to `retry` -- usually none, and then the launch costs a few microseconds instead of one nearly
// empty wave per read of the batch.  from_list = 2 (production batches): like 0 behind kSegBlocks leading workgroups
// -- dispatched first -- whose waves walk the SEGMENTS of the batch's long reads (tables built by batch_check_kernel:
// chk[1] = long reads, chk[9] = segments; without any those workgroups return at once); the read waves skip long reads.
constexpr u32 kSegBlocks = 1024;  // x 4 waves: a C4 batch has ~28 000 segments
// (struct LongReads: skx_kernels.hpp)
#define SKX_SKETCH_PARAMS                                                                                              \
    const uint8_t *__restrict__ bases, const u64 *__restrict__ offsets, u32 n_reads, u32 k_rt, u64 seed, u32 s, u64 max_ref,  \
        u64 *__restrict__ out_sk, u32 sk_stride, u32 *__restrict__ out_len, u32 *__restrict__ out_cnt_in, u32 from_list,      \
        u32 *__restrict__ retry, u32 *__restrict__ big, const u64 *__restrict__ filt, u32 filt_shift, u64 n_bases,            \
        u32 *__restrict__ chk, LongReads lr, KmerFilter kf, u32 pool_cap, u32 pool_fixed
#define SKX_SKETCH_ARGS \
    bases, offsets, n_reads, k_rt, seed, s, max_ref, out_sk, sk_stride, out_len, out_cnt_in, from_list, retry, big, filt, filt_shift, n_bases, chk, lr, kf, pool_cap, pool_fixed
template <int KT, int HCAP, bool INRANGE>
__device__ __forceinline__ void sketch_wave_body(SKX_SKETCH_PARAMS, unsigned char* smem, SketchTables<KT>* tb) {
    const bool packed = (from_list & 0x100u) != 0u;  // (bit 8: 4-bit packed input)
    from_list &= 0xFFu;
    // (whole workgroups that have nothing to do leave before the table is filled: the segment workgroups of a batch without
    // long reads, the list walk over an empty list)
    const bool seg_block = from_list == 2u && blockIdx.x < kSegBlocks;
    if (seg_block && chk[9] == 0u) return;
    if (from_list == 1u && retry[0] == 0u) return;
    fill_sketch_tables<KT>(tb);
    __syncthreads();
    const u32 wpb = blockDim.x >> 6;  // waves per block: 4, or 1 for the list walk (see launch_sketch)
    u32 w = blockIdx.x * wpb + (threadIdx.x >> 6);
    if constexpr (INRANGE) {
        if (seg_block) {
            if (chk[6] & 2u) return;  // the tables are incomplete (the host refuses the batch)
            const u32 n_seg = min(chk[9], lr.segs_cap);
            for (u32 sg = w; sg < n_seg; sg += kSegBlocks * 4u) {
                const u32 pos = lr.seg_tab[sg];
                if (pos >= lr.long_cap || lr.list[pos] >= n_reads || sg < lr.seg0[pos]) continue;  // (never: belt and braces)
                sketch_one_read<KT, HCAP, true, true>(smem, lr.list[pos], bases, offsets, k_rt, seed, s, max_ref, out_sk, sk_stride,
                                                      out_len, out_cnt_in, nullptr, nullptr, filt, filt_shift, n_bases, chk, tb,
                                                      packed, true, sg - lr.seg0[pos], lr.seg_h + (size_t)sg * kSegSlots,
                                                      lr.seg_cnt + sg, kf, pool_cap, pool_fixed);
                wave_sync();  // the wave's LDS region is reused by its next segment
            }
            return;
        }
    }
    if (from_list == 2u) w -= kSegBlocks * 4u;
    if (from_list != 1u) {
        if (w < n_reads)
            sketch_one_read<KT, HCAP, INRANGE>(smem, w, bases, offsets, k_rt, seed, s, max_ref, out_sk, sk_stride, out_len,
                                               out_cnt_in, retry, big, filt, filt_shift, n_bases, chk, tb, packed,
                                               from_list == 2u, 0u, nullptr, nullptr, kf, pool_cap, pool_fixed);
        return;
    }
    const u32 n = retry[0];
    for (u32 i = w; i < n; i += gridDim.x * wpb) {
        sketch_one_read<KT, HCAP, INRANGE>(smem, retry[1u + i], bases, offsets, k_rt, seed, s, max_ref, out_sk, sk_stride,
                                           out_len, out_cnt_in, nullptr, big, filt, filt_shift, n_bases, chk, tb, packed, false, 0u,
                                           nullptr, nullptr, kf, pool_cap, pool_fixed);
        wave_sync();  // the wave's LDS region is reused by its next read
    }
}
template <int KT, int HCAP, bool INRANGE>
__global__ __launch_bounds__(256, HCAP < kSketchCap ? 8 : 2) void sketch_wave_kernel(SKX_SKETCH_PARAMS) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ SketchTables<KT> tb;
    sketch_wave_body<KT, HCAP, INRANGE>(SKX_SKETCH_ARGS, smem, &tb);
}
// One wave per long read of a production batch (walking the list batch_check_kernel built): the segment waves' hashes are
// gathered into the wave's LDS buffer (kSketchCap entries) and finished like any other read's.  A segment slot that
// overflowed, or more hashes than the buffer holds, sends the read to the block sketcher (`big`), which reads it whole.
__global__ __launch_bounds__(64) void sketch_merge_kernel(const u64* __restrict__ offsets, u32 s, u64 max_ref,
                                                          u64* __restrict__ out_sk, u32 sk_stride, u32* __restrict__ out_len,
                                                          u32* __restrict__ out_cnt_in, u32* __restrict__ big,
                                                          const u64* __restrict__ filt, u32 filt_shift,
                                                          u32* __restrict__ chk, LongReads lr, u32 pool_cap, u32 pool_fixed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* hashes = reinterpret_cast<u64*>(smem);
    const u32 lane = lane_id();
    if (chk[6] & 2u) return;  // the tables are incomplete (the host refuses the batch)
    const u32 n_long = min(chk[1], lr.long_cap);
    for (u32 pos = blockIdx.x; pos < n_long; pos += gridDim.x) {
        const u32 r = lr.list[pos], seg0 = lr.seg0[pos];
        const u32 n_seg = (u32)((offsets[r + 1] - offsets[r] + kSketchCap - 1u) / kSketchCap);
        u32 m = 0;
        bool bad = seg0 + n_seg > lr.segs_cap;
        for (u32 i0 = 0; i0 < n_seg && !bad; i0 += 64u) {
            const bool on = i0 + lane < n_seg;
            const u32 c = on ? lr.seg_cnt[seg0 + i0 + lane] : 0u;  // (64 segments' counts at once)
            if (__ballot(c == kSegOvf)) { bad = true; break; }
            const u32 incl = wave_incl_scan(c);
            const u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);
            if (m + tot > (u32)kSketchCap) { bad = true; break; }
            const u64* src = lr.seg_h + (size_t)(seg0 + i0 + lane) * kSegSlots;
            u64* dst = hashes + m + incl - c;
            for (u32 j = 0; __ballot(j < c); ++j)
                if (j < c) dst[j] = src[j];
            m += tot;
        }
        wave_sync();
        if (bad) {
            if (lane == 0) { out_len[r] = kSketchRetry; out_cnt_in[r] = 0; list_append(big, r); }
        } else {
            sketch_finish<kSketchCap, true>(hashes, m, r, s, max_ref, out_sk, sk_stride, out_len, out_cnt_in, filt, filt_shift, chk, pool_cap, pool_fixed);
        }
        wave_sync();  // the buffer is reused by the wave's next read
    }
}
// Experiment (SKX_SKETCH_ROOM=1): the fast in-range variant held to 5 waves per SIMD by its register allocation instead of
// by unused dynamic LDS.  It gives the scan its stand-alone speed inside the pipeline and costs the step 12 %: the
// 102-register waves leave no room in the register file for the other kernels' waves on the SIMDs they occupy.
template <int KT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 5))) void sketch_wave_kernel_capped(SKX_SKETCH_PARAMS) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ SketchTables<KT> tb;
    sketch_wave_body<KT, kSketchSmallHashes, true>(SKX_SKETCH_ARGS, smem, &tb);
}

// =====================================================================================
// the block sketcher: reads whose hashes do not fit one wave's LDS (and full sketches of long sequences)
// =====================================================================================
// One 1024-thread block per read of the `big` list, 16 384 hash slots in LDS (128 KB), nothing in global memory.
// The read is streamed in chunks like above; accepted hashes are appended to the buffer, and whenever it cannot take
// another 1024 the block sorts it, drops duplicates and keeps the `keep` smallest -- from then on only hashes <= the
// largest kept one are accepted (finch's "h <= heap.max", mash.rs push) -- so the buffer ends up holding the
// min(keep, #distinct) smallest distinct hashes above `floor`.  A sketch larger than kBigKeep takes further passes over
// the read, each starting above the previous pass's largest hash.  Exact for any read length and any s.
constexpr u32 kBigHashes = 16384;
constexpr u32 kBigKeep = 12288;
constexpr u32 kBigChunk = 4096;  // raw bytes normalised per chunk
constexpr size_t kBigLds = (size_t)kBigHashes * 8 + kBigChunk + 64;

struct BlockScratch {
    u32 wsum[16];
    u32 n;      // hashes in the buffer
    u32 nb;     // codes in the chunk buffer
    u32 out;    // running count of a block-wide compaction
    u32 cnt;    // a second running count (in-range / member hashes)
};

// block-wide: position of this thread's flag among all set flags (in thread order) + total; 1024 threads
__device__ __forceinline__ u32 block_rank(bool flag, BlockScratch& sh, u32& total) {
    const u32 lane = lane_id(), wv = threadIdx.x >> 6;
    const u64 mask = __ballot(flag);
    if (lane == 0) sh.wsum[wv] = (u32)__popcll(mask);
    __syncthreads();
    u32 before = 0, tot = 0;
#pragma unroll
    for (u32 w = 0; w < 16u; ++w) { const u32 c = sh.wsum[w]; before += w < wv ? c : 0u; tot += c; }
    __syncthreads();  // wsum is reused by the next call
    total = tot;
    return before + (u32)__popcll(mask & lanemask_lt());
}

template <int KT, bool INRANGE>
__global__ __launch_bounds__(1024) void sketch_block_kernel(const uint8_t* __restrict__ bases, const u64* __restrict__ offsets,
                                                            const u32* __restrict__ big, u32 n_big, u32 k_rt, u64 seed, u32 s,
                                                            u64 max_ref, u64* __restrict__ out_sk, u32 sk_stride,
                                                            u32* __restrict__ out_len, u32* __restrict__ out_cnt_in,
                                                            const u64* __restrict__ filt, u32 filt_shift, u32 packed,
                                                            u32* __restrict__ chk, u32 pool_cap, u32 pool_fixed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ BlockScratch sh;
    __shared__ u32 s_off, s_ok;
    u64* hashes = reinterpret_cast<u64*>(smem);
    uint8_t* codes = smem + (size_t)kBigHashes * 8;
    const u32 tid = threadIdx.x, lane = lane_id();
    const u32 k = KT > 0 ? (u32)KT : k_rt;

    for (u32 bi = blockIdx.x; bi < n_big; bi += gridDim.x) {
        const u32 r = big[1u + bi];
        const u64 o0 = offsets[r];
        const u32 lraw = (u32)(offsets[r + 1] - o0);
        const uint8_t* rd = bases + o0;
        u64* out = out_sk + (size_t)r * sk_stride;
        const bool pool = INRANGE && filt != nullptr && sk_stride == 0u;
        bool pool_ok = true;
        if (pool) {
            // pool mode (sketch_finish): the row is reserved before its size is known -- at most min(s, bases) entries
            if (tid == 0) {
                const u32 want = min(s, max(lraw, 1u));
                const u32 part = blockIdx.x % kPoolParts, part_cap = (pool_cap - pool_fixed) / kPoolParts;
                const u32 off = atomicAdd(&chk[16u + 16u * part], want);
                s_off = pool_fixed + part * part_cap + off;
                s_ok = (off + want <= part_cap && off + want >= off) ? 1u : 0u;
                if (!s_ok) atomicOr(&chk[6], 4u);
            }
            __syncthreads();
            out = out_sk + (size_t)s_off;
            pool_ok = s_ok != 0u;
            if (!pool_ok) {  // (the batch is repeated with a larger pool)
                if (tid == 0) { out_len[r] = 0; out_cnt_in[r] = 0; }
                __syncthreads();
                continue;
            }
        }
        u64 floor_v = 0;
        bool have_floor = false;
        u32 outn = 0;         // sketch entries emitted so far (distinct hashes, ascending)
        u32 wrote = 0, cin = 0;

        for (;;) {  // one pass over the read per kBigKeep sketch entries
            const u32 keep = min(s - outn, kBigKeep);
            u64 thr = INRANGE ? max_ref : ~0ull;
            bool truncated = false;
            if (tid == 0) { sh.n = 0; sh.nb = 0; }
            __syncthreads();

            // sort + distinct + keep the `keep` smallest; leaves sh.n = entries kept (all threads return with it in sync)
            auto compact = [&]() {
                const u32 n = sh.n;
                u32 p2 = 2;
                while (p2 < n) p2 <<= 1;
                for (u32 i = n + tid; i < p2; i += 1024u) hashes[i] = kPad;
                __syncthreads();
                for (u32 size = 2; size <= p2; size <<= 1) {
                    for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
                        for (u32 t = tid; t < (p2 >> 1); t += 1024u) {
                            const u32 i = 2u * t - (t & (stride - 1u));
                            const u32 j = i + stride;
                            const bool up = (i & size) == 0u;
                            const u64 a = hashes[i], b = hashes[j];
                            if ((a > b) == up) { hashes[i] = b; hashes[j] = a; }
                        }
                        __syncthreads();
                    }
                }
                // distinct, in place: an element only ever moves towards the front, one round of 1024 at a time
                u32 nd = 0;
                for (u32 base = 0; base < n; base += 1024u) {
                    const u32 i = base + tid;
                    const bool v = i < n;
                    const u64 h = v ? hashes[i] : 0;
                    const bool head = v && (i == 0 || hashes[i - 1] != h);
                    u32 tot;
                    const u32 pos = nd + block_rank(head, sh, tot);  // (its barriers order the reads above before the writes)
                    if (head) hashes[pos] = h;
                    nd += tot;
                    __syncthreads();
                }
                if (nd > keep) { nd = keep; thr = hashes[keep - 1u]; truncated = true; }
                __syncthreads();
                if (tid == 0) sh.n = nd;
                __syncthreads();
            };

            for (u32 cbase = 0;; cbase += kBigChunk) {
                const u32 cend = min(lraw, cbase + kBigChunk);
                // 1. normalise the chunk behind the carried codes (sh.nb holds the carry)
                for (u32 base = cbase; base < cend; base += 1024u) {
                    const u32 idx = base + tid;
                    const u32 code = idx >= cend ? 5u : packed ? packed_code(bases, o0 + idx) : classify_base((u32)rd[idx]);
                    const bool kp = code != 5u;
                    u32 tot;
                    const u32 pos = sh.nb + block_rank(kp, sh, tot);
                    if (kp) codes[pos] = (uint8_t)code;
                    __syncthreads();
                    if (tid == 0) sh.nb += tot;
                    __syncthreads();
                }
                const u32 nb = sh.nb;
                const u32 nk = nb >= k ? nb - k + 1u : 0u;
                // 2. hash the windows ending in this chunk, 1024 at a time
                for (u32 p0 = 0; p0 < nk; p0 += 1024u) {
                    if (sh.n + 1024u > kBigHashes) compact();  // (uniform: sh.n only changes between barriers)
                    const u32 p = p0 + tid;
                    bool valid = p < nk;
                    u64 fwd = 0, rc = 0;
                    u32 bad = 0;
                    if (valid) {
#pragma unroll
                        for (u32 j = 0; j < (KT > 0 ? (u32)KT : 32u); ++j) {
                            if (j < k) {
                                u32 c = codes[p + j];
                                bad |= c >> 2;
                                c &= 3u;
                                fwd = (fwd << 2) | c;
                                rc |= (u64)(3u - c) << (2 * j);
                            }
                        }
                    }
                    const u64 h = hash_canonical_packed<KT>(fwd < rc ? fwd : rc, k, seed);
                    valid = valid && bad == 0 && h <= thr && (!have_floor || h > floor_v);
                    u32 tot;
                    const u32 pos = sh.n + block_rank(valid, sh, tot);
                    if (valid) hashes[pos] = h;
                    __syncthreads();
                    if (tid == 0) sh.n += tot;
                    __syncthreads();
                }
                if (cend >= lraw) break;
                // 3. the last k-1 codes open the next chunk
                const u32 kc = min(nb, k - 1u);
                const u32 cv = tid < kc ? (u32)codes[nb - kc + tid] : 0u;
                __syncthreads();
                if (tid < kc) codes[tid] = (uint8_t)cv;
                if (tid == 0) sh.nb = kc;
                __syncthreads();
            }
            compact();
            const u32 n = sh.n;  // the next `n` sketch entries, ascending and distinct, in hashes[0 .. n)

            // emit (production: only the hashes some genome holds are written, AFTER the truncation to s)
            for (u32 base = 0; base < n; base += 1024u) {
                const u32 i = base + tid;
                const bool v = i < n;
                const u64 h = v ? hashes[i] : 0;
                if (INRANGE && filt != nullptr) {
                    bool kp = false;
                    if (v) kp = filter_hit(filt, filt_shift, h);
                    u32 tot;
                    const u32 pos = wrote + block_rank(kp, sh, tot);
                    if (kp) out[pos] = h;
                    wrote += tot;
                } else {
                    if (v) out[outn + i] = h;
                    u32 tot;
                    (void)block_rank(v && h <= max_ref, sh, tot);
                    cin += tot;
                }
            }
            outn += n;
            if (!truncated || outn >= s) break;   // exhausted, or the sketch is complete
            floor_v = hashes[n - 1u];
            have_floor = true;
            __syncthreads();
        }
        if (tid == 0) {
            out_len[r] = pool ? s_off : min(outn, s);
            out_cnt_in[r] = (INRANGE && filt != nullptr) ? wrote : cin;
        }
        __syncthreads();
    }
    (void)lane;
}

// =====================================================================================
// device-side look at a batch's offsets, so the host reads back 24 bytes instead of every offset
// =====================================================================================
// chk[0] = 0xFFFFFFFF - (first r with offsets[r+1] < offsets[r])   (0: offsets are monotonic)
// chk[1] = number of LONG reads (is_split_long: more than kLongSplit bases, inside the batch), chk[9] = their segments
// chk[2..3] = offsets[0], chk[4..5] = offsets[n_reads]               (chk zeroed by the caller / the previous publish)
// lr.list != NULL (production batches): every long read is listed (lr.list / lr.seg0, in arrival order of the atomics) and
// gets a run of segment slots, each pointing back at it (lr.seg_tab) -- the work list of the segment waves and of
// sketch_merge_kernel.  A batch that would need more slots than exist (only possible with offsets that are not
// monotonic: n_bases <= the stream's max_batch_bases bounds the sum of the lengths) is flagged in chk[6].
__global__ void batch_check_kernel(const u64* __restrict__ offsets, u32 n_reads, u64 n_bases, u32* __restrict__ chk,
                                   u32* __restrict__ cnt_tail, LongReads lr) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    const u64 lo = offsets[0];
    for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
        const u64 o0 = offsets[r], o1 = offsets[r + 1];
        if (o1 < o0) atomicMax(&chk[0], 0xFFFFFFFFu - r);
        else if (lr.list && is_split_long(o0, o1, lo, n_bases)) {
            const u32 n_seg = (u32)((o1 - o0 + kSketchCap - 1u) / kSketchCap);
            const u32 pos = atomicAdd(&chk[1], 1u);
            const u32 seg0 = atomicAdd(&chk[9], n_seg);
            if (pos >= lr.long_cap || seg0 + n_seg > lr.segs_cap || seg0 + n_seg < seg0) { atomicOr(&chk[6], 2u); continue; }
            lr.list[pos] = r;
            lr.seg0[pos] = seg0;
            for (u32 i = 0; i < n_seg; ++i) lr.seg_tab[seg0 + i] = pos;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const u64 a = offsets[0], b = offsets[n_reads];
        chk[2] = (u32)a; chk[3] = (u32)(a >> 32); chk[4] = (u32)b; chk[5] = (u32)(b >> 32);
        for (u32 i = 0; i < kPoolParts; ++i) chk[16u + 16u * i] = 0;  // the batch's rows start at the beginning of every sub-pool
        *cnt_tail = 0;  // entry n_reads of the per-read pair counts: the exclusive scan runs over n_reads + 1 entries
    }
}
// Hands the few words the host needs per push -- chk[0..7] and the total pair count -- to PAGE-LOCKED HOST memory and
// raises a sequence number there; the host spins on it.  A blit copy + stream synchronisation for the same 36 bytes
// cost ~100 us of idle front stream per push (kernel timeline), this costs a launch.  Also re-arms the device-side
// counters (chk, the retry list) for the next push.
// n (<= 64) words from device memory into page-locked host memory, by the device: what a small asynchronous device-to-host
// copy is for -- except that hipMemcpyAsync made the HOST wait for everything queued in front of it (round 4, kernel trace: the
// thread that queues a group's eight rankings stood at the 8-byte copy of every ranking's live counters until that ranking's
// prefix kernels were done, and the sketch of the next batch was queued 2.7 ms late, once per group)
// (d_src is zeroed as it is read: the counters it serves are published once per use and re-armed here, without a memset)
__global__ void store_host_words_kernel(volatile u32* __restrict__ h_dst, u32* __restrict__ d_src, u32 n) {
    if (blockIdx.x == 0 && threadIdx.x < n) { h_dst[threadIdx.x] = d_src[threadIdx.x]; d_src[threadIdx.x] = 0u; }
    __threadfence_system();
}
__global__ void publish_kernel(u32* __restrict__ chk, u32* __restrict__ retry, u32* __restrict__ big,
                               const u32* __restrict__ total_pairs, volatile u32* __restrict__ h_pub, u32 seq,
                               const u32* __restrict__ dict_ctr) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (big) { chk[7] = big[0]; big[0] = 0; }  // (reads that needed the block sketcher: a statistic for the host)
    for (int i = 0; i < 8; ++i) h_pub[i] = chk[i];
    h_pub[9] = chk[9];  // (segments of the batch's long reads: a statistic)
    for (int i = 0; i < 16; ++i) chk[i] = 0;  // (the sub-pools' bump counters behind these 16 words live on: a block-sketcher round
                                              // of the same batch reserves behind the rows already there; batch_check_kernel
                                              // restarts them with the next batch)
    if (retry) retry[0] = 0;
    h_pub[8] = *total_pairs;
    h_pub[10] = dict_ctr ? dict_ctr[2] + (dict_ctr[1] & 1u) : 0xFFFFFFFFu;  // distinct keys of the speculative gather = |Q| of the pass
    {
        // row entries the batch asked the pool for: kPoolParts x the fullest sub-pool's request (what a pool must hold for the
        // same batch to fit), also when the pool was too small
        u32 worst = 0;
        for (u32 i = 0; i < kPoolParts; ++i) worst = max(worst, chk[16u + 16u * i]);
        h_pub[kChkPool] = (u32)min((u64)worst * kPoolParts, (u64)0xFFFFFFFFu);
    }
    __threadfence_system();
    h_pub[15] = seq;
    __threadfence_system();
}

// =====================================================================================
// exclusive scan of the per-read pair counts (n_reads + 1 entries) -> pair offsets
// =====================================================================================
// Two small launches: (a) every block scans its 1024 entries and leaves its total, (b) every block adds the totals of
// the blocks before it.  (Replaces a library scan: two launches on the push's critical path instead of three.)
// (256-thread blocks, four entries per thread: a 1024-thread block needs 16 free wave slots on ONE CU and waits tens of
// microseconds for them next to the other streams' kernels -- measured; 4-wave blocks slip in)
__device__ __forceinline__ u32 block256_excl_scan4(const u32 (&c)[4], u32 (&excl)[4], u32* wtot /* [4] shared */) {
    const u32 lane = lane_id(), wv = threadIdx.x >> 6;
    const u32 mine = c[0] + c[1] + c[2] + c[3];
    const u32 incl = wave_incl_scan(mine);
    if (lane == 63u) wtot[wv] = incl;
    __syncthreads();
    u32 before = 0, total = 0;
#pragma unroll
    for (u32 w = 0; w < 4u; ++w) { before += w < wv ? wtot[w] : 0u; total += wtot[w]; }
    u32 run = before + incl - mine;
#pragma unroll
    for (int j = 0; j < 4; ++j) { excl[j] = run; run += c[j]; }
    return total;
}
__global__ __launch_bounds__(256) void count_scan_a_kernel(const u32* __restrict__ in, u32* __restrict__ out, u32 n,
                                                           u32* __restrict__ bsum) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    __shared__ u32 wtot[4];
    const u32 i0 = blockIdx.x * 1024u + threadIdx.x * 4u;
    u32 c[4], e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = i0 + j < n ? in[i0 + j] : 0u;
    const u32 total = block256_excl_scan4(c, e, wtot);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i0 + j < n) out[i0 + j] = e[j];
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void count_scan_b_kernel(u32* __restrict__ out, u32 n, const u32* __restrict__ bsum) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    __shared__ u32 part[4];
    __shared__ u32 s_before;
    const u32 lane = lane_id(), wv = threadIdx.x >> 6;
    u32 t = 0;
    for (u32 b = threadIdx.x; b < blockIdx.x; b += 256u) t += bsum[b];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) t += (u32)__shfl_xor((int)t, d, 64);
    if (lane == 0) part[wv] = t;
    __syncthreads();
    if (threadIdx.x == 0) s_before = part[0] + part[1] + part[2] + part[3];
    __syncthreads();
    if (blockIdx.x) {
        const u32 i0 = blockIdx.x * 1024u + threadIdx.x * 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < n) out[i0 + j] += s_before;
    }
}

// =====================================================================================
// membership filter: which read hashes occur in ANY reference genome
// =====================================================================================
// A read hash that no genome holds adds 0 to every shared-hash count, so dropping it before the dictionary changes
// no result -- and for real reads (sequencing errors, novel k-mers) that is most of them: at C2 / B=49152 the
// in-range read hashes number ~517 k per pass of which ~10 k exist in the collection; everything downstream (Q, the
// bit matrices, the pair lists) shrinks by that factor.  The filter is a direct-mapped bitmap over hash >> shift
// (hashes are uniform, so no second hash function is needed): no false negatives, and a false positive only costs
// an all-zero row of M.  Built once per reference from the resident matrix.  (filter_mask / filter_hit: skx_common.hpp)
// count != NULL: linear counting instead -- set bit (v >> shift) of a plain bitmap and count the bits newly set: the number of
// distinct values, as long as the bitmap is sparse (skx_ref_create sizes the real filter from it).
__global__ void filter_build_kernel(const u64* __restrict__ vals, u64 n, u32 shift, u64* __restrict__ words,
                                    bool markers_are_values, unsigned long long* __restrict__ count) {
    u32 fresh = 0;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 v = vals[i];
        if (!markers_are_values && v >= kEmpty) continue;  // padding / empty cells of the tiled matrix
        if (count) {
            const u64 idx = v >> shift, bit = 1ull << (idx & 63u);
            if (!(words[idx >> 6] & bit)) fresh += (atomicOr(&words[idx >> 6], bit) & bit) ? 0u : 1u;  // (most values are repeats: test first)
        } else {
            const u64 m = filter_mask(v);
            if ((words[v >> shift] & m) != m) atomicOr(&words[v >> shift], m);
        }
    }
    if (count) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) fresh += (u32)__shfl_xor((int)fresh, d, 64);
        if (lane_id() == 0 && fresh) atomicAdd(count, (unsigned long long)fresh);
    }
}

// One wave per read: keeps, in order, those of the first cnt[r] hashes of the read's sketch row that pass the
// filter (in-place compaction: a hash only ever moves towards the front) and stores the new count.
__global__ __launch_bounds__(256) void filter_apply_kernel(u64* __restrict__ sk, u32 sk_stride, u32* __restrict__ cnt,
                                                           u32 n_reads, const u64* __restrict__ bits, u32 shift) {
    const u32 r = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = lane_id();
    if (r >= n_reads) return;
    u64* row = sk + (size_t)r * sk_stride;
    const u32 n = cnt[r];
    u32 kept = 0;
    for (u32 i0 = 0; i0 < n; i0 += 64u) {
        const u32 i = i0 + lane;
        u64 h = 0;
        bool keep = false;
        if (i < n) {
            h = row[i];
            keep = filter_hit(bits, shift, h);
        }
        const u64 b = __ballot(keep);
        if (keep) row[kept + (u32)__popcll(b & ((1ull << lane) - 1ull))] = h;
        kept += (u32)__popcll(b);
    }
    if (lane == 0) cnt[r] = kept;
}

// =====================================================================================
// dictionary of the batch's query hashes
// =====================================================================================
// ---- purpose-built dictionary ----------------------------------------------------------------------------------
// The surviving pairs of a pass repeat a few distinct hashes many times (C2 / B=98304: ~450 k pairs, ~11 k distinct),
// and a general radix sort of all of them costs 16 launches on the front stream's critical path.  Instead:
//   dict_insert   (fused with the pair gather) every pair hash goes into an open-addressing hash SET in HBM
//   dict_count    used slots: bucket = top bits of the hash (uniform), count per bucket, place inside the bucket
//   dict_scan_a/b exclusive scan of the 2^17 bucket counts (two levels) -> bucket bases, |Q|
//   dict_scatter  used slots -> Q by bucket (and the set is emptied)
//   dict_bucket_sort  one thread per bucket: insertion sort of its handful of keys  => Q sorted and distinct
constexpr u32 kDictBuckets = 1u << 17;
__device__ __forceinline__ u32 dict_bucket(u64 key, u32 bshift) { return (u32)min((u64)(kDictBuckets - 1u), key >> bshift); }

// pair_cap: the pair arrays' capacity.  The kernel can be queued before the host knows how many pairs the reads have
// (right behind the sketcher, on its stream); when they do not fit one pass it does nothing and the host, which learns
// the count a moment later, cuts the batch into passes and inserts per pass.
// sk_stride == 0: pool mode -- read r's row starts at sk + row_off[r] (sketch_finish)
// One THREAD per read (a read has a handful of pairs: one wave per read made this 98 304 nearly empty waves per C2 batch, each
// waiting for a wave slot next to the other streams' kernels -- 100-300 us on the critical chain behind the sketcher).
// ctr[2] counts the keys new to the set: |Q| of the pass, which the batch summary hands to the host.
// (Tried on top: listing the new keys and sorting the list by ONE workgroup in LDS instead of the five kernels that walk the
// hash set -- a bitonic network took 0.68 ms next to the other streams, a five-barrier bucket sort 0.25 ms: a lone workgroup
// gets a sliver of one busy CU; no gain over the chain it replaced, and quadratic on skewed hashes.  Dropped.)
__global__ __launch_bounds__(256) void dict_insert_kernel(const u64* __restrict__ sk, u32 sk_stride, const u32* __restrict__ poff,
                                                          u32 r_begin, u32 r_end, u32 p_base, u64* __restrict__ pair_h,
                                                          u32* __restrict__ pair_r, u64* __restrict__ ht, u32 ht_mask,
                                                          u32* __restrict__ ctr, u32 pair_cap, const u32* __restrict__ row_off,
                                                          PairBase base) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    const u32 r = r_begin + blockIdx.x * 256u + threadIdx.x;
    // base: the batch shares its pass with the batches before it -- its pairs go behind theirs (whose pair counts are only
    // known on the device when this is queued), into the same hash set
    u64 p_off64 = 0;
#pragma unroll
    for (int i = 0; i < kPairBaseMax; ++i)
        if (base.p[i]) p_off64 += *base.p[i];
    if ((u64)(poff[r_end] - p_base) + p_off64 > pair_cap) return;  // (uniform)
    const u32 p_off = (u32)p_off64;
    u32 a = 0, b = 0;
    if (r < r_end) { a = poff[r]; b = poff[r + 1]; }
    const u64* row = sk_stride ? sk + (size_t)r * sk_stride : (r < r_end ? sk + (size_t)row_off[r] : sk);
    pair_h += p_off; pair_r += p_off;
    u32 fresh = 0;  // keys this thread put into the set
    for (u32 j = 0; j < b - a; ++j) {
        const u64 key = row[j];
        pair_h[a - p_base + j] = key;
        pair_r[a - p_base + j] = r - r_begin;
        if (key == kPad) { atomicOr(&ctr[1], 1u); continue; }  // the empty marker itself: appended to Q at the end
        u32 slot = (u32)(key ^ (key >> 29)) & ht_mask;
        for (;;) {
            // (a look before the compare-and-swap: a slot only ever goes from empty to its key inside a pass, so a key seen is final --
            // most pairs repeat a key that is already there, and a load is served where the atomic has to go to the memory side)
            u64 prev = __hip_atomic_load(&ht[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == kPad) prev = atomicCAS(&ht[slot], kPad, key);
            if (prev == kPad) { ++fresh; break; }  // a new key
            if (prev == key) break;
            slot = (slot + 1u) & ht_mask;
        }
    }
    // ctr[2] = distinct keys in the set (|Q| of the pass): one add per wave, not per key (truth-strain batches bring 160 k new keys)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) fresh += __shfl_xor(fresh, d);
    if ((threadIdx.x & 63u) == 0 && fresh) atomicAdd(&ctr[2], fresh);
}
// used slots: count per bucket; the slot remembers its place inside the bucket (atomics spread over 2^17 addresses)
__global__ __launch_bounds__(256) void dict_count_kernel(const u64* __restrict__ ht, u32 ht_slots, u32 bshift,
                                                         u32* __restrict__ slot_off, u32* __restrict__ bcount) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    for (u32 slot = blockIdx.x * 256u + threadIdx.x; slot < ht_slots; slot += gridDim.x * 256u) {
        const u64 key = ht[slot];
        if (key != kPad) slot_off[slot] = atomicAdd(&bcount[dict_bucket(key, bshift)], 1u);
    }
}
// exclusive scan of the bucket counts, two levels: (a) inside every block of 1024 buckets, (b) over the block totals
__global__ __launch_bounds__(256) void dict_scan_a_kernel(u32* __restrict__ bcount, u32* __restrict__ bbase,
                                                          u32* __restrict__ btot) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    __shared__ u32 wtot[4];
    const u32 b0 = blockIdx.x * 1024u + threadIdx.x * 4u;
    u32 c[4], e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = bcount[b0 + j]; bcount[b0 + j] = 0; }  // zero again for the next pass
    const u32 total = block256_excl_scan4(c, e, wtot);
#pragma unroll
    for (int j = 0; j < 4; ++j) bbase[b0 + j] = e[j];
    if (threadIdx.x == 0) btot[blockIdx.x] = total;
}
__global__ __launch_bounds__(128) void dict_scan_b_kernel(u32* __restrict__ btot, const u32* __restrict__ ctr,
                                                          u32* __restrict__ n_q) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    static_assert(kDictBuckets / 1024u == 128u, "one thread per block of buckets");
    __shared__ u32 part[128];
    const u32 t = threadIdx.x;
    const u32 c = btot[t];
    part[t] = c;
    __syncthreads();
    for (u32 d = 1; d < 128u; d <<= 1) {
        const u32 v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    btot[t] = part[t] - c;  // exclusive: first position of the block's buckets in Q
    if (t == 127u) { btot[128] = part[t]; *n_q = part[t] + (ctr[1] & 1u); }
}
// used slots -> Q by bucket; the set is emptied for the next pass
__global__ __launch_bounds__(256) void dict_scatter_kernel(u64* __restrict__ ht, u32 ht_slots, u32 bshift,
                                                           const u32* __restrict__ slot_off, const u32* __restrict__ bbase,
                                                           const u32* __restrict__ btot, const u32* __restrict__ ctr,
                                                           u64* __restrict__ q) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    if (blockIdx.x == 0 && threadIdx.x == 0 && (ctr[1] & 1u)) q[btot[128]] = kPad;  // the largest possible hash goes last
    for (u32 slot = blockIdx.x * 256u + threadIdx.x; slot < ht_slots; slot += gridDim.x * 256u) {
        const u64 key = ht[slot];
        if (key != kPad) {
            const u32 b = dict_bucket(key, bshift);
            q[btot[b >> 10] + bbase[b] + slot_off[slot]] = key;
            ht[slot] = kPad;
        }
    }
}
__global__ void dict_bucket_sort_kernel(u64* __restrict__ q, const u32* __restrict__ bbase, const u32* __restrict__ btot,
                                        u32* __restrict__ ctr) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= kDictBuckets) return;
    const u32 a = btot[b >> 10] + bbase[b];
    const u32 z = (b + 1u < kDictBuckets) ? btot[(b + 1u) >> 10] + bbase[b + 1u] : btot[128];
    for (u32 i = a + 1; i < z; ++i) {  // buckets hold a handful of keys (uniform hashes, 2^17 buckets)
        const u64 v = q[i];
        u32 j = i;
        while (j > a && q[j - 1] > v) { q[j] = q[j - 1]; --j; }
        q[j] = v;
    }
    if (b == 0) { ctr[1] = 0; ctr[2] = 0; }  // (the "saw the all-ones hash" flag and the key count; their readers ran in earlier kernels)
}

__device__ __forceinline__ u32 lower_bound_u64(const u64* __restrict__ a, u32 n, u64 v) {
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ u32 upper_bound_u64(const u64* __restrict__ a, u32 n, u64 v) {
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// qrow != NULL: the dictionary was split (classify kernels): position in Q -> row of the bit matrix
// bbase / btot: the dictionary's bucket bases (dict_scan_a/b: still those of this pass's Q) -- a hash's bucket starts at
// btot[bucket >> 10] + bbase[bucket] and holds a handful of keys in ascending order: two or three reads instead of the 19 dependent
// ones of a binary search over 470 k keys (truth-strain pass: 0.19 ms of the front half, alone on the chip)
__global__ void pair_q_kernel(const u64* __restrict__ pair_h, u32 n_pairs, const u64* __restrict__ q,
                              const u32* __restrict__ n_q, u32* __restrict__ pair_q, const u32* __restrict__ qrow,
                              const u32* __restrict__ bbase, const u32* __restrict__ btot, u32 bshift) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_pairs) {
        const u64 h = pair_h[p];
        const u32 nq = *n_q, b = dict_bucket(h, bshift);
        u32 pos = btot[b >> 10] + bbase[b];
        while (pos < nq && q[pos] < h) ++pos;  // (every pair's hash is in Q)
        pair_q[p] = qrow ? qrow[pos] : pos;
    }
}

// win[2*bt] = qa, win[2*bt+1] = qb : Q[qa..qb) are the query hashes inside [lo[bt], hi[bt]]
__global__ void window_kernel(const u64* __restrict__ lo, const u64* __restrict__ hi, u32 n_bt,
                              const u64* __restrict__ q, const u32* __restrict__ n_q, u32* __restrict__ win,
                              volatile u32* __restrict__ h_nq) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    const u32 bt = blockIdx.x * blockDim.x + threadIdx.x;
    if (bt == 0 && h_nq) { *h_nq = *n_q; __threadfence_system(); }  // |Q| for the host (page-locked memory), a hint only
    if (bt >= n_bt) return;
    const u32 nq = *n_q;
    // (a band without any real hash -- only possible at the end of a tile's columns -- gets the empty window [nq, nq): the
    // first index of the windows of a tile then never decreases from band to band, which word_bands_kernel relies on)
    u32 qa = nq, qb = nq;
    if (lo[bt] <= hi[bt]) { qa = lower_bound_u64(q, nq, lo[bt]); qb = upper_bound_u64(q, nq, hi[bt]); }
    win[2 * bt] = qa;
    win[2 * bt + 1] = qb;
}

// reference hashes >= kEmpty were lifted out of the matrix at upload (they would alias the table's
// markers); set their bits here.  exc_g / exc_h: (genome, hash) pairs.
__global__ void exceptions_kernel(const u32* __restrict__ exc_g, const u64* __restrict__ exc_h, u32 n_exc,
                                  const u64* __restrict__ q, const u32* __restrict__ n_q, u64* __restrict__ m_bits,
                                  u32 n_pad, u32* __restrict__ m_dirty, const u32* __restrict__ qrow, u32 row0, u32 n_fixed) {
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_exc) return;
    const u32 nq = n_q ? *n_q : n_fixed;  // (static dictionary: q = its tail of lifted hashes, n_fixed of them, rows from row0 on)
    u32 pos = lower_bound_u64(q, nq, exc_h[e]);
    if (pos < nq && q[pos] == exc_h[e]) {
        pos += row0;
        if (qrow) pos = qrow[pos];
        atomicOr(&m_bits[(size_t)(pos >> 6) * n_pad + exc_g[e]], 1ull << (pos & 63u));
        if (m_dirty) *m_dirty = 1u;  // (the transpose must look at M)
    }
}

// =====================================================================================
// rare-hash index of the reference (round 5)
// =====================================================================================
// The scan prices a pass by the size of its dictionary: a (band, tile) block holds its slice of Q in LDS, and the lean kernel's
// one-pass probe takes 254 entries.  That fits a collection whose genomes share most of their hashes (C2's random-hash clone tree:
// ~10 k distinct member hashes per batch).  SURVEY.md 8(d)'s SNP clone tree at k = 16 does not look like that: 40 000 strains x
// 1 400 private SNPs x 16 windows put ~4.4 M distinct hashes below the largest reference hash -- 45 % of ALL canonical 16-mers in that
// range -- so every sequencing error has an even chance to hit SOMEBODY's private hash: ~160 k distinct member hashes per batch of
// 98 304 reads, 470 k for eight batches, 9 000 entries per slice, 8 ms per scan instead of 0.55.  Almost all of those hashes
// are RARE: one strain holds them, or one lineage.  Nothing needs to stream 3.2 GB of matrix to find out WHICH genome holds a hash
// that a handful hold: skx_ref_create builds, once, a hash table over the distinct reference hashes with the number of genomes that
// hold each (kt_key / kt_cnt) and, for those held by at most R genomes (policy "rare_hash_genomes", default 1024), the list of
// those genomes (postings).  A pass then splits its dictionary: DENSE hashes (held by more than R genomes) go to the scan as
// before -- rows [0, nd) of the bit matrix --, RARE ones get the rows behind them and their bits straight from the postings
// (sparse_fill_kernel: one atomicOr into M per posting).  Same bits, hence the same rows and table; the scan's dictionary is
// back to the ~10 k hashes the collection shares.
constexpr u32 kRareDense = 0xFFFFFFFFu;   // kt_off of a key held by more than R genomes: no postings, the scan finds it
constexpr u32 kSlotNone = 0xFFFFFFFFu;    // sslot of a query hash that no genome holds (a false positive of the membership filter)
__device__ __forceinline__ u32 kt_start(u64 key, u32 mask) { return (u32)((key ^ (key >> 29)) * 0x9E3779B1u) & mask; }

// every real hash of the tiled matrix: insert its key, count the genomes that hold it (columns hold distinct hashes)
// spmask (several species resident, or NULL): bit sp of spmask[slot] = some genome of species sp holds the key (grp_sp: species of a
// rank group of 512 genomes; per_tile = s x 256 elements of the tiled matrix per tile of 256 genomes)
__global__ __launch_bounds__(256) void rare_count_kernel(const u64* __restrict__ mat, u64 n_elems, u64* __restrict__ key,
                                                         u32* __restrict__ cnt, u32 mask, u32* __restrict__ overflow,
                                                         unsigned long long* __restrict__ spmask, const u32* __restrict__ grp_sp, u64 per_tile) {
    for (u64 e = (u64)blockIdx.x * 256u + threadIdx.x; e < n_elems; e += (u64)gridDim.x * 256u) {
        const u64 h = mat[e];
        if (h >= kEmpty) continue;  // padding; hashes >= kEmpty were lifted into the exception list and stay with the scan path
        u32 slot = kt_start(h, mask), tries = 0;
        for (;;) {
            u64 prev = key[slot];
            if (prev == kPad) prev = atomicCAS(&key[slot], kPad, h);
            if (prev == kPad || prev == h) {
                atomicAdd(&cnt[slot], 1u);
                if (spmask) {
                    const unsigned long long bit = 1ull << grp_sp[(u32)(e / per_tile) / (kRankWords * 64u / kTileGenomes)];
                    if (!(spmask[slot] & bit)) atomicOr(&spmask[slot], bit);
                }
                break;
            }
            slot = (slot + 1u) & mask;
            if (++tries > mask) { *overflow = 1u; break; }
        }
    }
}
// the genomes of every rare key (off[slot] != kRareDense): post[off + position], position from the key's cursor
__global__ __launch_bounds__(256) void rare_fill_kernel(const u64* __restrict__ mat, u64 n_elems, u32 s,
                                                        const u64* __restrict__ key, const u32* __restrict__ off,
                                                        u32* __restrict__ cursor, u32* __restrict__ post, u32 mask) {
    const u64 per_tile = (u64)s * kTileGenomes;
    for (u64 e = (u64)blockIdx.x * 256u + threadIdx.x; e < n_elems; e += (u64)gridDim.x * 256u) {
        const u64 h = mat[e];
        if (h >= kEmpty) continue;
        u32 slot = kt_start(h, mask);
        while (key[slot] != h) slot = (slot + 1u) & mask;  // (every real hash was inserted by rare_count_kernel)
        const u32 o = off[slot];
        if (o == kRareDense) continue;
        const u32 g = (u32)(e / per_tile) * kTileGenomes + (u32)(e % kTileGenomes);  // padded genome index of mat[t][i][c]
        post[o + atomicAdd(&cursor[slot], 1u)] = g;
    }
}

// the keys the scan has to find (held by more genomes than the index lists): out[0 .. *n) in no particular order; *n may exceed cap
__global__ __launch_bounds__(256) void collect_dense_kernel(const u64* __restrict__ key, const u32* __restrict__ off, u64 slots,
                                                            u64* __restrict__ out, u32* __restrict__ out_slot, u32* __restrict__ n, u32 cap) {
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < slots; i += (u64)gridDim.x * 256u) {
        const u64 k = key[i];
        if (k == kPad || off[i] != kRareDense) continue;
        const u32 at = atomicAdd(n, 1u);
        if (at < cap) { out[at] = k; out_slot[at] = (u32)i; }
    }
}
// the static dictionary's slice per (band, tile): several species resident -> the dictionary is the species' sorted segments one after
// the other (seg[2 sp], seg[2 sp + 1] = first row and number of hashes of species sp's), and a tile looks in ITS species' segment only
__global__ void window_seg_kernel(const u64* __restrict__ lo, const u64* __restrict__ hi, u32 n_bt, u32 n_tiles,
                                  const u64* __restrict__ q, const u32* __restrict__ seg, const u32* __restrict__ grp_sp, u32* __restrict__ win) {
    const u32 bt = blockIdx.x * blockDim.x + threadIdx.x;
    if (bt >= n_bt) return;
    const u32 sp = grp_sp[(bt % n_tiles) / (kRankWords * 64u / kTileGenomes)];
    const u32 a = seg[2u * sp], n = seg[2u * sp + 1u];
    u32 qa = a + n, qb = a + n;
    if (lo[bt] <= hi[bt]) { qa = a + lower_bound_u64(q + a, n, lo[bt]); qb = a + upper_bound_u64(q + a, n, hi[bt]); }
    win[2 * bt] = qa;
    win[2 * bt + 1] = qb;
}
// ---- long lists as bit rows.  A hash held by more than kShortList genomes (a lineage's: ~200) costs a pass one atomic per genome
// on its list and batch -- 8.7 M per C2 pass of the SNP workload, 3.9 ms at the ~2.3 G/s scattered device-scope atomics reach.  The
// lists themselves differ from hash to hash (a strain's own SNP removes a lineage hash from that one strain), so they cannot be
// shared; but as BIT ROWS over the genomes (mlong[row][genome word]: 5 KB per hash at C2, 1.3 GB for its 263 k long lists) a pass
// adds them up with bit-sliced counters -- coalesced 512-byte reads, no atomics (gain_long_kernel) -- and finds the candidates on a
// row by ANDing its words with the candidates' (cand_long_kernel).
constexpr u32 kCtrStride = 16;           // words between the per-batch counters of a pass (nlrow, nqc): one 64-byte line each
constexpr u32 kShortList = 8;            // genomes a list may have to be walked by its row's lane alone
constexpr u32 kLongFlag = 0x80000000u;   // sslot[2 i + 1]: the hash has a bit row, sslot[2 i] = its index
__global__ __launch_bounds__(256) void mlong_build_kernel(const u32* __restrict__ lslot, u32 n_long, const u32* __restrict__ off,
                                                          const u32* __restrict__ cnt, const u32* __restrict__ post, u64* __restrict__ mlong,
                                                          u32 n_gw) {
    const u32 w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = lane_id();
    if (w >= n_long) return;
    const u32 slot = lslot[w], o = off[slot], n = cnt[slot];
    for (u32 j = lane; j < n; j += 64u) {
        const u32 g = post[o + j];
        atomicOr(&mlong[(size_t)w * n_gw + (g >> 6)], 1ull << (g & 63u));
    }
}

// the bit rows transposed (RareIndex::mlongT): one wave per (64 bit rows, genome word) -- lane = bit row on the way in, = genome on the
// way out (transpose64 is defined further down)
__device__ __forceinline__ u64 transpose64(u64 x, u32 lane);
__global__ __launch_bounds__(256) void mlong_transpose_kernel(const u64* __restrict__ mlong, u32 n_long, u32 n_gw, u64* __restrict__ mlongT,
                                                              u32 n_lw) {
    const u32 lw = blockIdx.x, gw = blockIdx.y * 4u + (threadIdx.x >> 6), lane = lane_id();
    if (gw >= n_gw) return;
    const u32 row = lw * 64u + lane;
    const u64 x = row < n_long ? mlong[(size_t)row * n_gw + gw] : 0ull;
    const u64 y = transpose64(x, lane);
    mlongT[((size_t)gw * 64u + lane) * n_lw + lw] = y;
}
// ---- long lists as (pattern, exceptions) (round 6).  The long lists of a clonal collection are near-duplicates: every lineage-level hash of a
// lineage is held by the lineage's ~200 strains minus the odd strain whose own SNP removed it (SURVEY 8(d)'s tree at C2: 263 k long lists,
// 258 k of them distinct -- and 97 % within 14 genomes of the most frequent list of their lineage, median 2: tools/diag_patterns.py).
// skx_ref_create groups the long lists by a MinHash signature of the list (list_sig_kernel; two lists of Jaccard similarity J share it
// with probability J), takes the most frequent exact list of a group as the group's PATTERN (host: a few hundred thousand 12-byte
// records), and writes every list of the group as that pattern plus its symmetric difference (pat_exceptions_kernel: XOR of two bit rows)
// when that is at most kPatExcMax genomes: prec[lid] = {pattern, n, n x (genome | in the pattern but NOT on the list << 31)}.  The patterns'
// bits transposed form a small matrix PM[word of 64 patterns][genome] laid out like M.  For such a row a pass needs neither the bit row
// nor the list:  gain_b[g] = sum_p hist_b[p] * PM[p][g]  (hist_b[p] = occurrences of the pattern's rows among batch b's pairs: the same
// kernel that adds up the dense rows of M, on ~200 rows instead of 130 k bit rows of 5 KB)  +/- the count at the row's exceptions; and
// a compact ranking maps the row to its pattern's row of the compact matrix unless one of the exceptions is a candidate.  Whatever the
// grouping finds, rows and table are exact: a list is stored as EXACTLY pattern xor exceptions, or left to its bit row.
constexpr u32 kPatFlag = 0x40000000u;    // sslot[2 i + 1], with kLongFlag: the row's list is a pattern + exceptions (prec[bit row])
constexpr u32 kLenMask = 0x3FFFFFFFu;    // ... the list's length
constexpr u32 kPatNone = 0xFFFFFFFFu;
constexpr u32 kPatRec = 16;              // u32 words of a record: pattern, exceptions, the exceptions
constexpr u32 kPatExcMax = kPatRec - 2u;
__device__ __forceinline__ u32 mix32(u32 x) {
    x = (x ^ (x >> 16)) * 0x7FEB352Du;
    x = (x ^ (x >> 15)) * 0x846CA68Bu;
    return x ^ (x >> 16);
}
// sig[row] = min over the list of mix32(genome), content[row] = an order-independent 64-bit digest of the list: one wave per bit row
__global__ __launch_bounds__(256) void list_sig_kernel(const u32* __restrict__ lslot, u32 n_long, const u32* __restrict__ off,
                                                       const u32* __restrict__ cnt, const u32* __restrict__ post, u32* __restrict__ sig,
                                                       u64* __restrict__ content) {
    const u32 w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = lane_id();
    if (w >= n_long) return;
    const u32 slot = lslot[w], o = off[slot], n = cnt[slot];
    u32 m = 0xFFFFFFFFu;
    u64 c = 0;
    for (u32 j = lane; j < n; j += 64u) {
        const u32 g = post[o + j], x = mix32(g);
        m = min(m, x);
        c += ((u64)x * 0x9E3779B97F4A7C15ull) ^ ((u64)g << 32);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { m = min(m, (u32)__shfl_xor((int)m, d, 64)); c += shfl_xor64(c, d); }
    if (lane == 0u) { sig[w] = m; content[w] = c; }
}
// prec[row] = the row's list as its pattern (pat_of[row], or none) + exceptions: one wave per bit row XORs it with the pattern's
__global__ __launch_bounds__(256) void pat_exceptions_kernel(const u64* __restrict__ mlong, u32 n_gw, u32 n_long, const u32* __restrict__ pat_of,
                                                             const u32* __restrict__ pat_rep, u32* __restrict__ prec, u32* __restrict__ n_done) {
    const u32 w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = lane_id();
    if (w >= n_long) return;
    u32* rec = prec + (size_t)w * kPatRec;
    const u32 pat = pat_of[w];
    if (pat == kPatNone) { if (lane == 0u) rec[0] = kPatNone; return; }
    const u64* a = mlong + (size_t)w * n_gw;
    const u64* b = mlong + (size_t)pat_rep[pat] * n_gw;
    u32 total = 0;
    for (u32 i = lane; i < n_gw; i += 64u) total += (u32)__popcll(a[i] ^ b[i]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) total += (u32)__shfl_xor((int)total, d, 64);
    if (total > kPatExcMax) { if (lane == 0u) rec[0] = kPatNone; return; }
    u32 at = 0;
    for (u32 i0 = 0; i0 < n_gw; i0 += 64u) {
        const u32 i = i0 + lane;
        const u64 bw = i < n_gw ? b[i] : 0ull;
        u64 x = i < n_gw ? (a[i] ^ bw) : 0ull;
        u64 bal = __ballot(x != 0ull);
        while (bal) {  // (a handful of words of the row differ at all)
            const u32 src = (u32)__builtin_ctzll(bal);
            bal &= bal - 1ull;
            u64 xs = readlane64(x, (int)src);
            const u64 bs = readlane64(bw, (int)src);
            while (xs) {
                const u32 bit = (u32)__builtin_ctzll(xs);
                xs &= xs - 1ull;
                if (lane == 0u) rec[2u + at] = ((i0 + src) * 64u + bit) | (((bs >> bit) & 1ull) ? 0x80000000u : 0u);
                ++at;
            }
        }
    }
    if (lane == 0u) { rec[0] = pat; rec[1] = total; atomicAdd(n_done, 1u); }
}
// PM[w][g] bit j = pattern 64 w + j holds genome g (the patterns' bit rows transposed, in M's layout): one wave per (64 patterns, genome word)
__global__ __launch_bounds__(256) void pat_matrix_kernel(const u64* __restrict__ mlong, const u32* __restrict__ pat_rep, u32 n_pat, u32 n_gw,
                                                         u64* __restrict__ pm, u32 n_pad) {
    const u32 pw = blockIdx.x, gw = blockIdx.y * 4u + (threadIdx.x >> 6), lane = lane_id();
    if (gw >= n_gw) return;
    const u32 p = pw * 64u + lane;
    const u64 x = p < n_pat ? mlong[(size_t)pat_rep[p] * n_gw + gw] : 0ull;
    pm[(size_t)pw * n_pad + gw * 64u + lane] = transpose64(x, lane);
}
// ---- a pass's dictionary, split.  classify_a: look every query hash up (qinfo[q] = its slot | kSlotNone; dense: bit 31 of qloc[q]),
// block-local exclusive count of the dense ones; classify_b: one block scans the block totals, publishes nd / ns;
// classify_c: Qd (the dense hashes, still ascending), qrow[q] = the hash's row of the bit matrix (dense rows first, in Qd order,
// then -- from the next multiple of 64 on -- the others in Q order), sslot[row - nd64] = key-table slot of a rare hash.
__global__ __launch_bounds__(256) void classify_a_kernel(const u64* __restrict__ q, const u32* __restrict__ n_q, RareIndex ri,
                                                         u32* __restrict__ qinfo, u32* __restrict__ qloc, u32* __restrict__ bsum) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ u32 wtot[4];
    const u32 nq = *n_q;
    const u32 q0 = blockIdx.x * 1024u + threadIdx.x * 4u;
    if (blockIdx.x * 1024u >= nq) return;
    u32 c[4], e[4], info[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = 0; info[j] = kSlotNone;
        if (q0 + j < nq) {
            const u64 h = q[q0 + j];
            if (h >= kEmpty) c[j] = 1u;  // (lifted hashes: exceptions_kernel sets their bits; they ride with the dense rows)
            else {
                u32 slot = kt_start(h, ri.mask);
                for (;;) {
                    const u64 kk = ri.key[slot];
                    if (kk == h) { info[j] = slot; c[j] = ri.off[slot] == kRareDense ? 1u : 0u; break; }
                    if (kk == kPad) break;
                    slot = (slot + 1u) & ri.mask;
                }
            }
        }
    }
    const u32 total = block256_excl_scan4(c, e, wtot);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (q0 + j < nq) { qinfo[q0 + j] = info[j]; qloc[q0 + j] = e[j] | (c[j] << 31); }
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
constexpr u32 kNoStatic = 0xFFFFFFFFu;
__global__ __launch_bounds__(1024) void classify_b_kernel(u32* __restrict__ bsum, const u32* __restrict__ n_q, u32* __restrict__ n_d,
                                                          volatile u32* __restrict__ h_words, u32 static_nd) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ u32 part[1024];
    const u32 nq = *n_q, nb = (nq + 1023u) / 1024u, t = threadIdx.x;
    const u32 per = (nb + 1023u) / 1024u;  // block totals per thread (nq <= 2^22: at most 4)
    u32 mine = 0;
    for (u32 i = 0; i < per; ++i) { const u32 b = t * per + i; if (b < nb) mine += bsum[b]; }
    part[t] = mine;
    __syncthreads();
    for (u32 d = 1; d < 1024u; d <<= 1) {
        const u32 v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u32 run = part[t] - mine;
    for (u32 i = 0; i < per; ++i) { const u32 b = t * per + i; if (b < nb) { const u32 v = bsum[b]; bsum[b] = run; run += v; } }
    if (t == 1023u) {
        // (static_nd: the reference's STATIC dense dictionary -- the dense rows are its rows, whichever of them this pass asks for)
        const u32 nd_pass = part[t], nd = static_nd != kNoStatic ? static_nd : nd_pass, nd64 = (nd + 63u) & ~63u;
        n_d[0] = nd; n_d[1] = nq - nd_pass; n_d[2] = nd64; n_d[3] = nd64 + (nq - nd_pass);  // dense rows, other rows, first other row, rows in all
        if (h_words) { h_words[0] = nq; h_words[1] = nd_pass; __threadfence_system(); }
    }
}
__global__ __launch_bounds__(256) void classify_c_kernel(const u64* __restrict__ q, const u32* __restrict__ n_q,
                                                         const u32* __restrict__ qinfo, const u32* __restrict__ qloc,
                                                         const u32* __restrict__ bsum, const u32* __restrict__ n_d,
                                                         u64* __restrict__ qd, u32* __restrict__ qrow, u32* __restrict__ sslot, RareIndex ri) {
    __builtin_amdgcn_s_setprio(3);
    const u32 nq = *n_q, i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nq) return;
    const u32 loc = qloc[i], dr = bsum[i >> 10] + (loc & 0x7FFFFFFFu), nd64 = n_d[2];
    if (loc >> 31) {
        if (ri.qs) {  // its row of the static dictionary (every dense hash is in it): by key slot, the lifted hashes by search in their tail
            const u32 slot = qinfo[i];
            qrow[i] = slot != kSlotNone ? ri.srow[slot] : ri.tail0 + lower_bound_u64(ri.qs + ri.tail0, ri.n_sd - ri.tail0, q[i]);
        }
        else { qd[dr] = q[i]; qrow[i] = dr; }
    }
    else {  // (the other rows start on a word boundary; sslot[2 sr], [2 sr + 1] = start and length of the hash's genome list -- or, length with
        // kLongFlag: the index of its bit row)
        const u32 sr = i - dr, slot = qinfo[i];
        qrow[i] = nd64 + sr;
        const u32 np = slot != kSlotNone ? ri.cnt[slot] : 0u;
        const bool lng = ri.mlong != nullptr && np > kShortList;  // (a long list: the row carries the index of its bit row instead)
        const u32 lid = lng ? ri.lid[slot] : 0u;
        const bool pat = lng && ri.prec != nullptr && ri.prec[(size_t)lid * kPatRec] != kPatNone;  // (... which is a pattern + exceptions)
        sslot[2u * sr] = slot == kSlotNone ? 0u : lng ? lid : ri.off[slot];
        sslot[2u * sr + 1u] = lng ? (np | kLongFlag | (pat ? kPatFlag : 0u)) : np;
    }
}
// a reference without the index: every hash is the scan's, rows = positions in Q
__global__ void nd_from_nq_kernel(const u32* __restrict__ n_q, u32* __restrict__ n_d, volatile u32* __restrict__ h_words) {
    const u32 nq = *n_q;
    n_d[0] = nq; n_d[1] = 0; n_d[2] = (nq + 63u) & ~63u; n_d[3] = nq;
    if (h_words) { h_words[0] = nq; h_words[1] = nq; __threadfence_system(); }
}
// bits of the rare rows: M[row][g] for every genome g on the hash's list.  A wave takes 64 rows; rows with long lists are walked by
// the whole wave, 64 postings at a time.
__global__ __launch_bounds__(256) void sparse_fill_kernel(const u32* __restrict__ sslot, const u32* __restrict__ n_d, RareIndex ri,
                                                          u64* __restrict__ m_bits, u32 n_pad, u32* __restrict__ m_dirty,
                                                          const u32* __restrict__ only_if) {
    __builtin_amdgcn_s_setprio(2);
    if (only_if && !*only_if) return;  // (no batch of the pass ranks on the full matrix: nobody reads these rows)
    const u32 nd = n_d[2], ns = n_d[1], lane = lane_id();  // (nd: the first row behind the dense ones)
    const u32 wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    bool wrote = false;
    for (u32 r0 = wave * 64u; r0 < ns; r0 += n_waves * 64u) {
        const u32 sr = r0 + lane;
        u32 off = 0, cnt = 0;
        if (sr < ns) {
            const uint2 e = reinterpret_cast<const uint2*>(sslot)[sr];
            off = e.x; cnt = e.y;
            if (cnt & kLongFlag) { cnt &= kLenMask; off = ri.off[ri.lslot[off]]; }  // (a bit row: M wants the list)
        }
        const u32 row = nd + sr;
        u64* const mrow = m_bits + (size_t)(row >> 6) * n_pad;
        const u64 bit = 1ull << (row & 63u);
        if (cnt && cnt <= 8u) {
            for (u32 j = 0; j < cnt; ++j) atomicOr(&mrow[ri.post[off + j]], bit);
            wrote = true;
        }
        u64 longs = __ballot(cnt > 8u);
        while (longs) {
            const u32 src = (u32)__builtin_ctzll(longs);
            longs &= longs - 1ull;
            const u32 o = __shfl(off, (int)src), c = __shfl(cnt, (int)src), rw = nd + r0 + src;
            u64* const mr = m_bits + (size_t)(rw >> 6) * n_pad;
            const u64 b = 1ull << (rw & 63u);
            for (u32 j = lane; j < c; j += 64u) atomicOr(&mr[ri.post[o + j]], b);
            wrote = true;
        }
    }
    if (wrote && m_dirty) *m_dirty = 1u;
}

// The rare rows of a pass straight into the group-major matrix Mq -- not through M and the transpose.  A rare row's bits are known
// without the scan: a bit row of the reference's index (long lists: Mq[grp][row] IS words 8 grp .. 8 grp + 7 of it) or at most
// kShortList genomes.  Through M (sparse_fill_kernel + transpose_bits_kernel) a truth-strain pass of 469 k rows paid one scattered
// atomic per posting (2.0 ms) and a transpose of 2.4 GB of mostly zeros (1.7 ms) whenever one of its batches ranked on everything.
// Here: one block per 64 rows (= one word of rowany per group); thread (row, word of the group) walks the groups, copying or
// zero-filling -- 512-byte runs per wave, 2 KB per block and group --, then writes the short rows' few words over its own zeros.
// rowany[grp][word of these rows] is written for every group (plain stores), grp_any[grp] += rows of the group that hold a bit.
//
// Zero-filling is most of that: a truth-strain pass's rows are almost all short lists (somebody's private SNP hash met by a sequencing
// error: eight genomes at most), 381 k rows x 293 groups x 64 B at C4 = 7 GB of zeros per batch.  `ext` = {row stride, clean rows} of
// this buffer set's (Mq, rowany) pair says where that is not needed: under the SAME row stride as the passes before, a row below
// `clean` holds bits for a group only if the set's rowany -- still the previous pass's here -- says so (the transposes and this kernel
// keep it so: a row's words are written whenever its old or its new flag is set).  Such a block of 64 rows writes only the words that
// are non-zero now or were flagged before; the other blocks (a new stride, rows never written under it) are written in full as before.
// stride 0xFFFFFFFF = arrays zeroed at allocation: clean under any stride.  mq_extent_kernel, behind this kernel, brings `ext` up to date.
constexpr u32 kRareGrpChunk = 512;  // groups per turn (rowany words of a block in LDS)
constexpr u32 kAnyStride = 0xFFFFFFFFu;
__global__ __launch_bounds__(256) void rare_to_mq_kernel(const u32* __restrict__ sslot, const u32* __restrict__ n_d, RareIndex ri,
                                                         u64* __restrict__ mq, u32 nq_rows, u32 n_gw, u64* __restrict__ rowany,
                                                         u32* __restrict__ grp_any, const u32* __restrict__ only_if, u32 chunk,
                                                         const u32* __restrict__ ext) {
    __builtin_amdgcn_s_setprio(2);
    if (only_if && !*only_if) return;  // (no batch of the pass ranks on the full matrix)
    __shared__ u32 lpost[64][kShortList];
    __shared__ unsigned long long lany[kRareGrpChunk];
    __shared__ unsigned long long lold[kRareGrpChunk];  // the rows' flags before this pass (all ones: write every word)
    const u32 clean = (ext && (ext[0] == kAnyStride || ext[0] == nq_rows)) ? ext[1] : 0u;
    const u32 nd64 = n_d[2], ns = n_d[1], n_grp = n_gw / kRankWords, n_words = nq_rows >> 6;
    const u32 r = threadIdx.x >> 3, cw = threadIdx.x & 7u, lane = lane_id();
    static_assert(kRankWords == 8 && kShortList == 8, "thread = (row, word of the group) = (row, posting)");
    u32 acc = 0;  // (threads 0 .. chunk - 1 of the first turn) rows of "their" group that hold a bit, over the block's row blocks
    for (u32 sr0 = blockIdx.x * 64u; sr0 < ns; sr0 += gridDim.x * 64u) {
        // this thread's two rows: sr0 + r and sr0 + r + 32
        u32 lid[2], np[2];
        bool lng[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32 sr = sr0 + r + 32u * h;
            lid[h] = 0; np[h] = 0; lng[h] = false;
            u32 mine = 0xFFFFFFFFu;
            if (sr < ns) {
                const uint2 e = reinterpret_cast<const uint2*>(sslot)[sr];
                lng[h] = (e.y & kLongFlag) != 0u;
                np[h] = e.y & kLenMask;
                lid[h] = e.x;
                if (!lng[h] && cw < np[h]) mine = ri.post[e.x + cw];
            }
            lpost[r + 32u * h][cw] = mine;
        }
        const u32 row0 = nd64 + sr0;
        const bool known = row0 + 64u <= clean;  // (block-uniform)
        for (u32 g0 = 0; g0 < n_grp; g0 += chunk) {  // (chunk = kRareGrpChunk; the experiments build can force the turns on small references)
            const u32 g1 = min(n_grp, g0 + chunk);
            for (u32 i = threadIdx.x; i < g1 - g0; i += 256u) {
                lany[i] = 0ull;
                lold[i] = known ? rowany[(size_t)(g0 + i) * n_words + (row0 >> 6)] : ~0ull;
            }
            __syncthreads();
            // long rows: copy; short rows: zeros.  Four groups in flight per thread (the loads of a bit row are 64-byte pieces)
            for (u32 g = g0; g < g1; g += 4u) {
                u64 v[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        v[u][h] = (lng[h] && g + u < g1) ? __builtin_nontemporal_load(&ri.mlong[(size_t)lid[h] * n_gw + (size_t)(g + u) * kRankWords + cw]) : 0ull;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (g + u >= g1) break;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (v[u][h] != 0ull || ((lold[g + u - g0] >> (r + 32u * h)) & 1ull))
                            mq[((size_t)(g + u) * nq_rows + row0 + r + 32u * h) * kRankWords + cw] = v[u][h];
                        // rows of this wave (8 of them: 8 lanes each) that hold a bit for the group
                        u64 m = __ballot(v[u][h] != 0ull);
                        if (m) {
                            m |= m >> 4; m |= m >> 2; m |= m >> 1; m &= 0x0101010101010101ull;
                            const u64 rows8 = (m * 0x0102040810204080ull) >> 56;  // bit i = row i of the wave
                            if (lane == 0u) atomicOr(&lany[g + u - g0], rows8 << ((threadIdx.x >> 6) * 8u + 32u * h));
                        }
                    }
                }
            }
            // short rows: the words their postings fall into are written once more, by the thread that wrote the zero (thread (row, word
            // of the group) owns that word of every group: same thread, same address, program order -- no fence between the two stores)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32 rr = r + 32u * h;
                if (lng[h] || np[h] == 0u) continue;
                for (u32 j = 0; j < kShortList; ++j) {
                    const u32 pj = lpost[rr][j];
                    if (pj == 0xFFFFFFFFu) break;
                    const u32 gw = pj >> 6, g = gw / kRankWords;
                    if (gw % kRankWords != cw || g < g0 || g >= g1) continue;
                    u64 val = 0;
#pragma unroll
                    for (u32 k = 0; k < kShortList; ++k) {
                        const u32 pk = lpost[rr][k];
                        if (pk != 0xFFFFFFFFu && (pk >> 6) == gw) val |= 1ull << (pk & 63u);
                    }
                    mq[((size_t)g * nq_rows + row0 + rr) * kRankWords + cw] = val;
                    atomicOr(&lany[g - g0], 1ull << rr);
                }
            }
            __syncthreads();
            for (u32 i = threadIdx.x; i < g1 - g0; i += 256u) {
                const u64 a = lany[i];
                rowany[(size_t)(g0 + i) * n_words + (row0 >> 6)] = a;
                if (g0 == 0u && n_grp <= chunk && i == threadIdx.x) acc += (u32)__popcll(a);
                else if (a) atomicAdd(&grp_any[g0 + i], (u32)__popcll(a));
            }
            __syncthreads();  // (lany and lpost are reused)
        }
    }
    if (acc) atomicAdd(&grp_any[threadIdx.x], acc);
}

// {row stride, clean rows} of a buffer set's (Mq, rowany) pair after a pass that wrote it through the transposes + rare_to_mq_kernel
__global__ void mq_extent_kernel(u32* __restrict__ ext, u32 nq_rows, const u32* __restrict__ n_d, const u32* __restrict__ only_if) {
    if (only_if && !*only_if) return;
    const u32 rows = n_d[2] + ((n_d[1] + 63u) & ~63u);
    if (ext[0] == kAnyStride || ext[0] == nq_rows) { ext[0] = nq_rows; ext[1] = max(ext[1], rows); }
    else { ext[0] = nq_rows; ext[1] = rows; }  // (a new stride: what lies behind these rows belongs to the old layout)
}

// =====================================================================================
// the reference scan
// =====================================================================================
// One block per (band, tile): 256 lanes = 256 genome columns, rows [b*rb, (b+1)*rb).
// LDS: the band's slice of Q itself (sorted, up to CAP entries, 8 B each, plus an end sentinel) and an
// interpolation directory: the hash range [slice[0], slice[n-1]] is cut into power-of-two wide buckets and
// dir[b] = index of the first entry whose bucket is >= b.  Query hashes are uniform, so a probe is one u16
// directory read plus ~1.5 entry reads, and the position found IS the index into Q (no value array, no
// hashing, no CAS build; twice the entries of an open-addressing table in the same LDS).
// A lane meets its column's hits in ascending q, so it ORs them into one 64-bit word in a register and
// flushes that word to M[word][genome] when the word index moves on.  The OR is atomic: the neighbouring
// band can own bits of the same word.
// Measured alternatives that were all slower on MI355X ("scan kernel experiments"): plain
// stores for the words only one band can touch, parking finished words in LDS and writing them out after
// the last row (per lane or as contiguous 2 KB rows), taller bands, runs of bands per block.
// ABLATE (profiling aid, results invalid unless 0; env SKX_SCAN_ABLATE):
//   1 = no write to M, 2 = no probe at all (pure streaming), 3 = probe but ignore hits
// SPLIT: only the first and the last word a lane touches in a band can be shared with a neighbouring band; with
// SPLIT those go (atomically) to m_bits and every word in between -- complete, owned by the lane -- is stored
// plainly into m_int; the transpose ORs the two arrays.  Atomics cost a memory-side transaction each (~90 G/s),
// plain stores into exclusively owned words a fraction of that; the two kinds never share a cache line.
template <int CAP, int ABLATE, bool SPLIT>
__global__ __launch_bounds__(256) void scan_kernel(const u64* __restrict__ mat, u32 s, u32 n_tiles, u32 rb,
                                                   const u64* __restrict__ q, const u32* __restrict__ win,
                                                   u64* __restrict__ m_bits, u64* __restrict__ m_int, u32 n_pad) {
    constexpr u32 kBuckets = CAP > 2048 ? 4096 : 2048;  // directory entries (power of two); CAP = 2040 -> 20 KB of LDS
    __shared__ u64 slice[CAP + 1];
    __shared__ unsigned short dir[kBuckets + 1];
    const u32 bt = blockIdx.x;
    const u32 t = bt % n_tiles, b = bt / n_tiles, c = threadIdx.x;
    const u32 qa = win[2 * bt], qb = win[2 * bt + 1];
    if (qa >= qb) return;
    const u32 i0 = b * rb, rows = min(s, i0 + rb) - i0;
    const u64* col = mat + ((size_t)t * s + i0) * kTileGenomes + c;
    const u32 g = t * kTileGenomes + c;

    // a slice larger than CAP is walked in sub-window passes over the same rows (correct for any slice
    // size; slower, because the band is streamed once per pass)
    for (u32 sub = qa; sub < qb; sub += CAP) {
        const u32 n = min((u32)CAP, qb - sub);
        const u64 lo = q[sub], hi = q[sub + n - 1];
        // bucket(h) = (h - lo) >> shift, with (hi - lo) >> shift < kBuckets
        const u32 span_bits = 64u - (u32)__clzll((hi - lo) | 1ull);
        const u32 shift = span_bits > (u32)__builtin_ctz(kBuckets) ? span_bits - (u32)__builtin_ctz(kBuckets) : 0u;
        for (u32 j = c; j < n; j += 256u) slice[j] = q[sub + j];
        if (c == 0) slice[n] = kPad;
        __syncthreads();
        for (u32 j = c; j <= n; j += 256u) {
            // entry j opens every bucket in (bucket(j-1), bucket(j)]; the sentinel closes the rest
            const u32 bj = j < n ? (u32)((slice[j] - lo) >> shift) : kBuckets;
            const u32 bp = j == 0 ? 0xFFFFFFFFu : (u32)((slice[j - 1] - lo) >> shift);
            for (u32 x = bp + 1u; x <= bj; ++x) dir[x] = (unsigned short)j;
        }
        __syncthreads();

        u32 cur_w = 0xFFFFFFFFu;  // absolute word index (q >> 6)
        u64 cur_bits = 0;
        bool first = true;        // the next flush is this lane's first of the (sub-)pass
        const bool multi = qb - qa > (u32)CAP;  // several sub-window passes: hits are not in q order across passes
        auto hit = [&](u32 qi) {
            const u32 w = qi >> 6;
            if (w != cur_w) {
                if (ABLATE != 1 && cur_bits) {
                    if (SPLIT && !first && !multi) m_int[(size_t)cur_w * n_pad + g] = cur_bits;
                    else atomicOr(&m_bits[(size_t)cur_w * n_pad + g], cur_bits);
                    first = false;
                }
                cur_w = w; cur_bits = 0;
            }
            cur_bits |= 1ull << (qi & 63u);
        };
        auto probe = [&](u64 hv) {
            if (ABLATE == 2) { cur_bits ^= hv; return; }
            if (hv < lo || hv > hi) return;  // also drops the padding value
            u32 j = dir[(u32)((hv - lo) >> shift)];
            u64 e = slice[j];
            while (e < hv) e = slice[++j];   // the sentinel (all ones) ends every walk
            if (ABLATE == 3) { cur_bits ^= e; return; }
            if (e == hv) hit(sub + j);
        };
        // software-pipelined: the next 8 rows are in flight while the current 8 are probed
        u32 i = 0;
        if (rows >= 8u) {
            u64 h[8];
#pragma unroll
            for (u32 u = 0; u < 8u; ++u) h[u] = col[(size_t)u * kTileGenomes];
            for (i = 8u; i + 8u <= rows; i += 8u) {
                u64 hn[8];
#pragma unroll
                for (u32 u = 0; u < 8u; ++u) hn[u] = col[(size_t)(i + u) * kTileGenomes];
#pragma unroll
                for (u32 u = 0; u < 8u; ++u) probe(h[u]);
#pragma unroll
                for (u32 u = 0; u < 8u; ++u) h[u] = hn[u];
            }
#pragma unroll
            for (u32 u = 0; u < 8u; ++u) probe(h[u]);
        }
        for (; i < rows; ++i) probe(col[(size_t)i * kTileGenomes]);

        if (ABLATE >= 1) {
            if (cur_bits == 0x123456789ull) m_bits[g] = cur_bits;  // keep the work alive
        } else if (cur_bits) {
            atomicOr(&m_bits[(size_t)cur_w * n_pad + g], cur_bits);
        }
        __syncthreads();  // slice and directory are rebuilt by the next sub-window
    }
}

// =====================================================================================
// the reference scan, lean probe (default for sparse dictionaries)
// =====================================================================================
// scan_kernel's probe costs ~55 issued instructions per 64-lane element (31 VALU, 20 SALU for the exec-mask juggling of
// its branches and walk loop, 4 LDS): measured, that -- not HBM and not latency -- is what bounds it: 352 M wave
// instructions per launch over 1024 SIMDs at ~4 cycles each = the kernel's duration.  And once the probe is lean the
// write-back shows: every lane flushing its 8-byte word by atomicOr at its own moment costs 0.10 of 0.64 ms (plain
// stores at the same moments still 0.06: partial lines).  Same geometry here (one block per (band, tile), the best
// memory behaviour of everything tried), with
//   * a probe of ~20 instructions without branches: slices of at most kLeanCap = 254 entries, one-byte directory of
//     2048 buckets; while it is built the block checks that no bucket holds more than two entries (measured at C2, slices of
//     ~200 entries in 4096 buckets: 13 % of the blocks fail and take the walk-loop probe over the same tables); then an
//     element's match, if any, is slice[dir[bucket]] or its successor: one ds_read_u8, one ds_read2_b64, two 64-bit
//     compares.  The bucket is (high word of (h - lo)) >> (shift - 32) clamped to the last bucket, whose entries are
//     the end sentinels (kEmpty: no matrix cell holds it -- real hashes >= kEmpty live in the exception list, padding
//     is kPad), so out-of-range elements and padding need no test of their own;
//   * single-owner output: the block ORs its hit bits into an LDS tile acc[word of the slice][genome] (one ds_or_b64
//     per hit, no word tracking) and, when the band is done, stores the <= kLeanWords words of its slice for all 256
//     genomes with plain, fully coalesced stores into ITS OWN slab of `hbuf` -- written exactly once, zero words
//     included, so nothing has to be cleared.  Bands of a tile overlap in q, so a word of M is the OR of the slabs of the
//     2-3 bands that reach it: transpose_bits_kernel does that OR (word_bands_kernel tells it which bands).
// A band whose slice has more than kLeanCap entries (dense dictionaries; the host normally sends those to scan_kernel's
// split / big-table variants) is streamed once per sub-window with the walk probe and atomicOr into M, and raises
// *m_dirty so that the transpose also reads M.
constexpr u32 kLeanCap = 254;
constexpr u32 kLeanBuckets = 2048;
constexpr u32 kLeanBucketsMax = 4096;  // directory of a slice with more than kLeanBigFrom entries
constexpr u32 kLeanBigFrom = 96;
constexpr u32 kLeanMultiCap = 1022;    // entries per sub-window of a slice beyond kLeanCap (tables in the result tile's LDS)
static_assert((kLeanMultiCap + 2) * 8 <= 5 * 256 * 8 && (2048 + 2) * 2 <= kLeanBucketsMax + 8, "multi-window tables alias the lean ones");
constexpr u32 kLeanWords = 5;  // query words a slice of <= kLeanCap entries can touch
// BIG instance (round 4; passes whose slices mostly exceed kLeanCap -- C4 with eight batches per pass: ~390 entries per slice):
// the same branch-free probe over slices of up to kLeanCapBig entries -- a two-byte directory of 4096 buckets and THREE
// consecutive entries per probe (more than three entries in one bucket: the walk probe, as before), nine result words.  31 KB
// of LDS per block instead of 16.  Slices beyond that still take the multi-window path.
constexpr u32 kLeanCapBig = 510;
constexpr u32 kLeanWordsBig = 9;
constexpr u32 kLeanBucketsBig = 4096;

// NT: bit 0 = non-temporal slab stores, bit 1 = non-temporal loads of the matrix (it is streamed once per pass: marking its
// lines evict-first keeps them from flushing what the kernels running beside the scan gather from -- Mq, the pair lists)
template <int ABLATE, int NT = 0, bool BIG = false>
__global__ __launch_bounds__(256, SKX_SCAN_OCC) void scan_lean_kernel(const u64* __restrict__ mat, u32 s, u32 n_tiles, u32 rb,
                                                        const u64* __restrict__ q, const u32* __restrict__ win,
                                                        u64* __restrict__ m_bits, u32 n_pad, u64* __restrict__ hbuf,
                                                        u32* __restrict__ m_dirty, u32 prio) {
    // next to the VALU-bound sketch / ranking kernels (three-stream pipeline) the scan's few instructions should not
    // queue behind theirs: its loads are what keeps HBM busy
    if (prio) __builtin_amdgcn_s_setprio(3);
    constexpr u32 CAP = BIG ? kLeanCapBig : kLeanCap, WORDS = BIG ? kLeanWordsBig : kLeanWords;
    static_assert(!BIG || (NT & 4), "the BIG instance writes into M only");
    static_assert((kLeanMultiCap + 3) * 8 <= WORDS * 256 * 8, "multi-window slice aliases the result tile");
    __shared__ u64 slice[CAP + 3];
    __shared__ unsigned char dir[BIG ? 2 * (kLeanBucketsBig + 8) : kLeanBucketsMax + 8];
    __shared__ u64 acc[WORDS][kTileGenomes];
    __shared__ u32 deep;  // some bucket holds more than two (BIG: three) entries (or the slice spans < 2^43): walk probe
    const u32 bt = blockIdx.x;
    const u32 t = bt % n_tiles, b = bt / n_tiles, c = threadIdx.x;
    const u32 qa = win[2 * bt], qb = win[2 * bt + 1];
    if (qa >= qb) return;
    const u32 i0 = b * rb, rows = min(s, i0 + rb) - i0;
    const u32 g = t * kTileGenomes + c;
    const bool multi = qb - qa > CAP;  // several sub-window passes: the atomic path into M
    if (multi && c == 0) *m_dirty = 1u;
    const u32 w0 = qa >> 6, n_w = ((qb - 1u) >> 6) - w0 + 1u;  // (single window: n_w <= kLeanWords)
    if (!multi) {
#pragma unroll
        for (u32 k = 0; k < WORDS; ++k) acc[k][c] = 0;  // (only this lane ever touches column c: no barrier needed)
    }

    // A slice of more than kLeanCap entries (C4: nearly half of the blocks -- the order statistics of 256 genomes spread a
    // band's hash range to three times one genome's) used to stream its band once per 254 entries: 1.5 x the matrix in HBM
    // reads at C4 (PMC).  Those blocks need no result tile (their hits go to M), so its 10 KB hold a slice of up to
    // kLeanMultiCap entries and the directory's 4 KB become 2048 two-byte entries: one pass for slices up to 1022.
    u64* const sl = multi ? reinterpret_cast<u64*>(&acc[0][0]) : slice;
    unsigned short* const dir16 = reinterpret_cast<unsigned short*>(dir);
    const u32 cap = multi ? kLeanMultiCap : CAP;
    for (u32 sub = qa; sub < qb; sub += cap) {
        const u32 n = min(cap, qb - sub);
        const u64 lo = q[sub], hi = q[sub + n - 1];
        // bucket(h) = (h - lo) >> shift, with (hi - lo) >> shift < n_bk; longer slices get the larger directory (the chance
        // of three entries in one bucket grows with n^3 / n_bk^2: 1 % at n = 64 / 2048 buckets, 13 % at n = 150, and a
        // block that fails the test pays the walk probe for its whole band)
        const u32 bk_bits = multi ? (u32)__builtin_ctz(kLeanBuckets)
                            : BIG ? (u32)__builtin_ctz(kLeanBucketsBig)
                                  : (n > kLeanBigFrom ? (u32)__builtin_ctz(kLeanBucketsMax) : (u32)__builtin_ctz(kLeanBuckets));
        const u32 n_bk = 1u << bk_bits;
        const u32 span_bits = 64u - (u32)__clzll((hi - lo) | 1ull);
        const u32 shift = span_bits > bk_bits ? span_bits - bk_bits : 0u;
        for (u32 i = c; i < n; i += kTileGenomes) sl[i] = q[sub + i];
        if (c == 0) { sl[n] = kEmpty; sl[n + 1] = kEmpty; sl[n + 2] = kEmpty; deep = shift < 32u ? 1u : 0u; }
        __syncthreads();
        for (u32 i = c; i <= n; i += kTileGenomes) {
            // entry i opens every bucket in (bucket(i-1), bucket(i)]; the sentinel closes the rest
            const u32 bj = i < n ? (u32)((sl[i] - lo) >> shift) : n_bk;
            const u32 bp = i == 0 ? 0xFFFFFFFFu : (u32)((sl[i - 1] - lo) >> shift);
            if (multi || BIG) { for (u32 x = bp + 1u; x <= bj; ++x) dir16[x] = (unsigned short)i; }
            else { for (u32 x = bp + 1u; x <= bj; ++x) dir[x] = (unsigned char)i; }
            constexpr u32 kProbe = BIG ? 3u : 2u;  // entries one probe looks at
            if (i >= kProbe && i < n && (u32)((sl[i - kProbe] - lo) >> shift) == bj) deep = 1u;  // (benign race: same value)
        }
        __syncthreads();
        const bool lean = deep == 0u && !multi;
        const u32 sh_hi = shift - 32u;  // (lean only: shift >= 32)

        // ---- multi-window path: word in a register, atomicOr when it moves on (as scan_kernel)
        u32 cur_w = 0xFFFFFFFFu;
        u64 cur_bits = 0;
        auto hit_atomic = [&](u32 qi) {
            const u32 w = qi >> 6;
            if (w != cur_w) {
                if (ABLATE != 1 && cur_bits) atomicOr(&m_bits[(size_t)cur_w * n_pad + g], cur_bits);
                cur_w = w; cur_bits = 0;
            }
            cur_bits |= 1ull << (qi & 63u);
        };
        // ---- single-window path: bit (qi - 64 * w0) of this lane's column of the LDS tile
        const u32 lo_lo = (u32)lo;
        // (kept in vector registers on purpose: as scalar operands they would be re-moved for every element -- one
        // scalar operand per VALU instruction on this ISA, and the borrow / the mask already is one)
        u32 lo_hi = (u32)(lo >> 32), one = 1, zero = 0, rel = sub - (w0 << 6);
        asm volatile("" : "+v"(lo_hi), "+v"(one), "+v"(zero), "+v"(rel));
        u64* my_acc = &acc[0][c];
        auto probe_lean = [&](u64 hv) {
            if (ABLATE == 2) { cur_bits ^= hv; return; }
            // high word of (hv - lo), from the halves
            const u32 dh = (u32)(hv >> 32) - lo_hi - ((u32)hv < lo_lo ? 1u : 0u);
            const u32 bk = min(dh >> sh_hi, n_bk);
            const u32 j = BIG ? (u32)dir16[bk] : (u32)dir[bk];
            if constexpr (BIG) {
                u32 j1 = j + 1u, j2 = j + 2u;
                asm volatile("" : "+v"(j1), "+v"(j2));  // (three ds_read_b64: see below)
                const u64 e0 = slice[j], e1 = slice[j1], e2 = slice[j2];
                if (ABLATE == 3) { cur_bits ^= e0 ^ e1 ^ e2; return; }
                const bool m1 = e1 == hv, m2 = e2 == hv;
                if ((e0 == hv) || m1 || m2) {
                    const u32 qr = rel + j + (m1 ? 1u : 0u) + (m2 ? 2u : 0u);
                    __hip_atomic_fetch_or(&my_acc[(size_t)(qr >> 6) * kTileGenomes], make_u64(one, zero) << (qr & 63u),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                return;
            }
#if SKX_SCAN_SPLIT_READS
            // two ds_read_b64 instead of the ds_read2_b64 the compiler makes of adjacent entries: a wave's 64 random 16-byte
            // reads go through the LDS as 2 x 4 groups of 16 lanes (128 B per clock), two 8-byte reads as 2 x 2 groups of 32
            // over twice the banks (256 B per clock) -- MI355X_MICROARCH.md, LDS -- and with most elements hitting (the sample's
            // own species) the LDS pipe, not HBM, was what a CU ran out of
            u32 j1 = j + 1u;
            asm volatile("" : "+v"(j1));
            const u64 e0 = slice[j], e1 = slice[j1];
#else
            const u64 e0 = slice[j], e1 = slice[j + 1u];
#endif
            if (ABLATE == 3) { cur_bits ^= e0 ^ e1; return; }
            const bool m1 = e1 == hv;
            if ((e0 == hv) || m1) {
                const u32 qr = rel + j + (m1 ? 1u : 0u);
                __hip_atomic_fetch_or(&my_acc[(size_t)(qr >> 6) * kTileGenomes], make_u64(one, zero) << (qr & 63u),
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        auto probe_walk = [&](u64 hv) {
            if (ABLATE == 2) { cur_bits ^= hv; return; }
            if (hv < lo || hv > hi) return;  // also drops the padding value
            const u32 bk = (u32)((hv - lo) >> shift);
            u32 j = (multi || BIG) ? (u32)dir16[bk] : (u32)dir[bk];
            u64 e = sl[j];
            while (e < hv) e = sl[++j];   // the sentinel ends every walk (hv <= hi < kEmpty)
            if (ABLATE == 3) { cur_bits ^= e; return; }
            if (e == hv) {
                if (multi) hit_atomic(sub + j);
                else {
                    const u32 qr = rel + j;
                    my_acc[(size_t)(qr >> 6) * kTileGenomes] |= 1ull << (qr & 63u);
                }
            }
        };
        // software-pipelined: the next kU rows are in flight while the current kU are probed (kU = 8: 16 loads in flight per
        // lane.  Measured with -DSKX_SCAN_ROWS=16 / 32: more bytes in flight per wave at fewer resident waves loses --
        // 75 -> 67 -> 52 M reads/s at C2, scan 0.90 -> 1.2 -> 1.6 ms in the pipeline)
        constexpr u32 kU = SKX_SCAN_ROWS;
        const u64* const band = mat + ((size_t)t * s + i0) * kTileGenomes;  // wave-uniform
        auto row = [&](u32 r) -> u64 {
            const u64* rp = band + (size_t)r * kTileGenomes;  // (uniform: scalar arithmetic)
            return (NT & 2) ? __builtin_nontemporal_load(&rp[c]) : rp[c];
        };
        u32 i = 0;
        if (rows >= kU) {
            u64 h[kU];
#pragma unroll
            for (u32 u = 0; u < kU; ++u) h[u] = row(u);
            for (i = kU; i + kU <= rows; i += kU) {
                u64 hn[kU];
#pragma unroll
                for (u32 u = 0; u < kU; ++u) hn[u] = row(i + u);
                if (lean) {
#pragma unroll
                    for (u32 u = 0; u < kU; ++u) probe_lean(h[u]);
                } else {
#pragma unroll
                    for (u32 u = 0; u < kU; ++u) probe_walk(h[u]);
                }
#pragma unroll
                for (u32 u = 0; u < kU; ++u) h[u] = hn[u];
            }
            if (lean) {
#pragma unroll
                for (u32 u = 0; u < kU; ++u) probe_lean(h[u]);
            } else {
#pragma unroll
                for (u32 u = 0; u < kU; ++u) probe_walk(h[u]);
            }
        }
        for (; i < rows; ++i) probe_walk(row(i));

        if (ABLATE >= 1) {
            if (cur_bits == 0x123456789ull) m_bits[g] = cur_bits;  // keep the work alive
        } else if (multi && cur_bits) {
            atomicOr(&m_bits[(size_t)cur_w * n_pad + g], cur_bits);
        }
        __syncthreads();  // slice and directory are rebuilt by the next sub-window
    }
    if (!multi && ABLATE != 1 && (NT & 4)) {
        // results straight into M[word][genome] by atomicOr, zero words skipped (NT & 4).  Measured (tools/ubench/stream_rates.hip,
        // the kernel's geometry as a pure stream): ANY plain store stream next to the matrix reads costs a fifth of the read
        // rate (slabs: 0.63-0.69 of peak against 0.81 for the reads alone, whatever the slabs' size), while a compact M --
        // 52 MB at C2, every word hit again by the next two or three bands of its tile -- stays on chip: 0.81-0.83 with
        // non-temporal matrix loads.  The transpose then reads M only (and zeroes it again).
        if (c == 0) *m_dirty = 1u;
        for (u32 k = 0; k < n_w; ++k) {
            const u64 v = acc[k][c];
            if (v) atomicOr(&m_bits[(size_t)(w0 + k) * n_pad + g], v);
        }
    } else if (!multi && ABLATE != 1) {
        // this block's slab: [kLeanWords][256] words, the first n_w of them are read by the transpose
        u64* out = hbuf + (size_t)bt * kLeanWords * kTileGenomes + c;
        for (u32 k = 0; k < n_w; ++k) {
            if (NT & 1) __builtin_nontemporal_store(acc[k][c], &out[(size_t)k * kTileGenomes]);
            else out[(size_t)k * kTileGenomes] = acc[k][c];
        }
    }
}

// =====================================================================================
// the reference scan, RUNS of bands (round 4 experiment: it LOST -- 0.57 of peak against the lean kernel's 0.68 -- and is
// compiled into the experiments build only, knob SKX_SCAN_RUN)
// =====================================================================================
// What the lean kernel above still paid for was its RESULTS: measured with the kernel's geometry as a pure stream
// (tools/ubench/scan_setup.hip, profiles/r04_scan_setup.txt) the set-up of a block -- window, slice, directory, two barriers --
// costs nothing (0.92 of the 8 TB/s peak with and without it), the probe on every element nothing either (0.90 with hits OR-ed
// into the LDS tile), and five atomicOr per lane and band into a compact M take it to 0.77: exactly where the real kernel sat
// (0.70).  A word of M is hit by the two or three adjacent bands of a tile whose slices overlap it, i.e. written two or three
// times.  Here ONE workgroup walks a RUN of `run` consecutive bands of its tile with ONE slice (the union of their windows: the
// windows of adjacent bands overlap by two thirds, so four bands need twice one band's entries, not four times) and ONE result
// tile, and flushes every word of the union once: ~8 words per four bands instead of 20.  The same tables serve the large slices
// of passes that eight batches share (C4: ~350 entries per band): up to kRunCap entries per window with a two-byte directory of
// 8192 buckets and a THREE-entry probe (entry dir[bucket] and its two successors; more than three entries in one bucket -- or a
// window of more than kRunCap entries: further windows, the rows streamed again -- take the walk probe / the window loop).
// 48 KB of LDS: three workgroups per CU, which the stream needs (two: 0.77; three and more: 0.91 -- same file).
constexpr u32 kRunCap = 768;      // entries per window
constexpr u32 kRunWords = 13;     // query words a window of kRunCap entries can touch
constexpr u32 kRunBuckets = 8192;
template <int NT>
__global__ __launch_bounds__(256) void scan_run_kernel(const u64* __restrict__ mat, u32 s, u32 n_tiles, u32 rb, u32 n_bands, u32 run,
                                                        const u64* __restrict__ q, const u32* __restrict__ win,
                                                        u64* __restrict__ m_bits, u32 n_pad, u32* __restrict__ m_dirty, u32 prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    __shared__ u64 slice[kRunCap + 4];
    __shared__ unsigned short dir[kRunBuckets + 8];
    __shared__ u64 acc[kRunWords][kTileGenomes];
    __shared__ u32 deep;
    const u32 bt = blockIdx.x;
    const u32 t = bt % n_tiles, r = bt / n_tiles, c = threadIdx.x;
    const u32 b0 = r * run, b1 = min(n_bands, b0 + run);
    // the union of the run's windows (uniform: scalar loads; empty bands hold [nq, nq) or lo > hi)
    u32 qa = 0xFFFFFFFFu, qb = 0;
    for (u32 b = b0; b < b1; ++b) {
        const u32 a = win[2 * (b * n_tiles + t)], z = win[2 * (b * n_tiles + t) + 1];
        if (a < z) { qa = min(qa, a); qb = max(qb, z); }
    }
    if (qa >= qb) return;
    const u32 i0 = b0 * rb, rows = min(s, b1 * rb) - i0;
    const u32 g = t * kTileGenomes + c;
    if (c == 0 && bt == 0) *m_dirty = 1u;
    const u64* const band = mat + ((size_t)t * s + i0) * kTileGenomes;  // wave-uniform
    for (u32 sub = qa; sub < qb; sub += kRunCap) {
        const u32 n = min(kRunCap, qb - sub);
        const u64 lo = q[sub], hi = q[sub + n - 1];
        constexpr u32 bk_bits = 13;
        static_assert((1u << bk_bits) == kRunBuckets, "directory size");
        const u32 span_bits = 64u - (u32)__clzll((hi - lo) | 1ull);
        const u32 shift = span_bits > bk_bits ? span_bits - bk_bits : 0u;
        for (u32 i = c; i < n; i += kTileGenomes) slice[i] = q[sub + i];
        if (c < 3u) slice[n + c] = kEmpty;
        if (c == 0) deep = shift < 32u ? 1u : 0u;
        const u32 w0 = sub >> 6, n_w = ((sub + n - 1u) >> 6) - w0 + 1u;  // (<= kRunWords)
#pragma unroll
        for (u32 k = 0; k < kRunWords; ++k) acc[k][c] = 0;  // (only this lane ever touches column c)
        __syncthreads();
        for (u32 i = c; i <= n; i += kTileGenomes) {
            // entry i opens every bucket in (bucket(i-1), bucket(i)]; the sentinel closes the rest
            const u32 bj = i < n ? (u32)((slice[i] - lo) >> shift) : kRunBuckets;
            const u32 bp = i == 0 ? 0xFFFFFFFFu : (u32)((slice[i - 1] - lo) >> shift);
            for (u32 x = bp + 1u; x <= bj; ++x) dir[x] = (unsigned short)i;
            if (i >= 3u && i < n && (u32)((slice[i - 3] - lo) >> shift) == bj) deep = 1u;  // four entries in one bucket (benign race: same value)
        }
        __syncthreads();
        const bool lean = deep == 0u;
        const u32 sh_hi = shift - 32u;  // (lean only: shift >= 32)
        const u32 lo_lo = (u32)lo;
        u32 lo_hi = (u32)(lo >> 32), one = 1, zero = 0, rel = sub - (w0 << 6);
        asm volatile("" : "+v"(lo_hi), "+v"(one), "+v"(zero), "+v"(rel));  // (vector registers on purpose: see scan_lean_kernel)
        u64* my_acc = &acc[0][c];
        auto probe_lean = [&](u64 hv) {
            const u32 dh = (u32)(hv >> 32) - lo_hi - ((u32)hv < lo_lo ? 1u : 0u);  // high word of (hv - lo)
            const u32 bk = min(dh >> sh_hi, kRunBuckets);
            const u32 j = dir[bk];
            const u64 e0 = slice[j], e1 = slice[j + 1u], e2 = slice[j + 2u];
            const bool m1 = e1 == hv, m2 = e2 == hv;
            if ((e0 == hv) || m1 || m2) {
                const u32 qr = rel + j + (m1 ? 1u : 0u) + (m2 ? 2u : 0u);
                __hip_atomic_fetch_or(&my_acc[(size_t)(qr >> 6) * kTileGenomes], make_u64(one, zero) << (qr & 63u),
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        auto probe_walk = [&](u64 hv) {
            if (hv < lo || hv > hi) return;  // also drops the padding value
            u32 j = dir[(u32)((hv - lo) >> shift)];
            u64 e = slice[j];
            while (e < hv) e = slice[++j];   // the sentinel ends every walk (hv <= hi < kEmpty)
            if (e == hv) {
                const u32 qr = rel + j;
                my_acc[(size_t)(qr >> 6) * kTileGenomes] |= 1ull << (qr & 63u);
            }
        };
        constexpr u32 kU = SKX_SCAN_ROWS;
        auto row = [&](u32 rr) -> u64 {
            const u64* rp = band + (size_t)rr * kTileGenomes;  // (uniform: scalar arithmetic)
            return (NT & 2) ? __builtin_nontemporal_load(&rp[c]) : rp[c];
        };
        u32 i = 0;
        if (rows >= kU) {
            u64 h[kU];
#pragma unroll
            for (u32 u = 0; u < kU; ++u) h[u] = row(u);
            for (i = kU; i + kU <= rows; i += kU) {
                u64 hn[kU];
#pragma unroll
                for (u32 u = 0; u < kU; ++u) hn[u] = row(i + u);
                if (lean) {
#pragma unroll
                    for (u32 u = 0; u < kU; ++u) probe_lean(h[u]);
                } else {
#pragma unroll
                    for (u32 u = 0; u < kU; ++u) probe_walk(h[u]);
                }
#pragma unroll
                for (u32 u = 0; u < kU; ++u) h[u] = hn[u];
            }
            if (lean) {
#pragma unroll
                for (u32 u = 0; u < kU; ++u) probe_lean(h[u]);
            } else {
#pragma unroll
                for (u32 u = 0; u < kU; ++u) probe_walk(h[u]);
            }
        }
        for (; i < rows; ++i) probe_walk(row(i));
        // the window's words, once: zero words skipped (M stays all-zero where nothing hit)
        for (u32 k = 0; k < n_w; ++k) {
            const u64 v = acc[k][c];
            if (v) atomicOr(&m_bits[(size_t)(w0 + k) * n_pad + g], v);
        }
        __syncthreads();  // slice, directory and tile are rebuilt by the next window
    }
}

// Which bands of a tile reach query word w?  A band's slice [qa, qb) covers the words qa >> 6 .. (qb - 1) >> 6; qa grows
// with the band (the smallest hash of a band does), qb need not (ragged columns), so the answer is the candidate range
// [first band whose PREFIX MAXIMUM of last words reaches w, last band whose first word is <= w] -- the transpose tests
// each band of the range -- here, once per (word, tile), so that transpose_bits_kernel only follows ready-made slab
// indices.  One block per tile; wb[w * n_tiles + t] = four slab word indices (kWbNone = unused).
constexpr u32 kWordBandsMax = 2048;  // bands per tile the LDS tables hold (s <= 131 072 at 64 rows per band)
constexpr u32 kWbNone = 0xFFFFFFFFu, kWbRange = 0xFFFFFFFEu;
// lo != NULL: the block first computes the windows of its tile's bands itself (window_kernel's work: win[2 bt], win[2 bt + 1] =
// the slice of Q inside [lo[bt], hi[bt]]) -- one launch instead of two on the chain in front of the scan.
__global__ __launch_bounds__(256) void word_bands_kernel(u32* __restrict__ win, u32 n_tiles, u32 n_bands,
                                                         const u32* __restrict__ n_q, u32* __restrict__ wb,
                                                         const u64* __restrict__ lo, const u64* __restrict__ hi,
                                                         const u64* __restrict__ q, volatile u32* __restrict__ h_nq) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    __shared__ u32 first_w[kWordBandsMax], pmax_w[kWordBandsMax];
    __shared__ u32 wtot[4];
    const u32 t = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    if (lo) {
        const u32 nq = *n_q;
        if (t == 0 && tid == 0 && h_nq) { *h_nq = nq; __threadfence_system(); }  // |Q| for the host (page-locked memory), a hint only
        for (u32 b = tid; b < n_bands; b += 256u) {
            const u32 bt = b * n_tiles + t;
            u32 qa = nq, qb = nq;  // (an empty band gets the empty window [nq, nq): window_kernel)
            if (lo[bt] <= hi[bt]) { qa = lower_bound_u64(q, nq, lo[bt]); qb = upper_bound_u64(q, nq, hi[bt]); }
            win[2 * bt] = qa;
            win[2 * bt + 1] = qb;
        }
        __syncthreads();  // (the block reads its own writes below: same workgroup, global memory)
    }
    const u32 n_words = (*n_q + 63u) >> 6;
    // first word / (last word + 1) per band; empty bands reach nothing
    u32 carry = 0;  // prefix maximum of (last word + 1) so far
    for (u32 b0 = 0; b0 < n_bands; b0 += 256u) {
        const u32 b = b0 + tid;
        u32 fw = 0xFFFFFFFFu, lw1 = 0;
        if (b < n_bands) {
            const u32 qa = win[2 * (b * n_tiles + t)], qb = win[2 * (b * n_tiles + t) + 1];
            fw = qa >> 6;  // (monotone in b, also for empty bands: window_kernel)
            if (qa < qb) lw1 = ((qb - 1u) >> 6) + 1u;
        }
        // inclusive prefix maximum over the block
        u32 v = lw1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 o = (u32)__shfl_up((int)v, d, 64);
            if ((int)lane >= d) v = max(v, o);
        }
        if (lane == 63u) wtot[wv] = v;
        __syncthreads();
        u32 before = carry;
        for (u32 x = 0; x < wv; ++x) before = max(before, wtot[x]);
        v = max(v, before);
        if (b < n_bands) { first_w[b] = fw; pmax_w[b] = v; }
        const u32 tot = max(max(max(wtot[0], wtot[1]), max(wtot[2], wtot[3])), carry);
        __syncthreads();
        carry = tot;
    }
    __syncthreads();
    for (u32 w = tid; w < n_words; w += 256u) {
        // hi = last band with first_w <= w  (upper bound - 1);  lo = first band with pmax_w >= w + 1
        u32 a = 0, z = n_bands;
        while (a < z) { const u32 m = (a + z) >> 1; if (first_w[m] <= w) a = m + 1; else z = m; }
        const u32 hi = a;  // (one past)
        a = 0; z = n_bands;
        while (a < z) { const u32 m = (a + z) >> 1; if (pmax_w[m] < w + 1u) a = m + 1; else z = m; }
        const u32 lo = a;
        // up to four slab words (index into hbuf in units of 256 u64) that hold bits of query word w for this tile;
        // more than four bands reaching one word (ragged / dense references): e[3] = kWbRange, e[0] = lo | last << 16
        uint4 e = make_uint4(kWbNone, kWbNone, kWbNone, kWbNone);
        u32 cnt = 0;
        for (u32 b = lo; b < hi; ++b) {
            const u32 bt = b * n_tiles + t;
            const u32 qa = win[2 * bt], qb = win[2 * bt + 1];
            if (qa < qb && qb - qa <= kLeanCap && (qa >> 6) <= w && w <= ((qb - 1u) >> 6)) {
                const u32 word = bt * kLeanWords + (w - (qa >> 6));
                if (cnt == 0) e.x = word; else if (cnt == 1) e.y = word; else if (cnt == 2) e.z = word; else if (cnt == 3) e.w = word;
                ++cnt;
            }
        }
        if (cnt > 4u) e = make_uint4(lo | ((hi - 1u) << 16), kWbNone, kWbNone, kWbRange);
        reinterpret_cast<uint4*>(wb)[(size_t)w * n_tiles + t] = e;
    }
}

// =====================================================================================
// M[word][genome] (bit j of the word = query 64*word + j)  ->  Mq[query][genome word]
// =====================================================================================
// Partner exchange x[lane ^ D] for the butterfly, without the LDS crossbar (one LDS pipe per CU is shared by
// all waves): gfx950's v_permlane32_swap / v_permlane16_swap for D = 32 / 16, DPP row ops for D <= 8.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <int D>
__device__ __forceinline__ u32 xor_lane(u32 v, u32 lane) {
    if constexpr (D == 32) {
        const u32x2_t r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // r.x = [lo|lo], r.y = [hi|hi]
        return (lane & 32u) ? r.x : r.y;
    } else if constexpr (D == 16) {
        const u32x2_t r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // per 32-lane half, at 16
        return (lane & 16u) ? r.x : r.y;
    } else if constexpr (D == 8) {
        return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);            // row_ror:8
    } else if constexpr (D == 4) {
        int p = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);                  // row_shl:4 -> lanes 0-3, 8-11
        return (u32)__builtin_amdgcn_update_dpp(p, (int)v, 0x114, 0xF, 0xA, false);              // row_shr:4 -> lanes 4-7, 12-15
    } else if constexpr (D == 2) {
        return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);               // quad_perm [2,3,0,1]
    } else {
        return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);               // quad_perm [1,0,3,2]
    }
}

// 64x64 bit transpose across the wave: lane l holds row l; six butterfly steps swap the off-diagonal d x d
// blocks between lanes l and l^d; afterwards lane l holds column l (bit j = old bit l of lane j).
__device__ __forceinline__ u64 transpose64(u64 x, u32 lane) {
#define SKX_BFLY(D, LO)                                                                             \
    {                                                                                              \
        const u64 p = ((u64)xor_lane<D>((u32)(x >> 32), lane) << 32) | xor_lane<D>((u32)x, lane);   \
        x = (lane & D) ? ((x & ~(LO)) | ((p >> D) & (LO))) : ((x & (LO)) | ((p << D) & ~(LO)));    \
    }
    SKX_BFLY(32, 0x00000000FFFFFFFFull)  // LO = bits whose index has bit D clear
    SKX_BFLY(16, 0x0000FFFF0000FFFFull)
    SKX_BFLY(8, 0x00FF00FF00FF00FFull)
    SKX_BFLY(4, 0x0F0F0F0F0F0F0F0Full)
    SKX_BFLY(2, 0x3333333333333333ull)
    SKX_BFLY(1, 0x5555555555555555ull)
#undef SKX_BFLY
    return x;
}

// One block = 4 waves = one rank group (8 genome words = 512 genomes, two per wave) x kWordsPerBlock consecutive query
// words, strided over the word groups of the dictionary.  All words of a group are requested first; per word every wave
// transposes its two 64x64 bit blocks with the butterfly, and the block stages [64 queries][8 words] in LDS and writes
// 64 x 64 B = 4 KB contiguous of the group-major Mq.
constexpr u32 kWordsPerBlock = 4;
// The kernel also restores the "all zero between passes" state of the word arrays (only words that were set are
// written back): no memset of 2 x |M| bytes per pass.
// grp_any[grp] (zero on entry) is raised when the group's slice of the matrix holds any bit at all: rank groups without
// one -- a whole species the sample does not belong to, for instance -- are skipped by every kernel of the back half.
// hbuf != NULL: scan_lean_kernel's per-(band, tile) slabs are OR-ed in (the bands word_bands_kernel lists in wb, each
// tested against its window), and M itself is only read when *m_dirty says somebody wrote it.
__global__ __launch_bounds__(256) void transpose_bits_kernel(u64* __restrict__ m_bits, u64* __restrict__ m_int,
                                                             u32 n_pad, u32 n_words, u64* __restrict__ mq, u32 n_gw,
                                                             const u32* __restrict__ n_q, u32* __restrict__ grp_any,
                                                             const u64* __restrict__ hbuf, const u32* __restrict__ wb,
                                                             const u32* __restrict__ win, u32 n_tiles,
                                                             const u32* __restrict__ m_dirty, u64* __restrict__ rowany,
                                                             const u32* __restrict__ only_if, u32 keep_m) {
    __builtin_amdgcn_s_setprio(2);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    if (only_if && !*only_if) return;  // (every batch of the pass ranks on its candidates' compact matrix: nobody reads this one)
    __shared__ u64 tile[2][64][kRankWords + 1];
    // the grid is sized by the pair count (all the host knows); only the first ceil(nq / 64) words exist -- pairs
    // index Q, so rows of Mq beyond nq are never read
    // (... and the grid's y extent comes from the host's ESTIMATE of nq -- tens of thousands of 512-thread blocks that
    // only find out they have nothing to do cost the scan stream 0.3-0.4 ms next to the other streams' kernels -- so a
    // block strides over the word groups: any estimate is correct, a good one is fast)
    const u32 live_words = min(n_words, (*n_q + 63u) >> 6);
    const u32 grp = blockIdx.x;
    const u32 wv = threadIdx.x >> 6, lane = lane_id();
    u32 nz_rows = 0;  // (wave 0, lane 0) rows of this group that hold any bit, over the block's words
    for (u32 w0 = blockIdx.y * kWordsPerBlock; w0 < live_words; w0 += gridDim.y * kWordsPerBlock) {
    const u32 w1 = min(live_words, w0 + kWordsPerBlock);
    const bool read_m = hbuf == nullptr || *m_dirty != 0u;
    // (four waves, two genome words each: a 512-thread block needs eight free wave slots on one CU at once, which it
    // waits for next to the other streams' 256-thread blocks)
    auto load = [&](u32 w, u32 half) -> u64 {
        const u32 gw = grp * kRankWords + wv + 4u * half;
        const bool on = gw < n_gw;
        const size_t col = (size_t)gw * 64u + lane;
        const u32 ref_tile = gw / (kTileGenomes / 64u), tc = (gw % (kTileGenomes / 64u)) * 64u + lane;  // this lane's genome in its tile
        if (!on || w >= w1) return 0;
        u64 x = 0;
        if (read_m) {
            x = m_bits[(size_t)w * n_pad + col];
            if (x && !keep_m) m_bits[(size_t)w * n_pad + col] = 0;  // (keep_m: the static dense rows, kept for the passes to come)
            if (m_int) {
                const u64 y = m_int[(size_t)w * n_pad + col];
                if (y) { m_int[(size_t)w * n_pad + col] = 0; x |= y; }
            }
        }
        if (hbuf) {
            const uint4 e = reinterpret_cast<const uint4*>(wb)[(size_t)w * n_tiles + ref_tile];  // (wave-uniform)
            if (e.w != kWbRange) {
                if (e.x != kWbNone) x |= hbuf[(size_t)e.x * kTileGenomes + tc];
                if (e.y != kWbNone) x |= hbuf[(size_t)e.y * kTileGenomes + tc];
                if (e.z != kWbNone) x |= hbuf[(size_t)e.z * kTileGenomes + tc];
                if (e.w != kWbNone) x |= hbuf[(size_t)e.w * kTileGenomes + tc];
            } else {  // more than four bands reach this word: walk the candidate range
                for (u32 b = e.x & 0xFFFFu; b <= (e.x >> 16); ++b) {
                    const u32 bt = b * n_tiles + ref_tile;
                    const u32 qa = win[2 * bt], qb = win[2 * bt + 1];
                    if (qa < qb && qb - qa <= kLeanCap && (qa >> 6) <= w && w <= ((qb - 1u) >> 6))
                        x |= hbuf[((size_t)bt * kLeanWords + (w - (qa >> 6))) * kTileGenomes + tc];
                }
            }
        }
        return x;
    };
    const u32 row = threadIdx.x >> 3, cw = threadIdx.x & 7u;
    // all of the group's words are requested before the first is used: next to the other streams' kernels a memory round
    // trip takes several microseconds, and one per word in a row was most of this kernel's time
    u64 xs[kWordsPerBlock][2];
#pragma unroll
    for (u32 i = 0; i < kWordsPerBlock; ++i) { xs[i][0] = load(w0 + i, 0u); xs[i][1] = load(w0 + i, 1u); }
#pragma unroll
    for (u32 i = 0; i < kWordsPerBlock; ++i) {
        const u32 w = w0 + i;
        if (w >= w1) break;
        const u32 bsel = i & 1u;
        tile[bsel][lane][wv] = transpose64(xs[i][0], lane);
        tile[bsel][lane][wv + 4u] = transpose64(xs[i][1], lane);
        __syncthreads();  // (double-buffered tile: one barrier per word is enough)
        if (wv == 0u) {
            // rowany[grp][w]: bit r = query row 64 w + r holds a bit for some genome of this group (the ranking's kernels skip
            // the all-zero rows of sparse groups); the group's count of such rows replaces the old "any bit" flag
            u64 any = 0;
#pragma unroll
            for (u32 c = 0; c < (u32)kRankWords; ++c) any |= tile[bsel][lane][c];
            const u64 rm = __ballot(any != 0ull);
            if (lane == 0u) {
                if (rowany) rowany[(size_t)grp * n_words + w] = rm;
                nz_rows += (u32)__popcll(rm);
            }
        }
        if (grp * kRankWords + cw < n_gw) {
            mq[mq_index(grp * kRankWords + cw, w * 64u + row, n_words * 64u)] = tile[bsel][row][cw];
            mq[mq_index(grp * kRankWords + cw, w * 64u + row + 32u, n_words * 64u)] = tile[bsel][row + 32u][cw];
        }
    }
    __syncthreads();  // (the tile buffers are reused by the block's next word group)
    }
    if (wv == 0u && lane == 0u && nz_rows) atomicAdd(&grp_any[grp], nz_rows);  // rows with any bit: 0 = the group is dead this pass
}

// =====================================================================================
// running table and per-read ranking
// =====================================================================================
// Reads of a pass are cut into segments of seg_len reads.  pair_r/pair_q are sorted by read;
// poff[r] (absolute, minus p_base) delimits read r's pairs.
//
// acc += bit `lane` of a wave-uniform 64-bit mask: ONE VALU op (the mask is the carry-in of v_addc).
__device__ __forceinline__ u32 add_lane_bit(u32 acc, u64 mask) {
    u32 out;
    asm("v_addc_co_u32_e64 %0, vcc, 0, %1, %2" : "=v"(out) : "v"(acc), "s"(mask) : "vcc");
    return out;
}

// One pair's kRankWords mask words (one rank group = 512 genomes): 64 contiguous bytes = one memory sector.
struct __attribute__((aligned(16))) MaskVec { u64 w[kRankWords]; };

// Lane j of the wave fetches the mask words of pair p0 + j: 64 independent 64-byte gathers in flight.
// mq_g = this group's [nq_rows][kRankWords] slice of the group-major bit matrix.
__device__ __forceinline__ MaskVec gather_vec(const u64* __restrict__ mq_g, u32 q, bool on) {
    MaskVec m;
#pragma unroll
    for (int j = 0; j < kRankWords; ++j) m.w[j] = 0;
#if SKX_MQ_NT  // (measurement: tools/build_variant.sh mqnt -DSKX_MQ_NT=1 -- non-temporal gathers of the ranking's rows)
    if (on) {
#pragma unroll
        for (int j = 0; j < kRankWords; ++j) m.w[j] = __builtin_nontemporal_load(mq_g + (size_t)q * kRankWords + j);
    }
#else
    if (on) m = *reinterpret_cast<const MaskVec*>(mq_g + (size_t)q * kRankWords);
#endif
    return m;
}

// seg_sum: inc[seg][g] = sum over the segment's pairs of bit(Mq[q][g]).
// One wave per (rank group of 512 genomes = 8 mask words, segment), counting with BIT-SLICED counters instead of
// transposing the pair x genome bit matrix:
//   * lane = (sub = lane / 8, word j = lane % 8): one load instruction fetches the 64-byte mask rows of 8 pairs
//     (pair p0 + 8u + sub for row slot u), 8 lanes per row;
//   * each lane adds its rows (about 37 for a 64-read segment at C2) into counter planes ones / twos / fours / ...
//     with a Harley-Seal carry-save tree: 7 CSAs per 8 rows, bit g of plane b = bit b of the number of this lane's
//     rows that hit genome g of word j;
//   * the 8 sub-slots of a word are merged by a bit-sliced reduce-scatter -- lanes l and l^32 exchange the halves
//     they do not keep (one v_permlane32_swap per plane), then l^16 (v_permlane16_swap), then l^8 (DPP row_ror:8),
//     each followed by a ripple add of the two plane stacks -- after which lane (sub, j) holds 9 planes x 8 bits:
//     the totals of genomes sub*8 .. sub*8+7 of word j;
//   * those 8 counts are extracted once per block of <= 448 pairs and stored as 32 contiguous bytes per lane.
constexpr u32 kSparseQueue = 256;  // query rows a wave parks in LDS while it compacts a sparse group's pairs (<= kBlockPairs)
// carry-save adder of three bit planes: L = A ^ B ^ C, H = majority(A, B, C) -- both symmetric three-input functions, i.e. ONE
// v_bitop3_b32 each per 32-bit half on gfx950 (truth tables 0x96 / 0xE8 under any operand order).  Written as xor / and / or
// the compiler shares A ^ B between the two and ends up with five instructions per half (ISA inspected: 96 -> 54 VALU
// instructions per step of 64 pairs).
__device__ __forceinline__ u64 bitop3_u64_xor3(u64 a, u64 b, u64 c) {
    return make_u64(__builtin_amdgcn_bitop3_b32((u32)a, (u32)b, (u32)c, 0x96),
                    __builtin_amdgcn_bitop3_b32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32), 0x96));
}
__device__ __forceinline__ u64 bitop3_u64_maj(u64 a, u64 b, u64 c) {
    return make_u64(__builtin_amdgcn_bitop3_b32((u32)a, (u32)b, (u32)c, 0xE8),
                    __builtin_amdgcn_bitop3_b32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32), 0xE8));
}
#define SKX_CSA(H, L, A, B, C)                           \
    {                                                    \
        const u64 a_ = (A), b_ = (B), c_ = (C);          \
        H = bitop3_u64_maj(a_, b_, c_);                  \
        L = bitop3_u64_xor3(a_, b_, c_);                 \
    }
// x (N planes) += y (N planes), both bit-sliced little-endian; the carry out becomes plane N of x
template <int N>
__device__ __forceinline__ void add_planes(u32 (&x)[12], const u32 (&y)[12]) {
    u32 c = 0;
#pragma unroll
    for (int b = 0; b < N; ++b) {  // a full adder per plane: sum = xor3, carry = majority (one v_bitop3_b32 each)
        const u32 cn = __builtin_amdgcn_bitop3_b32(x[b], y[b], c, 0xE8);
        x[b] = __builtin_amdgcn_bitop3_b32(x[b], y[b], c, 0x96);
        c = cn;
    }
    x[N] = c;
}
// CHUNK (round 3): the same counting for a whole CHUNK per workgroup -- a wave takes a quarter chunk (seg_len = 256 reads),
// eight planes hold up to 255 rows per lane (one extraction per ~1 984 pairs instead of one per segment: the extraction was
// 60 % of a segment wave's instructions), only the chunk sums leave.  The pruned rankings run THIS over everything and the
// per-segment kernel only over the (chunk, rank group)s that can hold a candidate (chunk_group_live: known once the chunk
// sums are prefixed) -- far from the start of a sample that is the leaders' groups and little else, and nobody reads the
// per-segment increments of the others.
// (the test is described with the chunk-level pruning below)
__device__ __forceinline__ bool chunk_group_live(const u64* __restrict__ gmax, const u64* __restrict__ lead_val,
                                                 u32 n_half, u32 c, u32 grp, const Species& sp) {
    const u64 lv = lead_val[c * sp.n_sp + sp.of_grp[grp]];
    const u64* gm = gmax + (size_t)(c + 1u) * n_half + 2u * grp;  // (n_half = 2 x rank groups: n_pad is a multiple of 512)
    return gm[0] >= lv || gm[1] >= lv;
}

struct ChunkLive { const u64* gmax = nullptr; const u64* lead_val = nullptr; u32 n_half = 0; };
template <bool CHUNK>
__global__ __launch_bounds__(256) void seg_sum_kernel(const u32* __restrict__ pair_q, const u32* __restrict__ poff,
                                                      u32 p_base, u32 r_begin, u32 n_reads, u32 seg_len,
                                                      const u64* __restrict__ mq, u32 n_gw, u32 n_pad,
                                                      u32 nq_rows, u32* __restrict__ inc, const u32* __restrict__ grp_any,
                                                      u32* __restrict__ qsum, const u64* __restrict__ rowany,
                                                      const u32* __restrict__ n_q, ChunkLive cl, Species sp) {
    constexpr int NP = CHUNK ? 8 : 6;  // counter planes per lane: up to 2^NP - 1 rows per lane and block
    __builtin_amdgcn_s_setprio(SKX_SEGSUM_PRIO);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    static_assert(kRankWords == 8, "lane = (sub, word) layout assumes 8 words per rank group");
    const u32 lane = lane_id();
    const u32 n_seg = (n_reads + seg_len - 1) / seg_len, n_grp = (n_gw + kRankWords - 1) / kRankWords;
    // XCD-aware task order: workgroups go round-robin to the 8 XCDs, each with its own 4 MB L2.  XCD x takes the
    // groups x, x+8, ... and walks them one after the other (all segments of a group before the next group), so the
    // rows being gathered -- one group's slice of Mq, 64 B x |Q| -- stay in that XCD's L2.
    // A workgroup = four consecutive segments of one group; their sum is added to the chunk's row of `qsum` (= the chunk
    // sums csum_raw[chunk][g], zeroed by the caller: four workgroups per chunk, one coalesced atomic add each) as the
    // increments leave -- no separate pass that re-reads all 16 segment rows of every chunk (247 MB per batch).
    // (Summed through 8 KB of LDS first.  Every wave adding its own counts straight to global memory -- no LDS, four times
    // the atomics, scattered 32 bytes per lane -- was measured: 74 -> 39 M reads/s.)
    __shared__ u32 red[4][kRankWords * 64];
    __shared__ u32 squeue[4][kSparseQueue];
    const u32 xcd = blockIdx.x & 7u, wv = threadIdx.x >> 6;
    const u32 n_q4 = (n_seg + 3u) / 4u, blk = blockIdx.x >> 3;
    const u32 grp = (blk / n_q4) * 8u + xcd, seg4 = blk % n_q4;
    const u32 seg = __builtin_amdgcn_readfirstlane(seg4 * 4u + wv);
    if (grp >= n_grp || !grp_any[grp]) return;  // (a group without any bit: nobody reads its increments; the whole block leaves)
    // per-segment increments are only read for (chunk, group)s that can hold a candidate (seg_prefix / the pruned rankings test
    // exactly this): the whole block -- four segments of one chunk -- leaves
    if (!CHUNK && cl.gmax && !chunk_group_live(cl.gmax, cl.lead_val, cl.n_half, (seg4 * 4u) >> 4, grp, sp)) return;
    const bool on = seg < n_seg;
    const u32 sub = lane >> 3, j = lane & 7u;
    const u32 ra = seg * seg_len, rz = min(n_reads, ra + seg_len);
    const u32 pa = on ? poff[r_begin + ra] - p_base : 0u, pz = on ? poff[r_begin + rz] - p_base : 0u;
    u32 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0;
    constexpr u32 kBlockPairs = CHUNK ? 8u * 248u : 8u * 56u;  // rows per lane and block <= 2^NP - 1: the planes hold the lane's counts
    // Addressing kept off the VALU: the group's slice of Mq and the pair list are wave-uniform bases (scalar registers), a
    // lane's row word is at byte offset (q << 6 | j * 8) and its slot of the pair list at 4 * (p0 + sub) + 32 u -- both fit
    // 32 bits (a pass holds at most 2^22 pairs / rows), so a load costs one VALU instruction instead of five.  Full steps
    // of 64 pairs take no bounds tests at all; only a block's last, partial step masks the rows past its end (re-read
    // from the block's last pair: always a valid address).  (Conditional loads had become sixteen exec-mask branches per
    // step: 160 VALU instructions per step, 80 of them the counting itself.)
    const char* const mq_base = reinterpret_cast<const char*>(mq + (size_t)grp * nq_rows * kRankWords);
    const char* const pq_base = reinterpret_cast<const char*>(pair_q);
    const u32 j8 = j * 8u;
#if SKX_MQ_NT
    auto row_at = [&](u32 q) -> u64 { return __builtin_nontemporal_load(reinterpret_cast<const u64*>(mq_base + (size_t)((q << 6) | j8))); };
#else
    auto row_at = [&](u32 q) -> u64 { return *reinterpret_cast<const u64*>(mq_base + (size_t)((q << 6) | j8)); };
#endif
    // one block: cnt <= kBlockPairs pairs, q_at(i) = the query row of its i-th pair (i < cnt)
    auto count_block = [&](u32 cnt, auto q_at) {
        u64 ones = 0, twos = 0, fours = 0;
        u64 hi[NP - 3];  // eights, sixteens, ...
#pragma unroll
        for (int b = 0; b < NP - 3; ++b) hi[b] = 0;
        auto count8 = [&](const u64 (&x)[8]) {
            u64 twos_a, twos_b, fours_a, fours_b, eights_a;
            SKX_CSA(twos_a, ones, ones, x[0], x[1])
            SKX_CSA(twos_b, ones, ones, x[2], x[3])
            SKX_CSA(fours_a, twos, twos, twos_a, twos_b)
            SKX_CSA(twos_a, ones, ones, x[4], x[5])
            SKX_CSA(twos_b, ones, ones, x[6], x[7])
            SKX_CSA(fours_b, twos, twos, twos_a, twos_b)
            SKX_CSA(eights_a, fours, fours, fours_a, fours_b)
            u64 c = eights_a;  // ripple into the higher planes (no carry out of the last one: rows per lane <= 2^NP - 1)
#pragma unroll
            for (int b = 0; b < NP - 3; ++b) {
                const u64 t = hi[b] & c;
                hi[b] ^= c;
                c = t;
            }
        };
        // the query indices of the next 64 pairs are requested before the current rows are counted (one memory round trip
        // per step instead of two in a row; requesting the rows ahead as well needs 82 VGPRs and lost: 417 -> 725 us)
        auto load_q = [&](u32 i0, u32 (&qv)[8]) {
            if (i0 + 64u <= cnt) {
#pragma unroll
                for (u32 u = 0; u < 8u; ++u) qv[u] = q_at(i0 + 8u * u + sub);
            } else {
#pragma unroll
                for (u32 u = 0; u < 8u; ++u) qv[u] = q_at(min(i0 + 8u * u + sub, cnt - 1u));
            }
        };
        u32 qn[8];
        u32 i0 = 0;
        load_q(0, qn);
        for (; i0 + 64u <= cnt; i0 += 64u) {  // full steps
            u64 x[8];
#pragma unroll
            for (u32 u = 0; u < 8u; ++u) x[u] = row_at(qn[u]);
            if (i0 + 64u < cnt) load_q(i0 + 64u, qn);
            count8(x);
        }
        if (i0 < cnt) {  // the last, partial step
            u64 x[8];
#pragma unroll
            for (u32 u = 0; u < 8u; ++u) {
                const u64 row = row_at(qn[u]);
                x[u] = i0 + 8u * u + sub < cnt ? row : 0ull;
            }
            count8(x);
        }
        // ---- merge the 8 sub-slots of every word (reduce-scatter over lane bits 5, 4, 3), planes get narrower
        u64 pl[NP];
        pl[0] = ones; pl[1] = twos; pl[2] = fours;
#pragma unroll
        for (int b = 0; b < NP - 3; ++b) pl[3 + b] = hi[b];
        u32 x[12], y[12];
        // lanes l / l^32: lower lanes keep genomes 0..31 of the word, upper lanes 32..63
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            const u32x2_t r = __builtin_amdgcn_permlane32_swap((u32)pl[b], (u32)(pl[b] >> 32), false, false);
            x[b] = r.x; y[b] = r.y;  // lower lanes: (own low half, partner's low half); upper: (partner's high, own high)
        }
        add_planes<NP>(x, y);  // NP + 1 planes x 32 bits
        // lanes l / l^16: 16 genomes each
#pragma unroll
        for (int b = 0; b < NP + 1; ++b) {
            const u32x2_t r = __builtin_amdgcn_permlane16_swap(x[b] & 0xFFFFu, x[b] >> 16, false, false);
            x[b] = r.x; y[b] = r.y;
        }
        add_planes<NP + 1>(x, y);  // NP + 2 planes x 16 bits
        // lanes l / l^8: 8 genomes each
        const bool up = (lane & 8u) != 0u;
#pragma unroll
        for (int b = 0; b < NP + 2; ++b) {
            const u32 lo = x[b] & 0xFFu, hi8 = x[b] >> 8;
            const u32 send = up ? lo : hi8;
            x[b] = up ? hi8 : lo;
            y[b] = (u32)__builtin_amdgcn_update_dpp(0, (int)send, 0x128, 0xF, 0xF, false);  // row_ror:8 = lane ^ 8
        }
        add_planes<NP + 2>(x, y);  // NP + 3 planes x 8 bits: bit i of plane b = bit b of the count of genome sub*8 + i
        // counts out of the planes: the 8 x 8 bits of planes 0..7 are one 64-bit bit matrix (byte b = plane b, bit i = genome
        // i); its transpose has genome i's low 8 count bits in byte i (three masked-swap steps instead of 8 x 9 single-bit
        // extractions), the planes from 8 on add their bits one by one
        {
            u64 t = make_u64(x[0] | (x[1] << 8) | (x[2] << 16) | (x[3] << 24), x[4] | (x[5] << 8) | (x[6] << 16) | (x[7] << 24));
            u64 y2 = (t ^ (t >> 7)) & 0x00AA00AA00AA00AAull;
            t ^= y2 ^ (y2 << 7);
            y2 = (t ^ (t >> 14)) & 0x0000CCCC0000CCCCull;
            t ^= y2 ^ (y2 << 14);
            y2 = (t ^ (t >> 28)) & 0x00000000F0F0F0F0ull;
            t ^= y2 ^ (y2 << 28);
            const u32 tl = (u32)t, th = (u32)(t >> 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32 a = (tl >> (8 * i)) & 0xFFu, b2 = (th >> (8 * i)) & 0xFFu;
#pragma unroll
                for (int b = 8; b < NP + 3; ++b) {
                    a += ((x[b] >> i) & 1u) << b;
                    b2 += ((x[b] >> (i + 4)) & 1u) << b;
                }
                acc[i] += a;
                acc[i + 4] += b2;
            }
        }
    };
    // SPARSE groups (few of the dictionary's rows hold a bit for any genome of the group: a species the sample does not belong
    // to still shares a handful of its k-mers with every other species' genomes -- at k = 16 two random 2.8 Mb genomes share
    // ~3 600 -- so none of its rank groups is without bits, but 98 % of a segment's rows are zero for them): the pairs whose
    // row holds anything (rowany, one bit per (group, query row), written by the transpose) are compacted into the wave's LDS
    // queue first, and only those are counted -- usually none or a single partial step per segment instead of four full ones.
    const u32 nq = *n_q;
    const bool sparse = rowany != nullptr && grp_any[grp] * 4u < nq;
    if (sparse) {
        const u64* ra_g = rowany + (size_t)grp * (nq_rows >> 6);
        u32* queue = squeue[wv];
        const u64 lt = lanemask_lt();
        u32 qcount = 0;
        for (u32 p0 = pa; p0 < pz; p0 += 64u) {
            const u32 p = p0 + lane;
            const bool ok = p < pz;
            const u32 q = ok ? pair_q[p] : 0u;
            const bool nz = ok && ((ra_g[q >> 6] >> (q & 63u)) & 1ull);
            const u64 m = __ballot(nz);
            if (m) {
                if (nz) queue[qcount + (u32)__popcll(m & lt)] = q;
                qcount = __builtin_amdgcn_readfirstlane(qcount + (u32)__popcll(m));
                if (qcount > kSparseQueue - 64u) {
                    wave_sync();
                    count_block(qcount, [&](u32 i) -> u32 { return queue[i]; });
                    qcount = 0;
                    wave_sync();
                }
            }
        }
        if (qcount) {
            wave_sync();
            count_block(qcount, [&](u32 i) -> u32 { return queue[i]; });
        }
    } else {
        for (u32 b0 = pa; b0 < pz; b0 += kBlockPairs) {
            const u32 cnt = min(pz, b0 + kBlockPairs) - b0;
            const char* at = pq_base + (size_t)(b0 * 4u);
            count_block(cnt, [&](u32 i) -> u32 { return *reinterpret_cast<const u32*>(at + (size_t)(i * 4u)); });
        }
    }
    const u32 gw = grp * kRankWords + j;
    if (!CHUNK && on && gw < n_gw) {  // words past n_gw hold no genomes
        u32* out = inc + (size_t)seg * n_pad + gw * 64u + sub * 8u;
        *reinterpret_cast<uint4*>(out) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<uint4*>(out + 4) = make_uint4(acc[4], acc[5], acc[6], acc[7]);
    }
    if (qsum) {
        u32* mine = &red[wv][j * 64u + sub * 8u];
        *reinterpret_cast<uint4*>(mine) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<uint4*>(mine + 4) = make_uint4(acc[4], acc[5], acc[6], acc[7]);
        __syncthreads();
#pragma unroll
        for (u32 h = 0; h < 2u; ++h) {
            const u32 t = threadIdx.x + 256u * h;
            const u32 v = red[0][t] + red[1][t] + red[2][t] + red[3][t];
            if (grp * kRankWords * 64u + t < n_pad && v) {
                if (CHUNK) qsum[(size_t)seg4 * n_pad + grp * kRankWords * 64u + t] = v;  // (the workgroup IS the chunk)
                else atomicAdd(&qsum[(size_t)(seg4 >> 2) * n_pad + grp * kRankWords * 64u + t], v);
            }
        }
    }
}
#undef SKX_CSA

// Chunk-level pruning.  gmax[c][h] = largest value any genome of HALF rank group h (256 genomes) has as chunk c begins
// (row n_chunks: as the pass ends), lead_val[c] = the value of the k-th ranked genome as chunk c begins.  No genome of
// the half group can be among the first k at any read of chunk c unless the best value at the END of the chunk reaches
// lead_val[c] (sums never decrease; the k-th best never decreases either).
// Species: several reference collections share the matrix, each padded to whole rank groups; every species has its own
// ranking, so the leaders / bounds are per (chunk, species): lead_val[c * n_sp + sp], leader[(c * n_sp + sp) * k + j].
// sp.g0[sp] = first (padded) genome index, sp.n[sp] = real genomes, sp.of_grp[grp] = species of a rank group.
// (struct Species: skx_kernels.hpp)
// Segment start values, relative to the table at the start of the pass (32 bits: a pass gains at most its pair
// count), in three levels so no thread walks a long chain and nothing is read twice:
//   seg_sum_kernel     : csum_raw[c][g] = sum of inc over the 16 segments of chunk c (atomic adds of its workgroups' sums)
//   chunk_prefix_kernel: csum[.][g] = exclusive prefix of csum_raw over the chunks;  cum_out[g] = cum_in[g] + total
//   seg_prefix_kernel  : rel[seg][g] = csum[chunk of seg][g] + inc of the chunk's earlier segments
// so that the running sum of genome g before the first read of segment seg is cum_in[g] + rel[seg][g].
// chunk_prefix: one block per 256 genomes (half a rank group; n_pad is a multiple of 512).  The chunk sums are fetched 32
// at a time into registers and the prefixes written elsewhere (in place every load waited for the store before it: 96
// memory round trips in a row, 200+ us next to the other streams' kernels, measured).  gmax != NULL: the best value of
// the block's genomes at every chunk boundary (gmax[c][half], row n_chunks = as the pass ends) comes out of the same
// registers (chunk_group_live's bound) instead of a separate 30 000-block launch.
__global__ __launch_bounds__(256) void chunk_prefix_kernel(const u32* __restrict__ csum_raw, u32* __restrict__ csum, u32 n_chunks,
                                                           u32 n_pad, const u64* __restrict__ cum_in, u64* __restrict__ cum_out,
                                                           u64* __restrict__ gmax, u32 n_half) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    constexpr u32 kR = 32;  // chunk sums requested per round trip
    __shared__ u64 part[4][kR + 1];
    const u32 g = blockIdx.x * 256u + threadIdx.x, lane = lane_id(), wv = threadIdx.x >> 6;
    const u64 base = cum_in[g];  // (padding genomes: 0, and their sums stay 0)
    u32 run = 0;
    for (u32 c0 = 0; c0 <= n_chunks; c0 += kR) {
        const u32 nc = min(kR, n_chunks - c0);  // (the last round may have none: only the closing row)
        u32 t[kR];
#pragma unroll
        for (u32 i = 0; i < kR; ++i) t[i] = i < nc ? csum_raw[(size_t)(c0 + i) * n_pad + g] : 0u;  // (not in place: no load waits for a store)
        const u32 rows = min(kR + 1u, n_chunks + 1u - c0);  // boundaries handled this round (row n_chunks closes the pass)
#pragma unroll
        for (u32 i = 0; i < kR; ++i) {
            if (i < rows) {
                if (i < nc) csum[(size_t)(c0 + i) * n_pad + g] = run;
                if (gmax) {
                    u64 v = base + run;
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) v = max(v, shfl_xor64(v, d));
                    if (lane == 0) part[wv][i] = v;
                }
                run += t[i];
            }
        }
        if (gmax) {
            const u32 rows16 = min(kR, rows);
            __syncthreads();
            if (threadIdx.x < rows16)
                gmax[(size_t)(c0 + threadIdx.x) * n_half + blockIdx.x] =
                    max(max(part[0][threadIdx.x], part[1][threadIdx.x]), max(part[2][threadIdx.x], part[3][threadIdx.x]));
            __syncthreads();
        }
        if (nc < kR) break;
    }
    cum_out[g] = base + run;
}
// lead_seg[seg][species] = the smallest value, as segment seg begins, among the k genomes that ranked first when its chunk
// began: the bound the pruned ranking kernels measure candidates against (they recompute it; seg_prefix_kernel uses it to decide
// per word whether ANY genome of it can be a candidate).  Computed by chunk_leader_merge_kernel, one wave per (chunk, species).
// live != NULL (top-1 ranking): live[seg][genome word] = some genome of the word ENDS segment seg at or above lead_val of
// the segment's chunk (the leader's value as the chunk began -- the leader only grows, so that is a lower bound of
// every bound the ranking uses inside the chunk).  Words that cannot are never looked at by rank_seg_top1_kernel, and
// their start values are not even stored: once a sample has a clear best match that is nearly all of them, and the
// 2 x 247 MB per batch that the ranking read only to find no candidate (measured: 119 000 of 121 000 waves) stay unread.
__global__ __launch_bounds__(256) void seg_prefix_kernel(const u32* __restrict__ inc, const u32* __restrict__ csum,
                                                         u32 n_seg, u32 n_pad, u32* __restrict__ rel,
                                                         const u64* __restrict__ gmax, const u64* __restrict__ lead_val,
                                                         u32 n_half, Species sp, const u32* __restrict__ grp_any,
                                                         const u64* __restrict__ cum_in, unsigned char* __restrict__ live,
                                                         const u64* __restrict__ lead_seg, u32* __restrict__ live_ctr) {
    __builtin_amdgcn_s_setprio(2);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    const u32 g = blockIdx.x * 256u + threadIdx.x, c = blockIdx.y;
    if (g >= n_pad) return;
    const bool grp_dead = !grp_any[blockIdx.x >> 1];
    const bool chunk_live = !grp_dead && (!gmax || chunk_group_live(gmax, lead_val, n_half, c, blockIdx.x >> 1, sp));
    // (a sample of every 16th (chunk, half group): which share of them can hold a candidate -- the host picks the counting
    // scheme of later batches by it, launch_chunk_sum)
    if (live_ctr && threadIdx.x == 0 && ((blockIdx.x | blockIdx.y) & 3u) == 0u) {
        atomicAdd(&live_ctr[1], 1u);
        if (chunk_live) atomicAdd(&live_ctr[0], 1u);
    }
    if (grp_dead) return;  // (a group without any bit starts every segment at the pass-start table)
    // a (chunk, rank group) without any possible candidate is never looked at by the ranking: skip its start values
    // (both halves of a live group are written: the ranking reads the whole group)
    if (!chunk_live) return;
    const u32 s0 = c * 16u, s1 = min(n_seg, s0 + 16u);
    u32 run = csum[(size_t)c * n_pad + g];
    if (!live) {
#pragma unroll 16
        for (u32 sgi = s0; sgi < s1; ++sgi) {
            rel[(size_t)sgi * n_pad + g] = run;
            run += inc[(size_t)sgi * n_pad + g];
        }
        return;
    }
    const u32 spi = sp.of_grp[blockIdx.x >> 1];
    const u64 base = cum_in[g];
    const u32 gw = g >> 6, n_gw = n_pad >> 6;
    u32 t[16];
#pragma unroll
    for (u32 i = 0; i < 16u; ++i) t[i] = s0 + i < s1 ? inc[(size_t)(s0 + i) * n_pad + g] : 0u;
#pragma unroll
    for (u32 i = 0; i < 16u; ++i) {
        if (s0 + i < s1) {
            const u32 start = run;
            run += t[i];
            // (the segment's own bound, chunk_leader_merge_kernel: exactly the candidate test of the ranking kernels, per word)
            const u64 lv = lead_seg[(size_t)(s0 + i) * sp.n_sp + spi];
            const bool any = __ballot(base + run >= lv) != 0ull;  // (padding genomes: base 0, never gain)
            if (any) rel[(size_t)(s0 + i) * n_pad + g] = start;
            if (lane_id() == 0) live[(size_t)(s0 + i) * n_gw + gw] = any ? 1 : 0;
        }
    }
}

// (sum desc, index asc) ordering: a ranks before b
__device__ __forceinline__ bool ranks_before(u64 sa, u32 ia, u64 sb, u32 ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// wave-wide best (sum, idx) among lanes with `ok`; every lane returns the winner.
// idx == 0xFFFFFFFF marks "none".
__device__ __forceinline__ void wave_best(u64& sum, u32& idx, bool ok) {
    if (!ok) { sum = 0; idx = 0xFFFFFFFFu; }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const u64 os = shfl_xor64(sum, m);
        const u32 oi = (u32)__shfl_xor((int)idx, m, 64);
        const bool take = (oi != 0xFFFFFFFFu) && (idx == 0xFFFFFFFFu || ranks_before(os, oi, sum, idx));
        if (take) { sum = os; idx = oi; }
    }
}

// leader[c * top_k + j] = the genome ranked j-th (sum desc, index asc) as chunk c (16 segments = 1024 reads) begins,
// lead_val[c] = the value of the top_k-th, from the pass-start table and the prefixed chunk sums.  Two steps so the
// first one consists of many small blocks -- a 1024-thread block needs 16 free wave slots on one CU and starves for
// hundreds of microseconds next to the front half's sketch kernel, 4-wave blocks slip in: (a) every one of
// kLeaderParts blocks per chunk finds the first top_k of its slice of the genomes, (b) one wave per chunk merges the
// kLeaderParts x top_k candidates.
constexpr u32 kLeaderParts = 32;
constexpr u32 kTopkFastMax = 16;  // (= kTopkFast, defined with the top-k kernel below)
__global__ __launch_bounds__(256) void chunk_leader_part_kernel(const u64* __restrict__ cum_in, const u32* __restrict__ csum,
                                                                 u32 n_pad, Species sp, u32 top_k,
                                                                 u64* __restrict__ part_sum, u32* __restrict__ part_idx) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    __shared__ u64 ssum[4];
    __shared__ u32 sidx[4];
    __shared__ u64 wsum;
    __shared__ u32 widx;
    // blockIdx.x = chunk * n_sp + species: the arrays below are indexed by it directly
    const u32 cs = blockIdx.x, c = cs / sp.n_sp, spi = cs % sp.n_sp, part = blockIdx.y, tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const u32 sp_lo = sp.g0[spi], sp_hi = sp_lo + sp.n[spi];
    const u32 per = (sp.n[spi] + kLeaderParts - 1) / kLeaderParts, g_lo = min(sp_hi, sp_lo + part * per), g_hi = min(sp_hi, g_lo + per);
    u64 ps = 0; u32 pi = 0; bool first = true;
    for (u32 j = 0; j < top_k; ++j) {
        u64 bs = 0; u32 bi = 0xFFFFFFFFu;
        for (u32 g = g_lo + tid; g < g_hi; g += 256u) {
            const u64 v = cum_in[g] + csum[(size_t)c * n_pad + g];
            if (!first && !ranks_before(ps, pi, v, g)) continue;  // already taken in an earlier round
            if (bi == 0xFFFFFFFFu || v > bs) { bs = v; bi = g; }  // ascending g: ties keep the lower index
        }
        wave_best(bs, bi, bi != 0xFFFFFFFFu);
        if (lane == 0) { ssum[wv] = bs; sidx[wv] = bi; }
        __syncthreads();
        if (wv == 0) {
            u64 s2 = lane < 4 ? ssum[lane] : 0; u32 i2 = lane < 4 ? sidx[lane] : 0xFFFFFFFFu;
            wave_best(s2, i2, i2 != 0xFFFFFFFFu);
            if (lane == 0) {
                wsum = s2; widx = i2;
                part_sum[((size_t)cs * kLeaderParts + part) * top_k + j] = s2;
                part_idx[((size_t)cs * kLeaderParts + part) * top_k + j] = i2;  // 0xFFFFFFFF: the slice is exhausted
            }
        }
        __syncthreads();
        ps = wsum; pi = widx; first = widx == 0xFFFFFFFFu ? first : false;
        if (widx == 0xFFFFFFFFu) {  // nothing left in this slice: the remaining rounds are empty too
            for (u32 jj = j + 1; jj < top_k && tid == 0; ++jj) part_idx[((size_t)cs * kLeaderParts + part) * top_k + jj] = 0xFFFFFFFFu;
            break;
        }
        __syncthreads();
    }
}
// one wave per chunk
// lead_seg != NULL: the wave goes on with the per-segment bounds of its (chunk, species) -- one launch less on the ranking chain
__global__ __launch_bounds__(64) void chunk_leader_merge_kernel(const u64* __restrict__ part_sum, const u32* __restrict__ part_idx,
                                                                u32 top_k, u32* __restrict__ leader, u64* __restrict__ lead_val,
                                                                const u32* __restrict__ inc, const u32* __restrict__ csum, u32 n_seg,
                                                                u32 n_pad, const u64* __restrict__ cum_in, Species sp,
                                                                const u32* __restrict__ grp_any, u64* __restrict__ lead_seg,
                                                                u32 mode /* 0: both halves, 1: leaders only, 2: lead_seg only */) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    __shared__ u32 s_leader[kTopkFastMax];
    const u32 c = blockIdx.x, lane = lane_id(), n_cand = kLeaderParts * top_k;
    u64 ps = 0; u32 pi = 0; bool first = true;
    if (mode == 2u) {  // (the leaders were found by an earlier launch: the per-segment increments did not exist yet)
        if (lane < top_k && lane < kTopkFastMax) s_leader[lane] = leader[c * top_k + lane];
    } else
    for (u32 j = 0; j < top_k; ++j) {
        u64 bs = 0; u32 bi = 0xFFFFFFFFu;
        for (u32 x = lane; x < n_cand; x += 64u) {
            const u64 v = part_sum[(size_t)c * n_cand + x];
            const u32 g = part_idx[(size_t)c * n_cand + x];
            if (g == 0xFFFFFFFFu) continue;
            if (!first && !ranks_before(ps, pi, v, g)) continue;
            if (bi == 0xFFFFFFFFu || ranks_before(v, g, bs, bi)) { bs = v; bi = g; }
        }
        wave_best(bs, bi, bi != 0xFFFFFFFFu);
        if (lane == 0) { leader[c * top_k + j] = bi; if (j < kTopkFastMax) s_leader[j] = bi; }
        ps = bs; pi = bi; first = false;
    }
    if (mode != 2u && lane == 0) lead_val[c] = ps;  // value of the top_k-th ranked genome as the chunk begins (top_k <= n_genomes)
    if (lead_seg == nullptr || mode == 1u) return;
    // lead_seg[seg][species] (chunk_leader_merge_kernel): lane = segment of the chunk
    wave_sync();
    const u32 ch = c / sp.n_sp, spi = c % sp.n_sp;
    const u32 seg = ch * 16u + lane;
    if (lane >= 16u || seg >= n_seg) return;
    u64 lead = ~0ull;
    for (u32 j = 0; j < top_k; ++j) {
        const u32 gl = s_leader[j];
        u64 v = cum_in[gl];
        if (grp_any[gl / (kRankWords * 64u)]) {
            u32 run = csum[(size_t)ch * n_pad + gl];
            for (u32 s2 = ch * 16u; s2 < seg; ++s2) run += inc[(size_t)s2 * n_pad + gl];
            v += run;
        }
        lead = min(lead, v);
    }
    lead_seg[(size_t)seg * sp.n_sp + spi] = lead;
}

// rank_seg (generic, top_k > 16): walk a segment's reads in order from its start values; after every read emit this genome
// group's top_k candidates cand_sum/cand_idx[(r * n_gw + gw) * top_k + j].  One wave per (gw, seg).
__global__ __launch_bounds__(256) void rank_seg_kernel(const u32* __restrict__ pair_q, const u32* __restrict__ pair_r,
                                                       const u32* __restrict__ poff, u32 p_base, u32 r_begin,
                                                       u32 n_reads, u32 seg_len, const u64* __restrict__ mq,
                                                       u32 n_gw, u32 n_pad, Species sp,
                                                       const u64* __restrict__ cum_in, const u32* __restrict__ rel,
                                                       u32 top_k, u64* __restrict__ cand_sum,
                                                       u32* __restrict__ cand_idx, u32 nq_rows, const u32* __restrict__ grp_any) {
    const u32 wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = lane_id();
    const u32 n_seg = (n_reads + seg_len - 1) / seg_len;
    const u32 gw = wave % n_gw, seg = wave / n_gw;
    if (seg >= n_seg) return;
    const u32 ra = seg * seg_len, rz = min(n_reads, ra + seg_len);
    const bool dead = !grp_any[gw / kRankWords];  // no bit in the whole group: the table does not move, nothing to replay
    const u32 pa = poff[r_begin + ra] - p_base, pz = dead ? pa : poff[r_begin + rz] - p_base;
    const u32 g = gw * 64u + lane;
    const u32 spi = sp.of_grp[gw / kRankWords];
    const bool real = g < sp.g0[spi] + sp.n[spi];  // (padding genomes of the species' last groups never rank)
    u64 state = cum_in[g] + (dead ? 0u : rel[(size_t)seg * n_pad + g]);
    u32 cur = ra;  // next read to emit

    auto emit = [&](u32 r) {
        u64 ps = 0; u32 pi = 0; bool first = true;
        for (u32 j = 0; j < top_k; ++j) {
            // candidates strictly after the previous winner in rank order
            const bool ok = real && (first || ranks_before(ps, pi, state, g));
            u64 bs = state; u32 bi = g;
            wave_best(bs, bi, ok);
            if (lane == 0) {
                const size_t o = ((size_t)r * n_gw + gw) * top_k + j;
                cand_sum[o] = bs; cand_idx[o] = bi;
            }
            if (bi == 0xFFFFFFFFu) {  // exhausted: fill the rest
                for (u32 jj = j + 1; jj < top_k; ++jj)
                    if (lane == 0) { const size_t o = ((size_t)r * n_gw + gw) * top_k + jj; cand_sum[o] = 0; cand_idx[o] = 0xFFFFFFFFu; }
                break;
            }
            ps = bs; pi = bi; first = false;
        }
    };

    for (u32 p0 = pa; p0 < pz; p0 += 64u) {
        const u32 n = min(64u, pz - p0);
        u64 myword = 0; u32 myread = 0;
        if (lane < n) {
            myword = mq[mq_index(gw, pair_q[p0 + lane], nq_rows)];
            myread = pair_r[p0 + lane];
        }
        for (u32 j = 0; j < n; ++j) {
            const u64 word = readlane64(myword, (int)j);
            const u32 rd = __builtin_amdgcn_readlane(myread, (int)j);
            while (cur < rd) { emit(cur); ++cur; }
            state += (word >> lane) & 1ull;
        }
    }
    while (cur < rz) { emit(cur); ++cur; }
}

// topk_merge: per (read, species), merge the species' candidates into the final top_k.  One wave per (read, species).
// Candidates of read r: cand[(r * n_units + u) * top_k + j], u = rank group (pruned path, per_grp = 1) or genome word
// (generic path, per_grp = kRankWords); a species owns the units of its rank groups.  Output rows hold genome indices
// LOCAL to the species: out[((out_r0 + r) * n_sp + sp) * top_k + j].
// Every unit's list is in rank order with its "none" entries at the end (rank_seg_topk_kernel / rank_seg_kernel write them so), so
// this is a k-way merge: a unit takes part with the HEAD of its list only -- next[u], one byte of LDS per unit, owned by the lane
// that walks the unit -- and a round costs one entry per unit, not top_k.  (Until round 6 every round walked all top_k entries of
// every unit: 16 x 1 264 entries x 98 304 reads for a full top-16 ranking at C2, 4-9 ms; profiles/r06_topk.txt.)
__global__ __launch_bounds__(256) void topk_merge_kernel(const u64* __restrict__ cand_sum,
                                                         const u32* __restrict__ cand_idx, u32 n_reads, u32 n_units,
                                                         u32 per_grp, u32 top_k, u32* __restrict__ out_idx,
                                                         u64* __restrict__ out_sum, u32 out_r0, Species sp,
                                                         const unsigned char* __restrict__ has /* [seg][n_units] or NULL */,
                                                         u32 next_stride) {
    extern __shared__ unsigned char s_next[];  // [4 waves][next_stride]
    const u32 w = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = lane_id();
    const u32 r = w / sp.n_sp, spi = w % sp.n_sp;
    if (r >= n_reads) return;
    const u32 g_lo = sp.g0[spi], grp0 = g_lo / (kRankWords * 64u), grp1 = (g_lo + sp.n[spi] + kRankWords * 64u - 1u) / (kRankWords * 64u);
    const u32 n_u = (grp1 - grp0) * per_grp;
    const u64* cs = cand_sum + ((size_t)r * n_units + (size_t)grp0 * per_grp) * top_k;
    const u32* ci = cand_idx + ((size_t)r * n_units + (size_t)grp0 * per_grp) * top_k;
    if (n_u * top_k <= 64u) {
        // Few enough entries for one per lane (the candidates' compact problems: two rank groups per species): an entry's place in the row
        // is the number of entries that rank before it -- one load, no rounds.
        const u32 u = lane / top_k;
        bool ok = lane < n_u * top_k && !(has && !has[(size_t)(r >> 6) * n_units + grp0 + u]);
        u64 s_ = 0; u32 i_ = 0xFFFFFFFFu;
        if (ok) { s_ = cs[lane]; i_ = ci[lane]; }
        ok = ok && i_ != 0xFFFFFFFFu;
        const u64 okm = __ballot(ok);
        u32 before = 0;
        for (u64 m = okm; m; m &= m - 1ull) {
            const int e = __builtin_ctzll(m);
            const u64 se = readlane64(s_, e); const u32 ie = __builtin_amdgcn_readlane(i_, e);
            before += ranks_before(se, ie, s_, i_) ? 1u : 0u;
        }
        const size_t o = ((size_t)(out_r0 + r) * sp.n_sp + spi) * top_k;
        if (ok && before < top_k) { out_idx[o + before] = i_ - g_lo; out_sum[o + before] = s_; }
        const u32 n_ok = (u32)__popcll(okm);
        if (lane >= n_ok && lane < top_k) { out_idx[o + lane] = 0xFFFFFFFFu - g_lo; out_sum[o + lane] = 0; }  // (nothing left: see below)
        return;
    }
    unsigned char* next = s_next + (threadIdx.x >> 6) * next_stride;
    // (a unit the pruned ranking reported nothing for -- its entries were not even written -- starts exhausted; has: unit = rank group)
    for (u32 u = lane; u < n_u; u += 64u) next[u] = (unsigned char)((has && !has[(size_t)(r >> 6) * n_units + grp0 + u]) ? top_k : 0u);
    for (u32 j = 0; j < top_k; ++j) {
        u64 bs = 0; u32 bi = 0xFFFFFFFFu, bu = 0;
        for (u32 u = lane; u < n_u; u += 64u) {  // (lane l owns units l, l + 64, ...: no other lane reads or writes their next[])
            const u32 nx = next[u];
            if (nx >= top_k) continue;
            const u64 s_ = cs[(size_t)u * top_k + nx]; const u32 i_ = ci[(size_t)u * top_k + nx];
            if (i_ == 0xFFFFFFFFu) { next[u] = (unsigned char)top_k; continue; }
            if (bi == 0xFFFFFFFFu || ranks_before(s_, i_, bs, bi)) { bs = s_; bi = i_; bu = u; }
        }
        const u32 mine = bi;
        wave_best(bs, bi, bi != 0xFFFFFFFFu);
        if (bi != 0xFFFFFFFFu && mine == bi) next[bu] += 1;  // (genome indices are unique: one lane)
        if (lane == 0) {  // (nothing left: index "none" - g_lo and sum 0, as the rows of a species with fewer genomes than top_k always were)
            out_idx[((size_t)(out_r0 + r) * sp.n_sp + spi) * top_k + j] = bi - g_lo;
            out_sum[((size_t)(out_r0 + r) * sp.n_sp + spi) * top_k + j] = bs;
        }
    }
}

#ifdef SKX_EXPERIMENTS
// instrumentation of rank_seg_top1_kernel (experiments build only): [0..4] waves by exit (chunk dead, no live word, no candidate,
// few candidates, replay), [5] candidates of replaying waves, [6] pairs they replayed, [8..15] replaying waves by candidates
// (1, 2-4, 5-16, 17-64, 65-512), [16 + c] replaying waves of chunk c (c < 112)
__device__ unsigned long long g_rank_dbg[128];
__device__ int g_rank_dbg_on;  // (off unless skx_debug_rank_counters switched it on: the atomics triple the kernel's time)
#define SKX_DBG_ADD(i, v) do { if (g_rank_dbg_on) atomicAdd(&g_rank_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define SKX_DBG_ADD(i, v)
#endif
// ---- top-1 fast path -------------------------------------------------------------------
// wave-wide max of a u32 (DPP within rows of 16 lanes, then 4 readlanes); every lane returns it.
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));  // row_mirror
    const u32 a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const u32 c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

// rank_seg for top_k == 1.  One wave per (rank group of 512 genomes, segment); lane l holds genomes
// (grp*8 + j)*64 + l, j = 0..7.  Within one segment a genome gains at most G = (#pairs of the segment), so only
// genomes within G of the group's best starting sum can ever lead it, and for those
//   key = ((G - (best - start) + gained + 1) << 9) | ((7 - j) << 6) | (63 - l)
// fits 32 bits and orders exactly like (sum desc, genome index asc).  The segment's <= 64 results are kept one
// per lane and stored once: best_sum/best_idx[grp * n_reads + r].
__global__ __launch_bounds__(256) void rank_seg_top1_kernel(const u32* __restrict__ pair_q,
                                                            const u32* __restrict__ pair_r,
                                                            const u32* __restrict__ poff, u32 p_base, u32 r_begin,
                                                            u32 n_reads, u32 seg_len /* == 64 */,
                                                            const u64* __restrict__ mq, u32 n_gw, u32 n_pad,
                                                            Species sp, const u64* __restrict__ cum_in,
                                                            const u32* __restrict__ rel,
                                                            u64* __restrict__ best_sum, u32* __restrict__ best_idx,
                                                            u32 nq_rows, const u32* __restrict__ inc,
                                                            const u32* __restrict__ leader, const u64* __restrict__ gmax,
                                                            const u64* __restrict__ lead_val, const u32* __restrict__ grp_any,
                                                            const unsigned char* __restrict__ live, unsigned char* __restrict__ has,
                                                            const u64* __restrict__ rowany, const u32* __restrict__ n_q) {
    __shared__ u32 sq_q[4][kSparseQueue], sq_r[4][kSparseQueue];  // a sparse group's pairs worth replaying (per wave)
    __builtin_amdgcn_s_setprio(SKX_RANK1_PRIO);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    constexpr int NW = kRankWords, SH = 6 + 3;
    static_assert(kRankWords == 8, "key layout assumes 8 words per lane");
    const u32 wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256u + threadIdx.x) >> 6), lane = lane_id();
    const u32 n_seg = (n_reads + seg_len - 1) / seg_len, n_grp = (n_gw + NW - 1) / NW;
    // (groups interleaved over the workgroups, NOT the XCD-aware order of seg_sum_kernel: after pruning the work sits
    // in the few groups that hold candidates, and walking one group per XCD would leave 7 XCDs idle -- measured)
    const u32 grp = wave % n_grp, seg = wave / n_grp;
    if (seg >= n_seg) return;
    const u64* mq_g = mq + (size_t)grp * nq_rows * NW;
    const u32 ra = seg * seg_len, rz = min(n_reads, ra + seg_len);
    // Everything that can end this wave at once is requested together, before anything else is loaded: the chunk-level
    // bound, the group's "any bit" flag and the eight `live` bytes of seg_prefix_kernel.  (119 000 of the 121 000 waves of
    // a C2 batch leave here; with the leader lookups -- two dependent round trips -- in front of the test they held
    // their wave slots three times as long.)
    const u32 gwl = grp * NW + (lane & (NW - 1));
    const bool live_byte = live && lane < (u32)NW && gwl < n_gw && live[(size_t)seg * n_gw + gwl] != 0;
    // a group without any bit in the pass (e.g. a species the sample does not belong to): its sums do not move -- no
    // start values / increments were written for it, no pairs are replayed, one key serves the whole segment
    const bool dead = !grp_any[grp];
    if (!chunk_group_live(gmax, lead_val, n_pad / 256u, seg >> 4, grp, sp)) {  // (its start values were not even written)
        if (lane == 0) has[(size_t)seg * n_grp + grp] = 0;  // nothing to report: the merge skips this (segment, group)
        if (lane == 0) SKX_DBG_ADD(0, 1);
        return;
    }
    // words none of whose genomes can reach even the chunk's leader bound by the end of the segment (seg_prefix_kernel):
    // no start values were stored for them, nothing of theirs is loaded -- usually that is the whole group
    u32 livew = 0xFFu;
    if (!dead && live) {
        livew = (u32)__ballot(live_byte) & 0xFFu;
        if (livew == 0u) {
            if (lane == 0) has[(size_t)seg * n_grp + grp] = 0;
            if (lane == 0) SKX_DBG_ADD(1, 1);
            return;
        }
    }
    const u32 spi = sp.of_grp[grp], sp_end = sp.g0[spi] + sp.n[spi];
    const u32 pa = poff[r_begin + ra] - p_base;
    u32 pz = dead ? pa : poff[r_begin + rz] - p_base;
    const u32 g0 = grp * NW * 64u + lane;
    // Pruning (exact).  Sums never decrease and a genome ends the segment at start + inc, so with ANY lower bound
    // `lead` of the leading sum over the segment, only genomes with start + inc >= lead can lead at one of its
    // reads.  The bound: the segment start value of the genome that led when the segment's chunk of 1024 reads began
    // (one scalar load; for a sample with a stable best match that IS the leading sum) or this group's own best
    // start, whichever is larger
    // (the latter also keeps every start of the group <= lead, which the 32-bit keys need).  Once a sample has a
    // clear best match a handful of genomes are left: a group without any reports "none" straight away, and inside
    // a live group words without any are neither loaded nor counted.  Non-candidates (and padding) get value 0.
    const u32 gl = leader[(seg >> 4) * sp.n_sp + spi];  // (top_k == 1: one leader per chunk and species)
    u64 lead = cum_in[gl] + (grp_any[gl / (NW * 64u)] ? rel[(size_t)seg * n_pad + gl] : 0u);
    const u32 gain = pz - pa;
    u64 st0[NW];
    u32 ic[NW];
    bool real[NW];
    u64 grp_best = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const u32 g = g0 + (u32)j * 64u;
        real[j] = g < sp_end && ((livew >> j) & 1u);  // (padding genomes of the species' last group never rank)
        st0[j] = real[j] ? cum_in[g] + (dead ? 0u : rel[(size_t)seg * n_pad + g]) : 0;
        ic[j] = (real[j] && !dead) ? inc[(size_t)seg * n_pad + g] : 0;
        grp_best = max(grp_best, st0[j]);
    }
    // Nobody in this group gains anything in this segment (a species that shares a k-mer with the sample now and then:
    // most of its (group, segment)s): the sums stand still, one key serves every read -- nothing to gather or replay.
    {
        u32 any_inc = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) any_inc |= ic[j];
        if (__ballot(any_inc != 0u) == 0ull) pz = pa;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) grp_best = max(grp_best, shfl_xor64(grp_best, d));
    lead = max(lead, grp_best);
    u32 val[NW];
    u32 wmask = 0;  // wave-uniform: words holding at least one candidate
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const bool cand = real[j] && st0[j] + ic[j] >= lead;   // (then lead - st0 <= inc <= gain)
        val[j] = cand ? gain - (u32)(lead - st0[j]) + 1u : 0u;
        if (__ballot(cand)) wmask |= 1u << j;
    }
    if (wmask == 0) {
        if (lane == 0) has[(size_t)seg * n_grp + grp] = 0;  // nothing to report: the merge skips this (segment, group)
        if (lane == 0) SKX_DBG_ADD(2, 1);
        return;
    }
    const u64 base = lead - gain - 1u;  // winner sum = base + (key >> SH)   (mod 2^64)
    u32 tiec[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) tiec[j] = ((u32)(NW - 1 - j) << 6) | (63u - lane);
    u32 cur = ra, res_key = 0;
    // Few candidates in the whole group -- one is the usual case once a sample has a clear best match (the leader's group,
    // every segment): a candidate's value after read r is its start plus the hits among the pairs of reads <= r.  Per
    // candidate: one row WORD per pair (not the group's 64 bytes), one ballot per 64 pairs, a prefix popcount per read; the
    // read's best is the maximum over the candidates' keys.  No transposes, no read-by-read replay (~70 instructions per
    // candidate instead of ~3000 per wave, and a wave that ends in microseconds instead of being the kernel's tail).
    constexpr u32 kFewCands = SKX_FEW_CANDS;
    u32 n_cands = 0;
    u64 cbal[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        cbal[j] = __ballot(val[j] != 0u);
        n_cands += (u32)__popcll(cbal[j]);
    }
    const bool single = n_cands <= kFewCands;  // (n_cands >= 1 here: wmask != 0)
#ifdef SKX_EXPERIMENTS
    if (lane == 0) {
        SKX_DBG_ADD(single ? 3 : 4, 1);
        if (!single) {
            SKX_DBG_ADD(5, n_cands);
            SKX_DBG_ADD(6, pz - pa);
            SKX_DBG_ADD(8 + (n_cands <= 1 ? 0 : n_cands <= 4 ? 1 : n_cands <= 16 ? 2 : n_cands <= 64 ? 3 : 4), 1);
            if ((seg >> 4) < 112u) SKX_DBG_ADD(16 + (seg >> 4), 1);
        }
    }
#endif
    if (single) {
        const u32 pe = lane < rz - ra ? poff[r_begin + ra + lane + 1u] - p_base - pa : 0u;  // pairs of reads <= this lane's
        if (n_cands == 1u) {  // one candidate: one row WORD per pair is all that is needed
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                if (cbal[j] == 0ull) continue;
                const u32 lc = (u32)__builtin_ctzll(cbal[j]);
                const u32 v0 = (u32)__builtin_amdgcn_readlane((int)val[j], (int)lc);
                u32 cnt = 0;
                for (u32 p0 = pa; p0 < pz; p0 += 64u) {
                    const u32 p = p0 + lane;
                    const bool v = p < pz;
                    const u64 word = v ? mq_g[(size_t)pair_q[p] * NW + (u32)j] : 0ull;
                    const u64 hm = __ballot(v && ((word >> lc) & 1ull));
                    const u32 off = p0 - pa, nlow = pe > off ? min(64u, pe - off) : 0u;
                    const u64 lm = nlow >= 64u ? ~0ull : ((1ull << nlow) - 1ull);
                    cnt += (u32)__popcll(hm & lm);
                }
                res_key = max(res_key, ((v0 + cnt) << SH) | ((u32)(NW - 1 - j) << 6) | (63u - lc));
            }
        } else {
            // a handful (first batches of a sample: instrumented, half of the replaying waves held 2-8 candidates): every pair's
            // 64-byte row is fetched ONCE (as the replay does), each candidate takes one ballot per 64 pairs out of it, and the
            // per-read counts of the candidates sit in the wave's LDS (slot x lane, private to the lane: no synchronisation)
            static_assert(kFewCands <= 8u && kSparseQueue >= 256u, "eight count slots of 64 lanes in the wave's two queue arrays");
            u32* const cnt_lo = sq_q[threadIdx.x >> 6] + lane;  // slots 0..3 at [slot * 64]
            u32* const cnt_hi = sq_r[threadIdx.x >> 6] + lane;  // slots 4..7
#pragma unroll
            for (u32 sl = 0; sl < 4u; ++sl) { cnt_lo[sl * 64u] = 0; cnt_hi[sl * 64u] = 0; }
            for (u32 p0 = pa; p0 < pz; p0 += 64u) {
                const u32 p = p0 + lane;
                const bool v = p < pz;
                const MaskVec rowv = gather_vec(mq_g, v ? pair_q[p] : 0u, v);
                const u32 off = p0 - pa, nlow = pe > off ? min(64u, pe - off) : 0u;
                const u64 lm = nlow >= 64u ? ~0ull : ((1ull << nlow) - 1ull);
                u32 slot = 0;  // (wave-uniform)
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    u64 bal = cbal[j];
                    while (bal) {
                        const u32 lc = (u32)__builtin_ctzll(bal);
                        bal &= bal - 1ull;
                        const u64 hm = __ballot((rowv.w[j] >> lc) & 1ull);  // (rows past the end are zero)
                        u32* const c = (slot < 4u ? cnt_lo : cnt_hi) + (slot & 3u) * 64u;
                        *c += (u32)__popcll(hm & lm);
                        ++slot;
                    }
                }
            }
            u32 slot = 0;
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                u64 bal = cbal[j];
                while (bal) {
                    const u32 lc = (u32)__builtin_ctzll(bal);
                    bal &= bal - 1ull;
                    const u32 v0 = (u32)__builtin_amdgcn_readlane((int)val[j], (int)lc);
                    const u32 cnt = ((slot < 4u ? cnt_lo : cnt_hi) + (slot & 3u) * 64u)[0];
                    res_key = max(res_key, ((v0 + cnt) << SH) | ((u32)(NW - 1 - j) << 6) | (63u - lc));
                    ++slot;
                }
            }
        }
    }

    auto emit_upto = [&](u32 r_stop) {  // reads [cur, r_stop) all see the current state
        if (cur >= r_stop) return;
        u32 k = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) k = max(k, (val[j] << SH) | tiec[j]);
        const u32 key = wave_max_u32(k);
        if (lane >= cur - ra && lane < r_stop - ra) res_key = key;
        cur = r_stop;
    };

    auto replay = [&](const u64 (&x)[NW], u32 rv, u32 n) {  // x: transposed chunk, rv: read index of pair `lane`
        for (u32 j = 0; j < n;) {
            const u32 rd = __builtin_amdgcn_readlane(rv, (int)j);
            emit_upto(rd);                                   // reads before rd see the state without rd's pairs
            const u64 m = __ballot(lane < n && rv == rd);    // rd's pairs inside this chunk (contiguous from j)
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if (wmask >> w & 1u) val[w] += __popcll(x[w] & m);  // (value 0 stays 0 only where nobody is a candidate)
            j += __popcll(m);
        }
    };
    if (!single) {
        // The segment's pairs go through a queue in the wave's LDS, 64 at a time, and are replayed from there in chunks of 64
        // (the rows of the next chunk are requested while the current one is replayed).  SPARSE groups (seg_sum_kernel: nearly
        // all of the segment's rows are zero for this group's genomes) queue only the pairs whose row holds a bit (rowany) --
        // the reads in between see an unchanged state, and emit_upto is lazy anyway.
        const bool sparse = rowany != nullptr && grp_any[grp] * 4u < *n_q;
        const u64* ra_g = rowany + (size_t)grp * (nq_rows >> 6);
        u32* qq = sq_q[threadIdx.x >> 6];
        u32* qr = sq_r[threadIdx.x >> 6];
        const u64 lt = lanemask_lt();
        u32 qcount = 0;
        auto flush = [&]() {
            wave_sync();
            MaskVec nxt = gather_vec(mq_g, lane < qcount ? qq[lane] : 0u, lane < qcount);
            u32 rnxt = lane < qcount ? qr[lane] : 0u;
            for (u32 i0 = 0; i0 < qcount; i0 += 64u) {
                const u32 n = min(64u, qcount - i0);
                const MaskVec cur_m = nxt;
                const u32 rv = rnxt;
                const u32 in = i0 + 64u + lane;
                nxt = gather_vec(mq_g, in < qcount ? qq[in] : 0u, in < qcount);
                rnxt = in < qcount ? qr[in] : 0u;
                u64 x[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j)  // live words only; a word none of whose rows holds a bit needs no transpose
                    x[j] = ((wmask >> j & 1u) && __ballot(cur_m.w[j] != 0ull)) ? transpose64(cur_m.w[j], lane) : 0;
                replay(x, rv, n);
            }
            qcount = 0;
            wave_sync();
        };
        for (u32 p0 = pa;; p0 += 64u) {  // (one flush site: the replay code exists once)
            const bool more = p0 < pz;
            if (more) {
                const u32 p = p0 + lane;
                const bool ok = p < pz;
                const u32 q = ok ? pair_q[p] : 0u;
                const bool nz = ok && (!sparse || ((ra_g[q >> 6] >> (q & 63u)) & 1ull));
                const u64 m = __ballot(nz);
                if (nz) {
                    const u32 at = qcount + (u32)__popcll(m & lt);
                    qq[at] = q;
                    qr[at] = pair_r[p];
                }
                qcount = __builtin_amdgcn_readfirstlane(qcount + (u32)__popcll(m));
            }
            if (qcount && (!more || qcount > kSparseQueue - 64u)) flush();
            if (!more) break;
        }
        emit_upto(rz);
    }
    if (lane == 0) has[(size_t)seg * n_grp + grp] = 1;
    if (lane < rz - ra) {
        const size_t o = (size_t)grp * n_reads + ra + lane;
        // A non-candidate sitting in a live word counts up from 0 and can out-number this group's candidates; what
        // is reported for it is below `lead` (value <= gain), hence below the true leader's entry (>= lead, from the
        // leader's own group, where its value >= gain + 1 beats every non-candidate): the merge never picks it.
        const u32 wl = 63u - (res_key & 63u), wj = (u32)(NW - 1) - ((res_key >> 6) & (u32)(NW - 1));
        best_sum[o] = base + (u64)(res_key >> SH);
        best_idx[o] = (grp * NW + wj) * 64u + wl;
    }
}

// wave-wide max of a u64 given as (hi, lo); every lane returns it
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
    const u32 hi = (u32)(v >> 32);
    const u32 mh = wave_max_u32(hi);
    const u32 ml = wave_max_u32(hi == mh ? (u32)v : 0u);
    return ((u64)mh << 32) | ml;
}

// rank_seg for 2 <= top_k <= kTopkFast, pruned like the top-1 kernel.  One wave per (rank group of 512 genomes,
// segment).  The k-th best sum never decreases either, and it is at least the smallest segment start value among the
// k genomes that ranked first when the segment's chunk began (`lead`); so only genomes with start + inc >= lead can be
// in the top k at any read of the segment.  Those candidates are replayed read by read with 64-bit keys
// (sum << 9 | tie) -- the leader may be far ahead of the k-th, so the keys cannot be made relative as for top-1 --
// and every read takes k rounds of "largest key below the previous winner" over the group's candidates.
// Non-candidates never take part.  Output: cand_sum / cand_idx[(r * n_grp + grp) * top_k + j], idx 0xFFFFFFFF = none.
constexpr u32 kTopkFast = 16;
constexpr u32 kFewCands = 8;  // a group with at most this many candidates counts them one by one instead of replaying the segment
static_assert(kFewCands <= kTopkFast, "res[] holds the few-candidates path's rows");
static_assert(kTopkFast == kTopkFastMax, "chunk_leader_merge_kernel's LDS copy of the leaders");
__global__ __launch_bounds__(256) void rank_seg_topk_kernel(const u32* __restrict__ pair_q, const u32* __restrict__ pair_r,
                                                            const u32* __restrict__ poff, u32 p_base, u32 r_begin,
                                                            u32 n_reads, const u64* __restrict__ mq, u32 n_gw, u32 n_pad,
                                                            Species sp, const u64* __restrict__ cum_in,
                                                            const u32* __restrict__ rel, u32 top_k,
                                                            u64* __restrict__ cand_sum, u32* __restrict__ cand_idx,
                                                            u32 nq_rows, const u32* __restrict__ inc,
                                                            const u32* __restrict__ leader, const u64* __restrict__ gmax,
                                                            const u64* __restrict__ lead_val, const u32* __restrict__ grp_any,
                                                            const unsigned char* __restrict__ live, unsigned char* __restrict__ has) {
    constexpr int NW = kRankWords, SH = 6 + 3;
    const u32 wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256u + threadIdx.x) >> 6), lane = lane_id();
    const u32 n_seg = (n_reads + 63u) / 64u, n_grp = (n_gw + NW - 1) / NW;
    const u32 grp = wave % n_grp, seg = wave / n_grp;
    if (seg >= n_seg) return;
    const u64* mq_g = mq + (size_t)grp * nq_rows * NW;
    const u32 ra = seg * 64u, rz = min(n_reads, ra + 64u);
    if (!chunk_group_live(gmax, lead_val, n_pad / 256u, seg >> 4, grp, sp)) {  // (its start values were not even written)
        if (lane == 0) has[(size_t)seg * n_grp + grp] = 0;  // nothing to report: the merge skips this (segment, group)
        return;
    }
    const bool dead = !grp_any[grp];  // no bit in the whole group: nothing was written for it, nothing to replay
    const u32 pa = poff[r_begin + ra] - p_base, pz = dead ? pa : poff[r_begin + rz] - p_base;
    const u32 g0 = grp * NW * 64u + lane;
    // words none of whose genomes can reach the chunk's k-th value by the end of the segment (seg_prefix_kernel's flags:
    // no start values were stored for them): usually the whole group, which then reports "none" without loading anything
    u32 livew = 0xFFu;
    if (!dead && live) {
        const u32 gwl = grp * NW + (lane & (NW - 1));
        livew = (u32)__ballot(lane < (u32)NW && gwl < n_gw && live[(size_t)seg * n_gw + gwl] != 0) & 0xFFu;
        if (livew == 0u) {
            if (lane == 0) has[(size_t)seg * n_grp + grp] = 0;
            return;
        }
    }
    // lower bound of the k-th best sum (of this group's species) over the segment
    const u32 spi = sp.of_grp[grp], sp_end = sp.g0[spi] + sp.n[spi];
    u64 lead = ~0ull;
    for (u32 j = 0; j < top_k; ++j) {
        const u32 gl = leader[((seg >> 4) * sp.n_sp + spi) * top_k + j];
        lead = min(lead, cum_in[gl] + (grp_any[gl / (NW * 64u)] ? rel[(size_t)seg * n_pad + gl] : 0u));
    }
    u64 sum[NW];
    bool cand[NW];
    u32 wmask = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const u32 g = g0 + (u32)j * 64u;
        const bool real = g < sp_end && ((livew >> j) & 1u);
        sum[j] = real ? cum_in[g] + (dead ? 0u : rel[(size_t)seg * n_pad + g]) : 0;
        const u32 ic = (real && !dead) ? inc[(size_t)seg * n_pad + g] : 0;
        cand[j] = real && sum[j] + ic >= lead;
        if (__ballot(cand[j])) wmask |= 1u << j;
    }
    u64 res[kTopkFast];  // lane l: the k winners (keys) of read ra + l
#pragma unroll
    for (u32 j = 0; j < kTopkFast; ++j) res[j] = 0;
    // One candidate in the whole group (instrumented at C2, k = 5: every replaying wave): its value after read r is its
    // start plus the hits among the pairs of reads <= r -- one row WORD per pair (not the group's 64 bytes), one ballot per
    // 64 pairs, a prefix popcount per read.  No transposes, no read-by-read replay: ~70 instructions instead of ~3000.
    u32 n_cands = 0;
    u32 wc = 0;
    u64 cbal = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const u64 b_ = __ballot(cand[w]);
        n_cands += (u32)__popcll(b_);
        if (b_) { wc = (u32)w; cbal = b_; }
    }
    if (n_cands == 1u) {
        const u32 lc = (u32)__builtin_ctzll(cbal);
        u64 sv = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w)
            if ((u32)w == wc) sv = sum[w];
        const u64 start = make_u64((u32)__builtin_amdgcn_readlane((int)(u32)sv, (int)lc), (u32)__builtin_amdgcn_readlane((int)(u32)(sv >> 32), (int)lc));
        // pairs of the segment up to and including this lane's read
        const u32 pe = lane < rz - ra ? poff[r_begin + ra + lane + 1u] - p_base - pa : 0u;
        u32 cnt = 0;
        for (u32 p0 = pa; p0 < pz; p0 += 64u) {
            const u32 p = p0 + lane;
            const bool v = p < pz;
            const u64 word = v ? mq_g[(size_t)pair_q[p] * NW + wc] : 0ull;
            const u64 hm = __ballot(v && ((word >> lc) & 1ull));
            const u32 off = p0 - pa, nlow = pe > off ? min(64u, pe - off) : 0u;
            const u64 lm = nlow >= 64u ? ~0ull : ((1ull << nlow) - 1ull);
            cnt += (u32)__popcll(hm & lm);
        }
        if (lane < rz - ra) res[0] = ((start + cnt + 1ull) << SH) | ((u64)(NW - 1 - wc) << 6) | (u64)(63u - lc);
    } else if (n_cands >= 2u && n_cands <= kFewCands) {
        // A handful of candidates in the group (a full ranking of a clone-tree sample: the leading lineage's strains are spread over
        // all the rank groups, two or three to each): the same count as above per candidate -- every lane ends up with the values of
        // all of them after ITS read -- and the read's order is a selection among <= kFewCands keys in registers.  No read-by-read
        // replay with its wave-wide maxima: ~4 k instructions per wave instead of ~28 k (profiles/r06_topk.txt).
        u64 bal[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) bal[w] = __ballot(cand[w]);
        u32 cw[kFewCands], cl[kFewCands];
        u64 cst[kFewCands];
#pragma unroll
        for (u32 c = 0; c < kFewCands; ++c) {  // candidate c in (word, lane) order; everything here is wave-uniform
            u32 fw = (u32)NW;
#pragma unroll
            for (int w = NW - 1; w >= 0; --w) if (bal[w]) fw = (u32)w;
            u64 b_ = 0, sv = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) if ((u32)w == fw) { b_ = bal[w]; sv = sum[w]; }
            const u32 l = b_ ? (u32)__builtin_ctzll(b_) : 0u;
#pragma unroll
            for (int w = 0; w < NW; ++w) if ((u32)w == fw) bal[w] = b_ & (b_ - 1ull);
            cw[c] = fw < (u32)NW ? fw : 0u;
            cl[c] = l;
            cst[c] = readlane64(sv, (int)l);
        }
        const u32 pe = lane < rz - ra ? poff[r_begin + ra + lane + 1u] - p_base - pa : 0u;  // the segment's pairs up to and including this lane's read
        u32 cnt[kFewCands];
#pragma unroll
        for (u32 c = 0; c < kFewCands; ++c) cnt[c] = 0;
        for (u32 p0 = pa; p0 < pz; p0 += 64u) {
            const u32 p = p0 + lane;
            const bool v = p < pz;
            const u64* row = mq_g + (size_t)(v ? pair_q[p] : 0u) * NW;
            const u32 off = p0 - pa, nlow = pe > off ? min(64u, pe - off) : 0u;
            const u64 lm = nlow >= 64u ? ~0ull : ((1ull << nlow) - 1ull);
#pragma unroll
            for (u32 c = 0; c < kFewCands; ++c) {
                if (c < n_cands) {
                    const u64 word = v ? row[cw[c]] : 0ull;
                    const u64 hm = __ballot(v && ((word >> cl[c]) & 1ull));
                    cnt[c] += (u32)__popcll(hm & lm);
                }
            }
        }
        u64 key[kFewCands];
#pragma unroll
        for (u32 c = 0; c < kFewCands; ++c)
            key[c] = c < n_cands ? (((cst[c] + cnt[c] + 1ull) << SH) | ((u64)((u32)NW - 1u - cw[c]) << 6) | (u64)(63u - cl[c])) : 0ull;
        u64 prev = ~0ull;
#pragma unroll
        for (u32 j = 0; j < kFewCands; ++j) {
            if (j < top_k) {
                u64 best = 0;
#pragma unroll
                for (u32 c = 0; c < kFewCands; ++c) if (key[c] < prev) best = max(best, key[c]);
                res[j] = best;  // (keys are distinct; 0 = fewer than j + 1 candidates: nothing is below 0)
                prev = best;
            }
        }
    } else if (n_cands > kFewCands && n_cands <= 64u) {
        // At most one candidate per lane (a leader has emerged: the compact problems of a clone-tree sample carry 20-60): the candidates
        // move to lanes 0 .. n_cands-1 in (word, lane) order, the replay keeps ONE sum per lane, and a read's order is each candidate's
        // count of the candidates ahead of it (n_cands independent compares) instead of top_k dependent wave-wide maxima; the rows are
        // stored as they are made.  ~5 n_cands + 30 instructions per read instead of ~45 top_k (profiles/r06_topk.txt).
        __shared__ u32 s_cid[4][64];
        const u32 wv = threadIdx.x >> 6;
        {
            u32 base = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const u64 b_ = __ballot(cand[w]);
                if (cand[w]) s_cid[wv][base + (u32)__popcll(b_ & lanemask_lt())] = ((u32)w << 6) | lane;
                base += (u32)__popcll(b_);
            }
        }
        wave_sync();
        const bool isc = lane < n_cands;
        const u32 cid = isc ? s_cid[wv][lane] : 0u;
        const u32 wc_ = cid >> 6, lc_ = cid & 63u;
        auto pick = [&](const u64 (&v)[NW]) -> u64 {  // v[wc_] of lane lc_
            u64 out = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if ((wmask >> w) & 1u) {
                    const u32 lo = (u32)__shfl((int)(u32)v[w], (int)lc_, 64), hi = (u32)__shfl((int)(u32)(v[w] >> 32), (int)lc_, 64);
                    if (wc_ == (u32)w) out = make_u64(lo, hi);
                }
            }
            return out;
        };
        u64 sumc = pick(sum);
        const u32 gidx = (grp * NW + wc_) * 64u + lc_;
        const u64 tie = ((u64)((u32)NW - 1u - wc_) << 6) | (u64)(63u - lc_);
        u32 cur = ra;
        auto emit_upto = [&](u32 r_stop) {  // reads [cur, r_stop) all see the current state
            if (cur >= r_stop) return;
            const u64 key = isc ? (((sumc + 1ull) << SH) | tie) : 0ull;
            u32 ahead = 0;
            for (u32 e = 0; e < n_cands; ++e) ahead += readlane64(key, (int)e) > key ? 1u : 0u;
            for (u32 r = cur; r < r_stop; ++r) {
                const size_t o = ((size_t)r * n_grp + grp) * top_k;
                if (isc) { if (ahead < top_k) { cand_sum[o + ahead] = sumc; cand_idx[o + ahead] = gidx; } }
                else if (lane < top_k) { cand_sum[o + lane] = 0; cand_idx[o + lane] = 0xFFFFFFFFu; }  // (fewer than top_k candidates)
            }
            cur = r_stop;
        };
        MaskVec nxt = gather_vec(mq_g, (pa + lane < pz) ? pair_q[pa + lane] : 0u, pa + lane < pz);
        u32 rnxt = (pa + lane < pz) ? pair_r[pa + lane] : 0u;
        for (u32 p0 = pa; p0 < pz; p0 += 64u) {
            const u32 n = min(64u, pz - p0);
            const MaskVec cur_m = nxt;
            const u32 rv = rnxt;
            const u32 pn = p0 + 64u + lane;
            nxt = gather_vec(mq_g, pn < pz ? pair_q[pn] : 0u, pn < pz);
            rnxt = pn < pz ? pair_r[pn] : 0u;
            u64 x[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) x[j] = (wmask >> j & 1u) ? transpose64(cur_m.w[j], lane) : 0;  // live words only
            const u64 xc = pick(x);  // this lane's candidate across the chunk's pairs
            for (u32 j = 0; j < n;) {
                const u32 rd = __builtin_amdgcn_readlane(rv, (int)j);
                emit_upto(rd);                                   // reads before rd see the state without rd's pairs
                const u64 m = __ballot(lane < n && rv == rd);    // rd's pairs inside this chunk (contiguous from j)
                sumc += (u64)__popcll(xc & m);
                j += __popcll(m);
            }
        }
        emit_upto(rz);
        if (lane == 0) has[(size_t)seg * n_grp + grp] = 1;
        return;
    } else if (wmask != 0) {
        u32 cur = ra;
        auto emit_upto = [&](u32 r_stop) {  // reads [cur, r_stop) all see the current state
            if (cur >= r_stop) return;
            const bool mine = lane >= cur - ra && lane < r_stop - ra;
            // keys of this lane's candidates, once per emit (words without any candidate -- usually seven of the eight -- are
            // skipped: wave-uniform); then one wave-wide maximum per rank, stopping at the first rank nobody fills
            u64 keys[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w)
                keys[w] = ((wmask >> w) & 1u) && cand[w] ? (((sum[w] + 1ull) << SH) | ((u64)(NW - 1 - w) << 6) | (63u - lane)) : 0ull;
            u64 prev = ~0ull;
            bool more = true;  // wave-uniform
#pragma unroll
            for (u32 j = 0; j < kTopkFast; ++j) {
                if (j < top_k) {
                    u64 best = 0;
                    if (more) {
                        u64 k = 0;
#pragma unroll
                        for (int w = 0; w < NW; ++w)
                            if ((wmask >> w) & 1u) { if (keys[w] < prev) k = max(k, keys[w]); }
                        best = wave_max_u64(k);  // 0: fewer than j+1 candidates in this group
                        more = best != 0ull;
                    }
                    if (mine) res[j] = best;
                    prev = best;                  // (0 ends it: nothing is below 0)
                }
            }
            cur = r_stop;
        };
        MaskVec nxt = gather_vec(mq_g, (pa + lane < pz) ? pair_q[pa + lane] : 0u, pa + lane < pz);
        u32 rnxt = (pa + lane < pz) ? pair_r[pa + lane] : 0u;
        for (u32 p0 = pa; p0 < pz; p0 += 64u) {
            const u32 n = min(64u, pz - p0);
            const MaskVec cur_m = nxt;
            const u32 rv = rnxt;
            const u32 pn = p0 + 64u + lane;
            nxt = gather_vec(mq_g, pn < pz ? pair_q[pn] : 0u, pn < pz);
            rnxt = pn < pz ? pair_r[pn] : 0u;
            u64 x[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) x[j] = (wmask >> j & 1u) ? transpose64(cur_m.w[j], lane) : 0;  // live words only
            for (u32 j = 0; j < n;) {
                const u32 rd = __builtin_amdgcn_readlane(rv, (int)j);
                emit_upto(rd);                                   // reads before rd see the state without rd's pairs
                const u64 m = __ballot(lane < n && rv == rd);    // rd's pairs inside this chunk (contiguous from j)
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    if (wmask >> w & 1u) sum[w] += __popcll(x[w] & m);
                j += __popcll(m);
            }
        }
        emit_upto(rz);
    }
    if (lane == 0) has[(size_t)seg * n_grp + grp] = wmask != 0 ? 1 : 0;
    if (wmask == 0) return;
    if (lane < rz - ra) {
        const size_t o = ((size_t)(ra + lane) * n_grp + grp) * top_k;
#pragma unroll
        for (u32 j = 0; j < kTopkFast; ++j) {
            if (j < top_k) {
                const u64 key = res[j];
                const u32 wl = 63u - (u32)(key & 63u), wj = (u32)(NW - 1) - (u32)((key >> 6) & (u64)(NW - 1));
                cand_sum[o + j] = key ? (key >> SH) - 1ull : 0ull;
                cand_idx[o + j] = key ? (grp * NW + wj) * 64u + wl : 0xFFFFFFFFu;
            }
        }
    }
}

// merge for top_k == 1: one lane per (read, species), walking the species' rank groups in index order.
// out[(out_r0 + r) * n_sp + sp] = (genome index local to the species, sum)
__global__ __launch_bounds__(256) void top1_merge_kernel(const u64* __restrict__ best_sum,
                                                         const u32* __restrict__ best_idx, u32 n_reads,
                                                         u32* __restrict__ out_idx, u64* __restrict__ out_sum,
                                                         u32 out_r0, Species sp, const unsigned char* __restrict__ has, u32 n_grp) {
    __builtin_amdgcn_s_setprio(3);  // short / latency-bound link of a chain: do not queue behind the VALU-bound kernels beside it
    const u32 t = blockIdx.x * 256u + threadIdx.x;
    const u32 spi = t / n_reads, r = t % n_reads;  // (reads innermost: coalesced candidate loads)
    if (spi >= sp.n_sp) return;
    const u32 g_lo = sp.g0[spi], grp0 = g_lo / (kRankWords * 64u), grp1 = (g_lo + sp.n[spi] + kRankWords * 64u - 1u) / (kRankWords * 64u);
    u64 bs = 0; u32 bi = 0xFFFFFFFFu;
    // the groups the ranking reported anything for (this read's segment): their flags first, 64 independent byte loads at a
    // time, then only those groups' entries -- usually one or two of 79 (a test per group in front of its loads made every
    // load wait for the one before it)
    const unsigned char* hs = has + (size_t)(r >> 6) * n_grp;
    for (u32 base = grp0; base < grp1; base += 64u) {
        u64 mask = 0;
#pragma unroll 8
        for (u32 i = 0; i < 64u; ++i)
            if (base + i < grp1) mask |= (u64)(hs[base + i] != 0) << i;
        while (mask) {
            const u32 grp = base + (u32)__builtin_ctzll(mask);
            mask &= mask - 1ull;
            const u64 s_ = best_sum[(size_t)grp * n_reads + r];
            const u32 i_ = best_idx[(size_t)grp * n_reads + r];
            // groups come in ascending index order: a later group wins only with a strictly larger sum
            if (i_ != 0xFFFFFFFFu && (bi == 0xFFFFFFFFu || s_ > bs)) { bs = s_; bi = i_; }
        }
    }
    out_idx[(size_t)(out_r0 + r) * sp.n_sp + spi] = bi - g_lo;
    out_sum[(size_t)(out_r0 + r) * sp.n_sp + spi] = bs;
}

// rank the table itself: one block per species, top_k rounds.  (skx_stream_rank; also the all-reduced table)
// out[sp * top_k + j] = (genome index local to the species, sum)
__global__ __launch_bounds__(1024) void rank_table_kernel(const u64* __restrict__ cum, Species sp, u32 top_k,
                                                          u32* __restrict__ out_idx, u64* __restrict__ out_sum) {
    __shared__ u64 ssum[16];
    __shared__ u32 sidx[16];
    __shared__ u64 wsum;
    __shared__ u32 widx;
    const u32 tid = threadIdx.x, lane = lane_id(), wv = tid >> 6, spi = blockIdx.x;
    const u32 g_lo = sp.g0[spi], g_hi = g_lo + sp.n[spi];
    u64 ps = 0; u32 pi = 0; bool first = true;
    for (u32 j = 0; j < top_k; ++j) {
        u64 bs = 0; u32 bi = 0xFFFFFFFFu;
        for (u32 g = g_lo + tid; g < g_hi; g += 1024u) {
            const u64 s_ = cum[g];
            if (!first && !ranks_before(ps, pi, s_, g)) continue;
            if (bi == 0xFFFFFFFFu || ranks_before(s_, g, bs, bi)) { bs = s_; bi = g; }
        }
        wave_best(bs, bi, bi != 0xFFFFFFFFu);
        if (lane == 0) { ssum[wv] = bs; sidx[wv] = bi; }
        __syncthreads();
        if (wv == 0) {
            u64 s2 = lane < 16 ? ssum[lane] : 0; u32 i2 = lane < 16 ? sidx[lane] : 0xFFFFFFFFu;
            wave_best(s2, i2, i2 != 0xFFFFFFFFu);
            if (lane == 0) { wsum = s2; widx = i2; out_idx[spi * top_k + j] = i2 - g_lo; out_sum[spi * top_k + j] = s2; }
        }
        __syncthreads();
        ps = wsum; pi = widx; first = false;
        __syncthreads();
    }
}

// parity/debug: shared[r][g] = sum over read r's pairs of bit(Mq[q][g]).  One block per read.
// g runs over the n_real real genomes (species concatenated, no padding); real2pad maps it into the padded order.
__global__ __launch_bounds__(256) void shared_debug_kernel(const u32* __restrict__ pair_q, const u32* __restrict__ poff,
                                                           u32 p_base, u32 r_begin, const u64* __restrict__ mq,
                                                           u32 n_real, const u32* __restrict__ real2pad,
                                                           u32* __restrict__ shared, u32 out_r0, u32 nq_rows) {
    const u32 r = blockIdx.x;
    const u32 pa = poff[r_begin + r] - p_base, pz = poff[r_begin + r + 1] - p_base;
    for (u32 g = threadIdx.x; g < n_real; g += 256u) {
        const u32 gp = real2pad[g];
        u32 acc = 0;
        for (u32 p = pa; p < pz; ++p)
            acc += (u32)((mq[mq_index(gp >> 6, pair_q[p], nq_rows)] >> (gp & 63u)) & 1ull);
        shared[(size_t)(out_r0 + r) * n_real + g] = acc;
    }
}

// the table as the caller sees it (real genomes, species concatenated) <-> the padded table of the kernels
__global__ void add_table_kernel(u64* __restrict__ cum, const u64* __restrict__ add, u32 n_real, const u32* __restrict__ real2pad) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_real) cum[real2pad[g]] += add[g];
}
__global__ void gather_table_kernel(const u64* __restrict__ cum, u64* __restrict__ out, u32 n_real, const u32* __restrict__ real2pad) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_real) out[g] = cum[real2pad[g]];
}

// =====================================================================================
// the table without the ranking, and the genomes a batch's ranking has to look at (round 5)
// =====================================================================================
// Rounds 1-4 took the running table out of the ranking chain: per-segment increments, chunk sums, prefixes -- every pair's 64-byte
// row of Mq fetched once per rank group (232 k pairs x 79 groups per C2 batch) -- and the chain of batch i + 1 waited for the table
// the chain of batch i left.  But the table only needs how OFTEN each distinct hash occurs in a batch: a batch of ~100 k reads
// holds ~10 k distinct dense hashes, so   gain_b[g] = sum over rows q of cnt_b[q] * M[q][g]   reads the bit matrix once (50 MB)
// for all batches of a pass, and   table_{b+1} = table_b + gain_b   is known for every batch of the pass before any ranking starts.
// With the tables known, so is the set of genomes a batch's ranking has to look at at all: sums never decrease, so a genome
// whose value at the END of batch b is below the k-th best value at its START cannot be among the first k at any read of it
// (at least k genomes stay at or above that bound throughout).  The CANDIDATES of batch b -- table_{b+1}[g] >= k-th best of
// table_b -- are listed in reference order; when a species has at most kCandCap of them the per-read ranking runs on a compact
// problem (those genomes only: the same kernels on a bit matrix of two rank groups instead of 79), else on everything as before.
// Reads of a real sample come from one strain: after a few hundred thousand reads the candidates are its lineage.  The bench's
// near-tie (reads from the common ancestor of all 40 000 genomes) never gets there and keeps the full ranking -- both are exact.
constexpr u32 kCandNone = 0xFFFFFFFFu;
constexpr u32 kPatWords = 128;  // genome words of a compact problem the long-list path of cand_sparse_kernel holds in LDS (8 species)

// cnt[b][row] = occurrences of the row among batch b's pairs (zero on entry)
__global__ __launch_bounds__(256) void pass_hist_kernel(const u32* __restrict__ pair_q, PassBatches pb, u32* __restrict__ cnt, u32 row_stride) {
    __builtin_amdgcn_s_setprio(3);
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= pb.p_off[pb.n]) return;
    u32 b = 0;
#pragma unroll
    for (u32 i = 1; i < kPassBatchesMax; ++i) b += (i < pb.n && p >= pb.p_off[i]) ? 1u : 0u;
    atomicAdd(&cnt[(size_t)b * row_stride + pair_q[p]], 1u);
}
// gain[b][g] += sum over the dense rows of cnt[b][row] * M[row][g].  Lane = genome, the counts are wave-uniform (scalar loads).
// grid: (n_pad / 256, word chunks); M is read before the transpose re-zeroes it.
constexpr u32 kGainWords = 8;  // query words per block
constexpr u32 kGainSparseStride = SKX_GAIN_SPARSE_STRIDE;  // u32 words between two genomes' entries of the rare rows' gain array
// segw (or NULL; static dictionaries of several species): segw[2 sp], [2 sp + 1] = the words that hold species sp's rows -- a block
// of 256 genomes belongs to one species (grp_sp) and reads no other species' words (all zero for its genomes)
template <u32 NB>
__global__ __launch_bounds__(256) void gain_dense_kernel(const u64* __restrict__ m_bits, const u64* __restrict__ m_int, u32 n_pad,
                                                         const u32* __restrict__ n_d, const u32* __restrict__ cnt, u32 row_stride,
                                                         u32* __restrict__ gain, const u32* __restrict__ segw, const u32* __restrict__ grp_sp) {
    __builtin_amdgcn_s_setprio(2);
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    const u32 nd = n_d[0], n_words = (nd + 63u) >> 6;
    u32 w0 = blockIdx.y * kGainWords;
    if (w0 >= n_words) return;
    u32 w1 = min(n_words, w0 + kGainWords);
    if (segw) {
        const u32 sp = grp_sp[blockIdx.x / (kRankWords * 64u / 256u)];
        w0 = max(w0, segw[2u * sp]); w1 = min(w1, segw[2u * sp + 1u]);
        if (w0 >= w1) return;
    }
    u32 acc[NB];
#pragma unroll
    for (u32 b = 0; b < NB; ++b) acc[b] = 0;
    for (u32 w = w0; w < w1; ++w) {
        u64 m = m_bits[(size_t)w * n_pad + g];
        if (m_int) m |= m_int[(size_t)w * n_pad + g];  // (the split scan variant keeps interior words in a second array)
        if (__ballot(m != 0ull) == 0ull) continue;
        const u32 lo = (u32)m, hi = (u32)(m >> 32);
        const u32 rows = min(64u, nd - w * 64u);
        const u32* c = cnt + (size_t)w * 64u;
        for (u32 j = 0; j < rows; ++j) {
            const u32 bit = j < 32u ? __builtin_amdgcn_ubfe(lo, j, 1u) : __builtin_amdgcn_ubfe(hi, j - 32u, 1u);
#pragma unroll
            for (u32 b = 0; b < NB; ++b) acc[b] += __umul24(bit, c[(size_t)b * row_stride + j]);
        }
    }
#pragma unroll
    for (u32 b = 0; b < NB; ++b)
        if (acc[b]) atomicAdd(&gain[(size_t)b * n_pad + g], acc[b]);
}
// ... and the rows behind the dense ones (rare hashes): cnt[b][row] to every genome on the hash's list: one walk over the lists, one
// atomic per posting and batch the row occurs in.  (Device-scope atomics are executed at the memory side, ~2.3 G/s scattered: 8.7 M
// postings per C2 pass of the SNP workload = 3.7 ms, most of them the ~200 genomes of SOME lineage whose lineage-level hash a
// sequencing error hit.  Tried: workgroups that own a range of 2048 genomes and add in LDS -- every range walks all the lists:
// 4.9 ms.  What would remove it: lists shared between hashes stored once -- all lineage-level hashes of a lineage have the SAME
// list --, counts summed per list first.  DESIGN.md section 9.)
__global__ __launch_bounds__(256) void gain_sparse_kernel(const u32* __restrict__ sslot, const u32* __restrict__ n_d, RareIndex ri,
                                                          const u32* __restrict__ cnt, u32 row_stride, u32 n_b, u32 n_pad,
                                                          u32* __restrict__ gain, LongRows lr, PatRows pr) {
    __builtin_amdgcn_s_setprio(2);
    const u32 nd64 = n_d[2], ns = n_d[1], lane = lane_id();
    const u32 wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    for (u32 r0 = wave * 64u; r0 < ns; r0 += n_waves * 64u) {
        const u32 sr = r0 + lane;
        u32 off = 0, np = 0;
        u32 c[kPassBatchesMax];
        u32 any = 0;
#pragma unroll
        for (u32 b = 0; b < kPassBatchesMax; ++b) c[b] = 0;
        if (sr < ns) {
            const uint2 e = reinterpret_cast<const uint2*>(sslot)[sr];
            off = e.x; np = e.y;
            if (np) {
#pragma unroll
                for (u32 b = 0; b < kPassBatchesMax; ++b)
                    if (b < n_b) { c[b] = cnt[(size_t)b * row_stride + nd64 + sr]; any |= c[b]; }
            }
        }
        // rows whose list is a pattern + exceptions (prec): the pattern's count goes up by the row's, the exceptions take the difference
        // (gain_x: wrapping adds -- the pattern's part, added by the dense-rows kernel on PM, makes every entry a true count again);
        // listed per batch from the END of the bit rows' region for cand_pat_map_kernel
        const bool is_pat = pr.hist != nullptr && (np & (kLongFlag | kPatFlag)) == (kLongFlag | kPatFlag);
        if (__ballot(is_pat)) {
            u64 bal[kPassBatchesMax];
            u32 mine = 0;
#pragma unroll
            for (u32 b = 0; b < kPassBatchesMax; ++b) {
                bal[b] = __ballot(b < n_b && is_pat && c[b] != 0u);
                if (lane == b) mine = (u32)__popcll(bal[b]);
            }
            u32 base = 0;
            if (mine) base = atomicAdd(&pr.nprow[lane * kCtrStride], mine);
#pragma unroll
            for (u32 b = 0; b < kPassBatchesMax; ++b) {
                const u32 bb = (u32)__shfl((int)base, (int)b);
                if ((bal[b] >> lane) & 1ull)
                    lr.lrow[(size_t)b * lr.lrow_stride + (lr.lrow_stride - 1u - (bb + (u32)__popcll(bal[b] & lanemask_lt())))] = make_uint2(off, sr);
            }
            if (is_pat && any) {
                const u32* rec = ri.prec + (size_t)off * kPatRec;
                const uint4 r0 = *reinterpret_cast<const uint4*>(rec);  // {pattern, exceptions, first two of them}
                const u32 pat = r0.x, n_exc = r0.y;
#pragma unroll
                for (u32 b = 0; b < kPassBatchesMax; ++b)
                    if (c[b]) atomicAdd(&pr.hist[(size_t)b * pr.hist_stride + pat], c[b]);
                for (u32 j = 0; j < n_exc; ++j) {
                    const u32 e = j == 0u ? r0.z : j == 1u ? r0.w : rec[2u + j];
                    const u32 g = e & 0x7FFFFFFFu;
                    const bool neg = (e >> 31) != 0u;
#pragma unroll
                    for (u32 b = 0; b < kPassBatchesMax; ++b)
                        if (c[b]) atomicAdd(&pr.gain_x[(size_t)b * n_pad + g], neg ? 0u - c[b] : c[b]);
                }
            }
        }
        const bool is_long = (np & kLongFlag) != 0u && !is_pat;
        if (is_pat) np = 0;
        if (__ballot(is_long)) {
            // rows with a bit row: listed per batch for gain_long_kernel / cand_long_kernel (one walk over the rare rows instead of two)
            if (lr.lrow) {
                // (lane b asks for batch b's slots: ONE round trip for the eight counters -- one after the other, all in one cache line,
                // they were 57 k serialised returning atomics per pass: the waves of this kernel spent 97 % of their cycles waiting for
                // them, 0.73 ms alone on the chip.  The counters sit kCtrStride words apart.)
                u64 bal[kPassBatchesMax];
                u32 mine = 0;
#pragma unroll
                for (u32 b = 0; b < kPassBatchesMax; ++b) {
                    bal[b] = __ballot(b < n_b && is_long && c[b] != 0u);
                    if (lane == b) mine = (u32)__popcll(bal[b]);
                }
                u32 base = 0;
                if (mine) base = atomicAdd(&lr.nlrow[lane * kCtrStride], mine);
#pragma unroll
                for (u32 b = 0; b < kPassBatchesMax; ++b) {
                    const u32 bb = (u32)__shfl((int)base, (int)b);
                    if ((bal[b] >> lane) & 1ull) {
                        lr.lrow[(size_t)b * lr.lrow_stride + bb + (u32)__popcll(bal[b] & lanemask_lt())] = make_uint2(off, sr);
                        if (lr.inb) atomicOr(&lr.inb[(size_t)b * lr.n_lw + (off >> 6)], 1ull << (off & 63u));  // (off: the row's bit row)
                    }
                }
            }
            if (is_long) np = 0;
        }
        if (!any) np = 0;
        if (np && np <= 8u) {
            for (u32 j = 0; j < np; ++j) {
                const u32 g = ri.post[off + j];
#pragma unroll
                for (u32 b = 0; b < kPassBatchesMax; ++b)
                    if (c[b]) atomicAdd(&gain[((size_t)b * n_pad + g) * kGainSparseStride], c[b]);
            }
        }
        u64 longs = __ballot(np > 8u);
        while (longs) {
            const u32 src = (u32)__builtin_ctzll(longs);
            longs &= longs - 1ull;
            const u32 o = __shfl(off, (int)src), n = __shfl(np, (int)src);
            u32 cc[kPassBatchesMax];
#pragma unroll
            for (u32 b = 0; b < kPassBatchesMax; ++b) cc[b] = __shfl(c[b], (int)src);
            for (u32 j = lane; j < n; j += 64u) {
                const u32 g = ri.post[o + j];
#pragma unroll
                for (u32 b = 0; b < kPassBatchesMax; ++b)
                    if (cc[b]) atomicAdd(&gain[((size_t)b * n_pad + g) * kGainSparseStride], cc[b]);
            }
        }
    }
}
// tab[0] = the table the pass starts from, tab[b + 1] = tab[b] + gain[b]
// the long-list rows of every batch, compacted: lrow[b][i] = {bit row, sparse row} for the rows that occur in batch b
__global__ __launch_bounds__(256) void long_rows_kernel(const u32* __restrict__ sslot, const u32* __restrict__ n_d, const u32* __restrict__ cnt,
                                                        u32 row_stride, u32 n_b, LongRows lr) {
    __builtin_amdgcn_s_setprio(3);
    const u32 nd64 = n_d[2], ns = n_d[1], lane = lane_id();
    for (u32 sr0 = (blockIdx.x * 256u + threadIdx.x) & ~63u; sr0 < ns; sr0 += gridDim.x * 256u) {
        const u32 sr = sr0 + lane;
        uint2 e = make_uint2(0u, 0u);
        if (sr < ns) e = reinterpret_cast<const uint2*>(sslot)[sr];
        const bool lng = (e.y & kLongFlag) != 0u;
        if (!__ballot(lng)) continue;
        for (u32 b = 0; b < n_b; ++b) {
            const bool in = lng && cnt[(size_t)b * row_stride + nd64 + sr] != 0u;
            const u64 bal = __ballot(in);
            if (!bal) continue;
            u32 base = 0;
            if (lane == 0u) base = atomicAdd(&lr.nlrow[b * kCtrStride], (u32)__popcll(bal));
            base = (u32)__shfl((int)base, 0);
            if (in) lr.lrow[(size_t)b * lr.lrow_stride + base + (u32)__popcll(bal & lanemask_lt())] = make_uint2(e.x, sr);
        }
    }
}
// gain_l[b][g] (zero on entry) += sum over batch b's long-list rows of cnt[b][row] x (bit g of the row).  A workgroup owns 64 genome words
// (4 096 genomes) of one batch and a share of its rows (gridDim.z): its four waves split them, every lane keeps the counts of its word's 64 genomes BIT-SLICED (plane p =
// bit p of the 64 counts: adding a row of weight c is a ripple-carry add of the row's word at the planes of c's set bits), the
// waves' planes are added in LDS and the counts extracted once: coalesced 512-byte reads of the rows, plain stores, no atomics.
constexpr u32 kGlPlanes = 23;  // counts below 2^23 (a batch has at most 2^22 pairs)
__device__ __forceinline__ void planes_add(u64 (&pl)[kGlPlanes], u64 m, u32 k0) {
    u64 carry = m;
#pragma unroll
    for (u32 p = 0; p < kGlPlanes; ++p) {
        if (p >= k0) {
            const u64 t = pl[p] & carry;
            pl[p] ^= carry;
            carry = t;
            if ((p & 3u) == 3u && __ballot(carry != 0ull) == 0ull) return;  // (the carry dies after a few planes)
        }
    }
}
__global__ __launch_bounds__(256) void gain_long_kernel(LongRows lr, RareIndex ri, const u32* __restrict__ n_d, const u32* __restrict__ cnt,
                                                        u32 row_stride, u32 n_pad, u32* __restrict__ gain_l) {
    __builtin_amdgcn_s_setprio(2);
    __shared__ u64 xch[3][kGlPlanes][64];
    const u32 b = blockIdx.y, lane = lane_id(), wv = threadIdx.x >> 6;
    const u32 n_gw = ri.n_gw, wg = min(blockIdx.x * 64u + lane, n_gw - 1u);
    const bool mine = blockIdx.x * 64u + lane < n_gw;
    const u32 nd64 = n_d[2], nr = lr.nlrow[b * kCtrStride];
    const uint2* rows = lr.lrow + (size_t)b * lr.lrow_stride;
    const u32* cb = cnt + (size_t)b * row_stride + nd64;
    u64 pl[kGlPlanes];
#pragma unroll
    for (u32 p = 0; p < kGlPlanes; ++p) pl[p] = 0ull;
    constexpr u32 U = 8;  // rows in flight per wave
    // (the rows are split over gridDim.z workgroups x 4 waves; every load of a round is issued before the first is used: no branch
    // around them -- rows past the end repeat the last one with weight 0)
    const u32 n_split = gridDim.z * 4u, me = blockIdx.z * 4u + wv;
    if (nr == 0u) return;
    for (u32 i0 = me * U; i0 < nr; i0 += n_split * U) {
        u64 m[U];
        u32 c[U];
        uint2 e[U];
#pragma unroll
        for (u32 u = 0; u < U; ++u) e[u] = rows[min(i0 + u, nr - 1u)];
#pragma unroll
        for (u32 u = 0; u < U; ++u) { c[u] = cb[e[u].y]; m[u] = ri.mlong[(size_t)e[u].x * n_gw + wg]; }
#pragma unroll
        for (u32 u = 0; u < U; ++u) { if (i0 + u >= nr) c[u] = 0u; if (!mine) m[u] = 0ull; }
#pragma unroll
        for (u32 u = 0; u < U; ++u) {
            u32 cc = (u32)__builtin_amdgcn_readfirstlane((int)c[u]);
            while (cc) {  // (weight c = its set bits, one ripple add each; almost always 1)
                const u32 k = (u32)__builtin_ctz(cc);
                cc &= cc - 1u;
                if (__ballot(m[u] != 0ull)) planes_add(pl, m[u], k);
            }
        }
    }
    // waves 1..3 hand their planes to wave 0 (bit-sliced addition: a full adder per plane)
    if (wv) {
#pragma unroll
        for (u32 p = 0; p < kGlPlanes; ++p) xch[wv - 1u][p][lane] = pl[p];
    }
    __syncthreads();
    if (wv == 0u) {
        for (u32 o = 0; o < 3u; ++o) {
            u64 carry = 0ull;
#pragma unroll
            for (u32 p = 0; p < kGlPlanes; ++p) {
                const u64 a = pl[p], x = xch[o][p][lane];
                pl[p] = a ^ x ^ carry;
                carry = (a & x) | (carry & (a ^ x));
            }
        }
        if (mine) {
            u32* out = gain_l + (size_t)b * n_pad + (size_t)wg * 64u;
            for (u32 bit = 0; bit < 64u; ++bit) {
                u32 v = 0;
#pragma unroll
                for (u32 p = 0; p < kGlPlanes; ++p) v |= (u32)((pl[p] >> bit) & 1ull) << p;
                if (v) atomicAdd(&out[bit], v);  // (gridDim.z workgroups share the word: gain_l is zero on entry)
            }
        }
    }
}
// the candidates of a batch by genome word (one workgroup per batch)
__global__ __launch_bounds__(1024) void cand_words_kernel(const u32* __restrict__ cand, u32 n_pad_c, u32 n_gw, u64* __restrict__ cw,
                                                          u32* __restrict__ cbase, u32* __restrict__ cwl, u32* __restrict__ ncwl) {
    __builtin_amdgcn_s_setprio(3);
    const u32 b = blockIdx.x;
    u64* w = cw + (size_t)b * n_gw;
    u32* base = cbase + (size_t)b * n_gw;
    for (u32 c = threadIdx.x; c < n_pad_c; c += 1024u) {
        const u32 g = cand[(size_t)b * n_pad_c + c];
        if (g != kCandNone) { atomicOr(&w[g >> 6], 1ull << (g & 63u)); atomicMin(&base[g >> 6], c); }
    }
    __syncthreads();
    for (u32 i0 = 0; i0 < n_gw; i0 += 1024u) {
        const u32 i = i0 + threadIdx.x;
        const bool nz = i < n_gw && w[i] != 0ull;
        const u64 bal = __ballot(nz);
        u32 at = 0;
        if (bal && lane_id() == 0u) at = atomicAdd(&ncwl[b], (u32)__popcll(bal));
        at = (u32)__shfl((int)at, 0);
        if (nz) cwl[(size_t)b * n_pad_c + at + (u32)__popcll(bal & lanemask_lt())] = i;
    }
}
// the long-list rows of the compact problems: a row of batch b whose bit row meets b's candidates gets a row behind the dense ones
// of b's compact matrix; its words are put together in LDS (candidate slot = the word's first slot + the rank of the bit among the
// word's candidates) and written with plain stores.  grid: (blocks, batches)
// which long-list rows of a batch hold one of its candidates: one wave per (batch, candidate slot)
__global__ __launch_bounds__(256) void cand_hit_kernel(const u32* __restrict__ cand, u32 n_pad_c, const u32* __restrict__ bad, RareIndex ri,
                                                       const u64* __restrict__ inb, u64* __restrict__ hit) {
    __builtin_amdgcn_s_setprio(2);
    const u32 b = blockIdx.y, c = blockIdx.x * 4u + (threadIdx.x >> 6), lane = lane_id();
    if (c >= n_pad_c || bad[b]) return;
    const u32 g = cand[(size_t)b * n_pad_c + c];
    if (g == kCandNone) return;
    const u64* col = ri.mlongT + (size_t)g * ri.n_lw;
    const u64* in = inb + (size_t)b * ri.n_lw;
    u64* out = hit + (size_t)b * ri.n_lw;
    for (u32 w = lane; w < ri.n_lw; w += 64u) {
        const u64 x = col[w] & in[w];
        if (x) atomicOr(&out[w], x);
    }
}

__global__ __launch_bounds__(256) void cand_long_kernel(LongRows lr, RareIndex ri, const u32* __restrict__ n_d, const u64* __restrict__ cw,
                                                        const u32* __restrict__ cbase, const u32* __restrict__ cwl, const u32* __restrict__ ncwl,
                                                        u32 n_pad_c, u32* __restrict__ bad, u32* __restrict__ nqc, u32* __restrict__ smap,
                                                        u32 smap_stride, u64* __restrict__ mqc, size_t mqc_stride, u32 rows_c,
                                                        u64* __restrict__ rowany_c, u32 rowany_stride, u32* __restrict__ grp_any_c, u32 n_grp_c,
                                                        const u64* __restrict__ hit) {
    __builtin_amdgcn_s_setprio(2);
    __shared__ u64 pat[4][kPatWords];
    const u32 b = blockIdx.y, lane = lane_id(), wv = threadIdx.x >> 6;
    if (bad[b]) return;
    const u32 nd64 = n_d[2], nr = lr.nlrow[b * kCtrStride], n_gw = ri.n_gw, n_gw_c = n_grp_c * kRankWords;
    const u32 nw = ncwl[b];
    if (nw == 0u) return;
    const uint2* rows = lr.lrow + (size_t)b * lr.lrow_stride;
    const u64* cwb = cw + (size_t)b * n_gw;
    const u32* cbb = cbase + (size_t)b * n_gw;
    const u32* wl = cwl + (size_t)b * n_pad_c;
    u64* pw = pat[wv];
    const bool in_lds = n_gw_c <= kPatWords;
    for (u32 i = blockIdx.x * 4u + wv; i < nr; i += gridDim.x * 4u) {
        const uint2 e = rows[i];
        // (cand_hit_kernel has the answer for every row of the batch at once: 45 k rows x ~180 scattered words each were 35 M L2 requests
        // per pass, 0.58 ms; the candidates' rows of the transposed matrix ANDed with the batch's row mask are 53 MB of coalesced reads)
        if (hit && !((hit[(size_t)b * ri.n_lw + (e.x >> 6)] >> (e.x & 63u)) & 1ull)) continue;
        const u64* mrow = ri.mlong + (size_t)e.x * n_gw;
        // does the row meet any candidate at all?  (almost never: the hash of some other clade)
        u64 any = 0ull;
        for (u32 k0 = 0; k0 < nw; k0 += 64u) {
            const u32 k = k0 + lane;
            u64 bits = 0ull;
            if (k < nw) { const u32 wg = wl[k]; bits = mrow[wg] & cwb[wg]; }
            any |= __ballot(bits != 0ull);
        }
        if (!any) continue;
        u32 x = 0;
        if (lane == 0u) {
            const u32 crow = nd64 + atomicAdd(&nqc[b * kCtrStride], 1u);
            if (crow + 1u >= rows_c) { atomicOr(&bad[b], 2u); x = kCandNone; }
            else { smap[(size_t)b * smap_stride + e.y] = crow + 1u; x = crow; }
        }
        const u32 cr = (u32)__shfl((int)x, 0);
        if (cr == kCandNone) return;  // (the batch's matrix is full: it ranks on everything)
        if (in_lds) {
            for (u32 w = lane; w < n_gw_c; w += 64u) pw[w] = 0ull;
            wave_sync();
        }
        for (u32 k0 = 0; k0 < nw; k0 += 64u) {
            const u32 k = k0 + lane;
            if (k < nw) {
                const u32 wg = wl[k];
                const u64 cm = cwb[wg];
                u64 bits = mrow[wg] & cm;
                const u32 s0 = cbb[wg];
                while (bits) {
                    const u32 bit = (u32)__builtin_ctzll(bits);
                    bits &= bits - 1ull;
                    const u32 slot = s0 + (u32)__popcll(cm & ((1ull << bit) - 1ull));
                    if (in_lds) atomicOr(&pw[slot >> 6], 1ull << (slot & 63u));
                    else atomicOr(&mqc[(size_t)b * mqc_stride + mq_index(slot >> 6, cr, rows_c)], 1ull << (slot & 63u));
                }
            }
        }
        if (in_lds) {
            wave_sync();
            for (u32 w0 = 0; w0 < n_gw_c; w0 += 64u) {
                const u32 w = w0 + lane;
                const u64 v = w < n_gw_c ? pw[w] : 0ull;
                if (v) mqc[(size_t)b * mqc_stride + mq_index(w, cr, rows_c)] = v;
                const u64 nzw = __ballot(v != 0ull);
                const u32 l0 = (lane / kRankWords) * kRankWords;
                if (lane == l0 && w < n_gw_c && ((nzw >> l0) & ((1ull << kRankWords) - 1ull)) != 0ull) {
                    const u32 grp = w / kRankWords;
                    atomicOr(&rowany_c[(size_t)b * rowany_stride + (size_t)grp * (rows_c >> 6) + (cr >> 6)], 1ull << (cr & 63u));
                    u32* ga = grp_any_c + (size_t)b * n_grp_c + grp;
                    if (*ga == 0u) atomicOr(ga, 1u);
                }
            }
            wave_sync();
        } else {
            // (compact problems wider than the LDS pattern: the bits went straight into the matrix; flag every group -- a flag for a
            // group without a bit only costs the ranking a look at an all-zero row)
            for (u32 grp = lane; grp < n_grp_c; grp += 64u) {
                atomicOr(&rowany_c[(size_t)b * rowany_stride + (size_t)grp * (rows_c >> 6) + (cr >> 6)], 1ull << (cr & 63u));
                u32* ga = grp_any_c + (size_t)b * n_grp_c + grp;
                if (*ga == 0u) atomicOr(ga, 1u);
            }
        }
    }
}
// ---- pattern rows of the compact problems (round 6).  Row nd64 + p of batch b's compact matrix = pattern p's bits at b's candidates
// (one wave per (pattern, batch) walks the pattern's list -- its representative's genome list -- once per pass; pcw[b][p][word] keeps the
// words for the rows below).  A row of the pass whose list is pattern + exceptions then needs no row of its own unless one of its
// exceptions is a candidate of the batch: smap -> the pattern's row; else a new row = the pattern's words with those bits flipped.
// The mapped rows start behind the patterns': the pass's nqc counters start at n_pat rounded up to 64 (pat_nqc_init_kernel).
__global__ void pat_nqc_init_kernel(u32* __restrict__ nqc, u32 v) {
    if (threadIdx.x < kPassBatchesMax) nqc[threadIdx.x * kCtrStride] = v;
}
__global__ __launch_bounds__(256) void cand_pat_rows_kernel(RareIndex ri, const u32* __restrict__ n_d, const u32* __restrict__ candmask,
                                                            const u32* __restrict__ candslot, u32 n_pad, u32* __restrict__ bad,
                                                            u64* __restrict__ pcw, u64* __restrict__ mqc, size_t mqc_stride, u32 rows_c,
                                                            u64* __restrict__ rowany_c, u32 rowany_stride, u32* __restrict__ grp_any_c,
                                                            u32 n_grp_c) {
    __builtin_amdgcn_s_setprio(2);
    __shared__ u64 pat[4][kPatWords];
    const u32 b = blockIdx.y, lane = lane_id(), wv = threadIdx.x >> 6, p = blockIdx.x * 4u + wv;
    if (p >= ri.n_pat || bad[b]) return;
    const u32 nd64 = n_d[2], n_gw_c = n_grp_c * kRankWords, npat64 = (ri.n_pat + 63u) & ~63u;
    if (nd64 + npat64 + 64u >= rows_c) { if (lane == 0u) atomicOr(&bad[b], 2u); return; }  // (no room for the patterns' rows: the batch ranks on everything)
    const u32 cr = nd64 + p;
    u64* pw = pat[wv];
    for (u32 w = lane; w < n_gw_c; w += 64u) pw[w] = 0ull;
    wave_sync();
    const u32 slot = ri.lslot[ri.pat_rep[p]], o = ri.off[slot], n = ri.cnt[slot];
    for (u32 j = lane; j < n; j += 64u) {
        const u32 g = ri.post[o + j];
        if ((candmask[g] >> b) & 1u) {
            const u32 s_ = candslot[(size_t)b * n_pad + g];
            atomicOr(&pw[s_ >> 6], 1ull << (s_ & 63u));
        }
    }
    wave_sync();
    u64* out = pcw + ((size_t)b * ri.n_pat + p) * n_gw_c;
    for (u32 w0 = 0; w0 < n_gw_c; w0 += 64u) {
        const u32 w = w0 + lane;
        const u64 v = w < n_gw_c ? pw[w] : 0ull;
        if (w < n_gw_c) out[w] = v;
        if (v) mqc[(size_t)b * mqc_stride + mq_index(w, cr, rows_c)] = v;
        const u64 nzw = __ballot(v != 0ull);
        const u32 l0 = (lane / kRankWords) * kRankWords;
        if (lane == l0 && w < n_gw_c && ((nzw >> l0) & ((1ull << kRankWords) - 1ull)) != 0ull) {
            const u32 grp = w / kRankWords;
            atomicOr(&rowany_c[(size_t)b * rowany_stride + (size_t)grp * (rows_c >> 6) + (cr >> 6)], 1ull << (cr & 63u));
            u32* ga = grp_any_c + (size_t)b * n_grp_c + grp;
            if (*ga == 0u) atomicOr(ga, 1u);
        }
    }
}
// one THREAD per listed pattern row of a batch (grid: (blocks, batches))
__global__ __launch_bounds__(256) void cand_pat_map_kernel(LongRows lr, PatRows pr, RareIndex ri, const u32* __restrict__ n_d,
                                                           const u32* __restrict__ candmask, const u32* __restrict__ candslot, u32 n_pad,
                                                           u32* __restrict__ bad, u32* __restrict__ nqc, u32* __restrict__ smap, u32 smap_stride,
                                                           const u64* __restrict__ pcw, u64* __restrict__ mqc, size_t mqc_stride, u32 rows_c,
                                                           u64* __restrict__ rowany_c, u32 rowany_stride, u32* __restrict__ grp_any_c,
                                                           u32 n_grp_c) {
    __builtin_amdgcn_s_setprio(2);
    const u32 b = blockIdx.y;
    if (bad[b]) return;
    const u32 nd64 = n_d[2], nr = pr.nprow[b * kCtrStride], n_gw_c = n_grp_c * kRankWords;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < nr; i += gridDim.x * 256u) {
        const uint2 e = lr.lrow[(size_t)b * lr.lrow_stride + (lr.lrow_stride - 1u - i)];
        const u32* rec = ri.prec + (size_t)e.x * kPatRec;
        const u32 p = rec[0], n_exc = rec[1];
        u32 fl[kPatExcMax];
        u32 nf = 0;
#pragma unroll
        for (u32 j = 0; j < kPatExcMax; ++j) {
            fl[j] = kCandNone;
            if (j < n_exc) {
                const u32 g = rec[2u + j] & 0x7FFFFFFFu;
                if ((candmask[g] >> b) & 1u) { fl[j] = candslot[(size_t)b * n_pad + g]; ++nf; }
            }
        }
        if (nf == 0u) { smap[(size_t)b * smap_stride + e.y] = nd64 + p + 1u; continue; }
        const u32 cr = nd64 + atomicAdd(&nqc[b * kCtrStride], 1u);
        if (cr + 1u >= rows_c) { atomicOr(&bad[b], 2u); continue; }  // (the batch's matrix is full: it ranks on everything)
        smap[(size_t)b * smap_stride + e.y] = cr + 1u;
        const u64* src = pcw + ((size_t)b * ri.n_pat + p) * n_gw_c;
        for (u32 grp = 0; grp < n_grp_c; ++grp) {
            u64 any = 0ull;
#pragma unroll
            for (u32 k = 0; k < kRankWords; ++k) {
                const u32 w = grp * kRankWords + k;
                u64 v = src[w];
#pragma unroll
                for (u32 j = 0; j < kPatExcMax; ++j) v ^= (fl[j] >> 6) == w ? 1ull << (fl[j] & 63u) : 0ull;  // (kCandNone >> 6 is no word)
                if (v) mqc[(size_t)b * mqc_stride + mq_index(w, cr, rows_c)] = v;
                any |= v;
            }
            if (any) {
                atomicOr(&rowany_c[(size_t)b * rowany_stride + (size_t)grp * (rows_c >> 6) + (cr >> 6)], 1ull << (cr & 63u));
                u32* ga = grp_any_c + (size_t)b * n_grp_c + grp;
                if (*ga == 0u) atomicOr(ga, 1u);
            }
        }
    }
}
// gain_s (or NULL): the rare rows' part, one entry per kGainSparseStride words (gain_sparse_kernel)
// (prev may lie INSIDE tab -- the table a pass on the same buffer set left as its last row: every thread reads prev[g] before it writes
// column g and touches no other column, so the in-place case is well defined; hence no __restrict__ on the two)
__global__ __launch_bounds__(256) void pass_tables_kernel(const u64* prev, const u32* __restrict__ gain,
                                                          const u32* __restrict__ gain_s, const u32* __restrict__ gain_l, u32 n_b, u32 n_pad,
                                                          u64* tab) {
    __builtin_amdgcn_s_setprio(3);
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_pad) return;
    u64 t = prev[g];
    tab[g] = t;
    for (u32 b = 0; b < n_b; ++b) {
        t += gain[(size_t)b * n_pad + g];
        if (gain_s) t += gain_s[((size_t)b * n_pad + g) * kGainSparseStride];
        if (gain_l) t += gain_l[(size_t)b * n_pad + g];  // (the long-list rows' part: gain_long_kernel)
        tab[(size_t)(b + 1u) * n_pad + g] = t;
    }
}

// block-wide helpers of the candidate selection (1024 threads = 16 waves)
__device__ __forceinline__ u64 block1024_max_u64(u64 v, u64* sh /* [16] */) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, shfl_xor64(v, d));
    __syncthreads();
    if (lane_id() == 0u) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 r = sh[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) r = max(r, sh[i]);
    return r;
}
__device__ __forceinline__ u32 block1024_sum_u32(u32 v, u32* sh /* [16] */) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += (u32)__shfl_xor((int)v, d, 64);
    __syncthreads();
    if (lane_id() == 0u) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    u32 r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += sh[i];
    return r;
}
// the top_k-th best value among t0[g0 .. g0 + n): at most top_k rounds of "largest value below the previous one", counting multiplicities
__device__ __forceinline__ u64 species_theta(const u64* __restrict__ t0, u32 g0, u32 n, u32 top_k, u64* sh64, u32* sh32) {
    const u32 tid = threadIdx.x;
    u64 theta = 0, prev = 0;
    u32 remaining = max(top_k, 1u);
    bool first = true;
    // (every round settles at least one of the top_k places: at most top_k rounds -- round 5 stopped after 64 and returned 0 for larger
    // top_k with distinct values: exact still, but every genome a candidate for ever)
    for (u32 round = 0, n_rounds = max(top_k, 1u); round < n_rounds; ++round) {
        u64 m = 0;
        u32 any = 0;
        for (u32 g = g0 + tid; g < g0 + n; g += 1024u) {
            const u64 v = t0[g];
            if (first || v < prev) { m = max(m, v); any = 1u; }
        }
        m = block1024_max_u64(m, sh64);
        any = block1024_sum_u32(any, sh32);
        if (!any) { theta = 0; break; }  // (fewer than top_k distinct positions: everything counts)
        u32 c = 0;
        for (u32 g = g0 + tid; g < g0 + n; g += 1024u) c += (t0[g] == m) ? 1u : 0u;
        c = block1024_sum_u32(c, sh32);
        if (c >= remaining) { theta = m; break; }
        remaining -= c;
        prev = m;
        first = false;
        theta = 0;
    }
    return theta;
}
// How many candidates did a batch have whose ranking ran on everything (legacy passes: t0 / t1 = the table before / after it)?  The
// largest count over the species goes to the host (h_out[0], then h_out[1] = seq): a hint for the NEXT passes only.
__global__ __launch_bounds__(1024) void cand_count_kernel(const u64* __restrict__ t0, const u64* __restrict__ t1, Species sp, u32 top_k,
                                                          u32* __restrict__ d_max, volatile u32* __restrict__ h_out, u32 seq) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ u64 sh64[16];
    __shared__ u32 sh32[16];
    u32 worst = 0;
    for (u32 spi = 0; spi < sp.n_sp; ++spi) {
        const u32 g0 = sp.g0[spi], n = sp.n[spi];
        const u64 theta = species_theta(t0, g0, n, top_k, sh64, sh32);
        u32 c = 0;
        for (u32 g = g0 + threadIdx.x; g < g0 + n; g += 1024u) c += t1[g] >= theta ? 1u : 0u;
        worst = max(worst, block1024_sum_u32(c, sh32));
    }
    if (threadIdx.x == 0u) {
        (void)d_max;
        h_out[0] = worst;
        __threadfence_system();
        h_out[1] = seq;
        __threadfence_system();
    }
}
// One workgroup per (batch, species): theta = the k-th best value of the species as the batch begins; candidates = real genomes
// whose value as it ENDS reaches theta, in reference order.  cand[(b n_sp + sp) cap + i] = padded genome index, candslot[b][g] =
// sp cap + i (or none), tabc[b][sp cap + i] = the genome's start value (slots behind the candidates: 0, never ranked: ncand bounds
// the species), ncand[b n_sp + sp] = min(count, cap), bad[b] |= 1 when a species has more than cap.
__global__ __launch_bounds__(1024) void cand_select_kernel(const u64* __restrict__ tab, u32 n_pad, Species sp, u32 top_k, u32 cap,
                                                           u32* __restrict__ cand, u32* __restrict__ candslot, u64* __restrict__ tabc,
                                                           u32* __restrict__ ncand, u32* __restrict__ bad, u32* __restrict__ candmask) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ u64 sh64[16];
    __shared__ u32 sh32[16];
    __shared__ u32 wbase[17];
    const u32 b = blockIdx.x / sp.n_sp, spi = blockIdx.x % sp.n_sp, tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const u32 g0 = sp.g0[spi], n = sp.n[spi];
    const u32 g_end = spi + 1u < sp.n_sp ? sp.g0[spi + 1u] : n_pad;
    const u64* t0 = tab + (size_t)b * n_pad;
    const u64* t1 = t0 + n_pad;
    const u64 theta = species_theta(t0, g0, n, top_k, sh64, sh32);
    // ordered compaction
    u32 base = 0;
    u32* my_cand = cand + (size_t)(b * sp.n_sp + spi) * cap;
    u64* my_tabc = tabc + (size_t)b * sp.n_sp * cap + (size_t)spi * cap;
    for (u32 c0 = g0; c0 < g_end; c0 += 1024u) {
        const u32 g = c0 + tid;
        const bool is = g < g0 + n && t1[g] >= theta;
        const u64 bal = __ballot(is);
        const u32 before = (u32)__popcll(bal & lanemask_lt()), wtot = (u32)__popcll(bal);
        __syncthreads();
        if (lane == 0u) wbase[wv] = wtot;
        __syncthreads();
        u32 wb = 0, tot = 0;
#pragma unroll
        for (u32 i = 0; i < 16u; ++i) { const u32 v = wbase[i]; wb += i < wv ? v : 0u; tot += v; }
        const u32 pos = base + wb + before;
        if (g < g_end) {
            u32 slot = kCandNone;
            if (is && pos < cap) { my_cand[pos] = g; my_tabc[pos] = t0[g]; slot = spi * cap + pos; atomicOr(&candmask[g], 1u << b); }
            candslot[(size_t)b * n_pad + g] = slot;  // (candmask[g], zero on entry: the batches g is a candidate of)
        }
        base += tot;
    }
    for (u32 i = min(base, cap) + tid; i < cap; i += 1024u) { my_cand[i] = kCandNone; my_tabc[i] = 0; }
    if (tid == 0u) {
        ncand[b * sp.n_sp + spi] = min(base, cap);
        if (base > cap) atomicOr(&bad[b], 1u);
    }
}
// M_c[b][w][c] = M[w][cand[b][c]] for the dense words (batches the selection left in compact mode); slots without a candidate: 0
__global__ __launch_bounds__(256) void cand_gather_m_kernel(const u64* __restrict__ m_bits, const u64* __restrict__ m_int, u32 n_pad,
                                                            const u32* __restrict__ n_d,
                                                            const u32* __restrict__ cand, u32 n_pad_c, const u32* __restrict__ bad,
                                                            u64* __restrict__ mc, u32 words_c) {
    __builtin_amdgcn_s_setprio(2);
    const u32 b = blockIdx.z, w = blockIdx.y, c = blockIdx.x * 256u + threadIdx.x;
    if (bad[b] || w >= ((n_d[0] + 63u) >> 6) || c >= n_pad_c) return;
    const u32 g = cand[(size_t)b * n_pad_c + c];
    u64 x = 0;
    if (g != kCandNone) { x = m_bits[(size_t)w * n_pad + g]; if (m_int) x |= m_int[(size_t)w * n_pad + g]; }
    mc[((size_t)b * words_c + w) * n_pad_c + c] = x;
}
// the rare rows of the compact problems: a rare hash whose genome list meets batch b's candidates gets a row behind the dense ones of
// b's compact matrix (nqc[b] counts them; smap[b][sparse row] = that row + 1), its bits go straight into the group-major matrix
// Mq_c[b] (zero on entry behind the dense rows, as are rowany_c / grp_any_c); rows of hashes no candidate holds stay unmapped (the
// pair remap sends them to an all-zero row).  More rows than the matrix holds: bad[b] |= 2.  ONE walk over the postings for all
// batches: candmask[g] says which batches genome g is a candidate of (almost always none).
__global__ __launch_bounds__(256) void cand_sparse_kernel(const u32* __restrict__ sslot, const u32* __restrict__ n_d, RareIndex ri,
                                                          const u32* __restrict__ candmask, const u32* __restrict__ candslot, u32 n_pad,
                                                          u32* __restrict__ bad, u32 n_b, u32* __restrict__ nqc, u32* __restrict__ smap,
                                                          u32 smap_stride, u64* __restrict__ mqc, size_t mqc_stride, u32 rows_c,
                                                          u64* __restrict__ rowany_c, u32 rowany_stride, u32* __restrict__ grp_any_c,
                                                          u32 n_grp_c) {
    __builtin_amdgcn_s_setprio(2);
    __shared__ u64 pat[4][kPassBatchesMax][kPatWords];  // per wave: the words of a long-list row, per batch
    const u32 nd64 = n_d[2], ns = n_d[1], lane = lane_id();
    const u32 n_gw_c = n_grp_c * kRankWords;
    const u32 wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    // (atomics are executed at the memory side, ~0.4 ns each whatever the address: one per BIT, one per (row, batch) for the row
    // itself and one per (row, batch, rank group) for the row flags -- not one per posting)
    auto put = [&](u32 b, u32 crow, u32 slot) { atomicOr(&mqc[(size_t)b * mqc_stride + mq_index(slot >> 6, crow, rows_c)], 1ull << (slot & 63u)); };
    auto flag = [&](u32 b, u32 crow, u32 grp) {
        atomicOr(&rowany_c[(size_t)b * rowany_stride + (size_t)grp * (rows_c >> 6) + (crow >> 6)], 1ull << (crow & 63u));
        u32* ga = grp_any_c + (size_t)b * n_grp_c + grp;
        if (*ga == 0u) atomicOr(ga, 1u);
    };
    auto new_row = [&](u32 b, u32 sr) -> u32 {
        if (bad[b]) return kCandNone;
        const u32 crow = nd64 + atomicAdd(&nqc[b * kCtrStride], 1u);
        if (crow + 1u >= rows_c) { atomicOr(&bad[b], 2u); return kCandNone; }
        smap[(size_t)b * smap_stride + sr] = crow + 1u;  // (0 = not mapped)
        return crow;
    };
    for (u32 r0 = wave * 64u; r0 < ns; r0 += n_waves * 64u) {
        const u32 sr = r0 + lane;
        u32 off = 0, np = 0;
        if (sr < ns) { const uint2 e = reinterpret_cast<const uint2*>(sslot)[sr]; off = e.x; np = e.y; }
        if (np & kLongFlag) np = 0;  // (rows with a bit row: cand_long_kernel)
        if (np && np <= 8u) {
            u32 crow[kPassBatchesMax];
#pragma unroll
            for (u32 b = 0; b < kPassBatchesMax; ++b) crow[b] = kCandNone;
            // (tried: the list and its genomes' candidate masks requested at once instead of entry by entry -- the kernels 5 % shorter, the
            // stream not: 103.8 M reads/s either way)
            for (u32 j = 0; j < np; ++j) {
                const u32 g = ri.post[off + j];
                u32 m = candmask[g];
                while (m) {
                    const u32 b = (u32)__builtin_ctz(m);
                    m &= m - 1u;
                    if (b >= n_b) break;
                    u32 cr = kCandNone;
#pragma unroll
                    for (u32 i = 0; i < kPassBatchesMax; ++i) cr = i == b ? crow[i] : cr;
                    const bool fresh = cr == kCandNone;
                    if (fresh) {
                        cr = new_row(b, sr);
#pragma unroll
                        for (u32 i = 0; i < kPassBatchesMax; ++i) if (i == b) crow[i] = cr == kCandNone ? 0xFFFFFFFEu : cr;
                    }
                    if (cr >= 0xFFFFFFFEu) continue;  // (the batch's matrix is full: it ranks on everything)
                    const u32 s_ = candslot[(size_t)b * n_pad + g];
                    put(b, cr, s_);
                    flag(b, cr, (s_ >> 6) / kRankWords);  // (short lists: a flag per bit is at most 8 per row)
                }
            }
        }
        u64 longs = __ballot(np > 8u);
        while (longs) {
            const u32 src = (u32)__builtin_ctzll(longs);
            longs &= longs - 1ull;
            const u32 o = __shfl(off, (int)src), n = __shfl(np, (int)src);
            if (n_gw_c <= kPatWords) {
                // Long lists (a lineage's hash: ~200 genomes, all of them candidates when it is the sample's own lineage): the row's
                // words are put together in LDS -- per batch, the bits of the candidates on the list -- and written ONCE, with plain
                // stores (the row is this wave's alone): 3.2 M atomicOr per C2 pass became 0.26 M stores.
                u64* pw = &pat[threadIdx.x >> 6][0][0];
                for (u32 i = lane; i < n_b * n_gw_c; i += 64u) pw[(i / n_gw_c) * kPatWords + i % n_gw_c] = 0ull;
                wave_sync();
                u32 seen = 0;  // batches some candidate of which is on the list
                for (u32 j0 = 0; j0 < n; j0 += 64u) {
                    const u32 j = j0 + lane;
                    const u32 g = j < n ? ri.post[o + j] : 0u;
                    u32 m = j < n ? candmask[g] : 0u;
                    seen |= m;
                    while (m) {
                        const u32 b = (u32)__builtin_ctz(m);
                        m &= m - 1u;
                        if (b >= n_b) break;
                        const u32 s_ = candslot[(size_t)b * n_pad + g];
                        atomicOr(&pw[b * kPatWords + (s_ >> 6)], 1ull << (s_ & 63u));
                    }
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) seen |= (u32)__shfl_xor((int)seen, d, 64);
                wave_sync();
                while (seen) {
                    const u32 b = (u32)__builtin_ctz(seen);
                    seen &= seen - 1u;
                    if (b >= n_b) break;
                    u32 x = 0;
                    if (lane == 0u) x = new_row(b, r0 + src);
                    const u32 cr = (u32)__shfl((int)x, 0);
                    if (cr == kCandNone) continue;  // (the batch's matrix is full: it ranks on everything)
                    for (u32 w0 = 0; w0 < n_gw_c; w0 += 64u) {
                        const u32 w = w0 + lane;
                        const u64 v = w < n_gw_c ? pw[b * kPatWords + w] : 0ull;
                        if (v) mqc[(size_t)b * mqc_stride + mq_index(w, cr, rows_c)] = v;
                        // one flag per rank group with a bit: the first lane of every run of kRankWords words
                        const u64 nz = __ballot(v != 0ull);
                        const u32 grp_lane0 = (lane / kRankWords) * kRankWords;
                        const bool first = ((nz >> grp_lane0) & ((1ull << kRankWords) - 1ull)) != 0ull && lane == grp_lane0;
                        if (first && w < n_gw_c) flag(b, cr, w / kRankWords);
                    }
                }
                wave_sync();
                continue;
            }
            u32 crow[kPassBatchesMax];           // (wave-uniform)
            u64 flagged[kPassBatchesMax];        // rank groups (mod 64) the row is flagged for already, per batch
#pragma unroll
            for (u32 b = 0; b < kPassBatchesMax; ++b) { crow[b] = kCandNone; flagged[b] = 0; }
            for (u32 j0 = 0; j0 < n; j0 += 64u) {
                const u32 j = j0 + lane;
                const u32 g = j < n ? ri.post[o + j] : 0u;
                const u32 m = j < n ? candmask[g] : 0u;
                u32 all = m;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) all |= (u32)__shfl_xor((int)all, d, 64);
                while (all) {
                    const u32 b = (u32)__builtin_ctz(all);
                    all &= all - 1u;
                    if (b >= n_b) break;
                    u32 cr = kCandNone;
                    u64 fl = 0;
#pragma unroll
                    for (u32 i = 0; i < kPassBatchesMax; ++i) { cr = i == b ? crow[i] : cr; fl = i == b ? flagged[i] : fl; }
                    if (cr == kCandNone) {
                        u32 x = 0;
                        if (lane == 0u) x = new_row(b, r0 + src);
                        cr = (u32)__shfl((int)x, 0);
                        if (cr == kCandNone) cr = 0xFFFFFFFEu;
                    }
                    const bool mine = (m >> b) & 1u;
                    u32 s_ = kCandNone;
                    if (cr < 0xFFFFFFFEu && mine) { s_ = candslot[(size_t)b * n_pad + g]; put(b, cr, s_); }
                    u64 hit = cr < 0xFFFFFFFEu ? __ballot(mine) : 0ull;
                    while (hit) {  // one flag per (row, batch, group): the first lane of every group that is new to the row
                        const u32 l0 = (u32)__builtin_ctzll(hit);
                        const u32 grp = (u32)__shfl((int)((s_ >> 6) / kRankWords), (int)l0);
                        hit &= ~__ballot(mine && (s_ >> 6) / kRankWords == grp);
                        if (!((fl >> (grp & 63u)) & 1ull) || grp >= 64u) {
                            if (lane == l0) flag(b, cr, grp);
                            fl |= 1ull << (grp & 63u);
                        }
                    }
#pragma unroll
                    for (u32 i = 0; i < kPassBatchesMax; ++i) if (i == b) { crow[i] = cr; flagged[i] = fl; }
                }
            }
        }
    }
}
// decisions of a pass, for the host: mode[b] = 1 (compact ranking) / 0 (everything), the candidates' largest count, whether any
// batch needs the full bit matrix (the transpose of M and the fill of its rare rows only run then), sequence number last
__global__ void cand_publish_kernel(const u32* __restrict__ bad, u32 force_full, const u32* __restrict__ ncand, const u32* __restrict__ nqc,
                                    const u32* __restrict__ n_d, u32 n_b, u32 n_sp, u32 rows_c, u32* __restrict__ mode, u32* __restrict__ any_full,
                                    u32* __restrict__ nqc_total, volatile u32* __restrict__ h_pub, u32 seq) {
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 any = 0;
    const u32 nd64 = n_d[2];
    for (u32 b = 0; b < n_b; ++b) {
        const bool full = bad[b] != 0u || ((force_full >> b) & 1u) || nd64 + 64u >= rows_c;
        mode[b] = full ? 0u : 1u;
        any |= full ? 1u : 0u;
        u32 mx = 0;
        for (u32 s_ = 0; s_ < n_sp; ++s_) mx = max(mx, ncand[b * n_sp + s_]);
        nqc_total[b] = nd64 + nqc[b * kCtrStride];   // rows of the compact problem (dense rows, padded to 64, + the mapped rare rows)
        h_pub[b] = full ? 0u : 1u;
        h_pub[kPassBatchesMax + b] = mx;
    }
    *any_full = any;
    h_pub[2 * kPassBatchesMax] = any;
    h_pub[2 * kPassBatchesMax + 2] = n_b;
    __threadfence_system();
    h_pub[2 * kPassBatchesMax + 1] = seq;
    __threadfence_system();
}
// nobody needs the full bit matrix this pass: M's dense words go back to all-zero here instead of in the transpose
__global__ __launch_bounds__(256) void m_clear_kernel(u64* __restrict__ m_bits, u64* __restrict__ m_int, u32 n_pad,
                                                      const u32* __restrict__ n_d, const u32* __restrict__ any_full) {
    if (*any_full) return;
    const u32 n_words = (n_d[0] + 63u) >> 6;
    const size_t n = (size_t)n_words * n_pad;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        if (m_bits[i]) m_bits[i] = 0ull;
        if (m_int && m_int[i]) m_int[i] = 0ull;
    }
}
// pairs of a compact batch: row of the pass's matrix -> row of the compact one (dense rows keep theirs; rare rows through smap,
// unmapped ones to the all-zero last row)
__global__ __launch_bounds__(256) void cand_pair_rows_kernel(const u32* __restrict__ pair_q, u32 n_pairs, const u32* __restrict__ n_d,
                                                             const u32* __restrict__ smap, u32 rows_c, u32* __restrict__ pair_qc) {
    __builtin_amdgcn_s_setprio(3);
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_pairs) return;
    const u32 row = pair_q[p], nd64 = n_d[2];
    u32 out = row;
    if (row >= nd64) { const u32 m = smap[row - nd64]; out = m ? m - 1u : rows_c - 1u; }
    pair_qc[p] = out;
}
// rows of a compact batch: candidate slot (local to the species of the compact problem) -> genome index local to the species
__global__ __launch_bounds__(256) void cand_rows_back_kernel(u32* __restrict__ out_idx, u32 n_rows /* reads x species x top */, u32 n_sp,
                                                             u32 top_k, const u32* __restrict__ cand, u32 cap, const u32* __restrict__ g0) {
    __builtin_amdgcn_s_setprio(3);
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_rows) return;
    const u32 spi = (i / top_k) % n_sp;
    const u32 slot = out_idx[i];
    if (slot < cap) out_idx[i] = cand[(size_t)spi * cap + slot] - g0[spi];
}

// =====================================================================================
// launchers
// =====================================================================================
static inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }
static int getenv_int_early(const char* name, int dflt) { const char* e = knob(name); return e ? atoi(e) : dflt; }
static const u32 kWalkBlocks = (u32)(getenv_int_early("SKX_WALK_BLOCKS", SKX_WALK_BLOCKS));  // workgroups of the kernels that walk the rare rows' genome lists
static int env_int(const char* name, int dflt) { const char* e = knob(name); return e ? atoi(e) : dflt; }

void launch_ref_tile(hipStream_t st, const u64* src, const u32* eff_len, u64* dst, u32 s, u32 pad_base, u32 g_count) {
    dim3 grid(cdiv(s, 32), cdiv(g_count, 32));
    hipLaunchKernelGGL(ref_tile_kernel, grid, dim3(256), 0, st, src, eff_len, dst, s, pad_base, g_count);
}
void launch_band_bounds(hipStream_t st, const u64* mat, u32 s, u32 n_tiles, u32 rb, u32 n_bands, u64* lo, u64* hi) {
    hipLaunchKernelGGL(band_bounds_kernel, dim3(n_tiles * n_bands), dim3(256), 0, st, mat, s, n_tiles, rb, lo, hi);
}

static size_t sketch_wave_lds(int hcap, bool prefilter) { return 4 * (size_t)(hcap * 8 + kSketchCap + 128 + (prefilter ? kPfQueue * 4 : 0)); }

hipError_t launch_sketch(hipStream_t st, const uint8_t* bases, const u64* offsets, u32 n_reads, u32 k, u64 seed, u32 s,
                         u64 max_ref, bool inrange_only, u64* out_sk, u32 sk_stride, u32* out_len, u32* out_cnt_in,
                         const u64* filt, u32 filt_shift, u32* retry, u32* big, u64 n_bases, u32* chk, int leave_room, bool packed,
                         const LongReads* long_reads, const KmerFilter* kmer_filter, int phase, u32 pool_cap, u32 pool_fixed) {
    if (n_reads == 0) return hipSuccess;
    // leave_room: the previous pass's scan is still running on another stream.  The fast variant then asks for extra
    // dynamic LDS per block (env SKX_SKETCH_LDS_PAD, default 19 KB: 4 instead of 8 of its blocks fit a CU, and when one
    // leaves, the LDS it frees goes to the -- higher-priority -- scan's blocks first), which leaves wave slots, registers
    // and LDS for the scan: the VALU-bound sketch and the HBM-bound scan share every CU instead of taking turns.
    // End of round 2, C2, same box: pad 11 KB 76.1 M reads/s with the scan at 1.00 ms (0.40 of peak) in the pipeline; 18 KB
    // sits on the threshold (either 0.41 or 0.50); **19-20 KB 76.3-77.5 M with the scan at 0.71-0.75 ms (0.53-0.56)**; 22 KB
    // 75.5 M, 0.55; 23-28 KB 70-72 M with the scan at its stand-alone 0.61-0.65 ms (0.61-0.65).  C4: 21.3 M / 0.40 at
    // 11 KB, 21.1 M / 0.53 at 18 KB, 20.6 M / 0.58 at 20 KB.
    // SKX_SKETCH_ROOM: 2 (default) = unused dynamic LDS per block (SKX_SKETCH_LDS_PAD bytes), 1 = the register-capped
    // variant (5 waves per SIMD; measured: the scan then runs at its stand-alone speed inside the pipeline, 0.61-0.63 ms =
    // 0.64-0.66 of peak, but the step takes 1.48-1.50 ms instead of 1.32-1.33: 66 M reads/s instead of 74 M), 0 = never
    static const int room_mode = env_int("SKX_SKETCH_ROOM", 2);
    // (leave_room = 2: the batch was enqueued behind another one -- the scan of that one WILL run beside this sketch for
    // most of its time: the larger pad; 1: a push found the previous scan still in flight -- it overlaps only the start of
    // this sketch: 11 KB, SKX_SKETCH_LDS_PAD_PUSH (measured: 72-73 M reads/s through skx_stream_push_device, 64-65 M with 19 KB))
    static const size_t lds_pad_enq = (size_t)env_int("SKX_SKETCH_LDS_PAD", 19456);
    static const size_t lds_pad_push = (size_t)env_int("SKX_SKETCH_LDS_PAD_PUSH", 11264);
    const size_t lds_pad_env = leave_room >= 2 ? lds_pad_enq : lds_pad_push;
    static const size_t lds_pad_capped = (size_t)env_int("SKX_SKETCH_LDS_PAD_CAPPED", 0);
    const size_t lds_pad = !leave_room ? 0 : room_mode == 2 ? lds_pad_env : room_mode == 1 ? lds_pad_capped : 0;
    const bool capped = leave_room && room_mode == 1;
    const bool with_pf = kmer_filter && kmer_filter->words && inrange_only && k == 16;
    const size_t lds = sketch_wave_lds(kSketchCap, with_pf), lds_small = sketch_wave_lds(kSketchSmallHashes, with_pf) + lds_pad;
    dim3 grid(cdiv(n_reads, 4));
#define SKX_SK(KT, IR) sketch_wave_kernel<KT, kSketchCap, IR>
#define SKX_SK_SMALL(KT) sketch_wave_kernel<KT, kSketchSmallHashes, true>
#define SKX_BLK(KT, IR) sketch_block_kernel<KT, IR>
    // > 64 KiB of dynamic LDS needs the opt-in (gfx950 has 160 KiB per CU).  The attribute belongs to the (function,
    // device) pair and handles on different devices may be driven from different threads: one latch per device.
    {
        static std::mutex mu;
        static unsigned long long done[4] = {0, 0, 0, 0};  // devices 0..255
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lock(mu);
        if (dev < 0 || dev >= 256 || !((done[dev >> 6] >> (dev & 63)) & 1ull)) {
            const void* wave_fns[] = {(const void*)&SKX_SK(16, false), (const void*)&SKX_SK(16, true),
                                      (const void*)&SKX_SK(0, false), (const void*)&SKX_SK(0, true)};
            for (const void* f : wave_fns) {
                e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sketch_wave_lds(kSketchCap, true));
                if (e != hipSuccess) return e;
            }
            const void* blk_fns[] = {(const void*)&SKX_BLK(16, false), (const void*)&SKX_BLK(16, true),
                                     (const void*)&SKX_BLK(0, false), (const void*)&SKX_BLK(0, true)};
            for (const void* f : blk_fns) {
                e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBigLds);
                if (e != hipSuccess) return e;
            }
            if (dev >= 0 && dev < 256) done[dev >> 6] |= 1ull << (dev & 63);
        }
    }
    const dim3 list_grid(std::min<u32>(n_reads, 1024u)), blk_grid(std::min<u32>(n_reads, 256u));
    const dim3 grid2(kSegBlocks + cdiv(n_reads, 4));  // segment workgroups first; needs the tables of batch_check_kernel
    const LongReads lr = long_reads ? *long_reads : LongReads{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0u};
    const KmerFilter kf = (kmer_filter && inrange_only && k == 16) ? *kmer_filter : KmerFilter{nullptr, 0u};
    // (the list walk -- usually over an empty list -- goes out as ONE-wave blocks with a quarter of the LDS, 18.6 KB: a
    // 74 KB block would wait for a CU the scan's blocks have left, i.e. for the end of the scan running beside it:
    // 140-180 us on the sketch stream, measured)
#define SKX_SK_LAUNCH(KERNEL, LDS, FROM_LIST)                                                                              \
    hipLaunchKernelGGL((KERNEL), (FROM_LIST) == 1u ? list_grid : (FROM_LIST) == 2u ? grid2 : grid,                        \
                       dim3((FROM_LIST) == 1u ? 64 : 256), (FROM_LIST) == 1u ? (LDS) / 4 : (LDS), st, bases,                \
                       offsets, n_reads, k, seed, s, max_ref, out_sk, sk_stride, out_len, out_cnt_in,                          \
                       (u32)(FROM_LIST) | (packed ? 0x100u : 0u), retry, big, filt, filt_shift, n_bases, chk, lr, kf, pool_cap, pool_fixed)
    (void)blk_grid;
    if (inrange_only) {
        // fast variant first (256 hash slots: full occupancy); reads it flags are redone with 2048 slots, what still
        // does not fit is left on the `big` list for launch_sketch_block -- both lists live on the device, usually empty
        // (with the long-read tables: segment workgroups in front, long reads skipped by the read waves and finished by
        // sketch_merge_kernel -- one-wave workgroups with a 16 KB buffer each, walking the list; without long reads they
        // return at once)
        // phase bit 0: the main kernel; bit 1: the two list walks behind it (long-read merge, 2048-slot retry) -- the caller
        // may put them on another stream (behind an event), so the next batch's main kernel follows this one back to back
        const u32 first = (chk && lr.list) ? 2u : 0u;
        if (phase & 1) {
            if (k == 16) {
                if (capped) SKX_SK_LAUNCH(sketch_wave_kernel_capped<16>, lds_small, first); else SKX_SK_LAUNCH(SKX_SK_SMALL(16), lds_small, first);
            } else {
                if (capped) SKX_SK_LAUNCH(sketch_wave_kernel_capped<0>, lds_small, first); else SKX_SK_LAUNCH(SKX_SK_SMALL(0), lds_small, first);
            }
        }
        if (phase & 2) {
            if (first == 2u)
                hipLaunchKernelGGL(sketch_merge_kernel, dim3(512), dim3(64), (size_t)kSketchCap * 8, st, offsets, s, max_ref, out_sk, sk_stride,
                                   out_len, out_cnt_in, big, filt, filt_shift, chk, lr, pool_cap, pool_fixed);
            if (k == 16) SKX_SK_LAUNCH(SKX_SK(16, true), lds, 1u); else SKX_SK_LAUNCH(SKX_SK(0, true), lds, 1u);
        }
    } else {
        if (k == 16) SKX_SK_LAUNCH(SKX_SK(16, false), lds, 0u); else SKX_SK_LAUNCH(SKX_SK(0, false), lds, 0u);
    }
#undef SKX_SK_LAUNCH
#undef SKX_BLK
#undef SKX_SK_SMALL
#undef SKX_SK
    return hipGetLastError();
}

// The reads launch_sketch left on the `big` list (n_big of them; the caller read the count back).  A separate call because
// the kernel needs 135 KB of LDS per block: queued unconditionally it waits for a drained CU even when the list is empty
// (40-80 us on the sketch stream per push, measured) -- and it almost always is.
hipError_t launch_sketch_block(hipStream_t st, const uint8_t* bases, const u64* offsets, const u32* big, u32 n_big, u32 k, u64 seed,
                               u32 s, u64 max_ref, bool inrange_only, u64* out_sk, u32 sk_stride, u32* out_len, u32* out_cnt_in,
                               const u64* filt, u32 filt_shift, bool packed, u32* chk, u32 pool_cap, u32 pool_fixed) {
    if (n_big == 0) return hipSuccess;
    const dim3 blk_grid(std::min<u32>(n_big, 256u));
#define SKX_BLK_LAUNCH(KERNEL)                                                                                              \
    hipLaunchKernelGGL((KERNEL), blk_grid, dim3(1024), kBigLds, st, bases, offsets, big, n_big, k, seed, s, max_ref, out_sk, \
                       sk_stride, out_len, out_cnt_in, filt, filt_shift, packed ? 1u : 0u, chk, pool_cap, pool_fixed)
    if (k == 16) { if (inrange_only) SKX_BLK_LAUNCH((sketch_block_kernel<16, true>)); else SKX_BLK_LAUNCH((sketch_block_kernel<16, false>)); }
    else { if (inrange_only) SKX_BLK_LAUNCH((sketch_block_kernel<0, true>)); else SKX_BLK_LAUNCH((sketch_block_kernel<0, false>)); }
#undef SKX_BLK_LAUNCH
    return hipGetLastError();
}

void launch_kmer_filter_build(hipStream_t st, u64 seed, u64 max_ref, const u64* filt, u32 filt_shift, u32* n_keys, u32* words, u32 shift) {
    if (words) hipLaunchKernelGGL((kmer_filter_build_kernel<true>), dim3(65536), dim3(256), 0, st, seed, max_ref, filt, filt_shift, n_keys, words, shift);
    else hipLaunchKernelGGL((kmer_filter_build_kernel<false>), dim3(65536), dim3(256), 0, st, seed, max_ref, filt, filt_shift, n_keys, words, shift);
}
void launch_count_scan(hipStream_t st, const u32* in, u32* out, u32 n, u32* bsum) {
    if (n == 0) return;
    const u32 nb = cdiv(n, 1024);
    hipLaunchKernelGGL(count_scan_a_kernel, dim3(nb), dim3(256), 0, st, in, out, n, bsum);
    if (nb > 1) hipLaunchKernelGGL(count_scan_b_kernel, dim3(nb), dim3(256), 0, st, out, n, bsum);
}

void launch_dict_insert(hipStream_t st, const u64* sk, u32 sk_stride, const u32* poff, u32 r_begin, u32 r_end, u32 p_base,
                        u64* pair_h, u32* pair_r, u64* ht, u32 ht_slots, u32* ctr, u32 pair_cap, const u32* row_off,
                        PairBase base) {
    if (r_end <= r_begin) return;
    hipLaunchKernelGGL(dict_insert_kernel, dim3(cdiv(r_end - r_begin, 256)), dim3(256), 0, st, sk, sk_stride, poff, r_begin,
                       r_end, p_base, pair_h, pair_r, ht, ht_slots - 1u, ctr, pair_cap, row_off, base);
}
static u32 dict_bshift(u64 max_ref) {
    const u32 bits = 64u - (u32)__builtin_clzll(max_ref | 1ull);
    return bits > 17u ? bits - 17u : 0u;  // hashes <= max_ref  =>  hash >> bshift < 2^17
}
void launch_dict_rest(hipStream_t st, u64* ht, u32 ht_slots, u64 max_ref, u32* slot_off, u32* bcount, u32* bbase, u32* btot,
                      u32* ctr, u64* q, u32* n_q) {
    const u32 bshift = dict_bshift(max_ref);
    const u32 walk = std::min<u32>(cdiv(ht_slots, 256), 8192u);
    hipLaunchKernelGGL(dict_count_kernel, dim3(walk), dim3(256), 0, st, ht, ht_slots, bshift, slot_off, bcount);
    hipLaunchKernelGGL(dict_scan_a_kernel, dim3(kDictBuckets / 1024u), dim3(256), 0, st, bcount, bbase, btot);
    hipLaunchKernelGGL(dict_scan_b_kernel, dim3(1), dim3(128), 0, st, btot, ctr, n_q);
    hipLaunchKernelGGL(dict_scatter_kernel, dim3(walk), dim3(256), 0, st, ht, ht_slots, bshift, slot_off, bbase, btot, ctr, q);
    hipLaunchKernelGGL(dict_bucket_sort_kernel, dim3(kDictBuckets / 256), dim3(256), 0, st, q, bbase, btot, ctr);
}
u32 dict_buckets() { return kDictBuckets; }
void launch_pair_q(hipStream_t st, const u64* pair_h, u32 n_pairs, const u64* q, const u32* n_q, u32* pair_q, const u32* qrow, const u32* bbase,
                   const u32* btot, u64 max_ref) {
    if (n_pairs == 0) return;
    hipLaunchKernelGGL(pair_q_kernel, dim3(cdiv(n_pairs, 256)), dim3(256), 0, st, pair_h, n_pairs, q, n_q, pair_q, qrow, bbase, btot, dict_bshift(max_ref));
}
void launch_rare_count(hipStream_t st, const u64* mat, u64 n_elems, u64* key, u32* cnt, u32 mask, u32* overflow, u64* spmask, const u32* grp_sp, u32 s) {
    if (n_elems == 0) return;
    hipLaunchKernelGGL(rare_count_kernel, dim3((u32)std::min<u64>((n_elems + 255) / 256, 1u << 16)), dim3(256), 0, st, mat, n_elems, key, cnt, mask, overflow,
                       spmask, grp_sp, (u64)s * kTileGenomes);
}
void launch_rare_fill(hipStream_t st, const u64* mat, u64 n_elems, u32 s, const u64* key, const u32* off, u32* cursor, u32* post, u32 mask) {
    if (n_elems == 0) return;
    hipLaunchKernelGGL(rare_fill_kernel, dim3((u32)std::min<u64>((n_elems + 255) / 256, 1u << 16)), dim3(256), 0, st, mat, n_elems, s, key, off, cursor, post, mask);
}
void launch_collect_dense(hipStream_t st, const u64* key, const u32* off, u64 slots, u64* out, u32* out_slot, u32* n, u32 cap) {
    hipLaunchKernelGGL(collect_dense_kernel, dim3((u32)std::min<u64>((slots + 255) / 256, 4096)), dim3(256), 0, st, key, off, slots, out, out_slot, n, cap);
}
void launch_window_seg(hipStream_t st, const u64* lo, const u64* hi, u32 n_bt, u32 n_tiles, const u64* q, const u32* seg, const u32* grp_sp, u32* win) {
    hipLaunchKernelGGL(window_seg_kernel, dim3(cdiv(n_bt, 256)), dim3(256), 0, st, lo, hi, n_bt, n_tiles, q, seg, grp_sp, win);
}
void launch_classify(hipStream_t st, const u64* q, const u32* n_q, u32 q_bound, const RareIndex& ri, u32* qinfo, u32* qloc, u32* bsum,
                     u64* qd, u32* n_d, u32* qrow, u32* sslot, u32* h_words) {
    const u32 nb = std::max(1u, cdiv(q_bound, 1024));
    hipLaunchKernelGGL(classify_a_kernel, dim3(nb), dim3(256), 0, st, q, n_q, ri, qinfo, qloc, bsum);
    hipLaunchKernelGGL(classify_b_kernel, dim3(1), dim3(1024), 0, st, bsum, n_q, n_d, h_words, ri.qs ? ri.n_sd : kNoStatic);
    hipLaunchKernelGGL(classify_c_kernel, dim3(std::max(1u, cdiv(q_bound, 256))), dim3(256), 0, st, q, n_q, qinfo, qloc, bsum, n_d, qd, qrow, sslot, ri);
}
void launch_sparse_fill(hipStream_t st, const u32* sslot, const u32* n_d, const RareIndex& ri, u64* m_bits, u32 n_pad, u32* m_dirty, u32 rows_bound,
                        const u32* only_if) {
    const u32 blocks = std::max(1u, std::min(cdiv(rows_bound, 256), kWalkBlocks));
    hipLaunchKernelGGL(sparse_fill_kernel, dim3(blocks), dim3(256), 0, st, sslot, n_d, ri, m_bits, n_pad, m_dirty, only_if);
}
void launch_rare_to_mq(hipStream_t st, const u32* sslot, const u32* n_d, const RareIndex& ri, u64* mq, u32 nq_rows, u32 n_pad, u64* rowany,
                       u32* grp_any, u32 rows_bound, const u32* only_if, u32* ext) {
    const u32 blocks = std::max(1u, std::min(cdiv(rows_bound, 64), 1024u));
    const u32 chunk = std::min(kRareGrpChunk, std::max(1u, (u32)env_int("SKX_RARE_CHUNK", (int)kRareGrpChunk)));  // experiment knob: groups per turn
    static const bool sparse_off = env_int("SKX_RARE_SPARSE_WRITES", 1) == 0;  // experiment knob: 0 = every word of every row, as until round 6
    hipLaunchKernelGGL(rare_to_mq_kernel, dim3(blocks), dim3(256), 0, st, sslot, n_d, ri, mq, nq_rows, n_pad / 64, rowany, grp_any, only_if, chunk,
                       sparse_off ? nullptr : ext);
    if (ext) hipLaunchKernelGGL(mq_extent_kernel, dim3(1), dim3(1), 0, st, ext, nq_rows, n_d, only_if);
}
u32 mq_any_stride() { return kAnyStride; }
void launch_nd_from_nq(hipStream_t st, const u32* n_q, u32* n_d, u32* h_words) {
    hipLaunchKernelGGL(nd_from_nq_kernel, dim3(1), dim3(1), 0, st, n_q, n_d, h_words);
}
void launch_pass_hist(hipStream_t st, const u32* pair_q, const PassBatches& pb, u32* cnt, u32 row_stride) {
    const u32 n = pb.p_off[pb.n];
    if (n == 0) return;
    hipLaunchKernelGGL(pass_hist_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, pair_q, pb, cnt, row_stride);
}
void launch_gain_dense(hipStream_t st, const u64* m_bits, const u64* m_int, u32 n_pad, const u32* n_d, u32 rows_bound, const u32* cnt, u32 row_stride,
                       u32 n_b, u32* gain, const u32* segw, const u32* grp_sp) {
    const dim3 grid(n_pad / 256, std::max(1u, cdiv(cdiv(rows_bound, 64), kGainWords)));
#define SKX_GAIN(NB) hipLaunchKernelGGL(gain_dense_kernel<NB>, grid, dim3(256), 0, st, m_bits, m_int, n_pad, n_d, cnt, row_stride, gain, segw, grp_sp)
    switch (n_b) {
        case 1: SKX_GAIN(1); break; case 2: SKX_GAIN(2); break; case 3: SKX_GAIN(3); break; case 4: SKX_GAIN(4); break;
        case 5: SKX_GAIN(5); break; case 6: SKX_GAIN(6); break; case 7: SKX_GAIN(7); break; default: SKX_GAIN(8); break;
    }
#undef SKX_GAIN
}
// (the walks over the genome lists are bound by the latency of their dependent loads and atomics, not by their waves: a grid that fills
// the chip only keeps the other streams' kernels -- the next group's sketches -- out of the wave slots: 56 M reads/s with 1832 blocks,
// measured.  walk_scale: the caller knows that nothing runs beside this pass)
void launch_gain_sparse(hipStream_t st, const u32* n_d, u32 rows_bound, const u32* cnt, u32 row_stride, u32 n_b, u32 n_pad, u32* gain_s,
                        const u32* sslot, const RareIndex& ri, const LongRows* lr, u32 walk_scale, const PatRows* pr) {
    hipLaunchKernelGGL(gain_sparse_kernel, dim3(std::max(1u, std::min(cdiv(rows_bound, 256), kWalkBlocks * std::max(1u, walk_scale)))), dim3(256), 0, st,
                       sslot, n_d, ri, cnt, row_stride, n_b, n_pad, gain_s, lr ? *lr : LongRows{nullptr, nullptr, 0u},
                       (pr && lr && ri.prec) ? *pr : PatRows{nullptr, 0u, nullptr, nullptr});
}
void launch_list_sig(hipStream_t st, const u32* lslot, u32 n_long, const u32* off, const u32* cnt, const u32* post, u32* sig, u64* content) {
    if (n_long == 0) return;
    hipLaunchKernelGGL(list_sig_kernel, dim3(cdiv(n_long, 4)), dim3(256), 0, st, lslot, n_long, off, cnt, post, sig, content);
}
void launch_pat_exceptions(hipStream_t st, const u64* mlong, u32 n_gw, u32 n_long, const u32* pat_of, const u32* pat_rep, u32* prec, u32* n_done) {
    if (n_long == 0) return;
    hipLaunchKernelGGL(pat_exceptions_kernel, dim3(cdiv(n_long, 4)), dim3(256), 0, st, mlong, n_gw, n_long, pat_of, pat_rep, prec, n_done);
}
void launch_pat_matrix(hipStream_t st, const u64* mlong, const u32* pat_rep, u32 n_pat, u32 n_gw, u64* pm, u32 n_pad) {
    if (n_pat == 0) return;
    hipLaunchKernelGGL(pat_matrix_kernel, dim3(cdiv(n_pat, 64), cdiv(n_gw, 4)), dim3(256), 0, st, mlong, pat_rep, n_pat, n_gw, pm, n_pad);
}
u32 pat_record_words() { return kPatRec; }
u32 pat_words_max() { return kPatWords; }
void launch_pat_nqc_init(hipStream_t st, u32* nqc, u32 v) { hipLaunchKernelGGL(pat_nqc_init_kernel, dim3(1), dim3(64), 0, st, nqc, v); }
void launch_cand_pat_rows(hipStream_t st, const RareIndex& ri, const u32* n_d, const u32* candmask, const u32* candslot, u32 n_pad, u32* bad, u32 n_b,
                          u64* pcw, u64* mqc, size_t mqc_stride, u32 rows_c, u64* rowany_c, u32 rowany_stride, u32* grp_any_c, u32 n_grp_c) {
    if (ri.n_pat == 0) return;
    hipLaunchKernelGGL(cand_pat_rows_kernel, dim3(cdiv(ri.n_pat, 4), n_b), dim3(256), 0, st, ri, n_d, candmask, candslot, n_pad, bad, pcw, mqc, mqc_stride,
                       rows_c, rowany_c, rowany_stride, grp_any_c, n_grp_c);
}
void launch_cand_pat_map(hipStream_t st, const LongRows& lr, const PatRows& pr, const RareIndex& ri, const u32* n_d, const u32* candmask,
                         const u32* candslot, u32 n_pad, u32* bad, u32 n_b, u32* nqc, u32* smap, u32 smap_stride, const u64* pcw, u64* mqc,
                         size_t mqc_stride, u32 rows_c, u64* rowany_c, u32 rowany_stride, u32* grp_any_c, u32 n_grp_c, u32 rows_bound, u32 walk_scale) {
    const u32 blocks = std::max(1u, std::min(cdiv(rows_bound, 256), 64u * std::max(1u, walk_scale)));
    hipLaunchKernelGGL(cand_pat_map_kernel, dim3(blocks, n_b), dim3(256), 0, st, lr, pr, ri, n_d, candmask, candslot, n_pad, bad, nqc, smap, smap_stride,
                       pcw, mqc, mqc_stride, rows_c, rowany_c, rowany_stride, grp_any_c, n_grp_c);
}
u32 gain_sparse_stride() { return kGainSparseStride; }
u32 pass_counter_bytes() { return kPassBatchesMax * kCtrStride * 4u; }
void launch_mlong_build(hipStream_t st, const u32* lslot, u32 n_long, const u32* off, const u32* cnt, const u32* post, u64* mlong, u32 n_gw) {
    if (n_long == 0) return;
    hipLaunchKernelGGL(mlong_build_kernel, dim3(cdiv(n_long, 4)), dim3(256), 0, st, lslot, n_long, off, cnt, post, mlong, n_gw);
}
void launch_long_rows(hipStream_t st, const u32* sslot, const u32* n_d, u32 rows_bound, const u32* cnt, u32 row_stride, u32 n_b, const LongRows& lr) {
    hipLaunchKernelGGL(long_rows_kernel, dim3(std::max(1u, std::min(cdiv(rows_bound, 256), (u32)env_int("SKX_G_LONGROWS", 1024)))), dim3(256), 0, st, sslot, n_d, cnt, row_stride, n_b, lr);
}
void launch_gain_long(hipStream_t st, const LongRows& lr, const RareIndex& ri, const u32* n_d, const u32* cnt, u32 row_stride, u32 n_b, u32 n_pad,
                      u32* gain_l, u32 walk_scale) {
    // (z: the batch's rows are split over this many workgroups -- 8 beside the other streams' kernels; alone on the chip 32 read the bit
    // rows a third faster: one pass per batch 39 -> 47 M reads/s)
    hipLaunchKernelGGL(gain_long_kernel, dim3(cdiv(ri.n_gw, 64), n_b, (u32)env_int("SKX_G_GAINLONG", 8) * std::min(4u, std::max(1u, walk_scale))), dim3(256), 0, st, lr, ri, n_d, cnt, row_stride, n_pad, gain_l);
}
void launch_mlong_transpose(hipStream_t st, const u64* mlong, u32 n_long, u32 n_gw, u64* mlongT, u32 n_lw) {
    if (n_long == 0) return;
    hipLaunchKernelGGL(mlong_transpose_kernel, dim3(n_lw, cdiv(n_gw, 4)), dim3(256), 0, st, mlong, n_long, n_gw, mlongT, n_lw);
}
void launch_cand_hit(hipStream_t st, const u32* cand, u32 n_pad_c, u32 n_b, const u32* bad, const RareIndex& ri, const u64* inb, u64* hit) {
    hipLaunchKernelGGL(cand_hit_kernel, dim3(cdiv(n_pad_c, 4), n_b), dim3(256), 0, st, cand, n_pad_c, bad, ri, inb, hit);
}
void launch_cand_words(hipStream_t st, const u32* cand, u32 n_pad_c, u32 n_b, u32 n_gw, u64* cw, u32* cbase, u32* cwl, u32* ncwl) {
    hipLaunchKernelGGL(cand_words_kernel, dim3(n_b), dim3(1024), 0, st, cand, n_pad_c, n_gw, cw, cbase, cwl, ncwl);
}
void launch_cand_long(hipStream_t st, const LongRows& lr, const RareIndex& ri, const u32* n_d, const u64* cw, const u32* cbase, const u32* cwl,
                      const u32* ncwl, u32 n_pad_c, u32* bad, u32 n_b, u32* nqc, u32* smap, u32 smap_stride, u64* mqc, size_t mqc_stride,
                      u32 rows_c, u64* rowany_c, u32 rowany_stride, u32* grp_any_c, u32 n_grp_c, u32 walk_scale, const u64* hit) {
    hipLaunchKernelGGL(cand_long_kernel, dim3((u32)env_int("SKX_G_CANDLONG", 192) * std::min(4u, std::max(1u, walk_scale)), n_b), dim3(256), 0, st, lr, ri, n_d, cw, cbase, cwl, ncwl, n_pad_c, bad, nqc, smap, smap_stride, mqc,
                       mqc_stride, rows_c, rowany_c, rowany_stride, grp_any_c, n_grp_c, hit);
}
void launch_pass_tables(hipStream_t st, const u64* prev, const u32* gain, const u32* gain_s, const u32* gain_l, u32 n_b, u32 n_pad, u64* tab) {
    hipLaunchKernelGGL(pass_tables_kernel, dim3(cdiv(n_pad, 256)), dim3(256), 0, st, prev, gain, gain_s, gain_l, n_b, n_pad, tab);
}
void launch_cand_select(hipStream_t st, const u64* tab, u32 n_pad, const Species& sp, u32 n_b, u32 top_k, u32 cap, u32* cand, u32* candslot,
                        u64* tabc, u32* ncand, u32* bad, u32* candmask) {
    hipLaunchKernelGGL(cand_select_kernel, dim3(n_b * sp.n_sp), dim3(1024), 0, st, tab, n_pad, sp, top_k, cap, cand, candslot, tabc, ncand, bad, candmask);
}
void launch_cand_count(hipStream_t st, const u64* t0, const u64* t1, const Species& sp, u32 top_k, u32* h_out, u32 seq) {
    hipLaunchKernelGGL(cand_count_kernel, dim3(1), dim3(1024), 0, st, t0, t1, sp, top_k, nullptr, h_out, seq);
}
void launch_cand_gather_m(hipStream_t st, const u64* m_bits, const u64* m_int, u32 n_pad, const u32* n_d, u32 rows_bound, const u32* cand, u32 n_pad_c,
                          const u32* bad, u32 n_b, u64* mc, u32 words_c) {
    const u32 words = std::min(words_c, std::max(1u, cdiv(rows_bound, 64)));
    hipLaunchKernelGGL(cand_gather_m_kernel, dim3(cdiv(n_pad_c, 256), words, n_b), dim3(256), 0, st, m_bits, m_int, n_pad, n_d, cand, n_pad_c, bad, mc, words_c);
}
void launch_cand_sparse(hipStream_t st, const u32* sslot, const u32* n_d, u32 rows_bound, const RareIndex& ri, const u32* candmask,
                        const u32* candslot, u32 n_pad, u32* bad, u32 n_b, u32* nqc, u32* smap, u32 smap_stride, u64* mqc, size_t mqc_stride,
                        u32 rows_c, u64* rowany_c, u32 rowany_stride, u32* grp_any_c, u32 n_grp_c, u32 walk_scale) {
    hipLaunchKernelGGL(cand_sparse_kernel, dim3(std::max(1u, std::min(cdiv(rows_bound, 256), kWalkBlocks * std::max(1u, walk_scale)))), dim3(256), 0, st, sslot, n_d, ri, candmask,
                       candslot, n_pad, bad, n_b, nqc, smap, smap_stride, mqc, mqc_stride, rows_c, rowany_c, rowany_stride, grp_any_c, n_grp_c);
}
void launch_cand_publish(hipStream_t st, const u32* bad, u32 force_full, const u32* ncand, const u32* nqc, const u32* n_d, u32 n_b, u32 n_sp,
                         u32 rows_c, u32* mode, u32* any_full, u32* nqc_total, u32* h_pub, u32 seq) {
    hipLaunchKernelGGL(cand_publish_kernel, dim3(1), dim3(1), 0, st, bad, force_full, ncand, nqc, n_d, n_b, n_sp, rows_c, mode, any_full, nqc_total,
                       h_pub, seq);
}
void launch_m_clear(hipStream_t st, u64* m_bits, u64* m_int, u32 n_pad, const u32* n_d, const u32* any_full) {
    hipLaunchKernelGGL(m_clear_kernel, dim3(512), dim3(256), 0, st, m_bits, m_int, n_pad, n_d, any_full);
}
void launch_cand_pair_rows(hipStream_t st, const u32* pair_q, u32 n_pairs, const u32* n_d, const u32* smap, u32 rows_c, u32* pair_qc) {
    if (n_pairs == 0) return;
    hipLaunchKernelGGL(cand_pair_rows_kernel, dim3(cdiv(n_pairs, 256)), dim3(256), 0, st, pair_q, n_pairs, n_d, smap, rows_c, pair_qc);
}
void launch_cand_rows_back(hipStream_t st, u32* out_idx, u32 n_reads, u32 n_sp, u32 top_k, const u32* cand, u32 cap, const u32* g0) {
    const u32 n = n_reads * n_sp * top_k;
    if (n == 0) return;
    hipLaunchKernelGGL(cand_rows_back_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, out_idx, n, n_sp, top_k, cand, cap, g0);
}
void launch_window(hipStream_t st, const u64* lo, const u64* hi, u32 n_bt, const u64* q, const u32* n_q, u32* win, u32* h_nq) {
    hipLaunchKernelGGL(window_kernel, dim3(cdiv(n_bt, 256)), dim3(256), 0, st, lo, hi, n_bt, q, n_q, win, h_nq);
}
void launch_exceptions(hipStream_t st, const u32* exc_g, const u64* exc_h, u32 n_exc, const u64* q, const u32* n_q,
                       u64* m_bits, u32 n_pad, u32* m_dirty, const u32* qrow, u32 row0, u32 n_fixed) {
    if (n_exc == 0) return;
    hipLaunchKernelGGL(exceptions_kernel, dim3(cdiv(n_exc, 256)), dim3(256), 0, st, exc_g, exc_h, n_exc, q, n_q,
                       m_bits, n_pad, m_dirty, qrow, row0, n_fixed);
}

bool scan_lean_wants_slabs();
bool scan_lean_applies(u32 n_bands, bool split, bool big_table) {
    static const int lean_env = env_int("SKX_SCAN_LEAN", 1);
    // (the slab form needs n_bands <= kWordBandsMax for its word -> bands lists; results into M need no list)
    return lean_env && !split && !big_table && (!scan_lean_wants_slabs() || n_bands <= kWordBandsMax);
}
u32 scan_lean_words() { return kLeanWords; }
// The lean kernel puts its results straight into M (atomicOr at the end of a block, no slabs; the transpose then reads M only and
// zeroes it again).  Round 3 did so only while the pass's whole M was smaller than the infinity cache (C2: 52-100 MB) and kept
// per-block slabs for C4 (378 MB).  Round 4: the size of M does not matter -- the launch walks the matrix band by band, and a band's
// blocks touch the same handful of word rows of M (a few MB, whatever the dictionary's size), which are written back once when
// the launch has moved on: measured at C4 (tools/ab.sh, same box, twice each) slabs 45.8-46.2 M reads/s, scan alone 0.62-0.63
// of the 8 TB/s peak; into M 47.4 M, 0.66-0.67, a lone batch +4 %, and no slabs to allocate (0.94 GB per stream at C4; the counter
// WRITE_SIZE still sees 1.2 GB per launch -- now the atomicOr requests of the ~3 bands that reach every word).  The slab form survives in
// the experiments build only (SKX_SCAN_NT = 0..3, SKX_SCAN_ABLATE), which then also allocates the slabs.
bool scan_lean_wants_slabs() {
#ifdef SKX_EXPERIMENTS
    static const int nt = env_int("SKX_SCAN_NT", -1);  // experiment knob: bit 2 = results into M, 0..3 = slabs
    static const int ablate = env_int("SKX_SCAN_ABLATE", 0);
    return ablate != 0 || (nt >= 0 && (nt & 4) == 0);
#else
    return false;
#endif
}
bool scan_lean_into_m() { return !scan_lean_wants_slabs(); }
void launch_word_bands(hipStream_t st, u32* win, u32 n_tiles, u32 n_bands, const u32* n_q, u32* wb, const u64* lo, const u64* hi,
                       const u64* q, u32* h_nq) {
    hipLaunchKernelGGL(word_bands_kernel, dim3(n_tiles), dim3(256), 0, st, win, n_tiles, n_bands, n_q, wb, lo, hi, q, h_nq);
}

u32 scan_run_cap() { return kRunCap; }
u32 scan_lean_cap(bool big) { return big ? kLeanCapBig : kLeanCap; }
void launch_scan(hipStream_t st, const u64* mat, u32 s, u32 n_tiles, u32 rb, u32 n_bands, const u64* q, const u32* win,
                 u64* m_bits, u64* m_int /* NULL = everything atomically into m_bits */, u32 n_pad, bool big_table,
                 bool lean /* sparse dictionaries: scan_lean_kernel */, u64* hbuf /* its slabs (experiments build; NULL: results into M) */,
                 u32* m_dirty, bool into_m /* lean kernel: results by atomicOr into m_bits instead of slabs (scan_lean_into_m) */,
                 u32 run /* > 0: scan_run_kernel, one workgroup per `run` consecutive bands of a tile (results into m_bits) */,
                 bool big_slices /* lean kernel into M: the instance for slices of up to scan_lean_cap(true) entries */) {
    dim3 grid(n_tiles * n_bands), block(256);
#ifdef SKX_EXPERIMENTS
    // (experiments that lost -- the runs-of-bands kernel, the lean kernel's BIG instance -- are not compiled into the product library)
    if (run) {
        const dim3 rgrid(n_tiles * cdiv(n_bands, run));
        hipLaunchKernelGGL((scan_run_kernel<2>), rgrid, block, 0, st, mat, s, n_tiles, rb, n_bands, run, q, win, m_bits, n_pad, m_dirty, 1u);
        return;
    }
    if (lean && (into_m || !hbuf) && big_slices) {
        hipLaunchKernelGGL((scan_lean_kernel<0, 6, true>), grid, block, 0, st, mat, s, n_tiles, rb, q, win, m_bits, n_pad, hbuf, m_dirty, 1u);
        return;
    }
    // profiling aids (results invalid unless 0): the ablated kernels are not even compiled into the product library
    static const int ablate = env_int("SKX_SCAN_ABLATE", 0);
    static const int nt_env = env_int("SKX_SCAN_NT", -1);  // 1 = non-temporal slab stores, 2 = matrix loads, 3 = both, +4 = results into M
    const int nt = nt_env > 0 ? nt_env : 0;
    static const u32 prio = (u32)env_int("SKX_SCAN_PRIO", 1);
    if (lean && hbuf && (ablate || nt)) {
#define SKX_SCAN_L(A, N) hipLaunchKernelGGL((scan_lean_kernel<A, N>), grid, block, 0, st, mat, s, n_tiles, rb, q, win, m_bits, n_pad, hbuf, m_dirty, prio)
        if (ablate == 1) SKX_SCAN_L(1, 0); else if (ablate == 2) SKX_SCAN_L(2, 0); else if (ablate == 3) SKX_SCAN_L(3, 0);
        else if (nt == 1) SKX_SCAN_L(0, 1); else if (nt == 2) SKX_SCAN_L(0, 2); else if (nt == 3) SKX_SCAN_L(0, 3);
        else if (nt == 4) SKX_SCAN_L(0, 4); else if (nt == 5) SKX_SCAN_L(0, 5); else if (nt == 6) SKX_SCAN_L(0, 6); else SKX_SCAN_L(0, 7);
#undef SKX_SCAN_L
        return;
    }
    if (!lean && ablate) {
#define SKX_SCAN_A(CAP, A, SP) hipLaunchKernelGGL((scan_kernel<CAP, A, SP>), grid, block, 0, st, mat, s, n_tiles, rb, q, win, m_bits, m_int, n_pad)
        if (big_table && m_int) { if (ablate == 1) SKX_SCAN_A(4088, 1, true); else if (ablate == 2) SKX_SCAN_A(4088, 2, true); else SKX_SCAN_A(4088, 3, true); }
        else if (ablate == 1) SKX_SCAN_A(2040, 1, false); else if (ablate == 2) SKX_SCAN_A(2040, 2, false); else SKX_SCAN_A(2040, 3, false);
#undef SKX_SCAN_A
        return;
    }
    if (lean && !into_m && hbuf) {  // round 3's slab form
        hipLaunchKernelGGL((scan_lean_kernel<0, 0>), grid, block, 0, st, mat, s, n_tiles, rb, q, win, m_bits, n_pad, hbuf, m_dirty, prio);
        return;
    }
#else
    const u32 prio = 1u;
    (void)run; (void)big_slices; (void)into_m;
#endif
    // sparse dictionaries (the host asked for neither the split nor the big-table variant): the lean probe, results into M
    if (lean) {
        hipLaunchKernelGGL((scan_lean_kernel<0, 6>), grid, block, 0, st, mat, s, n_tiles, rb, q, win, m_bits, n_pad, hbuf, m_dirty, prio);
        return;
    }
    // dense passes: 2040-entry slices, interior words by plain stores when the caller has the second array; very dense:
    // 4088-entry slices (40 KB of LDS, 3 blocks per CU) instead of re-streaming the band
    if (big_table && m_int) hipLaunchKernelGGL((scan_kernel<4088, 0, true>), grid, block, 0, st, mat, s, n_tiles, rb, q, win, m_bits, m_int, n_pad);
    else if (m_int) hipLaunchKernelGGL((scan_kernel<2040, 0, true>), grid, block, 0, st, mat, s, n_tiles, rb, q, win, m_bits, m_int, n_pad);
    else hipLaunchKernelGGL((scan_kernel<2040, 0, false>), grid, block, 0, st, mat, s, n_tiles, rb, q, win, m_bits, m_int, n_pad);
}
void launch_transpose_bits(hipStream_t st, u64* m_bits, u64* m_int, u32 n_pad, u32 n_words, u64* mq, const u32* n_q, u32* grp_any,
                           const u64* hbuf, const u32* wb, const u32* win, u32 n_tiles, const u32* m_dirty, u64 nq_est, u64* rowany,
                           const u32* only_if, bool keep_m) {
    if (n_words == 0) return;
    const u32 n_gw = n_pad / 64;
    // y extent: twice the estimated dictionary size (the blocks stride, see the kernel), at most what the pairs allow
    const u32 y_all = cdiv(n_words, kWordsPerBlock);
    const u32 y_est = (u32)std::min<u64>(y_all, std::max<u64>(16, cdiv((u32)std::min<u64>(2 * nq_est / 64 + 1, 0xFFFFFFF0u), kWordsPerBlock)));
    hipLaunchKernelGGL(transpose_bits_kernel, dim3(cdiv(n_gw, kRankWords), std::min(y_est, 65535u)), dim3(256), 0, st,
                       m_bits, m_int, n_pad, n_words, mq, n_gw, n_q, grp_any, hbuf, wb, win, n_tiles, m_dirty, rowany, only_if, keep_m ? 1u : 0u);
}
void launch_batch_check(hipStream_t st, const u64* offsets, u32 n_reads, u64 n_bases, u32* chk, u32* cnt_tail, const LongReads* long_reads) {
    const LongReads lr = long_reads ? *long_reads : LongReads{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0u};
    hipLaunchKernelGGL(batch_check_kernel, dim3(std::min<u32>(cdiv(n_reads, 256), 1024u)), dim3(256), 0, st, offsets, n_reads, n_bases, chk,
                       cnt_tail, lr);
}
u32 chk_words() { return kChkWords; }
u32 pool_row_fixed() { return kRowFixed; }
u32 long_read_split() { return kLongSplit; }
u32 long_read_seg_slots() { return kSegSlots; }
void launch_store_host_words(hipStream_t st, u32* h_dst, u32* d_src, u32 n) {
    hipLaunchKernelGGL(store_host_words_kernel, dim3(1), dim3(64), 0, st, h_dst, d_src, n);
}
void launch_publish(hipStream_t st, u32* chk, u32* retry, u32* big, const u32* total_pairs, u32* h_pub, u32 seq, const u32* dict_ctr) {
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(1), 0, st, chk, retry, big, total_pairs, h_pub, seq, dict_ctr);
}
void launch_filter_build(hipStream_t st, const u64* vals, u64 n, u32 shift, u64* words, bool markers_are_values, unsigned long long* count) {
    if (n == 0) return;
    const u32 blocks = (u32)std::min<u64>((n + 255) / 256, 1u << 16);
    hipLaunchKernelGGL(filter_build_kernel, dim3(blocks), dim3(256), 0, st, vals, n, shift, words, markers_are_values, count);
}
void launch_filter_apply(hipStream_t st, u64* sk, u32 sk_stride, u32* cnt, u32 n_reads, const u64* bits, u32 shift) {
    if (n_reads == 0) return;
    hipLaunchKernelGGL(filter_apply_kernel, dim3(cdiv(n_reads, 4)), dim3(256), 0, st, sk, sk_stride, cnt, n_reads, bits, shift);
}
void launch_seg_sum(hipStream_t st, const u32* pair_q, const u32* poff, u32 p_base, u32 r_begin, u32 n_reads,
                    u32 seg_len, const u64* mq, u32 n_pad, u32 nq_rows, u32* inc, const u32* grp_any,
                    u32* qsum /* [ceil(n_seg / 16)][n_pad]: the chunk sums, zero on entry; NULL: not wanted */, const u64* rowany,
                    const u32* n_q, const Species& sp, const u64* gmax /* != NULL: only (chunk, group)s that can hold a candidate */,
                    const u64* lead_val) {
    const u32 n_gw = n_pad / 64, n_seg = cdiv(n_reads, seg_len);
    const u32 n_grp = cdiv(n_gw, kRankWords);
    ChunkLive cl;
    cl.gmax = gmax; cl.lead_val = lead_val; cl.n_half = n_pad / 256;
    // 8 XCDs x ceil(groups / 8) groups each x ceil(n_seg / 4) workgroups of 4 waves (= 4 consecutive segments)
    hipLaunchKernelGGL(seg_sum_kernel<false>, dim3(8u * cdiv(n_grp, 8) * cdiv(n_seg, 4)), dim3(256), 0, st, pair_q, poff, p_base,
                       r_begin, n_reads, seg_len, mq, n_gw, n_pad, nq_rows, inc, grp_any, qsum, rowany, n_q, cl, sp);
}
// chunk sums only (csum_raw[chunk][g], zero on entry), one workgroup per (rank group, chunk of 16 segments)
void launch_chunk_sum(hipStream_t st, const u32* pair_q, const u32* poff, u32 p_base, u32 r_begin, u32 n_reads,
                      const u64* mq, u32 n_pad, u32 nq_rows, const u32* grp_any, u32* csum_raw, const u64* rowany, const u32* n_q,
                      const Species& sp) {
    const u32 n_gw = n_pad / 64, quarter = 4u * kSegLen, n_quarters = cdiv(n_reads, quarter);
    const u32 n_grp = cdiv(n_gw, kRankWords);
    hipLaunchKernelGGL(seg_sum_kernel<true>, dim3(8u * cdiv(n_grp, 8) * cdiv(n_quarters, 4)), dim3(256), 0, st, pair_q, poff, p_base,
                       r_begin, n_reads, quarter, mq, n_gw, n_pad, nq_rows, nullptr, grp_any, csum_raw, rowany, n_q, ChunkLive(), sp);
}
// part: 0 = the whole chain; 1 = what needs only the chunk sums (prefix over the chunks, the table, per-chunk leaders and
// bounds); 2 = what needs the per-segment increments (per-segment bounds, start values, live flags)
void launch_seg_prefix(hipStream_t st, const u32* inc, u32 n_seg, u32 n_pad, const Species& sp, const u64* cum_in, u64* cum_out,
                       u32* rel, u32* csum /* [ceil(n_seg/16)][n_pad] scratch */, u32* csum_raw /* same size */, u32 prune_top_k, u32* leader,
                       u64* lead_val, u64* gmax, u64* part_sum, u32* part_idx, const u32* grp_any,
                       unsigned char* live /* [n_seg][n_pad / 64] or NULL: every start value is stored */,
                       u64* lead_seg /* [n_seg][n_sp] scratch (with live) */, int part, u32* live_ctr, hipEvent_t ev_table) {
    const u32 n_chunks = cdiv(n_seg, 16);
    dim3 grid(cdiv(n_pad, 256), n_chunks);
    if (part != 2) {
        hipLaunchKernelGGL(chunk_prefix_kernel, dim3(n_pad / 256), dim3(256), 0, st, csum_raw, csum, n_chunks, n_pad, cum_in, cum_out,
                           prune_top_k ? gmax : nullptr, n_pad / 256);
        if (ev_table) (void)hipEventRecord(ev_table, st);  // the table the next batch starts from exists from here on
        if (prune_top_k) {
            // who leads (per species) as each chunk of 16 segments begins (bound for the pruning), and which (chunk, group)s can matter
            hipLaunchKernelGGL(chunk_leader_part_kernel, dim3(n_chunks * sp.n_sp, kLeaderParts), dim3(256), 0, st, cum_in, csum, n_pad,
                               sp, prune_top_k, part_sum, part_idx);
            hipLaunchKernelGGL(chunk_leader_merge_kernel, dim3(n_chunks * sp.n_sp), dim3(64), 0, st, part_sum, part_idx, prune_top_k, leader,
                               lead_val, inc, csum, n_seg, n_pad, cum_in, sp, grp_any, live ? lead_seg : nullptr,
                               part == 1 ? 1u : 0u);  // (part 0: + seg_lead's work)
        }
    }
    if (part == 1) return;
    if (part == 2 && prune_top_k && live)
        hipLaunchKernelGGL(chunk_leader_merge_kernel, dim3(n_chunks * sp.n_sp), dim3(64), 0, st, part_sum, part_idx, prune_top_k, leader,
                           lead_val, inc, csum, n_seg, n_pad, cum_in, sp, grp_any, lead_seg, 2u);
    hipLaunchKernelGGL(seg_prefix_kernel, grid, dim3(256), 0, st, inc, csum, n_seg, n_pad, rel, prune_top_k ? gmax : nullptr,
                       lead_val, n_pad / 256, sp, grp_any, cum_in, prune_top_k ? live : nullptr, lead_seg,
                       prune_top_k ? live_ctr : nullptr);  // (live: the caller's choice, top-1 path only)
}
// (the first level of launch_seg_prefix's three: needs only the increments, not the running table -- queued with seg_sum)
void launch_rank_seg(hipStream_t st, const u32* pair_q, const u32* pair_r, const u32* poff, u32 p_base, u32 r_begin,
                     u32 n_reads, u32 seg_len, const u64* mq, u32 n_pad, u32 nq_rows, const Species& sp, const u64* cum_in,
                     const u32* rel, u32 top_k, u64* cand_sum, u32* cand_idx, const u32* grp_any) {
    const u32 n_gw = n_pad / 64, n_seg = cdiv(n_reads, seg_len);
    hipLaunchKernelGGL(rank_seg_kernel, dim3(cdiv((u64)n_seg * n_gw, 4)), dim3(256), 0, st, pair_q, pair_r, poff,
                       p_base, r_begin, n_reads, seg_len, mq, n_gw, n_pad, sp, cum_in, rel, top_k, cand_sum,
                       cand_idx, nq_rows, grp_any);
}
void launch_rank_seg_top1(hipStream_t st, const u32* pair_q, const u32* pair_r, const u32* poff, u32 p_base, u32 r_begin,
                          u32 n_reads, const u64* mq, u32 n_pad, u32 nq_rows, const Species& sp, const u64* cum_in,
                          const u32* rel, u64* best_sum, u32* best_idx, const u32* inc, const u32* leader, const u64* gmax,
                          const u64* lead_val, const u32* grp_any, const unsigned char* live,
                          unsigned char* has /* [segments][rank groups]: the kernel reported something for it */,
                          const u64* rowany, const u32* n_q) {
    const u32 n_gw = n_pad / 64, n_seg = cdiv(n_reads, 64), n_grp = cdiv(n_gw, kRankWords);
    hipLaunchKernelGGL(rank_seg_top1_kernel, dim3(cdiv((u64)n_seg * n_grp, 4)), dim3(256), 0, st, pair_q, pair_r, poff,
                       p_base, r_begin, n_reads, 64u, mq, n_gw, n_pad, sp, cum_in, rel, best_sum, best_idx, nq_rows, inc, leader,
                       gmax, lead_val, grp_any, live, has, rowany, n_q);
}
void launch_rank_seg_topk(hipStream_t st, const u32* pair_q, const u32* pair_r, const u32* poff, u32 p_base, u32 r_begin,
                          u32 n_reads, const u64* mq, u32 n_pad, u32 nq_rows, const Species& sp, const u64* cum_in,
                          const u32* rel, u32 top_k, u64* cand_sum, u32* cand_idx, const u32* inc, const u32* leader,
                          const u64* gmax, const u64* lead_val, const u32* grp_any, const unsigned char* live,
                          unsigned char* has) {
    const u32 n_gw = n_pad / 64, n_seg = cdiv(n_reads, 64), n_grp = cdiv(n_gw, kRankWords);
    hipLaunchKernelGGL(rank_seg_topk_kernel, dim3(cdiv((u64)n_seg * n_grp, 4)), dim3(256), 0, st, pair_q, pair_r, poff,
                       p_base, r_begin, n_reads, mq, n_gw, n_pad, sp, cum_in, rel, top_k, cand_sum, cand_idx, nq_rows,
                       inc, leader, gmax, lead_val, grp_any, live, has);
}
u32 rank_topk_fast_max() { return kTopkFast; }
u32 rank_leader_parts() { return kLeaderParts; }
void launch_top1_merge(hipStream_t st, const u64* best_sum, const u32* best_idx, u32 n_reads, u32* out_idx, u64* out_sum,
                       u32 out_r0, const Species& sp, const unsigned char* has, u32 n_grp) {
    if (n_reads == 0) return;
    hipLaunchKernelGGL(top1_merge_kernel, dim3(cdiv((u64)n_reads * sp.n_sp, 256)), dim3(256), 0, st, best_sum, best_idx, n_reads,
                       out_idx, out_sum, out_r0, sp, has, n_grp);
}
void launch_topk_merge(hipStream_t st, const u64* cand_sum, const u32* cand_idx, u32 n_reads, u32 n_units, u32 per_grp,
                       u32 top_k, u32* out_idx, u64* out_sum, u32 out_r0, const Species& sp, const unsigned char* has) {
    if (n_reads == 0) return;
    const u32 next_stride = (n_units + 3u) & ~3u;  // (a species owns at most all of the units)
    hipLaunchKernelGGL(topk_merge_kernel, dim3(cdiv((u64)n_reads * sp.n_sp, 4)), dim3(256), 4u * next_stride, st, cand_sum, cand_idx, n_reads,
                       n_units, per_grp, top_k, out_idx, out_sum, out_r0, sp, has, next_stride);
}
void launch_rank_table(hipStream_t st, const u64* cum, const Species& sp, u32 top_k, u32* out_idx, u64* out_sum) {
    hipLaunchKernelGGL(rank_table_kernel, dim3(sp.n_sp), dim3(1024), 0, st, cum, sp, top_k, out_idx, out_sum);
}
void launch_shared_debug(hipStream_t st, const u32* pair_q, const u32* poff, u32 p_base, u32 r_begin, u32 n_reads,
                         const u64* mq, u32 nq_rows, u32 n_real, const u32* real2pad, u32* shared, u32 out_r0) {
    if (n_reads == 0) return;
    hipLaunchKernelGGL(shared_debug_kernel, dim3(n_reads), dim3(256), 0, st, pair_q, poff, p_base, r_begin, mq, n_real, real2pad,
                       shared, out_r0, nq_rows);
}
void launch_add_table(hipStream_t st, u64* cum, const u64* add, u32 n_real, const u32* real2pad) {
    hipLaunchKernelGGL(add_table_kernel, dim3(cdiv(n_real, 256)), dim3(256), 0, st, cum, add, n_real, real2pad);
}
void launch_gather_table(hipStream_t st, const u64* cum, u64* out, u32 n_real, const u32* real2pad) {
    hipLaunchKernelGGL(gather_table_kernel, dim3(cdiv(n_real, 256)), dim3(256), 0, st, cum, out, n_real, real2pad);
}

}  // namespace skx

#ifdef SKX_EXPERIMENTS
namespace skx {
void rank_debug_counters(unsigned long long* out, bool reset) {
    const int on = 1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rank_dbg_on), &on, sizeof on);  // (counting starts with the first call)
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rank_dbg), sizeof(unsigned long long) * 128);
    if (reset) {
        unsigned long long z[128] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rank_dbg), z, sizeof z);
    }
}
}  // namespace skx
#endif
