// skx_kernels.hip -- hand-written gfx950 kernels for sketchy's streaming path.
//
// Pipeline of one scoring pass over a batch of B reads (DESIGN.md has the full picture):
//
//   sketch_wave_kernel     one wavefront per read: normalise, canonical k-mers, murmur3,
//                          bitonic sort in LDS, dedup, truncate to s          (A3-A6 of SURVEY 8(a))
//   gather_pairs_kernel    the part of every read sketch that can match at all (hash <= largest
//                          reference hash) becomes (read, hash) pairs
//   [sort + unique]        -> Q: the batch's sorted distinct query hashes        (skx_prim.hip)
//   pair_q_kernel          pair -> index of its hash in Q
//   window_kernel          per reference tile-band: the slice [qa,qb) of Q its hashes can meet
//   scan_kernel            THE roofline kernel: streams the resident s x N matrix once, probes an
//                          LDS table of the slice, ORs hit bits into M[word][genome]   (A2)
//   transpose_bits_kernel  M[word][genome] -> Mq[query][genome word] (64x64 bit transposes)
//   seg_sum / seg_prefix / rank_seg / topk_merge
//                          running table and per-read (sum desc, index asc) top-k       (A1, A7)
//
// Integer work throughout (u64 hash compares, bit counts): no MFMA.
#include "skx_common.hpp"
#include "skx_kernels.hpp"

namespace skx {

// =====================================================================================
// reference upload: genome-major columns -> tiled rank-major matrix
// =====================================================================================
// dst[t][i][c] = (i < eff_len[g]) ? src[g*s + i] : kPad   with g = t*256 + c.
// 32x32 LDS transpose so both sides are coalesced.  grid: (ceil(s/32), n_tile_genomes/32)
__global__ __launch_bounds__(256) void ref_tile_kernel(const u64* __restrict__ src, const u32* __restrict__ eff_len,
                                                       u64* __restrict__ dst, u32 s, u32 g_base, u32 n_genomes,
                                                       u32 g_count) {
    __shared__ u64 tile[32][33];
    const u32 tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;  // 32 x 8
    const u32 i0 = blockIdx.x * 32u, gl0 = blockIdx.y * 32u;  // gl: genome index local to this chunk
    for (u32 yy = ty; yy < 32u; yy += 8u) {
        const u32 gl = gl0 + yy, g = g_base + gl, i = i0 + tx;
        u64 v = kPad;
        if (gl < g_count && g < n_genomes && i < s && i < eff_len[g]) v = src[(size_t)gl * s + i];
        tile[yy][tx] = v;
    }
    __syncthreads();
    for (u32 yy = ty; yy < 32u; yy += 8u) {
        const u32 i = i0 + yy, gl = gl0 + tx, g = g_base + gl;
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
// read sketching: one wavefront per read
// =====================================================================================
// LDS per wave: hashes[CAP] u64, then codes[CAP + 64] bytes.
template <int KT, int CAP>
__global__ __launch_bounds__(256) void sketch_wave_kernel(const uint8_t* __restrict__ bases,
                                                          const u64* __restrict__ offsets, u32 n_reads, u32 k_rt,
                                                          u64 seed, u32 s, u64 max_ref, u64* __restrict__ out_sk,
                                                          u32 sk_stride, u32* __restrict__ out_len,
                                                          u32* __restrict__ out_cnt_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 kPerWave = CAP * 8 + CAP + 64;
    const u32 wv = threadIdx.x >> 6, lane = lane_id();
    const u32 r = blockIdx.x * 4u + wv;
    if (r >= n_reads) return;
    u64* hashes = reinterpret_cast<u64*>(smem + (size_t)wv * kPerWave);
    uint8_t* codes = smem + (size_t)wv * kPerWave + CAP * 8;
    const u32 k = KT > 0 ? (u32)KT : k_rt;
    const u64 o0 = offsets[r], o1 = offsets[r + 1];
    const u32 lraw = (u32)(o1 - o0);  // host guarantees lraw <= CAP + k - 1
    const u64 lt = lanemask_lt();

    // 1. normalise into 2-bit codes (whitespace dropped, everything not ACGTU -> 4)
    u32 nb = 0;
    for (u32 base = 0; base < lraw; base += 64u) {
        const u32 idx = base + lane;
        const u32 ch = idx < lraw ? (u32)bases[o0 + idx] : (u32)' ';
        const u32 code = classify_base(ch);
        const bool keep = code != 5u;
        const u64 mask = __ballot(keep);
        if (keep) codes[nb + __popcll(mask & lt)] = (uint8_t)code;
        nb += __popcll(mask);
    }
    wave_sync();

    // 2. canonical k-mer hashes, compacted in position order
    u32 m = 0;
    const u32 nk = nb >= k ? nb - k + 1u : 0u;
    for (u32 base = 0; base < nk; base += 64u) {
        const u32 p = base + lane;
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
        valid = valid && (bad == 0);
        const u64 canon = fwd < rc ? fwd : rc;
        const u64 h = hash_canonical_packed<KT>(canon, k, seed);
        const u64 mask = __ballot(valid);
        if (valid) hashes[m + __popcll(mask & lt)] = h;
        m += __popcll(mask);
    }
    // pad to a power of two (>= 64) for the bitonic network
    u32 p2 = 64;
    while (p2 < m) p2 <<= 1;
    for (u32 i = m + lane; i < p2; i += 64u) hashes[i] = kPad;
    wave_sync();

    // 3. bitonic sort ascending
    if (m > 1) {
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

    // 4. distinct, truncate to s, count the part that can meet the reference at all
    u32 outn = 0, cin = 0;
    u64* out = out_sk + (size_t)r * sk_stride;
    for (u32 base = 0; base < m && outn < s; base += 64u) {
        const u32 i = base + lane;
        const bool v = i < m;
        const u64 h = v ? hashes[i] : 0;
        const bool head = v && (i == 0 || hashes[i - 1] != h);
        const u64 mask = __ballot(head);
        const u32 pos = outn + __popcll(mask & lt);
        const bool take = head && pos < s;
        if (take) out[pos] = h;
        cin += __popcll(__ballot(take && h <= max_ref));
        outn += __popcll(mask);
    }
    if (lane == 0) {
        out_len[r] = min(outn, s);
        out_cnt_in[r] = cin;
    }
}

// =====================================================================================
// dictionary of the batch's query hashes
// =====================================================================================
// pair_h[p], pair_r[p] for p in [poff[r]-p_base, ...): the first cnt_in[r] hashes of read r's sketch
__global__ void gather_pairs_kernel(const u64* __restrict__ sk, u32 sk_stride, const u32* __restrict__ poff,
                                    u32 r_begin, u32 r_end, u32 p_base, u64* __restrict__ pair_h,
                                    u32* __restrict__ pair_r) {
    // one wave per read
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = lane_id();
    const u32 r = r_begin + wave;
    if (r >= r_end) return;
    const u32 a = poff[r], b = poff[r + 1];
    for (u32 j = lane; j < b - a; j += 64u) {
        pair_h[a - p_base + j] = sk[(size_t)r * sk_stride + j];
        pair_r[a - p_base + j] = r - r_begin;
    }
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

__global__ void pair_q_kernel(const u64* __restrict__ pair_h, u32 n_pairs, const u64* __restrict__ q,
                              const u32* __restrict__ n_q, u32* __restrict__ pair_q) {
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_pairs) pair_q[p] = lower_bound_u64(q, *n_q, pair_h[p]);
}

// win[2*bt] = qa, win[2*bt+1] = qb : Q[qa..qb) are the query hashes inside [lo[bt], hi[bt]]
__global__ void window_kernel(const u64* __restrict__ lo, const u64* __restrict__ hi, u32 n_bt,
                              const u64* __restrict__ q, const u32* __restrict__ n_q, u32* __restrict__ win) {
    const u32 bt = blockIdx.x * blockDim.x + threadIdx.x;
    if (bt >= n_bt) return;
    const u32 nq = *n_q;
    u32 qa = 0, qb = 0;
    if (lo[bt] <= hi[bt]) { qa = lower_bound_u64(q, nq, lo[bt]); qb = upper_bound_u64(q, nq, hi[bt]); }
    win[2 * bt] = qa;
    win[2 * bt + 1] = qb;
}

// reference hashes >= kEmpty were lifted out of the matrix at upload (they would alias the table's
// markers); set their bits here.  exc_g / exc_h: (genome, hash) pairs.
__global__ void exceptions_kernel(const u32* __restrict__ exc_g, const u64* __restrict__ exc_h, u32 n_exc,
                                  const u64* __restrict__ q, const u32* __restrict__ n_q, u64* __restrict__ m_bits,
                                  u32 n_pad) {
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_exc) return;
    const u32 nq = *n_q;
    const u32 pos = lower_bound_u64(q, nq, exc_h[e]);
    if (pos < nq && q[pos] == exc_h[e])
        atomicOr(&m_bits[(size_t)(pos >> 6) * n_pad + exc_g[e]], 1ull << (pos & 63u));
}

// =====================================================================================
// the reference scan
// =====================================================================================
// One block per (band, tile): 256 lanes = 256 genome columns, rows [b*rb, (b+1)*rb).
// LDS: open-addressing table of the band's slice of Q (keys 8 B + local index 2 B per slot).
// A lane meets its column's hits in ascending q, so it ORs them into one 64-bit word in a
// register and flushes that word to M[word][genome] when the word index moves on.
template <int TSLOTS>
__global__ __launch_bounds__(256) void scan_kernel(const u64* __restrict__ mat, u32 s, u32 n_tiles, u32 rb,
                                                   const u64* __restrict__ q, const u32* __restrict__ win,
                                                   u64* __restrict__ m_bits, u32 n_pad) {
    __shared__ u64 keys[TSLOTS];
    __shared__ unsigned short vals[TSLOTS];
    constexpr u32 kMask = TSLOTS - 1;
    constexpr u32 kSub = TSLOTS / 2;  // entries per table build (load factor <= 0.5)
    const u32 bt = blockIdx.x;
    const u32 t = bt % n_tiles, b = bt / n_tiles, c = threadIdx.x;
    const u32 qa = win[2 * bt], qb = win[2 * bt + 1];
    if (qa >= qb) return;
    const u32 i0 = b * rb, i1 = min(s, i0 + rb);
    const u64* col = mat + ((size_t)t * s + i0) * kTileGenomes + c;
    const u32 g = t * kTileGenomes + c;

    for (u32 sub = qa; sub < qb; sub += kSub) {
        const u32 n = min(kSub, qb - sub);
        for (u32 j = c; j < TSLOTS; j += 256u) keys[j] = kEmpty;
        __syncthreads();
        for (u32 j = c; j < n; j += 256u) {
            const u64 h = q[sub + j];
            u32 slot = (u32)h & kMask;
            for (;;) {
                const u64 old = atomicCAS(&keys[slot], kEmpty, h);
                if (old == kEmpty) { vals[slot] = (unsigned short)j; break; }
                slot = (slot + 1u) & kMask;
            }
        }
        __syncthreads();

        u32 cur_w = 0xFFFFFFFFu;
        u64 cur_bits = 0;
        const u32 rows = i1 - i0;
        u32 i = 0;
        // 8 independent loads in flight per lane
        for (; i + 8u <= rows; i += 8u) {
            u64 h[8];
#pragma unroll
            for (u32 u = 0; u < 8u; ++u) h[u] = col[(size_t)(i + u) * kTileGenomes];
#pragma unroll
            for (u32 u = 0; u < 8u; ++u) {
                u32 slot = (u32)h[u] & kMask;
                u64 e = keys[slot];
                while (e != kEmpty && e != h[u]) { slot = (slot + 1u) & kMask; e = keys[slot]; }
                if (e == h[u]) {
                    const u32 qi = sub + vals[slot];
                    const u32 w = qi >> 6;
                    if (w != cur_w) {
                        if (cur_bits) atomicOr(&m_bits[(size_t)cur_w * n_pad + g], cur_bits);
                        cur_w = w; cur_bits = 0;
                    }
                    cur_bits |= 1ull << (qi & 63u);
                }
            }
        }
        for (; i < rows; ++i) {
            const u64 hh = col[(size_t)i * kTileGenomes];
            u32 slot = (u32)hh & kMask;
            u64 e = keys[slot];
            while (e != kEmpty && e != hh) { slot = (slot + 1u) & kMask; e = keys[slot]; }
            if (e == hh) {
                const u32 qi = sub + vals[slot];
                const u32 w = qi >> 6;
                if (w != cur_w) {
                    if (cur_bits) atomicOr(&m_bits[(size_t)cur_w * n_pad + g], cur_bits);
                    cur_w = w; cur_bits = 0;
                }
                cur_bits |= 1ull << (qi & 63u);
            }
        }
        if (cur_bits) atomicOr(&m_bits[(size_t)cur_w * n_pad + g], cur_bits);
        __syncthreads();  // table is rebuilt by the next sub-window
    }
}

// =====================================================================================
// M[word][genome] (bit j of the word = query 64*word + j)  ->  Mq[query][genome word]
// =====================================================================================
// one wave per (word w, genome group gw): 64 ballots transpose a 64x64 bit block.
__global__ __launch_bounds__(256) void transpose_bits_kernel(const u64* __restrict__ m_bits, u32 n_pad, u32 n_words,
                                                             u64* __restrict__ mq, u32 n_gw) {
    const u32 wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = lane_id();
    const u32 gw = wave % n_gw, w = wave / n_gw;
    if (w >= n_words) return;
    const u64 word = m_bits[(size_t)w * n_pad + gw * 64u + lane];
    u64 mine = 0;
#pragma unroll
    for (u32 j = 0; j < 64u; ++j) {
        const u64 bal = __ballot((word >> j) & 1ull);
        if (lane == j) mine = bal;
    }
    mq[(size_t)(w * 64u + lane) * n_gw + gw] = mine;
}

// =====================================================================================
// running table and per-read ranking
// =====================================================================================
// Reads of a pass are cut into segments of seg_len reads.  pair_r/pair_q are sorted by read;
// poff[r] (absolute, minus p_base) delimits read r's pairs.
//
// seg_sum: inc[seg][g] = sum over the segment's pairs of bit(Mq[q][g]).  One wave per (gw, seg).
__global__ __launch_bounds__(256) void seg_sum_kernel(const u32* __restrict__ pair_q, const u32* __restrict__ poff,
                                                      u32 p_base, u32 r_begin, u32 n_reads, u32 seg_len,
                                                      const u64* __restrict__ mq, u32 n_gw, u32 n_pad,
                                                      u32* __restrict__ inc) {
    const u32 wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = lane_id();
    const u32 n_seg = (n_reads + seg_len - 1) / seg_len;
    const u32 gw = wave % n_gw, seg = wave / n_gw;
    if (seg >= n_seg) return;
    const u32 ra = seg * seg_len, rz = min(n_reads, ra + seg_len);
    const u32 pa = poff[r_begin + ra] - p_base, pz = poff[r_begin + rz] - p_base;
    u32 acc = 0;
    for (u32 p0 = pa; p0 < pz; p0 += 64u) {
        const u32 n = min(64u, pz - p0);
        u64 myword = 0;
        if (lane < n) myword = mq[(size_t)pair_q[p0 + lane] * n_gw + gw];
        for (u32 j = 0; j < n; ++j) {
            const u64 word = readlane64(myword, (int)j);
            acc += (u32)((word >> lane) & 1ull);
        }
    }
    inc[(size_t)seg * n_pad + gw * 64u + lane] = acc;
}

// start[seg][g] = cum[g] + sum_{seg' < seg} inc[seg'][g];  cum[g] += sum of all.  One thread per genome.
__global__ void seg_prefix_kernel(const u32* __restrict__ inc, u32 n_seg, u32 n_pad, u64* __restrict__ cum,
                                  u64* __restrict__ start) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_pad) return;
    u64 run = cum[g];
    for (u32 sgi = 0; sgi < n_seg; ++sgi) {
        start[(size_t)sgi * n_pad + g] = run;
        run += inc[(size_t)sgi * n_pad + g];
    }
    cum[g] = run;
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

// rank_seg: walk a segment's reads in order from start[seg]; after every read emit this genome
// group's top_k candidates cand_sum/cand_idx[(r * n_gw + gw) * top_k + j].  One wave per (gw, seg).
__global__ __launch_bounds__(256) void rank_seg_kernel(const u32* __restrict__ pair_q, const u32* __restrict__ pair_r,
                                                       const u32* __restrict__ poff, u32 p_base, u32 r_begin,
                                                       u32 n_reads, u32 seg_len, const u64* __restrict__ mq,
                                                       u32 n_gw, u32 n_pad, u32 n_genomes,
                                                       const u64* __restrict__ start, u32 top_k,
                                                       u64* __restrict__ cand_sum, u32* __restrict__ cand_idx) {
    const u32 wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = lane_id();
    const u32 n_seg = (n_reads + seg_len - 1) / seg_len;
    const u32 gw = wave % n_gw, seg = wave / n_gw;
    if (seg >= n_seg) return;
    const u32 ra = seg * seg_len, rz = min(n_reads, ra + seg_len);
    const u32 pa = poff[r_begin + ra] - p_base, pz = poff[r_begin + rz] - p_base;
    const u32 g = gw * 64u + lane;
    const bool real = g < n_genomes;
    u64 state = start[(size_t)seg * n_pad + g];
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
        if (lane < n) { myword = mq[(size_t)pair_q[p0 + lane] * n_gw + gw]; myread = pair_r[p0 + lane]; }
        for (u32 j = 0; j < n; ++j) {
            const u64 word = readlane64(myword, (int)j);
            const u32 rd = __builtin_amdgcn_readlane(myread, (int)j);
            while (cur < rd) { emit(cur); ++cur; }
            state += (word >> lane) & 1ull;
        }
    }
    while (cur < rz) { emit(cur); ++cur; }
}

// topk_merge: per read, merge the n_gw*top_k candidates into the final top_k.  One wave per read.
__global__ __launch_bounds__(256) void topk_merge_kernel(const u64* __restrict__ cand_sum,
                                                         const u32* __restrict__ cand_idx, u32 n_reads, u32 n_cand,
                                                         u32 top_k, u32* __restrict__ out_idx,
                                                         u64* __restrict__ out_sum, u32 out_r0) {
    const u32 r = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = lane_id();
    if (r >= n_reads) return;
    const u64* cs = cand_sum + (size_t)r * n_cand;
    const u32* ci = cand_idx + (size_t)r * n_cand;
    u64 ps = 0; u32 pi = 0; bool first = true;
    for (u32 j = 0; j < top_k; ++j) {
        u64 bs = 0; u32 bi = 0xFFFFFFFFu;
        for (u32 c = lane; c < n_cand; c += 64u) {
            const u64 s_ = cs[c]; const u32 i_ = ci[c];
            if (i_ == 0xFFFFFFFFu) continue;
            if (!first && !ranks_before(ps, pi, s_, i_)) continue;
            if (bi == 0xFFFFFFFFu || ranks_before(s_, i_, bs, bi)) { bs = s_; bi = i_; }
        }
        wave_best(bs, bi, bi != 0xFFFFFFFFu);
        if (lane == 0) {
            out_idx[(size_t)(out_r0 + r) * top_k + j] = bi;
            out_sum[(size_t)(out_r0 + r) * top_k + j] = bs;
        }
        ps = bs; pi = bi; first = false;
    }
}

// rank the table itself: one block, top_k rounds.  (skx_stream_rank; also the all-reduced table)
__global__ __launch_bounds__(1024) void rank_table_kernel(const u64* __restrict__ cum, u32 n_genomes, u32 top_k,
                                                          u32* __restrict__ out_idx, u64* __restrict__ out_sum) {
    __shared__ u64 ssum[16];
    __shared__ u32 sidx[16];
    __shared__ u64 wsum;
    __shared__ u32 widx;
    const u32 tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    u64 ps = 0; u32 pi = 0; bool first = true;
    for (u32 j = 0; j < top_k; ++j) {
        u64 bs = 0; u32 bi = 0xFFFFFFFFu;
        for (u32 g = tid; g < n_genomes; g += 1024u) {
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
            if (lane == 0) { wsum = s2; widx = i2; out_idx[j] = i2; out_sum[j] = s2; }
        }
        __syncthreads();
        ps = wsum; pi = widx; first = false;
        __syncthreads();
    }
}

// parity/debug: shared[r][g] = sum over read r's pairs of bit(Mq[q][g]).  One block per read.
__global__ __launch_bounds__(256) void shared_debug_kernel(const u32* __restrict__ pair_q, const u32* __restrict__ poff,
                                                           u32 p_base, u32 r_begin, const u64* __restrict__ mq,
                                                           u32 n_gw, u32 n_genomes, u32* __restrict__ shared,
                                                           u32 out_r0) {
    const u32 r = blockIdx.x;
    const u32 pa = poff[r_begin + r] - p_base, pz = poff[r_begin + r + 1] - p_base;
    for (u32 g = threadIdx.x; g < n_genomes; g += 256u) {
        u32 acc = 0;
        for (u32 p = pa; p < pz; ++p) acc += (u32)((mq[(size_t)pair_q[p] * n_gw + (g >> 6)] >> (g & 63u)) & 1ull);
        shared[(size_t)(out_r0 + r) * n_genomes + g] = acc;
    }
}

__global__ void add_table_kernel(u64* __restrict__ cum, const u64* __restrict__ add, u32 n) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) cum[g] += add[g];
}

// =====================================================================================
// launchers
// =====================================================================================
static inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

void launch_ref_tile(hipStream_t st, const u64* src, const u32* eff_len, u64* dst, u32 s, u32 g_base, u32 n_genomes,
                     u32 g_count) {
    dim3 grid(cdiv(s, 32), cdiv(g_count, 32));
    hipLaunchKernelGGL(ref_tile_kernel, grid, dim3(256), 0, st, src, eff_len, dst, s, g_base, n_genomes, g_count);
}
void launch_band_bounds(hipStream_t st, const u64* mat, u32 s, u32 n_tiles, u32 rb, u32 n_bands, u64* lo, u64* hi) {
    hipLaunchKernelGGL(band_bounds_kernel, dim3(n_tiles * n_bands), dim3(256), 0, st, mat, s, n_tiles, rb, lo, hi);
}

size_t sketch_wave_lds_bytes() { return 4 * (size_t)(kSketchCap * 8 + kSketchCap + 64); }

void launch_sketch_wave(hipStream_t st, const uint8_t* bases, const u64* offsets, u32 n_reads, u32 k, u64 seed, u32 s,
                        u64 max_ref, u64* out_sk, u32 sk_stride, u32* out_len, u32* out_cnt_in) {
    if (n_reads == 0) return;
    const size_t lds = sketch_wave_lds_bytes();
    dim3 grid(cdiv(n_reads, 4));
    static bool attr_set = false;  // > 64 KiB of dynamic LDS needs the opt-in (gfx950 has 160 KiB per CU)
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sketch_wave_kernel<16, kSketchCap>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sketch_wave_kernel<0, kSketchCap>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (k == 16)
        hipLaunchKernelGGL((sketch_wave_kernel<16, kSketchCap>), grid, dim3(256), lds, st, bases, offsets, n_reads, k,
                           seed, s, max_ref, out_sk, sk_stride, out_len, out_cnt_in);
    else
        hipLaunchKernelGGL((sketch_wave_kernel<0, kSketchCap>), grid, dim3(256), lds, st, bases, offsets, n_reads, k,
                           seed, s, max_ref, out_sk, sk_stride, out_len, out_cnt_in);
}

void launch_gather_pairs(hipStream_t st, const u64* sk, u32 sk_stride, const u32* poff, u32 r_begin, u32 r_end,
                         u32 p_base, u64* pair_h, u32* pair_r) {
    if (r_end <= r_begin) return;
    hipLaunchKernelGGL(gather_pairs_kernel, dim3(cdiv(r_end - r_begin, 4)), dim3(256), 0, st, sk, sk_stride, poff,
                       r_begin, r_end, p_base, pair_h, pair_r);
}
void launch_pair_q(hipStream_t st, const u64* pair_h, u32 n_pairs, const u64* q, const u32* n_q, u32* pair_q) {
    if (n_pairs == 0) return;
    hipLaunchKernelGGL(pair_q_kernel, dim3(cdiv(n_pairs, 256)), dim3(256), 0, st, pair_h, n_pairs, q, n_q, pair_q);
}
void launch_window(hipStream_t st, const u64* lo, const u64* hi, u32 n_bt, const u64* q, const u32* n_q, u32* win) {
    hipLaunchKernelGGL(window_kernel, dim3(cdiv(n_bt, 256)), dim3(256), 0, st, lo, hi, n_bt, q, n_q, win);
}
void launch_exceptions(hipStream_t st, const u32* exc_g, const u64* exc_h, u32 n_exc, const u64* q, const u32* n_q,
                       u64* m_bits, u32 n_pad) {
    if (n_exc == 0) return;
    hipLaunchKernelGGL(exceptions_kernel, dim3(cdiv(n_exc, 256)), dim3(256), 0, st, exc_g, exc_h, n_exc, q, n_q,
                       m_bits, n_pad);
}
void launch_scan(hipStream_t st, const u64* mat, u32 s, u32 n_tiles, u32 rb, u32 n_bands, const u64* q, const u32* win,
                 u64* m_bits, u32 n_pad) {
    hipLaunchKernelGGL((scan_kernel<kScanSlots>), dim3(n_tiles * n_bands), dim3(256), 0, st, mat, s, n_tiles, rb, q,
                       win, m_bits, n_pad);
}
void launch_transpose_bits(hipStream_t st, const u64* m_bits, u32 n_pad, u32 n_words, u64* mq) {
    if (n_words == 0) return;
    const u32 n_gw = n_pad / 64;
    hipLaunchKernelGGL(transpose_bits_kernel, dim3(cdiv((u64)n_words * n_gw, 4)), dim3(256), 0, st, m_bits, n_pad,
                       n_words, mq, n_gw);
}
void launch_seg_sum(hipStream_t st, const u32* pair_q, const u32* poff, u32 p_base, u32 r_begin, u32 n_reads,
                    u32 seg_len, const u64* mq, u32 n_pad, u32* inc) {
    const u32 n_gw = n_pad / 64, n_seg = cdiv(n_reads, seg_len);
    hipLaunchKernelGGL(seg_sum_kernel, dim3(cdiv((u64)n_seg * n_gw, 4)), dim3(256), 0, st, pair_q, poff, p_base,
                       r_begin, n_reads, seg_len, mq, n_gw, n_pad, inc);
}
void launch_seg_prefix(hipStream_t st, const u32* inc, u32 n_seg, u32 n_pad, u64* cum, u64* start) {
    hipLaunchKernelGGL(seg_prefix_kernel, dim3(cdiv(n_pad, 256)), dim3(256), 0, st, inc, n_seg, n_pad, cum, start);
}
void launch_rank_seg(hipStream_t st, const u32* pair_q, const u32* pair_r, const u32* poff, u32 p_base, u32 r_begin,
                     u32 n_reads, u32 seg_len, const u64* mq, u32 n_pad, u32 n_genomes, const u64* start, u32 top_k,
                     u64* cand_sum, u32* cand_idx) {
    const u32 n_gw = n_pad / 64, n_seg = cdiv(n_reads, seg_len);
    hipLaunchKernelGGL(rank_seg_kernel, dim3(cdiv((u64)n_seg * n_gw, 4)), dim3(256), 0, st, pair_q, pair_r, poff,
                       p_base, r_begin, n_reads, seg_len, mq, n_gw, n_pad, n_genomes, start, top_k, cand_sum,
                       cand_idx);
}
void launch_topk_merge(hipStream_t st, const u64* cand_sum, const u32* cand_idx, u32 n_reads, u32 n_cand, u32 top_k,
                       u32* out_idx, u64* out_sum, u32 out_r0) {
    if (n_reads == 0) return;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(cdiv(n_reads, 4)), dim3(256), 0, st, cand_sum, cand_idx, n_reads,
                       n_cand, top_k, out_idx, out_sum, out_r0);
}
void launch_rank_table(hipStream_t st, const u64* cum, u32 n_genomes, u32 top_k, u32* out_idx, u64* out_sum) {
    hipLaunchKernelGGL(rank_table_kernel, dim3(1), dim3(1024), 0, st, cum, n_genomes, top_k, out_idx, out_sum);
}
void launch_shared_debug(hipStream_t st, const u32* pair_q, const u32* poff, u32 p_base, u32 r_begin, u32 n_reads,
                         const u64* mq, u32 n_pad, u32 n_genomes, u32* shared, u32 out_r0) {
    if (n_reads == 0) return;
    hipLaunchKernelGGL(shared_debug_kernel, dim3(n_reads), dim3(256), 0, st, pair_q, poff, p_base, r_begin, mq,
                       n_pad / 64, n_genomes, shared, out_r0);
}
void launch_add_table(hipStream_t st, u64* cum, const u64* add, u32 n) {
    hipLaunchKernelGGL(add_table_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, cum, add, n);
}

}  // namespace skx
