/*
 * oracle.c -- CPU restatement of sketchy's streaming read-vs-reference MinHash path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call anything in oracle/.  The
 * product path (sketchy_amd/, libsketchy_hip.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference (esteinig/sketchy v0.6.0, Rust) ships no tests, golden
 * vectors or fixtures for this path, and its toolchain (cargo/rustc) and its crates
 * (finch 0.4.1, murmurhash3 0.0.5, needletail 0.4.1 -- Cargo.lock:220-236, :341-345,
 * :360-372) are absent here, so it cannot be run to generate any.  This file restates
 * the published algorithms of those crates and follows the reference's own call sites;
 * it is pinned only by public MurmurHash3 known answers (SMHasher verification value
 * 0x6384BA69, mmh3.hash64("foo")) and by self-consistency vectors (tests/golden/).
 *
 * What follows what (paths under /root/reference):
 *   orc_murmur3_x64_128   murmurhash3 0.0.5 src/mmh3_128.rs (canonical MurmurHash3_x64_128,
 *                         u64 seed into both lanes); finch hash_f keeps .0
 *                         (finch 0.4.1 src/sketch_schemes/hashing.rs)
 *   orc_normalize         needletail 0.4.1 src/sequence.rs normalize(seq, iupac=false)
 *   orc_sketch_heap       finch 0.4.1 src/sketch_schemes/mash.rs MashSketcher::{new,process,
 *                         push,to_vec}; called at src/sketchy.rs:331-335
 *   orc_sketch_sort       the net semantics of the above, an
 *                         independent second implementation used to cross-check the heap one
 *   orc_common_hashes     src/sketchy.rs:419-459 (_common_hashes; min_scale = 0 for Mash
 *                         sketches, :84-87, so the tail loops :441-457 are not taken)
 *   orc_stream            src/sketchy.rs:317-356 (_sum_of_shared_hashes) with the stable
 *                         descending sort of :348 and the top rows of :391
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ murmur3 x64 128 */

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static inline uint64_t load_le64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}

ORC_EXPORT void orc_murmur3_x64_128(const uint8_t *key, uint64_t len, uint64_t seed, uint64_t out[2]) {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t nblocks = len / 16;
    for (uint64_t b = 0; b < nblocks; ++b) {
        uint64_t k1 = load_le64(key + 16 * b), k2 = load_le64(key + 16 * b + 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729ULL;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5ULL;
    }
    const uint8_t *tail = key + 16 * nblocks;
    uint64_t k1 = 0, k2 = 0;
    switch (len & 15) {
    case 15: k2 ^= (uint64_t)tail[14] << 48; /* fallthrough */
    case 14: k2 ^= (uint64_t)tail[13] << 40; /* fallthrough */
    case 13: k2 ^= (uint64_t)tail[12] << 32; /* fallthrough */
    case 12: k2 ^= (uint64_t)tail[11] << 24; /* fallthrough */
    case 11: k2 ^= (uint64_t)tail[10] << 16; /* fallthrough */
    case 10: k2 ^= (uint64_t)tail[9] << 8;   /* fallthrough */
    case 9:  k2 ^= (uint64_t)tail[8];
             k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; /* fallthrough */
    case 8:  k1 ^= (uint64_t)tail[7] << 56; /* fallthrough */
    case 7:  k1 ^= (uint64_t)tail[6] << 48; /* fallthrough */
    case 6:  k1 ^= (uint64_t)tail[5] << 40; /* fallthrough */
    case 5:  k1 ^= (uint64_t)tail[4] << 32; /* fallthrough */
    case 4:  k1 ^= (uint64_t)tail[3] << 24; /* fallthrough */
    case 3:  k1 ^= (uint64_t)tail[2] << 16; /* fallthrough */
    case 2:  k1 ^= (uint64_t)tail[1] << 8;  /* fallthrough */
    case 1:  k1 ^= (uint64_t)tail[0];
             k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    out[0] = h1; out[1] = h2;
}

/* finch hash_f: murmurhash3_x64_128(item, seed).0 */
static inline uint64_t hash_f(const uint8_t *kmer, uint32_t k, uint64_t seed) {
    uint64_t o[2];
    orc_murmur3_x64_128(kmer, k, seed, o);
    return o[0];
}

/* ------------------------------------------------------------------ normalise / canonical */

/* needletail normalize(seq, false): returns number of bytes written to out (<= n). */
ORC_EXPORT uint64_t orc_normalize(const uint8_t *seq, uint64_t n, uint8_t *out) {
    uint64_t w = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint8_t c = seq[i], o;
        switch (c) {
        case 'A': case 'C': case 'G': case 'T': case 'N': case '-': o = c; break;
        case 'a': o = 'A'; break;
        case 'c': o = 'C'; break;
        case 'g': o = 'G'; break;
        case 't': case 'u': case 'U': o = 'T'; break;
        case '.': case '~': o = '-'; break;
        case ' ': case '\t': case '\r': case '\n': continue; /* whitespace removed */
        default: o = 'N'; break; /* IUPAC codes are not allowed -> N */
        }
        out[w++] = o;
    }
    return w;
}

static inline uint8_t complement(uint8_t c) {
    switch (c) {
    case 'A': return 'T';
    case 'T': return 'A';
    case 'C': return 'G';
    case 'G': return 'C';
    default:  return c; /* N, - unchanged; such windows are skipped anyway */
    }
}

static inline int is_acgt(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

/*
 * Enumerate the canonical k-mers of an (un-normalised) sequence in position order and
 * write their hashes to out (capacity >= n).  Returns the number of valid k-mers.
 * needletail CanonicalKmers: windows with any non-ACGT byte are skipped;
 * canonical = fwd if fwd < rc (bytewise) else rc.
 */
ORC_EXPORT uint64_t orc_kmer_hashes(const uint8_t *seq, uint64_t n, uint32_t k, uint64_t seed,
                                    uint64_t *out, uint8_t *is_rc /* may be NULL */) {
    if (k == 0 || k > 255) return 0;
    uint8_t *norm = (uint8_t *)malloc(n + 1), *rc = (uint8_t *)malloc(n + 1);
    uint64_t L = orc_normalize(seq, n, norm), m = 0;
    for (uint64_t i = 0; i < L; ++i) rc[i] = complement(norm[L - 1 - i]);
    if (L >= k) {
        uint64_t bad = 0; /* number of non-ACGT bytes in the current window */
        for (uint64_t i = 0; i < k - 1; ++i) bad += !is_acgt(norm[i]);
        for (uint64_t p = 0; p + k <= L; ++p) {
            bad += !is_acgt(norm[p + k - 1]);
            if (bad == 0) {
                const uint8_t *fwd = norm + p, *rev = rc + (L - p - k);
                int use_rc = !(memcmp(fwd, rev, k) < 0);
                out[m] = hash_f(use_rc ? rev : fwd, k, seed);
                if (is_rc) is_rc[m] = (uint8_t)use_rc;
                ++m;
            }
            bad -= !is_acgt(norm[p]);
        }
    }
    free(norm); free(rc);
    return m;
}

/* ------------------------------------------------------------------ bottom-s sketchers */

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

/* net semantics: ascending min(s, #distinct) smallest distinct hashes */
static uint64_t bottom_s_sort(uint64_t *h, uint64_t m, uint64_t s, uint64_t *out) {
    qsort(h, m, sizeof(uint64_t), cmp_u64);
    uint64_t w = 0;
    for (uint64_t i = 0; i < m && w < s; ++i)
        if (i == 0 || h[i] != h[i - 1]) out[w++] = h[i];
    return w;
}

ORC_EXPORT uint64_t orc_sketch_sort(const uint8_t *seq, uint64_t n, uint32_t k, uint64_t seed,
                                    uint64_t s, uint64_t *out /* capacity >= s */) {
    uint64_t *h = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
    uint64_t m = orc_kmer_hashes(seq, n, k, seed, h, NULL);
    uint64_t w = bottom_s_sort(h, m, s, out);
    free(h);
    return w;
}

/*
 * Faithful MashSketcher: binary max-heap of hashes bounded by s plus a membership set
 * (finch keys a HashMap<u64,_> by hash with a pass-through hasher; here an open-addressing
 * table with tombstone-free deletion by rebuild is overkill -- membership is tested on the
 * heap's content through a small chained table).
 */
typedef struct { uint64_t *heap; uint64_t len, cap; uint64_t *tab; uint8_t *used; uint64_t tcap; } msk_t;

static void set_insert(msk_t *m, uint64_t h) {
    uint64_t i = (h * 0x9E3779B97F4A7C15ULL) >> 32;
    for (i &= m->tcap - 1; m->used[i] == 1; i = (i + 1) & (m->tcap - 1)) {}
    m->used[i] = 1; m->tab[i] = h;
}
static int set_contains(const msk_t *m, uint64_t h) {
    uint64_t i = (h * 0x9E3779B97F4A7C15ULL) >> 32;
    for (i &= m->tcap - 1; m->used[i] != 0; i = (i + 1) & (m->tcap - 1))
        if (m->used[i] == 1 && m->tab[i] == h) return 1;
    return 0;
}
static void set_remove(msk_t *m, uint64_t h) {
    uint64_t i = (h * 0x9E3779B97F4A7C15ULL) >> 32;
    for (i &= m->tcap - 1; m->used[i] != 0; i = (i + 1) & (m->tcap - 1))
        if (m->used[i] == 1 && m->tab[i] == h) { m->used[i] = 2; return; } /* tombstone */
}
static void heap_push(msk_t *m, uint64_t h) {
    uint64_t i = m->len++;
    m->heap[i] = h;
    while (i > 0) {
        uint64_t p = (i - 1) / 2;
        if (m->heap[p] >= m->heap[i]) break;
        uint64_t t = m->heap[p]; m->heap[p] = m->heap[i]; m->heap[i] = t; i = p;
    }
}
static uint64_t heap_pop(msk_t *m) {
    uint64_t top = m->heap[0];
    m->heap[0] = m->heap[--m->len];
    uint64_t i = 0;
    for (;;) {
        uint64_t l = 2 * i + 1, r = l + 1, b = i;
        if (l < m->len && m->heap[l] > m->heap[b]) b = l;
        if (r < m->len && m->heap[r] > m->heap[b]) b = r;
        if (b == i) break;
        uint64_t t = m->heap[b]; m->heap[b] = m->heap[i]; m->heap[i] = t; i = b;
    }
    return top;
}

ORC_EXPORT uint64_t orc_sketch_heap(const uint8_t *seq, uint64_t n, uint32_t k, uint64_t seed,
                                    uint64_t s, uint64_t *out /* capacity >= s */) {
    uint64_t *h = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
    uint64_t nk = orc_kmer_hashes(seq, n, k, seed, h, NULL);
    msk_t m;
    m.cap = s + 1; m.len = 0;
    m.heap = (uint64_t *)malloc((m.cap + 1) * sizeof(uint64_t));
    /* tombstones accumulate with evictions: size for every push ever made */
    m.tcap = 64; while (m.tcap < 4 * (nk + 1)) m.tcap <<= 1;
    m.tab = (uint64_t *)malloc(m.tcap * sizeof(uint64_t));
    m.used = (uint8_t *)calloc(m.tcap, 1);
    for (uint64_t i = 0; i < nk; ++i) {
        uint64_t nh = h[i];
        /* MashSketcher::push */
        int add = (m.len == 0) || (nh <= m.heap[0]) || (m.len < s);
        if (!add) continue;
        if (set_contains(&m, nh)) continue; /* count bump only */
        heap_push(&m, nh);
        set_insert(&m, nh);
        if (m.len > s) { uint64_t ev = heap_pop(&m); set_remove(&m, ev); }
    }
    /* to_vec: into_sorted_vec ascending */
    uint64_t w = m.len;
    memcpy(out, m.heap, w * sizeof(uint64_t));
    qsort(out, w, sizeof(uint64_t), cmp_u64);
    free(h); free(m.heap); free(m.tab); free(m.used);
    return w;
}

/* ------------------------------------------------------------------ intersection */

/* src/sketchy.rs:425-438 two-pointer merge; both ascending */
ORC_EXPORT uint64_t orc_common_hashes(const uint64_t *ref_h, uint64_t nref, const uint64_t *query_h, uint64_t nq) {
    uint64_t i = 0, j = 0, common = 0;
    while (i < nq && j < nref) {
        if (query_h[i] < ref_h[j]) ++i;
        else if (query_h[i] > ref_h[j]) ++j;
        else { ++common; ++i; ++j; }
    }
    return common;
}

/* ------------------------------------------------------------------ stable rank */

/* stable merge sort of idx by sum descending == Rust sort_by(|a,b| b.1.cmp(&a.1)) on (i,sum[i]) */
static void stable_rank(const uint64_t *sum, uint32_t n, uint32_t *idx, uint32_t *tmp) {
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;
    for (uint32_t w = 1; w < n; w *= 2) {
        for (uint32_t lo = 0; lo < n; lo += 2 * w) {
            uint32_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            uint32_t a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                /* take from the right run only if strictly greater: keeps equal sums in index order */
                if (sum[idx[b]] > sum[idx[a]]) tmp[o++] = idx[b++]; else tmp[o++] = idx[a++];
            }
            while (a < mid) tmp[o++] = idx[a++];
            while (b < hi) tmp[o++] = idx[b++];
        }
        memcpy(idx, tmp, n * sizeof(uint32_t));
    }
}

ORC_EXPORT void orc_stable_rank(const uint64_t *sum, uint32_t n, uint32_t *idx_out) {
    uint32_t *tmp = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
    stable_rank(sum, n, idx_out, tmp);
    free(tmp);
}

/* ------------------------------------------------------------------ streaming driver */

/*
 * _sum_of_shared_hashes (src/sketchy.rs:317-356) over a packed batch of reads.
 *   ref_hashes: genome g's ascending hashes at [g*s, g*s + col_len[g])
 *   cum:        in/out running table (u64[n_genomes]); start with zeros for a fresh stream
 * Optional outputs (NULL to skip): topk_idx/topk_sum [n_reads][top_k], per_read_shared
 * [n_reads][n_genomes], sketches [n_reads][s] + sketch_len [n_reads].
 * rank_every_read = 0 skips the per-read sort (used only to time the scoring loop alone).
 */
/* `stride`: genome g's hashes start at ref_hashes + g * stride (col_len[g] <= stride).  The reference keeps every sketch in
 * its own Vec, of any length; the read sketch size s is the length of the FIRST one (src/sketchy.rs:82, :520-527), so a
 * collection can hold columns longer (or shorter) than s and they are still walked in full by _common_hashes (:425-438). */
ORC_EXPORT int orc_stream_strided(uint32_t k, uint64_t seed, uint32_t s, uint32_t stride, uint32_t n_genomes,
                                  const uint64_t *ref_hashes, const uint32_t *col_len,
                                  const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                                  uint32_t top_k, uint64_t *cum,
                                  uint32_t *topk_idx, uint64_t *topk_sum, uint32_t *per_read_shared,
                                  uint64_t *sketches, uint32_t *sketch_len, int rank_every_read) {
    if (top_k > n_genomes) return -1; /* the reference panics on [..top] (src/sketchy.rs:391) */
    uint64_t *sk = (uint64_t *)malloc(((uint64_t)s + 1) * sizeof(uint64_t));
    uint32_t *idx = (uint32_t *)malloc(((uint64_t)n_genomes + 1) * sizeof(uint32_t));
    uint32_t *tmp = (uint32_t *)malloc(((uint64_t)n_genomes + 1) * sizeof(uint32_t));
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint8_t *seq = bases + offsets[r];
        uint64_t n = offsets[r + 1] - offsets[r];
        uint64_t len = orc_sketch_heap(seq, n, k, seed, s, sk); /* fresh sketcher per read, :331 */
        if (sketches) memcpy(sketches + (uint64_t)r * s, sk, len * sizeof(uint64_t));
        if (sketch_len) sketch_len[r] = (uint32_t)len;
        for (uint32_t g = 0; g < n_genomes; ++g) { /* :337-347 */
            uint64_t sh = orc_common_hashes(ref_hashes + (uint64_t)g * stride, col_len[g], sk, len);
            cum[g] += sh;
            if (per_read_shared) per_read_shared[(uint64_t)r * n_genomes + g] = (uint32_t)sh;
        }
        if (rank_every_read && top_k > 0) { /* :348-349 */
            stable_rank(cum, n_genomes, idx, tmp);
            for (uint32_t t = 0; t < top_k; ++t) {
                if (topk_idx) topk_idx[(uint64_t)r * top_k + t] = idx[t];
                if (topk_sum) topk_sum[(uint64_t)r * top_k + t] = cum[idx[t]];
            }
        }
    }
    free(sk); free(idx); free(tmp);
    return 0;
}
ORC_EXPORT int orc_stream(uint32_t k, uint64_t seed, uint32_t s, uint32_t n_genomes,
                          const uint64_t *ref_hashes, const uint32_t *col_len,
                          const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                          uint32_t top_k, uint64_t *cum,
                          uint32_t *topk_idx, uint64_t *topk_sum, uint32_t *per_read_shared,
                          uint64_t *sketches, uint32_t *sketch_len, int rank_every_read) {
    return orc_stream_strided(k, seed, s, s, n_genomes, ref_hashes, col_len, bases, offsets, n_reads, top_k, cum, topk_idx,
                              topk_sum, per_read_shared, sketches, sketch_len, rank_every_read);
}

/*
 * The same driver with the N intersections of a read (src/sketchy.rs:337-347) spread over host threads (OpenMP over
 * genomes).  The reference itself is single-threaded on this path; this is the generous "all host cores" upper bound
 * of SURVEY.md 8(d)(ii) for bench.py's cpu_baseline_all_cores leg.  Same results (integer sums per genome).
 */
ORC_EXPORT int orc_stream_mt(uint32_t k, uint64_t seed, uint32_t s, uint32_t n_genomes,
                             const uint64_t *ref_hashes, const uint32_t *col_len,
                             const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                             uint32_t top_k, uint64_t *cum, uint32_t *topk_idx, uint64_t *topk_sum, int n_threads) {
    if (top_k > n_genomes) return -1;
    if (n_threads < 1) n_threads = 1;
    uint64_t *sk = (uint64_t *)malloc(((uint64_t)s + 1) * sizeof(uint64_t));
    uint32_t *idx = (uint32_t *)malloc(((uint64_t)n_genomes + 1) * sizeof(uint32_t));
    uint32_t *tmp = (uint32_t *)malloc(((uint64_t)n_genomes + 1) * sizeof(uint32_t));
    for (uint32_t r = 0; r < n_reads; ++r) {
        uint64_t len = orc_sketch_heap(bases + offsets[r], offsets[r + 1] - offsets[r], k, seed, s, sk);
        int64_t g;
#pragma omp parallel for schedule(static) num_threads(n_threads)
        for (g = 0; g < (int64_t)n_genomes; ++g)
            cum[g] += orc_common_hashes(ref_hashes + (uint64_t)g * s, col_len[g], sk, len);
        if (top_k > 0) {
            stable_rank(cum, n_genomes, idx, tmp);
            for (uint32_t t = 0; t < top_k; ++t) {
                if (topk_idx) topk_idx[(uint64_t)r * top_k + t] = idx[t];
                if (topk_sum) topk_sum[(uint64_t)r * top_k + t] = cum[idx[t]];
            }
        }
    }
    free(sk); free(idx); free(tmp);
    return 0;
}

/* ------------------------------------------------------------------ fast exact checker (full-size parity) */

/*
 * orc_stream_fast -- the rows and the table of orc_stream (src/sketchy.rs:317-356), computed batch-wise so that a
 * FULL-SIZE batch (98 304 reads against 40 000 x 10 000 hashes) takes seconds on the host's cores instead of the
 * 4x10^8 merge steps per read of the literal loop.  Still test infrastructure, still exact, and pinned against
 * orc_stream itself in tests/test_oracle.py (C0, C1-sized and random ragged cases).
 *
 * What it restates and why it is the same function:
 *   - every read is sketched by orc_sketch_heap (finch's MashSketcher, fresh per read, :331-335) at the full size s;
 *   - _common_hashes (:425-438) is a plain set intersection, and a read hash above the reference's largest hash is in
 *     no column, so only the sketch's prefix <= max_ref can count (the sketch is ascending: a prefix);
 *   - for a block of reads, Q = the distinct in-range hashes; member[q][g] = (Q[q] in column g) comes from ONE
 *     two-pointer merge of Q against every column (the same merge as :425-438, with Q in place of a read's sketch);
 *     shared(read, g) = number of the read's hashes q with member[q][g] (:341), sum[g] += shared in read order;
 *   - after every read the first top_k of (sum desc, genome index asc) (:348, :391) by a linear selection.
 * Host threads split the GENOMES (not the reads): every thread replays all reads in order over its own genome range
 * and keeps that range's best rows; the ranges' rows are merged per read.
 *
 * index_min: block dictionaries of at least this many hashes go through the bucket index (0 = default 4096; tests force 1).
 * stats (may be NULL): [0] reads with no in-range hash, [1] pairs (read, in-range hash), [2] sum over blocks of |Q|,
 * [3] blocks, [4] set bits of all member matrices.
 */
typedef struct { uint64_t sum; uint32_t idx; } orc_row_t;

static inline int row_before(uint64_t sa, uint32_t ia, uint64_t sb, uint32_t ib) { /* (sum desc, index asc) */
    return sa > sb || (sa == sb && ia < ib);
}

static int fast_block(uint32_t k, uint64_t seed, uint32_t s, uint32_t stride, uint32_t n_genomes, const uint64_t *ref,
                      const uint32_t *col_len, uint64_t max_ref, int have_ref, const uint8_t *bases, const uint64_t *offsets,
                      uint32_t r0, uint32_t r1, uint32_t top_k, uint64_t *cum, uint32_t *topk_idx, uint64_t *topk_sum,
                      int n_threads, uint64_t mem_limit, uint32_t index_min, uint64_t *stats) {
    const uint32_t nr = r1 - r0;
    uint64_t **lists = (uint64_t **)calloc(nr, sizeof(uint64_t *));
    uint32_t *lens = (uint32_t *)calloc(nr + 1, sizeof(uint32_t));
    int64_t i;
    /* 1. sketches (full size s, faithful sketcher), in-range prefix kept */
#pragma omp parallel num_threads(n_threads)
    {
        uint64_t *sk = (uint64_t *)malloc(((uint64_t)s + 1) * sizeof(uint64_t));
        uint64_t *hh = NULL, hh_cap = 0;
#pragma omp for schedule(dynamic, 64)
        for (i = 0; i < (int64_t)nr; ++i) {
            uint32_t r = r0 + (uint32_t)i;
            const uint64_t n = offsets[r + 1] - offsets[r];
            uint64_t len, keep = 0;
            if (n <= s) {
                /* A read with at most s bases has at most s k-mers: MashSketcher never evicts (heap.len() never exceeds s,
                 * finch push), so its sketch is ALL its distinct hashes, ascending -- no heap needed to know the prefix
                 * <= max_ref: the distinct hashes <= max_ref, sorted.  (Longer reads go through the faithful sketcher below;
                 * tests/test_oracle.py pins both branches against orc_stream.) */
                if (n + 1 > hh_cap) { free(hh); hh_cap = n + 1; hh = (uint64_t *)malloc(hh_cap * sizeof(uint64_t)); }
                uint64_t m = orc_kmer_hashes(bases + offsets[r], n, k, seed, hh, NULL), w = 0;
                if (have_ref) for (uint64_t j = 0; j < m; ++j) if (hh[j] <= max_ref) hh[w++] = hh[j];
                len = bottom_s_sort(hh, w, s, sk);
                keep = len;
            } else {
                len = orc_sketch_heap(bases + offsets[r], n, k, seed, s, sk);
                if (have_ref) while (keep < len && sk[keep] <= max_ref) ++keep;
            }
            if (keep) {
                lists[i] = (uint64_t *)malloc(keep * sizeof(uint64_t));
                memcpy(lists[i], sk, keep * sizeof(uint64_t));
            }
            lens[i] = (uint32_t)keep;
        }
        free(sk); free(hh);
    }
    uint64_t npairs = 0, empty = 0;
    uint64_t *poff = (uint64_t *)malloc(((uint64_t)nr + 1) * sizeof(uint64_t));
    for (uint32_t j = 0; j < nr; ++j) { poff[j] = npairs; npairs += lens[j]; empty += lens[j] == 0; }
    poff[nr] = npairs;
    /* 2. Q = sorted distinct in-range hashes of the block */
    uint64_t *Q = (uint64_t *)malloc((npairs + 1) * sizeof(uint64_t));
    for (uint32_t j = 0; j < nr; ++j) if (lens[j]) memcpy(Q + poff[j], lists[j], lens[j] * sizeof(uint64_t));
    qsort(Q, npairs, sizeof(uint64_t), cmp_u64);
    uint64_t nq = 0;
    for (uint64_t j = 0; j < npairs; ++j) if (j == 0 || Q[j] != Q[j - 1]) Q[nq++] = Q[j];
    const uint64_t W = ((uint64_t)n_genomes + 63) / 64;
    if (nq * W * 8 > mem_limit && nr > 1) { /* member matrix too large: halve the block (sketches are recomputed) */
        for (uint32_t j = 0; j < nr; ++j) free(lists[j]);
        free(lists); free(lens); free(poff); free(Q);
        uint32_t mid = r0 + nr / 2;
        int rc = fast_block(k, seed, s, stride, n_genomes, ref, col_len, max_ref, have_ref, bases, offsets, r0, mid, top_k, cum,
                            topk_idx, topk_sum, n_threads, mem_limit, index_min, stats);
        if (rc) return rc;
        return fast_block(k, seed, s, stride, n_genomes, ref, col_len, max_ref, have_ref, bases, offsets, mid, r1, top_k, cum,
                          topk_idx, topk_sum, n_threads, mem_limit, index_min, stats);
    }
    /* 3. pair -> index into Q */
    uint32_t *pq = (uint32_t *)malloc((npairs + 1) * sizeof(uint32_t));
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (i = 0; i < (int64_t)nr; ++i)
        for (uint32_t j = 0; j < lens[i]; ++j) {
            uint64_t h = lists[i][j], lo = 0, hi = nq;
            while (lo < hi) { uint64_t m = (lo + hi) / 2; if (Q[m] < h) lo = m + 1; else hi = m; }
            pq[poff[i] + j] = (uint32_t)lo;
        }
    /* 4. member[q][word]: bit (g & 63) of word g / 64 = Q[q] in column g; a thread owns whole words.
     * Small Q: the two-pointer merge of :428-437, Q against the column.  Large Q (hundreds of thousands of distinct hashes: walking
     * all of Q once per column is what a full-size batch spent its time in): the same set intersection through a bucket index over Q
     * -- first[h >> shift] = first position of Q at or above that bucket -- so every column hash looks at the few entries of its
     * bucket.  (tests/test_oracle.py runs both against orc_stream: the C0 / ragged cases take the merge, C1-sized ones the index.) */
    uint64_t *member = (uint64_t *)calloc(nq * W + 1, sizeof(uint64_t));
    uint64_t bits_set = 0;
    uint32_t *first = NULL;
    int shift = 0;
    uint64_t nbk = 0;
    if (nq >= index_min) {
        nbk = 1; while (nbk < 2 * nq) nbk <<= 1;
        while (shift < 63 && (Q[nq - 1] >> shift) >= nbk) ++shift;
        first = (uint32_t *)malloc((nbk + 2) * sizeof(uint32_t));
        uint64_t pos = 0;
        for (uint64_t bkt = 0; bkt <= nbk; ++bkt) {
            while (pos < nq && (Q[pos] >> shift) < bkt) ++pos;
            first[bkt] = (uint32_t)pos;
        }
    }
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads) reduction(+ : bits_set)
    for (i = 0; i < (int64_t)W; ++i) {
        uint32_t g1 = (uint32_t)((i + 1) * 64 < (int64_t)n_genomes ? (i + 1) * 64 : n_genomes);
        for (uint32_t g = (uint32_t)i * 64; g < g1; ++g) {
            const uint64_t *col = ref + (uint64_t)g * stride;
            uint64_t a = 0, b = 0, na = col_len[g];
            const uint64_t bit = 1ULL << (g & 63);
            if (first) {
                for (a = 0; a < na; ++a) {
                    const uint64_t h = col[a], bkt = h >> shift;
                    if (bkt >= nbk) break; /* the column ascends: everything from here on is above every query hash */
                    for (b = first[bkt]; b < first[bkt + 1]; ++b)
                        if (Q[b] == h) { member[b * W + (uint64_t)i] |= bit; ++bits_set; break; }
                }
                continue;
            }
            while (a < na && b < nq) { /* src/sketchy.rs:428-437 */
                if (Q[b] < col[a]) ++b;
                else if (Q[b] > col[a]) ++a;
                else { member[b * W + (uint64_t)i] |= bit; ++bits_set; ++a; ++b; }
            }
        }
    }
    free(first);
    /* 5. replay in read order, genomes split over threads */
    if (top_k == 0 || (!topk_idx && !topk_sum)) {
        /* table only: sum[g] += sum over q of multiplicity(q) * member[q][g] */
        uint32_t *mult = (uint32_t *)calloc(nq + 1, sizeof(uint32_t));
        for (uint64_t j = 0; j < npairs; ++j) ++mult[pq[j]];
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
        for (i = 0; i < (int64_t)W; ++i)
            for (uint64_t q = 0; q < nq; ++q) {
                uint64_t bits = member[q * W + (uint64_t)i];
                while (bits) { cum[(uint64_t)i * 64 + (uint64_t)__builtin_ctzll(bits)] += mult[q]; bits &= bits - 1; }
            }
        free(mult);
    } else {
        int T = n_threads;
        if ((uint64_t)T > W) T = (int)W;
        orc_row_t *loc = (orc_row_t *)malloc((uint64_t)T * nr * top_k * sizeof(orc_row_t));
        uint32_t *loc_n = (uint32_t *)calloc((uint64_t)T * nr, sizeof(uint32_t));
#pragma omp parallel for schedule(static, 1) num_threads(T)
        for (i = 0; i < (int64_t)T; ++i) {
            const uint64_t wa = W * (uint64_t)i / (uint64_t)T, wb = W * (uint64_t)(i + 1) / (uint64_t)T;
            const uint32_t ga = (uint32_t)(wa * 64), gb = (uint32_t)(wb * 64 < n_genomes ? wb * 64 : n_genomes);
            for (uint32_t j = 0; j < nr; ++j) {
                for (uint64_t p = poff[j]; p < poff[j + 1]; ++p) {
                    const uint64_t *row = member + (uint64_t)pq[p] * W;
                    for (uint64_t w = wa; w < wb; ++w) {
                        uint64_t bits = row[w];
                        while (bits) { ++cum[w * 64 + (uint64_t)__builtin_ctzll(bits)]; bits &= bits - 1; }
                    }
                }
                orc_row_t *best = loc + ((uint64_t)i * nr + j) * top_k;
                uint32_t nb = 0;
                for (uint32_t g = ga; g < gb; ++g) { /* ascending g: an equal sum never displaces an earlier genome */
                    uint64_t v = cum[g];
                    if (nb == top_k && !(v > best[nb - 1].sum)) continue;
                    uint32_t at = nb < top_k ? nb++ : top_k - 1;
                    while (at > 0 && v > best[at - 1].sum) { best[at] = best[at - 1]; --at; }
                    best[at].sum = v; best[at].idx = g;
                }
                loc_n[(uint64_t)i * nr + j] = nb;
            }
        }
        /* merge the ranges' rows: ranges ascend in genome index, each list is (sum desc, index asc) */
#pragma omp parallel for schedule(static) num_threads(n_threads)
        for (i = 0; i < (int64_t)nr; ++i) {
            uint32_t head[256];
            for (int t = 0; t < T; ++t) head[t] = 0;
            for (uint32_t o = 0; o < top_k; ++o) {
                int bt = -1;
                for (int t = 0; t < T; ++t) {
                    if (head[t] >= loc_n[(uint64_t)t * nr + (uint64_t)i]) continue;
                    const orc_row_t *c = loc + ((uint64_t)t * nr + (uint64_t)i) * top_k + head[t];
                    if (bt < 0) { bt = t; continue; }
                    const orc_row_t *b = loc + ((uint64_t)bt * nr + (uint64_t)i) * top_k + head[bt];
                    if (row_before(c->sum, c->idx, b->sum, b->idx)) bt = t;
                }
                const orc_row_t *b = loc + ((uint64_t)bt * nr + (uint64_t)i) * top_k + head[bt]++;
                if (topk_idx) topk_idx[((uint64_t)r0 + (uint64_t)i) * top_k + o] = b->idx;
                if (topk_sum) topk_sum[((uint64_t)r0 + (uint64_t)i) * top_k + o] = b->sum;
            }
        }
        free(loc); free(loc_n);
    }
    if (stats) { stats[0] += empty; stats[1] += npairs; stats[2] += nq; stats[3] += 1; stats[4] += bits_set; }
    for (uint32_t j = 0; j < nr; ++j) free(lists[j]);
    free(lists); free(lens); free(poff); free(Q); free(pq); free(member);
    return 0;
}

ORC_EXPORT int orc_stream_fast(uint32_t k, uint64_t seed, uint32_t s, uint32_t stride, uint32_t n_genomes,
                               const uint64_t *ref_hashes, const uint32_t *col_len,
                               const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                               uint32_t top_k, uint64_t *cum, uint32_t *topk_idx, uint64_t *topk_sum,
                               int n_threads, uint32_t block_reads, uint32_t index_min, uint64_t *stats) {
    if (top_k > n_genomes) return -1; /* the reference panics on [..top] (src/sketchy.rs:391) */
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (block_reads == 0) block_reads = 32768;
    uint64_t max_ref = 0;
    int have_ref = 0;
    for (uint32_t g = 0; g < n_genomes; ++g)
        if (col_len[g]) {
            uint64_t last = ref_hashes[(uint64_t)g * stride + col_len[g] - 1]; /* columns ascend (:416-418) */
            if (!have_ref || last > max_ref) max_ref = last;
            have_ref = 1;
        }
    for (uint32_t r0 = 0; r0 < n_reads; r0 += block_reads) {
        uint32_t r1 = n_reads - r0 > block_reads ? r0 + block_reads : n_reads;
        int rc = fast_block(k, seed, s, stride, n_genomes, ref_hashes, col_len, max_ref, have_ref, bases, offsets, r0, r1, top_k,
                            cum, topk_idx, topk_sum, n_threads, 3ULL << 30, index_min ? index_min : 4096u, stats);
        if (rc) return rc;
    }
    return 0;
}
