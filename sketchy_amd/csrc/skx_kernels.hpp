// skx_kernels.hpp -- launch wrappers of the gfx950 kernels (skx_kernels.hip).  Internal to libsketchy_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <stdlib.h>

namespace skx {
typedef unsigned long long u64;
typedef unsigned int u32;

// Experiment / test knobs: environment variables that pick a kernel variant, a stream arrangement, a pass size or -- the
// profiling aids SKX_SCAN_ABLATE and SKX_NO_FILTER -- change what the kernels compute.  They exist only in the
// -DSKX_EXPERIMENTS build (sketchy_amd/libsketchy_hip_exp.so, used by tests/ and tools/ through SKX_LIB_PATH); the
// product library reads no environment variable at all.
inline const char* knob(const char* name) {
#ifdef SKX_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

constexpr int kSketchCap = 2048;   // k-mers per read the one-wave-per-read sketcher holds in LDS
constexpr u32 kSegLen = 64;        // reads per ranking segment
constexpr int kRankWords = 8;      // genome words (x64 genomes) per ranking wave = one 64-byte sector of Mq per pair

// Species: several reference collections share one matrix, each padded to whole rank groups (512 genomes); every species
// has its own ranking.  Device arrays: g0[sp] first padded genome index, n[sp] real genomes, of_grp[rank group] species.
struct Species { const u32* g0; const u32* n; const u32* of_grp; u32 n_sp; };

// reference upload: src / eff_len start at the chunk's first genome, which sits at padded index pad_base
void launch_ref_tile(hipStream_t st, const u64* src, const u32* eff_len, u64* dst, u32 s, u32 pad_base, u32 g_count);
void launch_band_bounds(hipStream_t st, const u64* mat, u32 s, u32 n_tiles, u32 rb, u32 n_bands, u64* lo, u64* hi);

// Long reads of a production batch are split over waves (skx_kernels.hip, "long reads: split over waves"): device tables
// built by launch_batch_check and consumed by launch_sketch.  Capacities: a read is long beyond long_read_split() raw
// bytes and takes ceil(length / kSketchCap) segment slots, so a batch of at most B bases needs at most
// B / long_read_split() + 1 list entries and 5 B / (4 kSketchCap) + 1 slots (the caller sizes them from max_batch_bases).
struct LongReads {
    u32* list;     // [long_cap] read index
    u32* seg0;     // [long_cap] first segment slot of the read
    u32* seg_tab;  // [segs_cap] position in `list` of the read a segment slot belongs to
    u32* seg_cnt;  // [segs_cap] hashes the segment kept
    u64* seg_h;    // [segs_cap][long_read_seg_slots()]
    u32 segs_cap, long_cap;
};
u32 long_read_split();
u32 long_read_seg_slots();
u32 pool_row_fixed();  // entries of a read's fixed slot at the start of the row pool
u32 chk_words();  // u32 words of a `chk` block: 16 flags / counters + the row pool's bump counters (one cache line each)

// k-mer prefilter (k = 16): a two-probe blocked Bloom table over the 2-bit code of the CANONICAL k-mer, holding every
// k-mer whose MurmurHash3 value passes the reference's membership filter (<= max_ref and bit set) -- built once per
// reference by enumerating all 4^16 / 2 canonical 16-mers (~10 ms of VALU work on an MI355X).  A read with at most s k-mer
// windows cannot be truncated (its sketch is ALL its distinct hashes), so only k-mers that can end up as (read, hash) pairs
// matter: the sketcher looks every window up in this table (one 4-byte gather) and runs murmur3 only on the survivors,
// compacted through the wave's LDS -- ~30 instead of ~100 VALU instructions per 64 windows.  Exact: the table has no
// false negatives; false positives are hashed and then dropped by the in-range / membership tests as before.
struct KmerFilter { const u32* words; u32 shift; };  // word = (code * kKmerMix) >> shift; bits (m & 31), ((m >> 5) & 31)
// count (pass 1) / insert (pass 2) the canonical 16-mers whose hash passes (max_ref, filt); *n_keys accumulates in pass 1
void launch_kmer_filter_build(hipStream_t st, u64 seed, u64 max_ref, const u64* filt, u32 filt_shift, u32* n_keys /* pass 1 */,
                              u32* words /* pass 2, zeroed */, u32 shift);

// sketching: every read of the batch, any length -- one wave per read (256 hash slots, then 2048 for the reads that
// overflow); what still does not fit (and, for full sketches, every read with more than kSketchCap k-mers) is left on the
// device-side list `big` for launch_sketch_block (one block per read).  `retry` / `big`: [0] = count, zero on entry.
// n_bases: bytes the caller vouches for from offsets[0] on; a read reaching outside is skipped and flagged in chk[6]
// (chk may be NULL).  long_reads (production batches, with chk): the tables launch_batch_check filled for this batch.
// Returns the launch status (it also opts the big-LDS kernels in, once per device).
hipError_t launch_sketch(hipStream_t st, const uint8_t* bases, const u64* offsets, u32 n_reads, u32 k, u64 seed, u32 s,
                         u64 max_ref, bool inrange_only, u64* out_sk, u32 sk_stride, u32* out_len, u32* out_cnt_in,
                         const u64* filt /* membership filter applied in inrange_only mode, or NULL */, u32 filt_shift,
                         u32* retry /* [1 + n_reads] */, u32* big /* [1 + n_reads] */, u64 n_bases, u32* chk,
                         int leave_room /* 0 no, 1 a scan overlaps the start, 2 a scan runs beside most of it */, bool packed = false,
                         const LongReads* long_reads = nullptr, const KmerFilter* kmer_filter = nullptr /* production, k = 16 */,
                         int phase = 3 /* production batches: 1 = the main kernel only, 2 = only the list walks behind it, 3 = both */,
                         u32 pool_cap = 0 /* sk_stride == 0 (production): out_sk is a pool of this many entries, out_len receives
                                             every row's start, chk[6] |= 4 on overflow */,
                         u32 pool_fixed = 0 /* its first entries: pool_row_fixed() per read, taken without reservation */);
// the block sketcher for the n_big reads launch_sketch left on `big` (big[1 ..]); the caller reads the count back first
hipError_t launch_sketch_block(hipStream_t st, const uint8_t* bases, const u64* offsets, const u32* big, u32 n_big, u32 k, u64 seed,
                               u32 s, u64 max_ref, bool inrange_only, u64* out_sk, u32 sk_stride, u32* out_len, u32* out_cnt_in,
                               const u64* filt, u32 filt_shift, bool packed = false, u32* chk = nullptr, u32 pool_cap = 0,
                               u32 pool_fixed = 0);
// exclusive scan of n counts (out[i] = sum of in[0..i)); bsum: [ceil(n / 1024)] scratch
void launch_count_scan(hipStream_t st, const u32* in, u32* out, u32 n, u32* bsum);

// Rare-hash index of a reference (skx_kernels.hip, "rare-hash index"): open-addressing table over the distinct reference hashes
// (key, all-ones = empty; mask = slots - 1), cnt = genomes that hold the hash, off = start of its genome list in post (padded genome
// indices) -- or 0xFFFFFFFF for a hash held by more genomes than the policy "rare_hash_genomes": those stay with the scan.
struct RareIndex { const u64* key; const u32* off; const u32* cnt; const u32* post; u32 mask;
                   // long lists (more than 8 genomes) also as BIT ROWS: mlong[lid][n_gw] u64, bit g of a row = genome g holds the hash;
                   // lid[key slot] = row (0xFFFFFFFF: none), lslot[row] = key slot.  NULL when there was no room for the rows.
                   const u64* mlong = nullptr; const u32* lid = nullptr; const u32* lslot = nullptr; u32 n_gw = 0;
                   // ... and the same bits TRANSPOSED: mlongT[genome][n_lw] u64, bit r of a genome's row = bit row r holds the genome
                   // (n_lw = ceil(rows / 64) words; NULL when there was no room): which of a batch's long-list rows meet a candidate is
                   // then an AND of the candidate's row with the batch's row mask, not a walk over every row (launch_cand_hit)
                   const u64* mlongT = nullptr; u32 n_lw = 0;
                   // ... and most long lists as (pattern, exceptions) (round 6; skx_kernels.hip, "long lists as (pattern, exceptions)"):
                   // prec[bit row][pat_record_words()] = {pattern or 0xFFFFFFFF, n, n x (genome | not-on-the-list << 31)}; pat_rep[pattern] =
                   // the bit row whose list IS the pattern; pm[word of 64 patterns][n_pad] the patterns' bits in M's layout; d_npat[0] = n_pat
                   const u32* prec = nullptr; const u32* pat_rep = nullptr; const u64* pm = nullptr; const u32* d_npat = nullptr; u32 n_pat = 0;
                   // the STATIC dense dictionary (round 6): qs[0 .. n_sd) = every hash the scan can be asked for (held by more genomes than
                   // the index lists, + the lifted hashes), ascending; a pass's dense rows are ITS rows (launch_classify); NULL: per-pass dictionaries
                   // (several species: the species' sorted segments one after the other, each starting on a multiple of 64 rows, padded with
                   // all-ones; srow[key slot] = the hash's row; the lifted hashes sorted in the tail [tail0, n_sd))
                   const u64* qs = nullptr; u32 n_sd = 0; const u32* srow = nullptr; u32 tail0 = 0; };
void launch_collect_dense(hipStream_t st, const u64* key, const u32* off, u64 slots, u64* out, u32* out_slot, u32* n, u32 cap);
void launch_window_seg(hipStream_t st, const u64* lo, const u64* hi, u32 n_bt, u32 n_tiles, const u64* q, const u32* seg, const u32* grp_sp, u32* win);
// pattern rows of a pass: hist[b][hist_stride] = occurrences of every pattern among batch b's pairs, gain_x[b][n_pad] += / -= the rows'
// counts at their exceptions (wrapping), nprow[b] (one counter per batch, like LongRows::nlrow) rows listed from the END of lrow[b]
struct PatRows { u32* hist; u32 hist_stride; u32* gain_x; u32* nprow; };
void launch_list_sig(hipStream_t st, const u32* lslot, u32 n_long, const u32* off, const u32* cnt, const u32* post, u32* sig, u64* content);
void launch_pat_exceptions(hipStream_t st, const u64* mlong, u32 n_gw, u32 n_long, const u32* pat_of, const u32* pat_rep, u32* prec, u32* n_done);
void launch_pat_matrix(hipStream_t st, const u64* mlong, const u32* pat_rep, u32 n_pat, u32 n_gw, u64* pm, u32 n_pad);
u32 pat_record_words();
u32 pat_words_max();  // genome words of a compact problem the pattern path handles (more species than that: the stream does not use the patterns)
// long-list rows of a pass, per batch: lrow[b][i] = {bit row, sparse row of the pass's matrix} (nlrow[b] of them, lrow_stride apart)
struct LongRows { uint2* lrow; u32* nlrow; u32 lrow_stride; u64* inb = nullptr; u32 n_lw = 0; };  // inb[b][n_lw] (or NULL; zero on entry): bit r = bit row r is on batch b's list  // (nlrow: one counter per batch, pass_counter_bytes() in all)
u32 pass_counter_bytes();  // size of the per-batch counter arrays of a pass (nlrow, nqc: one cache line per batch)
void launch_mlong_build(hipStream_t st, const u32* lslot, u32 n_long, const u32* off, const u32* cnt, const u32* post, u64* mlong, u32 n_gw);
void launch_long_rows(hipStream_t st, const u32* sslot, const u32* n_d, u32 rows_bound, const u32* cnt, u32 row_stride, u32 n_b, const LongRows& lr);
// gain_l[b][g] (zero on entry) += sum over batch b's long-list rows of cnt x bit: bit-sliced counters over the rows' 64-bit words
void launch_gain_long(hipStream_t st, const LongRows& lr, const RareIndex& ri, const u32* n_d, const u32* cnt, u32 row_stride, u32 n_b, u32 n_pad,
                      u32* gain_l, u32 walk_scale = 1);
// the candidates of every batch by genome word: cw[b][w] their bits, cbase[b][w] the slot of the word's first one, cwl[b][.] / ncwl[b] the
// words that hold any (cw, ncwl zero on entry; cbase all-ones)
void launch_mlong_transpose(hipStream_t st, const u64* mlong, u32 n_long, u32 n_gw, u64* mlongT, u32 n_lw);
// hit[b][n_lw] (zero on entry) |= (row of candidate g in the transposed bit rows) & inb[b], for every candidate of every batch: the
// long-list rows of batch b that hold any of its candidates -- cand_long_kernel then only looks at those
void launch_cand_hit(hipStream_t st, const u32* cand, u32 n_pad_c, u32 n_b, const u32* bad, const RareIndex& ri, const u64* inb, u64* hit);
void launch_cand_words(hipStream_t st, const u32* cand, u32 n_pad_c, u32 n_b, u32 n_gw, u64* cw, u32* cbase, u32* cwl, u32* ncwl);
void launch_cand_long(hipStream_t st, const LongRows& lr, const RareIndex& ri, const u32* n_d, const u64* cw, const u32* cbase, const u32* cwl,
                      const u32* ncwl, u32 n_pad_c, u32* bad, u32 n_b, u32* nqc, u32* smap, u32 smap_stride, u64* mqc, size_t mqc_stride,
                      u32 rows_c, u64* rowany_c, u32 rowany_stride, u32* grp_any_c, u32 n_grp_c, u32 walk_scale = 1, const u64* hit = nullptr /* launch_cand_hit's, or NULL: every listed row is tested */);
// build, two passes over the tiled matrix (n_elems = n_tiles * s * 256): count (key / cnt zeroed: all-ones / 0; *overflow raised when
// the table is too small), then -- offsets from the counts, cursor zeroed -- fill
void launch_rare_count(hipStream_t st, const u64* mat, u64 n_elems, u64* key, u32* cnt, u32 mask, u32* overflow,
                       u64* spmask = nullptr /* [slots], zero on entry: bit sp = a genome of species sp holds the key */, const u32* grp_sp = nullptr, u32 s = 0);
void launch_rare_fill(hipStream_t st, const u64* mat, u64 n_elems, u32 s, const u64* key, const u32* off, u32* cursor, u32* post, u32 mask);
// a pass's dictionary split into the hashes the scan looks for (qd ascending, n_d[0] of them: rows [0, n_d[0]) of the bit matrix)
// and the others (n_d[1]; rows from n_d[2] on, sslot[2 i], [2 i + 1] = start and length of row i's genome list); qrow[position in q] = row.
// q_bound: host's upper bound of *n_q (sizes the grids); qinfo / qloc: [q_bound] scratch; bsum: [q_bound / 1024 + 1] scratch;
// h_words (page-locked, or NULL): [0] = *n_q, [1] = n_d[0]
void launch_classify(hipStream_t st, const u64* q, const u32* n_q, u32 q_bound, const RareIndex& ri, u32* qinfo, u32* qloc, u32* bsum,
                     u64* qd, u32* n_d, u32* qrow, u32* sslot, u32* h_words);
// bits of the rows behind the dense ones, from the genome lists: atomicOr into m_bits (and *m_dirty = 1)
// the rare rows straight into the group-major matrix (rows n_d[2] .. of Mq, their words of rowany, grp_any): references with bit rows
// for every list longer than 8 genomes (RareIndex::mlong) -- then M holds the dense rows only and launch_transpose_bits is given n_d + 2
// ext = {row stride, clean rows} of the buffer set's (Mq, rowany) pair (device; see rare_to_mq_kernel): read, then brought up to date
void launch_rare_to_mq(hipStream_t st, const u32* sslot, const u32* n_d, const RareIndex& ri, u64* mq, u32 nq_rows, u32 n_pad, u64* rowany,
                       u32* grp_any, u32 rows_bound, const u32* only_if, u32* ext);
u32 mq_any_stride();  // the stride value of arrays that were zeroed at allocation (clean under any stride)
void launch_sparse_fill(hipStream_t st, const u32* sslot, const u32* n_d, const RareIndex& ri, u64* m_bits, u32 n_pad, u32* m_dirty,
                        u32 rows_bound, const u32* only_if = nullptr /* device flag: run only when it is non-zero */);
// n_d = {dense rows, other rows, first other row (dense rows rounded up to 64), rows in all} -- what launch_classify leaves; for a
// reference without the index: everything dense, rows = positions in q
void launch_nd_from_nq(hipStream_t st, const u32* n_q, u32* n_d, u32* h_words);

// ---- the table without the ranking, and the candidates of a batch (skx_kernels.hip, "the table without the ranking")
static const u32 kPassBatchesMax = 8;          // batches of a pass (stream_coalesce)
static const u32 kCandCap = 1024;              // candidates per species the compact ranking takes (two rank groups)
static const u32 kCandRows = 131072;           // rows of a compact bit matrix (dense rows + the rare rows some candidate holds)
struct PassBatches { u32 n; u32 p_off[kPassBatchesMax + 1]; };  // pairs [p_off[b], p_off[b + 1]) of the pass's lists are batch b's
// cnt[b][row] (zero on entry; row_stride entries per batch) = occurrences of the row among batch b's pairs
void launch_pass_hist(hipStream_t st, const u32* pair_q, const PassBatches& pb, u32* cnt, u32 row_stride);
// gain[b][g] (zero on entry) += sum over rows of cnt[b][row] * (row's bit for g): dense rows from m_bits (BEFORE the transpose
// re-zeroes it), the others from the genome lists (ri / sslot, or NULL)
// gain_s (zero on entry, [n_b][n_pad] entries gain_sparse_stride() words apart): the rare rows' part
void launch_gain_dense(hipStream_t st, const u64* m_bits, const u64* m_int /* or NULL */, u32 n_pad, const u32* n_d, u32 rows_bound, const u32* cnt,
                       u32 row_stride, u32 n_b, u32* gain, const u32* segw = nullptr /* [2 n_sp] word range of every species' rows (static dictionary) */,
                       const u32* grp_sp = nullptr);
// (lr: also lists the rows with a bit row per batch, nlrow zero on entry; walk_scale: multiplies the workgroups of the list walk --
// a pass with nothing beside it may fill the chip)
// (pr: the rows whose list is pattern + exceptions add to hist / gain_x and are listed for launch_cand_pat_map; hist, gain_x, nprow zero on entry)
void launch_gain_sparse(hipStream_t st, const u32* n_d, u32 rows_bound, const u32* cnt, u32 row_stride, u32 n_b, u32 n_pad, u32* gain_s,
                        const u32* sslot, const RareIndex& ri, const LongRows* lr, u32 walk_scale, const PatRows* pr = nullptr);
// pattern rows of the compact problems: nqc starts at n_pat rounded up to 64 (the mapped rows come behind the patterns' rows);
// launch_cand_pat_rows: row n_d[2] + p of mqc[b] = pattern p at batch b's candidates, pcw[b][p][n_grp_c * 8] the same words;
// launch_cand_pat_map: smap of every listed pattern row -> its pattern's row, or a new row (pattern's words with the exceptions that are
// candidates flipped)
void launch_pat_nqc_init(hipStream_t st, u32* nqc, u32 v);
void launch_cand_pat_rows(hipStream_t st, const RareIndex& ri, const u32* n_d, const u32* candmask, const u32* candslot, u32 n_pad, u32* bad, u32 n_b,
                          u64* pcw, u64* mqc, size_t mqc_stride, u32 rows_c, u64* rowany_c, u32 rowany_stride, u32* grp_any_c, u32 n_grp_c);
void launch_cand_pat_map(hipStream_t st, const LongRows& lr, const PatRows& pr, const RareIndex& ri, const u32* n_d, const u32* candmask,
                         const u32* candslot, u32 n_pad, u32* bad, u32 n_b, u32* nqc, u32* smap, u32 smap_stride, const u64* pcw, u64* mqc,
                         size_t mqc_stride, u32 rows_c, u64* rowany_c, u32 rowany_stride, u32* grp_any_c, u32 n_grp_c, u32 rows_bound, u32 walk_scale = 1);
u32 gain_sparse_stride();
// tab[0] = prev, tab[b + 1] = tab[b] + gain[b]   ([n_b + 1][n_pad])
void launch_pass_tables(hipStream_t st, const u64* prev, const u32* gain, const u32* gain_s /* or NULL */, const u32* gain_l /* or NULL */, u32 n_b,
                        u32 n_pad, u64* tab);
// per (batch, species): the genomes whose value at the end of the batch reaches the top_k-th best value at its start, in reference
// order: cand[(b n_sp + sp) cap + i], candslot[b][g] (0xFFFFFFFF: none), tabc[b][sp cap + i] start values, ncand[b n_sp + sp],
// bad[b] (zero on entry) |= 1 when a species has more than cap
// candmask[g] (zero on entry) |= 1 << b for every candidate g of batch b
void launch_cand_select(hipStream_t st, const u64* tab, u32 n_pad, const Species& sp, u32 n_b, u32 top_k, u32 cap, u32* cand, u32* candslot,
                        u64* tabc, u32* ncand, u32* bad, u32* candmask);
// candidates a batch would have had (t0 / t1: the table before / after it): the largest count over the species -> h_out[0], seq -> h_out[1]
void launch_cand_count(hipStream_t st, const u64* t0, const u64* t1, const Species& sp, u32 top_k, u32* h_out, u32 seq);
// mc[b][w][c] = m_bits[w][cand[b][c]] (dense words; n_pad_c = n_sp * cap columns, words_c words per batch)
void launch_cand_gather_m(hipStream_t st, const u64* m_bits, const u64* m_int /* or NULL */, u32 n_pad, const u32* n_d, u32 rows_bound, const u32* cand,
                          u32 n_pad_c, const u32* bad, u32 n_b, u64* mc, u32 words_c);
// rare rows of the compact problems: rows behind the dense ones for the hashes some candidate holds (nqc[b] of them, smap[b][.]),
// their bits into mqc[b] / rowany_c[b] / grp_any_c[b] (all zero on entry); bad[b] |= 2 when they do not fit rows_c
void launch_cand_sparse(hipStream_t st, const u32* sslot, const u32* n_d, u32 rows_bound, const RareIndex& ri, const u32* candmask,
                        const u32* candslot, u32 n_pad, u32* bad, u32 n_b, u32* nqc, u32* smap, u32 smap_stride, u64* mqc, size_t mqc_stride,
                        u32 rows_c, u64* rowany_c, u32 rowany_stride, u32* grp_any_c, u32 n_grp_c, u32 walk_scale = 1);
// mode[b] = 1 compact / 0 everything, *any_full, nqc_total[b] = rows of the compact problem; h_pub (page-locked): [b] mode,
// [8 + b] largest candidate count, [16] any_full, [18] = n_b, [17] = seq (written last)
void launch_cand_publish(hipStream_t st, const u32* bad, u32 force_full, const u32* ncand, const u32* nqc, const u32* n_d, u32 n_b, u32 n_sp,
                         u32 rows_c, u32* mode, u32* any_full, u32* nqc_total, u32* h_pub, u32 seq);
void launch_m_clear(hipStream_t st, u64* m_bits, u64* m_int /* or NULL */, u32 n_pad, const u32* n_d, const u32* any_full);
void launch_cand_pair_rows(hipStream_t st, const u32* pair_q, u32 n_pairs, const u32* n_d, const u32* smap, u32 rows_c, u32* pair_qc);
void launch_cand_rows_back(hipStream_t st, u32* out_idx, u32 n_reads, u32 n_sp, u32 top_k, const u32* cand, u32 cap, const u32* g0);

// dictionary
// qrow (launch_classify) != NULL: pair_q receives ROWS of the bit matrix instead of positions in q
void launch_pair_q(hipStream_t st, const u64* pair_h, u32 n_pairs, const u64* q, const u32* n_q, u32* pair_q, const u32* qrow /* or NULL */,
                   const u32* bbase, const u32* btot, u64 max_ref /* launch_dict_rest's, of the same pass */);
// Dictionary of a pass, in two steps.  (1) launch_dict_insert gathers the pairs of reads [r_begin, r_end) (pair_h, pair_r)
// and inserts every pair hash into the hash set -- one wave per read, no knowledge of the pair count needed: it can be
// queued right behind the sketcher; if the reads have more than pair_cap pairs it does nothing.  (2) launch_dict_rest
// turns the set into q / n_q, the sorted distinct dictionary, and empties it.
// ht: hash set of ht_slots (power of two, >= 2 x pairs) u64, all-ones between passes; slot_off: [ht_slots] scratch;
// bcount: [dict_buckets()] zero between passes; bbase: [dict_buckets()]; btot: [129]; ctr: [4] zero between passes
// (ctr[1]: the all-ones hash was seen, ctr[2]: distinct keys inserted so far)
// base: device words whose sum is the number of pairs already in the lists (a batch sharing the pass of the batches before it:
// their pair counts are only known on the device when this is queued)
static const int kPairBaseMax = 7;
struct PairBase { const u32* p[kPairBaseMax] = {}; };
void launch_dict_insert(hipStream_t st, const u64* sk, u32 sk_stride, const u32* poff, u32 r_begin, u32 r_end, u32 p_base,
                        u64* pair_h, u32* pair_r, u64* ht, u32 ht_slots, u32* ctr, u32 pair_cap,
                        const u32* row_off = nullptr /* sk_stride == 0: row r starts at sk + row_off[r] */,
                        PairBase base = PairBase());
void launch_dict_rest(hipStream_t st, u64* ht, u32 ht_slots, u64 max_ref, u32* slot_off, u32* bcount, u32* bbase, u32* btot,
                      u32* ctr, u64* q, u32* n_q);
u32 dict_buckets();
void launch_window(hipStream_t st, const u64* lo, const u64* hi, u32 n_bt, const u64* q, const u32* n_q, u32* win,
                   u32* h_nq /* page-locked host word that receives *n_q, or NULL */);
void launch_exceptions(hipStream_t st, const u32* exc_g, const u64* exc_h, u32 n_exc, const u64* q, const u32* n_q,
                       u64* m_bits, u32 n_pad, u32* m_dirty /* raised when a bit was set, or NULL */, const u32* qrow = nullptr,
                       u32 row0 = 0 /* q[0]'s row */, u32 n_fixed = 0 /* n_q == NULL: entries of q */);

// scan + transpose
// lean: scan_lean_kernel (sparse dictionaries: every production pass) -- every (band, tile) block ORs the words of its slice into
// m_bits when its band is done (one coalesced atomicOr per non-zero word) and raises *m_dirty; !lean: scan_kernel variants into
// m_bits / m_int.  Experiments build only: hbuf != NULL with into_m == false = round 3's slab form (every block stores its words
// into its own slab hbuf[(band * n_tiles + tile) * scan_lean_words() * 256 ...]; scan_lean_wants_slabs() says whether a knob asks
// for it, and the stream then allocates the slabs).
bool scan_lean_applies(u32 n_bands, bool split, bool big_table);
u32 scan_lean_words();
bool scan_lean_wants_slabs();
bool scan_lean_into_m();
// run > 0: scan_run_kernel -- one workgroup per `run` consecutive bands of a tile, results OR-ed into m_bits (no slabs, no hbuf);
// windows of up to scan_run_cap() entries in one pass over the rows
u32 scan_run_cap();
void launch_scan(hipStream_t st, const u64* mat, u32 s, u32 n_tiles, u32 rb, u32 n_bands, const u64* q, const u32* win,
                 u64* m_bits, u64* m_int, u32 n_pad, bool big_table, bool lean, u64* hbuf, u32* m_dirty, bool into_m = true, u32 run = 0, bool big_slices = false);
u32 scan_lean_cap(bool big);  // entries of a (band, tile) slice the lean kernel's one-pass probe holds (big: the instance for passes with large slices)
// wb[w * n_tiles + t] = (first | last << 16) band of tile t whose slice can reach query word w (first > last: none)
// lo != NULL: also computes the windows (launch_window's work) first: one launch instead of two in front of the scan
void launch_word_bands(hipStream_t st, u32* win, u32 n_tiles, u32 n_bands, const u32* n_q, u32* wb, const u64* lo = nullptr,
                       const u64* hi = nullptr, const u64* q = nullptr, u32* h_nq = nullptr);
// also re-zeroes m_bits / m_int; words beyond *n_q are skipped; grp_any[rank group] (zero on entry) receives the number of
// query rows that hold a bit for some genome of the group -- the ranking kernels skip groups with none (grp_any arguments
// below) and compact the pairs of groups with few (rowany)
// hbuf != NULL: also ORs the lean kernel's slabs in (wb from launch_word_bands, win = the windows) and reads m_bits only
// when *m_dirty != 0
void launch_transpose_bits(hipStream_t st, u64* m_bits, u64* m_int, u32 n_pad, u32 n_words, u64* mq,
                           const u32* n_q, u32* grp_any, const u64* hbuf, const u32* wb, const u32* win, u32 n_tiles,
                           const u32* m_dirty, u64 nq_est /* the host's estimate of the dictionary size: sizes the grid */,
                           u64* rowany /* [rank groups][n_words]: bit r of word w = row 64 w + r holds a bit in the group; or NULL */,
                           const u32* only_if = nullptr /* device flag: run only when it is non-zero */,
                           bool keep_m = false /* do NOT re-zero m_bits (static dense rows kept for later passes: policy reuse_membership) */);
// chk[0..5], [9] (zero on entry): non-monotonic marker, long-read count, offsets[0], offsets[n_reads], segment count;
// long_reads != NULL: also lists the batch's long reads and their segments (chk[6] |= 2 if they do not fit the tables)
void launch_batch_check(hipStream_t st, const u64* offsets, u32 n_reads, u64 n_bases, u32* chk, u32* cnt_tail /* zeroed */,
                        const LongReads* long_reads);
// h_pub (page-locked host memory, 16 words): [0..7] = chk (then zeroed, as is retry[0]), [8] = *total_pairs, [15] = seq last
// ([7] = number of reads the block sketcher took; `big` is re-armed like `retry`)
// [10] = distinct keys in the hash set dict_ctr belongs to (the speculative gather's |Q|; 0xFFFFFFFF without dict_ctr)
// n <= 64 words, device memory -> page-locked (coherent) host memory, written by a kernel: never blocks the queueing thread
// (reads AND zeroes d_src: a counter block that is published once per use needs no memset of its own)
void launch_store_host_words(hipStream_t st, u32* h_dst, u32* d_src, u32 n);
void launch_publish(hipStream_t st, u32* chk, u32* retry, u32* big, const u32* total_pairs, u32* h_pub, u32 seq,
                    const u32* dict_ctr /* counters of the set the speculative gather filled, or NULL */);
// membership filter over the union of the reference hashes (blocked Bloom filter, skx_common.hpp: filter_mask / filter_hit);
// count != NULL: `words` is a plain bitmap over v >> shift instead and *count receives the bits newly set (distinct values)
void launch_filter_build(hipStream_t st, const u64* vals, u64 n, u32 shift, u64* words, bool markers_are_values,
                         unsigned long long* count = nullptr);
void launch_filter_apply(hipStream_t st, u64* sk, u32 sk_stride, u32* cnt, u32 n_reads, const u64* bits, u32 shift);

// ranking
void launch_seg_sum(hipStream_t st, const u32* pair_q, const u32* poff, u32 p_base, u32 r_begin, u32 n_reads,
                    u32 seg_len, const u64* mq, u32 n_pad, u32 nq_rows, u32* inc, const u32* grp_any,
                    u32* csum_raw /* [ceil(n_seg / 16)][n_pad] chunk sums, zero on entry; NULL: not wanted */,
                    const u64* rowany /* of launch_transpose_bits, or NULL */, const u32* n_q, const Species& sp,
                    const u64* gmax = nullptr /* != NULL: only the (chunk, rank group)s that can hold a candidate (launch_seg_prefix part 1) */,
                    const u64* lead_val = nullptr);
// the chunk sums alone (csum_raw zero on entry): a workgroup per (rank group, chunk), one extraction per quarter chunk
void launch_chunk_sum(hipStream_t st, const u32* pair_q, const u32* poff, u32 p_base, u32 r_begin, u32 n_reads,
                      const u64* mq, u32 n_pad, u32 nq_rows, const u32* grp_any, u32* csum_raw, const u64* rowany, const u32* n_q,
                      const Species& sp);
// prune_top_k > 0 (1..rank_topk_fast_max()): also find the first prune_top_k genomes as each chunk of 16 segments begins
// (leader [n_chunks * k], lead_val [n_chunks]) and every half rank group's best value per chunk boundary (gmax
// [(n_chunks + 1) * n_pad / 256]); start values are then only written for (chunk, group)s that can hold a candidate.
// part: 0 = everything; 1 = what needs only the chunk sums; 2 = what needs the per-segment increments (after part 1)
void launch_seg_prefix(hipStream_t st, const u32* inc, u32 n_seg, u32 n_pad, const Species& sp, const u64* cum_in, u64* cum_out,
                       u32* rel /* [n_seg][n_pad]: segment start values minus cum_in */, u32* csum, u32* csum_raw, u32 prune_top_k,
                       u32* leader /* [n_chunks * n_sp * k] */, u64* lead_val /* [n_chunks * n_sp] */, u64* gmax,
                       u64* part_sum, u32* part_idx /* [n_chunks * n_sp * rank_leader_parts() * k] scratch */,
                       const u32* grp_any, unsigned char* live /* [n_seg][n_pad / 64], pruned rankings only; else NULL */,
                       u64* lead_seg /* [n_seg][n_sp] scratch */, int part = 0,
                       u32* live_ctr = nullptr /* [2], pruned rankings: += (chunk, half group)s that can hold a candidate / tested (a sample) */,
                       hipEvent_t ev_table = nullptr /* recorded right behind chunk_prefix: cum_out is complete (parts 0 and 1) */);
u32 rank_leader_parts();
void launch_rank_seg(hipStream_t st, const u32* pair_q, const u32* pair_r, const u32* poff, u32 p_base, u32 r_begin,
                     u32 n_reads, u32 seg_len, const u64* mq, u32 n_pad, u32 nq_rows, const Species& sp, const u64* cum_in,
                     const u32* rel, u32 top_k, u64* cand_sum, u32* cand_idx, const u32* grp_any);
void launch_rank_seg_top1(hipStream_t st, const u32* pair_q, const u32* pair_r, const u32* poff, u32 p_base, u32 r_begin,
                          u32 n_reads, const u64* mq, u32 n_pad, u32 nq_rows, const Species& sp, const u64* cum_in,
                          const u32* rel, u64* best_sum, u32* best_idx, const u32* inc, const u32* leader, const u64* gmax,
                          const u64* lead_val, const u32* grp_any, const unsigned char* live, unsigned char* has,
                          const u64* rowany /* of launch_transpose_bits, or NULL */, const u32* n_q);
// 2 <= top_k <= rank_topk_fast_max(): pruned, one wave per (rank group, segment);
// cand_sum / cand_idx[(r * n_grp + grp) * top_k + j]
void launch_rank_seg_topk(hipStream_t st, const u32* pair_q, const u32* pair_r, const u32* poff, u32 p_base, u32 r_begin,
                          u32 n_reads, const u64* mq, u32 n_pad, u32 nq_rows, const Species& sp, const u64* cum_in,
                          const u32* rel, u32 top_k, u64* cand_sum, u32* cand_idx, const u32* inc, const u32* leader,
                          const u64* gmax, const u64* lead_val, const u32* grp_any, const unsigned char* live, unsigned char* has);
u32 rank_topk_fast_max();
// rows come out per (read, species) with genome indices local to the species: out[((out_r0 + r) * n_sp + sp) * top_k + j]
void launch_top1_merge(hipStream_t st, const u64* best_sum, const u32* best_idx, u32 n_reads, u32* out_idx, u64* out_sum,
                       u32 out_r0, const Species& sp, const unsigned char* has, u32 n_grp);
// n_units candidates-units per read (rank groups: per_grp = 1; genome words: per_grp = kRankWords)
void launch_topk_merge(hipStream_t st, const u64* cand_sum, const u32* cand_idx, u32 n_reads, u32 n_units, u32 per_grp,
                       u32 top_k, u32* out_idx, u64* out_sum, u32 out_r0, const Species& sp, const unsigned char* has /* [segments][rank groups] of the pruned kernels, or NULL */);
void launch_rank_table(hipStream_t st, const u64* cum, const Species& sp, u32 top_k, u32* out_idx /* [n_sp][top_k] */, u64* out_sum);
// n_real real genomes (species concatenated); real2pad[g] = padded index
void launch_shared_debug(hipStream_t st, const u32* pair_q, const u32* poff, u32 p_base, u32 r_begin, u32 n_reads,
                         const u64* mq, u32 nq_rows, u32 n_real, const u32* real2pad, u32* shared, u32 out_r0);
void launch_add_table(hipStream_t st, u64* cum, const u64* add, u32 n_real, const u32* real2pad);
void launch_gather_table(hipStream_t st, const u64* cum, u64* out, u32 n_real, const u32* real2pad);

}  // namespace skx
