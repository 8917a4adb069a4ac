/*
 * sketchy_hip.h -- C ABI of libsketchy_hip.so: the MI355X (gfx950) implementation of
 * sketchy's streaming read-vs-reference MinHash path.
 *
 * The reference (esteinig/sketchy v0.6.0) has no FFI/plugin interface; it is one Rust
 * binary.  The boundary below sits exactly at the seam its hot loop uses internally
 * (paths are under the reference tree):
 *
 *   skx_ref_create         <- the in-RAM reference collection `Vec<Sketch>` returned by
 *                             Sketchy::_read_sketch (src/sketchy.rs:497-536) and consumed at
 *                             :337-339; k / seed / s are the sketch params of :82, :520-527
 *   skx_stream_create      <- state of Sketchy::_sum_of_shared_hashes (src/sketchy.rs:326-327):
 *                             `sum_of_shared_hashes: Vec<u64>` + read counter; `top` is
 *                             PredictConfig::top (src/sketchy.rs:43-50)
 *   skx_stream_push        <- one or more iterations of the loop body src/sketchy.rs:328-354:
 *                             create_sketcher/process/to_vec (:331-335, finch MashSketcher),
 *                             N x _common_hashes (:337-341, :419-459), stable sort (:348),
 *                             first `top` rows (:391)
 *   skx_stream_table[_add] <- read / seed `sum_of_shared_hashes` (:326, :341)
 *   skx_common_hashes      <- Sketchy::_common_hashes (src/sketchy.rs:419-459) for sketch
 *                             collections (the `shared` all-pairs use at :251-261)
 *   skx_sketch_reads       <- finch SketchScheme::{process,to_vec} as called at :331-335
 *
 * Conventions: every function returns 0 (SKX_OK) or a negative SKX_ERR_* code and never
 * throws; skx_last_error() returns a thread-local message for the last failure.  Handles
 * are opaque and owned by the library.  Host buffers are caller-owned and only borrowed for
 * the duration of the call.  A stream handle is driven by one host thread at a time (the
 * reference loop is single-threaded); different handles may be used from different threads.
 * There is NO CPU fallback: with no usable gfx950 device every entry point that needs one
 * fails with SKX_ERR_NO_DEVICE.
 */
#ifndef SKETCHY_HIP_H
#define SKETCHY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKX_OK 0
#define SKX_ERR_INVALID (-1)      /* bad argument (NULL, k out of range, top > n_genomes ...) */
#define SKX_ERR_NO_DEVICE (-2)    /* no HIP device / device index out of range */
#define SKX_ERR_HIP (-3)          /* a HIP runtime call failed; see skx_last_error() */
#define SKX_ERR_UNSORTED (-4)     /* a reference column is not strictly ascending */
#define SKX_ERR_CAPACITY (-5)     /* batch exceeds the capacity given at skx_stream_create */
#define SKX_ERR_COMM (-6)         /* RCCL failure */
#define SKX_ERR_UNSUPPORTED (-7)  /* parameter combination outside what the kernels support */

#define SKX_MAX_K 32u             /* k-mer length 1..32 (sketchy default 16, src/cli.rs:39-41) */
#define SKX_MAX_TOP 64u           /* rows ranked per read (sketchy default 1, src/cli.rs:118-119) */
#define SKX_MAX_SPECIES 64u       /* reference collections resident together in one skx_ref */

typedef struct skx_ref skx_ref;
typedef struct skx_stream skx_stream;
typedef struct skx_comm skx_comm;

/* ---- library / device ------------------------------------------------------------- */
const char *skx_last_error(void);
const char *skx_version(void);
int skx_device_count(void);
/* name (<= name_cap bytes), number of CUs and bytes of device memory of `device` */
int skx_device_info(int device, char *name, size_t name_cap, int *compute_units, uint64_t *total_mem);
/* PCI address of `device` as "domain:bus:device.function" (>= 13 bytes): what a host that feeds the device from page-locked
 * memory looks up under /sys/bus/pci/devices/<id>/ (numa_node, local_cpulist) to keep its threads on the device's NUMA node */
int skx_device_pci_bus_id(int device, char *bus_id, size_t cap);

/*
 * Process-wide policies, consulted when a reference or a stream is CREATED (no effect on existing handles, never on
 * results -- only on memory and speed).  Names:
 *   "kmer_prefilter"  0 off (default) / 1 on / 2 on when its table fits 32 KB: for k = 16, skx_ref_create also builds a Bloom
 *                     table over the canonical 16-mers whose hash can meet the reference (every 4^16 / 2 of them is hashed
 *                     once, ~10 ms on an MI355X; 8 table bits per key); the production sketcher then hashes only the windows
 *                     of a read that pass it (reads with at most s windows -- longer ones can be truncated and take the plain
 *                     loop).  Rows, tables and debug outputs are identical either way.  One gather per window: it pays only
 *                     while the table sits in the L1 caches (a 16 KB table: +10 % reads/s; 128 KB: -10 %; 8 MB: -50 %), i.e.
 *                     for references with a few thousand distinct hashes -- not for a species-wide collection.
 *   "filter_bits_per_hash"  table bits of the membership filter per DISTINCT reference hash, 4..4096 (default 32: ~0.03 % of
 *                     the read hashes no genome holds get through and cost an all-zero row each; 32 MB at 40 000 S. aureus-like
 *                     genomes x 10 000 hashes, which share all but ~6e6 of their 4e8 hashes).
 *   "stream_query_rows"  rows of a pass's bit matrices = distinct query hashes one pass can hold; 0 (default) = 65 536, never
 *                     fewer than one read can need (s).  A batch (or pair of batches) with more distinct in-range hashes is
 *                     cut into several passes.  The bit matrices take rows x n_genomes / 8 bytes, three times.
 *   "stream_coalesce" 1 .. 8 (default 8): how many batches of skx_stream_enqueue_device / skx_stream_submit (there: at most 4)
 *                     may share one scoring pass (see skx_stream_enqueue_device).  n batches per pass divide the scans of the
 *                     reference per read by n (C2, 98 304-read batches: 77 / 97 / 111 / 116 M reads/s at 1 / 2 / 4 / 8) at
 *                     the cost of up to n - 1 more batches of latency and n + 1 copies of the per-batch sketch rows (25 MB
 *                     each at C2).  A group whose pairs or distinct hashes turn out too many for one pass is processed
 *                     batch by batch, and the stream forms smaller groups from then on.
 *   "rank_lanes"      1 .. 4 (default 2): the rankings of the batches that share a pass run as chains on this many HIP streams,
 *                     each with its own scratch (the table a batch starts from is all a chain needs of the one before it: it waits
 *                     for that chain's second kernel, not for its end).  C2, 20 batches from a fresh table: 131 M reads/s on one
 *                     lane, 134 M on two, 129-131 M on three / four; ~0.6 GB of scratch per lane at C2, 2.3 GB at C4.
 *   "rare_hash_genomes"  0 (no rare-hash index) .. 2^20 (default 1024): see skx_ref_rare_index.
 *   "reuse_membership"  0 (default) / 1: references with a static dense dictionary (skx_ref_static_dense) -- the bits the scan finds
 *                     for them depend on the reference alone.  0: every scoring pass streams the reference once (8 x stride x genomes
 *                     bytes: the pass SURVEY 8(d)'s roofline figure is defined on).  1: a stream scans the reference once per buffer set
 *                     and keeps the rows; later passes read the 52 MB (C2) of kept bits instead of the 3.2 GB of hashes.
 *   "comm_timeout_ms" 0 (default: no watchdog) .. 86 400 000: skx_comm_create (ncclCommInitRank) and skx_stream_allreduce
 *                     (ncclAllReduce + its stream synchronisation) block for ever when a peer never arrives.  With a
 *                     timeout set, a call that has not returned in time prints which rank was stuck in what to stderr and
 *                     ends the PROCESS with exit code SKX_COMM_TIMEOUT_EXIT -- a hung collective cannot be cancelled, and a
 *                     rank that exits lets the launcher tear the job down.  Consulted at every call (not at creation).
 * Unknown names fail with SKX_ERR_INVALID.
 */
#define SKX_COMM_TIMEOUT_EXIT 86
int skx_set_option(const char *name, uint64_t value);
int skx_get_option(const char *name, uint64_t *value);

/* ---- reference sketch collection, resident in HBM --------------------------------- */
/*
 * hashes: genome g's ascending distinct 64-bit hashes at [g*stride, g*stride + col_len[g]), col_len[g] <= stride.
 * k, seed, s are the sketch parameters reads are sketched with (src/sketchy.rs:331): the reference takes them from the
 * FIRST sketch of the collection -- s := number of hashes of sketch 0 (src/sketchy.rs:82, :520-527) -- while the other
 * sketches of a collection may be longer or shorter and are still intersected in full (:425-438).  So `s` (read sketch
 * size) and `stride` (capacity of a column = the longest sketch) are separate; a host mirroring the reference passes
 * s = col_len[0].  The library keeps its own device copy (a tiled, rank-major stride x N matrix); `hashes` may be
 * freed after the call.
 */
int skx_ref_create(skx_ref **out, int device, uint32_t k, uint64_t seed, uint32_t s, uint32_t stride, uint32_t n_genomes,
                   const uint64_t *hashes, const uint32_t *col_len);
/*
 * Several reference collections ("species": one sketch file each, src/sketchy.rs:81-82) resident together and scored
 * in ONE pass per batch.  The reference binary takes one sketch file per `predict` run, so a multi-species deployment
 * runs one predict per species over the same reads; here every read is sketched once and scanned against all of them,
 * and every species keeps its own running table and its own (sum desc, index asc) ranking.  All collections share
 * k, seed, s (the read sketch depends on them) and the column stride.  hashes[sp] / col_len[sp] are laid out as for
 * skx_ref_create.
 * Everywhere below, "n_genomes" of such a reference is the total over its species and genome-indexed arrays
 * (running table, per_read_shared, skx_common_hashes) hold the species one after the other; ranked rows come per
 * species, with genome indices local to the species.
 */
int skx_ref_create_multi(skx_ref **out, int device, uint32_t k, uint64_t seed, uint32_t s, uint32_t stride, uint32_t n_species,
                         const uint32_t *n_genomes, const uint64_t *const *hashes, const uint32_t *const *col_len);
int skx_ref_n_genomes(const skx_ref *ref, uint32_t *n_genomes);
int skx_ref_n_species(const skx_ref *ref, uint32_t *n_species);
int skx_ref_species_genomes(const skx_ref *ref, uint32_t species, uint32_t *n_genomes);
/* read sketch size and column stride the reference was created with */
int skx_ref_sketch_size(const skx_ref *ref, uint32_t *s, uint32_t *stride);
/* k-mer prefilter of the reference: keys it holds (0: none was built) and the bytes of its table */
int skx_ref_kmer_filter(const skx_ref *ref, uint64_t *n_keys, uint64_t *table_bytes);
/*
 * Rare-hash index of the reference (policy "rare_hash_genomes", default 1024, 0 = none): the distinct hashes of the collection
 * (n_keys), how many of them at most that many genomes hold (n_rare_keys), the entries of their genome lists (n_postings) and the
 * device memory of the whole index.  A pass looks its rare query hashes up there instead of asking the scan for them (the scan's
 * dictionary then holds the hashes many genomes share); results are the same with and without it.  All zero: none was built.
 */
int skx_ref_rare_index(const skx_ref *ref, uint64_t *n_keys, uint64_t *n_rare_keys, uint64_t *n_postings, uint64_t *bytes);
/*
 * ... and how its long genome lists (more than 8 genomes) are stored: n_long_lists of them have a bit row over the genomes;
 * n_pattern_lists of those are ALSO written as one of n_patterns shared patterns plus at most 14 exceptions (the lineage-level hashes
 * of a clonal collection: "the lineage's strains minus the odd one") -- a pass adds such rows up per pattern instead of per row; bytes =
 * the device memory of the records and the pattern matrix.  Exact either way; all zero: no bit rows / no patterns.
 */
int skx_ref_patterns(const skx_ref *ref, uint64_t *n_long_lists, uint64_t *n_patterns, uint64_t *n_pattern_lists, uint64_t *bytes);
/*
 * The static dense dictionary: with the rare-hash index the scan is only ever asked for hashes held by more genomes than the index
 * lists, and which hashes those are depends on the reference alone -- n_hashes of them (0 with *is_static set: every hash is rare,
 * nothing is ever scanned for).  Their rows of the bit matrix are fixed, a pass needs no dictionary sort or windows for them, and its
 * scan is queued with its first batch's sketch instead of behind its dictionary.  *is_static = 0: per-pass dictionaries (no index, too
 * many dense hashes for the scan kernel's slices, or long lists without bit rows).
 */
int skx_ref_static_dense(const skx_ref *ref, int *is_static, uint64_t *n_hashes);
/* bytes of reference hashes one scoring pass streams from HBM (8*stride*n_genomes) */
int skx_ref_pass_bytes(const skx_ref *ref, uint64_t *bytes);
void skx_ref_destroy(skx_ref *ref);

/* ---- streaming predictor ----------------------------------------------------------- */
/*
 * top_k: rows ranked after every read (per species), 0..min(genomes of the smallest species, SKX_MAX_TOP)
 * (0 = no per-read ranking, table only).  max_batch_reads / max_batch_bases bound one skx_stream_push call.
 */
int skx_stream_create(skx_stream **out, const skx_ref *ref, uint32_t top_k, uint32_t max_batch_reads,
                      uint64_t max_batch_bases);
/*
 * Consume n_reads reads: read r is the raw (un-normalised) ASCII bytes bases[offsets[r] .. offsets[r+1]).
 * Outputs are optional (NULL to skip), caller-allocated host arrays:
 *   topk_idx [n_reads][n_species][top_k]  genome indices (within the species) after that read, order =
 *                                         (cumulative sum desc, index asc); n_species = 1 for skx_ref_create
 *   topk_sum [n_reads][n_species][top_k]  the cumulative sums of those genomes after that read
 *   per_read_shared [n_reads][n_genomes]  this read's shared-hash count per genome (parity/debug)
 *   sketches [n_reads][s], sketch_len [n_reads]  the read's bottom-s sketch, ascending (parity/debug)
 * Synchronous: on return the running table includes these reads.
 */
int skx_stream_push(skx_stream *st, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                    uint32_t *topk_idx, uint64_t *topk_sum, uint32_t *per_read_shared, uint64_t *sketches,
                    uint32_t *sketch_len);
/*
 * Same, with every pointer a DEVICE pointer on the stream's device (bench path: inputs already
 * resident in HBM; only the optional topk outputs are produced).  Asynchronous on the stream's
 * HIP stream except for the library's own internal synchronisation points; call
 * skx_stream_sync() before reading the outputs.
 */
int skx_stream_push_device(skx_stream *st, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads,
                           uint64_t n_bases, uint32_t *d_topk_idx, uint64_t *d_topk_sum);
/*
 * skx_stream_push_device in two halves, so that consecutive batches overlap on the device without the host in between:
 * the call queues the sketch of THIS batch first, then waits for the published summary of the batch(es) enqueued before
 * and queues their scan / ranking passes.  (skx_stream_push_device does both halves of one batch in one call: the sketch
 * stream then idles while the host queues the passes -- ~0.25 ms per 98,304-read batch, measured.)
 * With the option "stream_coalesce" = n (default 8) up to n batches enqueued back to back SHARE one pass -- one dictionary
 * of their distinct hashes, one scan of the reference, one transpose, then one ranking each, in order -- when their pairs
 * and distinct hashes together fit a pass; otherwise each takes its own.  The back half of a group is queued by the call
 * that opens the next group (behind that batch's sketch), or by the flush.
 * Consequences: rows of batch i are written by work queued during one of the calls i + 1 .. i + n; an error found in batch i (offsets
 * not monotonic, a read outside n_bases) is returned by one of those calls or by the flush -- batches enqueued BEFORE the
 * faulty one are processed, the faulty one and those enqueued after it (up to the call that reports the error) are
 * dropped.  d_bases / d_offsets of a batch must stay untouched until skx_stream_sync() (or any other flushing entry
 * point followed by a synchronisation).  skx_stream_flush() queues the outstanding half; every other entry point of the
 * stream (sync, table, rank, push, reset, stats ...) flushes first.  Rows and table are those of skx_stream_push_device
 * on the same batches in the same order (src/sketchy.rs:328-354 is sequential over reads).
 */
int skx_stream_enqueue_device(skx_stream *st, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads,
                              uint64_t n_bases, uint32_t *d_topk_idx, uint64_t *d_topk_sum);
int skx_stream_flush(skx_stream *st);
/*
 * 4-bit packed input (optional; the reference reads ASCII FASTX: src/sketchy.rs:328-333 -- this is a wire format for
 * hosts that feed the device over PCIe, where a 98 304-read batch of 1.5 kb reads is 147 MB of ASCII and the copy,
 * not the kernels, bounds a host-fed stream).  After skx_stream_set_packed_input(st, 1) every entry point of the stream
 * takes `bases` as two bases per byte, low nibble first, codes 0..3 = A C G T(U), any other value = a retained
 * non-ACGT byte (N, IUPAC, '-': breaks k-mers exactly as in the ASCII path); whitespace does not exist in this format.
 * `offsets` (and n_bases / max_batch_bases) count BASES, i.e. nibbles of the stream; a read may start on an odd
 * nibble.  skx_pack_bases() is the matching host-side packer: it appends the normalised bases of `ascii[0..n)`
 * (whitespace dropped) at nibble position `nibble_pos` of `packed` and returns the new position.  Rows, tables and
 * debug outputs are those of the same reads given as ASCII.
 */
int skx_stream_set_packed_input(skx_stream *st, int on);
uint64_t skx_pack_bases(const uint8_t *ascii, uint64_t n, uint8_t *packed, uint64_t nibble_pos);
/* the same up to the first line feed of ascii[0..n): packs what lies in front of it, *consumed = bytes taken, the line feed
 * included (n when there is none) -- a FASTX parser finds the end of a sequence line and packs it in one pass over its bytes */
uint64_t skx_pack_line(const uint8_t *ascii, uint64_t n, uint8_t *packed, uint64_t nibble_pos, uint64_t *consumed);
int skx_stream_sync(skx_stream *st);
/*
 * Host-fed pipeline.  skx_stream_submit() queues a batch from PAGE-LOCKED host buffers (skx_host_alloc; bases, offsets
 * and the optional row arrays) and returns without waiting for it: the host-to-device copy of batch i+1 runs on its
 * own stream into another staging slot (three; up to nine when batches share passes, "stream_coalesce" >= 2: each holds a batch's
 * bases, offsets and rows on the device) while batch i is sketched, scanned and ranked.  Processing lags one call
 * behind submission (submit(i) starts the copy of batch i, then runs batch i-1 through the kernels), so a single host
 * thread keeps the copy engine and the kernels busy at the same time -- what needletail's reader plus the loop of
 * src/sketchy.rs:328-354 would look like with the device in between.  Rows of batch `ticket` ([n_reads][n_species]
 * [top_k], as skx_stream_push) are in its arrays once skx_stream_wait(st, ticket) or skx_stream_drain(st) has returned;
 * the input buffers may be reused after the NEXT submit has returned (or after wait / drain).  Results are those of
 * the same batches pushed with skx_stream_push in the same order.  Do not interleave with skx_stream_push[_device]
 * without a drain in between.
 * Errors: a batch that turns out to be faulty (offsets not monotonic ...) makes the submit that hands it to the kernels -- or a
 * later wait / drain -- fail; batches submitted BEFORE it are processed and their rows delivered, the faulty one and those
 * submitted after it (up to the failing call, the failing call's own batch included) are dropped, and skx_stream_wait on a
 * dropped ticket fails instead of returning rows that were never written.
 * skx_stream_wait(ticket) on a batch that still waits for the partners of its group closes the group early (a pass for fewer
 * batches): a host that wants full groups waits for a ticket only once nine younger ones have been submitted -- by then the
 * batch is through and the call returns at once (sketchy_amd/host/sketchy_host.cpp does exactly that).
 */
int skx_stream_submit(skx_stream *st, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                      uint32_t *topk_idx, uint64_t *topk_sum, uint64_t *ticket);
int skx_stream_wait(skx_stream *st, uint64_t ticket);
int skx_stream_drain(skx_stream *st);
/* running sum-of-shared-hashes table, u64[n_genomes] (host) */
int skx_stream_table(skx_stream *st, uint64_t *cum);
/* cum[g] += add[g]: resume from a checkpoint, or offset a shard by the totals of earlier shards */
int skx_stream_table_add(skx_stream *st, const uint64_t *add);
/* zero the table and the read counter */
int skx_stream_reset(skx_stream *st);
/* reads consumed so far (the reference's `read` counter minus one, src/sketchy.rs:327/:350) */
int skx_stream_reads(const skx_stream *st, uint64_t *n_reads);
/*
 * Counters for tuning / reporting (no effect on results): out[0] (read, hash) pairs of the last push, [1] scoring
 * passes it took, [2] distinct query hashes of the most recent dictionary, [3] reads sketched by the block sketcher so
 * far, [4] passes so far, [5] of those with the lean scan kernel, [6] pair capacity of a pass, [7] rank groups (512 genomes)
 * that received any bit in the most recent pass, [8] long reads (more than 8192 bases) whose sketch was split over several
 * wavefronts so far, [9] the segments they were cut into, [10] batches that were sketched a second time because their rows
 * did not fit the stream's row pool (it grows to fit), [11] passes that served SEVERAL enqueued batches (option
 * "stream_coalesce"), [12] groups of enqueued batches that turned out too large for one pass together (or held an error) and were
 * processed one by one -- the stream forms smaller groups after that, [13] rows (distinct query hashes) the bit matrices of a
 * pass hold right now, [14] how often they grew because a batch held more distinct hashes than that (only without a
 * "stream_query_rows" policy; at most a quarter of the free device memory is taken), [15] of the most recent dictionary ([2]) the
 * hashes the scan looked for (all of them without a rare-hash index, skx_ref_rare_index; else those many genomes hold), [16] batches
 * whose per-read ranking ran on their CANDIDATES only (the genomes whose value at the end of the batch reaches the top-th best
 * value at its start: at most 1024 per species), [17] batches ranked on every genome.  Waits for the stream's queued work.
 */
#define SKX_N_STATS 18
int skx_stream_stats(skx_stream *st, uint64_t *out, uint32_t n_out);
/* rank the CURRENT table: first top_k of (sum desc, index asc) per species; idx/sum are host arrays [n_species][top_k] */
int skx_stream_rank(skx_stream *st, uint32_t top_k, uint32_t *idx, uint64_t *sum);
void skx_stream_destroy(skx_stream *st);

/*
 * Per-stage device time, measured with HIP events on the stream's own HIP stream when enabled.
 * Stages: 0 sketch (+ membership filter), 1 dictionary (distinct query hashes, windows), 2 reference scan (the roofline kernel),
 * 3 bit-matrix transpose, 4 rank (segment sums, prefix, per-read top-k, merge).
 */
#define SKX_N_STAGES 5
/* enabled: 0 off, 1 every stage, 2 only stage 2 (two event records per pass instead of twelve: ~2 % of a step) */
int skx_stream_set_profiling(skx_stream *st, int enabled);
/* ms[SKX_N_STAGES] accumulated milliseconds, launches[SKX_N_STAGES] timed intervals; resets the counters */
int skx_stream_profile(skx_stream *st, double *ms, uint64_t *launches);
/*
 * The reference scan ALONE on the device: `reps` launches back to back behind a synchronisation, their average in milliseconds.
 * (References with a static dense dictionary -- the hashes the scan can be asked for are a property of the reference; see
 * skx_ref_static_dense -- have their scans queued beside the sketches of the batches they serve, so no other entry point times one
 * by itself.)  SKX_ERR_INVALID for a stream that builds its scan's dictionary per pass.
 */
int skx_stream_scan_alone(skx_stream *st, uint32_t reps, double *ms_avg);

/* ---- stand-alone operators (parity / `shared` subcommand) --------------------------- */
/* sketch n_reads reads with (k, seed, s); outputs as in skx_stream_push (host arrays) */
int skx_sketch_reads(int device, uint32_t k, uint64_t seed, uint32_t s, const uint8_t *bases,
                     const uint64_t *offsets, uint32_t n_reads, uint64_t *sketches, uint32_t *sketch_len);
/*
 * common[q][g] = |query sketch q  intersect  reference genome g|  (src/sketchy.rs:419-438) for
 * n_query ascending sketches laid out like skx_ref_create's input (row stride q_stride).
 */
int skx_common_hashes(const skx_ref *ref, const uint64_t *query, const uint32_t *query_len, uint32_t n_query,
                      uint32_t q_stride, uint32_t *common);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI --------------------------------- */
#define SKX_COMM_ID_BYTES 128
int skx_comm_unique_id(uint8_t id[SKX_COMM_ID_BYTES]);                 /* rank 0; broadcast it out of band */
int skx_comm_create(skx_comm **out, int device, int rank, int n_ranks, const uint8_t id[SKX_COMM_ID_BYTES]);
/* ranks of the communicator as RCCL reports them (ncclCommCount) */
int skx_comm_n_ranks(const skx_comm *comm, int *n_ranks);
/* in-place sum of the running tables of all ranks' streams (u64, exact) */
int skx_stream_allreduce(skx_stream *st, skx_comm *comm);
void skx_comm_destroy(skx_comm *comm);

/* ---- raw device buffers (so a host without a HIP binding can stage inputs in HBM) ---- */
int skx_dev_malloc(int device, void **d_ptr, size_t bytes);
int skx_dev_free(int device, void *d_ptr);
int skx_dev_upload(int device, void *d_dst, const void *h_src, size_t bytes);
int skx_dev_download(int device, void *h_dst, const void *d_src, size_t bytes);
int skx_dev_synchronize(int device);
/* free / total bytes of device memory (hipMemGetInfo): what a host sizing several streams per GPU looks at */
int skx_dev_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes);
/* page-locked host memory for batch buffers handed to skx_stream_push / skx_sketch_reads: the H2D copy then runs at
 * full PCIe rate and asynchronously (what a double-buffered FASTX front-end wants; needletail's reader in
 * src/sketchy.rs:89-92 has no counterpart, the reference never leaves the host) */
int skx_host_alloc(int device, void **h_ptr, size_t bytes);
int skx_host_free(int device, void *h_ptr);

#ifdef __cplusplus
}
#endif
#endif /* SKETCHY_HIP_H */
