// skx_capi.hip -- implementation of the C ABI declared in include/sketchy_hip.h.
//
// Host-side orchestration of one skx_stream_push (the body of the reference's hot loop,
// src/sketchy.rs:328-354, for a whole batch of reads):
//   sketch every read (keeping only hashes some genome holds) -> normally ONE pass per push, else cut the batch
//   into passes that fit the pass workspace -> per pass: dictionary of the distinct query hashes, windows,
//   reference scan, bit transpose, running table + per-read top-k (pruned).
// No CPU fallback exists: without a HIP device the calls fail with SKX_ERR_NO_DEVICE.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "sketchy_hip.h"
#include "skx_common.hpp"
#include "skx_kernels.hpp"

using skx::u32;
using skx::u64;

#define SKX_API extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(SKX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define SKXCHK(expr)             \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != SKX_OK) return rc_; \
    } while (0)

SKX_API const char* skx_last_error(void) { return g_err.c_str(); }
#ifdef SKX_EXPERIMENTS
SKX_API const char* skx_version(void) { return "sketchy-hip 0.5.0 (gfx950, experiments build: reads SKX_* environment knobs)"; }
#else
SKX_API const char* skx_version(void) { return "sketchy-hip 0.5.0 (gfx950)"; }
#endif

SKX_API int skx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
static int use_device(int device) {
    int n = skx_device_count();
    if (n <= 0) return fail(SKX_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(SKX_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    return SKX_OK;
}
SKX_API int skx_device_info(int device, char* name, size_t name_cap, int* compute_units, uint64_t* total_mem) {
    SKXCHK(use_device(device));
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    if (name && name_cap) snprintf(name, name_cap, "%s (%s)", p.name, p.gcnArchName);
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (total_mem) *total_mem = p.totalGlobalMem;
    return SKX_OK;
}

SKX_API int skx_device_pci_bus_id(int device, char* bus_id, size_t cap) {
    if (!bus_id || cap < 13) return fail(SKX_ERR_INVALID, "bus_id needs room for 13 bytes");
    SKXCHK(use_device(device));
    HIPCHK(hipDeviceGetPCIBusId(bus_id, (int)cap, device));
    return SKX_OK;
}

// ------------------------------------------------------------------ raw device buffers
SKX_API int skx_dev_malloc(int device, void** d_ptr, size_t bytes) {
    if (!d_ptr) return fail(SKX_ERR_INVALID, "d_ptr is NULL");
    SKXCHK(use_device(device));
    HIPCHK(hipMalloc(d_ptr, bytes ? bytes : 1));
    return SKX_OK;
}
SKX_API int skx_dev_free(int device, void* d_ptr) {
    SKXCHK(use_device(device));
    HIPCHK(hipFree(d_ptr));
    return SKX_OK;
}
SKX_API int skx_dev_upload(int device, void* d_dst, const void* h_src, size_t bytes) {
    SKXCHK(use_device(device));
    HIPCHK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return SKX_OK;
}
SKX_API int skx_dev_download(int device, void* h_dst, const void* d_src, size_t bytes) {
    SKXCHK(use_device(device));
    HIPCHK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return SKX_OK;
}
SKX_API int skx_host_alloc(int device, void** h_ptr, size_t bytes) {
    if (!h_ptr) return fail(SKX_ERR_INVALID, "h_ptr is NULL");
    SKXCHK(use_device(device));
    HIPCHK(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return SKX_OK;
}
SKX_API int skx_host_free(int device, void* h_ptr) {
    SKXCHK(use_device(device));
    HIPCHK(hipHostFree(h_ptr));
    return SKX_OK;
}
SKX_API int skx_dev_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
    SKXCHK(use_device(device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return SKX_OK;
}
SKX_API int skx_dev_synchronize(int device) {
    SKXCHK(use_device(device));
    HIPCHK(hipDeviceSynchronize());
    return SKX_OK;
}

// ------------------------------------------------------------------ host-time accounting (experiments build: SKX_HOST_TIMES=1)
#ifdef SKX_EXPERIMENTS
#include <chrono>
struct HostTimes { double front = 0, wait = 0, back = 0; u64 n = 0; };
static HostTimes g_ht;
static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define SKX_T0() const double t0_ = now_us()
#define SKX_ACC(field) g_ht.field += now_us() - t0_
// SKX_HOST_TRACE=1: every marked point of the queueing thread with its time, printed when the stream is synchronised (which
// call of the host blocks, and for how long: the device timeline only shows that something was queued late)
struct HostMark { const char* what; int i; double t; };
static std::vector<HostMark> g_marks;
static const bool g_trace_on = skx::knob("SKX_HOST_TRACE") != nullptr;
#define SKX_MARK(what, i) do { if (g_trace_on) g_marks.push_back(HostMark{what, (int)(i), now_us()}); } while (0)
static void dump_marks() {
    if (!g_trace_on || g_marks.empty()) return;
    const double t0 = g_marks.front().t;
    double prev = t0;
    for (const HostMark& m : g_marks) { fprintf(stderr, "[skx host] %9.1f (+%7.1f) %s %d\n", m.t - t0, m.t - prev, m.what, m.i); prev = m.t; }
    g_marks.clear();
}
#else
#define SKX_MARK(what, i) do {} while (0)
static void dump_marks() {}
#define SKX_T0() do {} while (0)
#define SKX_ACC(field) do {} while (0)
#endif

// ------------------------------------------------------------------ policies (skx_set_option)
// (atomics: a host may create references / streams on several threads while another one sets a policy; every reader takes one
// consistent value at creation)
static std::atomic<u32> g_kmer_prefilter{0};  // k-mer prefilter of k = 16 references: 0 off, 1 on, 2 on when its table is small enough to pay
static std::atomic<u32> g_filter_bits_per_hash{32};  // membership filter: table bits per DISTINCT reference hash
static std::atomic<u32> g_stream_query_rows{0};       // rows of a pass's bit matrices (distinct query hashes per pass); 0 = default (65 536)
static std::atomic<u32> g_stream_coalesce{8};         // batches of skx_stream_enqueue_device / skx_stream_submit that may share one pass (1 .. 8)
static const int kRankLanesMax = 4;
static std::atomic<u32> g_rank_lanes{2};   // ranking lanes of a stream that enqueues (1 .. 4): chains of consecutive batches that run side by side
static std::atomic<u32> g_rare_hash_genomes{1024};   // rare-hash index of a reference: hashes held by at most this many genomes get genome lists (0 = no index)
static std::atomic<u32> g_reuse_membership{0};   // references with a static dense dictionary: 1 = a stream scans the reference ONCE per buffer set and keeps the rows
static std::atomic<u64> g_comm_timeout_ms{0};  // watchdog of skx_comm_create / skx_stream_allreduce: 0 = none (block for ever, as RCCL does)

SKX_API int skx_set_option(const char* name, uint64_t value) {
    if (!name) return fail(SKX_ERR_INVALID, "NULL option name");
    if (!strcmp(name, "kmer_prefilter")) {
        if (value > 2) return fail(SKX_ERR_INVALID, "kmer_prefilter must be 0 (off), 1 (on) or 2 (on when its table is small)");
        g_kmer_prefilter = (u32)value;
        return SKX_OK;
    }
    if (!strcmp(name, "stream_coalesce")) {
        if (value < 1 || value > (uint64_t)skx::kPairBaseMax + 1) return fail(SKX_ERR_INVALID, "stream_coalesce must be 1 .. 8");
        g_stream_coalesce = (u32)value;
        return SKX_OK;
    }
    if (!strcmp(name, "stream_query_rows")) {
        if (value > (1u << 22)) return fail(SKX_ERR_INVALID, "stream_query_rows must be 0 (default) .. 2^22");
        g_stream_query_rows = (u32)value;
        return SKX_OK;
    }
    if (!strcmp(name, "filter_bits_per_hash")) {
        if (value < 4 || value > 4096) return fail(SKX_ERR_INVALID, "filter_bits_per_hash must be 4..4096");
        g_filter_bits_per_hash = (u32)value;
        return SKX_OK;
    }
    if (!strcmp(name, "rank_lanes")) {
        if (value < 1 || value > (uint64_t)kRankLanesMax) return fail(SKX_ERR_INVALID, "rank_lanes must be 1 .. %d", kRankLanesMax);
        g_rank_lanes = (u32)value;
        return SKX_OK;
    }
    if (!strcmp(name, "rare_hash_genomes")) {
        if (value > (1u << 20)) return fail(SKX_ERR_INVALID, "rare_hash_genomes must be 0 (no rare-hash index) .. 2^20");
        g_rare_hash_genomes = (u32)value;
        return SKX_OK;
    }
    if (!strcmp(name, "reuse_membership")) {
        if (value > 1) return fail(SKX_ERR_INVALID, "reuse_membership must be 0 (every pass scans the reference) or 1 (static dense rows are kept)");
        g_reuse_membership = (u32)value;
        return SKX_OK;
    }
    if (!strcmp(name, "comm_timeout_ms")) {
        if (value > 86400000ull) return fail(SKX_ERR_INVALID, "comm_timeout_ms must be 0 (no watchdog) .. 86400000");
        g_comm_timeout_ms = value;
        return SKX_OK;
    }
    return fail(SKX_ERR_INVALID, "unknown option '%s'", name);
}
SKX_API int skx_get_option(const char* name, uint64_t* value) {
    if (!name || !value) return fail(SKX_ERR_INVALID, "NULL argument");
    if (!strcmp(name, "kmer_prefilter")) { *value = g_kmer_prefilter; return SKX_OK; }
    if (!strcmp(name, "filter_bits_per_hash")) { *value = g_filter_bits_per_hash; return SKX_OK; }
    if (!strcmp(name, "stream_query_rows")) { *value = g_stream_query_rows; return SKX_OK; }
    if (!strcmp(name, "stream_coalesce")) { *value = g_stream_coalesce; return SKX_OK; }
    if (!strcmp(name, "comm_timeout_ms")) { *value = g_comm_timeout_ms; return SKX_OK; }
    if (!strcmp(name, "rank_lanes")) { *value = g_rank_lanes; return SKX_OK; }
    if (!strcmp(name, "rare_hash_genomes")) { *value = g_rare_hash_genomes; return SKX_OK; }
    if (!strcmp(name, "reuse_membership")) { *value = g_reuse_membership; return SKX_OK; }
    return fail(SKX_ERR_INVALID, "unknown option '%s'", name);
}

// ------------------------------------------------------------------ reference
struct skx_ref {
    int device = 0;
    u32 k = 0, s = 0 /* rows of the matrix = column stride */, s_read = 0 /* sketch size of the reads (|sketch 0| in the reference) */;
    u32 n_genomes = 0 /* real genomes, all species */, n_tiles = 0, n_pad = 0, rb = 0, n_bands = 0;
    u64 seed = 0;
    // species (reference collections scanned together); each padded to whole rank groups of 512 genomes
    u32 n_species = 0;
    std::vector<u32> sp_n, sp_g0, sp_real0;  // real genomes / first padded index / first index in the caller's order
    u32 min_species = 0;                     // smallest species (bounds top_k)
    u32 max_species = 0;                     // largest species (more genomes than a compact ranking takes: a fresh table cannot rank compactly)
    u32 *d_sp_g0 = nullptr, *d_sp_n = nullptr, *d_grp_sp = nullptr;
    u32* d_real2pad = nullptr;               // [n_genomes] caller's genome index -> padded index
    u64* d_mat = nullptr;  // [n_tiles][s][256]
    u64 *d_lo = nullptr, *d_hi = nullptr;  // [n_bands * n_tiles]
    u64 max_ref = 0;       // largest hash in the matrix (queries above it cannot match)
    bool any = false;      // at least one real hash
    u32 n_exc = 0;         // hashes >= kEmpty lifted out of the matrix
    u32* d_exc_g = nullptr;
    u64* d_exc_h = nullptr;
    // membership filter over the union of all reference hashes: bit (h >> filt_shift) of d_filt
    u64* d_filt = nullptr;  // 64-bit words; word h >> filt_shift
    u32 filt_shift = 0;
    u64 filt_bits = 0, n_distinct = 0;  // table bits; distinct reference hashes (linear-counting estimate)
    // k-mer prefilter (k = 16): Bloom table over the canonical 16-mers whose hash passes the membership filter
    u32* d_kf = nullptr;
    u32 kf_shift = 0, kf_keys = 0;
    // rare-hash index (skx_kernels.hip): table over the distinct reference hashes + genome lists of those few genomes hold
    u64* d_kt_key = nullptr;
    u32 *d_kt_cnt = nullptr, *d_kt_off = nullptr, *d_post = nullptr;
    u32 kt_mask = 0, rare_max = 0;
    u64 n_keys = 0, n_rare_keys = 0, n_postings = 0;  // distinct hashes (exact), of those rare, entries of their lists
    // long lists (more than 8 genomes) also as bit rows over the genomes (skx_kernels.hip, "long lists as bit rows")
    u64* d_mlong = nullptr;
    u64* d_mlongT = nullptr;  // the bit rows transposed: [n_pad][n_lw] (or NULL: no room -- the candidates' rows are then found row by row)
    u32 *d_lid = nullptr, *d_lslot = nullptr;
    u64 n_long = 0;
    u32 n_lw = 0;
    // most long lists as (pattern, exceptions) (skx_kernels.hip, "long lists as (pattern, exceptions)"; build_patterns below)
    u32 *d_prec = nullptr, *d_pat_rep = nullptr, *d_npat = nullptr;
    u64* d_pm = nullptr;
    u32 n_pat = 0;
    u64 n_pat_lists = 0;  // long lists stored as a pattern + at most 14 exceptions
    // every rare hash's bits come from its list or bit row WITHOUT M (rare_to_mq): all long lists have a bit row, or there is no long list
    bool rare_direct = false;
    // the static dense dictionary (build_static_dense): the hashes the scan can be asked for are a property of the REFERENCE
    u64* d_qs = nullptr;      // [n_sd] ascending
    u32* d_nsd = nullptr;     // {n_sd, 0, n_sd rounded up to 64, n_sd}: what the kernels take as n_q / n_d of the static rows
    u32* d_win_s = nullptr;   // [n_bands * n_tiles][2] its slice per (band, tile)
    u32 n_sd = 0;
    bool static_dense = false;
    // (several species: the dictionary is the species' sorted segments one after the other -- a tile of species A only needs A's hashes:
    // slices of ~200 entries again instead of ~1 000, the lean scan kernel instead of the dense-dictionary one)
    u32* d_srow = nullptr;    // [key-table slots] row of a dense hash
    u32* d_seg = nullptr;     // [2 n_species] first row / hashes of every species' segment
    u32* d_segw = nullptr;    // [2 n_species] the words of M that hold them
    u32 sd_tail0 = 0;         // first row of the lifted hashes (sorted tail)
    u64 n_forced_rare = 0;    // hashes held by more genomes than rare_hash_genomes that got a list anyway: they occur in several species
    skx::RareIndex rare_index() const {
        skx::RareIndex ri{d_kt_key, d_kt_off, d_kt_cnt, d_post, kt_mask, d_mlong, d_lid, d_lslot, n_pad / 64, d_mlongT, n_lw};
        ri.prec = d_prec; ri.pat_rep = d_pat_rep; ri.pm = d_pm; ri.d_npat = d_npat; ri.n_pat = n_pat;
        if (static_dense) { ri.qs = d_qs; ri.n_sd = n_sd; ri.srow = d_srow; ri.tail0 = sd_tail0; }
        return ri;
    }
    skx::KmerFilter kmer_filter() const { return skx::KmerFilter{d_kf, kf_shift}; }
    skx::Species species() const { return skx::Species{d_sp_g0, d_sp_n, d_grp_sp, n_species}; }
};

static void ref_free(skx_ref* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    (void)hipFree(r->d_mat); (void)hipFree(r->d_lo); (void)hipFree(r->d_hi);
    (void)hipFree(r->d_exc_g); (void)hipFree(r->d_exc_h); (void)hipFree(r->d_filt); (void)hipFree(r->d_kf);
    (void)hipFree(r->d_sp_g0); (void)hipFree(r->d_sp_n); (void)hipFree(r->d_grp_sp); (void)hipFree(r->d_real2pad);
    (void)hipFree(r->d_kt_key); (void)hipFree(r->d_kt_cnt); (void)hipFree(r->d_kt_off); (void)hipFree(r->d_post);
    (void)hipFree(r->d_mlong); (void)hipFree(r->d_mlongT); (void)hipFree(r->d_lid); (void)hipFree(r->d_lslot);
    (void)hipFree(r->d_prec); (void)hipFree(r->d_pat_rep); (void)hipFree(r->d_npat); (void)hipFree(r->d_pm);
    (void)hipFree(r->d_qs); (void)hipFree(r->d_nsd); (void)hipFree(r->d_win_s);
    (void)hipFree(r->d_srow); (void)hipFree(r->d_seg); (void)hipFree(r->d_segw);
    delete r;
}

static const u32 kGroupGenomes = skx::kRankWords * 64u;  // 512: a species starts on a rank-group boundary

// Most long lists of a clonal collection as (pattern, exceptions) -- skx_kernels.hip, "long lists as (pattern, exceptions)".  Needs the
// bit rows (the exceptions are the XOR of two of them); anything that goes wrong leaves the reference without patterns: every long list
// keeps its bit row, as in round 5.  Host work: one 12-byte record per long list (C2's SNP tree: 263 k).
static const u32 kPatMax = 16384;        // patterns of a reference (rows of the compact matrices, words of the pattern matrix)
static const u32 kPatGroupMin = 4;       // lists a signature group must hold to get a pattern
static void build_patterns(skx_ref* r) {
    static const int pat_env = skx::knob("SKX_PATTERNS") ? atoi(skx::knob("SKX_PATTERNS")) : 1;  // experiment knob: 0 = bit rows only
    if (!pat_env || !r->d_mlong || r->n_long < 64 || r->n_long >= 0x7FFFFFFFull) return;
    const u32 n_long = (u32)r->n_long, n_gw = r->n_pad / 64;
    u32 *d_sig = nullptr, *d_pat_of = nullptr, *d_done = nullptr;
    u64* d_content = nullptr;
    auto drop = [&]() {
        (void)hipFree(r->d_prec); (void)hipFree(r->d_pat_rep); (void)hipFree(r->d_npat); (void)hipFree(r->d_pm);
        r->d_prec = r->d_pat_rep = r->d_npat = nullptr; r->d_pm = nullptr; r->n_pat = 0; r->n_pat_lists = 0;
        (void)hipGetLastError();
    };
    bool ok = hipMalloc(&d_sig, (size_t)n_long * 4) == hipSuccess && hipMalloc(&d_content, (size_t)n_long * 8) == hipSuccess;
    try {
        std::vector<u32> sig(n_long), pat_of(n_long, 0xFFFFFFFFu), rep;
        std::vector<u64> content(n_long);
        if (ok) {
            skx::launch_list_sig(nullptr, r->d_lslot, n_long, r->d_kt_off, r->d_kt_cnt, r->d_post, d_sig, d_content);
            ok = hipGetLastError() == hipSuccess && hipMemcpy(sig.data(), d_sig, (size_t)n_long * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(content.data(), d_content, (size_t)n_long * 8, hipMemcpyDeviceToHost) == hipSuccess;
        }
        if (ok) {
            // the most frequent exact list of every signature group (ties: the lower bit row), groups by size
            struct Exact { u32 count = 0, first = 0, sig = 0; };
            struct Group { u32 size = 0, best_count = 0, best_first = 0; };
            std::unordered_map<u64, Exact> exact;
            std::unordered_map<u32, Group> groups;
            exact.reserve(n_long);
            for (u32 i = 0; i < n_long; ++i) {
                Exact& e = exact[content[i] ^ ((u64)sig[i] * 0xD6E8FEB86659FD93ull)];
                if (!e.count) { e.first = i; e.sig = sig[i]; }
                e.count += 1;
                groups[sig[i]].size += 1;
            }
            for (const auto& kv : exact) {
                const Exact& e = kv.second;
                Group& g = groups[e.sig];
                if (e.count > g.best_count || (e.count == g.best_count && e.first < g.best_first)) { g.best_count = e.count; g.best_first = e.first; }
            }
            std::vector<std::pair<u32, u32>> order;  // (size, signature)
            for (const auto& kv : groups) if (kv.second.size >= kPatGroupMin) order.push_back({kv.second.size, kv.first});
            std::sort(order.begin(), order.end(), [](const std::pair<u32, u32>& a, const std::pair<u32, u32>& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
            if (order.size() > kPatMax) order.resize(kPatMax);
            std::unordered_map<u32, u32> pat_of_sig;
            for (const auto& o : order) { pat_of_sig[o.second] = (u32)rep.size(); rep.push_back(groups[o.second].best_first); }
            for (u32 i = 0; i < n_long; ++i) { const auto it = pat_of_sig.find(sig[i]); if (it != pat_of_sig.end()) pat_of[i] = it->second; }
            ok = !rep.empty();
        }
        const u32 n_pat = (u32)rep.size(), n_pw = (n_pat + 63) / 64;
        size_t mem_free = 0, mem_total = 0;
        (void)hipMemGetInfo(&mem_free, &mem_total);
        ok = ok && ((size_t)n_long * skx::pat_record_words() * 4 + (size_t)n_pw * r->n_pad * 8) <= mem_free / 8;
        ok = ok && hipMalloc(&d_pat_of, (size_t)n_long * 4) == hipSuccess && hipMalloc(&d_done, 4) == hipSuccess &&
             hipMalloc(&r->d_pat_rep, (size_t)n_pat * 4) == hipSuccess && hipMalloc(&r->d_npat, 16) == hipSuccess &&
             hipMalloc(&r->d_prec, (size_t)n_long * skx::pat_record_words() * 4) == hipSuccess &&
             hipMalloc(&r->d_pm, (size_t)n_pw * r->n_pad * 8) == hipSuccess &&
             hipMemcpy(d_pat_of, pat_of.data(), (size_t)n_long * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(r->d_pat_rep, rep.data(), (size_t)n_pat * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemset(d_done, 0, 4) == hipSuccess;
        if (ok) {
            const u32 words[4] = {n_pat, 0u, (n_pat + 63u) & ~63u, n_pat};
            ok = hipMemcpy(r->d_npat, words, 16, hipMemcpyHostToDevice) == hipSuccess;
        }
        if (ok) {
            skx::launch_pat_exceptions(nullptr, r->d_mlong, n_gw, n_long, d_pat_of, r->d_pat_rep, r->d_prec, d_done);
            skx::launch_pat_matrix(nullptr, r->d_mlong, r->d_pat_rep, n_pat, n_gw, r->d_pm, r->n_pad);
            u32 done = 0;
            ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipMemcpy(&done, d_done, 4, hipMemcpyDeviceToHost) == hipSuccess;
            // (a collection whose long lists do not cluster: the pattern machinery would only add launches)
            ok = ok && (u64)done * 8 >= n_long;
            if (ok) { r->n_pat = n_pat; r->n_pat_lists = done; }
        }
    } catch (const std::bad_alloc&) { ok = false; }
    (void)hipFree(d_sig); (void)hipFree(d_content); (void)hipFree(d_pat_of); (void)hipFree(d_done);
    if (!ok) drop();
}

// The static dense dictionary.  With the rare-hash index a pass asks the scan only for hashes held by more genomes than the index
// lists -- and WHICH hashes those are is a property of the reference (C2: 10 140 of 893 260 keys).  So the scan's dictionary, its
// slice per (band, tile) and the rows of the bit matrix they get are built here, once: a pass maps its dense query hashes onto
// those fixed rows (launch_classify), needs no sorted dictionary, windows or row compaction for them, and -- the point -- its scan
// no longer depends on the batch at all: it is queued when the pass's first batch is (batch_front), beside the sketches.  A lone
// batch used to run sketch -> dictionary -> scan -> ranking strictly one after the other.  The scan still streams 8 x s x N bytes
// per pass.  Needs: the index; every rare row's bits obtainable without M (rare_direct); few enough dense hashes for the lean scan
// kernel's slices.  Else the reference keeps per-pass dictionaries, as in rounds 1-5.
static void build_static_dense(skx_ref* r, const std::vector<u64>& exc_h, const std::vector<u64>* spmask) {
    static const int sd_env = skx::knob("SKX_STATIC_DENSE") ? atoi(skx::knob("SKX_STATIC_DENSE")) : 1;  // experiment knob: 0 = per-pass dictionaries
    if (!sd_env || !r->d_kt_key || !r->rare_direct) return;
    const u64 slots = (u64)r->kt_mask + 1;
    const u32 n_sp = r->n_species;
    // (the lean kernel's one-pass probe takes slices of 254 entries; a (band, tile) slice holds ~3.1 x rows per band x |Q of its species| / s
    // of them.  Beyond ~190 entries per band's worth the host would pick the split / big-table variants: such references keep per-pass
    // dictionaries)
    const u64 cap_sp = (u64)191 * r->s / std::max<u32>(r->rb, 1u), cap = std::min<u64>(131072, cap_sp * n_sp);
    u64* d_keys = nullptr;
    u32 *d_slots = nullptr, *d_n = nullptr;
    bool ok = cap >= 1 && hipMalloc(&d_keys, (cap + 1) * 8) == hipSuccess && hipMalloc(&d_slots, (cap + 1) * 4) == hipSuccess &&
              hipMalloc(&d_n, 4) == hipSuccess && hipMemset(d_n, 0, 4) == hipSuccess;
    u32 n = 0;
    if (ok) {
        skx::launch_collect_dense(nullptr, r->d_kt_key, r->d_kt_off, slots, d_keys, d_slots, d_n, (u32)cap);
        ok = hipGetLastError() == hipSuccess && hipMemcpy(&n, d_n, 4, hipMemcpyDeviceToHost) == hipSuccess && n <= cap;
    }
    std::vector<u64> keys(n);
    std::vector<u32> kslot(n);
    ok = ok && (n == 0 || (hipMemcpy(keys.data(), d_keys, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess &&
                           hipMemcpy(kslot.data(), d_slots, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess));
    (void)hipFree(d_keys); (void)hipFree(d_slots); (void)hipFree(d_n);
    if (!ok) { (void)hipGetLastError(); return; }
    try {
        // every dense hash into its species' segment (a hash that several species hold was given a list by the index build: none is left here)
        std::vector<std::vector<std::pair<u64, u32>>> per(n_sp);
        for (u32 i = 0; i < n; ++i) {
            u32 sp = 0;
            if (n_sp > 1) {
                const u64 m = spmask ? (*spmask)[kslot[i]] : 0;
                if (m == 0 || (m & (m - 1)) != 0) return;  // (held by several species and too widely for a list: per-pass dictionaries)
                sp = (u32)__builtin_ctzll(m);
            }
            per[sp].push_back({keys[i], kslot[i]});
        }
        std::vector<u64> ex(exc_h);  // the lifted hashes ride with the dense rows (exceptions_kernel sets their bits): sorted tail
        std::sort(ex.begin(), ex.end());
        ex.erase(std::unique(ex.begin(), ex.end()), ex.end());
        std::vector<u32> seg(2 * n_sp), segw(2 * n_sp);
        u32 row = 0;
        for (u32 sp = 0; sp < n_sp; ++sp) {
            if (per[sp].size() > cap_sp) return;
            std::sort(per[sp].begin(), per[sp].end());
            seg[2 * sp] = row; seg[2 * sp + 1] = (u32)per[sp].size();
            segw[2 * sp] = row / 64; segw[2 * sp + 1] = (row + (u32)per[sp].size() + 63) / 64;
            row = (row + (u32)per[sp].size() + 63u) & ~63u;
        }
        // (the tail starts on the word boundary behind the last segment; without lifted hashes the dictionary ends with the last segment's last hash)
        const u32 t0 = row, n_sd = ex.empty() ? (n_sp ? seg[2 * (n_sp - 1)] + seg[2 * (n_sp - 1) + 1] : 0u) : t0 + (u32)ex.size();
        std::vector<u64> qs(std::max<u32>(n_sd, 1u), 0xFFFFFFFFFFFFFFFFull);   // (padding between the segments: all-ones, inside no window)
        std::vector<u32> srow(slots, 0xFFFFFFFFu);
        for (u32 sp = 0; sp < n_sp; ++sp)
            for (u32 i = 0; i < per[sp].size(); ++i) { qs[seg[2 * sp] + i] = per[sp][i].first; srow[per[sp][i].second] = seg[2 * sp] + i; }
        for (u32 i = 0; i < ex.size(); ++i) qs[t0 + i] = ex[i];
        const u32 n_bt = r->n_bands * r->n_tiles;
        const u32 words[4] = {n_sd, 0u, (n_sd + 63u) & ~63u, n_sd};
        ok = hipMalloc(&r->d_qs, (size_t)qs.size() * 8) == hipSuccess && hipMalloc(&r->d_nsd, 16) == hipSuccess &&
             hipMalloc(&r->d_win_s, (size_t)n_bt * 8) == hipSuccess && hipMalloc(&r->d_srow, (size_t)slots * 4) == hipSuccess &&
             hipMalloc(&r->d_seg, (size_t)2 * n_sp * 4) == hipSuccess && hipMalloc(&r->d_segw, (size_t)2 * n_sp * 4) == hipSuccess &&
             hipMemcpy(r->d_qs, qs.data(), qs.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(r->d_srow, srow.data(), (size_t)slots * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(r->d_seg, seg.data(), (size_t)2 * n_sp * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(r->d_segw, segw.data(), (size_t)2 * n_sp * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(r->d_nsd, words, 16, hipMemcpyHostToDevice) == hipSuccess;
        if (ok) {
            skx::launch_window_seg(nullptr, r->d_lo, r->d_hi, n_bt, r->n_tiles, r->d_qs, r->d_seg, r->d_grp_sp, r->d_win_s);
            ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
        }
        if (ok) { r->n_sd = n_sd; r->sd_tail0 = t0; r->static_dense = true; }
    } catch (const std::bad_alloc&) { ok = false; }
    if (!ok) {
        (void)hipFree(r->d_qs); (void)hipFree(r->d_nsd); (void)hipFree(r->d_win_s); (void)hipFree(r->d_srow); (void)hipFree(r->d_seg); (void)hipFree(r->d_segw);
        r->d_qs = nullptr; r->d_nsd = r->d_win_s = r->d_srow = r->d_seg = r->d_segw = nullptr;
        (void)hipGetLastError();
    }
}

SKX_API int skx_ref_create_multi(skx_ref** out, int device, uint32_t k, uint64_t seed, uint32_t s_read, uint32_t stride,
                                 uint32_t n_species, const uint32_t* n_genomes, const uint64_t* const* hashes,
                                 const uint32_t* const* col_len) {
    if (!out || !n_genomes || !hashes || !col_len) return fail(SKX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (s_read < 1) return fail(SKX_ERR_INVALID, "read sketch size s must be >= 1");
    const uint32_t s = stride;  // rows of the resident matrix
    if (k < 1 || k > SKX_MAX_K) return fail(SKX_ERR_INVALID, "k=%u outside 1..%u", k, SKX_MAX_K);
    if (n_species < 1 || n_species > SKX_MAX_SPECIES) return fail(SKX_ERR_INVALID, "n_species=%u outside 1..%u", n_species, SKX_MAX_SPECIES);
    if (s < 1) return fail(SKX_ERR_INVALID, "empty reference (stride=0)");
    u64 total = 0, total_pad = 0;
    for (u32 sp = 0; sp < n_species; ++sp) {
        if (!hashes[sp] || !col_len[sp]) return fail(SKX_ERR_INVALID, "NULL argument (species %u)", sp);
        if (n_genomes[sp] < 1) return fail(SKX_ERR_INVALID, "empty reference (species %u has no genomes)", sp);
        total += n_genomes[sp];
        total_pad += ((u64)n_genomes[sp] + kGroupGenomes - 1) / kGroupGenomes * kGroupGenomes;
    }
    if (total_pad >= (1ull << 31)) return fail(SKX_ERR_CAPACITY, "%llu genomes exceed what the kernels index", (unsigned long long)total);
    SKXCHK(use_device(device));

    skx_ref* r = new skx_ref;
    r->device = device; r->k = k; r->seed = seed; r->s = s; r->s_read = s_read; r->n_genomes = (u32)total; r->n_species = n_species;
    r->n_pad = (u32)total_pad;
    r->n_tiles = r->n_pad / skx::kTileGenomes;
    static const int rb_env = skx::knob("SKX_RB") ? atoi(skx::knob("SKX_RB")) : 0;  // experiment knob
    r->rb = rb_env >= 8 ? (u32)rb_env : 64;  // rows per band (measured best of 64/128/256/512 on MI355X)
    r->n_bands = (s + r->rb - 1) / r->rb;
    r->min_species = 0xFFFFFFFFu;
    {
        u32 g0 = 0, real0 = 0;
        for (u32 sp = 0; sp < n_species; ++sp) {
            r->sp_n.push_back(n_genomes[sp]); r->sp_g0.push_back(g0); r->sp_real0.push_back(real0);
            r->min_species = std::min(r->min_species, n_genomes[sp]);
            r->max_species = std::max(r->max_species, n_genomes[sp]);
            g0 += (n_genomes[sp] + kGroupGenomes - 1) / kGroupGenomes * kGroupGenomes;
            real0 += n_genomes[sp];
        }
    }

    // validate (the reference assumes ascending columns, src/sketchy.rs:416-418; here it is checked)
    std::vector<u32> eff(total);
    std::vector<u32> exc_g;
    std::vector<u64> exc_h;
    u64 max_ref = 0;
    bool any = false;
    for (u32 sp = 0; sp < n_species; ++sp) {
        for (u32 g = 0; g < n_genomes[sp]; ++g) {
            const u32 len = col_len[sp][g];
            if (len > s) { delete r; return fail(SKX_ERR_INVALID, "species %u: col_len[%u]=%u exceeds stride=%u", sp, g, len, s); }
            const uint64_t* col = hashes[sp] + (size_t)g * s;
            for (u32 i = 1; i < len; ++i)
                if (col[i] <= col[i - 1]) { delete r; return fail(SKX_ERR_UNSORTED, "species %u genome %u: hashes not strictly ascending at %u", sp, g, i); }
            u32 e = len;
            while (e > 0 && col[e - 1] >= skx::kEmpty) {  // would alias table markers: handled as exceptions
                exc_g.push_back(r->sp_g0[sp] + g); exc_h.push_back(col[e - 1]); --e;
            }
            eff[r->sp_real0[sp] + g] = e;
            if (e > 0) { any = true; max_ref = std::max<u64>(max_ref, col[e - 1]); }
        }
    }
    if (!exc_h.empty()) { any = true; max_ref = std::max<u64>(max_ref, *std::max_element(exc_h.begin(), exc_h.end())); }
    r->max_ref = max_ref; r->any = any;
    r->n_exc = (u32)exc_h.size();

    hipError_t e;
    u32* d_eff = nullptr;
    u64* d_stage = nullptr;
#define RCHK(expr) do { e = (expr); if (e != hipSuccess) { (void)hipFree(d_eff); (void)hipFree(d_stage); ref_free(r); return fail(SKX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e)); } } while (0)
    const size_t mat_elems = (size_t)r->n_tiles * s * skx::kTileGenomes;
    {
        // experiment knob SKX_MAT_ALLOC=1: physically contiguous memory for the matrix (hipDeviceMallocContiguous: large page-table
        // fragments -- the TLB-reach test of the 12 GB scan)
        static const int mat_alloc = skx::knob("SKX_MAT_ALLOC") ? atoi(skx::knob("SKX_MAT_ALLOC")) : 0;
        if (mat_alloc == 1) {
            if (hipExtMallocWithFlags((void**)&r->d_mat, mat_elems * 8, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); r->d_mat = nullptr; }
        }
        if (!r->d_mat) RCHK(hipMalloc(&r->d_mat, mat_elems * 8));
    }
    RCHK(hipMemset(r->d_mat, 0xFF, mat_elems * 8));  // kPad everywhere: short columns, padding genomes
    RCHK(hipMalloc(&r->d_lo, (size_t)r->n_bands * r->n_tiles * 8));
    RCHK(hipMalloc(&r->d_hi, (size_t)r->n_bands * r->n_tiles * 8));
    RCHK(hipMalloc(&d_eff, (size_t)total * 4));
    RCHK(hipMemcpy(d_eff, eff.data(), (size_t)total * 4, hipMemcpyHostToDevice));
    // species tables
    {
        const u32 n_grp = r->n_pad / kGroupGenomes;
        std::vector<u32> grp_sp(n_grp), real2pad(total);
        for (u32 sp = 0; sp < n_species; ++sp) {
            const u32 g_end = sp + 1 < n_species ? r->sp_g0[sp + 1] : r->n_pad;
            for (u32 grp = r->sp_g0[sp] / kGroupGenomes; grp < g_end / kGroupGenomes; ++grp) grp_sp[grp] = sp;
            for (u32 g = 0; g < n_genomes[sp]; ++g) real2pad[r->sp_real0[sp] + g] = r->sp_g0[sp] + g;
        }
        RCHK(hipMalloc(&r->d_sp_g0, (size_t)n_species * 4));
        RCHK(hipMalloc(&r->d_sp_n, (size_t)n_species * 4));
        RCHK(hipMalloc(&r->d_grp_sp, (size_t)n_grp * 4));
        RCHK(hipMalloc(&r->d_real2pad, (size_t)total * 4));
        RCHK(hipMemcpy(r->d_sp_g0, r->sp_g0.data(), (size_t)n_species * 4, hipMemcpyHostToDevice));
        RCHK(hipMemcpy(r->d_sp_n, r->sp_n.data(), (size_t)n_species * 4, hipMemcpyHostToDevice));
        RCHK(hipMemcpy(r->d_grp_sp, grp_sp.data(), (size_t)n_grp * 4, hipMemcpyHostToDevice));
        RCHK(hipMemcpy(r->d_real2pad, real2pad.data(), (size_t)total * 4, hipMemcpyHostToDevice));
    }
    // upload in chunks of whole tiles (bounded staging), re-tiling on the device
    const u32 chunk_tiles = std::max<u32>(1u, (u32)((256ull << 20) / ((size_t)skx::kTileGenomes * s * 8)));
    const u32 chunk_g = chunk_tiles * skx::kTileGenomes;
    RCHK(hipMalloc(&d_stage, (size_t)chunk_g * s * 8));
    for (u32 sp = 0; sp < n_species; ++sp) {
        for (u32 g0 = 0; g0 < n_genomes[sp]; g0 += chunk_g) {  // (chunks start on tile boundaries of the padded order)
            const u32 cnt = std::min<u32>(chunk_g, n_genomes[sp] - g0);
            RCHK(hipMemcpy(d_stage, hashes[sp] + (size_t)g0 * s, (size_t)cnt * s * 8, hipMemcpyHostToDevice));
            skx::launch_ref_tile(nullptr, d_stage, d_eff + r->sp_real0[sp] + g0, r->d_mat, s, r->sp_g0[sp] + g0, cnt);
            RCHK(hipGetLastError());
            RCHK(hipDeviceSynchronize());
        }
    }
    (void)hipFree(d_stage); d_stage = nullptr;
    (void)hipFree(d_eff); d_eff = nullptr;
    skx::launch_band_bounds(nullptr, r->d_mat, s, r->n_tiles, r->rb, r->n_bands, r->d_lo, r->d_hi);
    RCHK(hipGetLastError());
    if (r->n_exc) {
        RCHK(hipMalloc(&r->d_exc_g, (size_t)r->n_exc * 4));
        RCHK(hipMalloc(&r->d_exc_h, (size_t)r->n_exc * 8));
        RCHK(hipMemcpy(r->d_exc_g, exc_g.data(), (size_t)r->n_exc * 4, hipMemcpyHostToDevice));
        RCHK(hipMemcpy(r->d_exc_h, exc_h.data(), (size_t)r->n_exc * 8, hipMemcpyHostToDevice));
    }
    {
        // Membership filter (skx_common.hpp: a blocked Bloom filter, four bits per key inside one 64-bit word).  Sized from the
        // number of DISTINCT reference hashes -- the genomes of a collection share most of theirs: C2 holds 4e8 hashes, ~6e6
        // distinct -- which a linear-counting pass over the matrix gives first (a transient bitmap of 2^32 bits at most;
        // n = -B ln(1 - fill) corrects for the collisions).  Policy "filter_bits_per_hash" (default 32 per distinct hash:
        // ~0.03 % false positives, each one an all-zero row of the bit matrices and an entry of every slice it falls into;
        // rounds 1-2 spent 64 bits per NON-distinct hash on a direct-mapped bitmap: 4 GB at C2 where this takes 32 MB).
        const u32 max_bits = 64u - (u32)__builtin_clzll(max_ref | 1ull);
        u64 distinct = 0;
        {
            u32 lg = 16;
            while (lg < 32 && (1ull << lg) < 8ull * s * total) ++lg;
            const u32 shift = max_bits > lg ? max_bits - lg : 0u;
            u64* d_tmp = nullptr;
            unsigned long long* d_cnt = nullptr;
            RCHK(hipMalloc(&d_tmp, (1ull << lg) / 8));
            e = hipMalloc(&d_cnt, 8);
            if (e == hipSuccess) e = hipMemset(d_tmp, 0, (1ull << lg) / 8);
            if (e == hipSuccess) e = hipMemset(d_cnt, 0, 8);
            if (e == hipSuccess) {
                skx::launch_filter_build(nullptr, r->d_mat, mat_elems, shift, d_tmp, false, d_cnt);
                skx::launch_filter_build(nullptr, r->d_exc_h, r->n_exc, shift, d_tmp, true, d_cnt);
                e = hipGetLastError();
            }
            unsigned long long set_bits = 0;
            if (e == hipSuccess) e = hipMemcpy(&set_bits, d_cnt, 8, hipMemcpyDeviceToHost);
            (void)hipFree(d_tmp); (void)hipFree(d_cnt);
            RCHK(e);
            const double B = (double)(1ull << lg), fill = std::min(0.999999, (double)set_bits / B);
            distinct = (u64)(-B * std::log1p(-fill)) + 1;
        }
        r->n_distinct = distinct;
        static const u32 lg_cap = skx::knob("SKX_FILTER_LG") ? (u32)std::min(36, std::max(16, atoi(skx::knob("SKX_FILTER_LG")))) : 36u;  // experiment knob
        const u64 want = (u64)g_filter_bits_per_hash * distinct;
        u32 lg = 16;
        while (lg < lg_cap && (1ull << lg) < want) ++lg;
        const u32 lg_words = lg - 6;  // 64-bit words
        r->filt_shift = max_bits > lg_words ? max_bits - lg_words : 0u;
        r->filt_bits = 1ull << lg;
        RCHK(hipMalloc(&r->d_filt, r->filt_bits / 8));
        RCHK(hipMemset(r->d_filt, 0, r->filt_bits / 8));
        skx::launch_filter_build(nullptr, r->d_mat, mat_elems, r->filt_shift, r->d_filt, false);
        skx::launch_filter_build(nullptr, r->d_exc_h, r->n_exc, r->filt_shift, r->d_filt, true);
        RCHK(hipGetLastError());
    }
    RCHK(hipDeviceSynchronize());
    {
        // Rare-hash index (skx_kernels.hip, "rare-hash index of the reference"; policy "rare_hash_genomes", 0 = none): which genomes
        // hold a hash that only a few hold -- so that a pass asks the scan for the hashes many genomes share and nothing else.
        // Anything that goes wrong here (no memory, more list entries than 32 bits index) leaves the reference without the index:
        // every hash then goes to the scan, as in rounds 1-4.
        static const int rare_env = skx::knob("SKX_RARE_MAX") ? atoi(skx::knob("SKX_RARE_MAX")) : -1;  // experiment knob
        const u32 rare_max = rare_env >= 0 ? (u32)rare_env : g_rare_hash_genomes.load();
        if (rare_max && any) {
            u64 slots = 1024;
            while (slots < (u64)((double)r->n_distinct * 1.6) + 1024) slots <<= 1;
            u32* d_over = nullptr;
            u32* d_cursor = nullptr;
            u64* d_spm = nullptr;
            auto drop = [&]() {
                (void)hipFree(r->d_kt_key); (void)hipFree(r->d_kt_cnt); (void)hipFree(r->d_kt_off); (void)hipFree(r->d_post);
                r->d_kt_key = nullptr; r->d_kt_cnt = r->d_kt_off = r->d_post = nullptr;
                (void)hipGetLastError();
            };
            bool ok = hipMalloc(&d_over, 4) == hipSuccess;
            for (int attempt = 0; ok && attempt < 4; ++attempt, slots <<= 1) {
                if (slots > (1ull << 31)) { ok = false; break; }
                ok = hipMalloc(&r->d_kt_key, slots * 8) == hipSuccess && hipMalloc(&r->d_kt_cnt, slots * 4) == hipSuccess &&
                     hipMemset(r->d_kt_key, 0xFF, slots * 8) == hipSuccess && hipMemset(r->d_kt_cnt, 0, slots * 4) == hipSuccess &&
                     hipMemset(d_over, 0, 4) == hipSuccess;
                // (several species: which of them hold a key -- a hash many genomes of SEVERAL species hold gets a list whatever its
                // count, so that every hash left to the scan belongs to one species: build_static_dense)
                if (ok && n_species > 1) ok = hipMalloc(&d_spm, slots * 8) == hipSuccess && hipMemset(d_spm, 0, slots * 8) == hipSuccess;
                if (!ok) break;
                skx::launch_rare_count(nullptr, r->d_mat, mat_elems, r->d_kt_key, r->d_kt_cnt, (u32)(slots - 1), d_over, d_spm, r->d_grp_sp, s);
                u32 over = 0;
                ok = hipGetLastError() == hipSuccess && hipMemcpy(&over, d_over, 4, hipMemcpyDeviceToHost) == hipSuccess;
                if (ok && !over) break;
                drop();  // (the distinct count was an estimate: twice the slots)
                (void)hipFree(d_spm); d_spm = nullptr;
                if (attempt == 3) ok = false;
            }
            std::vector<u64> spm;
            if (ok && r->d_kt_key) try {
                std::vector<u32> cnt(slots), off(slots);
                ok = hipMemcpy(cnt.data(), r->d_kt_cnt, slots * 4, hipMemcpyDeviceToHost) == hipSuccess;
                if (ok && d_spm) { spm.resize(slots); ok = hipMemcpy(spm.data(), d_spm, slots * 8, hipMemcpyDeviceToHost) == hipSuccess; }
                (void)hipFree(d_spm); d_spm = nullptr;
                u64 total = 0, keys = 0, rare = 0, long_keys = 0, forced = 0;
                for (u64 i = 0; ok && i < slots; ++i) {
                    // (a hash of several species: listed up to 2^18 genomes; a handful at k = 16 -- unrelated species share ~0.1 % of their k-mers)
                    const bool multi = !spm.empty() && (spm[i] & (spm[i] - 1)) != 0 && cnt[i] > rare_max && cnt[i] <= (1u << 18);
                    if (cnt[i] && (cnt[i] <= rare_max || multi)) { off[i] = (u32)total; total += cnt[i]; ++rare; long_keys += cnt[i] > 8u ? 1 : 0; forced += multi ? 1 : 0; }
                    else off[i] = 0xFFFFFFFFu;
                    keys += cnt[i] ? 1 : 0;
                }
                r->n_forced_rare = forced;
                ok = ok && total < 0xFFFFFFF0ull;
                ok = ok && hipMalloc(&r->d_kt_off, slots * 4) == hipSuccess && hipMalloc(&r->d_post, std::max<u64>(total, 1) * 4) == hipSuccess &&
                     hipMalloc(&d_cursor, slots * 4) == hipSuccess && hipMemset(d_cursor, 0, slots * 4) == hipSuccess &&
                     hipMemcpy(r->d_kt_off, off.data(), slots * 4, hipMemcpyHostToDevice) == hipSuccess;
                if (ok) {
                    skx::launch_rare_fill(nullptr, r->d_mat, mat_elems, s, r->d_kt_key, r->d_kt_off, d_cursor, r->d_post, (u32)(slots - 1));
                    ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
                }
                if (ok) { r->kt_mask = (u32)(slots - 1); r->rare_max = rare_max; r->n_keys = keys; r->n_rare_keys = rare; r->n_postings = total; }
                // long lists as bit rows over the genomes (at most an eighth of the free memory; without them the lists are walked)
                static const int mlong_env = skx::knob("SKX_LONG_ROWS") ? atoi(skx::knob("SKX_LONG_ROWS")) : 1;  // experiment knob
                if (ok && mlong_env) {
                    std::vector<u32> lid(slots, 0xFFFFFFFFu), lslot;
                    for (u64 i = 0; i < slots; ++i)
                        if (off[i] != 0xFFFFFFFFu && cnt[i] > 8u) { lid[i] = (u32)lslot.size(); lslot.push_back((u32)i); }
                    const u64 n_long = lslot.size(), row_bytes = (u64)r->n_pad / 8;
                    size_t mem_free = 0, mem_total = 0;
                    (void)hipMemGetInfo(&mem_free, &mem_total);
                    bool lok = n_long > 0 && n_long * row_bytes <= mem_free / 8 &&
                               hipMalloc(&r->d_mlong, n_long * row_bytes) == hipSuccess && hipMemset(r->d_mlong, 0, n_long * row_bytes) == hipSuccess &&
                               hipMalloc(&r->d_lid, slots * 4) == hipSuccess && hipMalloc(&r->d_lslot, n_long * 4) == hipSuccess &&
                               hipMemcpy(r->d_lid, lid.data(), slots * 4, hipMemcpyHostToDevice) == hipSuccess &&
                               hipMemcpy(r->d_lslot, lslot.data(), n_long * 4, hipMemcpyHostToDevice) == hipSuccess;
                    if (lok) {
                        skx::launch_mlong_build(nullptr, r->d_lslot, (u32)n_long, r->d_kt_off, r->d_kt_cnt, r->d_post, r->d_mlong, r->n_pad / 64);
                        lok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
                    }
                    if (lok) r->n_long = n_long;
                    // most of them as (pattern, exceptions): such rows need no transposed bit rows (their candidates come from the pattern)
                    if (lok) build_patterns(r);
                    const bool mostly_patterns = lok && r->d_prec && r->n_pat_lists * 10 >= n_long * 9;
                    // ... and transposed (genome-major), when another eighth of what is free holds it -- unless nine lists in ten are patterns:
                    // the few bit rows a batch then lists are tested against the candidates row by row (C4's pool reference: 24 GB less)
                    static const int mlongt_env = skx::knob("SKX_LONG_ROWS_T") ? atoi(skx::knob("SKX_LONG_ROWS_T")) : 1;  // experiment knob (2: always)
                    if (lok && mlongt_env && (!mostly_patterns || mlongt_env == 2)) {
                        const u32 n_lw = (u32)((n_long + 63) / 64);
                        const u64 t_bytes = (u64)r->n_pad * n_lw * 8;
                        (void)hipMemGetInfo(&mem_free, &mem_total);
                        if (t_bytes <= mem_free / 8 && hipMalloc(&r->d_mlongT, t_bytes) == hipSuccess) {
                            skx::launch_mlong_transpose(nullptr, r->d_mlong, (u32)n_long, r->n_pad / 64, r->d_mlongT, n_lw);
                            if (hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess) r->n_lw = n_lw;
                            else { (void)hipFree(r->d_mlongT); r->d_mlongT = nullptr; }
                        } else r->d_mlongT = nullptr;
                        (void)hipGetLastError();
                    }
                    else if (!lok) { (void)hipFree(r->d_mlong); (void)hipFree(r->d_lid); (void)hipFree(r->d_lslot); r->d_mlong = nullptr; r->d_lid = r->d_lslot = nullptr; r->n_long = 0; (void)hipGetLastError(); }
                }
                r->rare_direct = ok && (r->d_mlong != nullptr || long_keys == 0);
            } catch (const std::bad_alloc&) {
                // (the host-side offsets of a very large key table did not fit the host's memory: no index, every hash goes to the scan)
                (void)hipFree(r->d_mlong); (void)hipFree(r->d_mlongT); (void)hipFree(r->d_lid); (void)hipFree(r->d_lslot);
                r->d_mlong = r->d_mlongT = nullptr; r->d_lid = r->d_lslot = nullptr; r->n_long = 0; r->n_lw = 0;
                (void)hipFree(r->d_prec); (void)hipFree(r->d_pat_rep); (void)hipFree(r->d_npat); (void)hipFree(r->d_pm);
                r->d_prec = r->d_pat_rep = r->d_npat = nullptr; r->d_pm = nullptr; r->n_pat = 0; r->n_pat_lists = 0;
                r->rare_direct = false;
                ok = false;
            }
            (void)hipFree(d_over); (void)hipFree(d_cursor);
            (void)hipFree(d_spm);
            if (!ok) { drop(); r->rare_direct = false; }
            else build_static_dense(r, exc_h, spm.empty() ? nullptr : &spm);
        }
    }
    const u32 pf_mode = skx::knob("SKX_KMER_PREFILTER") ? (u32)atoi(skx::knob("SKX_KMER_PREFILTER")) : g_kmer_prefilter.load();  // (experiment knob overrides the policy)
    // table bits per key, and how large a table still pays.  Every window of a read costs one random 4-byte gather, and that only
    // pays while the table sits in the CUs' L1 caches (32 KB).  Measured at C2 (98 304-read batches, eight batches per scan) with the
    // table blown up to the size a larger key set would need: no table 117 M reads/s; 16 KB 128 M; 128 KB 105 M; 1 MB 96 M; 4 MB
    // 86 M; 8 MB 59 M; 32 MB 38 M.  The SYNTHETIC C2 reference has only 11 600 keys (its variant hashes are random numbers, not
    // hashes of 16-mers; 8 bits per key = 16 KB) -- a real collection of that size (6e6 distinct hashes, every one a 16-mer's)
    // would need megabytes.  Hence: off by default; "2" builds the table only when it fits 32 KB (small panels).
    static const u32 kf_bits = skx::knob("SKX_KF_BITS") ? (u32)std::max(1, atoi(skx::knob("SKX_KF_BITS"))) : 8u;
    const u64 kf_auto_bits = 8ull << 15;  // 32 KB
    if (k == 16 && any && pf_mode) {
        // every canonical 16-mer whose hash passes the filter above (two passes over the 2^32 codes: count, then insert)
        u32* d_n = nullptr;
        RCHK(hipMalloc(&d_n, 4));
        RCHK(hipMemset(d_n, 0, 4));
        skx::launch_kmer_filter_build(nullptr, seed, max_ref, r->d_filt, r->filt_shift, d_n, nullptr, 0);
        RCHK(hipGetLastError());
        RCHK(hipMemcpy(&r->kf_keys, d_n, 4, hipMemcpyDeviceToHost));
        (void)hipFree(d_n);
        auto table_lg_words = [&](u32 bits) {
            u32 lg = 10;  // 4 KB at least
            while (lg < 27 && (32ull << lg) < (u64)bits * r->kf_keys) ++lg;
            return lg;
        };
        u32 lg_words = table_lg_words(kf_bits);
        if (pf_mode >= 2 && (32ull << lg_words) > kf_auto_bits) {  // auto: half the bits if that fits, else no table at all
            lg_words = table_lg_words(std::max(1u, kf_bits / 2));
            if ((32ull << lg_words) > kf_auto_bits) { lg_words = 0; r->kf_keys = 0; }
        }
        if (lg_words) {
            r->kf_shift = 32u - lg_words;
            RCHK(hipMalloc(&r->d_kf, (size_t)4 << lg_words));
            RCHK(hipMemset(r->d_kf, 0, (size_t)4 << lg_words));
            skx::launch_kmer_filter_build(nullptr, seed, max_ref, r->d_filt, r->filt_shift, nullptr, r->d_kf, r->kf_shift);
            RCHK(hipGetLastError());
            RCHK(hipDeviceSynchronize());
        }
    }
#undef RCHK
    *out = r;
    return SKX_OK;
}
SKX_API int skx_ref_create(skx_ref** out, int device, uint32_t k, uint64_t seed, uint32_t s, uint32_t stride, uint32_t n_genomes,
                           const uint64_t* hashes, const uint32_t* col_len) {
    if (!out || !hashes || !col_len) return fail(SKX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (s < 1 || stride < 1 || n_genomes < 1)
        return fail(SKX_ERR_INVALID, "empty reference (s=%u, stride=%u, n_genomes=%u)", s, stride, n_genomes);
    return skx_ref_create_multi(out, device, k, seed, s, stride, 1, &n_genomes, &hashes, &col_len);
}
SKX_API int skx_ref_sketch_size(const skx_ref* ref, uint32_t* s, uint32_t* stride) {
    if (!ref) return fail(SKX_ERR_INVALID, "NULL argument");
    if (s) *s = ref->s_read;
    if (stride) *stride = ref->s;
    return SKX_OK;
}
SKX_API int skx_ref_n_genomes(const skx_ref* ref, uint32_t* n_genomes) {
    if (!ref || !n_genomes) return fail(SKX_ERR_INVALID, "NULL argument");
    *n_genomes = ref->n_genomes;
    return SKX_OK;
}
SKX_API int skx_ref_n_species(const skx_ref* ref, uint32_t* n_species) {
    if (!ref || !n_species) return fail(SKX_ERR_INVALID, "NULL argument");
    *n_species = ref->n_species;
    return SKX_OK;
}
SKX_API int skx_ref_species_genomes(const skx_ref* ref, uint32_t species, uint32_t* n_genomes) {
    if (!ref || !n_genomes) return fail(SKX_ERR_INVALID, "NULL argument");
    if (species >= ref->n_species) return fail(SKX_ERR_INVALID, "species %u outside 0..%u", species, ref->n_species - 1);
    *n_genomes = ref->sp_n[species];
    return SKX_OK;
}
SKX_API int skx_ref_kmer_filter(const skx_ref* ref, uint64_t* n_keys, uint64_t* table_bytes) {
    if (!ref) return fail(SKX_ERR_INVALID, "NULL argument");
    if (n_keys) *n_keys = ref->d_kf ? ref->kf_keys : 0;
    if (table_bytes) *table_bytes = ref->d_kf ? (4ull << (32u - ref->kf_shift)) : 0;
    return SKX_OK;
}
SKX_API int skx_ref_rare_index(const skx_ref* ref, uint64_t* n_keys, uint64_t* n_rare_keys, uint64_t* n_postings, uint64_t* bytes) {
    if (!ref) return fail(SKX_ERR_INVALID, "NULL argument");
    const bool on = ref->d_kt_key != nullptr;
    if (n_keys) *n_keys = on ? ref->n_keys : 0;
    if (n_rare_keys) *n_rare_keys = on ? ref->n_rare_keys : 0;
    if (n_postings) *n_postings = on ? ref->n_postings : 0;
    if (bytes) *bytes = on ? ((u64)ref->kt_mask + 1) * 16 + std::max<u64>(ref->n_postings, 1) * 4 + (ref->d_mlong ? ref->n_long * (ref->n_pad / 8 + 4) + ((u64)ref->kt_mask + 1) * 4 : 0) + (ref->d_mlongT ? (u64)ref->n_pad * ref->n_lw * 8 : 0) : 0;
    return SKX_OK;
}
SKX_API int skx_ref_patterns(const skx_ref* ref, uint64_t* n_long_lists, uint64_t* n_patterns, uint64_t* n_pattern_lists, uint64_t* bytes) {
    if (!ref) return fail(SKX_ERR_INVALID, "NULL argument");
    if (n_long_lists) *n_long_lists = ref->d_mlong ? ref->n_long : 0;
    if (n_patterns) *n_patterns = ref->d_prec ? ref->n_pat : 0;
    if (n_pattern_lists) *n_pattern_lists = ref->d_prec ? ref->n_pat_lists : 0;
    if (bytes) *bytes = ref->d_prec ? ref->n_long * skx::pat_record_words() * 4 + (u64)((ref->n_pat + 63) / 64) * ref->n_pad * 8 + (u64)ref->n_pat * 4 : 0;
    return SKX_OK;
}
SKX_API int skx_ref_static_dense(const skx_ref* ref, int* is_static, uint64_t* n_hashes) {
    if (!ref) return fail(SKX_ERR_INVALID, "NULL argument");
    if (is_static) *is_static = ref->static_dense ? 1 : 0;
    if (n_hashes) *n_hashes = ref->static_dense ? ref->n_sd : 0;
    return SKX_OK;
}
SKX_API int skx_ref_pass_bytes(const skx_ref* ref, uint64_t* bytes) {
    if (!ref || !bytes) return fail(SKX_ERR_INVALID, "NULL argument");
    *bytes = 8ull * ref->s * ref->n_genomes;
    return SKX_OK;
}
SKX_API void skx_ref_destroy(skx_ref* ref) { ref_free(ref); }

// ------------------------------------------------------------------ stream
struct TimedSpan { int stage; hipEvent_t a, b; };

// A batch between its two halves.  batch_front queues the sketcher, the pair counts, the speculative pair gather and the
// published summary on the sketch stream; batch_back waits for the summary (the one host wait of a batch) and queues the
// passes.  skx_stream_push* run both halves in one call; skx_stream_enqueue_device runs the front half of batch i + 1
// BEFORE the back half of batch i, so the sketch stream never waits for the host between two batches.
static const int kGroupMax = skx::kPairBaseMax + 1;  // batches that may share one pass (option stream_coalesce)
struct PendingBatch {
    bool valid = false;
    int side = 0;  // which copy of the sketch buffers holds it
    const uint8_t* d_bases = nullptr;
    const u64* d_offsets = nullptr;
    u32 n_reads = 0;
    u64 n_bases = 0;
    u32* d_topk_idx = nullptr;
    u64* d_topk_sum = nullptr;
    u32 seq = 0;               // sequence number its summary is published under
    bool spec_insert = false;  // its pairs were gathered into buffer set spec_set / pair slot spec_slot right behind the sketcher
    int spec_set = 0, spec_slot = 0;
    bool pairable = false;     // it came through an entry point whose batches may share a pass (enqueue / submit, stream_coalesce >= 2)
    u32 max_group = 1;         // ... at most so many of them
    int gi = 0;                // its place in the group of batches that share a pass (0: it opened one): same set / slot as the
                               // group's first batch, its pairs behind those of the gi batches before it ...
    int prev_side[kGroupMax - 1] = {};   // ... whose sides hold the pair counts its gather starts behind
    u32 prev_reads[kGroupMax - 1] = {};
    bool inrange_only = true;
    bool rows_mode = false;    // its sketch rows are full-width rows (debug outputs / no filter), not reservations out of the pool
    u32 dbg_cap = 0xFFFFFFFFu;
    u32* h_shared = nullptr;   // host outputs of the synchronous parity / debug path
    u64* h_sketches = nullptr;
    u32* h_sketch_len = nullptr;
    void* slot = nullptr;      // skx_stream::Staged of a host-fed batch: its rows travel to the host behind the ranking
};

// The HIP streams of the pipeline, ONE set per device shared by every skx_stream on it.  Measured (tools/diag_second_stream.py,
// C2): a second skx_stream with four HIP streams of its own ran at 63 M reads/s beside a first one's 78 M -- its sketch and its
// scan took their stand-alone times, i.e. never overlapped: the runtime maps HIP streams onto 4 hardware queues per process
// (GPU_MAX_HW_QUEUES) and the second set aliased.  Kernels of different skx_streams queue behind each other on the shared HIP
// streams in host order; every skx_stream orders its own work with its own events, as before.  (A skx_stream_sync therefore
// also waits for work queued by the device's other streams.)
// lane_s[i]: the stream of ranking lane i + 1 (skx_stream::RankLane; lane 0 is hs2), created when the first stream that enqueues asks for it
struct SharedQueues { hipStream_t hs = nullptr, hs0 = nullptr, hs1 = nullptr, hs2 = nullptr, lane_s[kRankLanesMax - 1] = {}; int refs = 0; };
static std::mutex g_queues_mu;
static SharedQueues g_queues[64];
static void release_queues(int device) {
    std::lock_guard<std::mutex> lk(g_queues_mu);
    SharedQueues& q = g_queues[device & 63];
    if (--q.refs > 0) return;
    for (hipStream_t& h : q.lane_s) { if (h) (void)hipStreamDestroy(h); h = nullptr; }
    for (hipStream_t* h : {&q.hs1, &q.hs2, &q.hs0, &q.hs}) { if (*h) (void)hipStreamDestroy(*h); *h = nullptr; }
    q.refs = 0;
}
// A pass normally ranks ONE run of reads; batches enqueued back to back (up to stream_coalesce of them) share a pass (one
// dictionary, one scan of the reference, one transpose) and are ranked one after the other from it: `subs` lists them -- their pairs sit one behind the
// other in the pass's pair lists (p_off), each with its own pair offsets (d_poff, relative to its own first pair), reads and
// output rows.  The scan is the only cost of a step that does not grow with the reads: n batches per scan is what a batch
// of n times the size would give, without asking the caller for it.
struct SubPass {
    u32 ra = 0, rb = 0;           // reads [ra, rb) of the batch
    u32 p_off = 0, P = 0;         // its pairs: [p_off, p_off + P) of the pass's pair lists
    u32 p_base = 0;               // value of the batch's pair offset at read ra (non-speculative passes: absolute offsets)
    const u32* d_poff = nullptr;  // inserted passes: the copy of the batch's pair offsets the front half left in the slot
    u32* d_topk_idx = nullptr;
    u64* d_topk_sum = nullptr;
    u32* d_shared = nullptr;      // [rb-ra][n_genomes] or NULL
    int side = 0;
    void* slot = nullptr;         // host-fed batches: the staging slot whose rows go back to the host behind the ranking
};
static const int kSides = kGroupMax + 1;  // copies of the per-batch sketch outputs: the enqueued batches waiting for their shared pass + the one being sketched
                                          // (a stream uses stream_coalesce + 1 of them, allocated at first use)
static const u32 kStagedGroupMax = 4;
static const int kSlotsMax = 2 * kStagedGroupMax + 1;  // staging slots of the host-fed pipeline
struct skx_stream {
    const skx_ref* ref = nullptr;
    int device = 0;
    // Three HIP streams form a pipeline over passes, so the stages of consecutive batches overlap:
    //   hs0  sketch + dictionary (small VALU/latency-bound kernels; independent of the previous batch)
    //   hs   reference scan + bit transpose (HBM-bound)
    //   hs2  running table + per-read ranking (VALU-bound)
    // Everything handed from one stage to the next (Q, windows, pair lists, Mq, the pass's slice of the pair
    // offsets) is double-buffered; events order the hand-offs and the reuse of a buffer set two passes later.
    hipStream_t hs0 = nullptr, hs = nullptr, hs2 = nullptr;  // hs0 / hs2 alias hs at lower pipeline depths
    // hs1: what follows a batch's MAIN sketch kernel and nothing else depends on until the passes are queued -- long-read
    // merge, 2048-slot retry, pair counts / offsets, speculative pair gather, published summary.  On their own stream (behind
    // ev_main) so that the sketch stream runs main kernel after main kernel: those ~8 short, latency-bound kernels cost the
    // sketch stream 150-250 us per batch next to the other streams' work (kernel trace), a fifth of the step.
    hipStream_t hs1 = nullptr;  // aliases hs0 below pipeline depth 3
    bool tail_pass = false;     // the pass being queued closes a flush: no younger batch is sketched beside it
    hipEvent_t ev_main[kSides] = {};  // sketch stream: everything of the batch queued on hs0 is done (per side)
    int depth = 2;
    bool shared_queues = false;  // hs / hs0 / hs1 / hs2 (and the second ranking lane's stream) belong to the device's SharedQueues
    int buf = 0;
    hipEvent_t ev_dict[2] = {nullptr, nullptr}, ev_front[2] = {nullptr, nullptr}, ev_back[2] = {nullptr, nullptr};
    // per side of the sketch buffers (see d_sk below):
    hipEvent_t ev_sketch[kSides] = {};  // hs1: the batch's sketches, pair offsets (and speculative pair gather) are done
    hipEvent_t ev_skread[kSides] = {};  // scan stream: a pass that gathered its own pairs has read the sketch buffers
    bool sk_reader_pending[kSides] = {};
    int side = 0;                                  // the side d_sk ... h_chk currently name
    PendingBatch pend[kGroupMax];                  // skx_stream_enqueue_device: the batches whose back half is still to come (a group: they share a pass)
    int n_pend = 0;
    u32 group_cap = 1;       // batches a group may hold right now: stream_coalesce, lowered when groups turned out too large for a pass
    u32 clean_groups = 0;    // shared passes since it was last lowered
    double ppr_est = 0.0, qpr_est = 0.0;  // pairs / distinct query hashes per read of the recent batches (0: nothing seen yet)
    int side_next = 0;                             // sides are taken in turn: at most two pending batches + the one being sketched
    bool front_pending[2] = {false, false};
    bool back_pending[2] = {false, false};
    // The per-pass pair lists the ranking reads (read of every pair, the pass's pair offsets) rotate over THREE slots, and a
    // pass only waits for the ranking two passes back right before it overwrites what that ranking reads (pair -> query
    // index, Mq, group flags: after its scan).  So the gather / dictionary / scan of batch i + 1 never wait for the
    // ranking of batch i - 1, which is the longest chain of a step.
    hipEvent_t ev_pslot[3] = {nullptr, nullptr, nullptr};   // back stream: the pass using the slot has been ranked
    bool pslot_pending[3] = {false, false, false};
    int pslot = 0;                                          // slot of the next pass
    hipEvent_t ev_pairq[2] = {nullptr, nullptr};            // scan stream: the set's hash set / pair hashes have been consumed
    bool pairq_pending[2] = {false, false};
    bool packed = false;  // the stream's input is 4 bits per base (skx_stream_set_packed_input): offsets count bases
    u32 top_k = 0, max_reads = 0, sk_stride = 0;
    u64 max_bases = 0;
    u32 pcap = 0;        // pairs per pass
    u32 qcap = 0;        // distinct query hashes per pass = rows of the pass's bit matrices (|Q| <= pairs, usually far below)
    // the split dictionary of a pass (references with a rare-hash index; all on the scan stream, one copy): the hashes the scan looks
    // for, position in Q -> row, key-table slots of the other rows, scratch of the classify kernels; d_nd = {dense rows, other rows}
    u64* d_qd = nullptr;
    u32 *d_qrow = nullptr, *d_sslot = nullptr, *d_qinfo = nullptr, *d_qloc = nullptr, *d_cls_bsum = nullptr;
    // {dense rows, other rows, first other row, rows in all} of the pass whose FRONT HALF is on the scan stream: scan-stream scratch, one
    // copy.  The ranking chains read the per-set copy ps[b].nd, which a pass writes only behind its wait for the chains that last used the
    // set (wait_back) -- round 5 let the dictionary stage write ps[b].nd directly, ahead of that wait: a chain of the pass two back that was
    // still running could have seen the new pass's counts.
    u32* d_nd_hs = nullptr;
    u32* h_lcount = nullptr;   // page-locked: {candidates the latest batch of a legacy pass would have had, its sequence number}
    u32 lcount_seq = 0, lcount_seen = 0, lcount_floor = 0;  // (floor: counts older than the last reset do not count)
    u32* h_nq_sink = nullptr;  // page-locked: where the live sample of a compact chain goes (nobody reads it)
    u32* h_nd = nullptr;     // page-locked: [2 b] = |Q| of buffer set b's latest pass, [2 b + 1] = its dense rows
    // ---- the table without the ranking, the candidates of a batch (skx_kernels.hip; scan stream unless said otherwise)
    u32 *d_rowcnt = nullptr;     // [kPassBatchesMax][qcap] occurrences of every row among a batch's pairs
    u32 *d_gain = nullptr;       // [kPassBatchesMax][n_pad]
    u32 *d_gain_s = nullptr;     // the rare rows' part: [kPassBatchesMax][n_pad] entries gain_sparse_stride() words apart (references with the index)
    u32 *d_candslot = nullptr;   // [kPassBatchesMax][n_pad] candidate slot of a genome (or none)
    u32 *d_candmask = nullptr;   // [n_pad] batches of the pass a genome is a candidate of
    // long-list rows of a pass (references whose rare-hash index holds bit rows): the rows of every batch, their gains, the candidates by word
    uint2 *d_lrow = nullptr;     // [kPassBatchesMax][qcap + 128]
    u32 *d_nlrow = nullptr, *d_gain_l = nullptr, *d_cbase = nullptr, *d_cwl = nullptr, *d_ncwl = nullptr;
    u64 *d_cw = nullptr;
    u64 *d_inb = nullptr, *d_hit = nullptr;  // [kPassBatchesMax][ref->n_lw]: the bit rows on a batch's list / those that hold a candidate of it
    skx::LongRows long_rows() const { return skx::LongRows{d_lrow, d_nlrow, qcap + 128, d_inb, ref->n_lw}; }
    // rows whose list is a pattern + exceptions (references with patterns; a stream whose compact problems are wider than the pattern
    // kernels' LDS rows -- more than eight species -- does without: its dictionaries never flag a row)
    bool use_pat = false;
    u32 *d_hist = nullptr, *d_nprow = nullptr;  // [kPassBatchesMax][npat_pad] occurrences of every pattern per batch; rows listed per batch
    u64* d_pcw = nullptr;                        // [kPassBatchesMax][n_pat][n_grp_c * 8] the patterns at every batch's candidates
    u32 npat_pad = 0;
    skx::PatRows pat_rows() const { return skx::PatRows{d_hist, npat_pad, d_gain_l, d_nprow}; }
    skx::RareIndex rare_index() const {
        skx::RareIndex ri = ref->rare_index();
        if (!use_pat) { ri.prec = nullptr; ri.n_pat = 0; }
        if (!static_dense) { ri.qs = nullptr; ri.n_sd = 0; }
        return ri;
    }
    u32 *d_cbad = nullptr, *d_nqc = nullptr;  // [kPassBatchesMax] why a batch cannot rank compactly / mapped rare rows
    u32 *d_spc_g0 = nullptr, *d_spc_grp = nullptr;  // Species layout of the compact problems: kCandCap slots per species
    u32 n_pad_c = 0, n_grp_c = 0, cand_seq = 0;
    struct PassSet {  // per buffer set: what a pass hands to its ranking chains
        u64 *tab = nullptr;      // [kPassBatchesMax + 1][n_pad] the table as every batch begins, [n] = as the pass ends
        u32 *cand = nullptr;     // [kPassBatchesMax][n_pad_c]
        u64 *tabc = nullptr;     // [kPassBatchesMax][n_pad_c] the candidates' start values
        u32 *ncand = nullptr;    // [kPassBatchesMax][n_species]
        u32 *mode = nullptr, *any_full = nullptr, *nqc_total = nullptr;
        u64 *mc = nullptr;       // [kPassBatchesMax][kCandRows / 64][n_pad_c] the candidates' columns of M (dense words)
        u64 *mqc = nullptr;      // [kPassBatchesMax][n_grp_c][kCandRows][8] their group-major matrices
        u64 *rowany_c = nullptr; // [kPassBatchesMax][n_grp_c][kCandRows / 64]
        u32 *grp_any_c = nullptr;// [kPassBatchesMax][n_grp_c]
        u32 *smap = nullptr;     // [kPassBatchesMax][qcap] rare row -> row of the compact matrix + 1
        u32 *pair_qc = nullptr;  // [pcap] pairs of the compact batches in compact rows
        u32 *nd = nullptr;       // {dense rows, other rows, first other row, rows in all} of the pass's dictionary
        u32 *h_pub = nullptr;    // page-locked: modes, candidate counts, any_full, sequence number (cand_publish_kernel)
    } ps[2];
    // the ranking chains of the latest pass are queued once its candidates are known (queue_chains)
    struct PassChains {
        bool pending = false, ranked = false, has_cand = false, update_table = false, forced_full = false, legacy = false;
        SubPass subs[kGroupMax];
        int n_sub = 0, b = 0, slot = 0;
        u32 P = 0, nq_rows = 0, seq = 0;
        u64 nq_est = 0;
    } pcq[2];                 // FIFO: the passes whose chains are not queued yet (at most the latest two: one per buffer set)
    int pc_head = 0, pc_n = 0;
    // What the last pass whose candidates were published looked like: every batch with more candidates than a compact ranking takes
    // (the bench's near-tie: all 40 000 genomes, always) -> the next pass is told to rank on everything right away: its chains are
    // queued with the pass, nothing waits for a publication, the compact side's scratch is not even touched.  (Ranking on
    // everything is always exact; a sample that has just found its leader loses one pass of the compact ranking.)
    bool hint_all_overflow = false;
    u32 hint_seq = 0;  // sequence number of the pass the hint was taken from
    u32 passes_since_fresh = 0;  // passes queued since the table was last all zeros
    bool fresh_table = true;  // nothing has been added to the table since the stream was created / reset: every genome ties at zero
    u32 legacy_run = 0;  // passes in a row that took the table out of their ranking chains (rounds 1-4's way) instead
    u64 batches_compact = 0, batches_full = 0;  // batches ranked on their candidates / on everything (statistic)
    bool have_split_hint = false;
    double nd_frac = 1.0;    // dense rows / |Q| of the latest pass whose dictionary is known: which scan variant a pass gets
    u32 qcap_max = 0;        // rows the matrices may grow to (an eighth of the free device memory at creation)
    bool qcap_auto = false;  // no "stream_query_rows" policy: the matrices grow when a batch holds more distinct hashes (grow_query_rows)
    u64 qrows_grown = 0;     // how often they did (statistic)
    u32 coalesce = 1;    // enqueued batches that may share a pass (policy stream_coalesce at creation)
    u32 rpass = 0;       // reads per pass
    u64 reads_total = 0;
    // staging for host pushes
    uint8_t* d_bases = nullptr;
    u64* d_offsets = nullptr;
    // sketch outputs for the whole batch.  Two copies ("sides"): a batch queued with skx_stream_enqueue_device keeps its
    // side until its passes are queued, while the next batch is sketched into the other.  d_sk ... d_big / h_chk name the
    // side in use (use_side); side 1 is allocated by the first call that needs it.
    u64* d_sk = nullptr;
    u32 *d_len = nullptr, *d_cnt = nullptr, *d_poff = nullptr;
    bool side_ready[kSides] = {};  // the set is completely allocated (alloc_side is all or nothing)
    u64* sd_sk[kSides] = {};       // POOL of a side: production rows are exact-size reservations (sketch_finish, pool mode)
    u64 pool_cap[kSides] = {};        // ... its entries: a fixed slot per read first (pool_fixed), the reservable part behind
    u32 pool_fixed = 0;
    u64* sd_rows[kSides] = {};    // full-width rows [max_reads][sk_stride] of a side: debug outputs / skx_common_hashes; allocated on first use
    u32 cur_stride = 0;                      // what the kernels get as row stride for d_sk: 0 = pool mode
    u32 cur_pool_cap = 0, cur_pool_fixed = 0;
    u32 *sd_len[kSides] = {}, *sd_cnt[kSides] = {}, *sd_poff[kSides] = {}, *sd_big[kSides] = {};
    u32 *sd_chk[kSides] = {}, *sd_retry[kSides] = {};  // (per side: batch i's summary is published while batch i + 1 is sketched)
    skx::LongReads sd_lr[kSides] = {};
    // pass workspace (per buffer set: pair hashes, hash set and its counters, Q, windows, pair lists, Mq)
    u64 *d_pair_h[2] = {nullptr, nullptr}, *d_q[2] = {nullptr, nullptr};
    u32 *d_nq[2] = {nullptr, nullptr}, *d_win[2] = {nullptr, nullptr};
    u32 *d_pair_r[3] = {nullptr, nullptr, nullptr}, *d_pair_q[2] = {nullptr, nullptr}, *d_poff_pass[3] = {nullptr, nullptr, nullptr};
    u64 *d_m = nullptr, *d_mint = nullptr, *d_mq[2] = {nullptr, nullptr};
    // static dense dictionary (skx_ref::static_dense; build_static_dense): M holds the reference's static dense rows only -- two copies,
    // one per buffer set --, and the scan that fills set b's is queued as soon as the pass that will use set b is known to come
    // (queue_static_scan: when its first batch is sketched), not behind that pass's dictionary
    bool static_dense = false;
    u64* d_ms[2] = {nullptr, nullptr};
    bool m_ready[2] = {false, false};  // a scan into d_ms[b] is queued (or done) and nobody has consumed it yet
    // policy "reuse_membership" (off by default): the static dense rows of M depend on the reference alone, so a stream may scan the
    // reference once per buffer set and keep them -- no pass streams the 8 x s x N bytes again.  NOT what SURVEY 8(d)'s roofline figure
    // assumes (one scan of the reference per scoring pass): bench.py reports it as a side leg, `value` keeps scanning every pass.
    bool reuse_m = false;
    bool m_filled[2] = {false, false};
    u32* d_mdirty_s = nullptr;         // (a word for the early scan's "M was written" flag: nobody reads it)
    u32 sd64 = 0;                      // static dense rows, rounded up to 64: the rare rows of a pass start there
    u32 qhash_cap() const { return qcap - sd64; }  // distinct query hashes a pass's matrices hold beside the static rows
    // RANKING LANES (round 4).  The ranking of a batch is a chain of nine short, latency-bound kernels (counts, three prefix /
    // leader kernels, per-read arg-max, merge): alone it takes ~0.37 ms and leaves most of the chip idle, and the chains of
    // consecutive batches depend on each other through ONE thing only -- the table a batch starts from, which the previous
    // batch's chunk_prefix writes as its second kernel.  So consecutive batches take turns over kRankLanes lanes, each with its
    // own HIP stream and its own scratch; a chain waits for the event behind the previous chain's chunk_prefix (ev_cum) and
    // otherwise runs beside it.  The tables rotate over kRankLanes + 1 buffers: batch i reads T[i], writes T[i + 1]; T[i] is
    // written again by batch i + kRankLanes + 1, which follows batch i + 1's ... on a lane batch i's chain has long left.
    // Lane 0's stream is hs2, the stream every other user of the table is ordered on: a pass ends by joining the other lanes
    // into it.  What it buys: the tail of a stream (the last group's rankings used to run one after the other on an empty
    // chip) and the backlog the ranking stream built up beside the sketches.
    struct RankLane {
        hipStream_t s = nullptr;
        u32 *d_inc = nullptr;       // [segments of a pass][n_pad] per-segment increments (seg_sum)
        u32 *d_csum_raw = nullptr;  // [chunks][n_pad] chunk sums as counted, d_csum: their exclusive prefix (chunk_prefix)
        u32 *d_csum = nullptr;
        u32 *d_rel = nullptr;       // [segments of a pass][n_pad] segment start values relative to the pass-start table
        u32 *d_leader = nullptr;    // [chunks of 16 segments][species][top_k] genomes ranked first as the chunk begins
        u64 *d_lead_val = nullptr;  // [chunks][species] value of the top_k-th of them
        u64 *d_gmax = nullptr;      // [chunks + 1][half rank groups] best value inside 256 genomes at every chunk boundary
        u64 *d_lpart_sum = nullptr; // per-slice leader candidates (chunk_leader_part_kernel)
        u32 *d_lpart_idx = nullptr;
        u32 *d_live_ctr = nullptr;  // [2] sampled (chunk, half group)s that can hold a candidate / tested (seg_prefix_kernel)
        u64 *d_lead_seg = nullptr;  // [segments of a pass][species] the ranking's bound as every segment begins
        unsigned char *d_has = nullptr;   // [segments of a pass][rank groups] the pruned ranking kernels reported something
        unsigned char *d_live = nullptr;  // [segments of a pass][n_pad / 64] top-1 ranking: the word can hold a candidate
        u64 *d_cand_sum = nullptr;
        u32 *d_cand_idx = nullptr;
        u64 *d_cum_sink = nullptr;  // [n_pad] the table the chain's prefix kernels write (round 5: the real one comes from the pass's gains)
        hipEvent_t ev_cum = nullptr;   // its chain's chunk_prefix is done: the table the NEXT batch starts from is complete
        hipEvent_t ev_done = nullptr;  // its chain is done
        bool ready = false;
    };
    static const int kRankLanes = kRankLanesMax;
    RankLane lane[kRankLanes];
    int n_lanes = 1;             // lanes in use (the second one is allocated for streams that enqueue)
    bool own_lane_stream = false;  // the streams of lanes 1 .. are this skx_stream's own (not the device's shared set's)
    u32 n_cand_units = 0;        // candidate slots per read of the ranking's per-(read, unit) arrays
    u64 rank_seq = 0;            // batches ranked so far: batch i takes lane i % n_lanes
    RankLane* cum_writer = nullptr;  // the lane whose chunk_prefix wrote (or will have written) d_cum
    u64* d_tab[kRankLanes + 1] = {};  // the rotating tables
    int tab_cur = 0;             // d_cum == d_tab[tab_cur]
    u64* d_cum = nullptr;        // running table (current)
    u32* d_topk_idx = nullptr;
    u64* d_topk_sum = nullptr;
    u64* d_tab_tmp = nullptr;   // [n_genomes] staging of skx_stream_table / skx_stream_table_add
    u32* d_rank_idx = nullptr;  // [n_species][top] outputs of skx_stream_rank
    u64* d_rank_sum = nullptr;
    u32 rank_cap = 0;
    u32* h_poff = nullptr;   // pinned
    u64* h_offsets = nullptr;  // pinned
    // |Q| / pairs of the most recent pass whose dictionary has finished: the host only knows the pair count of a
    // pass when it picks the scan variant; reads of one sample share most of their matching hashes, so |Q| can be
    // far below it.  A hint only -- every variant gives the same bits.
    // dictionary builder scratch (launch_dictionary): hash set, per-key bucket offsets, bucket counts / bases, counters
    u64* d_ht[2] = {nullptr, nullptr};
    u32 ht_slots = 0;
    u32 *d_slot_off = nullptr, *d_bcount = nullptr, *d_bbase = nullptr, *d_btot = nullptr, *d_dict_ctr[2] = {nullptr, nullptr};
    u32* d_chk = nullptr;    // [8] device-side look at a batch's offsets (batch_check_kernel)
    u32* h_chk = nullptr;    // page-locked, coherent [16] per side: written by publish_kernel: d_chk, [8] = total pairs, [15] = sequence
    u32* h_chk_base = nullptr;
    u32 pub_seq = 0;         // sequence number of the latest publish
    bool chk_dirty[kSides] = {};  // per set: a push failed between arming and publishing -- re-zero the set's device-side counters before its next batch
    u64* d_rowany[2] = {nullptr, nullptr};   // [rank groups][qcap / 64] per buffer set: which query rows hold a bit for the group
    u32* d_mqext = nullptr;                  // per buffer set {row stride, clean rows} of (d_mq, d_rowany): rare_to_mq_kernel writes only what changed
    u32 mq_stride[2] = {0, 0};               // the row stride the set's last pass used (kept while it fits: the same stride = sparse writes)
    u32* d_grp_any[2] = {nullptr, nullptr};  // [rank groups + 1] per buffer set: the group's slice of the bit matrix holds any
                                             // bit; last word = "M itself was written this pass" (m_dirty)
    u64* d_hbuf = nullptr;                   // slabs of the lean scan kernel, [bands * tiles][scan_lean_words()][256]
    u32* d_wb[2] = {nullptr, nullptr};       // [query words][tiles] bands that can reach a word (launch_word_bands)
    u32* d_retry = nullptr;  // [1 + max_reads] reads the fast sketch variant hands to the full-size one ([0] = count)
    u32* d_big = nullptr;    // [1 + max_reads] reads the wave sketchers hand to the block sketcher ([0] = count)
    skx::LongReads lr{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0u};  // long reads of a production batch, split over waves
    u32* d_bsum = nullptr;   // block totals of the pair-count scan
    u64 reads_big = 0;       // reads that went through the block sketcher so far (statistic)
    u64 reads_split = 0, segs_split = 0;  // long reads split over waves so far, and their segments (statistic)
    u64 pool_grown = 0;                    // batches repeated with a larger row pool (statistic)
    u64 shared_passes = 0;                 // passes that served several enqueued batches (statistic)
    u64 groups_unshared = 0;               // groups of enqueued batches that did not fit one pass together and went one by one (statistic)
    u64 last_pairs = 0, last_passes = 0, total_passes = 0, lean_passes = 0;  // statistics (skx_stream_stats)
    // host-fed pipeline (skx_stream_submit): staging slots, a copy stream, one batch of lag
    struct Staged {
        bool pending = false;   // copied (or being copied) to the device, not yet processed
        bool in_flight = false; // processed, rows possibly still on their way to the host
        bool dropped = false;   // the batch was never scored: a batch enqueued before it failed (skx_stream_wait says so)
        u32 n_reads = 0;
        u64 n_bases = 0, ticket = 0;
        u32* out_idx = nullptr;
        u64* out_sum = nullptr;
        u32* d_rows_idx = nullptr;  // the slot's own device rows (batches may share a pass: each needs its rows until they are copied out)
        u64* d_rows_sum = nullptr;
        uint8_t* d_bases = nullptr;
        u64 *d_offsets = nullptr, *h_offsets = nullptr;
        hipEvent_t ev_copy = nullptr, ev_done = nullptr;
    } slot[kSlotsMax];  // (three: the copy of batch i starts once batch i - 3 is through, not i - 2; 2 n + 1 when groups of n batches
                        // share a pass -- the pass of the group that holds batch i - 2 n - 1 was queued n submits ago, not one)
    u32 n_slots = 3;
    hipStream_t hs_copy = nullptr;
    u64 next_ticket = 0;
    u32* h_nq = nullptr;     // pinned [4]
    u32 hint_pairs[2] = {0, 0};
    double nq_per_pair = 1.0;
    bool have_hint = false;  // false until one dictionary size has been seen
    // profiling
    int profiling = 0;  // 0 off, 1 every stage, 2 only the reference scan (the roofline kernel)
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> ev_pool;
    double ms[SKX_N_STAGES] = {0, 0, 0, 0, 0};
    u64 launches[SKX_N_STAGES] = {0, 0, 0, 0, 0};
};

static void free_side(skx_stream* st, int i);
static void free_lane(skx_stream* st, int i);
static void stream_free(skx_stream* st) {
    if (!st) return;
    (void)hipSetDevice(st->device);
    if (st->hs0) (void)hipStreamSynchronize(st->hs0);
    if (st->hs1) (void)hipStreamSynchronize(st->hs1);
    if (st->hs) (void)hipStreamSynchronize(st->hs);
    if (st->hs2) (void)hipStreamSynchronize(st->hs2);
    for (int i = 1; i < skx_stream::kRankLanes; ++i)
        if (st->lane[i].s) (void)hipStreamSynchronize(st->lane[i].s);
    for (int i = 0; i < kSides; ++i) (void)hipFree(st->sd_rows[i]);
    void* ptrs[] = {st->d_bases, st->d_offsets, st->d_pair_h[0], st->d_pair_h[1],
                    st->d_q[0], st->d_q[1], st->d_pair_r[0], st->d_pair_r[1], st->d_pair_q[0], st->d_pair_q[1],
                    st->d_poff_pass[0], st->d_poff_pass[1], st->d_poff_pass[2], st->d_pair_r[2], st->d_nq[0], st->d_nq[1], st->d_win[0], st->d_win[1], st->d_m, st->d_mint, st->d_mq[0], st->d_mq[1],
                    st->d_topk_idx, st->d_topk_sum, st->d_tab_tmp, st->d_rank_idx, st->d_rank_sum, st->d_bsum, st->d_grp_any[0],
                    st->d_grp_any[1], st->d_hbuf, st->d_wb[0], st->d_wb[1], st->d_rowany[0], st->d_rowany[1], st->d_mqext,
                    st->d_qd, st->d_qrow, st->d_sslot, st->d_qinfo, st->d_qloc, st->d_cls_bsum, st->d_nd_hs,
                    st->d_rowcnt, st->d_gain, st->d_gain_s, st->d_candslot, st->d_candmask, st->d_lrow, st->d_nlrow, st->d_gain_l, st->d_cbase, st->d_cwl, st->d_ncwl, st->d_cw, st->d_cbad, st->d_nqc, st->d_spc_g0, st->d_spc_grp, st->d_inb, st->d_hit, st->d_hist, st->d_nprow, st->d_pcw, st->d_ms[0], st->d_ms[1], st->d_mdirty_s};
    for (auto& q : st->ps) {
        for (void* x : {(void*)q.tab, (void*)q.cand, (void*)q.tabc, (void*)q.ncand, (void*)q.mode, (void*)q.any_full, (void*)q.nqc_total, (void*)q.mc, (void*)q.mqc,
                        (void*)q.rowany_c, (void*)q.grp_any_c, (void*)q.smap, (void*)q.pair_qc, (void*)q.nd}) (void)hipFree(x);
        if (q.h_pub) (void)hipHostFree(q.h_pub);
    }
    for (auto& t : st->d_tab) (void)hipFree(t);
    for (int i = 0; i < skx_stream::kRankLanes; ++i) free_lane(st, i);
    for (int i = 1; i < skx_stream::kRankLanes; ++i)
        if (st->own_lane_stream && st->lane[i].s) (void)hipStreamDestroy(st->lane[i].s);
    for (void* p : ptrs) (void)hipFree(p);
    if (st->h_poff) (void)hipHostFree(st->h_poff);
    if (st->h_offsets) (void)hipHostFree(st->h_offsets);
    if (st->h_nq) (void)hipHostFree(st->h_nq);
    if (st->h_nd) (void)hipHostFree(st->h_nd);
    if (st->h_nq_sink) (void)hipHostFree(st->h_nq_sink);
    if (st->h_lcount) (void)hipHostFree(st->h_lcount);
    if (st->h_chk_base) (void)hipHostFree(st->h_chk_base);
    for (int i = 0; i < kSides; ++i) {
        free_side(st, i);
        if (st->ev_main[i]) (void)hipEventDestroy(st->ev_main[i]);
    }
    if (!st->shared_queues && st->hs1 && st->hs1 != st->hs0) (void)hipStreamDestroy(st->hs1);
    (void)hipFree(st->d_ht[0]); (void)hipFree(st->d_ht[1]); (void)hipFree(st->d_dict_ctr[0]); (void)hipFree(st->d_dict_ctr[1]);
    (void)hipFree(st->d_slot_off); (void)hipFree(st->d_bcount); (void)hipFree(st->d_bbase);
    (void)hipFree(st->d_btot);
    for (auto& sl : st->slot) {
        (void)hipFree(sl.d_bases); (void)hipFree(sl.d_offsets); (void)hipFree(sl.d_rows_idx); (void)hipFree(sl.d_rows_sum);
        if (sl.h_offsets) (void)hipHostFree(sl.h_offsets);
        if (sl.ev_copy) (void)hipEventDestroy(sl.ev_copy);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    }
    if (st->hs_copy) (void)hipStreamDestroy(st->hs_copy);
    for (auto& sp : st->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto ev : st->ev_pool) (void)hipEventDestroy(ev);
    for (int i = 0; i < 2; ++i) {
        if (st->ev_dict[i]) (void)hipEventDestroy(st->ev_dict[i]);
        if (st->ev_pairq[i]) (void)hipEventDestroy(st->ev_pairq[i]);
        if (st->ev_front[i]) (void)hipEventDestroy(st->ev_front[i]);
        if (st->ev_back[i]) (void)hipEventDestroy(st->ev_back[i]);
    }
    for (int i = 0; i < kSides; ++i) {
        if (st->ev_sketch[i]) (void)hipEventDestroy(st->ev_sketch[i]);
        if (st->ev_skread[i]) (void)hipEventDestroy(st->ev_skread[i]);
    }
    for (auto ev : st->ev_pslot)
        if (ev) (void)hipEventDestroy(ev);
    if (st->shared_queues) {
        release_queues(st->device);
    } else {
        if (st->hs2 && st->hs2 != st->hs && st->hs2 != st->hs0) (void)hipStreamDestroy(st->hs2);
        if (st->hs0 && st->hs0 != st->hs) (void)hipStreamDestroy(st->hs0);
        if (st->hs) (void)hipStreamDestroy(st->hs);
    }
    delete st;
}

// one copy of the per-batch sketch buffers
static void free_side(skx_stream* st, int i) {
    void* ptrs[] = {st->sd_sk[i], st->sd_len[i], st->sd_cnt[i], st->sd_poff[i], st->sd_big[i], st->sd_retry[i], st->sd_chk[i],
                    st->sd_lr[i].list, st->sd_lr[i].seg0, st->sd_lr[i].seg_tab, st->sd_lr[i].seg_cnt, st->sd_lr[i].seg_h};
    for (void* q : ptrs) (void)hipFree(q);
    st->sd_sk[i] = nullptr; st->sd_len[i] = st->sd_cnt[i] = st->sd_poff[i] = st->sd_big[i] = st->sd_retry[i] = st->sd_chk[i] = nullptr;
    st->sd_lr[i] = skx::LongReads{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0u};
    st->pool_cap[i] = 0;
    st->side_ready[i] = false;
}
static hipError_t alloc_side_parts(skx_stream* st, int i);
// (all or nothing: a set that could not be completed is released again -- use_side only ever sees whole sets)
static hipError_t alloc_side(skx_stream* st, int i) {
    const hipError_t e = alloc_side_parts(st, i);
    if (e != hipSuccess) { free_side(st, i); (void)hipGetLastError(); }
    else st->side_ready[i] = true;
    return e;
}
static hipError_t alloc_side_parts(skx_stream* st, int i) {
    hipError_t e;
    // the pool: a fixed slot of 16 entries per read (C2 keeps 2.4 per read, C4 3.9: nearly every row fits its slot and costs no
    // atomic) and, behind it, a reservable part for the longer rows -- as large again, at least 2^20 entries, and never more than
    // the worst case needs (every read a full-width row; it is cut into 64 sub-pools, a workgroup of four reads uses sub-pool
    // blockIdx % 64), so small streams cannot overflow it at all; a batch that does is repeated with a larger pool (batch_back)
    {
        st->pool_fixed = (u32)std::min<u64>((u64)st->max_reads * skx::pool_row_fixed(), 0x7FFFFFF0ull);
        const u64 groups = ((u64)st->max_reads + 3) / 4, per_part = ((groups + 63) / 64) * 4 + 8;
        const u64 worst = 64 * per_part * st->sk_stride;
        const u64 res = std::min<u64>(worst, std::max<u64>((u64)st->max_reads * 16, 1u << 20));
        st->pool_cap[i] = std::min<u64>((u64)st->pool_fixed + std::max<u64>(res, 4096), 0xFFFFFF00ull);
    }
    if ((e = hipMalloc(&st->sd_sk[i], (size_t)st->pool_cap[i] * 8)) != hipSuccess) return e;
    if ((e = hipMalloc(&st->sd_len[i], ((size_t)st->max_reads + 1) * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(&st->sd_cnt[i], ((size_t)st->max_reads + 1) * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(&st->sd_poff[i], ((size_t)st->max_reads + 2) * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(&st->sd_big[i], ((size_t)st->max_reads + 1) * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(&st->sd_retry[i], ((size_t)st->max_reads + 1) * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(&st->sd_chk[i], (size_t)skx::chk_words() * 4)) != hipSuccess) return e;
    if ((e = hipMemset(st->sd_chk[i], 0, (size_t)skx::chk_words() * 4)) != hipSuccess) return e;
    if ((e = hipMemset(st->sd_retry[i], 0, 4)) != hipSuccess) return e;
    if (st->max_bases > skx::long_read_split()) {  // (a batch that can hold a long read at all)
        skx::LongReads& lr = st->sd_lr[i];
        lr.long_cap = (u32)std::min<u64>(st->max_reads, st->max_bases / skx::long_read_split() + 1);
        lr.segs_cap = (u32)std::min<u64>(0x7FFFFFFFu, 5 * (st->max_bases / (4ull * skx::kSketchCap)) + 2);
        if ((e = hipMalloc(&lr.list, (size_t)lr.long_cap * 4)) != hipSuccess) return e;
        if ((e = hipMalloc(&lr.seg0, (size_t)lr.long_cap * 4)) != hipSuccess) return e;
        if ((e = hipMalloc(&lr.seg_tab, (size_t)lr.segs_cap * 4)) != hipSuccess) return e;
        if ((e = hipMalloc(&lr.seg_cnt, (size_t)lr.segs_cap * 4)) != hipSuccess) return e;
        if ((e = hipMalloc(&lr.seg_h, (size_t)lr.segs_cap * skx::long_read_seg_slots() * 8)) != hipSuccess) return e;
    }
    return hipMemset(st->sd_big[i], 0, 4);  // (null stream; callers on the pipeline streams synchronise the device once)
}
static hipError_t use_side(skx_stream* st, int i) {
    if (!st->side_ready[i]) {
        // (not on the steady path: skx_stream_create allocates every set an enqueueing stream rotates over; this serves the
        // internal streams of skx_common_hashes, which start with one)
        hipError_t e = alloc_side(st, i);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) return e;
    }
    st->side = i;
    st->d_sk = st->sd_sk[i]; st->cur_stride = 0; st->cur_pool_cap = (u32)st->pool_cap[i]; st->cur_pool_fixed = st->pool_fixed;  // (pool mode; use_rows switches)
    st->d_len = st->sd_len[i]; st->d_cnt = st->sd_cnt[i]; st->d_poff = st->sd_poff[i];
    st->d_big = st->sd_big[i]; st->d_retry = st->sd_retry[i]; st->d_chk = st->sd_chk[i];
    st->lr = st->sd_lr[i];
    st->h_chk = st->h_chk_base + 16 * i;
    return hipSuccess;
}

// full-width rows for the side in use (debug outputs, unfiltered sketches, skx_common_hashes): allocated by the first call
// that needs them -- max_reads x min(s, longest read) x 8 bytes, 7.9 GB at C2, which a production stream never touches
static hipError_t use_rows(skx_stream* st) {
    const int i = st->side;
    if (!st->sd_rows[i]) {
        hipError_t e = hipMalloc(&st->sd_rows[i], (size_t)st->max_reads * st->sk_stride * 8);
        if (e != hipSuccess) return e;
    }
    st->d_sk = st->sd_rows[i]; st->cur_stride = st->sk_stride; st->cur_pool_cap = 0; st->cur_pool_fixed = 0;
    return hipSuccess;
}

// the scratch of one ranking lane (sizes: the stream's pass geometry, set before)
static hipError_t alloc_lane_parts(skx_stream* st, int i) {
    skx_stream::RankLane& L = st->lane[i];
    const skx_ref* ref = st->ref;
    // (sized for the larger of the two problems a chain may run on: every genome, or kCandCap candidates per species)
    const u32 n_sp = ref->n_species, top_k = st->top_k, n_pad = std::max<u32>(ref->n_pad, n_sp * skx::kCandCap);
    const u32 n_seg_max = (st->rpass + skx::kSegLen - 1) / skx::kSegLen, n_chunk_max = (n_seg_max + 15) / 16;
    const u32 k1 = std::max<u32>(top_k, 1);
    hipError_t e;
#define LCHK(expr) do { if ((e = (expr)) != hipSuccess) return e; } while (0)
    LCHK(hipMalloc(&L.d_inc, (size_t)n_seg_max * n_pad * 4));
    LCHK(hipMalloc(&L.d_rel, (size_t)n_seg_max * n_pad * 4));
    LCHK(hipMalloc(&L.d_live, (size_t)n_seg_max * (n_pad / 64)));
    LCHK(hipMalloc(&L.d_has, (size_t)n_seg_max * (n_pad / (skx::kRankWords * 64))));
    LCHK(hipMalloc(&L.d_lead_seg, (size_t)n_seg_max * n_sp * 8 + 64));
    LCHK(hipMalloc(&L.d_live_ctr, 8));
    LCHK(hipMemset(L.d_live_ctr, 0, 8));  // (from here on every ranking leaves it zero: store_host_words_kernel reads and resets)
    LCHK(hipMalloc(&L.d_csum, (size_t)n_chunk_max * n_pad * 4));
    LCHK(hipMalloc(&L.d_csum_raw, (size_t)n_chunk_max * n_pad * 4));
    LCHK(hipMalloc(&L.d_leader, (size_t)n_chunk_max * n_sp * k1 * 4 + 64));
    LCHK(hipMalloc(&L.d_lead_val, (size_t)n_chunk_max * n_sp * 8 + 64));
    LCHK(hipMalloc(&L.d_lpart_sum, (size_t)n_chunk_max * n_sp * skx::rank_leader_parts() * k1 * 8 + 64));
    LCHK(hipMalloc(&L.d_lpart_idx, (size_t)n_chunk_max * n_sp * skx::rank_leader_parts() * k1 * 4 + 64));
    LCHK(hipMalloc(&L.d_gmax, (size_t)(n_chunk_max + 1) * (n_pad / 256) * 8 + 64));
    if (top_k) {
        LCHK(hipMalloc(&L.d_cand_sum, (size_t)st->rpass * st->n_cand_units * top_k * 8));
        LCHK(hipMalloc(&L.d_cand_idx, (size_t)st->rpass * st->n_cand_units * top_k * 4));
    }
    LCHK(hipMalloc(&L.d_cum_sink, (size_t)n_pad * 8));
    LCHK(hipEventCreateWithFlags(&L.ev_cum, hipEventDisableTiming));
    LCHK(hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming));
#undef LCHK
    return hipSuccess;
}
static void free_lane(skx_stream* st, int i) {
    skx_stream::RankLane& L = st->lane[i];
    void* ptrs[] = {L.d_inc, L.d_rel, L.d_live, L.d_has, L.d_lead_seg, L.d_live_ctr, L.d_csum, L.d_csum_raw, L.d_leader, L.d_lead_val,
                    L.d_lpart_sum, L.d_lpart_idx, L.d_gmax, L.d_cand_sum, L.d_cand_idx, L.d_cum_sink};
    for (void* q : ptrs) (void)hipFree(q);
    if (L.ev_cum) (void)hipEventDestroy(L.ev_cum);
    if (L.ev_done) (void)hipEventDestroy(L.ev_done);
    const hipStream_t keep = L.s;  // (streams are created and destroyed with the skx_stream, not with the scratch)
    L = skx_stream::RankLane{};
    L.s = keep;
}
static hipError_t alloc_lane(skx_stream* st, int i) {
    const hipError_t e = alloc_lane_parts(st, i);
    if (e != hipSuccess) { free_lane(st, i); (void)hipGetLastError(); }
    else st->lane[i].ready = true;
    return e;
}

// pair_hint: pairs (read, hash some genome holds) a read is expected to contribute at most; sizes the pass workspace
// (a batch with more than it can hold is cut into several passes -- correct at any size)
// dense_queries: the pairs of a pass are (nearly) all distinct hashes (skx_common_hashes: whole sketches as queries) -- the
// matrices then get as many rows as the pass has pairs, bounded by a sixth of the free device memory
// enqueueing: the stream may be driven through skx_stream_enqueue_device / skx_stream_submit -- every buffer set its batches rotate
// over (stream_coalesce + 1) is allocated HERE: a hipMalloc and the device-wide synchronisation behind it have no place in
// the enqueue path (round 3 allocated them at first use: the first stream_coalesce + 1 enqueues of a stream each stalled
// every stream of the device -- the first repetition of the bench ran at 79 M reads/s, the others at 121 M)
static int stream_create_internal(skx_stream** out, const skx_ref* ref, u32 top_k, u32 max_reads, u64 max_bases,
                                  u32 sk_stride, u32 pair_hint, bool dense_queries = false, bool enqueueing = false) {
    SKXCHK(use_device(ref->device));
    skx_stream* st = new skx_stream;
    st->ref = ref; st->device = ref->device; st->top_k = top_k; st->max_reads = max_reads; st->max_bases = max_bases;
    st->sk_stride = sk_stride;
    const u32 n_pad = ref->n_pad, n_gw = n_pad / 64;
    // Pass capacity.  Pairs (read, hash some genome holds): 16 per read on average (C2: 2.4), at most 2^22 -- pair lists and
    // the hash set cost ~70 bytes per pair.  Rows of the bit matrices = DISTINCT query hashes of a pass (C2: 10 k of 232 k
    // pairs: the reads of a sample share their matching hashes): policy "stream_query_rows", default 65 536 -- the three
    // matrices of a pass (M, 2 x Mq) take rows x genomes / 8 bytes each: 1 GB in all at C2.  (Rounds 1-2 sized the matrices
    // by the PAIR capacity: 51 GB at C2.)  The host learns a batch's |Q| with its published summary (the speculative pair
    // gather counts its distinct keys); a batch with more pairs or more distinct hashes than a pass holds is cut into
    // several passes, each bounded by pairs <= min(pcap, qcap) -- correct at any size.
    static const u64 pass_pairs_env = skx::knob("SKX_PASS_PAIRS") ? (u64)atoll(skx::knob("SKX_PASS_PAIRS")) : 0;  // test knob
    u64 pc = std::min<u64>((u64)max_reads * std::min<u32>(sk_stride, pair_hint), 1u << 22);  // (a read contributes at most sk_stride pairs)
    if (pass_pairs_env) pc = pass_pairs_env;
    pc = std::max<u64>(pc, sk_stride);
    pc = (pc + 63) / 64 * 64;
    st->pcap = (u32)pc;
    const u32 rows_policy = g_stream_query_rows.load();
    u64 qc = rows_policy ? rows_policy : 65536;
    if (!rows_policy && ref->static_dense) qc += (ref->n_sd + 63u) & ~63u;  // (the static dense rows come on top of the distinct hashes of a pass)
    if (dense_queries) {
        size_t mem_free = 0, mem_total = 0;
        (void)hipMemGetInfo(&mem_free, &mem_total);
        qc = std::max<u64>(qc, std::min<u64>(pc, (u64)(mem_free / 6) / (3ull * n_pad / 8)));
    }
    qc = std::max<u64>(std::min<u64>(qc, pc), std::min<u64>(pc, sk_stride));  // (never below one read's worth: a read alone must fit a pass)
    qc = (qc + 63) / 64 * 64;
    st->qcap = (u32)qc;
    st->qcap_auto = rows_policy == 0 && !dense_queries;
    st->qcap_max = st->qcap;
    if (st->qcap_auto) {
        size_t mem_free = 0, mem_total = 0;
        (void)hipMemGetInfo(&mem_free, &mem_total);
        const u64 per_row = (u64)n_pad / 8 + 2ull * ((n_gw + skx::kRankWords - 1) / skx::kRankWords * skx::kRankWords) * 8 + 3ull * skx::kPassBatchesMax * 4 + 8;
        st->qcap_max = (u32)std::max<u64>(st->qcap, std::min<u64>(st->pcap, (u64)(mem_free / 4) / per_row) / 64 * 64);
    }
    static const u32 coalesce_env = skx::knob("SKX_COALESCE") ? (u32)atoi(skx::knob("SKX_COALESCE")) : 0u;  // experiment knob
    st->coalesce = coalesce_env ? std::min<u32>((u32)kGroupMax, std::max(1u, coalesce_env)) : g_stream_coalesce.load();
    st->group_cap = st->coalesce;
    // reads per pass: the whole batch if the ranking's per-segment arrays fit (inc / rel: 4 bytes per (64 reads, genome) each, at most
    // an eighth of the free device memory) -- a batch cut into two passes scans the reference twice
    static const u64 pass_reads_env = skx::knob("SKX_PASS_READS") ? (u64)atoll(skx::knob("SKX_PASS_READS")) : 0;  // test knob
    u64 pass_reads = 1u << 20, lanes_planned = 1;
    {
        size_t mem_free = 0, mem_total = 0;
        (void)hipMemGetInfo(&mem_free, &mem_total);
        // (every ranking lane has its own inc / rel / candidate arrays: the eighth of the free memory is shared by the lanes this
        // stream will ask for -- round 4 budgeted one lane's worth and allocated up to four)
        static const int lanes_env0 = skx::knob("SKX_RANK_LANES") ? atoi(skx::knob("SKX_RANK_LANES")) : 0;
        const int lanes_want = lanes_env0 ? lanes_env0 : (int)g_rank_lanes;
        lanes_planned = (enqueueing && top_k) ? (u64)std::max(1, std::min(lanes_want, (int)skx_stream::kRankLanes)) : 1;
        const u64 per_seg = ((u64)n_pad * 4 * 2 + n_pad / 64 + 64) * lanes_planned;
        pass_reads = std::min<u64>(pass_reads, std::max<u64>(4096, (u64)(mem_free / 8) / per_seg * skx::kSegLen));
    }
    if (pass_reads_env) pass_reads = pass_reads_env;
    u64 rp = std::min<u64>(max_reads, pass_reads);
    // candidate arrays of the ranking: per (read, rank group, row) for top_k <= 16, per (read, genome word, row) beyond
    const u32 n_sp = ref->n_species;
    const u32 n_gw_max = std::max<u32>(n_gw, n_sp * skx::kCandCap / 64);  // (... or of the candidates' compact problem, if that is larger)
    const u32 n_cand_units = (top_k >= 1 && top_k <= skx::rank_topk_fast_max()) ? (n_gw_max + skx::kRankWords - 1) / skx::kRankWords : n_gw_max;
    if (top_k) rp = std::min<u64>(rp, std::max<u64>(skx::kSegLen, (4ull << 30) / lanes_planned / ((u64)n_cand_units * top_k * 12)));
    rp = std::max<u64>(rp, 1);
    st->rpass = (u32)rp;
    const u32 n_bt = ref->n_bands * ref->n_tiles;

    hipError_t e;
#define SCHK(expr) do { e = (expr); if (e != hipSuccess) { stream_free(st); return fail(SKX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e)); } } while (0)
    // pipeline depth (env SKX_PIPELINE, default 3): 1 = one stream (strictly serial), 2 = {sketch, dictionary, scan,
    // transpose} | {ranking}, 3 = {sketch, dictionary} | {scan, transpose} | {ranking}.  With round 1's scan kernel
    // (issue-bound itself) three streams gained nothing; with the lean scan kernel, its waves at raised priority and the
    // sketch kernel leaving room on every CU (launch_sketch, leave_room) the VALU-bound sketch of batch i+1 and the
    // HBM-bound scan of batch i overlap for real: C2, same box: 54.5 -> 59.7 M reads/s from a fresh table, 56.8 -> 63.6 M
    // steady, the scan itself 0.62 -> 0.78 ms.
    static const int depth_env = skx::knob("SKX_PIPELINE") ? atoi(skx::knob("SKX_PIPELINE")) : 3;
    st->depth = depth_env < 1 ? 1 : depth_env > 4 ? 4 : depth_env;
    // the HBM-bound scan stream gets the higher priority (it needs its full occupancy); the others fill what is left
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // numerically lower = higher priority
    static const int prio_env = skx::knob("SKX_PRIO") ? atoi(skx::knob("SKX_PRIO")) : 1;
    if (!prio_env) prio_hi = prio_lo;
    // Experiment (env SKX_CU_SCAN = n, with SKX_PIPELINE = 3): partition the chip -- the scan stream may only use n of the
    // CUs (spread evenly over the XCDs: mask bit i is CU i / 8 of XCD i % 8), the sketch / dictionary and ranking
    // streams only the others, so the HBM-bound scan of batch i and the VALU-bound sketch of batch i + 1 run side by
    // side without sharing a CU.  Results in DESIGN.md; off by default.
    static const int cu_scan_env = skx::knob("SKX_CU_SCAN") ? atoi(skx::knob("SKX_CU_SCAN")) : 0;
    int n_cus = 0;
    (void)hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, st->device);
    if (cu_scan_env > 0 && cu_scan_env < n_cus && st->depth >= 3) {
        const u32 words = (u32)(n_cus + 31) / 32;
        std::vector<uint32_t> m_scan(words, 0u), m_rest(words, 0u);
        for (int i = 0; i < n_cus; ++i) (i < cu_scan_env ? m_scan : m_rest)[i / 32] |= 1u << (i % 32);
        SCHK(hipExtStreamCreateWithCUMask(&st->hs, words, m_scan.data()));
        SCHK(hipExtStreamCreateWithCUMask(&st->hs0, words, m_rest.data()));
        SCHK(hipExtStreamCreateWithCUMask(&st->hs2, words, m_rest.data()));
    } else if (st->depth == 3 && prio_env && !skx::knob("SKX_PRIO_RANK")) {
        // the product's configuration: the device's shared set (created by its first stream)
        std::lock_guard<std::mutex> lk(g_queues_mu);
        SharedQueues& q = g_queues[st->device & 63];
        if (q.refs == 0) {
            hipError_t e = hipStreamCreateWithPriority(&q.hs, hipStreamNonBlocking, prio_hi);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&q.hs0, hipStreamNonBlocking, prio_lo);
            static const int rank_hi_q = skx::knob("SKX_PRIO_RANK") ? atoi(skx::knob("SKX_PRIO_RANK")) : 0;  // experiment knob: 1 = as the scan, 2 = between
            const int prio_rank_q = rank_hi_q == 1 ? prio_hi : rank_hi_q == 2 ? (prio_lo + prio_hi) / 2 : prio_lo;
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&q.hs2, hipStreamNonBlocking, prio_rank_q);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&q.hs1, hipStreamNonBlocking, prio_lo);
            if (e != hipSuccess) {
                for (hipStream_t* h : {&q.hs1, &q.hs2, &q.hs0, &q.hs}) { if (*h) (void)hipStreamDestroy(*h); *h = nullptr; }
                SCHK(e);
            }
        }
        q.refs += 1;
        st->shared_queues = true;
        st->hs = q.hs; st->hs0 = q.hs0; st->hs2 = q.hs2; st->hs1 = q.hs1;
    } else {
    SCHK(hipStreamCreateWithPriority(&st->hs, hipStreamNonBlocking, prio_hi));
    if (st->depth >= 3) SCHK(hipStreamCreateWithPriority(&st->hs0, hipStreamNonBlocking, prio_lo)); else st->hs0 = st->hs;
    // (4: the ranking shares the SKETCH stream -- {sketch, ranking} | {dictionary, scan, transpose}: the two VALU-bound stages
    // take turns and only the HBM-bound scan runs beside them)
    if (st->depth == 4) st->hs2 = st->hs0;
    else if (st->depth >= 2) {
        static const int rank_hi = skx::knob("SKX_PRIO_RANK") ? atoi(skx::knob("SKX_PRIO_RANK")) : 0;  // experiment knob
        SCHK(hipStreamCreateWithPriority(&st->hs2, hipStreamNonBlocking, rank_hi ? prio_hi : prio_lo));
    } else st->hs2 = st->hs;
    }
    for (int i = 0; i < 2; ++i) {
        SCHK(hipEventCreateWithFlags(&st->ev_dict[i], hipEventDisableTiming));
        SCHK(hipEventCreateWithFlags(&st->ev_pairq[i], hipEventDisableTiming));
        SCHK(hipEventCreateWithFlags(&st->ev_front[i], hipEventDisableTiming));
        SCHK(hipEventCreateWithFlags(&st->ev_back[i], hipEventDisableTiming));
    }
    for (int i = 0; i < kSides; ++i) {
        SCHK(hipEventCreateWithFlags(&st->ev_sketch[i], hipEventDisableTiming));
        SCHK(hipEventCreateWithFlags(&st->ev_skread[i], hipEventDisableTiming));
        SCHK(hipEventCreateWithFlags(&st->ev_main[i], hipEventDisableTiming));
    }
    if (!st->shared_queues) {
        if (st->depth >= 3) SCHK(hipStreamCreateWithPriority(&st->hs1, hipStreamNonBlocking, prio_lo)); else st->hs1 = st->hs0;
    }
    SCHK(hipMalloc(&st->d_bases, std::max<u64>(max_bases, 1)));
    SCHK(hipMalloc(&st->d_offsets, ((size_t)max_reads + 1) * 8));
    {
        const int n_now = enqueueing ? (st->coalesce >= 2 ? (int)st->coalesce + 1 : 2) : 1;
        for (int i = 0; i < n_now; ++i) SCHK(alloc_side(st, i));
    }
    for (int i = 0; i < 2; ++i) SCHK(hipMalloc(&st->d_pair_h[i], (size_t)st->pcap * 8));
    for (int i = 0; i < 2; ++i) SCHK(hipMalloc(&st->d_q[i], (size_t)st->pcap * 8));
    for (int i = 0; i < 2; ++i) SCHK(hipMalloc(&st->d_pair_q[i], (size_t)st->pcap * 4));
    for (int i = 0; i < 3; ++i) {
        SCHK(hipMalloc(&st->d_pair_r[i], (size_t)st->pcap * 4));
        SCHK(hipMalloc(&st->d_poff_pass[i], (size_t)st->coalesce * ((size_t)st->rpass + 2) * 4));  // (the batches that share a pass: one region each)
        SCHK(hipEventCreateWithFlags(&st->ev_pslot[i], hipEventDisableTiming));
    }
    for (int i = 0; i < 2; ++i) {
        SCHK(hipMalloc(&st->d_nq[i], 64));
        SCHK(hipMalloc(&st->d_win[i], (size_t)n_bt * 8));
    }
    // (rows: qcap + 128 -- the rows behind the dense ones start on a word boundary, a pass of qcap hashes can reach 63 rows further)
    {
        const u32 sd64 = (ref->n_sd + 63u) & ~63u;
        st->static_dense = ref->static_dense && (u64)sd64 + 1024 <= st->qcap;
        if (st->static_dense) {
            st->sd64 = sd64;
            for (int i = 0; i < 2; ++i) {
                SCHK(hipMalloc(&st->d_ms[i], (size_t)(sd64 / 64 + 2) * n_pad * 8));
                SCHK(hipMemset(st->d_ms[i], 0, (size_t)(sd64 / 64 + 2) * n_pad * 8));  // all-zero until a scan fills it; the pass that consumes it zeroes it again
            }
            SCHK(hipMalloc(&st->d_mdirty_s, 64));
            st->reuse_m = g_reuse_membership.load() != 0;
        } else {
            SCHK(hipMalloc(&st->d_m, (size_t)(st->qcap / 64 + 2) * n_pad * 8));
            SCHK(hipMemset(st->d_m, 0, (size_t)(st->qcap / 64 + 2) * n_pad * 8));      // kept all-zero between passes
        }
    }
    // (d_mint, the second word array of the split scan variant, is allocated by the first pass that wants it)
    for (int i = 0; i < 2; ++i) {
        const size_t bytes = ((size_t)st->qcap + 128) * ((n_gw + skx::kRankWords - 1) / skx::kRankWords * skx::kRankWords) * 8;
        SCHK(hipMalloc(&st->d_mq[i], bytes));
        SCHK(hipMemset(st->d_mq[i], 0, bytes));  // (zero + zero flags = clean under any row stride: d_mqext)
    }
    {
        const u32 any[4] = {skx::mq_any_stride(), 0xFFFFFFFFu, skx::mq_any_stride(), 0xFFFFFFFFu};
        SCHK(hipMalloc(&st->d_mqext, sizeof(any)));
        SCHK(hipMemcpy(st->d_mqext, any, sizeof(any), hipMemcpyHostToDevice));
    }
    st->n_cand_units = n_cand_units;
    {
        // ranking lanes: the second one only for streams that enqueue (batches back to back are what it overlaps); experiment
        // knob SKX_RANK_LANES=1 keeps every chain on one lane
        static const int lanes_env = skx::knob("SKX_RANK_LANES") ? atoi(skx::knob("SKX_RANK_LANES")) : 0;
        const int want = lanes_env ? lanes_env : (int)g_rank_lanes;
        st->n_lanes = (enqueueing && top_k && st->depth >= 3) ? std::max(1, std::min(want, (int)skx_stream::kRankLanes)) : 1;
        st->lane[0].s = st->hs2;
        for (int i = 1; i < st->n_lanes; ++i) {
            if (st->shared_queues) {
                // (the error is checked OUTSIDE the lock: SCHK frees the stream, which releases the shared queues under this mutex)
                hipError_t lane_err = hipSuccess;
                {
                    std::lock_guard<std::mutex> lk(g_queues_mu);
                    SharedQueues& q = g_queues[st->device & 63];
                    static const int rank_hi_l = skx::knob("SKX_PRIO_RANK") ? atoi(skx::knob("SKX_PRIO_RANK")) : 0;
                    const int prio_rank_l = rank_hi_l == 1 ? prio_hi : rank_hi_l == 2 ? (prio_lo + prio_hi) / 2 : prio_lo;
                    if (!q.lane_s[i - 1]) {
                        lane_err = hipStreamCreateWithPriority(&q.lane_s[i - 1], hipStreamNonBlocking, prio_rank_l);
                        if (lane_err != hipSuccess) q.lane_s[i - 1] = nullptr;
                    }
                    st->lane[i].s = q.lane_s[i - 1];
                }
                SCHK(lane_err);
            } else {
                SCHK(hipStreamCreateWithPriority(&st->lane[i].s, hipStreamNonBlocking, prio_lo));
                st->own_lane_stream = true;
            }
        }
        SCHK(alloc_lane(st, 0));
        for (int i = 1; i < st->n_lanes; ++i)
            if (alloc_lane(st, i) != hipSuccess) { st->n_lanes = i; break; }  // no room for another lane's scratch: fewer lanes, same results
    }
    for (int i = 0; i <= st->n_lanes; ++i) {
        SCHK(hipMalloc(&st->d_tab[i], (size_t)n_pad * 8));
        SCHK(hipMemset(st->d_tab[i], 0, (size_t)n_pad * 8));
    }
    st->tab_cur = 0;
    st->d_cum = st->d_tab[0];
    if (top_k) {
        SCHK(hipMalloc(&st->d_topk_idx, (size_t)max_reads * n_sp * top_k * 4));
        SCHK(hipMalloc(&st->d_topk_sum, (size_t)max_reads * n_sp * top_k * 8));
    }
    SCHK(hipMalloc(&st->d_tab_tmp, (size_t)ref->n_genomes * 8));
    st->rank_cap = n_sp * std::max<u32>(SKX_MAX_TOP, top_k);
    SCHK(hipMalloc(&st->d_rank_idx, (size_t)st->rank_cap * 4));
    SCHK(hipMalloc(&st->d_rank_sum, (size_t)st->rank_cap * 8));
    SCHK(hipMalloc(&st->d_bsum, ((size_t)max_reads / 1024 + 2) * 4));
    for (int i = 0; i < 2; ++i) SCHK(hipMalloc(&st->d_grp_any[i], (size_t)(n_pad / (skx::kRankWords * 64) + 1) * 4));
    for (int i = 0; i < 2; ++i) {
        SCHK(hipMalloc(&st->d_rowany[i], (size_t)(n_pad / (skx::kRankWords * 64)) * (st->qcap / 64 + 2) * 8));
        SCHK(hipMemset(st->d_rowany[i], 0, (size_t)(n_pad / (skx::kRankWords * 64)) * (st->qcap / 64 + 2) * 8));
    }
    // (the slabs of round 3's form of the lean kernel, 252 MB at C2, 0.94 GB at C4: experiments build, when a knob asks for that form)
    if (skx::scan_lean_wants_slabs() && skx::scan_lean_applies(ref->n_bands, false, false)) {
        SCHK(hipMalloc(&st->d_hbuf, (size_t)n_bt * skx::scan_lean_words() * skx::kTileGenomes * 8));
        for (int i = 0; i < 2; ++i) SCHK(hipMalloc(&st->d_wb[i], ((size_t)st->qcap / 64 + 1) * ref->n_tiles * 16));
    }
    SCHK(hipHostMalloc((void**)&st->h_poff, ((size_t)max_reads + 2) * 4, hipHostMallocDefault));
    SCHK(hipHostMalloc((void**)&st->h_offsets, ((size_t)max_reads + 1) * 8, hipHostMallocDefault));
    st->ht_slots = 1024;
    while (st->ht_slots < 2ull * st->pcap) st->ht_slots <<= 1;
    for (int i = 0; i < 2; ++i) {
        SCHK(hipMalloc(&st->d_ht[i], (size_t)st->ht_slots * 8));
        SCHK(hipMemset(st->d_ht[i], 0xFF, (size_t)st->ht_slots * 8));  // all-ones = empty; the compaction empties it again
        SCHK(hipMalloc(&st->d_dict_ctr[i], 64));
        SCHK(hipMemset(st->d_dict_ctr[i], 0, 64));
    }
    SCHK(hipMalloc(&st->d_slot_off, (size_t)st->ht_slots * 4));
    SCHK(hipMalloc(&st->d_btot, 256 * 4));
    SCHK(hipMalloc(&st->d_bcount, (size_t)skx::dict_buckets() * 4));
    SCHK(hipMemset(st->d_bcount, 0, (size_t)skx::dict_buckets() * 4));
    SCHK(hipMalloc(&st->d_bbase, (size_t)skx::dict_buckets() * 4));
    SCHK(hipHostMalloc((void**)&st->h_chk_base, kSides * 16 * 4, hipHostMallocCoherent));  // kernels write it, the host polls it
    memset(st->h_chk_base, 0, kSides * 16 * 4);
    SCHK(use_side(st, 0));
    SCHK(hipHostMalloc((void**)&st->h_nq, 4 * 4, hipHostMallocCoherent));  // ([2], [3]: the ranking's live sample, d_live_ctr)
    st->h_nq[0] = st->h_nq[1] = 0;
    st->h_nq[2] = 1; st->h_nq[3] = 1;  // (nothing known yet: everything may hold a candidate)
    SCHK(hipHostMalloc((void**)&st->h_nd, 4 * 4, hipHostMallocCoherent));
    memset(st->h_nd, 0, 4 * 4);
    SCHK(hipHostMalloc((void**)&st->h_nq_sink, 4 * 4, hipHostMallocCoherent));
    SCHK(hipHostMalloc((void**)&st->h_lcount, 4 * 4, hipHostMallocCoherent));
    memset(st->h_lcount, 0, 4 * 4);
    SCHK(hipMalloc(&st->d_nd_hs, 64));
    SCHK(hipMemset(st->d_nd_hs, 0, 64));
    if (ref->d_kt_key) {  // the split dictionary of a pass (rare-hash index)
        SCHK(hipMalloc(&st->d_qd, (size_t)st->pcap * 8));
        SCHK(hipMalloc(&st->d_qrow, (size_t)st->pcap * 4));
        SCHK(hipMalloc(&st->d_sslot, (size_t)st->pcap * 8));  // (start, length) of a rare row's genome list
        SCHK(hipMalloc(&st->d_qinfo, (size_t)st->pcap * 4));
        SCHK(hipMalloc(&st->d_qloc, (size_t)st->pcap * 4));
        SCHK(hipMalloc(&st->d_cls_bsum, ((size_t)st->pcap / 1024 + 2) * 4));
    }
    {
        // the table without the ranking + the candidates of a batch (skx_kernels.hip): hs-only scratch once, what the chains read per set
        const u32 nb = skx::kPassBatchesMax, cap = skx::kCandCap, rows_c = skx::kCandRows;
        st->n_pad_c = n_sp * cap;
        st->n_grp_c = st->n_pad_c / (skx::kRankWords * 64);
        SCHK(hipMalloc(&st->d_rowcnt, (size_t)nb * ((size_t)st->qcap + 128) * 4));
        SCHK(hipMalloc(&st->d_gain, (size_t)nb * n_pad * 4));
        if (ref->d_kt_key) SCHK(hipMalloc(&st->d_gain_s, (size_t)nb * n_pad * 4 * skx::gain_sparse_stride()));
        SCHK(hipMalloc(&st->d_candslot, (size_t)nb * n_pad * 4));
        SCHK(hipMalloc(&st->d_candmask, (size_t)n_pad * 4));
        if (ref->d_mlong) {
            SCHK(hipMalloc(&st->d_lrow, (size_t)nb * ((size_t)st->qcap + 128) * 8));
            SCHK(hipMalloc(&st->d_nlrow, skx::pass_counter_bytes()));
            SCHK(hipMalloc(&st->d_gain_l, (size_t)nb * n_pad * 4));
            SCHK(hipMalloc(&st->d_cw, (size_t)nb * (n_pad / 64) * 8));
            SCHK(hipMalloc(&st->d_cbase, (size_t)nb * (n_pad / 64) * 4));
            SCHK(hipMalloc(&st->d_cwl, (size_t)nb * st->n_pad_c * 4));
            SCHK(hipMalloc(&st->d_ncwl, 64));
            if (ref->d_mlongT) {
                SCHK(hipMalloc(&st->d_inb, (size_t)nb * ref->n_lw * 8));
                SCHK(hipMalloc(&st->d_hit, (size_t)nb * ref->n_lw * 8));
            }
            if (ref->d_prec && ref->n_pat && st->n_grp_c * skx::kRankWords <= skx::pat_words_max()) {
                st->use_pat = true;
                st->npat_pad = (ref->n_pat + 63u) & ~63u;
                SCHK(hipMalloc(&st->d_hist, (size_t)nb * st->npat_pad * 4));
                SCHK(hipMalloc(&st->d_nprow, skx::pass_counter_bytes()));
                SCHK(hipMalloc(&st->d_pcw, (size_t)nb * ref->n_pat * st->n_grp_c * skx::kRankWords * 8));
            }
        }
        SCHK(hipMalloc(&st->d_cbad, 64));
        SCHK(hipMalloc(&st->d_nqc, skx::pass_counter_bytes()));
        std::vector<u32> g0c(n_sp), grpc(st->n_grp_c);
        for (u32 i = 0; i < n_sp; ++i) g0c[i] = i * cap;
        for (u32 i = 0; i < st->n_grp_c; ++i) grpc[i] = i / (cap / (skx::kRankWords * 64));
        SCHK(hipMalloc(&st->d_spc_g0, (size_t)n_sp * 4));
        SCHK(hipMalloc(&st->d_spc_grp, (size_t)st->n_grp_c * 4));
        SCHK(hipMemcpy(st->d_spc_g0, g0c.data(), (size_t)n_sp * 4, hipMemcpyHostToDevice));
        SCHK(hipMemcpy(st->d_spc_grp, grpc.data(), (size_t)st->n_grp_c * 4, hipMemcpyHostToDevice));
        for (auto& q : st->ps) {
            SCHK(hipMalloc(&q.tab, (size_t)(nb + 1) * n_pad * 8));
            SCHK(hipMalloc(&q.cand, (size_t)nb * st->n_pad_c * 4));
            SCHK(hipMalloc(&q.tabc, (size_t)nb * st->n_pad_c * 8));
            SCHK(hipMalloc(&q.ncand, (size_t)nb * n_sp * 4));
            SCHK(hipMalloc(&q.mode, 64)); SCHK(hipMalloc(&q.any_full, 64)); SCHK(hipMalloc(&q.nqc_total, 64));
            SCHK(hipMemset(q.any_full, 0, 64));
            SCHK(hipMalloc(&q.mc, (size_t)nb * (rows_c / 64) * st->n_pad_c * 8));
            SCHK(hipMalloc(&q.mqc, (size_t)nb * st->n_grp_c * rows_c * skx::kRankWords * 8));
            SCHK(hipMalloc(&q.rowany_c, (size_t)nb * st->n_grp_c * (rows_c / 64) * 8));
            SCHK(hipMalloc(&q.grp_any_c, (size_t)nb * st->n_grp_c * 4 + 64));
            SCHK(hipMalloc(&q.smap, (size_t)nb * ((size_t)st->qcap + 128) * 4));
            SCHK(hipMalloc(&q.pair_qc, (size_t)st->pcap * 4));
            SCHK(hipMalloc(&q.nd, 64));
            SCHK(hipMemset(q.nd, 0, 64));
            SCHK(hipHostMalloc((void**)&q.h_pub, 32 * 4, hipHostMallocCoherent));
            memset(q.h_pub, 0, 32 * 4);
        }
    }
    // the zero-fills above ran on the null stream, which the (non-blocking) pipeline streams do not wait for
    SCHK(hipDeviceSynchronize());
#undef SCHK
    *out = st;
    return SKX_OK;
}

SKX_API int skx_stream_create(skx_stream** out, const skx_ref* ref, uint32_t top_k, uint32_t max_batch_reads,
                              uint64_t max_batch_bases) {
    if (!out || !ref) return fail(SKX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    // the reference slices result_vec[..top] and panics when top > N (src/sketchy.rs:391)
    if (top_k > ref->min_species) return fail(SKX_ERR_INVALID, "top_k=%u exceeds n_genomes=%u", top_k, ref->min_species);
    if (top_k > SKX_MAX_TOP) return fail(SKX_ERR_INVALID, "top_k=%u exceeds SKX_MAX_TOP=%u", top_k, SKX_MAX_TOP);
    if (max_batch_reads < 1) return fail(SKX_ERR_INVALID, "max_batch_reads must be >= 1");
    // a read contributes at most min(s, #k-mers) hashes; short reads (<= kSketchCap k-mers) are the floor
    const u64 longest = std::max<u64>(max_batch_bases, (u64)skx::kSketchCap);
    const u32 sk_stride = (u32)std::min<u64>(ref->s_read, longest);
    if (max_batch_bases >= (1ull << 32)) return fail(SKX_ERR_CAPACITY, "max_batch_bases must be below 2^32");
    // real reads leave a handful of pairs each (C2: ~5 of ~1500 hashes are in range and held by some genome); 64 per read
    // keeps small streams small, and denser batches simply take more passes
    return stream_create_internal(out, ref, top_k, max_batch_reads, max_batch_bases, sk_stride, 64, false, true);
}
SKX_API void skx_stream_destroy(skx_stream* st) { stream_free(st); }

// ---- profiling spans
static int flush_pending(skx_stream* st);
SKX_API int skx_stream_sync(skx_stream* st);
static hipEvent_t get_event(skx_stream* st) {
    if (!st->ev_pool.empty()) { hipEvent_t ev = st->ev_pool.back(); st->ev_pool.pop_back(); return ev; }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}
struct Span {
    skx_stream* st; int stage; hipStream_t on; hipEvent_t a = nullptr; bool active = false;
    Span(skx_stream* s, int stg, hipStream_t on_ = nullptr) : st(s), stage(stg), on(on_ ? on_ : s->hs0) {
        active = st->profiling == 1 || (st->profiling == 2 && stage == 2);
        if (active) { a = get_event(st); (void)hipEventRecord(a, on); }
    }
    ~Span() {
        if (active) { hipEvent_t b = get_event(st); (void)hipEventRecord(b, on); st->spans.push_back({stage, a, b}); }
    }
};
static void collect_spans(skx_stream* st) {
#ifdef SKX_EXPERIMENTS
    // SKX_SPAN_DUMP=1: every recorded span with its start and end relative to the first one (the undistorted device timeline of
    // the stages: a kernel trace slows the queueing thread enough to change the picture)
    if (skx::knob("SKX_SPAN_DUMP") && !st->spans.empty()) {
        static const char* names[SKX_N_STAGES] = {"sketch", "dictionary", "scan", "transpose", "rank"};
        const hipEvent_t base = st->spans.front().a;
        for (auto& sp : st->spans) {
            float a = 0, b = 0;
            if (hipEventElapsedTime(&a, base, sp.a) == hipSuccess && hipEventElapsedTime(&b, base, sp.b) == hipSuccess)
                fprintf(stderr, "[skx span] %-10s %9.3f %9.3f  (%.3f ms)\n", names[sp.stage], a, b, b - a);
        }
        fprintf(stderr, "[skx span] ----\n");
    }
#endif
    for (auto& sp : st->spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { st->ms[sp.stage] += ms; st->launches[sp.stage] += 1; }
        st->ev_pool.push_back(sp.a); st->ev_pool.push_back(sp.b);
    }
    st->spans.clear();
}
SKX_API int skx_stream_set_profiling(skx_stream* st, int enabled) {
    if (!st) return fail(SKX_ERR_INVALID, "NULL stream");
    st->profiling = enabled < 0 ? 0 : enabled > 2 ? 1 : enabled;
    return SKX_OK;
}
SKX_API int skx_stream_profile(skx_stream* st, double* ms, uint64_t* launches) {
    if (!st) return fail(SKX_ERR_INVALID, "NULL stream");
    SKXCHK(use_device(st->device));
    SKXCHK(flush_pending(st));
    HIPCHK(hipStreamSynchronize(st->hs0));
    HIPCHK(hipStreamSynchronize(st->hs1));
    HIPCHK(hipStreamSynchronize(st->hs));
    HIPCHK(hipStreamSynchronize(st->hs2));
    collect_spans(st);
    for (int i = 0; i < SKX_N_STAGES; ++i) {
        if (ms) ms[i] = st->ms[i];
        if (launches) launches[i] = st->launches[i];
        st->ms[i] = 0; st->launches[i] = 0;
    }
    return SKX_OK;
}

// ---- one pass: reads [ra, rb) of the batch, pairs [p_base, p_base + P)
// inserted: the pairs of this pass were already gathered / inserted into the hash set on the sketch stream (the usual
// case: the whole batch is one pass and process_batch queued launch_dict_insert right behind the sketcher)
// q_rows: an upper bound of the pass's distinct query hashes (<= qcap): P itself, or -- when the host knows it from the
// speculative gather -- |Q|
// The bit matrices of a pass (M, Mq, row flags) have `qcap` rows: 65 536 by default -- 0.33 GB + 2 x 0.33 GB at C2, where a batch of ~100k
// reads holds ~10 k distinct member hashes when the collection's genomes share most of their hashes.  A collection with millions of
// strain-specific hashes (SURVEY.md 8(d)'s SNP clone tree at k = 16: 4.4 M distinct reference hashes below the largest one, 45 % of ALL
// canonical 16-mers in that range -- every sequencing error has an even chance to hit somebody's private hash) leaves ~160 k distinct
// hashes per batch: cut into passes of 65 536 pairs that was SEVEN scans per batch (10 M reads/s instead of 130 M).  Without a
// "stream_query_rows" policy the matrices therefore GROW when a batch (or a group of batches that would share a pass) needs more rows:
// everything in flight is waited for, the new arrays are allocated before the old ones are freed, at most a quarter of the free device
// memory is taken.  Not on the steady path: a stream grows once or twice, then its groups fit.
static int queue_chains(skx_stream* st, bool block);
static int queue_chains_until(skx_stream* st, int leave);
static void update_cand_hint(skx_stream* st);
static int staged_rows(skx_stream* st, void* slot);
static int grow_query_rows(skx_stream* st, u64 want_rows) {
    if (!st->qcap_auto) return SKX_OK;
    SKXCHK(queue_chains(st, true));  // (the latest pass's ranking still wants the arrays that are about to be replaced)
    const skx_ref* ref = st->ref;
    const u32 n_pad = ref->n_pad, n_gw = n_pad / 64;
    const size_t mq_words = (size_t)((n_gw + skx::kRankWords - 1) / skx::kRankWords * skx::kRankWords);
    for (hipStream_t h : {st->hs0, st->hs1, st->hs, st->hs2}) if (h) HIPCHK(hipStreamSynchronize(h));
    for (int i = 1; i < st->n_lanes; ++i) if (st->lane[i].s) HIPCHK(hipStreamSynchronize(st->lane[i].s));
    size_t mem_free = 0, mem_total = 0;
    HIPCHK(hipMemGetInfo(&mem_free, &mem_total));
    const u64 per_row = (st->static_dense ? 0 : (u64)n_pad / 8 * (st->d_mint ? 2 : 1)) + 2 * mq_words * 8 + 2 * (u64)(n_pad / (skx::kRankWords * 64)) / 8 + 1 +
                        5ull * skx::kPassBatchesMax * 4;  // (+ the per-batch row counters, the two rare-row maps, the long-row lists)
    want_rows += st->sd64;  // (the caller counts distinct query hashes: the static dense rows come on top)
    static const int grow_div = skx::knob("SKX_GROW_DIV") ? std::max(1, atoi(skx::knob("SKX_GROW_DIV"))) : 4;  // experiment knob
    u64 rows = std::min<u64>(want_rows, (u64)(mem_free / grow_div) / per_row);  // (a quarter of what is free: the one array family a dense stream cannot do without)
    rows = std::min<u64>(rows, st->pcap) / 64 * 64;
    if (rows <= st->qcap) return SKX_OK;  // no room (or nothing to gain): the batch is cut into passes as before
    u64 *m = nullptr, *mint = nullptr, *mq[2] = {nullptr, nullptr}, *ra[2] = {nullptr, nullptr};
    u32 *wb[2] = {nullptr, nullptr}, *cnt = nullptr, *sm[2] = {nullptr, nullptr};
    uint2* lrow = nullptr;
    auto undo = [&]() { for (void* q : {(void*)m, (void*)mint, (void*)mq[0], (void*)mq[1], (void*)ra[0], (void*)ra[1], (void*)wb[0], (void*)wb[1], (void*)cnt, (void*)sm[0], (void*)sm[1], (void*)lrow}) if (q) (void)hipFree(q); (void)hipGetLastError(); };
    hipError_t e = st->static_dense ? hipSuccess : hipMalloc(&m, (size_t)(rows / 64 + 2) * n_pad * 8);  // (static dense rows: M does not grow)
    if (e == hipSuccess && st->d_mint) e = hipMalloc(&mint, (size_t)(rows / 64 + 2) * n_pad * 8);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMalloc(&mq[i], ((size_t)rows + 128) * mq_words * 8);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMalloc(&ra[i], (size_t)(n_pad / (skx::kRankWords * 64)) * (rows / 64 + 2) * 8);
    for (int i = 0; i < 2 && e == hipSuccess && st->d_wb[i]; ++i) e = hipMalloc(&wb[i], ((size_t)rows / 64 + 1) * ref->n_tiles * 16);
    if (e == hipSuccess) e = hipMalloc(&cnt, (size_t)skx::kPassBatchesMax * ((size_t)rows + 128) * 4);
    if (e == hipSuccess && st->d_lrow) e = hipMalloc(&lrow, (size_t)skx::kPassBatchesMax * ((size_t)rows + 128) * 8);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMalloc(&sm[i], (size_t)skx::kPassBatchesMax * ((size_t)rows + 128) * 4);
    if (e == hipSuccess && m) e = hipMemset(m, 0, (size_t)(rows / 64 + 2) * n_pad * 8);  // (M is all-zero between passes)
    // (the group-major matrices and their row flags start all-zero: clean under any row stride -- d_mqext below)
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMemset(mq[i], 0, ((size_t)rows + 128) * mq_words * 8);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMemset(ra[i], 0, (size_t)(n_pad / (skx::kRankWords * 64)) * (rows / 64 + 2) * 8);
    if (e == hipSuccess) {
        const u32 any[4] = {skx::mq_any_stride(), 0xFFFFFFFFu, skx::mq_any_stride(), 0xFFFFFFFFu};
        e = hipMemcpy(st->d_mqext, any, sizeof(any), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && mint) e = hipMemset(mint, 0, (size_t)(rows / 64 + 2) * n_pad * 8);
    if (e == hipSuccess) e = hipDeviceSynchronize();  // (the zero-fills ran on the null stream)
    if (e != hipSuccess) { undo(); return SKX_OK; }   // (no memory for it: as before)
    if (m) { (void)hipFree(st->d_m); st->d_m = m; }
    if (mint) { (void)hipFree(st->d_mint); st->d_mint = mint; }
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(st->d_mq[i]); st->d_mq[i] = mq[i];
        (void)hipFree(st->d_rowany[i]); st->d_rowany[i] = ra[i];
        if (wb[i]) { (void)hipFree(st->d_wb[i]); st->d_wb[i] = wb[i]; }
        (void)hipFree(st->ps[i].smap); st->ps[i].smap = sm[i];
    }
    (void)hipFree(st->d_rowcnt); st->d_rowcnt = cnt;
    if (lrow) { (void)hipFree(st->d_lrow); st->d_lrow = lrow; }
    st->qcap = (u32)rows;
    st->mq_stride[0] = st->mq_stride[1] = 0;
    st->qrows_grown += 1;
    return SKX_OK;
}

// The scan of the reference for the pass that will use buffer set b (static dense dictionary only): M_s[b] = membership of the static
// dense hashes.  Independent of every batch, so it is queued as early as the pass is known to come -- batch_front, when the pass's
// first batch is sketched: an HBM-bound kernel beside the VALU-bound sketches -- or, failing that, by the pass itself.  M_s[b] is
// all-zero here: zeroed at creation, and whoever consumed it last (transpose / m_clear of the pass two back, queued on this stream
// before) zeroed what it read.
static int queue_static_scan(skx_stream* st, int b) {
    if (!st->static_dense || st->m_ready[b]) return SKX_OK;
    const skx_ref* ref = st->ref;
    st->m_ready[b] = true;
    if (ref->n_sd == 0) return SKX_OK;  // (every hash of the reference is rare: nothing to scan for)
    if (st->reuse_m && st->m_filled[b]) return SKX_OK;  // (policy reuse_membership: the rows of an earlier scan were kept)
    st->m_filled[b] = true;
    // (experiment knob SKX_SCAN_BIGSLICE=1, with SKX_RB=128: the lean kernel's instance for slices of up to 510 entries over bands of 128
    // rows = TWO of the default bands per workgroup, one result tile, every word flushed once per pair of bands)
    static const bool big_slices = skx::knob("SKX_SCAN_BIGSLICE") && atoi(skx::knob("SKX_SCAN_BIGSLICE")) != 0;
    {
        Span sp(st, 2, st->hs);
        skx::launch_scan(st->hs, ref->d_mat, ref->s, ref->n_tiles, ref->rb, ref->n_bands, ref->d_qs, ref->d_win_s, st->d_ms[b], nullptr, ref->n_pad,
                         false, true, nullptr, st->d_mdirty_s, true, 0u, big_slices);
    }
    {
        Span sp(st, 1, st->hs);
        skx::launch_exceptions(st->hs, ref->d_exc_g, ref->d_exc_h, ref->n_exc, ref->d_qs + ref->sd_tail0, nullptr, st->d_ms[b], ref->n_pad, st->d_mdirty_s, nullptr,
                               ref->sd_tail0, ref->n_sd - ref->sd_tail0);
    }
    HIPCHK(hipGetLastError());
    return SKX_OK;
}

static int run_pass_multi(skx_stream* st, SubPass* subs, int n_sub, bool update_table, bool inserted, u32 q_rows) {
    const skx_ref* ref = st->ref;
    hipStream_t hs0 = st->hs0, hs = st->hs;
    const skx::Species spc = ref->species();
    const u32 n_pad = ref->n_pad;
    u32 P = 0;
    for (int i = 0; i < n_sub; ++i) P += subs[i].P;
    const u32 ra = subs[0].ra, rb = subs[0].rb, p_base = subs[0].p_base;  // (the non-speculative gather below: single-batch passes only)
    if (!inserted && n_sub != 1) return fail(SKX_ERR_HIP, "internal: a shared pass needs its pairs gathered by the front halves");
    const u32 n_bt = ref->n_bands * ref->n_tiles;
    const u32 q_bound = std::min(P, q_rows);
    if (q_bound > st->qhash_cap()) return fail(SKX_ERR_HIP, "internal: pass of %u query rows exceeds the matrices' %u", q_bound, st->qhash_cap());
    // rows per group of the group-major bit matrix of this pass (the rows behind the dense ones start on a word boundary: up to 63 more
    // than the dictionary has hashes)
    const bool sdm = st->static_dense;  // the dense rows are the reference's static ones: rows [0, sd64), the pass's rare rows behind
    // (... and a buffer set keeps the stride of its last pass while that is large enough and not more than twice what this pass needs:
    // under an unchanged stride rare_to_mq_kernel writes only the words that changed -- d_mqext; a new stride is taken a quarter larger
    // than needed, so that the next passes fit it)
    const u32 nq_need = st->sd64 + ((q_bound + 63) / 64) * 64 + 64;
    const u32 nq_max = st->sd64 + ((st->qhash_cap() + 63) / 64) * 64 + 64;
    u32& nq_keep = st->mq_stride[st->buf];
    if (!(nq_keep >= nq_need && nq_keep <= 2 * nq_need + 4096)) nq_keep = std::min(nq_max, (nq_need + nq_need / 4 + 63) / 64 * 64);
    const u32 nq_rows = nq_keep;
    // the ranking of earlier passes whose candidates have been published by now; the pass TWO back (it used this pass's buffer set)
    // must be through -- the one before this may still be waiting for its scan
    // (the pass ONE back is queued BEHIND this pass's front half, further down: its eight chains are a hundred launches, 0.3-0.4 ms of
    // this thread, and the dictionary and scan of this pass need not wait for them)
    if (st->pc_n == 2) SKXCHK(queue_chains_until(st, 1));
    update_cand_hint(st);
    const int b = st->buf;                      // buffer set handed from stage to stage for this pass
    st->buf ^= 1;
    const int slot = st->pslot;                 // ... and its slot of the pair lists
    st->pslot = (st->pslot + 1) % 3;
    u32 *d_pair_r = st->d_pair_r[slot], *d_pair_q = st->d_pair_q[b], *d_poff = st->d_poff_pass[slot];
    u32 *d_nq = st->d_nq[b], *d_win = st->d_win[b];
    u64 *d_mq = st->d_mq[b], *d_q = st->d_q[b];
    u64* const d_m = sdm ? st->d_ms[b] : st->d_m;  // the pass's M: the static rows' (filled by queue_static_scan), or all of its dictionary's
    u32* d_grp_any = st->d_grp_any[b];
    skx_stream::PassSet& ps = st->ps[b];
    u32* const d_nd = st->d_nd_hs;  // (scan-stream scratch; ps.nd, the copy the chains read, is written in wait_back)
    const u32 qstride = st->qcap + 128;         // rows per batch of the per-row arrays (d_cnt, smap)

    // |Q| per pair from the latest pass whose dictionary is known to be complete (split dictionaries: also the dense share)
    const bool split_dict = ref->d_kt_key != nullptr;
    auto hint_nq = [&](int i) -> u32 { return split_dict ? st->h_nd[2 * i] : st->h_nq[i]; };
    auto take_hint = [&](int i, u32 pairs) {
        st->nq_per_pair = std::min(1.0, (double)hint_nq(i) / std::max<u32>(pairs, 1u));
        if (split_dict && st->h_nd[2 * i]) st->nd_frac = std::min(1.0, (double)st->h_nd[2 * i + 1] / st->h_nd[2 * i]);
        st->hint_pairs[i] = 0;
        st->have_hint = true;
    };
    for (int i = 0; i < 2; ++i)
        if (st->hint_pairs[i] && hipEventQuery(st->ev_dict[i]) == hipSuccess) take_hint(i, st->hint_pairs[i]);
    u64 nq_est = std::max<u64>(1, (u64)(P * st->nq_per_pair));
    // (a pass whose pairs the front halves gathered knows its |Q| exactly: the gather counted the keys new to the hash set.  The
    // per-pair figure of an earlier pass misleads when passes differ in size -- a group of eight batches has eight times the
    // pairs of a single batch but hardly more distinct hashes, and was sent to the dense-dictionary scan kernel)
    const bool nq_known = inserted && q_rows != 0xFFFFFFFFu;
    if (nq_known) nq_est = std::max<u64>(1, q_rows);
    // The lean scan kernel's results go straight into M (skx::scan_lean_into_m; the slab form of round 3 only when an experiment
    // knob asks for it, in which case the dictionary stage also builds the word -> bands lists).
    // Experiment knob SKX_SCAN_RUN (default 0 = the lean kernel): scan_run_kernel, RUNS of bands per workgroup -- measured slower
    // (DESIGN.md).  The run length from the expected entries of a band's slice -- ~3.1 x rows x |Q| / s: the order statistics of a
    // tile's 256 genomes spread a band's hash range to three times one genome's -- so that the union window of the run (adjacent
    // bands overlap by two thirds) stays inside one window of the kernel's tables with a third to spare; 1 .. 4; n > 0 forces n.
    static const int run_env = skx::knob("SKX_SCAN_RUN") ? atoi(skx::knob("SKX_SCAN_RUN")) : 0;
    u32 scan_run = 0;
    if (run_env != 0 && ref->n_bands <= 65535u) {
        const double per_band = (double)ref->rb * (double)nq_est * st->nd_frac / (double)std::max<u32>(ref->s, 1u);
        scan_run = 1;
        while (scan_run < 4u && (scan_run + 1 + 2.1) * per_band * 1.35 <= (double)skx::scan_run_cap()) ++scan_run;
        if (run_env > 0) scan_run = (u32)std::min(run_env, 64);
    }
    // (round 5: always into M -- the pass's tables and the candidates' columns are read from it; the slab form of round 3 is gone
    // from the experiments build too)
    const bool into_m = true;

    // ---- dictionary (scan stream hs; the pair gather possibly ran on the sketch stream already).  Set b was last used two
    // passes ago: by that pass's dictionary / scan / transpose on THIS stream (Q, windows, hash set: ordered by the stream)
    // and by its ranking on the back stream (pair -> query index, Mq, group flags) -- the wait for that ranking sits
    // further down, right before those are overwritten.
    (void)hs0;
    SKX_MARK("pass: begin, subs", n_sub);
    if (sdm && P > 0) SKXCHK(queue_static_scan(st, b));  // (normally queued long since: batch_front of the pass's first batch)
    for (int i = 0; i < n_sub; ++i) HIPCHK(hipStreamWaitEvent(hs, st->ev_sketch[subs[i].side], 0));  // the batches' sketches and pair offsets
    st->front_pending[b] = false;  // (this stream recorded it)
    // (inserted: the sketch stream also copied the pass's pair offsets -- this stream then never touches the sketch buffers,
    // which the next batch's sketch is free to overwrite)
    if (!inserted) {
        if (st->pslot_pending[slot]) { HIPCHK(hipStreamWaitEvent(hs, st->ev_pslot[slot], 0)); st->pslot_pending[slot] = false; }
        HIPCHK(hipMemcpyAsync(d_poff, st->d_poff + ra, ((size_t)(rb - ra) + 1) * 4, hipMemcpyDeviceToDevice, hs));
        subs[0].d_poff = d_poff;
    }
    const u64* scan_q = d_q;    // the dictionary the scan works on (its dense part when the reference has a rare-hash index)
    const u32* scan_nq = d_nq;
    if (P > 0) {
        Span sp(st, 1, hs);
        if (!inserted)
            skx::launch_dict_insert(hs, st->d_sk, st->cur_stride, st->d_poff, ra, rb, p_base, st->d_pair_h[b], d_pair_r, st->d_ht[b],
                                    st->ht_slots, st->d_dict_ctr[b], st->pcap, st->d_len);
        skx::launch_dict_rest(hs, st->d_ht[b], st->ht_slots, ref->max_ref, st->d_slot_off, st->d_bcount, st->d_bbase, st->d_btot,
                              st->d_dict_ctr[b], d_q, d_nq);
        if (split_dict) {
            // dense hashes -> rows [0, nd), what the scan looks for; the others -> the rows behind, filled from the genome lists
            skx::launch_classify(hs, d_q, d_nq, q_bound, st->rare_index(), st->d_qinfo, st->d_qloc, st->d_cls_bsum, st->d_qd, d_nd,
                                 st->d_qrow, st->d_sslot, st->h_nd + 2 * b);
            scan_q = st->d_qd; scan_nq = d_nd;
        } else {
            skx::launch_nd_from_nq(hs, d_nq, d_nd, st->h_nd + 2 * b);
        }
        if (sdm) {
            // (the scan's dictionary and its windows are the reference's)
        } else if (st->d_hbuf && !into_m) {  // (round 3's slab form only: windows + word -> bands in one launch; also hands |Q| to the host)
            skx::launch_word_bands(hs, d_win, ref->n_tiles, ref->n_bands, scan_nq, st->d_wb[b], ref->d_lo, ref->d_hi, scan_q, split_dict ? nullptr : &st->h_nq[b]);
        } else {
            skx::launch_window(hs, ref->d_lo, ref->d_hi, n_bt, scan_q, scan_nq, d_win, split_dict ? nullptr : &st->h_nq[b]);
        }
        st->hint_pairs[b] = P;
    }
    if (P == 0) HIPCHK(hipMemsetAsync(d_nd, 0, 16, hs));  // (no pairs: no dictionary ran)
    HIPCHK(hipGetLastError());
    if (!inserted) {  // this side of the sketch buffers may be overwritten once the gather above has run
        HIPCHK(hipEventRecord(st->ev_skread[st->side], hs));
        st->sk_reader_pending[st->side] = true;
    }
    HIPCHK(hipEventRecord(st->ev_dict[b], hs));

    // the very first pass of a stream has no hint: wait for its dictionary once rather than run the heaviest variant
    // (split dictionaries: until a pass has told which share of the hashes is dense, every pass is worth that wait once)
    if (!sdm && ((!st->have_hint && !nq_known && P > 0) || (split_dict && !st->have_split_hint && P > 0))) {
        HIPCHK(hipStreamSynchronize(hs));
        take_hint(b, P);
        st->have_split_hint = true;
        if (!nq_known) nq_est = std::max<u64>(1, hint_nq(b));
    }
    // what the SCAN looks for: the dense part of the dictionary
    const u64 nd_est = sdm ? std::max<u32>(1u, ref->n_sd) : split_dict ? std::max<u64>(1, (u64)((double)nq_est * st->nd_frac * 1.1) + 64) : nq_est;

    // ---- scan + transpose (same stream, HBM-bound)
    const u32 n_grp_all = n_pad / (skx::kRankWords * 64);
    u32* d_mdirty = d_grp_any + n_grp_all;
    SKX_MARK("pass: dictionary queued", 0);
    HIPCHK(hipMemsetAsync(d_mdirty, 0, 4, hs));  // raised by writers of M (read by the transpose, this stream)
    SKX_MARK("pass: memset mdirty", 0);
    // the ranking two passes back reads this set's pair -> query index, Mq and group flags: from here on they are rewritten
    auto wait_back = [&]() -> int {
        if (st->back_pending[b]) { HIPCHK(hipStreamWaitEvent(hs, st->ev_back[b], 0)); st->back_pending[b] = false; }
        HIPCHK(hipMemcpyAsync(ps.nd, d_nd, 16, hipMemcpyDeviceToDevice, hs));  // (the counts this pass's chains will read)
        HIPCHK(hipMemsetAsync(d_grp_any, 0, (size_t)n_grp_all * 4, hs));  // raised by the transpose
        if (P > 0) {
            skx::launch_pair_q(hs, st->d_pair_h[b], P, d_q, d_nq, d_pair_q, split_dict ? st->d_qrow : nullptr, st->d_bbase, st->d_btot, ref->max_ref);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(st->ev_pairq[b], hs));  // the set's hash set and pair hashes may be refilled
        st->pairq_pending[b] = true;
        return SKX_OK;
    };
    // ---- the table as every batch of the pass begins (and as the pass ends), and the genomes each batch's ranking has to look at
    // (skx_kernels.hip, "the table without the ranking").  All on the scan stream, before the transpose re-zeroes M.
    bool ranked = false, all_forced = false, all_ranked = true;
    for (int i = 0; i < n_sub; ++i) {
        const bool r = st->top_k && subs[i].d_topk_idx && subs[i].d_topk_sum;
        ranked = ranked || r; all_ranked = all_ranked && r && !subs[i].d_shared;
    }
    const u32* only_if = nullptr;  // device flag: does any batch of the pass rank on the FULL matrix?  (NULL: yes, unconditionally)
    u32 seq = 0;
    // A stream whose batches ALL have more candidates than a compact ranking takes (the bench's near-tie of 40 000 genomes) gains
    // nothing from knowing its tables early: three passes out of four then take the table out of their ranking chains, as in rounds
    // 1-4 (no row counts, no gains, no candidate selection: ~0.65 ms of scan-stream work per C2 pass, 9 % of the reads/s next to the
    // sketches) -- and every such chain counts what its batch's candidates would have been (cand_count_kernel: two reads of the
    // table), so the stream notices when a leader has emerged.
    static const int legacy_env = skx::knob("SKX_TABLE_LEGACY") ? atoi(skx::knob("SKX_TABLE_LEGACY")) : 1 << 30;  // experiment knob: passes in a row (0: never)
    static const int cand_env0 = skx::knob("SKX_CAND") ? atoi(skx::knob("SKX_CAND")) : 1;
    // (... and the FIRST pass of a sample: on a table of zeros every genome of a species is a candidate of the first batch, whatever
    // follows -- with more genomes than a compact ranking takes, the pass that would only find that out is not worth its wait)
    // (top_k >= 2: the full top-k ranking of a batch costs several times the top-1's -- there the pass that finds out that batches 3, 4, ...
    // of a sample already have a leader is worth its wait: --top 16 on the truth-strain stream 34.1 -> 41.8 M reads/s from a fresh table,
    // top-1 100.8 against 101.3 M)
    const bool fresh_overflow = st->fresh_table && ref->max_species > skx::kCandCap && st->top_k <= 1;
    // (the SECOND pass of a sample goes by a count from the second half of the first pass -- the first batches of ANY sample say
    // "everything" -- and looks for itself when none has arrived yet)
    const bool second_blind = st->passes_since_fresh == 1 && !((int)(st->lcount_seen - st->lcount_floor) > 0);
    const bool legacy = update_table && all_ranked && cand_env0 && legacy_env > 0 &&
                        (fresh_overflow || (st->hint_all_overflow && !second_blind && (int)st->legacy_run < legacy_env));
    st->legacy_run = legacy ? st->legacy_run + 1 : 0;
    if (update_table) { st->passes_since_fresh = st->fresh_table ? 1 : st->passes_since_fresh + 1; st->fresh_table = false; }
    const bool long_rows = split_dict && ref->d_mlong != nullptr;
    // (nothing is sketched beside a sample's last pass, or beside a synchronous push: the walks over the rare rows' lists and bit rows may
    // fill the chip -- beside the next group's sketches the same grids cost 3-5 % of the reads/s.  Measured on the truth-strain workload:
    // batches of twice the size 75 -> 91 M reads/s, 20 batches from a fresh table +1 %.  Tried and dropped: those gains on a stream of
    // their own beside the scan -- nothing gained from a fresh table, and the mere existence of one more HIP stream cost the steady state
    // 12 % on both workloads (176 -> 155 M, 133 -> 120 M reads/s: more streams than hardware queues).)
    static const int tail_env = skx::knob("SKX_TAIL_SCALE") ? atoi(skx::knob("SKX_TAIL_SCALE")) : 8;  // experiment knob
    const u32 walk_scale = st->tail_pass ? (u32)std::max(1, tail_env) : 1u;  // (nothing is sketched beside a sample's last pass: its list walks may fill the chip)
    skx::PassBatches pbt;
    pbt.n = (u32)n_sub;
    for (int i = 0; i < n_sub; ++i) pbt.p_off[i] = subs[i].p_off;
    pbt.p_off[n_sub] = P;
    auto rare_gains = [&](hipStream_t on) -> int {  // row counts known: the short lists' adds, the bit rows' bit-sliced sums
        const skx::RareIndex ri = st->rare_index();
        const skx::LongRows lrows = st->long_rows();
        const skx::PatRows prows = st->pat_rows();
        skx::launch_gain_sparse(on, d_nd, q_bound, st->d_rowcnt, qstride, (u32)n_sub, n_pad, st->d_gain_s, st->d_sslot, ri, long_rows ? &lrows : nullptr, walk_scale,
                                (long_rows && st->use_pat) ? &prows : nullptr);
        if (long_rows) skx::launch_gain_long(on, lrows, ri, d_nd, st->d_rowcnt, qstride, (u32)n_sub, n_pad, st->d_gain_l, walk_scale);
        // the pattern rows' part: gain_l[b][g] += sum over the patterns of hist[b][p] x PM[p][g] -- the dense rows' kernel on the pattern matrix
        if (long_rows && st->use_pat)
            skx::launch_gain_dense(on, ri.pm, nullptr, n_pad, ri.d_npat, ri.n_pat, st->d_hist, st->npat_pad, (u32)n_sub, st->d_gain_l);
        HIPCHK(hipGetLastError());
        return SKX_OK;
    };
    auto row_counts = [&]() -> int {
        HIPCHK(hipMemset2DAsync(st->d_rowcnt, (size_t)qstride * 4, 0, (size_t)nq_rows * 4, (size_t)n_sub, hs));
        skx::launch_pass_hist(hs, d_pair_q, pbt, st->d_rowcnt, qstride);
        if (split_dict) HIPCHK(hipMemsetAsync(st->d_gain_s, 0, (size_t)n_sub * n_pad * 4 * skx::gain_sparse_stride(), hs));
        if (long_rows) {  // the rows with a bit row: listed per batch (by the short lists' walk), added up with bit-sliced counters
            HIPCHK(hipMemsetAsync(st->d_nlrow, 0, skx::pass_counter_bytes(), hs));
            if (st->d_inb) {
                HIPCHK(hipMemsetAsync(st->d_inb, 0, (size_t)n_sub * ref->n_lw * 8, hs));
                HIPCHK(hipMemsetAsync(st->d_hit, 0, (size_t)n_sub * ref->n_lw * 8, hs));
            }
            HIPCHK(hipMemsetAsync(st->d_gain_l, 0, (size_t)n_sub * n_pad * 4, hs));
            if (st->use_pat) {
                HIPCHK(hipMemsetAsync(st->d_hist, 0, (size_t)n_sub * st->npat_pad * 4, hs));
                HIPCHK(hipMemsetAsync(st->d_nprow, 0, skx::pass_counter_bytes(), hs));
            }
        }
        return SKX_OK;
    };
    const u32 n_words = nq_rows / 64;
    bool split = false, lean_used = false;
    if (P > 0 && sdm) {
        st->total_passes += 1; st->lean_passes += 1;
        SKXCHK(wait_back());  // (the scan was queued with the pass's first batch, or at the top of this function)
    } else if (P > 0) {
        // many query words per band (dense batches): most flushed words are interior -> split arrays pay off
        static const int split_env = skx::knob("SKX_SCAN_SPLIT") ? atoi(skx::knob("SKX_SCAN_SPLIT")) : -1;
        split = split_env >= 0 ? split_env != 0 : (nd_est * ref->rb / ref->s >= 192);
        if (split && !st->d_mint) {  // dense dictionaries only: most streams never get here
            const size_t bytes = (size_t)(st->qcap / 64 + 2) * n_pad * 8;
            if (hipMalloc(&st->d_mint, bytes) == hipSuccess) {
                HIPCHK(hipMemsetAsync(st->d_mint, 0, bytes, hs));  // (the transpose re-zeroes what it reads)
            } else {
                (void)hipGetLastError();
                st->d_mint = nullptr;
                split = false;  // no room: the plain variant gives the same bits
            }
        }
        static const int big_env = skx::knob("SKX_SCAN_BIG") ? atoi(skx::knob("SKX_SCAN_BIG")) : -1;
        const bool big = big_env >= 0 ? big_env != 0 : (nd_est * ref->rb / ref->s >= 900);
        const bool run_scan = scan_run != 0 && !split && !big;  // (dense dictionaries: scan_kernel's variants, as before)
        const bool lean = run_scan || skx::scan_lean_applies(ref->n_bands, split, big);
        lean_used = lean;
        // Experiment knob SKX_SCAN_BIGSLICE=1: the lean kernel's BIG instance (slices of up to 510 entries in one pass of the
        // branch-free probe: two-byte directory, three entries per probe, nine result words, 31 KB of LDS).  A (band, tile) slice
        // holds ~3.1 x rows per band x |Q| / s entries on average -- C4: ~245 with one batch per pass, ~390 with eight, so most of
        // its blocks leave the 254-entry probe for the multi-window walk.  Measured at C4 (tools/ab.sh, twice each): the scan
        // of a shared pass 2.90 -> 2.65 ms in the pipeline (0.52 -> 0.57 of peak), but reads/s 48.1 -> 47.5-48.0 M from a fresh
        // table, 57.0 -> 53.8 M steady, single-batch passes 0.66-0.68 -> 0.60 alone: the walk costs less than three LDS reads per
        // element and twice the LDS per block do beside the other streams.  Off.
        static const int bigslice_env = skx::knob("SKX_SCAN_BIGSLICE") ? atoi(skx::knob("SKX_SCAN_BIGSLICE")) : 0;
        const bool big_slices = lean && into_m && bigslice_env != 0;
        st->total_passes += 1; st->lean_passes += lean ? 1 : 0;
        {
            Span sp(st, 2, hs);
            // sparse dictionaries: the lean kernel with its single-owner slabs; dense ones: scan_kernel's variants into M
            // d_m / d_mint are all zero here: zeroed at creation, and the transpose of every pass zeroes what it read
            skx::launch_scan(hs, ref->d_mat, ref->s, ref->n_tiles, ref->rb, ref->n_bands, scan_q, d_win, d_m,
                             split ? st->d_mint : nullptr, n_pad, big, lean && !run_scan, (lean && !run_scan) ? st->d_hbuf : nullptr, d_mdirty, into_m,
                             run_scan ? scan_run : 0u, big_slices);
        }
        {
            Span sp(st, 1, hs);
            // (after the scan: its persistent form writes complete words with plain stores, these OR single bits in)
            skx::launch_exceptions(hs, ref->d_exc_g, ref->d_exc_h, ref->n_exc, d_q, d_nq, d_m, n_pad, d_mdirty, split_dict ? st->d_qrow : nullptr);
        }
        SKXCHK(wait_back());
    }
    if (P == 0) SKXCHK(wait_back());

    if (update_table && !legacy && st->cum_writer) {  // (the table the last chain of a legacy pass left: the gains are added to it)
        HIPCHK(hipStreamWaitEvent(hs, st->cum_writer->ev_cum, 0));
        st->cum_writer = nullptr;
    }
    if (update_table && !legacy) {
        Span sp(st, 4, hs);
        const u32 n_sp = ref->n_species, cap = skx::kCandCap, rows_c = skx::kCandRows;
        const u64* m_int = split ? st->d_mint : nullptr;
        HIPCHK(hipMemsetAsync(st->d_gain, 0, (size_t)n_sub * n_pad * 4, hs));
        if (P > 0) {
            SKXCHK(row_counts());
            skx::launch_gain_dense(hs, d_m, m_int, n_pad, d_nd, sdm ? ref->n_sd : q_bound, st->d_rowcnt, qstride, (u32)n_sub, st->d_gain,
                                   (sdm && ref->n_species > 1) ? ref->d_segw : nullptr, ref->d_grp_sp);
            if (split_dict) SKXCHK(rare_gains(hs));
        }
        skx::launch_pass_tables(hs, st->d_cum, st->d_gain, (split_dict && P > 0) ? st->d_gain_s : nullptr,
                                (long_rows && P > 0) ? st->d_gain_l : nullptr, (u32)n_sub, n_pad, ps.tab);
        st->d_cum = ps.tab + (size_t)n_sub * n_pad;  // (readers: the next pass on this stream; everybody else behind ev_front / a flush)
        // candidates.  A batch that wants the per-read x per-genome debug matrix ranks on everything; so does every batch when the
        // experiment knob SKX_CAND=0 says so
        static const int cand_env = skx::knob("SKX_CAND") ? atoi(skx::knob("SKX_CAND")) : 1;
        static const int hint_env = skx::knob("SKX_CAND_HINT") ? atoi(skx::knob("SKX_CAND_HINT")) : 1;  // experiment knob: 0 = never predict
        (void)hint_env;
        all_forced = !cand_env;
        u32 force_full = all_forced ? 0xFFu : 0u;
        for (int i = 0; i < n_sub; ++i) if (subs[i].d_shared) force_full |= 1u << i;
        HIPCHK(hipMemsetAsync(st->d_cbad, 0, 64, hs));
        HIPCHK(hipMemsetAsync(st->d_nqc, 0, skx::pass_counter_bytes(), hs));
        const bool pat_rows = st->use_pat && long_rows && ranked && P > 0 && !all_forced;
        if (pat_rows) skx::launch_pat_nqc_init(hs, st->d_nqc, st->npat_pad);  // (the mapped rare rows come behind the patterns' rows)
        if (ranked) {
            HIPCHK(hipMemsetAsync(st->d_candmask, 0, (size_t)n_pad * 4, hs));
            skx::launch_cand_select(hs, ps.tab, n_pad, spc, (u32)n_sub, st->top_k, cap, ps.cand, st->d_candslot, ps.tabc, ps.ncand, st->d_cbad,
                                    st->d_candmask);
            HIPCHK(hipMemsetAsync(ps.grp_any_c, 0, (size_t)n_sub * st->n_grp_c * 4, hs));
            if (split_dict && P > 0 && !all_forced) {
                HIPCHK(hipMemsetAsync(ps.mqc, 0, (size_t)n_sub * st->n_grp_c * rows_c * skx::kRankWords * 8, hs));
                HIPCHK(hipMemsetAsync(ps.rowany_c, 0, (size_t)n_sub * st->n_grp_c * (rows_c / 64) * 8, hs));
                HIPCHK(hipMemset2DAsync(ps.smap, (size_t)qstride * 4, 0, (size_t)nq_rows * 4, (size_t)n_sub, hs));
                skx::launch_cand_sparse(hs, st->d_sslot, d_nd, q_bound, st->rare_index(), st->d_candmask, st->d_candslot, n_pad, st->d_cbad, (u32)n_sub, st->d_nqc,
                                        ps.smap, qstride, ps.mqc, (size_t)st->n_grp_c * rows_c * skx::kRankWords, rows_c, ps.rowany_c,
                                        st->n_grp_c * (rows_c / 64), ps.grp_any_c, st->n_grp_c, walk_scale);
                if (long_rows) {  // ... and the rows with a bit row: ANDed with the candidates' words
                    const u32 n_gw = n_pad / 64;
                    HIPCHK(hipMemsetAsync(st->d_cw, 0, (size_t)n_sub * n_gw * 8, hs));
                    HIPCHK(hipMemsetAsync(st->d_cbase, 0xFF, (size_t)n_sub * n_gw * 4, hs));
                    HIPCHK(hipMemsetAsync(st->d_ncwl, 0, 64, hs));
                    skx::launch_cand_words(hs, ps.cand, st->n_pad_c, (u32)n_sub, n_gw, st->d_cw, st->d_cbase, st->d_cwl, st->d_ncwl);
                    if (st->d_hit) skx::launch_cand_hit(hs, ps.cand, st->n_pad_c, (u32)n_sub, st->d_cbad, st->rare_index(), st->d_inb, st->d_hit);
                    skx::launch_cand_long(hs, st->long_rows(), st->rare_index(), d_nd, st->d_cw, st->d_cbase, st->d_cwl, st->d_ncwl, st->n_pad_c,
                                          st->d_cbad, (u32)n_sub, st->d_nqc, ps.smap, qstride, ps.mqc, (size_t)st->n_grp_c * rows_c * skx::kRankWords,
                                          rows_c, ps.rowany_c, st->n_grp_c * (rows_c / 64), ps.grp_any_c, st->n_grp_c, walk_scale, st->d_hit);
                }
                if (pat_rows) {  // ... and the rows that are a pattern + exceptions: the pattern's row, or the pattern's words with a few bits flipped
                    const skx::RareIndex ri = st->rare_index();
                    skx::launch_cand_pat_rows(hs, ri, d_nd, st->d_candmask, st->d_candslot, n_pad, st->d_cbad, (u32)n_sub, st->d_pcw, ps.mqc,
                                              (size_t)st->n_grp_c * rows_c * skx::kRankWords, rows_c, ps.rowany_c, st->n_grp_c * (rows_c / 64), ps.grp_any_c,
                                              st->n_grp_c);
                    skx::launch_cand_pat_map(hs, st->long_rows(), st->pat_rows(), ri, d_nd, st->d_candmask, st->d_candslot, n_pad, st->d_cbad, (u32)n_sub,
                                             st->d_nqc, ps.smap, qstride, st->d_pcw, ps.mqc, (size_t)st->n_grp_c * rows_c * skx::kRankWords, rows_c,
                                             ps.rowany_c, st->n_grp_c * (rows_c / 64), ps.grp_any_c, st->n_grp_c, q_bound, walk_scale);
                }
            }
        } else {
            HIPCHK(hipMemsetAsync(ps.ncand, 0, (size_t)n_sub * n_sp * 4, hs));
        }
        seq = ++st->cand_seq;
        skx::launch_cand_publish(hs, st->d_cbad, force_full, ps.ncand, st->d_nqc, d_nd, (u32)n_sub, n_sp, rows_c, ps.mode, ps.any_full,
                                 ps.nqc_total, ps.h_pub, seq);
        if (ranked && P > 0 && !all_forced)
            skx::launch_cand_gather_m(hs, d_m, m_int, n_pad, d_nd, sdm ? ref->n_sd : q_bound, ps.cand, st->n_pad_c, st->d_cbad, (u32)n_sub, ps.mc, rows_c / 64);
        only_if = ps.any_full;
        HIPCHK(hipGetLastError());
    }
    if (P > 0) {
        // the rows behind the dense ones of the FULL matrix: from the reference's rare-hash index.  With a bit row for every list of
        // more than eight genomes they go straight into the group-major matrix (rare_to_mq_kernel) and M -- hence the transpose --
        // holds the dense rows only; without (no room for the bit rows): bits from the genome lists into M, as the scan's.
        static const int direct_env = skx::knob("SKX_RARE_DIRECT") ? atoi(skx::knob("SKX_RARE_DIRECT")) : 1;  // experiment knob: 0 = through M
        const bool direct_rare = split_dict && ref->rare_direct && (direct_env != 0 || sdm);
        if (split_dict && !direct_rare) {
            Span sp(st, 1, hs);
            skx::launch_sparse_fill(hs, st->d_sslot, d_nd, st->rare_index(), d_m, n_pad, d_mdirty, q_bound, only_if);
        }
        if (!direct_rare) HIPCHK(hipMemsetAsync(st->d_mqext + 2 * b, 0, 8, hs));  // (the transpose alone writes the set: nothing is known about it afterwards)
        {
            Span sp(st, 3, hs);
            skx::launch_transpose_bits(hs, d_m, split ? st->d_mint : nullptr, n_pad, n_words, d_mq, direct_rare ? d_nd + 2 : d_nd + 3, d_grp_any,
                                       nullptr, st->d_wb[b], d_win, ref->n_tiles, d_mdirty, direct_rare ? nd_est : nq_est, st->d_rowany[b], only_if,
                                       sdm && st->reuse_m);
            if (direct_rare)
                skx::launch_rare_to_mq(hs, st->d_sslot, d_nd, st->rare_index(), d_mq, nq_rows, n_pad, st->d_rowany[b], d_grp_any, q_bound, only_if,
                                       st->d_mqext + 2 * b);
            if (only_if && !(sdm && st->reuse_m)) skx::launch_m_clear(hs, d_m, split ? st->d_mint : nullptr, n_pad, d_nd, only_if);
        }
        if (sdm) st->m_ready[b] = false;  // (consumed: transposed -- which zeroes what it reads -- or cleared)
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(st->ev_front[b], hs));
    st->front_pending[b] = true;
    (void)lean_used;

    // ---- the ranking chains are queued by queue_chains, once the pass's candidates are known
    SKXCHK(queue_chains(st, false));  // (the pass before this one, if its candidates have been published by now)
    skx_stream::PassChains& pc = st->pcq[(st->pc_head + st->pc_n) & 1];
    st->pc_n += 1;
    pc.pending = true; pc.ranked = ranked; pc.has_cand = update_table;
    pc.n_sub = n_sub; pc.b = b; pc.slot = slot; pc.P = P; pc.nq_rows = nq_rows; pc.seq = seq; pc.nq_est = nq_est;
    for (int i = 0; i < n_sub; ++i) pc.subs[i] = subs[i];
    pc.update_table = update_table;
    pc.forced_full = all_forced || legacy;
    pc.legacy = legacy;
    if (legacy) { pc.has_cand = false; all_forced = true; }
    SKX_MARK("pass: scan + transpose queued", 0);
    // (every batch ranks on everything whatever the candidates turn out to be: nothing to wait for -- the chains go out now, behind
    // those of the passes before, which then cannot wait either)
    if (all_forced) SKXCHK(queue_chains(st, true));
    return SKX_OK;
}

// The back half of a pass: its batches are ranked in order, consecutive ones on alternating LANES (skx_stream::RankLane).  Every
// chain starts from the table the pass's front half computed for it (ps.tab[i]: no chain waits for another one since round 5) and
// runs either on everything (Mq, 79 rank groups at C2) or -- when the batch's candidates fit -- on the compact problem: the
// candidates' columns (two rank groups per species), the same kernels, the slots mapped back to genome indices at the end.
// block: wait for the pass's published candidates (else return when they are not there yet).
static int queue_one(skx_stream* st, skx_stream::PassChains& pc, bool block, bool* done);
// hint_all_overflow from the most recent pass whose candidates have been published (either buffer set; forced passes publish what
// their candidates WOULD have been): did every batch have more candidates than the compact ranking takes?
static void update_cand_hint(skx_stream* st) {
    {   // the latest batch of a legacy pass whose count has arrived
        volatile u32* hl = st->h_lcount;
        const u32 seq = hl[1], count = hl[0];
        if (seq != st->lcount_seen && (int)(seq - st->lcount_floor) > 0 && hl[1] == seq) {
            st->hint_all_overflow = count > skx::kCandCap && st->top_k != 0;
            st->lcount_seen = seq;
        }
    }
    for (auto& q : st->ps) {
        volatile u32* hp = q.h_pub;
        const u32 seq = hp[2 * skx::kPassBatchesMax + 1], n_b = hp[2 * skx::kPassBatchesMax + 2];
        if (seq == 0 || (int)(seq - st->hint_seq) <= 0 || n_b == 0 || n_b > skx::kPassBatchesMax) continue;
        bool all_over = true;
        for (u32 i = 0; i < n_b; ++i) all_over = all_over && hp[skx::kPassBatchesMax + i] >= skx::kCandCap;
        if (hp[2 * skx::kPassBatchesMax + 1] != seq) continue;  // (rewritten while we looked)
        st->hint_all_overflow = all_over && st->top_k != 0;
        st->hint_seq = seq;
    }
}
// until at most `leave` passes wait for their chains (blocking)
static int queue_chains_until(skx_stream* st, int leave) {
    while (st->pc_n > leave) {
        bool done = false;
        SKXCHK(queue_one(st, st->pcq[st->pc_head], true, &done));
        st->pc_head ^= 1; st->pc_n -= 1;
    }
    return SKX_OK;
}
// block: all of them; else: those whose candidates have been published, in order
static int queue_chains(skx_stream* st, bool block) {
    if (block) return queue_chains_until(st, 0);
    while (st->pc_n > 0) {
        bool done = false;
        SKXCHK(queue_one(st, st->pcq[st->pc_head], false, &done));
        if (!done) break;
        st->pc_head ^= 1; st->pc_n -= 1;
    }
    return SKX_OK;
}
static int queue_one(skx_stream* st, skx_stream::PassChains& pc, bool block, bool* done) {
    *done = false;
    if (!pc.pending) { *done = true; return SKX_OK; }
    const skx_ref* ref = st->ref;
    const int b = pc.b, slot = pc.slot, n_sub = pc.n_sub;
    skx_stream::PassSet& ps = st->ps[b];
    u32 mode[kGroupMax] = {};
    if (pc.has_cand && !pc.forced_full) {
        volatile u32* hp = ps.h_pub;
        if (hp[2 * skx::kPassBatchesMax + 1] != pc.seq) {
            if (!block) return SKX_OK;  // (*done stays false)
            SKX_T0();
            for (u64 spins = 0; hp[2 * skx::kPassBatchesMax + 1] != pc.seq; ++spins)
                if ((spins & 0xFFFFu) == 0xFFFFu) {  // (look at the stream now and then so a fault cannot hang the caller)
                    const hipError_t e = hipStreamQuery(st->hs);
                    if (e != hipSuccess && e != hipErrorNotReady) return fail(SKX_ERR_HIP, "the scan stream failed: %s", hipGetErrorString(e));
                    if (e == hipSuccess && hp[2 * skx::kPassBatchesMax + 1] != pc.seq) {
                        (void)hipGetLastError();
                        if (hp[2 * skx::kPassBatchesMax + 1] != pc.seq) return fail(SKX_ERR_HIP, "internal: the pass's candidates were never published");
                    }
                    (void)hipGetLastError();
                }
            SKX_ACC(wait);
        }
        for (int i = 0; i < n_sub; ++i) mode[i] = hp[i];
    }
    update_cand_hint(st);
    pc.pending = false;
    *done = true;
    hipStream_t hs2 = st->hs2;
    const skx::Species spc = ref->species();
    const u32 n_pad = ref->n_pad, n_sp = ref->n_species;
    const u32 nq_rows = pc.nq_rows, P = pc.P;
    const bool update_table = pc.update_table;
    u32 *d_pair_r = st->d_pair_r[slot], *d_pair_q = st->d_pair_q[b];
    u64* d_mq = st->d_mq[b];
    u32* d_grp_any = st->d_grp_any[b];
    u32* const d_nq_rows = ps.nd + 3;  // rows of the pass's matrix (device)
    bool lane_used[skx_stream::kRankLanes] = {};
    for (int si = 0; si < n_sub; ++si) {
    SKX_MARK("rank: begin sub", si);
    const SubPass& sb = pc.subs[si];
    // (only the batches of a SHARED pass take turns: a pass of one batch has the next pass's scan and sketch beside its chain
    // already -- measured with one pass per batch, option stream_coalesce = 1: 83 M reads/s on one lane, 67 M on two)
    const int li = (update_table && n_sub > 1) ? (int)(st->rank_seq % (u64)st->n_lanes) : 0;
    skx_stream::RankLane& L = st->lane[li];
    hipStream_t ls = L.s;
    if (!lane_used[li]) { HIPCHK(hipStreamWaitEvent(ls, st->ev_front[b], 0)); lane_used[li] = true; }  // the pass's matrices and tables
    const u32 n_reads = sb.rb - sb.ra, sub_base = sb.p_base;
    u32* const d_topk_idx = sb.d_topk_idx;
    u64* const d_topk_sum = sb.d_topk_sum;
    const u32 out_r0 = sb.ra;
    const u32 n_seg = (n_reads + skx::kSegLen - 1) / skx::kSegLen;
    const bool ranked = st->top_k && d_topk_idx && d_topk_sum;
    const bool compact = update_table && ranked && mode[si] == 1u;
    // what the chain works on: everything, or the batch's candidates
    const u32 c_pad = compact ? st->n_pad_c : n_pad, c_gw = c_pad / 64, c_rows = compact ? skx::kCandRows : nq_rows;
    const skx::Species c_spc = compact ? skx::Species{st->d_spc_g0, ps.ncand + (size_t)si * n_sp, st->d_spc_grp, n_sp} : spc;
    const u64* c_mq = compact ? ps.mqc + (size_t)si * st->n_grp_c * skx::kCandRows * skx::kRankWords : d_mq;
    u32* c_grp_any = compact ? ps.grp_any_c + (size_t)si * st->n_grp_c : d_grp_any;
    const u64* c_rowany = compact ? ps.rowany_c + (size_t)si * st->n_grp_c * (skx::kCandRows / 64) : (P > 0 ? st->d_rowany[b] : nullptr);
    const u32* c_nq = compact ? ps.nqc_total + si : d_nq_rows;
    // the table the batch starts from: the pass's front half computed it (ps.tab / the candidates' tabc) -- or, legacy passes, the chain
    // before this one leaves it (rotating buffers, as in rounds 1-4)
    const bool legacy = pc.legacy && update_table;
    if (legacy && st->cum_writer && st->cum_writer != &L) HIPCHK(hipStreamWaitEvent(ls, st->cum_writer->ev_cum, 0));
    int tab_next = st->tab_cur;
    if (legacy) {
        do { tab_next = (tab_next + 1) % (st->n_lanes + 1); } while (st->d_tab[tab_next] == st->d_cum);
    }
    const u64* const cum_in = legacy ? st->d_cum : update_table ? (compact ? ps.tabc + (size_t)si * st->n_pad_c : ps.tab + (size_t)si * n_pad) : st->d_cum;
    const u32 *sub_pair_q = (compact ? ps.pair_qc : d_pair_q) + sb.p_off, *sub_pair_r = d_pair_r + sb.p_off, *sub_poff = sb.d_poff;
    if (update_table && ranked) { if (compact) st->batches_compact += 1; else st->batches_full += 1; }
    if (compact) {
        Span sp(st, 3, ls);
        // the candidates' dense rows: 64 x 64 bit transposes of their columns of M; the pairs in rows of the compact matrix
        if (P > 0)
            skx::launch_transpose_bits(ls, ps.mc + (size_t)si * (skx::kCandRows / 64) * st->n_pad_c, nullptr, st->n_pad_c, skx::kCandRows / 64,
                                       const_cast<u64*>(c_mq), ps.nd, c_grp_any, nullptr, nullptr, nullptr, 0, nullptr, pc.nq_est,
                                       const_cast<u64*>(c_rowany));
        skx::launch_cand_pair_rows(ls, d_pair_q + sb.p_off, sb.P, ps.nd, ps.smap + (size_t)si * ((size_t)st->qcap + 128), skx::kCandRows,
                                   ps.pair_qc + sb.p_off);
        HIPCHK(hipGetLastError());
    }
    const u32 prune_k = (ranked && st->top_k <= skx::rank_topk_fast_max()) ? st->top_k : 0u;
    // Pruned rankings count in two levels: the chunk sums of EVERYTHING first (one workgroup per (rank group, chunk of 1024
    // reads): a third less work than per segment), then -- once the chunk-level bounds are known -- the per-segment increments
    // of the (chunk, rank group)s that can hold a candidate.  Far from the start of a sample that is the leaders' groups and
    // little else; nobody reads the increments of the others (seg_prefix and the ranking kernels apply the same test).
    // It pays when most (chunk, group)s are dead -- C2: from the fifth batch of a sample on; the first batches, where nearly every
    // genome is still a candidate, would count everything twice.  The ranking samples the share of live (chunk, half group)s of
    // every batch (seg_prefix_kernel -> h_nq[2..3], a few batches stale when read here: both schemes are exact).
    static const int two_level_env = skx::knob("SKX_TWO_LEVEL") ? atoi(skx::knob("SKX_TWO_LEVEL")) : -1;  // experiment knob: 0 / 1 force
    const volatile u32* h_live = st->h_nq + 2;
    const u32 live_now = h_live[0], tested_now = h_live[1];
    static const u32 live_pct_env = skx::knob("SKX_LIVE_PCT") ? (u32)atoi(skx::knob("SKX_LIVE_PCT")) : 33u;  // experiment knob
    const bool mostly_dead = tested_now != 0 && (u64)live_now * 100 < (u64)tested_now * live_pct_env;
    // (a compact chain has two rank groups: one level)
    const bool two_level = !compact && prune_k != 0 && (two_level_env >= 0 ? two_level_env != 0 : mostly_dead);
    u32 *d_inc = L.d_inc, *d_csum_raw = L.d_csum_raw;
    if (update_table && ranked) {
        Span sp(st, 4, ls);
        // (the chunk sums are accumulated by seg_sum's workgroups: four atomic adds per chunk and genome)
        HIPCHK(hipMemsetAsync(d_csum_raw, 0, (size_t)((n_seg + 15) / 16) * c_pad * 4, ls));
        static const int ablate_rank = skx::knob("SKX_ABLATE_RANK") ? atoi(skx::knob("SKX_ABLATE_RANK")) : 0;  // measurement aid (results invalid): 1 = no counts at all
        if (ablate_rank >= 1) {
        } else if (two_level)
            skx::launch_chunk_sum(ls, sub_pair_q, sub_poff, sub_base, 0, n_reads, c_mq, c_pad, c_rows, c_grp_any, d_csum_raw, c_rowany, c_nq, c_spc);
        else
            skx::launch_seg_sum(ls, sub_pair_q, sub_poff, sub_base, 0, n_reads, skx::kSegLen, c_mq, c_pad, c_rows, d_inc, c_grp_any,
                                d_csum_raw, c_rowany, c_nq, c_spc);
        HIPCHK(hipGetLastError());
    }
    SKX_MARK("rank: seg_sum queued", si);
    static const int ablate_rank2 = skx::knob("SKX_ABLATE_RANK") ? atoi(skx::knob("SKX_ABLATE_RANK")) : 0;  // 2 = no ranking stage at all
    if (update_table && ranked && ablate_rank2 != 2) {
        Span sp(st, 4, ls);
        // (the chain's own sum of the table: nobody reads it when the pass's front half has computed the tables)
        u64* const cum_out = legacy ? st->d_tab[tab_next] : L.d_cum_sink;
        hipEvent_t ev_tab = legacy ? L.ev_cum : nullptr;
        // the top-1 kernel keeps (value relative to the leader) in 23 bits of a 32-bit key: at most 2 x 64 x s + 1 per
        // segment, so sketch sizes from 2^15 on take the 64-bit-key kernel (with k = 1) instead
        static const bool top1_wide_env = skx::knob("SKX_TOP1_WIDE") != nullptr;  // test knob: force the 64-bit-key kernel
        const bool top1_fast = st->top_k == 1 && ref->s_read < (1u << 15) && !top1_wide_env;
        static const bool live_env = !skx::knob("SKX_RANK_LIVE") || atoi(skx::knob("SKX_RANK_LIVE")) != 0;  // test knob
        const bool topk_fast = !top1_fast && st->top_k && st->top_k <= skx::rank_topk_fast_max();
        unsigned char* d_live = ((top1_fast || topk_fast) && live_env) ? L.d_live : nullptr;  // (the pruned kernels look at the flags)
        if (two_level) {
            skx::launch_seg_prefix(ls, d_inc, n_seg, c_pad, c_spc, cum_in, cum_out, L.d_rel, L.d_csum, d_csum_raw, prune_k,
                                   L.d_leader, L.d_lead_val, L.d_gmax, L.d_lpart_sum, L.d_lpart_idx, c_grp_any, d_live, L.d_lead_seg, 1, nullptr, ev_tab);
            skx::launch_seg_sum(ls, sub_pair_q, sub_poff, sub_base, 0, n_reads, skx::kSegLen, c_mq, c_pad, c_rows, d_inc, c_grp_any,
                                nullptr, c_rowany, c_nq, c_spc, L.d_gmax, L.d_lead_val);
            skx::launch_seg_prefix(ls, d_inc, n_seg, c_pad, c_spc, cum_in, cum_out, L.d_rel, L.d_csum, d_csum_raw, prune_k,
                                   L.d_leader, L.d_lead_val, L.d_gmax, L.d_lpart_sum, L.d_lpart_idx, c_grp_any, d_live, L.d_lead_seg, 2, L.d_live_ctr);
        } else {
            skx::launch_seg_prefix(ls, d_inc, n_seg, c_pad, c_spc, cum_in, cum_out, L.d_rel, L.d_csum, d_csum_raw, prune_k,
                                   L.d_leader, L.d_lead_val, L.d_gmax, L.d_lpart_sum, L.d_lpart_idx, c_grp_any, d_live, L.d_lead_seg, 0,
                                   L.d_live_ctr, ev_tab);
        }
        HIPCHK(hipGetLastError());
        SKX_MARK("rank: prefixes queued", si);
        // (the live sample reaches the host through a kernel that also re-arms the counters -- NOT hipMemcpyAsync / hipMemsetAsync;
        // a compact chain's sample says nothing about the full problem and is only re-armed)
        if (prune_k) skx::launch_store_host_words(ls, compact ? st->h_nq_sink : st->h_nq + 2, L.d_live_ctr, 2);
        st->rank_seq += 1;
        if (legacy) {
            // (what the batch's candidates would have been, for the passes to come: a hint, a few batches stale when it is read)
            // (two batches of a pass are asked: the middle one and the last -- the kernel is a lone workgroup that reads the table
            // twice per species, 150 us on a busy chip; a count from the second half of a sample's FIRST pass is what its second pass
            // goes by)
            st->tab_cur = tab_next; st->d_cum = cum_out; st->cum_writer = &L;
        }
        if (top1_fast) {
            skx::launch_rank_seg_top1(ls, sub_pair_q, sub_pair_r, sub_poff, sub_base, 0, n_reads, c_mq, c_pad, c_rows,
                                      c_spc, cum_in, L.d_rel, L.d_cand_sum, L.d_cand_idx, d_inc,
                                      L.d_leader, L.d_gmax, L.d_lead_val, c_grp_any, d_live, L.d_has, c_rowany, c_nq);
            skx::launch_top1_merge(ls, L.d_cand_sum, L.d_cand_idx, n_reads, d_topk_idx, d_topk_sum, out_r0, c_spc, L.d_has,
                                   (c_gw + skx::kRankWords - 1) / skx::kRankWords);
        } else if (st->top_k <= skx::rank_topk_fast_max()) {
            const u32 n_grp = (c_gw + skx::kRankWords - 1) / skx::kRankWords;
            skx::launch_rank_seg_topk(ls, sub_pair_q, sub_pair_r, sub_poff, sub_base, 0, n_reads, c_mq, c_pad, c_rows, c_spc,
                                      cum_in, L.d_rel, st->top_k, L.d_cand_sum, L.d_cand_idx, d_inc, L.d_leader,
                                      L.d_gmax, L.d_lead_val, c_grp_any, d_live, L.d_has);
            skx::launch_topk_merge(ls, L.d_cand_sum, L.d_cand_idx, n_reads, n_grp, 1, st->top_k, d_topk_idx, d_topk_sum, out_r0, c_spc,
                                   L.d_has);
        } else {
            skx::launch_rank_seg(ls, sub_pair_q, sub_pair_r, sub_poff, sub_base, 0, n_reads, skx::kSegLen, c_mq, c_pad, c_rows,
                                 c_spc, cum_in, L.d_rel, st->top_k, L.d_cand_sum, L.d_cand_idx, c_grp_any);
            skx::launch_topk_merge(ls, L.d_cand_sum, L.d_cand_idx, n_reads, c_gw, skx::kRankWords, st->top_k, d_topk_idx,
                                   d_topk_sum, out_r0, c_spc, nullptr);
        }
        if (legacy && (si == n_sub / 2 || si == n_sub - 1))  // (behind the batch's rows: nobody waits for the count)
            skx::launch_cand_count(ls, cum_in, cum_out, spc, st->top_k, st->h_lcount, ++st->lcount_seq);
        if (compact)  // candidate slots -> genome indices (local to the species)
            skx::launch_cand_rows_back(ls, d_topk_idx + (size_t)out_r0 * n_sp * st->top_k, n_reads, n_sp, st->top_k,
                                       ps.cand + (size_t)si * st->n_pad_c, skx::kCandCap, ref->d_sp_g0);
    }
    if (sb.d_shared)
        skx::launch_shared_debug(ls, d_pair_q + sb.p_off, sub_poff, sub_base, 0, n_reads, d_mq, nq_rows, ref->n_genomes, ref->d_real2pad, sb.d_shared, 0);
    HIPCHK(hipGetLastError());
    SKX_MARK("rank: end sub", si);
    }  // sub-passes
    // every lane this pass used joins hs2 (lane 0's stream), the stream everybody else who looks at the table or at this pass's
    // buffers is ordered on
    if (!lane_used[0]) HIPCHK(hipStreamWaitEvent(hs2, st->ev_front[b], 0));
    for (int li = 1; li < st->n_lanes; ++li)
        if (lane_used[li]) {
            HIPCHK(hipEventRecord(st->lane[li].ev_done, st->lane[li].s));
            HIPCHK(hipStreamWaitEvent(hs2, st->lane[li].ev_done, 0));
        }
    HIPCHK(hipEventRecord(st->ev_back[b], hs2));
    st->back_pending[b] = true;
    HIPCHK(hipEventRecord(st->ev_pslot[slot], hs2));
    st->pslot_pending[slot] = true;
    // host-fed batches: their rows go back to the host behind the ranking
    int rc = SKX_OK;
    for (int si = 0; si < n_sub; ++si) {
        const int r2 = staged_rows(st, pc.subs[si].slot);
        if (rc == SKX_OK && r2 != SKX_OK) rc = r2;
    }
    return rc;
}

static int run_pass(skx_stream* st, u32 ra, u32 rb, u32 p_base, u32 P, u32* d_topk_idx, u64* d_topk_sum,
                    u32* d_shared /* [rb-ra][n_genomes] or NULL */, bool update_table, bool inserted = false, u32 q_rows = 0xFFFFFFFFu,
                    void* slot = nullptr) {
    SubPass sb;
    sb.slot = slot;
    sb.ra = ra; sb.rb = rb; sb.p_off = 0; sb.P = P; sb.p_base = p_base; sb.d_topk_idx = d_topk_idx; sb.d_topk_sum = d_topk_sum;
    sb.d_shared = d_shared; sb.side = st->side;
    sb.d_poff = st->d_poff_pass[st->pslot];  // (inserted passes: the front half's copy; else run_pass_multi fills the slot itself)
    return run_pass_multi(st, &sb, 1, update_table, inserted, q_rows);
}

// partition [0, n_reads) into passes by the pair counts in h_poff; calls fn(ra, rb, p_base, P)
// (a pass cut here holds at most min(pcap, qcap) pairs, hence at most qcap distinct query hashes)
template <class F>
static int for_each_pass(skx_stream* st, u32 n_reads, u32 max_pass_reads, F fn) {
    const u32 cap = std::max<u32>(1u, std::min<u32>(st->rpass, max_pass_reads));
    const u32 pair_cap = std::min(st->pcap, st->qhash_cap());
    u32 ra = 0;
    while (ra < n_reads) {
        u32 rb = ra;
        while (rb < n_reads && rb - ra < cap && st->h_poff[rb + 1] - st->h_poff[ra] <= pair_cap) ++rb;
        if (rb == ra) return fail(SKX_ERR_CAPACITY, "read %u alone has %u candidate hashes > pass capacity %u", ra,
                                  st->h_poff[ra + 1] - st->h_poff[ra], pair_cap);
        SKXCHK(fn(ra, rb, st->h_poff[ra], st->h_poff[rb] - st->h_poff[ra]));
        ra = rb;
    }
    return SKX_OK;
}

// spin on the published sequence number (looking at the stream now and then so a fault cannot hang the caller)
static int wait_published(skx_stream* st, const PendingBatch& pb) {
    volatile u32* pub = st->h_chk_base + 16 * pb.side;
    hipStream_t hs = st->hs1;  // (the publish kernel's stream)
    for (u64 spins = 1; pub[15] != pb.seq; ++spins) {
        if ((spins & 0xFFF) == 0) {
            const hipError_t e = hipStreamQuery(hs);
            if (e == hipSuccess) {
                if (pub[15] == pb.seq) break;
                HIPCHK(hipStreamSynchronize(hs));
                if (pub[15] != pb.seq) return fail(SKX_ERR_HIP, "publish kernel finished without raising its sequence number");
            } else if (e != hipErrorNotReady) {
                return fail(SKX_ERR_HIP, "stream failed while waiting for the batch summary: %s", hipGetErrorString(e));
            }
        }
    }
    return SKX_OK;
}
static const u64* batch_filter(const skx_ref* ref) {
    static const bool no_filter = skx::knob("SKX_NO_FILTER") != nullptr;  // measurement aid
    return (ref->any && !no_filter) ? ref->d_filt : nullptr;
}
// counts -> (filter, for rows the sketchers did not filter themselves) -> pair offsets (poff[n_reads] = total pairs)
// -> speculative pair gather -> the few words the host needs, published to page-locked memory
// (the stream's d_sk ... names must be on the batch's side)
// everything of the batch queued on the sketch stream so far comes before what is queued on hs1 from here on
static int behind_the_sketch(skx_stream* st, int side) {
    if (st->hs1 == st->hs0) return SKX_OK;
    HIPCHK(hipEventRecord(st->ev_main[side], st->hs0));
    HIPCHK(hipStreamWaitEvent(st->hs1, st->ev_main[side], 0));
    return SKX_OK;
}
static int queue_counts_and_summary(skx_stream* st, PendingBatch& pb) {
    const skx_ref* ref = st->ref;
    hipStream_t hs = st->hs1;
    const u32 n_reads = pb.n_reads;
    const u64* filt = batch_filter(ref);
    SKXCHK(behind_the_sketch(st, pb.side));
    if (!ref->any) HIPCHK(hipMemsetAsync(st->d_cnt, 0, (size_t)n_reads * 4, hs));
    if (filt && !pb.inrange_only)
        skx::launch_filter_apply(hs, st->d_sk, st->sk_stride, st->d_cnt, n_reads, filt, ref->filt_shift);  // (rows mode)
    skx::launch_count_scan(hs, st->d_cnt, st->d_poff, n_reads + 1, st->d_bsum);
    // the whole batch is normally ONE pass: gather its pairs into the buffer set that pass will use and fill the set's
    // hash set right here, behind the sketcher -- the rest of the dictionary then runs on the scan stream and this
    // stream is free for the next batch's sketch (the kernel does nothing if the pairs do not fit one pass: the host
    // finds out after the wait and cuts the batch into passes)
    if (pb.spec_insert) {
        const int b = pb.spec_set, slot = pb.spec_slot;
        // (round 5: that ranking may not even be QUEUED yet -- chains wait for their pass's candidates: make sure it is, so that the
        // event below covers it.  Its pass was queued two groups ago: its candidates were published long since)
        for (int k = 0; k < st->pc_n; ++k)
            if (st->pcq[(st->pc_head + k) & 1].slot == slot) { SKXCHK(queue_chains_until(st, st->pc_n - k - 1)); break; }
        // the set's hash set / pair hashes were last read by the dictionary of the pass two back (scan stream), the slot's
        // pair lists by the ranking three passes back
        if (st->pairq_pending[b]) { HIPCHK(hipStreamWaitEvent(hs, st->ev_pairq[b], 0)); st->pairq_pending[b] = false; }
        if (st->pslot_pending[slot]) { HIPCHK(hipStreamWaitEvent(hs, st->ev_pslot[slot], 0)); st->pslot_pending[slot] = false; }
        // (gi > 0: the batch shares the pass of the gi batches before it -- same set, same slot, its pairs behind theirs, whose
        // counts sit at the end of those batches' pair offsets on the device; its own offsets go to the slot's region gi)
        skx::PairBase base;
        for (int i = 0; i < pb.gi; ++i) base.p[i] = st->sd_poff[pb.prev_side[i]] + pb.prev_reads[i];
        skx::launch_dict_insert(hs, st->d_sk, st->cur_stride, st->d_poff, 0, n_reads, 0, st->d_pair_h[b], st->d_pair_r[slot],
                                st->d_ht[b], st->ht_slots, st->d_dict_ctr[b], st->pcap, st->d_len, base);
        HIPCHK(hipMemcpyAsync(st->d_poff_pass[slot] + (size_t)pb.gi * ((size_t)st->rpass + 2), st->d_poff, ((size_t)n_reads + 1) * 4,
                              hipMemcpyDeviceToDevice, hs));
    }
    pb.seq = ++st->pub_seq;
    skx::launch_publish(hs, st->d_chk, st->d_retry, st->d_big, st->d_poff + n_reads, st->h_chk, pb.seq,
                        pb.spec_insert ? st->d_dict_ctr[pb.spec_set] : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(st->ev_sketch[pb.side], hs));
    return SKX_OK;
}

// the device work of a front half: offsets check (+ long-read tables), sketch, counts, speculative gather, published summary
// (the stream's names are on the batch's side and row mode; also used to REPEAT a batch whose rows did not fit the pool)
static int queue_front(skx_stream* st, PendingBatch& pb, int leave_room) {
    const skx_ref* ref = st->ref;
    hipStream_t hs = st->hs0;  // sketching and everything the host reads back run on the first pipeline stream
    const u32 n_reads = pb.n_reads;
    const u64 max_ref = ref->any ? ref->max_ref : 0;
    const u64* filt = batch_filter(ref);
    Span sp(st, 0);
    // offsets are looked at on the device (cheap); it also zeroes entry n_reads of the pair counts
    // (... and lists the long reads of a production batch with their segments: the sketcher splits those over waves)
    const skx::LongReads* lr = (pb.inrange_only && st->lr.list) ? &st->lr : nullptr;
    const skx::KmerFilter kf = ref->kmer_filter();
    skx::launch_batch_check(hs, pb.d_offsets, n_reads, pb.n_bases, st->d_chk, st->d_cnt + n_reads, lr);
    if (pb.h_sketches) HIPCHK(hipMemsetAsync(st->d_sk, 0, (size_t)n_reads * st->sk_stride * 8, hs));  // (rows mode)
    // every read, any length: wave sketchers (then the block sketcher for what overflowed: device-side lists)
    // production: the main kernel on the sketch stream, the list walks behind it (long-read merge, 2048-slot retry) on hs1
    // with the rest of the batch's front half -- the next batch's main kernel then follows this one directly
    for (int phase = 1; phase <= (pb.inrange_only ? 2 : 1); ++phase) {
        if (phase == 2) SKXCHK(behind_the_sketch(st, pb.side));
        HIPCHK(skx::launch_sketch(phase == 2 ? st->hs1 : hs, pb.d_bases, pb.d_offsets, n_reads, ref->k, ref->seed, ref->s_read, max_ref,
                                  pb.inrange_only, st->d_sk, st->cur_stride, st->d_len, st->d_cnt, filt, ref->filt_shift, st->d_retry,
                                  st->d_big, pb.n_bases, st->d_chk, leave_room, st->packed, lr, ref->d_kf ? &kf : nullptr,
                                  pb.inrange_only ? phase : 3, st->cur_pool_cap, st->cur_pool_fixed));
    }
    if (!pb.inrange_only) {
        // full sketches (debug outputs): reads with more k-mers than a wave holds are on the `big` list; this path is
        // not the fast one -- read the count back and run the block sketcher before the rows are copied out
        u32 n_big = 0;
        HIPCHK(hipMemcpyAsync(&n_big, st->d_big, 4, hipMemcpyDeviceToHost, hs));
        HIPCHK(hipStreamSynchronize(hs));
        HIPCHK(skx::launch_sketch_block(hs, pb.d_bases, pb.d_offsets, st->d_big, n_big, ref->k, ref->seed, ref->s_read, max_ref,
                                        false, st->d_sk, st->cur_stride, st->d_len, st->d_cnt, filt, ref->filt_shift, st->packed, st->d_chk,
                                        st->cur_pool_cap, st->cur_pool_fixed));
        st->reads_big += n_big;
        HIPCHK(hipMemsetAsync(st->d_big, 0, 4, hs));
    }
    // optional sketch outputs leave now: the filter compacts the rows in place
    if (pb.h_sketch_len) HIPCHK(hipMemcpyAsync(pb.h_sketch_len, st->d_len, (size_t)n_reads * 4, hipMemcpyDeviceToHost, hs));
    if (pb.h_sketches) {
        memset(pb.h_sketches, 0, (size_t)n_reads * ref->s_read * 8);
        HIPCHK(hipMemcpy2DAsync(pb.h_sketches, (size_t)ref->s_read * 8, st->d_sk, (size_t)st->sk_stride * 8,
                                (size_t)std::min(ref->s_read, st->sk_stride) * 8, n_reads, hipMemcpyDeviceToHost, hs));
    }
    return queue_counts_and_summary(st, pb);
}

// front half of a batch already resident on the device: everything up to the published summary, on the sketch stream.
// Only hashes some genome holds become pairs (exact: the others share nothing with anyone).  The wave kernel applies the
// filter itself in production mode; rows it did not filter (full sketches for the debug outputs) get the separate pass.
static int batch_front(skx_stream* st, PendingBatch& pb) {
    const skx_ref* ref = st->ref;
    hipStream_t hs = st->hs0;
    const u32 n_reads = pb.n_reads;
    // sides are taken in turn (two enqueued batches may be waiting for their shared pass while this one is sketched)
    // (synchronous pushes have flushed whatever was waiting: they alternate between two sides, so a stream that is only ever
    // pushed to never allocates the others)
    const int n_sides = pb.pairable ? (int)st->coalesce + 1 : 2;
    pb.side = st->side_next % n_sides;
    st->side_next = (pb.side + 1) % n_sides;
    HIPCHK(use_side(st, pb.side));
    pb.inrange_only = !(pb.h_sketches || pb.h_sketch_len);  // production: only what can meet the reference is built
    // production rows are reservations out of the side's pool; full sketches (and unfiltered ones) need full-width rows
    pb.rows_mode = !pb.inrange_only || batch_filter(ref) == nullptr;
    if (pb.rows_mode) HIPCHK(use_rows(st));
    // the per-read x per-genome debug matrix is produced in slabs of at most 256 MB
    pb.dbg_cap = pb.h_shared ? (u32)std::max<u64>(1, (256ull << 20) / ((u64)ref->n_genomes * 4)) : 0xFFFFFFFFu;
    static const bool spec_env = !skx::knob("SKX_SPEC_INSERT") || atoi(skx::knob("SKX_SPEC_INSERT")) != 0;  // test knob
    pb.spec_insert = spec_env && n_reads <= std::min(st->rpass, pb.dbg_cap);  // one pass unless the pairs turn out too many
    // Batches enqueued back to back SHARE a pass (run_pass_multi): the later ones' pairs are gathered into the first one's hash
    // set and pair lists.  Decided here, when a later one's front half is queued: the batches waiting must be one open group
    // with room left (stream_coalesce), speculative and production like this one, and the caller must have asked for it
    // (pb.pairable: the enqueue / submit entry points of a stream created with stream_coalesce >= 2).
    // (the FIRST group of a sample is kept to six batches: its pass ranks on everything -- nothing is known about the sample yet --, and
    // the sooner it has said what its candidates were, the sooner the passes behind it can rank compactly.  K = 20 batches: passes of
    // 6 + 8 + 6 instead of 8 + 8 + 4, the same three scans.  Measured, reads/s from a fresh table, first group of 4 / 6 / 8 batches:
    // truth-strain workload 104 / 102 / 97.5 M, ancestor workload -- no compact pass ever, but the shorter the LAST group, the
    // shorter the tail of ranking chains behind the last sketch -- 129 / 133 / 134.5 M.)
    static const u32 first_group_env = skx::knob("SKX_FIRST_GROUP") ? (u32)atoi(skx::knob("SKX_FIRST_GROUP")) : 6u;  // experiment knob
    // (... where a scan of the reference costs no more than the rankings it spares: the smaller first group can add one scan to the
    // sample -- K = 8 batches: 4 + 4 instead of 8.  Five species resident, 12 GB per scan: 36 M reads/s became 27 M.  Up to 6 GB.)
    const bool scan_is_cheap = (u64)ref->n_pad * ref->s * 8ull <= (6ull << 30);
    const u32 fresh_cap = (st->fresh_table && st->top_k && ref->max_species > skx::kCandCap && scan_is_cheap) ? std::max(1u, first_group_env) : 0xFFFFFFFFu;
    bool joins = pb.pairable && pb.spec_insert && pb.inrange_only && !pb.rows_mode && st->n_pend >= 1 &&
                 st->n_pend < (int)std::min(std::min(st->group_cap, pb.max_group), fresh_cap);
    if (joins && st->ppr_est > 0.0) {  // would the group still fit a pass?  (a group that does not is un-shared at a price: batch_back_group)
        u64 reads = pb.n_reads;
        for (int i = 0; i < st->n_pend; ++i) reads += st->pend[i].n_reads;
        // (distinct hashes: what the matrices can GROW to counts -- batch_back_group makes the room when the group closes)
        joins = reads * st->ppr_est * 1.25 <= (double)st->pcap && reads * st->qpr_est * 1.25 <= (double)(std::max(st->qcap, st->qcap_max) - st->sd64);
    }
    for (int i = 0; joins && i < st->n_pend; ++i) {
        const PendingBatch& q = st->pend[i];
        joins = q.pairable && q.gi == i && q.spec_insert && q.inrange_only && !q.rows_mode;
    }
    if (joins) {
        pb.gi = st->n_pend;
        pb.spec_set = st->pend[0].spec_set; pb.spec_slot = st->pend[0].spec_slot;
        for (int i = 0; i < pb.gi; ++i) { pb.prev_side[i] = st->pend[i].side; pb.prev_reads[i] = st->pend[i].n_reads; }
    } else {
        // (one pass is still to come for the batches waiting -- be it one batch or a group sharing it)
        pb.gi = 0;
        pb.spec_set = st->buf ^ (st->n_pend ? 1 : 0);
        pb.spec_slot = (st->pslot + (st->n_pend ? 1 : 0)) % 3;
    }
    // static dense dictionary: the scan of the pass this batch opens depends on no batch -- queue it NOW, ahead of the sketch: the
    // HBM-bound scan then runs beside the VALU-bound sketches of its own group (a lone batch: beside its own sketch) instead of behind
    // their dictionary
    // WHEN (policy of this build, experiment knob SKX_EARLY_SCAN): 1 = always here; 2 (default) = here only when no group is waiting for its
    // pass (the first group of a sample, a lone batch) -- else right BEHIND the waiting group's front half (enqueue_batch): the scan of
    // group k + 1 then runs beside that group's padded sketches in mid-stream instead of beside the unpadded first sketches of the sample;
    // 0 = never early (the pass launches it itself, as in round 5)
    static const int early_env = skx::knob("SKX_EARLY_SCAN") ? atoi(skx::knob("SKX_EARLY_SCAN")) : 2;
    bool early_scan = false;
    if (st->static_dense && pb.gi == 0 && ref->n_sd && !st->m_ready[pb.spec_set] && (early_env == 1 || (early_env == 2 && st->n_pend == 0))) {
        early_scan = !(st->reuse_m && st->m_filled[pb.spec_set]);  // (a scan really goes out beside this sketch)
        SKXCHK(queue_static_scan(st, pb.spec_set));
    }
    if (st->chk_dirty[pb.side]) {  // the batch that last used this set failed between arming and publishing: nothing re-armed its counters
        HIPCHK(hipMemsetAsync(st->d_chk, 0, (size_t)skx::chk_words() * 4, hs));  // (the whole block: the pool's bump counters sit behind word 16)
        HIPCHK(hipMemsetAsync(st->d_retry, 0, 4, hs));
        HIPCHK(hipMemsetAsync(st->d_big, 0, 4, hs));
    }
    st->chk_dirty[pb.side] = true;
    if (st->sk_reader_pending[pb.side]) {  // (rare: this side's previous batch was cut into passes that read it on the scan stream)
        HIPCHK(hipStreamWaitEvent(hs, st->ev_skread[pb.side], 0));
        st->sk_reader_pending[pb.side] = false;
    }
    // (three-stream pipeline: will the previous pass's scan be in flight?  then this sketch shares the CUs with it)
    // (2: the larger LDS pad -- fewer sketch waves per CU, room for the scan AND a heavy ranking chain beside it; 1: the smaller
    // one, once the ranking has become light: far from the start of a sample most (chunk, rank group)s are dead -- the share the
    // ranking samples for its own counting scheme -- and with a scan only every few batches the sketch is what should fill the
    // chip.  Measured at C2, eight batches per scan: 19 KB everywhere 120 M reads/s from a fresh table / 138 M steady; 11 KB
    // everywhere 118 / 150 M.)
    static const int room_env = skx::knob("SKX_ROOM_ADAPT") ? atoi(skx::knob("SKX_ROOM_ADAPT")) : 1;  // experiment knob
    static const u32 room_pct_env = skx::knob("SKX_ROOM_PCT") ? (u32)atoi(skx::knob("SKX_ROOM_PCT")) : 33u;  // experiment knob
    const bool ranking_light = room_env && st->h_nq[3] != 0 && (u64)st->h_nq[2] * 100 < (u64)st->h_nq[3] * room_pct_env;
    // (round 4: only when something WILL run beside this sketch.  The batches that join the FIRST group of a stream -- or the
    // first after a pause -- have nothing beside them: no pass is in flight and none is queued before their group closes; with the
    // pad they ran at half their occupancy alone on the chip, 0.49 instead of 0.40 ms each, seven times per fresh stream.
    // A batch that opens a new group while others wait gets the pad: this very call queues the waiting group's pass behind it.)
    bool scan_in_flight = false, pass_in_flight = false;
    if (st->depth >= 3)
        for (int i = 0; i < 2; ++i) {
            if (st->front_pending[i] && hipEventQuery(st->ev_front[i]) == hipErrorNotReady) scan_in_flight = pass_in_flight = true;
            if (st->back_pending[i] && hipEventQuery(st->ev_back[i]) == hipErrorNotReady) pass_in_flight = true;
        }
    (void)hipGetLastError();  // (hipErrorNotReady is not an error)
    static const int room_lazy_env = skx::knob("SKX_ROOM_LAZY") ? atoi(skx::knob("SKX_ROOM_LAZY")) : 1;  // experiment knob: 0 = round 3's rule
    int leave_room = 0;
    if (st->depth >= 3) {
        const bool opens_behind_a_group = st->n_pend && !joins;
        if (st->n_pend && (!room_lazy_env || opens_behind_a_group || pass_in_flight)) leave_room = ranking_light ? 1 : 2;
        else if (scan_in_flight) leave_room = 1;
        // (experiment knob SKX_EARLY_ROOM: the pad of a sketch whose pass's scan was queued just ahead of it; default 1 = the smaller pad)
        static const int early_room_env = skx::knob("SKX_EARLY_ROOM") ? atoi(skx::knob("SKX_EARLY_ROOM")) : 1;
        if (early_scan && leave_room < early_room_env) leave_room = early_room_env;
    }
    {
        SKX_T0();
        SKXCHK(queue_front(st, pb, leave_room));
        SKX_ACC(front);
    }
    st->chk_dirty[pb.side] = false;  // the publish kernel is queued: it re-arms the device-side counters
    pb.valid = true;
    return SKX_OK;
}

// a younger batch's speculative pair gather has to be undone: the batch before it needs more than one pass, so the
// buffer sets no longer alternate the way the gather assumed (rare)
static int cancel_speculation(skx_stream* st, PendingBatch& y, bool keep_flag = false) {
    if (!y.spec_insert) return SKX_OK;
    // (on the stream the gather was queued on: behind it)
    HIPCHK(hipMemsetAsync(st->d_ht[y.spec_set], 0xFF, (size_t)st->ht_slots * 8, st->hs1));
    HIPCHK(hipMemsetAsync(st->d_dict_ctr[y.spec_set], 0, 64, st->hs1));
    HIPCHK(hipStreamSynchronize(st->hs1));
    if (!keep_flag) y.spec_insert = false;
    return SKX_OK;
}

// back half: wait for the summary, then queue the passes.  `younger`: a batch whose front half is already queued.
static int batch_back(skx_stream* st, PendingBatch& pb, PendingBatch* younger) {
    const skx_ref* ref = st->ref;
    hipStream_t hs = st->hs0;
    const u32 n_reads = pb.n_reads;
    const u64 max_ref = ref->any ? ref->max_ref : 0;
    const u64* filt = batch_filter(ref);
    pb.valid = false;
    HIPCHK(use_side(st, pb.side));
    if (pb.rows_mode) HIPCHK(use_rows(st));
    // A refused batch must not leave its pairs behind: the speculative gather queued by the front half has already put the
    // batch's keys into the hash set of buffer set spec_set (it only looks at the pair capacity, not at the offsets), and
    // every later pass on that set assumes |Q| <= its own pair count.  Empty the set again before returning the error.
    auto refuse = [&](int code, const std::string& msg) -> int {
        (void)cancel_speculation(st, pb);
        return fail(code, "%s", msg.c_str());
    };
    // The published summary, and the two rare things it can ask for -- each followed by a new summary:
    //   * c[6] & 4: the batch's rows did not fit the side's pool (far denser than the 16 pairs per read it is sized for).  The
    //     summary says how many entries were asked for: allocate that, empty what the speculative gather made of the partial
    //     rows and run the whole front half again (same side, same buffer set: nothing else has looked at this batch yet);
    //   * c[7]: reads whose in-range hashes overflowed a wave's 2048 slots wait on the `big` list.  The block sketcher was not
    //     queued blindly (it needs a drained CU even to find the list empty): run it now, then counts, pair gather (into the
    //     same buffer set: the keys already there are a subset) and the summary once more.
    bool stats_done = false, outputs_synced = false;
    u64 big_counted = 0;
    for (int round = 0;; ++round) {
        if (round > 6) return fail(SKX_ERR_HIP, "internal: the batch's front half does not settle");
        {
            SKX_T0();
            SKXCHK(wait_published(st, pb));  // the one wait of a batch: 48 bytes, no copy, no stream synchronisation
            SKX_ACC(wait);
        }
        if ((pb.h_sketches || pb.h_sketch_len) && !outputs_synced) { HIPCHK(hipStreamSynchronize(hs)); outputs_synced = true; }  // (debug outputs: their copies must have landed)
        u32 c[12];
        for (int i = 0; i < 12; ++i) c[i] = st->h_chk[i];
        char buf[160];
        if (c[0]) {
            snprintf(buf, sizeof buf, "offsets not monotonic at read %u", 0xFFFFFFFFu - c[0]);
            return refuse(SKX_ERR_INVALID, buf);
        }
        if (c[6] & 2u) return refuse(SKX_ERR_INVALID, "the long reads of the batch do not fit the stream's tables (offsets not monotonic?)");
        if (c[6] & 1u) {
            snprintf(buf, sizeof buf, "a read lies outside the n_bases=%llu bytes given from offsets[0] on", (unsigned long long)pb.n_bases);
            return refuse(SKX_ERR_INVALID, buf);
        }
        if (!stats_done && pb.inrange_only && st->lr.list && (c[1] || c[9])) { st->reads_split += c[1]; st->segs_split += c[9]; stats_done = true; }
        if (c[6] & 4u) {
            if (pb.rows_mode) return fail(SKX_ERR_HIP, "internal: pool overflow reported for full-width rows");
            const u64 res_now = st->pool_cap[pb.side] - st->pool_fixed;
            const u64 res_need = std::max<u64>(c[11], res_now + 1);  // (c[11]: 64 x the fullest sub-pool's request)
            const u64 need = st->pool_fixed + res_need + res_need / 8 + 1024;
            if (need > 0xFFFFFF00ull) return refuse(SKX_ERR_CAPACITY, "the batch's sketch rows exceed what the pool can index");
            SKXCHK(cancel_speculation(st, pb, true));
            HIPCHK(hipStreamSynchronize(st->hs0));
            HIPCHK(hipStreamSynchronize(st->hs1));
            const u64 cap = need;
            u64* grown = nullptr;
            HIPCHK(hipMalloc(&grown, (size_t)cap * 8));  // (the new pool first: a failure leaves the set as it was)
            (void)hipFree(st->sd_sk[pb.side]);
            st->sd_sk[pb.side] = grown;
            st->pool_cap[pb.side] = cap;
            st->pool_grown += 1;
            st->reads_big -= big_counted; big_counted = 0;
            HIPCHK(use_side(st, pb.side));
            SKXCHK(queue_front(st, pb, 0));
            continue;
        }
        if (c[7]) {
            st->reads_big += c[7]; big_counted += c[7];
            HIPCHK(skx::launch_sketch_block(hs, pb.d_bases, pb.d_offsets, st->d_big, c[7], ref->k, ref->seed, ref->s_read, max_ref,
                                            pb.inrange_only, st->d_sk, st->cur_stride, st->d_len, st->d_cnt, filt, ref->filt_shift,
                                            st->packed, st->d_chk, st->cur_pool_cap, st->cur_pool_fixed));
            SKXCHK(queue_counts_and_summary(st, pb));
            continue;
        }
        break;
    }
    const u32 total_pairs = st->h_chk[8];
    st->last_pairs = total_pairs; st->last_passes = 0;


    // one pass: the reads fit, the pairs fit, and the distinct hashes fit the bit matrices -- their number is known when the
    // speculative gather ran (it counts its new keys; it did nothing when the pairs exceed a pass), else bounded by the pairs
    const u32 spec_keys = pb.spec_insert ? st->h_chk[10] : 0xFFFFFFFFu;
    const u32 q_rows = spec_keys != 0xFFFFFFFFu ? spec_keys : total_pairs;
    if (spec_keys != 0xFFFFFFFFu && q_rows > st->qhash_cap() && total_pairs <= st->pcap) {
        // more distinct hashes than the pass's matrices have rows: make room for a group of such batches (see grow_query_rows)
        // (enqueued batches share passes: room for a group of them; a stream that is pushed to needs one batch's worth)
        const u64 per_batch = (u64)q_rows + q_rows / 4;
        SKXCHK(grow_query_rows(st, pb.pairable ? std::max<u64>(per_batch, per_batch * std::min<u32>(st->coalesce, 8u) * 3 / 4) : per_batch));
    }
    const bool single = n_reads <= std::min(st->rpass, pb.dbg_cap) && total_pairs <= st->pcap && q_rows <= st->qhash_cap();
#ifdef SKX_EXPERIMENTS
    if (skx::knob("SKX_DEBUG_PASS"))
        fprintf(stderr, "[skx pass] reads %u rpass %u dbg_cap %u pairs %u pcap %u spec_insert %d spec_keys %u q_rows %u qcap %u single %d gi %d\n",
                n_reads, st->rpass, pb.dbg_cap, total_pairs, st->pcap, (int)pb.spec_insert, spec_keys, q_rows, st->qcap, (int)single, pb.gi);
#endif
    if (n_reads && pb.inrange_only) {  // (what the next groups are sized by)
        st->ppr_est = (double)total_pairs / n_reads;
        st->qpr_est = (double)std::min(q_rows, total_pairs) / n_reads;
    }
    // (the batch needs several passes although a speculative gather was queued for it: normally that gather did nothing -- it
    // saw more pairs than a pass holds -- but after the block-sketcher redo above the FIRST gather may have fitted while the
    // recount does not; either way the set must be empty before the passes insert their own pairs)
    if (!single && pb.spec_insert) SKXCHK(cancel_speculation(st, pb));
    bool inserted = pb.spec_insert && single;  // the gather queued by the front half did its work
    if (!single && younger) SKXCHK(cancel_speculation(st, *younger));
    if (inserted && (st->buf != pb.spec_set || st->pslot != pb.spec_slot)) return fail(SKX_ERR_HIP, "internal: buffer sets out of step");
    u32* d_shared = nullptr;
    auto one_pass = [&](u32 ra, u32 rb, u32 p_base, u32 P) -> int {
        if (pb.h_shared) {
            if (d_shared) { (void)hipFree(d_shared); d_shared = nullptr; }
            HIPCHK(hipMalloc(&d_shared, (size_t)(rb - ra) * ref->n_genomes * 4));
        }
        SKXCHK(run_pass(st, ra, rb, p_base, P, pb.d_topk_idx, pb.d_topk_sum, d_shared, true, inserted, inserted ? q_rows : 0xFFFFFFFFu,
                        rb == n_reads ? pb.slot : nullptr));
        inserted = false;
        st->last_passes += 1;
        if (pb.h_shared) {
            SKXCHK(queue_chains(st, true));  // (the debug matrix comes out of the pass's ranking chain)
            HIPCHK(hipMemcpyAsync(pb.h_shared + (size_t)ra * ref->n_genomes, d_shared, (size_t)(rb - ra) * ref->n_genomes * 4,
                                  hipMemcpyDeviceToHost, st->hs2));
            HIPCHK(hipStreamSynchronize(st->hs2));
        }
        return SKX_OK;
    };
    int pass_rc;
    SKX_T0();
    if (single) {
        pass_rc = one_pass(0, n_reads, 0, total_pairs);  // the whole batch is one pass: no per-read offsets needed
    } else {
        HIPCHK(hipMemcpyAsync(st->h_poff, st->d_poff, ((size_t)n_reads + 1) * 4, hipMemcpyDeviceToHost, st->hs1));  // (the offsets' stream)
        HIPCHK(hipStreamSynchronize(st->hs1));
        pass_rc = for_each_pass(st, n_reads, pb.dbg_cap, one_pass);
    }
    SKX_ACC(back);
#ifdef SKX_EXPERIMENTS
    g_ht.n += 1;
    if (skx::knob("SKX_HOST_TIMES") && (g_ht.n % 64) == 0)
        fprintf(stderr, "[skx host times] per batch over %llu: front %.1f us, wait for summary %.1f us, queue passes %.1f us\n",
                (unsigned long long)g_ht.n, g_ht.front / g_ht.n, g_ht.wait / g_ht.n, g_ht.back / g_ht.n);
#endif
    if (d_shared) (void)hipFree(d_shared);
    SKXCHK(pass_rc);
    st->reads_total += n_reads;
    return SKX_OK;
}

// back half of a GROUP of enqueued batches that share a pass (the first one's front half opened hash set and pair lists, the
// others' appended to them): their summaries, one dictionary / scan / transpose, a ranking each.  Anything out of the ordinary
// in a summary -- an error, a row pool that was too small, reads for the block sketcher, more pairs or distinct hashes together
// than a pass holds -- un-shares them: the joint set is emptied and each batch takes its own pass(es), gathering its pairs on
// the scan stream.
// n_done: how many of the batches (in order) were processed completely -- their rows exist and their table updates are
// applied even when a later batch of the group makes the call fail
static int batch_back_group(skx_stream* st, PendingBatch* g, int n, PendingBatch* younger, int* n_done) {
    *n_done = 0;
    if (n == 1) {
        SKXCHK(batch_back(st, g[0], younger));
        *n_done = 1;
        return SKX_OK;
    }
    {
        SKX_T0();
        for (int i = 0; i < n; ++i) SKXCHK(wait_published(st, g[i]));
        SKX_ACC(wait);
    }
    bool fits = st->buf == g[0].spec_set && st->pslot == g[0].spec_slot;
    u64 pairs = 0;
    u32 P[kGroupMax];
    for (int i = 0; i < n; ++i) {
        const volatile u32* c = st->h_chk_base + 16 * g[i].side;
        fits = fits && !(c[0] | c[6] | c[7]);
        P[i] = c[8];
        pairs += P[i];
    }
    const u32 q_rows = (st->h_chk_base + 16 * g[n - 1].side)[10];  // (the last summary counts the keys of the joint set)
    const bool clean = fits;
    if (clean && q_rows != 0xFFFFFFFFu && q_rows > st->qhash_cap() && pairs <= st->pcap)
        // (the joint set of the group, scaled to a full group when this one was smaller: see grow_query_rows)
        SKXCHK(grow_query_rows(st, ((u64)q_rows + q_rows / 4) * std::max<u32>(1u, (st->coalesce + (u32)n - 1) / (u32)n)));
    fits = fits && pairs <= st->pcap && q_rows != 0xFFFFFFFFu && q_rows <= st->qhash_cap();
    u64 reads = 0;
    for (int i = 0; i < n; ++i) reads += g[i].n_reads;
    if (clean && q_rows != 0xFFFFFFFFu && reads) {
        // (distinct hashes grow less than linearly with the reads: the per-read figure of a group overestimates a larger one's)
        st->ppr_est = (double)pairs / reads;
        st->qpr_est = (double)q_rows / reads;
    }
    if (clean && !fits) {  // too much for one pass: smaller groups from here on
        u64 acc = 0;
        int m = 0;
        while (m < n && acc + P[m] <= st->pcap / 10 * 9 && (double)(acc + P[m]) / std::max<u64>(pairs, 1) * q_rows <= st->qhash_cap() / 10 * 9) acc += P[m++];
        st->group_cap = (u32)std::max(1, std::min(m, n - 1));
        st->clean_groups = 0;
    } else if (fits && ++st->clean_groups >= 64 && st->group_cap < st->coalesce) {
        st->group_cap += 1;
        st->clean_groups = 0;
    }
    if (!fits) {
        st->groups_unshared += 1;
        HIPCHK(hipMemsetAsync(st->d_ht[g[0].spec_set], 0xFF, (size_t)st->ht_slots * 8, st->hs1));
        HIPCHK(hipMemsetAsync(st->d_dict_ctr[g[0].spec_set], 0, 64, st->hs1));
        HIPCHK(hipStreamSynchronize(st->hs1));
        for (int i = 0; i < n; ++i) { g[i].spec_insert = false; g[i].gi = 0; }
        if (younger) SKXCHK(cancel_speculation(st, *younger));  // (it counted on ONE pass ahead of its own)
        for (int i = 0; i < n; ++i) {
            SKXCHK(batch_back(st, g[i], nullptr));
            *n_done = i + 1;
        }
        return SKX_OK;
    }
    SubPass subs[kGroupMax];
    u32 p_off = 0;
    for (int i = 0; i < n; ++i) {
        const volatile u32* c = st->h_chk_base + 16 * g[i].side;
        g[i].valid = false;
        if (st->sd_lr[g[i].side].list) { st->reads_split += c[1]; st->segs_split += c[9]; }
        subs[i].ra = 0; subs[i].rb = g[i].n_reads; subs[i].p_base = 0; subs[i].side = g[i].side;
        subs[i].d_topk_idx = g[i].d_topk_idx; subs[i].d_topk_sum = g[i].d_topk_sum;
        subs[i].p_off = p_off; subs[i].P = P[i];
        subs[i].d_poff = st->d_poff_pass[g[0].spec_slot] + (size_t)i * ((size_t)st->rpass + 2);
        subs[i].slot = g[i].slot;
        p_off += P[i];
    }
    st->last_pairs = pairs; st->last_passes = 1; st->shared_passes += 1;
    {
        SKX_T0();
        SKXCHK(run_pass_multi(st, subs, n, true, true, q_rows));
        SKX_ACC(back);
    }
    st->reads_total += reads;
    *n_done = n;
    return SKX_OK;
}

// the back half of the enqueued batch(es), if there are any (every entry point that looks at the stream's state starts here)
static int staged_rows(skx_stream* st, void* slot);
static void staged_drop(void* slot);
static int pending_back(skx_stream* st, PendingBatch* younger) {
    PendingBatch g[kGroupMax];
    const int n = st->n_pend;
    for (int i = 0; i < n; ++i) g[i] = st->pend[i];
    st->n_pend = 0;
    if (n == 0) return SKX_OK;
    // (a group that went batch by batch may fail at batch i: batches 0 .. i - 1 are scored, their rows must still reach the
    // host; the slots of the others are marked so that skx_stream_wait reports the loss instead of returning stale rows)
    int done = 0;
    st->tail_pass = younger == nullptr;  // (a flush: nothing is sketched beside the group's pass)
    int rc = batch_back_group(st, g, n, younger, &done);
    st->tail_pass = false;
    const std::string msg = rc != SKX_OK ? g_err : std::string();
    // (host-fed batches: the rows of the scored ones go back to the host behind their ranking chains -- queue_chains)
    for (int i = done; i < n; ++i) staged_drop(g[i].slot);
    if (!msg.empty()) g_err = msg;
    return rc;
}
static int flush_pending(skx_stream* st) {
    const int rc = pending_back(st, nullptr);
    const std::string msg = rc != SKX_OK ? g_err : std::string();
    const int rc2 = queue_chains(st, true);  // (the ranking of the last pass: waits for its candidates)
    if (rc != SKX_OK) { g_err = msg; return rc; }
    return rc2;
}
// sketch + score + rank a batch already resident on the device, both halves (synchronous entry points).
// h_shared / h_sketches / h_sketch_len: optional HOST outputs (parity/debug).
static int process_batch(skx_stream* st, const uint8_t* d_bases, const u64* d_offsets, u32 n_reads, u64 n_bases, u32* d_topk_idx,
                         u64* d_topk_sum, u32* h_shared, u64* h_sketches, u32* h_sketch_len) {
    if (n_reads == 0) return SKX_OK;
    SKXCHK(flush_pending(st));
    PendingBatch pb;
    pb.d_bases = d_bases; pb.d_offsets = d_offsets; pb.n_reads = n_reads; pb.n_bases = n_bases;
    pb.d_topk_idx = d_topk_idx; pb.d_topk_sum = d_topk_sum;
    pb.h_shared = h_shared; pb.h_sketches = h_sketches; pb.h_sketch_len = h_sketch_len;
    SKXCHK(batch_front(st, pb));
    st->tail_pass = true;  // (a synchronous push: the batch's passes have the chip to themselves)
    const int rc = batch_back(st, pb, nullptr);
    st->tail_pass = false;
    return rc;
}
// ... and with the halves of consecutive batches interleaved: front(i + 1), then back(i) -- or, when batches share passes,
// front(i + 1) alone while the group of batch i has room left (its batches wait for their partners); the call that opens the
// next group runs the shared back half of the one before it, behind its own front half.  Errors of a batch surface here up to
// stream_coalesce calls late (or in the flush); the batches enqueued after it up to that call are dropped too.
static int enqueue_batch(skx_stream* st, const uint8_t* d_bases, const u64* d_offsets, u32 n_reads, u64 n_bases, u32* d_topk_idx,
                         u64* d_topk_sum, void* slot) {
    PendingBatch nw;
    nw.d_bases = d_bases; nw.d_offsets = d_offsets; nw.n_reads = n_reads; nw.n_bases = n_bases;
    nw.d_topk_idx = d_topk_idx; nw.d_topk_sum = d_topk_sum; nw.slot = slot;
    nw.pairable = st->coalesce >= 2;
    // (host-fed batches: every batch waiting for its group holds a staging slot -- bases, offsets, rows -- so groups stay at four)
    nw.max_group = slot ? std::min<u32>(st->coalesce, kStagedGroupMax) : st->coalesce;
    SKX_MARK("enqueue: begin", st->n_pend);
    SKXCHK(queue_chains(st, false));  // (the latest pass's ranking, if its candidates have been published by now)
    {
        const int rc_front = batch_front(st, nw);
        if (rc_front != SKX_OK) { staged_drop(slot); return rc_front; }  // (skx_stream_wait on its ticket then fails instead of handing out stale rows)
    }
    SKX_MARK("enqueue: front queued, gi", nw.gi);
    int rc = SKX_OK;
    const bool had_pending = st->n_pend > 0;
    if (nw.gi == 0) rc = pending_back(st, &nw);  // (it joined nobody: the batches waiting are complete)
    if (rc == SKX_OK && nw.gi == 0 && had_pending && st->static_dense && st->ref->n_sd) {
        // (the new group's scan: behind the front half of the group that has just been queued -- see batch_front)
        static const int early_env = skx::knob("SKX_EARLY_SCAN") ? atoi(skx::knob("SKX_EARLY_SCAN")) : 2;
        if (early_env == 2 && !st->m_ready[nw.spec_set]) rc = queue_static_scan(st, nw.spec_set);
    }
    if (rc != SKX_OK) {
        const std::string msg = g_err;
        (void)cancel_speculation(st, nw);
        staged_drop(slot);
        (void)hipStreamSynchronize(st->hs0);
        (void)hipStreamSynchronize(st->hs1);
        g_err = msg + " (a batch enqueued earlier; the batches enqueued after it were dropped too)";
        return rc;
    }
    st->pend[st->n_pend++] = nw;
    SKX_MARK("enqueue: end", st->n_pend);
    return SKX_OK;
}

SKX_API int skx_stream_push(skx_stream* st, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                            uint32_t* topk_idx, uint64_t* topk_sum, uint32_t* per_read_shared, uint64_t* sketches,
                            uint32_t* sketch_len) {
    if (!st || !offsets) return fail(SKX_ERR_INVALID, "NULL argument");
    if (n_reads == 0) return SKX_OK;
    if (n_reads > st->max_reads) return fail(SKX_ERR_CAPACITY, "n_reads=%u exceeds max_batch_reads=%u", n_reads, st->max_reads);
    for (u32 r = 0; r < n_reads; ++r)
        if (offsets[r + 1] < offsets[r]) return fail(SKX_ERR_INVALID, "offsets not monotonic at read %u", r);
    const u64 base0 = offsets[0], n_bases = offsets[n_reads] - base0;
    if (n_bases > st->max_bases) return fail(SKX_ERR_CAPACITY, "batch has %llu bases > max_batch_bases=%llu",
                                             (unsigned long long)n_bases, (unsigned long long)st->max_bases);
    if (n_bases && !bases) return fail(SKX_ERR_INVALID, "bases is NULL");
    if ((topk_idx || topk_sum) && st->top_k == 0) return fail(SKX_ERR_INVALID, "stream was created with top_k=0");
    SKXCHK(use_device(st->device));
    SKXCHK(flush_pending(st));
    hipStream_t hs = st->hs0;
    // rebase offsets to 0 on the way in (packed input: to the first byte that holds a base of the batch)
    const u64 byte0 = st->packed ? base0 >> 1 : base0, rebase = st->packed ? byte0 * 2 : base0;
    const u64 n_bytes = st->packed ? ((offsets[n_reads] + 1) >> 1) - byte0 : n_bases;
    for (u32 r = 0; r <= n_reads; ++r) st->h_offsets[r] = offsets[r] - rebase;
    HIPCHK(hipMemcpyAsync(st->d_offsets, st->h_offsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, hs));
    if (n_bases) HIPCHK(hipMemcpyAsync(st->d_bases, bases + byte0, n_bytes, hipMemcpyHostToDevice, hs));
    SKXCHK(process_batch(st, st->d_bases, st->d_offsets, n_reads, n_bases, st->d_topk_idx, st->d_topk_sum,
                         per_read_shared, reinterpret_cast<u64*>(sketches), sketch_len));
    SKXCHK(queue_chains(st, true));    // the ranking of the batch's last pass (waits for its candidates)
    HIPCHK(hipStreamSynchronize(hs));  // sketch copies (first stream)
    const size_t rows = (size_t)n_reads * st->ref->n_species * st->top_k;
    if (topk_idx) HIPCHK(hipMemcpyAsync(topk_idx, st->d_topk_idx, rows * 4, hipMemcpyDeviceToHost, st->hs2));
    if (topk_sum) HIPCHK(hipMemcpyAsync(topk_sum, st->d_topk_sum, rows * 8, hipMemcpyDeviceToHost, st->hs2));
    HIPCHK(hipStreamSynchronize(st->hs2));  // the rows are written by the back half
    if (st->profiling) collect_spans(st);
    return SKX_OK;
}

static int check_device_batch(skx_stream* st, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases,
                              const uint32_t* d_topk_idx, const uint64_t* d_topk_sum) {
    if (!st || !d_offsets) return fail(SKX_ERR_INVALID, "NULL argument");
    if (n_reads > st->max_reads) return fail(SKX_ERR_CAPACITY, "n_reads=%u exceeds max_batch_reads=%u", n_reads, st->max_reads);
    if (n_reads && n_bases && !d_bases) return fail(SKX_ERR_INVALID, "d_bases is NULL");
    if ((d_topk_idx || d_topk_sum) && st->top_k == 0) return fail(SKX_ERR_INVALID, "stream was created with top_k=0");
    if (n_bases > st->max_bases) return fail(SKX_ERR_CAPACITY, "batch has %llu bases > max_batch_bases=%llu",
                                             (unsigned long long)n_bases, (unsigned long long)st->max_bases);
    return SKX_OK;
}
SKX_API int skx_stream_push_device(skx_stream* st, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads,
                                   uint64_t n_bases, uint32_t* d_topk_idx, uint64_t* d_topk_sum) {
    SKXCHK(check_device_batch(st, d_bases, d_offsets, n_reads, n_bases, d_topk_idx, d_topk_sum));
    SKXCHK(use_device(st->device));
    if (n_reads == 0) return flush_pending(st);
    u32* ti = d_topk_idx ? d_topk_idx : st->d_topk_idx;
    u64* ts = d_topk_sum ? reinterpret_cast<u64*>(d_topk_sum) : st->d_topk_sum;
    return process_batch(st, d_bases, reinterpret_cast<const u64*>(d_offsets), n_reads, n_bases, ti, ts, nullptr, nullptr, nullptr);
}
SKX_API int skx_stream_enqueue_device(skx_stream* st, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads,
                                      uint64_t n_bases, uint32_t* d_topk_idx, uint64_t* d_topk_sum) {
    SKXCHK(check_device_batch(st, d_bases, d_offsets, n_reads, n_bases, d_topk_idx, d_topk_sum));
    if (n_reads == 0) return SKX_OK;
    SKXCHK(use_device(st->device));
    u32* ti = d_topk_idx ? d_topk_idx : st->d_topk_idx;
    u64* ts = d_topk_sum ? reinterpret_cast<u64*>(d_topk_sum) : st->d_topk_sum;
    return enqueue_batch(st, d_bases, reinterpret_cast<const u64*>(d_offsets), n_reads, n_bases, ti, ts, nullptr);
}
SKX_API int skx_stream_flush(skx_stream* st) {
    if (!st) return fail(SKX_ERR_INVALID, "NULL stream");
    SKXCHK(use_device(st->device));
    return flush_pending(st);
}

// ---- host-fed pipeline
// rows travel back behind the pass's ranking (same stream), before the next batch's ranking overwrites them
static int staged_rows(skx_stream* st, void* slot) {
    if (!slot) return SKX_OK;
    skx_stream::Staged& sl = *static_cast<skx_stream::Staged*>(slot);
    const size_t rows = (size_t)sl.n_reads * st->ref->n_species * st->top_k;
    if (sl.out_idx) HIPCHK(hipMemcpyAsync(sl.out_idx, sl.d_rows_idx, rows * 4, hipMemcpyDeviceToHost, st->hs2));
    if (sl.out_sum) HIPCHK(hipMemcpyAsync(sl.out_sum, sl.d_rows_sum, rows * 8, hipMemcpyDeviceToHost, st->hs2));
    HIPCHK(hipEventRecord(sl.ev_done, st->hs2));
    sl.in_flight = true;
    return SKX_OK;
}
static void staged_drop(void* slot) {
    if (slot) static_cast<skx_stream::Staged*>(slot)->dropped = true;
}
// front half of the slot's batch (and the back half of the one before it)
static int staged_process(skx_stream* st, skx_stream::Staged& sl) {
    if (!sl.pending) return SKX_OK;
    sl.pending = false;
    HIPCHK(hipStreamWaitEvent(st->hs0, sl.ev_copy, 0));  // the batch must have landed before the sketcher reads it
    return enqueue_batch(st, sl.d_bases, sl.d_offsets, sl.n_reads, sl.n_bases, sl.d_rows_idx, sl.d_rows_sum, &sl);
}
// ... until its rows are on the host
static int staged_finish(skx_stream* st, skx_stream::Staged& sl) {
    SKXCHK(staged_process(st, sl));
    for (int i = 0; i < st->n_pend; ++i)
        if (st->pend[i].slot == &sl) { SKXCHK(flush_pending(st)); break; }
    for (int k = 0; k < st->pc_n; ++k) {  // (its pass is queued, its ranking -- and the copy of its rows -- not yet)
        const skx_stream::PassChains& pc = st->pcq[(st->pc_head + k) & 1];
        bool mine = false;
        for (int i = 0; i < pc.n_sub; ++i) mine = mine || pc.subs[i].slot == &sl;
        if (mine) { SKXCHK(queue_chains_until(st, st->pc_n - k - 1)); break; }
    }
    if (sl.in_flight) { HIPCHK(hipEventSynchronize(sl.ev_done)); sl.in_flight = false; }
    return SKX_OK;
}
SKX_API int skx_stream_submit(skx_stream* st, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                              uint32_t* topk_idx, uint64_t* topk_sum, uint64_t* ticket) {
    if (!st || !offsets) return fail(SKX_ERR_INVALID, "NULL argument");
    if (n_reads == 0) return fail(SKX_ERR_INVALID, "empty batch");
    if (n_reads > st->max_reads) return fail(SKX_ERR_CAPACITY, "n_reads=%u exceeds max_batch_reads=%u", n_reads, st->max_reads);
    for (u32 r = 0; r < n_reads; ++r)
        if (offsets[r + 1] < offsets[r]) return fail(SKX_ERR_INVALID, "offsets not monotonic at read %u", r);
    const u64 base0 = offsets[0], n_bases = offsets[n_reads] - base0;
    if (n_bases > st->max_bases) return fail(SKX_ERR_CAPACITY, "batch has %llu bases > max_batch_bases=%llu",
                                             (unsigned long long)n_bases, (unsigned long long)st->max_bases);
    if (n_bases && !bases) return fail(SKX_ERR_INVALID, "bases is NULL");
    if ((topk_idx || topk_sum) && st->top_k == 0) return fail(SKX_ERR_INVALID, "stream was created with top_k=0");
    SKXCHK(use_device(st->device));
    if (!st->hs_copy) {  // first use: copy stream + the staging slots
        HIPCHK(hipStreamCreateWithFlags(&st->hs_copy, hipStreamNonBlocking));
        st->n_slots = st->coalesce >= 2 ? 2 * std::min<u32>(st->coalesce, kStagedGroupMax) + 1 : 3;
        for (u32 si = 0; si < st->n_slots; ++si) {
            auto& sl = st->slot[si];
            HIPCHK(hipMalloc(&sl.d_bases, std::max<u64>(st->max_bases, 1)));
            HIPCHK(hipMalloc(&sl.d_offsets, ((size_t)st->max_reads + 1) * 8));
            if (st->top_k) {
                HIPCHK(hipMalloc(&sl.d_rows_idx, (size_t)st->max_reads * st->ref->n_species * st->top_k * 4));
                HIPCHK(hipMalloc(&sl.d_rows_sum, (size_t)st->max_reads * st->ref->n_species * st->top_k * 8));
            }
            HIPCHK(hipHostMalloc((void**)&sl.h_offsets, ((size_t)st->max_reads + 1) * 8, hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&sl.ev_copy, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
        }
    }
    const u32 ns = st->n_slots;
    skx_stream::Staged& sl = st->slot[st->next_ticket % ns];
    skx_stream::Staged& other = st->slot[(st->next_ticket + ns - 1) % ns];  // (the previous ticket's)
    // Order (round 4): (1) this slot's previous batch (n_slots tickets ago) must be through -- passes queued, rows on the host;
    // (2) the copy of THIS batch starts; (3) the previous batch goes to the kernels (its sketch is queued -- and, when it completes
    // a group, the host waits for the group's summaries before it can queue their pass).  Round 3 did (3) first: the copy
    // engine stood still whenever the host waited in (3), and a stream of 32 768-read batches took 1.4 ms per batch where its
    // copies take 0.45 ms.
    SKXCHK(staged_finish(st, sl));
    const u64 byte0 = st->packed ? base0 >> 1 : base0, rebase = st->packed ? byte0 * 2 : base0;
    const u64 n_bytes = st->packed ? ((offsets[n_reads] + 1) >> 1) - byte0 : n_bases;
    for (u32 r = 0; r <= n_reads; ++r) sl.h_offsets[r] = offsets[r] - rebase;
    HIPCHK(hipMemcpyAsync(sl.d_offsets, sl.h_offsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, st->hs_copy));
    if (n_bases) HIPCHK(hipMemcpyAsync(sl.d_bases, bases + byte0, n_bytes, hipMemcpyHostToDevice, st->hs_copy));
    HIPCHK(hipEventRecord(sl.ev_copy, st->hs_copy));
    sl.pending = true; sl.dropped = false; sl.n_reads = n_reads; sl.n_bases = n_bases; sl.out_idx = topk_idx; sl.out_sum = reinterpret_cast<u64*>(topk_sum);
    sl.ticket = st->next_ticket;
    if (ticket) *ticket = st->next_ticket;
    st->next_ticket += 1;
    if (&other != &sl) {
        const int rc = staged_process(st, other);
        if (rc != SKX_OK) {  // (a batch submitted before this one failed: this one is dropped with it, as the header says)
            const std::string msg = g_err;
            (void)hipEventSynchronize(other.ev_copy);
            (void)hipEventSynchronize(sl.ev_copy);
            sl.pending = false; sl.dropped = true;
            g_err = msg;
            return rc;
        }
        // (the previous batch's host buffers are the caller's again when this call returns)
        if (other.ev_copy && other.ticket + 2 == st->next_ticket) HIPCHK(hipEventSynchronize(other.ev_copy));
    }
    return SKX_OK;
}
SKX_API int skx_stream_wait(skx_stream* st, uint64_t ticket) {
    if (!st) return fail(SKX_ERR_INVALID, "NULL stream");
    if (ticket >= st->next_ticket) return fail(SKX_ERR_INVALID, "ticket %llu was never issued", (unsigned long long)ticket);
    SKXCHK(use_device(st->device));
    const u32 ns = st->n_slots;
    skx_stream::Staged& sl = st->slot[ticket % ns];
    if (sl.ticket != ticket) return SKX_OK;  // the slot has moved on: that batch completed before it was reused
    for (u64 t = ticket >= ns - 1 ? ticket - (ns - 1) : 0; t < ticket; ++t) {  // (in submission order)
        skx_stream::Staged& older = st->slot[t % ns];
        if (older.pending && older.ticket == t) SKXCHK(staged_process(st, older));
    }
    SKXCHK(staged_finish(st, sl));
    if (sl.dropped) return fail(SKX_ERR_INVALID, "batch %llu was dropped: a batch submitted before it failed, its rows were never written",
                                (unsigned long long)ticket);
    if (st->profiling) collect_spans(st);
    return SKX_OK;
}
SKX_API int skx_stream_drain(skx_stream* st) {
    if (!st) return fail(SKX_ERR_INVALID, "NULL stream");
    SKXCHK(use_device(st->device));
    // oldest first
    for (u64 t = st->next_ticket >= st->n_slots ? st->next_ticket - st->n_slots : 0; t < st->next_ticket; ++t) {
        skx_stream::Staged& sl = st->slot[t % st->n_slots];
        if (sl.pending && sl.ticket == t) SKXCHK(staged_process(st, sl));
    }
    SKXCHK(flush_pending(st));
    for (auto& sl : st->slot)
        if (sl.in_flight) { HIPCHK(hipEventSynchronize(sl.ev_done)); sl.in_flight = false; }
    return skx_stream_sync(st);
}

SKX_API int skx_stream_set_packed_input(skx_stream* st, int on) {
    if (!st) return fail(SKX_ERR_INVALID, "NULL stream");
    SKXCHK(use_device(st->device));
    SKXCHK(flush_pending(st));
    for (auto& sl : st->slot)
        if (sl.pending || sl.in_flight) return fail(SKX_ERR_INVALID, "drain the submitted batches before changing the input format");
    st->packed = on != 0;
    return SKX_OK;
}
// host helper (no device involved): ASCII -> 4-bit codes, appended at nibble position `nibble_pos` of `packed`.
// A FASTX front-end packs every base it parses (sketchy_amd/host: ~40 GB/s of sequence at 25 M reads/s), so the body is a
// table walk with an AVX2 path (32 bases per step: classify by compares, pairs joined by one multiply-add) chosen at run time;
// both give the bytes the plain definition gives: skx::classify_base per byte, whitespace dropped, an even nibble starts its
// byte afresh, an odd one keeps the low nibble already there.
#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
namespace {
struct PackLut {
    uint8_t code[256];
    PackLut() { for (int c = 0; c < 256; ++c) code[c] = (uint8_t)skx::classify_base((u32)c); }
};
const PackLut g_pack_lut;
inline uint64_t pack_scalar(const uint8_t* ascii, uint64_t n, uint8_t* packed, uint64_t pos) {
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t code = g_pack_lut.code[ascii[i]];
        if (code == 5u) continue;  // whitespace does not exist in the packed format
        uint8_t& b = packed[pos >> 1];
        b = (pos & 1ull) ? (uint8_t)((b & 0x0Fu) | (code << 4)) : code;
        ++pos;
    }
    return pos;
}
// LINE: stop at the first line feed (not packed); *consumed = bytes taken, the line feed included
template <bool LINE>
__attribute__((target("avx2"))) uint64_t pack_avx2_t(const uint8_t* ascii, uint64_t n, uint8_t* packed, uint64_t pos, uint64_t* consumed) {
    uint64_t i = 0;
    if ((pos & 1ull) && n) {  // finish the open byte first: the vector path writes whole bytes
        while (i < n && (pos & 1ull)) {
            if (LINE && ascii[i] == '\n') { *consumed = i + 1; return pos; }
            pos = pack_scalar(ascii + i, 1, packed, pos); ++i;
        }
    }
    // per byte x, keyed by its low nibble (vpshufb as a 16-entry table):
    //   want[x & 15] = the one upper-cased base letter with that low nibble (A 41, C 43, T 54, U 55, G 47; 0xFF where there is none):
    //                  x is a base iff (x & 0xDF) == want[x & 15]
    //   wsp[x & 15]  = the one whitespace byte with that low nibble (space 20, tab 09, LF 0A, CR 0D)
    //   code         = ((x >> 1) ^ (x >> 2)) & 3  -- A 0, C 1, G 2, T / U 3 in either case (bits 1-2 of the letters) -- else 4
    const __m256i low = _mm256_set1_epi8(0x0F), up = _mm256_set1_epi8((char)0xDF), three = _mm256_set1_epi8(3), four = _mm256_set1_epi8(4),
                  join = _mm256_set1_epi16(0x1001);
    const __m256i want = _mm256_setr_epi8((char)0xFF, 0x41, (char)0xFF, 0x43, 0x54, 0x55, (char)0xFF, 0x47, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF,
                                          (char)0xFF, 0x41, (char)0xFF, 0x43, 0x54, 0x55, (char)0xFF, 0x47, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF);
    const __m256i wsp = _mm256_setr_epi8(0x20, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, 0x09, 0x0A, (char)0xFF, (char)0xFF, 0x0D, (char)0xFF, (char)0xFF,
                                         0x20, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, 0x09, 0x0A, (char)0xFF, (char)0xFF, 0x0D, (char)0xFF, (char)0xFF);
    while (i + 32 <= n) {
        const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(ascii + i));
        const __m256i nib = _mm256_and_si256(x, low);
        if (_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, _mm256_shuffle_epi8(wsp, nib)))) {
            // (a block with whitespace in it: byte by byte -- the open byte may end up odd)
            uint64_t take = 32;
            if (LINE) {
                const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, _mm256_set1_epi8('\n')));
                if (m) { const uint32_t j = (uint32_t)__builtin_ctz(m); *consumed = i + j + 1; return pack_scalar(ascii + i, j, packed, pos); }
            }
            pos = pack_scalar(ascii + i, take, packed, pos);
            i += take;
            while (i < n && (pos & 1ull)) {
                if (LINE && ascii[i] == '\n') { *consumed = i + 1; return pos; }
                pos = pack_scalar(ascii + i, 1, packed, pos); ++i;
            }
            continue;
        }
        const __m256i is_base = _mm256_cmpeq_epi8(_mm256_and_si256(x, up), _mm256_shuffle_epi8(want, nib));
        const __m256i c2 = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(x, 1), _mm256_srli_epi16(x, 2)), three);  // (bits shifted in from the neighbour byte land above bit 5)
        const __m256i code = _mm256_blendv_epi8(four, c2, is_base);
        const __m256i pairs = _mm256_maddubs_epi16(code, join);           // 16 x (even base | odd base << 4)
        const __m256i bytes = _mm256_permute4x64_epi64(_mm256_packus_epi16(pairs, pairs), 0x08);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(packed + (pos >> 1)), _mm256_castsi256_si128(bytes));
        pos += 32; i += 32;
    }
    if (LINE) {
        for (; i < n; ++i) {
            if (ascii[i] == '\n') { *consumed = i + 1; return pos; }
            pos = pack_scalar(ascii + i, 1, packed, pos);
        }
        *consumed = n;
        return pos;
    }
    return pack_scalar(ascii + i, n - i, packed, pos);
}
__attribute__((target("avx2"))) uint64_t pack_avx2(const uint8_t* ascii, uint64_t n, uint8_t* packed, uint64_t pos) {
    return pack_avx2_t<false>(ascii, n, packed, pos, nullptr);
}
}  // namespace
SKX_API uint64_t skx_pack_bases(const uint8_t* ascii, uint64_t n, uint8_t* packed, uint64_t nibble_pos) {
    static const bool has_avx2 = __builtin_cpu_supports("avx2");
    return has_avx2 ? pack_avx2(ascii, n, packed, nibble_pos) : pack_scalar(ascii, n, packed, nibble_pos);
}
// ... up to the first line feed: a FASTX parser's sequence line in ONE pass over its bytes (finding the end of the line and
// packing it are the same scan).  Packs ascii[0 .. m), m = the index of the first '\n' (or n when there is none);
// *consumed = m + 1 (the line feed included) or n.
namespace {
__attribute__((target("avx2"))) uint64_t pack_line_avx2(const uint8_t* ascii, uint64_t n, uint8_t* packed, uint64_t pos, uint64_t* consumed) {
    *consumed = n;
    return pack_avx2_t<true>(ascii, n, packed, pos, consumed);
}
}  // namespace
SKX_API uint64_t skx_pack_line(const uint8_t* ascii, uint64_t n, uint8_t* packed, uint64_t nibble_pos, uint64_t* consumed) {
    static const bool has_avx2 = __builtin_cpu_supports("avx2");
    uint64_t used = n;
    uint64_t pos;
    if (has_avx2) pos = pack_line_avx2(ascii, n, packed, nibble_pos, &used);
    else {
        const void* e = memchr(ascii, '\n', n);
        const uint64_t m = e ? (uint64_t)(static_cast<const uint8_t*>(e) - ascii) : n;
        used = e ? m + 1 : n;
        pos = pack_scalar(ascii, m, packed, nibble_pos);
    }
    if (consumed) *consumed = used;
    return pos;
}
#else
SKX_API uint64_t skx_pack_bases(const uint8_t*, uint64_t, uint8_t*, uint64_t);
SKX_API uint64_t skx_pack_line(const uint8_t*, uint64_t, uint8_t*, uint64_t, uint64_t*);
#endif
// the reference scan ALONE on the device (profiling aid: with the static dense dictionary a stream's scans run beside its sketches,
// so no entry point times one by itself any more): `reps` launches back to back behind a synchronisation, average milliseconds.
// The membership bits it leaves are the ones any pass of the stream needs (they depend on the reference only): kept for the next pass.
SKX_API int skx_stream_scan_alone(skx_stream* st, uint32_t reps, double* ms_avg) {
    if (!st || !ms_avg) return fail(SKX_ERR_INVALID, "NULL argument");
    *ms_avg = 0.0;
    SKXCHK(skx_stream_sync(st));
    if (!st->static_dense || st->ref->n_sd == 0 || reps == 0)
        return fail(SKX_ERR_INVALID, "this stream builds its scan's dictionary per pass (or the reference has no dense hash): time a synchronous push instead");
    const skx_ref* ref = st->ref;
    const int b = st->buf;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st->hs));
    static const bool big_slices = skx::knob("SKX_SCAN_BIGSLICE") && atoi(skx::knob("SKX_SCAN_BIGSLICE")) != 0;  // (experiment knob, as queue_static_scan)
    for (uint32_t i = 0; i < reps; ++i)
        skx::launch_scan(st->hs, ref->d_mat, ref->s, ref->n_tiles, ref->rb, ref->n_bands, ref->d_qs, ref->d_win_s, st->d_ms[b], nullptr, ref->n_pad,
                         false, true, nullptr, st->d_mdirty_s, true, 0u, big_slices);
    HIPCHK(hipEventRecord(e1, st->hs));
    skx::launch_exceptions(st->hs, ref->d_exc_g, ref->d_exc_h, ref->n_exc, ref->d_qs + ref->sd_tail0, nullptr, st->d_ms[b], ref->n_pad, st->d_mdirty_s, nullptr,
                           ref->sd_tail0, ref->n_sd - ref->sd_tail0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st->hs));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    st->m_ready[b] = true; st->m_filled[b] = true;
    *ms_avg = (double)ms / reps;
    return SKX_OK;
}
SKX_API int skx_stream_sync(skx_stream* st) {
    if (!st) return fail(SKX_ERR_INVALID, "NULL stream");
    SKXCHK(use_device(st->device));
    SKXCHK(flush_pending(st));
    HIPCHK(hipStreamSynchronize(st->hs0));
    HIPCHK(hipStreamSynchronize(st->hs1));
    HIPCHK(hipStreamSynchronize(st->hs));
    HIPCHK(hipStreamSynchronize(st->hs2));
    if (st->profiling) collect_spans(st);
    dump_marks();
    return SKX_OK;
}

SKX_API int skx_stream_table(skx_stream* st, uint64_t* cum) {
    if (!st || !cum) return fail(SKX_ERR_INVALID, "NULL argument");
    SKXCHK(use_device(st->device));
    SKXCHK(flush_pending(st));
    // the running table belongs to the back stream; the caller sees the real genomes, species concatenated
    const u32 n = st->ref->n_genomes;
    skx::launch_gather_table(st->hs2, st->d_cum, st->d_tab_tmp, n, st->ref->d_real2pad);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(cum, st->d_tab_tmp, (size_t)n * 8, hipMemcpyDeviceToHost, st->hs2));
    HIPCHK(hipStreamSynchronize(st->hs2));
    return SKX_OK;
}
SKX_API int skx_stream_table_add(skx_stream* st, const uint64_t* add) {
    if (!st || !add) return fail(SKX_ERR_INVALID, "NULL argument");
    SKXCHK(use_device(st->device));
    SKXCHK(flush_pending(st));
    const u32 n = st->ref->n_genomes;
    HIPCHK(hipMemcpyAsync(st->d_tab_tmp, add, (size_t)n * 8, hipMemcpyHostToDevice, st->hs2));
    skx::launch_add_table(st->hs2, st->d_cum, st->d_tab_tmp, n, st->ref->d_real2pad);
    st->fresh_table = false;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st->hs2));  // `add` is only borrowed for the call
    return SKX_OK;
}
SKX_API int skx_stream_reset(skx_stream* st) {
    if (!st) return fail(SKX_ERR_INVALID, "NULL stream");
    SKXCHK(use_device(st->device));
    SKXCHK(flush_pending(st));
    HIPCHK(hipStreamSynchronize(st->hs0));
    HIPCHK(hipStreamSynchronize(st->hs1));
    HIPCHK(hipStreamSynchronize(st->hs));
    HIPCHK(hipMemsetAsync(st->d_cum, 0, (size_t)st->ref->n_pad * 8, st->hs2));
    HIPCHK(hipStreamSynchronize(st->hs2));
    st->reads_total = 0;
    st->h_nq[2] = 1; st->h_nq[3] = 1;  // (a new sample: every genome is a candidate again)
    st->hint_all_overflow = false;     // (... and nothing is known about its candidates)
    st->hint_seq = st->cand_seq;
    st->legacy_run = 0;
    st->lcount_floor = st->lcount_seq;
    st->fresh_table = true;
    st->cum_writer = nullptr;          // (everything is synchronised)
    return SKX_OK;
}
SKX_API int skx_stream_reads(const skx_stream* st, uint64_t* n_reads) {
    if (!st || !n_reads) return fail(SKX_ERR_INVALID, "NULL argument");
    *n_reads = st->reads_total;
    for (int i = 0; i < st->n_pend; ++i) *n_reads += st->pend[i].n_reads;  // (an enqueued batch counts once it is accepted)
    return SKX_OK;
}
SKX_API int skx_stream_stats(skx_stream* st, uint64_t* out, uint32_t n_out) {
    if (!st || !out) return fail(SKX_ERR_INVALID, "NULL argument");
    // rank groups that received any bit in the most recent pass (waits for the stream)
    // (a deferred error of an enqueued batch surfaces here like in every other entry point that flushes)
    uint64_t live = 0;
    SKXCHK(use_device(st->device));
    SKXCHK(skx_stream_sync(st));
    {
        const u32 n_grp = st->ref->n_pad / (skx::kRankWords * 64);
        std::vector<u32> flags(n_grp, 0);
        HIPCHK(hipMemcpy(flags.data(), st->d_grp_any[st->buf ^ 1], (size_t)n_grp * 4, hipMemcpyDeviceToHost));
        for (u32 f : flags) live += f ? 1 : 0;
    }
    const uint64_t v[SKX_N_STATS] = {st->last_pairs, st->last_passes, (uint64_t)(st->ref->d_kt_key ? st->h_nd[2 * (st->buf ^ 1)] : st->h_nq[st->buf ^ 1]), st->reads_big,
                                     st->total_passes, st->lean_passes, st->pcap, live, st->reads_split, st->segs_split, st->pool_grown,
                                     st->shared_passes, st->groups_unshared, st->qcap, st->qrows_grown,
                                     (uint64_t)(st->ref->d_kt_key ? st->h_nd[2 * (st->buf ^ 1) + 1] : st->h_nq[st->buf ^ 1]),
                                     st->batches_compact, st->batches_full};
    for (uint32_t i = 0; i < n_out; ++i) out[i] = i < SKX_N_STATS ? v[i] : 0;
    return SKX_OK;
}
SKX_API int skx_stream_rank(skx_stream* st, uint32_t top_k, uint32_t* idx, uint64_t* sum) {
    if (!st || !idx || !sum) return fail(SKX_ERR_INVALID, "NULL argument");
    const skx_ref* ref = st->ref;
    if (top_k < 1 || top_k > ref->min_species) return fail(SKX_ERR_INVALID, "top_k=%u outside 1..n_genomes", top_k);
    SKXCHK(use_device(st->device));
    SKXCHK(flush_pending(st));
    const size_t rows = (size_t)ref->n_species * top_k;
    if (rows > st->rank_cap) {  // (beyond SKX_MAX_TOP rows per species: grow the scratch once)
        HIPCHK(hipStreamSynchronize(st->hs2));
        (void)hipFree(st->d_rank_idx); (void)hipFree(st->d_rank_sum);
        st->d_rank_idx = nullptr; st->d_rank_sum = nullptr; st->rank_cap = 0;
        HIPCHK(hipMalloc(&st->d_rank_idx, rows * 4));
        HIPCHK(hipMalloc(&st->d_rank_sum, rows * 8));
        st->rank_cap = (u32)rows;
    }
    skx::launch_rank_table(st->hs2, st->d_cum, ref->species(), top_k, st->d_rank_idx, st->d_rank_sum);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(idx, st->d_rank_idx, rows * 4, hipMemcpyDeviceToHost, st->hs2));
    HIPCHK(hipMemcpyAsync(sum, st->d_rank_sum, rows * 8, hipMemcpyDeviceToHost, st->hs2));
    HIPCHK(hipStreamSynchronize(st->hs2));
    return SKX_OK;
}

// ------------------------------------------------------------------ stand-alone operators
SKX_API int skx_sketch_reads(int device, uint32_t k, uint64_t seed, uint32_t s, const uint8_t* bases,
                             const uint64_t* offsets, uint32_t n_reads, uint64_t* sketches, uint32_t* sketch_len) {
    if (!offsets || !sketches || !sketch_len) return fail(SKX_ERR_INVALID, "NULL argument");
    if (k < 1 || k > SKX_MAX_K || s < 1) return fail(SKX_ERR_INVALID, "bad k/s");
    if (n_reads == 0) return SKX_OK;
    SKXCHK(use_device(device));
    const u64 base0 = offsets[0], n_bases = offsets[n_reads] - base0;
    if (n_bases && !bases) return fail(SKX_ERR_INVALID, "bases is NULL");
    std::vector<u64> off(n_reads + 1);
    u64 longest = 1;
    for (u32 r = 0; r <= n_reads; ++r) {
        off[r] = offsets[r] - base0;
        if (r && offsets[r] < offsets[r - 1]) return fail(SKX_ERR_INVALID, "offsets not monotonic at read %u", r - 1);
        if (r) longest = std::max<u64>(longest, offsets[r] - offsets[r - 1]);
    }
    const u32 stride = (u32)std::min<u64>(s, std::max<u64>(longest, 1));
    uint8_t* d_b = nullptr; u64 *d_o = nullptr, *d_sk = nullptr; u32 *d_len = nullptr, *d_cnt = nullptr, *d_lists = nullptr;
    int rc = SKX_OK;
    hipError_t e = hipSuccess;
    do {
        if ((e = hipMalloc(&d_b, std::max<u64>(n_bases, 1))) != hipSuccess) break;
        if ((e = hipMalloc(&d_o, ((size_t)n_reads + 1) * 8)) != hipSuccess) break;
        if ((e = hipMalloc(&d_sk, (size_t)n_reads * stride * 8)) != hipSuccess) break;
        if ((e = hipMalloc(&d_len, (size_t)n_reads * 4)) != hipSuccess) break;
        if ((e = hipMalloc(&d_cnt, (size_t)n_reads * 4)) != hipSuccess) break;
        if ((e = hipMalloc(&d_lists, 2 * ((size_t)n_reads + 1) * 4)) != hipSuccess) break;  // retry / big lists
        if (n_bases && (e = hipMemcpy(d_b, bases + base0, n_bases, hipMemcpyHostToDevice)) != hipSuccess) break;
        if ((e = hipMemcpy(d_o, off.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice)) != hipSuccess) break;
        if ((e = hipMemset(d_sk, 0, (size_t)n_reads * stride * 8)) != hipSuccess) break;
        if ((e = hipMemset(d_lists, 0, 2 * ((size_t)n_reads + 1) * 4)) != hipSuccess) break;
        if ((e = skx::launch_sketch(nullptr, d_b, d_o, n_reads, k, seed, s, 0, false, d_sk, stride, d_len, d_cnt, nullptr, 0,
                                    d_lists, d_lists + n_reads + 1, n_bases, nullptr, 0)) != hipSuccess) break;
        // sequences with more k-mers than a wave holds wait on the second list: the block sketcher
        u32 n_big = 0;
        if ((e = hipMemcpy(&n_big, d_lists + n_reads + 1, 4, hipMemcpyDeviceToHost)) != hipSuccess) break;
        if ((e = skx::launch_sketch_block(nullptr, d_b, d_o, d_lists + n_reads + 1, n_big, k, seed, s, 0, false, d_sk, stride, d_len,
                                          d_cnt, nullptr, 0)) != hipSuccess) break;
        memset(sketches, 0, (size_t)n_reads * s * 8);
        if ((e = hipMemcpy2D(sketches, (size_t)s * 8, d_sk, (size_t)stride * 8, (size_t)stride * 8, n_reads, hipMemcpyDeviceToHost)) != hipSuccess) break;
        if ((e = hipMemcpy(sketch_len, d_len, (size_t)n_reads * 4, hipMemcpyDeviceToHost)) != hipSuccess) break;
    } while (0);
    if (e != hipSuccess) rc = fail(SKX_ERR_HIP, "skx_sketch_reads: %s", hipGetErrorString(e));
    (void)hipFree(d_b); (void)hipFree(d_o); (void)hipFree(d_sk); (void)hipFree(d_len); (void)hipFree(d_cnt); (void)hipFree(d_lists);
    return rc;
}

SKX_API int skx_common_hashes(const skx_ref* ref, const uint64_t* query, const uint32_t* query_len, uint32_t n_query,
                              uint32_t q_stride, uint32_t* common) {
    if (!ref || !query || !query_len || !common) return fail(SKX_ERR_INVALID, "NULL argument");
    if (n_query == 0) return SKX_OK;
    u32 max_len = 1;
    for (u32 i = 0; i < n_query; ++i) {
        if (query_len[i] > q_stride) return fail(SKX_ERR_INVALID, "query_len[%u] exceeds q_stride", i);
        const uint64_t* row = query + (size_t)i * q_stride;
        for (u32 j = 1; j < query_len[i]; ++j)
            if (row[j] <= row[j - 1]) return fail(SKX_ERR_UNSORTED, "query %u not strictly ascending at %u", i, j);
        max_len = std::max(max_len, query_len[i]);
    }
    skx_stream* st = nullptr;
    SKXCHK(stream_create_internal(&st, ref, 0, n_query, 1, max_len, max_len, true));  // (whole sketches: every hash is a pair)
    int rc = SKX_OK;
    do {
        // candidate prefix of every query: hashes <= max_ref (ascending rows)
        std::vector<u32> cnt(n_query + 1, 0);
        st->h_poff[0] = 0;
        for (u32 i = 0; i < n_query; ++i) {
            const uint64_t* row = query + (size_t)i * q_stride;
            u32 c = ref->any ? (u32)(std::upper_bound(row, row + query_len[i], (uint64_t)ref->max_ref) - row) : 0;
            st->h_poff[i + 1] = st->h_poff[i] + c;
        }
        hipError_t e = use_rows(st);  // (the queries are full-width rows)
        if (e == hipSuccess) e = hipMemcpy2D(st->d_sk, (size_t)max_len * 8, query, (size_t)q_stride * 8, (size_t)max_len * 8, n_query, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(st->d_poff, st->h_poff, ((size_t)n_query + 1) * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) { rc = fail(SKX_ERR_HIP, "skx_common_hashes upload: %s", hipGetErrorString(e)); break; }
        u32* d_shared = nullptr;
        const u32 out_cap = (u32)std::max<u64>(1, (256ull << 20) / ((u64)ref->n_genomes * 4));
        rc = for_each_pass(st, n_query, out_cap, [&](u32 ra, u32 rb, u32 p_base, u32 P) -> int {
            if (d_shared) { (void)hipFree(d_shared); d_shared = nullptr; }
            HIPCHK(hipMalloc(&d_shared, (size_t)(rb - ra) * ref->n_genomes * 4));
            if (P == 0) {
                HIPCHK(hipMemsetAsync(d_shared, 0, (size_t)(rb - ra) * ref->n_genomes * 4, st->hs2));
            } else {
                SKXCHK(run_pass(st, ra, rb, p_base, P, nullptr, nullptr, d_shared, false));
                SKXCHK(queue_chains(st, true));
            }
            HIPCHK(hipMemcpyAsync(common + (size_t)ra * ref->n_genomes, d_shared, (size_t)(rb - ra) * ref->n_genomes * 4,
                                  hipMemcpyDeviceToHost, st->hs2));
            HIPCHK(hipStreamSynchronize(st->hs2));
            return SKX_OK;
        });
        if (d_shared) (void)hipFree(d_shared);
    } while (0);
    stream_free(st);
    return rc;
}

// ------------------------------------------------------------------ RCCL (loaded on first use)
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
    if (g_rccl.h) return SKX_OK;
    // soname first: a process that already loaded an RCCL (e.g. through torch) reuses that copy
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail(SKX_ERR_COMM, "cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(SKX_ERR_COMM, "librccl lacks the expected symbols");
    g_rccl.h = h;
    return SKX_OK;
}
struct skx_comm { int device; int rank, n_ranks; ncclComm_t comm; };

// RCCL's bring-up and its collectives block for ever when a peer never arrives (a rank that died, a wrong id, a fabric that
// does not come up): on a node where N ranks meet for the first time that is a job hanging until somebody's wall clock kills
// it.  With the option "comm_timeout_ms" set, the two calls that can block -- ncclCommInitRank and the all-reduce with its
// stream synchronisation -- run under a watchdog thread that, when the call has not returned in time, says which rank was
// stuck in what and ends the PROCESS with exit code SKX_COMM_TIMEOUT_EXIT (86).  A hung collective cannot be cancelled and
// the communicator is unusable afterwards; ending the rank lets the launcher tear the job down (never a re-exec: the process
// has initialised the GPU).  Off by default: a library does not exit its host unasked.
struct CommWatchdog {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    CommWatchdog(u64 ms, const char* what, int rank, int n_ranks, int device) {
        if (!ms) return;
        th = std::thread([this, ms, what, rank, n_ranks, device] {
            std::unique_lock<std::mutex> lk(mu);
            if (!cv.wait_for(lk, std::chrono::milliseconds(ms), [this] { return done; })) {
                fprintf(stderr, "[sketchy-hip] rank %d of %d (device %d): %s did not return within %llu ms (option comm_timeout_ms) -- "
                                "a peer is missing or the fabric is down; exiting with code %d\n",
                        rank, n_ranks, device, what, (unsigned long long)ms, SKX_COMM_TIMEOUT_EXIT);
                fflush(stderr);
                _exit(SKX_COMM_TIMEOUT_EXIT);
            }
        });
    }
    ~CommWatchdog() {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu); done = true; }
        cv.notify_all();
        th.join();
    }
};
static_assert(sizeof(ncclUniqueId) == SKX_COMM_ID_BYTES, "RCCL unique id size");

SKX_API int skx_comm_unique_id(uint8_t id[SKX_COMM_ID_BYTES]) {
    if (!id) return fail(SKX_ERR_INVALID, "NULL argument");
    SKXCHK(rccl_load());
    ncclUniqueId u;
    ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return fail(SKX_ERR_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    memcpy(id, &u, SKX_COMM_ID_BYTES);
    return SKX_OK;
}
SKX_API int skx_comm_create(skx_comm** out, int device, int rank, int n_ranks, const uint8_t id[SKX_COMM_ID_BYTES]) {
    if (!out || !id) return fail(SKX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(SKX_ERR_INVALID, "bad rank %d of %d", rank, n_ranks);
    SKXCHK(use_device(device));
    SKXCHK(rccl_load());
    ncclUniqueId u;
    memcpy(&u, id, SKX_COMM_ID_BYTES);
    ncclComm_t c;
    ncclResult_t r;
    {
        CommWatchdog wd(g_comm_timeout_ms, "ncclCommInitRank", rank, n_ranks, device);
        r = g_rccl.CommInitRank(&c, n_ranks, u, rank);
    }
    if (r != ncclSuccess) return fail(SKX_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    *out = new skx_comm{device, rank, n_ranks, c};
    return SKX_OK;
}
SKX_API int skx_comm_n_ranks(const skx_comm* comm, int* n_ranks) {
    if (!comm || !n_ranks) return fail(SKX_ERR_INVALID, "NULL argument");
    *n_ranks = comm->n_ranks;
    if (g_rccl.CommCount) {
        int n = 0;
        ncclResult_t r = g_rccl.CommCount(comm->comm, &n);
        if (r != ncclSuccess) return fail(SKX_ERR_COMM, "ncclCommCount: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
        *n_ranks = n;
    }
    return SKX_OK;
}
SKX_API int skx_stream_allreduce(skx_stream* st, skx_comm* comm) {
    if (!st || !comm) return fail(SKX_ERR_INVALID, "NULL argument");
    if (comm->device != st->device) return fail(SKX_ERR_INVALID, "communicator and stream are on different devices");
    SKXCHK(use_device(st->device));
    SKXCHK(flush_pending(st));  // (an enqueued batch whose passes are still to be queued belongs to the table)
    // one sum all-reduce of the u64 table (8*N bytes: latency-bound); in place
    // (the padded table: padding entries are 0 on every rank)
    CommWatchdog wd(g_comm_timeout_ms, "ncclAllReduce of the running table", comm->rank, comm->n_ranks, comm->device);
    ncclResult_t r = g_rccl.AllReduce(st->d_cum, st->d_cum, st->ref->n_pad, ncclUint64, ncclSum, comm->comm, st->hs2);
    if (r != ncclSuccess) return fail(SKX_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    HIPCHK(hipStreamSynchronize(st->hs2));
    return SKX_OK;
}
SKX_API void skx_comm_destroy(skx_comm* comm) {
    if (!comm) return;
    if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm->comm);
    delete comm;
}

#ifdef SKX_EXPERIMENTS
namespace skx { void rank_debug_counters(unsigned long long* out, bool reset); }
// experiments build only: counters of rank_seg_top1_kernel (see skx_kernels.hip); out[128]
SKX_API void skx_debug_rank_counters(unsigned long long* out, int reset) { skx::rank_debug_counters(out, reset != 0); }
#endif
