// skx_stub.cpp -- a stand-in for libsketchy_hip.so behind the C++ host, for the SANITIZER build of the host only (tests/: the
// host's parsers and its streaming pipeline run on the CPU under ASan + UBSan; no device, no kernels).  It is test infrastructure:
// nothing under sketchy_amd/ links it.
//
// What it "computes" makes the host's output checkable without a device: the row of a read is genome (n_bases mod n_genomes)
// with sum = FNV-1a over the read's 4-bit codes as the host packed them -- so the text the host prints pins, read by read and in
// order, exactly which bases reached the boundary (tests/test_host_cpu.py recomputes it from the sequences).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sketchy_hip.h"

struct skx_ref { uint32_t n = 0, s = 0; };
struct skx_stream { const skx_ref* ref; uint32_t top; uint32_t max_reads; uint64_t max_bases; bool packed = false; uint64_t next_ticket = 0; uint64_t reads = 0; };
static std::string g_err;
static int fail(int c, const char* m) { g_err = m; return c; }
static uint64_t fnv_codes(const uint8_t* bases, uint64_t a, uint64_t b, bool packed) {
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = a; i < b; ++i) {
        uint8_t c;
        if (packed) { c = (uint8_t)((bases[i >> 1] >> (4 * (i & 1))) & 0xF); if (c > 3) c = 4; }
        else {
            const uint8_t x = bases[i], u = x & 0xDF;
            if (x == ' ' || x == '\t' || x == '\r' || x == '\n') continue;
            c = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : (u == 'T' || u == 'U') ? 3 : 4;
        }
        h = (h ^ c) * 1099511628211ull;
    }
    return h;
}
extern "C" {
const char* skx_last_error(void) { return g_err.c_str(); }
const char* skx_version(void) { return "skx stub (tests)"; }
int skx_device_count(void) { return 1; }
int skx_device_pci_bus_id(int, char* bus_id, size_t cap) { if (cap) bus_id[0] = 0; return fail(SKX_ERR_NO_DEVICE, "stub: no device"); }
int skx_ref_create(skx_ref** out, int, uint32_t k, uint64_t, uint32_t s, uint32_t stride, uint32_t n, const uint64_t* hashes, const uint32_t* col_len) {
    if (!out || !hashes || !col_len || k < 1 || k > SKX_MAX_K || s < 1 || stride < 1 || n < 1) return fail(SKX_ERR_INVALID, "stub: bad reference");
    for (uint32_t g = 0; g < n; ++g) {
        if (col_len[g] > stride) return fail(SKX_ERR_INVALID, "stub: col_len exceeds stride");
        for (uint32_t i = 1; i < col_len[g]; ++i)
            if (hashes[(size_t)g * stride + i] <= hashes[(size_t)g * stride + i - 1]) return fail(SKX_ERR_UNSORTED, "stub: hashes not strictly ascending");
    }
    *out = new skx_ref{n, s};
    return SKX_OK;
}
void skx_ref_destroy(skx_ref* r) { delete r; }
int skx_stream_create(skx_stream** out, const skx_ref* ref, uint32_t top, uint32_t max_reads, uint64_t max_bases) {
    if (!out || !ref || top > ref->n || max_reads < 1) return fail(SKX_ERR_INVALID, "stub: bad stream");
    *out = new skx_stream{ref, top, max_reads, max_bases};
    return SKX_OK;
}
void skx_stream_destroy(skx_stream* s) { delete s; }
int skx_stream_set_packed_input(skx_stream* s, int on) { s->packed = on != 0; return SKX_OK; }
int skx_host_alloc(int, void** p, size_t bytes) { *p = malloc(bytes ? bytes : 1); return *p ? SKX_OK : fail(SKX_ERR_HIP, "stub: out of memory"); }
int skx_host_free(int, void* p) { free(p); return SKX_OK; }
static int score(skx_stream* st, const uint8_t* bases, const uint64_t* offsets, uint32_t n, uint32_t* idx, uint64_t* sum) {
    if (n > st->max_reads) return fail(SKX_ERR_CAPACITY, "stub: n_reads exceeds max_batch_reads");
    if (offsets[n] - offsets[0] > st->max_bases) return fail(SKX_ERR_CAPACITY, "stub: batch exceeds max_batch_bases");
    static const bool fast = getenv("SKX_STUB_FAST") != nullptr;  // (throughput runs of the front-end: no per-base work behind the boundary)
    for (uint32_t r = 0; r < n; ++r) {
        if (offsets[r + 1] < offsets[r]) return fail(SKX_ERR_INVALID, "stub: offsets not monotonic");
        const uint64_t h = fast ? 0 : fnv_codes(bases, offsets[r], offsets[r + 1], st->packed);
        uint64_t nb = 0;
        if (st->packed) nb = offsets[r + 1] - offsets[r];
        else for (uint64_t i = offsets[r]; i < offsets[r + 1]; ++i) nb += !(bases[i] == ' ' || bases[i] == '\t' || bases[i] == '\r' || bases[i] == '\n');
        for (uint32_t t = 0; t < st->top; ++t) {
            if (idx) idx[(size_t)r * st->top + t] = (uint32_t)((nb + t) % st->ref->n);
            if (sum) sum[(size_t)r * st->top + t] = h >> 8;
        }
    }
    st->reads += n;
    return SKX_OK;
}
int skx_stream_submit(skx_stream* st, const uint8_t* bases, const uint64_t* offsets, uint32_t n, uint32_t* idx, uint64_t* sum, uint64_t* ticket) {
    if (!st || !offsets || n == 0) return fail(SKX_ERR_INVALID, "stub: empty batch");
    const int rc = score(st, bases, offsets, n, idx, sum);
    if (ticket) *ticket = st->next_ticket;
    st->next_ticket++;
    return rc;
}
int skx_stream_wait(skx_stream* st, uint64_t ticket) { return ticket < st->next_ticket ? SKX_OK : fail(SKX_ERR_INVALID, "stub: ticket never issued"); }
int skx_stream_drain(skx_stream*) { return SKX_OK; }
int skx_stream_push(skx_stream* st, const uint8_t* bases, const uint64_t* offsets, uint32_t n, uint32_t* idx, uint64_t* sum, uint32_t*, uint64_t*, uint32_t*) {
    return n ? score(st, bases, offsets, n, idx, sum) : SKX_OK;
}
#ifndef SKX_STUB_NO_PACK  /* (throughput runs link the library's own packer instead: -DSKX_STUB_NO_PACK -lsketchy_hip) */
uint64_t skx_pack_bases(const uint8_t* ascii, uint64_t n, uint8_t* packed, uint64_t pos) {
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t x = ascii[i], u = x & 0xDF;
        if (x == ' ' || x == '\t' || x == '\r' || x == '\n') continue;
        const uint8_t c = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : (u == 'T' || u == 'U') ? 3 : 4;
        uint8_t& b = packed[pos >> 1];
        b = (pos & 1) ? (uint8_t)((b & 0x0F) | (c << 4)) : c;
        ++pos;
    }
    return pos;
}
uint64_t skx_pack_line(const uint8_t* ascii, uint64_t n, uint8_t* packed, uint64_t pos, uint64_t* consumed) {
    const void* e = memchr(ascii, '\n', n);
    const uint64_t m = e ? (uint64_t)(static_cast<const uint8_t*>(e) - ascii) : n;
    if (consumed) *consumed = e ? m + 1 : n;
    return skx_pack_bases(ascii, m, packed, pos);
}
#endif
int skx_sketch_reads(int, uint32_t, uint64_t, uint32_t s, const uint8_t*, const uint64_t*, uint32_t n, uint64_t* sketches, uint32_t* len) {
    memset(sketches, 0, (size_t)n * s * 8);
    for (uint32_t r = 0; r < n; ++r) len[r] = 0;
    return SKX_OK;
}
int skx_common_hashes(const skx_ref* ref, const uint64_t*, const uint32_t*, uint32_t nq, uint32_t, uint32_t* common) {
    memset(common, 0, (size_t)nq * ref->n * 4);
    return SKX_OK;
}
}
