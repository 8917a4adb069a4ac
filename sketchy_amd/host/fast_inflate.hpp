// fast_inflate.hpp -- a raw DEFLATE (RFC 1951) decoder for members that sit in memory whole, written for the BGZF front-end
// (formats.hpp, MappedFile::open_bgzf): input and output buffers are complete and the output size is known from the member's trailer,
// so the hot loop needs no streaming state -- a 64-bit bit buffer refilled eight bytes at a time, an 11-bit primary table for the
// literal / length code whose entries carry base value and extra-bit count (one lookup per symbol, two for the rare long codes), an
// 8-bit one for the distances, word-wise match copies, and a second 11-bit table that yields every literal whose whole code lies inside the
// index (up to four per lookup: four bases, one or two quality values).  zlib's inflate() does 0.14-0.25 GB/s per thread on FASTQ with noisy
// quality strings; this does 1.5-1.8 x that (`tests/stub/inflate_check speed`).  Nothing is trusted: every table entry, distance and length is checked against the
// buffers, a corrupt member makes inflate_raw return false (tests/test_host_cpu.py runs it under ASan + UBSan on mutated members).
// The reference reads gzip through needletail / flate2 (src/sketchy.rs:89-92); the bytes that come out are the same.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace sketchy {

class FastInflate {
  public:
    // true: `in` held exactly one complete raw-deflate stream (final block seen) that inflated to exactly out_len bytes
    bool inflate_raw(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len) {
        start(in, in_len, out, out_len);
        while (prepare() == kFast) fast_single();
        return ok();
    }

  private:
    // table entries (u32): bits 0-7 = bits to consume (0: no such code), 8-12 = extra bits (lengths, distances) or the subtable's index bits,
    // 16-.. = literal / length base / distance base / subtable start; flags below
    // (literal / length table: literal bit 31, end of block bit 30, subtable bit 29 -- start in bits 16-28; distance table: the base takes
    // bits 16-30, its subtable flag is bit 31)
    static constexpr uint32_t kLiteral = 0x80000000u, kEndOfBlock = 0x40000000u, kSubtable = 0x20000000u, kOffSubtable = 0x80000000u;
    static constexpr unsigned kLitBits = 11, kOffBits = 8, kPreBits = 7;
    static constexpr unsigned kLitSize = (1u << kLitBits) + 288u * 16u, kOffSize = (1u << kOffBits) + 32u * 128u, kPreSize = 1u << kPreBits;
    enum Kind { kKindLit, kKindOff, kKindPre };
    uint32_t dyn_lit[kLitSize], dyn_off[kOffSize], fixed_lit[kLitSize], fixed_off[kOffSize], pre_tab[kPreSize];
    uint64_t dyn_pack[1u << kLitBits], fixed_pack[1u << kLitBits];  // bits 0-7 = bits of the run, 8-15 = literals in it (0: none), 16-47 = the literals
    const uint32_t* lit = nullptr;
    const uint32_t* off = nullptr;
    const uint64_t* pack = nullptr;
    bool fixed_ready = false;
    // ---- the decoder's state between calls (a member is decoded by start, then prepare / fast_* in turns)
    enum { kDone = 0, kFast = 1 };
    const uint8_t* in_ = nullptr;
    const uint8_t* in_end_ = nullptr;
    uint8_t* out_ = nullptr;
    uint8_t* out0_ = nullptr;
    uint8_t* out_end_ = nullptr;
    uint64_t bb_ = 0;   // bit buffer, LSB first
    int bc_ = 0;        // valid bits in it (exact between calls)
    bool final_ = false, in_block_ = false, failed_ = false, finished_ = false;

    void start(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len) {
        in_ = in; in_end_ = in + in_len; out_ = out0_ = out; out_end_ = out + out_len;
        bb_ = 0; bc_ = 0; final_ = in_block_ = failed_ = finished_ = false;
    }
    bool ok() const { return finished_ && !failed_ && out_ == out_end_; }  // (bytes behind the stream's last block are the caller's: a trailer, padding)
    bool fast_room() const { return (size_t)(in_end_ - in_) >= 32 && (size_t)(out_end_ - out_) >= 320; }

    // careful refill: byte-wise, never past in_end (missing bits read as zero and are caught by `bc < 0` after they are consumed)
#define SKX_REFILL_SAFE() do { while (bc <= 56 && in < in_end) { bb |= (uint64_t)*in++ << bc; bc += 8; } } while (0)
#define SKX_TAKE(n) do { bb >>= (n); bc -= (int)(n); } while (0)
#define SKX_FAIL() do { failed_ = true; finished_ = true; in_ = in; out_ = out; bb_ = bb; bc_ = bc; return kDone; } while (0)
    // Everything that is not the fast symbol loop: block headers, code tables, stored blocks, and -- near the ends of the buffers -- the
    // symbols themselves, one at a time with every access checked.  Returns kFast inside a compressed block with room for the fast
    // loop on both sides, kDone when the member is finished (ok() has the verdict).
    int prepare() {
        if (finished_) return kDone;
        const uint8_t* in = in_;
        const uint8_t* const in_end = in_end_;
        uint8_t* out = out_;
        uint8_t* const out0 = out0_;
        uint8_t* const out_end = out_end_;
        uint64_t bb = bb_;
        int bc = bc_;
        for (;;) {
            if (!in_block_) {
                if (final_) { finished_ = true; in_ = in; out_ = out; bb_ = bb; bc_ = bc; return kDone; }
                SKX_REFILL_SAFE();
                const unsigned type = ((unsigned)bb >> 1) & 3u;
                final_ = ((unsigned)bb & 1u) != 0;
                SKX_TAKE(3);
                if (bc < 0) SKX_FAIL();
                if (type == 0) {  // stored: skip to the byte boundary, LEN, NLEN
                    SKX_TAKE(bc & 7);
                    SKX_REFILL_SAFE();
                    if (bc < 32) SKX_FAIL();
                    const unsigned len = (unsigned)bb & 0xFFFFu, nlen = ((unsigned)(bb >> 16)) & 0xFFFFu;
                    SKX_TAKE(32);
                    if ((len ^ nlen) != 0xFFFFu) SKX_FAIL();
                    in -= bc >> 3; bb = 0; bc = 0;  // (the bytes still in the bit buffer belong to the block: hand them back)
                    if ((size_t)(in_end - in) < len || (size_t)(out_end - out) < len) SKX_FAIL();
                    if (len) memcpy(out, in, len);
                    in += len; out += len;
                    continue;
                }
                if (type == 3) SKX_FAIL();
                if (type == 1) {
                    if (!fixed_ready) build_fixed();
                    lit = fixed_lit; off = fixed_off; pack = fixed_pack;
                } else {
                    // dynamic: HLIT, HDIST, HCLEN, the code-length code, then the two codes' lengths run-length coded with it
                    SKX_REFILL_SAFE();
                    const unsigned hlit = ((unsigned)bb & 31u) + 257u, hdist = (((unsigned)bb >> 5) & 31u) + 1u, hclen = (((unsigned)bb >> 10) & 15u) + 4u;
                    SKX_TAKE(14);
                    if (bc < 0 || hlit > 286u || hdist > 30u) SKX_FAIL();
                    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    uint8_t pl[19];
                    memset(pl, 0, sizeof pl);
                    for (unsigned i = 0; i < hclen; ++i) {
                        SKX_REFILL_SAFE();
                        pl[order[i]] = (uint8_t)(bb & 7u);
                        SKX_TAKE(3);
                    }
                    if (bc < 0) SKX_FAIL();
                    if (!build(pl, 19, pre_tab, kPreBits, kPreSize, kKindPre)) SKX_FAIL();
                    uint8_t lens[286 + 30 + 138];
                    unsigned n = 0;
                    const unsigned total = hlit + hdist;
                    while (n < total) {
                        SKX_REFILL_SAFE();
                        const uint32_t e = pre_tab[bb & ((1u << kPreBits) - 1u)];
                        const unsigned l = e & 0xFFu;
                        if (l == 0) SKX_FAIL();
                        SKX_TAKE(l);
                        const unsigned sym = e >> 16;
                        if (sym < 16) { lens[n++] = (uint8_t)sym; }
                        else {
                            unsigned rep, val = 0;
                            if (sym == 16) { if (n == 0) SKX_FAIL(); val = lens[n - 1]; rep = 3u + ((unsigned)bb & 3u); SKX_TAKE(2); }
                            else if (sym == 17) { rep = 3u + ((unsigned)bb & 7u); SKX_TAKE(3); }
                            else { rep = 11u + ((unsigned)bb & 127u); SKX_TAKE(7); }
                            if (n + rep > total) SKX_FAIL();
                            memset(lens + n, (int)val, rep);
                            n += rep;
                        }
                        if (bc < 0) SKX_FAIL();
                    }
                    if (lens[256] == 0) SKX_FAIL();  // (no end-of-block code)
                    if (!build(lens, hlit, dyn_lit, kLitBits, kLitSize, kKindLit)) SKX_FAIL();
                    if (!build(lens + hlit, hdist, dyn_off, kOffBits, kOffSize, kKindOff)) SKX_FAIL();
                    build_pack(dyn_lit, dyn_pack);
                    lit = dyn_lit; off = dyn_off; pack = dyn_pack;
                }
                in_block_ = true;
            }
            // inside a compressed block
            if ((size_t)(in_end - in) >= 32 && (size_t)(out_end - out) >= 320) { in_ = in; out_ = out; bb_ = bb; bc_ = bc; return kFast; }
            // careful: one symbol, every access checked
            SKX_REFILL_SAFE();
            uint32_t e = lit[bb & ((1u << kLitBits) - 1u)];
            if (e & kSubtable) {
                SKX_TAKE(kLitBits);
                e = lit[((e >> 16) & 0x1FFFu) + (bb & ((1u << ((e >> 8) & 0xFu)) - 1u))];
            }
            if ((e & 0xFFu) == 0) SKX_FAIL();
            SKX_TAKE(e & 0xFFu);
            if (bc < 0) SKX_FAIL();
            if (e & kLiteral) {
                if (out == out_end) SKX_FAIL();
                *out++ = (uint8_t)(e >> 16);
                continue;
            }
            if (e & kEndOfBlock) { in_block_ = false; continue; }
            const unsigned xl = (e >> 8) & 0x1Fu;
            const unsigned len = (e >> 16) + ((unsigned)bb & ((1u << xl) - 1u));
            SKX_TAKE(xl);
            SKX_REFILL_SAFE();
            uint32_t d = off[bb & ((1u << kOffBits) - 1u)];
            if (d & kOffSubtable) {
                SKX_TAKE(kOffBits);
                d = off[((d >> 16) & 0x1FFFu) + (bb & ((1u << ((d >> 8) & 0xFu)) - 1u))];
            }
            if ((d & 0xFFu) == 0) SKX_FAIL();
            SKX_TAKE(d & 0xFFu);
            const unsigned xd = (d >> 8) & 0xFu;
            const size_t dist = ((d >> 16) & 0x7FFFu) + (size_t)(bb & ((1ull << xd) - 1ull));
            SKX_TAKE(xd);
            if (bc < 0 || dist > (size_t)(out - out0) || len > (size_t)(out_end - out)) SKX_FAIL();
            const uint8_t* src = out - dist;
            for (unsigned i = 0; i < len; ++i) out[i] = src[i];
            out += len;
        }
    }
#undef SKX_FAIL
#undef SKX_REFILL_SAFE

    // One step of the fast symbol loop: a refill, then a run of up to sixteen literals (four table lookups of up to four literals each)
    // or one literal / length + distance pair.  0: go on, 1: end of block, 2: corrupt.  Needs 32 readable bytes at `in` and 320 writable
    // at `out` (eight-byte refills, word copies that run up to 7 bytes past a match).
    static inline __attribute__((always_inline)) int fast_step(const uint8_t*& in, uint8_t*& out, const uint8_t* out0, uint64_t& bb, int& bc,
                                                               const uint32_t* lit, const uint32_t* off, const uint64_t* pack) {
        // (the careful refill leaves exact counts; this one keeps bc in 56..63 with valid data above it)
#define SKX_REFILL_FAST() do { uint64_t w_; memcpy(&w_, in, 8); bb |= w_ << bc; in += (63 - bc) >> 3; bc |= 56; } while (0)
        SKX_REFILL_FAST();
        uint64_t pk = pack[bb & ((1u << kLitBits) - 1u)];
        if (pk & 0xFF00u) {
#define SKX_PACKED() do { const uint32_t w4_ = (uint32_t)(pk >> 16); memcpy(out, &w4_, 4); out += (pk >> 8) & 0xFFu; SKX_TAKE(pk & 0xFFu); pk = pack[bb & ((1u << kLitBits) - 1u)]; } while (0)
            SKX_PACKED();
            if (pk & 0xFF00u) {
                SKX_PACKED();
                if (pk & 0xFF00u) {
                    SKX_PACKED();
                    if (pk & 0xFF00u) { SKX_PACKED(); return 0; }
                }
            }
#undef SKX_PACKED
            SKX_REFILL_FAST();
        }
        uint32_t e = lit[bb & ((1u << kLitBits) - 1u)];
        if (e & kSubtable) {
            SKX_TAKE(kLitBits);
            e = lit[((e >> 16) & 0x1FFFu) + (bb & ((1u << ((e >> 8) & 0xFu)) - 1u))];
        }
        if ((e & 0xFFu) == 0) return 2;  // (a code the block's table does not hold)
        SKX_TAKE(e & 0xFFu);
        if (e & kLiteral) { *out++ = (uint8_t)(e >> 16); return 0; }
        if (e & kEndOfBlock) return 1;
        const unsigned xl = (e >> 8) & 0x1Fu;
        const unsigned len = (e >> 16) + ((unsigned)bb & ((1u << xl) - 1u));
        SKX_TAKE(xl);
        SKX_REFILL_FAST();
        uint32_t d = off[bb & ((1u << kOffBits) - 1u)];
        if (d & kOffSubtable) {
            SKX_TAKE(kOffBits);
            d = off[((d >> 16) & 0x1FFFu) + (bb & ((1u << ((d >> 8) & 0xFu)) - 1u))];
        }
        if ((d & 0xFFu) == 0) return 2;
        SKX_TAKE(d & 0xFFu);
        const unsigned xd = (d >> 8) & 0xFu;
        const size_t dist = ((d >> 16) & 0x7FFFu) + (size_t)(bb & ((1ull << xd) - 1ull));
        SKX_TAKE(xd);
        if (dist > (size_t)(out - out0)) return 2;
        const uint8_t* src = out - dist;
        uint8_t* dst = out;
        out += len;
        if (dist >= 8) {
            do { uint64_t w_; memcpy(&w_, src, 8); memcpy(dst, &w_, 8); src += 8; dst += 8; } while (dst < out);
        } else if (dist == 1) {
            const uint64_t w_ = 0x0101010101010101ull * src[0];
            do { memcpy(dst, &w_, 8); dst += 8; } while (dst < out);
        } else {
            do { *dst++ = *src++; } while (dst < out);
        }
        return 0;
#undef SKX_REFILL_FAST
    }
#undef SKX_TAKE
    void leave_fast(const uint8_t* in, uint8_t* out, uint64_t bb, int bc, int r) {
        in_ = in; out_ = out; bc_ = bc;
        bb_ = bb & (bc >= 64 ? ~0ull : ((1ull << bc) - 1ull));  // back to exact bit accounting
        if (r == 1) in_block_ = false;
        if (r == 2) { failed_ = true; finished_ = true; }
    }
    void fast_single() {
        const uint8_t* in = in_;
        uint8_t* out = out_;
        uint64_t bb = bb_;
        int bc = bc_, r;
        const uint8_t* const in_fast = in_end_ - 32;
        uint8_t* const out_fast = out_end_ - 320;
        do { r = fast_step(in, out, out0_, bb, bc, lit, off, pack); } while (r == 0 && in <= in_fast && out <= out_fast);
        leave_fast(in, out, bb, bc, r);
    }
    // (measured and dropped: TWO members decoded in one loop, their steps interleaved -- the loop is bound by its instruction count, not by the
    // latency of its dependent table lookups: +9 % on FASTQ with noisy quality strings, -5 % with constant ones)

    static uint32_t symbol_entry(Kind kind, unsigned sym) {
        static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        if (kind == kKindPre) return (uint32_t)sym << 16;
        if (kind == kKindOff) return sym < 30 ? ((uint32_t)dist_base[sym] << 16) | ((uint32_t)dist_extra[sym] << 8) : 0xFFFFFFFFu;
        if (sym < 256) return kLiteral | ((uint32_t)sym << 16);
        if (sym == 256) return kEndOfBlock;
        if (sym < 286) return ((uint32_t)len_base[sym - 257] << 16) | ((uint32_t)len_extra[sym - 257] << 8);
        return 0xFFFFFFFFu;  // (286, 287: in the fixed code's alphabet, never valid in data)
    }
    static unsigned reverse_bits(unsigned code, unsigned len) {
        unsigned r = 0;
        for (unsigned i = 0; i < len; ++i) { r = (r << 1) | (code & 1u); code >>= 1; }
        return r;
    }
    // canonical Huffman code of lens[0 .. n) -> lookup table (primary `bits` wide + subtables); false: over-subscribed / no code at all
    static bool build(const uint8_t* lens, unsigned n, uint32_t* tab, unsigned bits, unsigned size, Kind kind) {
        unsigned count[16];
        memset(count, 0, sizeof count);
        for (unsigned i = 0; i < n; ++i) { if (lens[i] > 15) return false; count[lens[i]]++; }
        unsigned next_code[16];
        unsigned code = 0, left = 1;
        count[0] = 0;
        for (unsigned l = 1; l <= 15; ++l) {
            left <<= 1;
            if (count[l] > left) return false;  // over-subscribed
            left -= count[l];
            code = (code + count[l - 1]) << 1;
            next_code[l] = code;
        }
        // (incomplete codes are legal for a distance code with one symbol -- and tolerated in general: codes that do not exist decode
        // to an entry of length 0, which the decoder refuses)
        memset(tab, 0, (size_t)size * sizeof(uint32_t));
        uint8_t sub_max[1u << 11];
        bool any_long = false;
        // pass 1: short codes into the primary table; the longest code behind every primary prefix
        unsigned codes[320];
        for (unsigned s = 0; s < n; ++s) {
            const unsigned l = lens[s];
            if (!l) continue;
            const unsigned r = reverse_bits(next_code[l]++, l);
            codes[s] = r;
            const uint32_t ent = symbol_entry(kind, s);
            if (l <= bits) {
                if (ent == 0xFFFFFFFFu) continue;  // (a code for a symbol that must not occur: left out, refused when met)
                for (unsigned i = r; i < (1u << bits); i += 1u << l) tab[i] = ent | l;
            } else {
                if (!any_long) { memset(sub_max, 0, sizeof sub_max); any_long = true; }
                const unsigned p = r & ((1u << bits) - 1u);
                if (l > sub_max[p]) sub_max[p] = (uint8_t)l;
            }
        }
        if (!any_long) return true;
        // pass 2: a subtable per prefix with long codes, then the long codes into them
        unsigned next = 1u << bits;
        for (unsigned p = 0; p < (1u << bits); ++p) {
            if (!sub_max[p]) continue;
            const unsigned sb = sub_max[p] - bits;
            if (next + (1u << sb) > size) return false;
            tab[p] = (kind == kKindOff ? kOffSubtable : kSubtable) | ((uint32_t)next << 16) | (sb << 8) | bits;
            next += 1u << sb;
        }
        for (unsigned s = 0; s < n; ++s) {
            const unsigned l = lens[s];
            if (l <= bits) continue;
            const uint32_t ent = symbol_entry(kind, s);
            if (ent == 0xFFFFFFFFu) continue;
            const unsigned r = codes[s], p = r & ((1u << bits) - 1u);
            const uint32_t pe = tab[p];
            const unsigned start = (pe >> 16) & 0x1FFFu, sb = (pe >> 8) & 0xFu;
            for (unsigned i = r >> bits; i < (1u << sb); i += 1u << (l - bits)) tab[start + i] = ent | (l - bits);
        }
        return true;
    }
    // pk[i] = the literals whose codes lie completely inside the index bits i (at most four), from the primary table
    static void build_pack(const uint32_t* tab, uint64_t* pk) {
        for (unsigned idx = 0; idx < (1u << kLitBits); ++idx) {
            unsigned v = idx, avail = kLitBits, bits = 0, n = 0;
            uint64_t payload = 0;
            while (n < 4) {
                const uint32_t e = tab[v];  // (the bits above `avail` are zero: right for every code of at most `avail` bits)
                const unsigned l = e & 0xFFu;
                if (!(e & kLiteral) || l > avail) break;
                payload |= (uint64_t)((e >> 16) & 0xFFu) << (8 * n);
                ++n; bits += l; v >>= l; avail -= l;
            }
            pk[idx] = n ? (payload << 16) | ((uint64_t)n << 8) | bits : 0ull;
        }
    }
    void build_fixed() {
        uint8_t l[288 + 32];
        for (unsigned i = 0; i < 144; ++i) l[i] = 8;
        for (unsigned i = 144; i < 256; ++i) l[i] = 9;
        for (unsigned i = 256; i < 280; ++i) l[i] = 7;
        for (unsigned i = 280; i < 288; ++i) l[i] = 8;
        for (unsigned i = 0; i < 32; ++i) l[288 + i] = 5;
        (void)build(l, 288, fixed_lit, kLitBits, kLitSize, kKindLit);
        (void)build(l + 288, 32, fixed_off, kOffBits, kOffSize, kKindOff);
        build_pack(fixed_lit, fixed_pack);
        fixed_ready = true;
    }
};

}  // namespace sketchy

// ---- CRC-32 (the gzip polynomial) of a member's inflated bytes.  zlib's table-driven crc32() runs at ~1 GB/s per thread -- a third of the
// time of the decoder above; carry-less multiplication folds 64 bytes per step (Gopal et al., "Fast CRC Computation for Generic
// Polynomials Using PCLMULQDQ", the constants are those of the reflected polynomial 0xEDB88320): ~10 GB/s.  Run-time dispatch; the head
// that is not a multiple of 16 bytes (and machines without the instruction) go through zlib.
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
namespace sketchy {
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_clmul_blocks(const uint8_t* buf, size_t len /* multiple of 16, >= 64 */, uint32_t state) {
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)state));
    x0 = _mm_load_si128((const __m128i*)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i*)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i*)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i*)poly);
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
inline bool crc32_have_clmul() { static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1"); return have; }
}  // namespace sketchy
#endif
namespace sketchy {
// crc32 of buf[0 .. len) as zlib's crc32(0, buf, len)
inline uint32_t crc32_fast(const uint8_t* buf, size_t len) {
#if defined(__x86_64__)
    if (len >= 80 && crc32_have_clmul()) {
        const size_t head = len & 15u;                      // (zlib for the odd head, folds for the rest: a multiple of 16, at least 64)
        uint32_t c = head ? (uint32_t)crc32(0L, buf, (uInt)head) : 0u;
        return ~crc32_clmul_blocks(buf + head, len - head, ~c);
    }
#endif
    return (uint32_t)crc32(0L, buf, (uInt)len);
}
}  // namespace sketchy
