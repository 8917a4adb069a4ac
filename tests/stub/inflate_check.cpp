// tests/stub/inflate_check.cpp -- sketchy_amd/host/fast_inflate.hpp against zlib: every strategy / level / size class of raw-deflate streams
// round-trips byte for byte and refuses wrong output sizes; 20 000 mutated members neither crash nor write out of bounds (built with
// ASan + UBSan by tests/test_host_cpu.py); crc32_fast against zlib's crc32; `inflate_check speed`: GB/s beside zlib's inflate.
#include <zlib.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "fast_inflate.hpp"
static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& in, int level, int strategy) {
    z_stream z{}; deflateInit2(&z, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> out(deflateBound(&z, in.size()) + 64);
    z.next_in = const_cast<uint8_t*>(in.data()); z.avail_in = in.size(); z.next_out = out.data(); z.avail_out = out.size();
    deflate(&z, Z_FINISH); out.resize(z.total_out); deflateEnd(&z); return out;
}
static std::vector<uint8_t> fastq(size_t n_bytes, std::mt19937_64& rng, bool noisy) {
    std::vector<uint8_t> v; v.reserve(n_bytes + 4000);
    size_t r = 0;
    while (v.size() < n_bytes) {
        char hdr[64]; int h = snprintf(hdr, sizeof hdr, "@read_%zu len=1500\n", r++);
        v.insert(v.end(), hdr, hdr + h);
        for (int i = 0; i < 1500; ++i) v.push_back("ACGT"[rng() & 3]);
        v.push_back('\n'); v.push_back('+'); v.push_back('\n');
        for (int i = 0; i < 1500; ++i) v.push_back(noisy ? (uint8_t)(33 + (rng() % 41)) : 'I');
        v.push_back('\n');
    }
    return v;
}
int main(int argc, char** argv) {
    const bool speed = argc > 1;
    std::mt19937_64 rng(7);
    sketchy::FastInflate fi;
    int fails = 0, cases = 0;
    auto check = [&](const std::vector<uint8_t>& data, int level, int strategy, const char* what) {
        auto c = deflate_raw(data, level, strategy);
        std::vector<uint8_t> out(data.size() + 1, 0xAB);
        bool ok = fi.inflate_raw(c.data(), c.size(), out.data(), data.size());
        ok = ok && (data.empty() || memcmp(out.data(), data.data(), data.size()) == 0) && out[data.size()] == 0xAB;
        // wrong output sizes must be refused
        if (!data.empty()) { std::vector<uint8_t> o2(data.size() + 8); if (fi.inflate_raw(c.data(), c.size(), o2.data(), data.size() - 1)) ok = false; if (fi.inflate_raw(c.data(), c.size(), o2.data(), data.size() + 1)) ok = false; }
        ++cases;
        if (!ok) { ++fails; printf("FAIL %s level %d strategy %d size %zu\n", what, level, strategy, data.size()); }
    };
    for (size_t n : {0ul, 1ul, 2ul, 7ul, 100ul, 319ul, 320ul, 321ul, 1000ul, 65280ul, 65536ul, 300000ul}) {
        std::vector<uint8_t> rnd(n), zeros(n, 0), text(n), dna(n);
        for (auto& b : rnd) b = rng();
        for (size_t i = 0; i < n; ++i) { text[i] = "the quick brown fox jumps over the lazy dog\n"[i % 44]; dna[i] = "ACGT"[rng() & 3]; }
        auto fq = fastq(n, rng, true); fq.resize(n);
        auto fq2 = fastq(n, rng, false); fq2.resize(n);
        for (int level : {0, 1, 4, 6, 9})
            for (int strat : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE}) {
                check(rnd, level, strat, "random"); check(zeros, level, strat, "zeros"); check(text, level, strat, "text");
                check(dna, level, strat, "dna"); check(fq, level, strat, "fastq noisy"); check(fq2, level, strat, "fastq const");
            }
    }
    // skewed alphabets: long codes (15 bits) and subtables
    for (int t = 0; t < 20; ++t) {
        std::vector<uint8_t> v(200000);
        for (auto& b : v) { double u = (rng() % 1000000) / 1e6; int s = 0; double p = 0.5; while (u > p && s < 255) { u -= p; p *= (t % 2 ? 0.75 : 0.6); ++s; } b = (uint8_t)(s * 37 + t); }
        check(v, 6, Z_HUFFMAN_ONLY, "skewed"); check(v, 9, Z_DEFAULT_STRATEGY, "skewed");
    }
    // mutations: never crash, never write outside (ASan); result may be either
    auto fq = fastq(65280, rng, true); fq.resize(65280);
    auto c = deflate_raw(fq, 6, Z_DEFAULT_STRATEGY);
    size_t accepted = 0;
    for (int t = 0; t < 20000; ++t) {
        auto m = c;
        int k = 1 + rng() % 3;
        for (int i = 0; i < k; ++i) m[rng() % m.size()] ^= (uint8_t)(1u << (rng() % 8));
        if (t % 5 == 0) m.resize(rng() % m.size());
        std::vector<uint8_t> out(fq.size());
        if (fi.inflate_raw(m.data(), m.size(), out.data(), out.size())) ++accepted;
    }
    for (size_t n : {0ul, 1ul, 15ul, 16ul, 63ul, 64ul, 79ul, 80ul, 81ul, 95ul, 96ul, 127ul, 128ul, 1000ul, 65280ul, 65281ul, 65295ul}) {
        std::vector<uint8_t> v(n); for (auto& b : v) b = rng();
        ++cases;
        if (sketchy::crc32_fast(v.data(), n) != (uint32_t)crc32(0L, v.data(), n)) { ++fails; printf("FAIL crc32 size %zu\n", n); }
    }
    printf("cases %d fails %d; mutated members accepted (same length, caught by the CRC): %zu of 20000\n", cases, fails, accepted);
    if (!speed) return fails ? 1 : 0;
    // speed (./inflate_check speed)
    for (bool noisy : {true, false}) {
        auto data = fastq(64u << 20, rng, noisy);
        std::vector<std::vector<uint8_t>> members; std::vector<size_t> usz;
        for (size_t a = 0; a < data.size(); a += 65280) { size_t n = std::min<size_t>(65280, data.size() - a); std::vector<uint8_t> part(data.begin() + a, data.begin() + a + n); members.push_back(deflate_raw(part, 6, Z_DEFAULT_STRATEGY)); usz.push_back(n); }
        std::vector<uint8_t> out(65536);
        size_t comp = 0; for (auto& m : members) comp += m.size();
        auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < members.size(); ++i) if (!fi.inflate_raw(members[i].data(), members[i].size(), out.data(), usz[i])) { printf("speed: FAIL\n"); return 1; }
        double s1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        z_stream z{}; inflateInit2(&z, -15);
        t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < members.size(); ++i) { inflateReset(&z); z.next_in = members[i].data(); z.avail_in = members[i].size(); z.next_out = out.data(); z.avail_out = usz[i]; if (inflate(&z, Z_FINISH) != Z_STREAM_END) return 2; }
        double s2 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        t0 = std::chrono::steady_clock::now();
        unsigned long crc = 0; for (size_t a = 0; a < data.size(); a += 65280) crc ^= crc32(0, data.data() + a, std::min<size_t>(65280, data.size() - a));
        double s3 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%s quality: ratio %.2f; fast_inflate %.3f GB/s, zlib %.3f GB/s (x%.2f), crc32 %.2f GB/s (%lx)\n", noisy ? "noisy" : "constant", (double)data.size() / comp, data.size() / s1 / 1e9, data.size() / s2 / 1e9, s2 / s1, data.size() / s3 / 1e9, crc);
    }
    return fails ? 1 : 0;
}
