// formats.hpp -- the data formats either side of the hot path, as the C++ host needs them:
//   * Mash `.msh` reference sketches (un-packed Cap'n Proto, Mash's MinHash.capnp) -- what finch's
//     read_mash_file gives Sketchy::_read_sketch (src/sketchy.rs:497-536): only name, length64, numValidKmers,
//     comment and hashes64 of every reference plus kmerSize / hashSeed are used
//   * the genotype table (tab separated, header row; src/sketchy.rs:538-571)
//   * FASTA / FASTQ records, optionally gzip-compressed (needletail's parse_fastx_file, src/sketchy.rs:89-92)
// [UPSTREAM-RECALL] The MinHash.capnp field layout below is restated from the public schema (it is not in the
// reference tree and no Mash/capnp tool exists in this image to confirm it): struct MinHash = 3 data words
// {kmerSize u32 @0, windowSize u32 @4, minHashesPerWindow u32 @8, bools @12, error f32 @16, hashSeed u32 @20
// (default 42, stored XOR 42)} + 4 pointers {referenceListOld, locusList, alphabet, referenceList};
// ReferenceList = 1 pointer {references}; Reference = 3 data words {length u32 @0, length64 u64 @8,
// numValidKmers u64 @16} + 7 pointers {sequence, quality, name, comment, hashes32, hashes64, counts32}.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <memory>
#include <zlib.h>

#include "fast_inflate.hpp"
#include <atomic>
#include <thread>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace sketchy {

struct Sketch {  // the fields of finch::serialization::Sketch the path touches
    std::string name, comment;
    uint64_t seq_length = 0, num_valid_kmers = 0;
    std::vector<uint64_t> hashes;  // ascending
    uint32_t kmer_length = 0;
    uint64_t hash_seed = 0;
};

// ------------------------------------------------------------------ Cap'n Proto (read side)
class CapnpMessage {
  public:
    explicit CapnpMessage(std::vector<uint64_t>&& words_) : words(std::move(words_)) {
        if (words.empty()) throw std::runtime_error("empty capnp message");
        const uint32_t* h = reinterpret_cast<const uint32_t*>(words.data());
        const uint64_t nseg = (uint64_t)h[0] + 1;
        if (nseg > words.size() * 2) throw std::runtime_error("capnp segment table truncated");
        const uint64_t header_words = (4 + 4 * nseg + 7) / 8;
        uint64_t pos = header_words;
        for (uint64_t i = 0; i < nseg; ++i) {
            if (header_words * 2 <= 1 + i) throw std::runtime_error("capnp segment table truncated");
            const uint64_t len = h[1 + i];
            if (pos + len > words.size()) throw std::runtime_error("capnp segment exceeds file");
            seg_off.push_back(pos); seg_len.push_back(len);
            pos += len;
        }
    }
    struct Loc { uint32_t seg; uint64_t off; };  // a word inside a segment
    struct Obj {                                  // a resolved struct or list
        int kind = -1;                            // 0 struct, 1 list, -1 null
        Loc at{0, 0};                             // first word of the content
        uint32_t data_words = 0, ptr_words = 0;   // struct (or list element, composite)
        uint32_t elem_code = 0; uint64_t count = 0;
    };
    uint64_t word(Loc l) const {
        if (l.seg >= seg_off.size() || l.off >= seg_len[l.seg]) throw std::runtime_error("capnp pointer out of bounds");
        return words[seg_off[l.seg] + l.off];
    }
    Obj root() const { return follow({0, 0}); }
    // resolve the pointer stored at `p`
    Obj follow(Loc p) const {
        uint64_t w = word(p);
        Loc base = p;  // offsets are relative to the word after `base`
        if ((w & 3) == 2) {  // far pointer
            const bool dbl = (w >> 2) & 1;
            Loc pad{(uint32_t)(w >> 32), (uint64_t)((w >> 3) & 0x1FFFFFFF)};
            if (!dbl) { w = word(pad); base = pad; }
            else {
                const uint64_t far = word(pad), tag = word({pad.seg, pad.off + 1});
                Loc content{(uint32_t)(far >> 32), (uint64_t)((far >> 3) & 0x1FFFFFFF)};
                return decode(tag, content, true);
            }
        }
        if (w == 0) return Obj();
        const int64_t off = (int64_t)((int32_t)(uint32_t)(w & 0xFFFFFFFFu)) >> 2;
        Loc content{base.seg, (uint64_t)((int64_t)base.off + 1 + off)};
        return decode(w, content, false);
    }
    Obj decode(uint64_t w, Loc content, bool) const {
        Obj o;
        o.at = content;
        if ((w & 3) == 0) { o.kind = 0; o.data_words = (w >> 32) & 0xFFFF; o.ptr_words = (w >> 48) & 0xFFFF; }
        else if ((w & 3) == 1) {
            o.kind = 1; o.elem_code = (w >> 32) & 7; o.count = w >> 35;
            if (o.elem_code == 7) {  // composite: tag word first
                const uint64_t tag = word(content);
                const uint64_t total_words = w >> 35;     // (of a composite list the pointer holds the WORD count of the content)
                o.count = (tag >> 2) & 0x3FFFFFFF;
                o.data_words = (tag >> 32) & 0xFFFF; o.ptr_words = (tag >> 48) & 0xFFFF;
                o.at = {content.seg, content.off + 1};
                if (o.count * (uint64_t)(o.data_words + o.ptr_words) > total_words) throw std::runtime_error("capnp composite list larger than its pointer says");
                check_span(o, total_words);
            }
        } else throw std::runtime_error("unsupported capnp pointer kind");
        return o;
    }
    uint64_t data_u64(const Obj& s, uint32_t word_idx) const { return word_idx < s.data_words ? word({s.at.seg, s.at.off + word_idx}) : 0; }
    uint32_t data_u32(const Obj& s, uint32_t byte_off) const { return (uint32_t)(data_u64(s, byte_off / 8) >> (8 * (byte_off % 8))); }
    Obj ptr(const Obj& s, uint32_t idx) const {
        if (s.kind != 0 && !(s.kind == 1 && s.elem_code == 7)) return Obj();
        if (idx >= s.ptr_words) return Obj();
        return follow({s.at.seg, s.at.off + s.data_words + idx});
    }
    Obj element(const Obj& l, uint64_t i) const {  // composite list element as a struct
        if (l.kind != 1 || l.elem_code != 7 || i >= l.count) throw std::runtime_error("capnp list element out of range");
        Obj e; e.kind = 0; e.data_words = l.data_words; e.ptr_words = l.ptr_words;
        e.at = {l.at.seg, l.at.off + i * (l.data_words + l.ptr_words)};
        return e;
    }
    // (a list's content must lie inside its segment BEFORE anything is allocated for it: a flipped count must not become a
    // multi-gigabyte resize)
    void check_span(const Obj& l, uint64_t n_words) const {
        if (l.at.seg >= seg_off.size() || l.at.off > seg_len[l.at.seg] || n_words > seg_len[l.at.seg] - l.at.off)
            throw std::runtime_error("capnp list exceeds its segment");
    }
    std::string text(const Obj& l) const {
        if (l.kind != 1 || l.elem_code != 2 || l.count == 0) return std::string();
        check_span(l, (l.count + 7) / 8);
        std::string s((size_t)l.count - 1, '\0');
        for (uint64_t i = 0; i + 1 < l.count; ++i) s[i] = (char)(word({l.at.seg, l.at.off + i / 8}) >> (8 * (i % 8)));
        return s;
    }
    std::vector<uint64_t> list_u64(const Obj& l) const {
        std::vector<uint64_t> v;
        if (l.kind != 1 || l.elem_code != 5) return v;
        check_span(l, l.count);
        v.resize((size_t)l.count);
        for (uint64_t i = 0; i < l.count; ++i) v[i] = word({l.at.seg, l.at.off + i});
        return v;
    }
  private:
    std::vector<uint64_t> words;
    std::vector<uint64_t> seg_off, seg_len;
};

inline std::vector<uint64_t> read_words(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("failed to open file: " + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<uint64_t> w((size_t)(n + 7) / 8, 0);
    if (n > 0 && !f.read(reinterpret_cast<char*>(w.data()), n)) throw std::runtime_error("failed to read file: " + path);
    return w;
}

// finch read_mash_file + the parameter fix-up of src/sketchy.rs:513-530 (s := number of hashes)
inline std::vector<Sketch> read_mash_file(const std::string& path) {
    CapnpMessage m(read_words(path));
    const auto root = m.root();
    if (root.kind != 0) throw std::runtime_error("not a Mash sketch file: " + path);
    const uint32_t kmer = m.data_u32(root, 0);
    const uint32_t seed = m.data_u32(root, 20) ^ 42u;
    auto rl = m.ptr(root, 3);                          // referenceList
    if (rl.kind != 0) rl = m.ptr(root, 0);             // referenceListOld
    const auto refs = m.ptr(rl, 0);
    std::vector<Sketch> out;
    if (refs.kind != 1) return out;
    if (refs.elem_code != 7) throw std::runtime_error("not a Mash sketch file (reference list is not a struct list): " + path);
    out.reserve((size_t)refs.count);
    for (uint64_t i = 0; i < refs.count; ++i) {
        const auto r = m.element(refs, i);
        Sketch s;
        s.name = m.text(m.ptr(r, 2));
        s.comment = m.text(m.ptr(r, 3));
        s.seq_length = m.data_u64(r, 1);
        if (s.seq_length == 0) s.seq_length = m.data_u32(r, 0);
        s.num_valid_kmers = m.data_u64(r, 2);
        s.hashes = m.list_u64(m.ptr(r, 5));            // hashes64 only, as finch does
        // A genuine Mash sketch with k <= 16 stores its hashes as hashes32 (pointer 4); finch's reader takes hashes64 only and
        // would hand sketchy a collection of EMPTY sketches -- every read would score 0 against everything,
        // silently.  Say so instead.
        if (s.hashes.empty()) {
            const auto h32 = m.ptr(r, 4);
            if (h32.kind == 1 && h32.count > 0)
                throw std::runtime_error("sketch '" + s.name + "' in " + path + " stores 32-bit hashes only (hashes32: a Mash sketch with k <= 16); "
                                         "this reader, like finch's, takes hashes64 -- build the reference with `sketchy sketch` (64-bit hashes)");
        }
        s.kmer_length = kmer; s.hash_seed = seed;
        out.push_back(std::move(s));
    }
    return out;
}

// single-segment writer (tests, and `sketch`-style output later)
inline void write_mash_file(const std::string& path, const std::vector<Sketch>& sk, uint32_t kmer, uint32_t seed) {
    std::vector<uint64_t> w;
    auto struct_ptr = [](int64_t off, uint32_t dw, uint32_t pw) { return (uint64_t)((uint32_t)(off << 2)) | ((uint64_t)dw << 32) | ((uint64_t)pw << 48); };
    auto list_ptr = [](int64_t off, uint32_t code, uint64_t count) { return (uint64_t)((uint32_t)(off << 2) | 1u) | ((uint64_t)code << 32) | (count << 35); };
    w.push_back(0);                                    // root pointer (word 0)
    const size_t root = w.size(); w.resize(root + 3 + 4, 0);
    w[0] = struct_ptr(0, 3, 4);
    w[root + 0] = (uint64_t)kmer;                      // kmerSize @0, windowSize @4 = 0
    w[root + 2] = (uint64_t)(seed ^ 42u) << 32;        // hashSeed at byte 20
    const size_t rl = w.size(); w.resize(rl + 1, 0);   // ReferenceList {references}
    w[root + 3 + 3] = struct_ptr((int64_t)rl - (int64_t)(root + 3 + 3) - 1, 0, 1);
    const size_t tag = w.size();
    const uint64_t n = sk.size(), esz = 3 + 7;
    w.resize(tag + 1 + n * esz, 0);
    w[rl] = list_ptr((int64_t)tag - (int64_t)rl - 1, 7, n * esz);
    w[tag] = ((uint64_t)n << 2) | (3ull << 32) | (7ull << 48);
    for (uint64_t i = 0; i < n; ++i) {
        const size_t e = tag + 1 + i * esz;
        w[e + 0] = (uint32_t)std::min<uint64_t>(sk[i].seq_length, 0xFFFFFFFFu);
        w[e + 1] = sk[i].seq_length;
        w[e + 2] = sk[i].num_valid_kmers;
        auto put_text = [&](size_t pidx, const std::string& s) {
            const size_t at = w.size(); const uint64_t cnt = s.size() + 1;
            w.resize(at + (cnt + 7) / 8, 0);
            memcpy(&w[at], s.data(), s.size());
            w[e + 3 + pidx] = list_ptr((int64_t)at - (int64_t)(e + 3 + pidx) - 1, 2, cnt);
        };
        put_text(2, sk[i].name);
        put_text(3, sk[i].comment);
        const size_t at = w.size();
        w.insert(w.end(), sk[i].hashes.begin(), sk[i].hashes.end());
        w[e + 3 + 5] = list_ptr((int64_t)at - (int64_t)(e + 3 + 5) - 1, 5, sk[i].hashes.size());
    }
    if (w.size() - 1 >= (1ull << 29)) throw std::runtime_error("sketch too large for the single-segment writer");
    std::ofstream f(path, std::ios::binary);
    const uint32_t hdr[2] = {0u, (uint32_t)w.size()};
    f.write(reinterpret_cast<const char*>(hdr), 8);
    f.write(reinterpret_cast<const char*>(w.data()), (std::streamsize)w.size() * 8);
    if (!f) throw std::runtime_error("failed to write " + path);
}

// ------------------------------------------------------------------ genotype table
struct Genotypes {
    std::string header;                                          // columns 1.. joined by tab (src/sketchy.rs:557)
    std::unordered_map<std::string, std::vector<std::string>> map;  // name -> columns 1..   (:561-571)
    size_t rows = 0;
};
inline std::vector<std::string> split_tab(const std::string& line) {
    std::vector<std::string> f; size_t a = 0;
    for (;;) { size_t b = line.find('\t', a); f.push_back(line.substr(a, b == std::string::npos ? b : b - a)); if (b == std::string::npos) break; a = b + 1; }
    return f;
}
inline std::string join_tab(const std::vector<std::string>& v, size_t from = 0) {
    std::string s; for (size_t i = from; i < v.size(); ++i) { if (i > from) s += '\t'; s += v[i]; } return s;
}
inline Genotypes read_genotypes(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("failed to open genotype file: " + path);
    Genotypes g; std::string line; bool first = true;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        auto cols = split_tab(line);
        if (first) { g.header = join_tab(cols, 1); first = false; continue; }
        g.map[cols[0]] = std::vector<std::string>(cols.begin() + 1, cols.end());
        g.rows++;
    }
    return g;
}

// ------------------------------------------------------------------ FASTA / FASTQ (+gzip)
// Block-buffered: 4 MB gzread blocks, lines found with memchr (the per-line gzgets of the first version topped out
// far below what the device path consumes).
class FastxReader {
  public:
    explicit FastxReader(const std::string& path) : buf(4u << 20) {  // "-" = stdin; gzip is detected by zlib itself
        gz = path == "-" ? gzdopen(0, "rb") : gzopen(path.c_str(), "rb");
        if (!gz) throw std::runtime_error("failed to open Fastx file: " + path);
        gzbuffer(gz, 1 << 20);
    }
    ~FastxReader() { if (gz) gzclose(gz); }
    FastxReader(const FastxReader&) = delete;
    FastxReader& operator=(const FastxReader&) = delete;
    // next record's sequence bytes (as in the file: case kept, line breaks of multi-line FASTA removed)
    bool next(std::string& seq) {
        while (!have_pending) { if (!getline(pending)) return false; have_pending = !pending.empty(); }
        seq.clear();
        if (pending[0] == '>') {
            have_pending = false;
            while (getline(line)) { if (!line.empty() && line[0] == '>') { pending.swap(line); have_pending = true; break; } seq += line; }
            return true;
        }
        if (pending[0] == '@') {
            have_pending = false;
            if (!getline(seq)) throw std::runtime_error("truncated FASTQ record");
            if (!getline(line) || line.empty() || line[0] != '+') throw std::runtime_error("malformed FASTQ record (no '+' line)");
            if (!getline(line)) throw std::runtime_error("truncated FASTQ record");
            return true;
        }
        throw std::runtime_error("input is neither FASTA nor FASTQ");
    }
  private:
    bool fill() {
        if (eof) return false;
        const int n = gzread(gz, buf.data(), (unsigned)buf.size());
        if (n < 0) throw std::runtime_error("read error in Fastx input");
        pos = 0; end = (size_t)n;
        if (n == 0) eof = true;
        return n > 0;
    }
    // one line without its terminator (\n or \r\n); false at end of input with nothing read
    bool getline(std::string& out) {
        out.clear();
        bool any = false;
        for (;;) {
            if (pos == end && !fill()) break;
            any = true;
            const char* p = buf.data() + pos;
            const char* nl = static_cast<const char*>(memchr(p, '\n', end - pos));
            if (nl) {
                out.append(p, (size_t)(nl - p));
                pos += (size_t)(nl - p) + 1;
                if (!out.empty() && out.back() == '\r') out.pop_back();
                return true;
            }
            out.append(p, end - pos);
            pos = end;
        }
        if (any && !out.empty() && out.back() == '\r') out.pop_back();
        return any && !out.empty();
    }
    gzFile gz = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false, have_pending = false;
    std::string pending, line;
};

// ------------------------------------------------------------------ FASTX in memory: the parallel front-end's view
// An uncompressed regular file is mapped and cut into CHUNKS at record boundaries; every chunk is parsed by whichever thread
// picks it up (sketchy_host.cpp).  needletail reads one record at a time on the thread that scores it (src/sketchy.rs:328-333);
// the records, their order and their bytes are the same.
bool& bgzf_force_zlib();
class MappedFile {
  public:
    MappedFile() = default;
    ~MappedFile();
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    // false: not a regular file (a pipe, stdin) or empty / compressed -- the caller streams it instead
    bool open(const std::string& path);
    // A BGZF file (bgzip, htslib: gzip members of at most 64 KB that carry their compressed size in a "BC" extra field, so the member
    // boundaries are known without inflating anything) is inflated by `threads` threads into an anonymous mapping and then looks
    // like an uncompressed mapped file.  false: not BGZF (plain gzip has ONE member: a sequential stream), or larger than a quarter
    // of the machine's memory uncompressed -- the caller streams it.  needletail picks its decoder from the magic bytes just the same
    // (src/sketchy.rs:89-92).
    bool open_bgzf(const std::string& path, unsigned threads);
    bool inflated() const { return anon; }
    bool eof_marker = true;  // open_bgzf: the file ends with BGZF's empty end-of-file member
    const char* data() const { return base; }
    size_t size() const { return len; }
    // map the pages of [a, b) now (MADV_POPULATE_READ where the kernel has it; else a hint): a parser thread calls it for its chunk
    void prefetch(size_t a, size_t b) const;
  private:
    const char* base = nullptr;
    size_t len = 0;
    bool anon = false;  // an anonymous mapping holding inflated data (open_bgzf)
};

// first byte of the first record that starts at or behind `pos` (a line start), or `end` when there is none.
// FASTQ (four-line records, as needletail's fast path reads them): a line that starts with '@' whose second-next line starts
// with '+' -- a quality line may start with '@' too, but then the line after it is the next header and the one after that a
// sequence, which cannot start with '+'.  FASTA: a line that starts with '>'.
inline const char* next_record_start(const char* begin, const char* pos, const char* end, bool fastq) {
    const char* p = pos;
    if (p > begin && p[-1] != '\n') {  // inside a line: go to the next line start
        p = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
        if (!p) return end;
        ++p;
    }
    while (p < end) {
        if (!fastq) {
            if (*p == '>') return p;
        } else if (*p == '@') {
            const char* l1 = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
            const char* l2 = l1 ? static_cast<const char*>(memchr(l1 + 1, '\n', (size_t)(end - l1 - 1))) : nullptr;
            if (l2 && l2 + 1 < end && l2[1] == '+') return p;
            if (!l2) return end;  // (no complete record behind p)
        }
        p = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
        if (!p) return end;
        ++p;
    }
    return end;
}

}  // namespace sketchy

inline sketchy::MappedFile::~MappedFile() { if (base) munmap(const_cast<char*>(base), len); }
inline bool sketchy::MappedFile::open(const std::string& path) {
    if (path == "-") return false;
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("failed to open Fastx file: " + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) { ::close(fd); return false; }
    unsigned char magic[2] = {0, 0};
    if (pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b)) { ::close(fd); return false; }  // gzip: streamed
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) return false;
    (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    base = static_cast<const char*>(m); len = (size_t)st.st_size;
    return true;
}
// (measurement aid of tools/bgzf_rate.cpp: every member through zlib, as in round 5)
inline bool& sketchy::bgzf_force_zlib() { static bool f = false; return f; }
inline bool sketchy::MappedFile::open_bgzf(const std::string& path, unsigned threads) {
    if (path == "-") return false;
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 28) { ::close(fd); return false; }
    void* cm = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (cm == MAP_FAILED) return false;
    const unsigned char* c = static_cast<const unsigned char*>(cm);
    const size_t clen = (size_t)st.st_size;
    struct Block { size_t coff, csize, uoff, usize; uint32_t crc; };
    std::vector<Block> blocks;
    size_t pos = 0, total = 0;
    bool ok = true;
    while (pos < clen) {  // every member: 1f 8b 08 04 .. XLEN, extra subfield 'B' 'C' 02 00 BSIZE (member size - 1), ..., CRC32, ISIZE
        if (clen - pos < 18 || c[pos] != 0x1f || c[pos + 1] != 0x8b || c[pos + 2] != 8 || !(c[pos + 3] & 4)) { ok = false; break; }
        const size_t xlen = c[pos + 10] | ((size_t)c[pos + 11] << 8);
        if (clen - pos < 12 + xlen + 8) { ok = false; break; }
        size_t bsize = 0;
        for (size_t x = pos + 12; x + 4 <= pos + 12 + xlen;) {
            const size_t slen = c[x + 2] | ((size_t)c[x + 3] << 8);
            if (c[x] == 'B' && c[x + 1] == 'C' && slen == 2 && x + 6 <= pos + 12 + xlen) bsize = (c[x + 4] | ((size_t)c[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || pos + bsize > clen) { ok = false; break; }
        const unsigned char* tail = c + pos + bsize - 4;
        const size_t isize = tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
        if (isize > (1u << 16)) { ok = false; break; }
        const unsigned char* ct = tail - 4;  // (the member's trailer: CRC32 of the inflated bytes, then ISIZE)
        blocks.push_back(Block{pos + 12 + xlen, bsize - 12 - xlen - 8, total, isize,
                               (uint32_t)ct[0] | ((uint32_t)ct[1] << 8) | ((uint32_t)ct[2] << 16) | ((uint32_t)ct[3] << 24)});
        total += isize;
        pos += bsize;
    }
    long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
    if (!ok || total < 2 || (pages > 0 && psz > 0 && total > (size_t)pages * (size_t)psz / 4)) { munmap(cm, clen); return false; }
    void* um = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (um == MAP_FAILED) { munmap(cm, clen); return false; }
#ifdef MADV_HUGEPAGE
    (void)madvise(um, total, MADV_HUGEPAGE);  // (first touch by all threads at once: 2 MB pages where the kernel gives them -- 512 x fewer faults)
#endif
    char* u = static_cast<char*>(um);
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    // every member: fast_inflate.hpp's decoder (whole member in, whole block out: 1.5-2 x zlib's inflate per thread); a member it
    // refuses goes through zlib once more, so that a stream only zlib understands still loads and a corrupt one fails with zlib's
    // verdict; then the CRC32 of the inflated bytes against the member's trailer (round 5 compared the length only: a corrupted block
    // that still inflated to the right length was scored silently -- needletail / flate2 reject it, src/sketchy.rs:89-92)
    auto work = [&] {
        std::unique_ptr<FastInflate> fi(new FastInflate);
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) { bad = true; return; }  // (raw deflate: header and trailer were walked above)
        for (;;) {
            const size_t b0 = next.fetch_add(64);
            if (b0 >= blocks.size() || bad) break;
            {
                // the pages this grab writes to (64 members = ~4 MB), populated in ONE call: sixteen threads taking a page fault every
                // 4 KB of a fresh anonymous mapping queue on the address space's lock -- round 5's 16 threads inflated no faster than 8
                const size_t b1 = std::min(blocks.size(), b0 + 64) - 1;
                const size_t lo = blocks[b0].uoff & ~(size_t)4095, hi = blocks[b1].uoff + blocks[b1].usize;
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
                if (hi > lo) (void)madvise(u + lo, hi - lo, MADV_POPULATE_WRITE);  // (kernels before 5.14: EINVAL, the faults happen one by one as before)
            }
            for (size_t b = b0; b < std::min(blocks.size(), b0 + 64); ++b) {
                const Block& k = blocks[b];
                if (k.usize == 0) continue;
                unsigned char* dst = reinterpret_cast<unsigned char*>(u + k.uoff);
                if (bgzf_force_zlib() || !fi->inflate_raw(c + k.coff, k.csize, dst, k.usize)) {
                    inflateReset(&z);
                    z.next_in = const_cast<unsigned char*>(c + k.coff); z.avail_in = (unsigned)k.csize;
                    z.next_out = dst; z.avail_out = (unsigned)k.usize;
                    if (inflate(&z, Z_FINISH) != Z_STREAM_END || z.avail_out != 0) { bad = true; break; }
                }
                if (crc32_fast(dst, k.usize) != k.crc) { bad = true; break; }
            }
        }
        inflateEnd(&z);
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < std::max(1u, threads); ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    munmap(cm, clen);
    if (bad) { munmap(um, total); throw std::runtime_error("failed to inflate BGZF file (corrupt member or CRC mismatch): " + path); }
    // (bgzip ends every file with an empty member: without it the file was probably cut at a block boundary -- still scored, said aloud)
    eof_marker = !blocks.empty() && blocks.back().usize == 0;
    base = u; len = total; anon = true;
    return true;
}
inline void sketchy::MappedFile::prefetch(size_t a, size_t b) const {
    if (!base || b <= a || anon) return;  // (inflated data is resident already)
    const size_t page = 4096, lo = a & ~(page - 1);
    char* p = const_cast<char*>(base) + lo;
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
    if (madvise(p, b - lo, MADV_POPULATE_READ) != 0) (void)madvise(p, b - lo, MADV_WILLNEED);
}
