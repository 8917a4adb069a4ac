// sketchy_host.cpp -- C++ host above the C ABI, mirroring the reference's interface for the predict path
// (Rust is unavailable in this image; the reference is compiled code, so the host is C++).
//
//   sketchy::PredictConfig            src/sketchy.rs:43-50
//   Sketchy::predict                  src/sketchy.rs:66-124   (reference + genotypes + FASTX, header line, mode switch)
//   Sketchy::_sum_of_shared_hashes    src/sketchy.rs:317-356  (streaming: rows after every read)  -> skx_stream_push
//   Sketchy::_shared_hashes           src/sketchy.rs:281-315  (offline: one pooled sketch)         -> skx_sketch_reads + skx_common_hashes
//   Sketchy::_print_results           src/sketchy.rs:358-402  (rows / consensus)
//   Sketchy::shared                   src/sketchy.rs:238-279                                       -> skx_common_hashes
//   Sketchy::info (names only)        src/sketchy.rs:172-208
//   Sketchy::check                    src/sketchy.rs:212-236
//   Sketchy::sketch + _sketch_files   src/sketchy.rs:128-167, :465-494  (genome files -> Mash .msh)      -> skx_sketch_reads
// Command line: the reference's flag names and defaults (src/cli.rs:25-132).
//
// Streaming (`predict -s`) runs as a pipeline of four stages (the reference reads, scores and prints one record at a time
// on one thread, src/sketchy.rs:328-354):
//   parse   N threads: an uncompressed input file is mapped and cut at record boundaries into chunks of ~batch reads; a
//           thread parses its chunk and PACKS the bases as it goes (4 bits each, skx_pack_bases) into a page-locked slot.
//           gzip input and stdin are parsed by one thread (the stream is sequential), everything behind it is the same
//   device  one thread: skx_stream_submit, in file order -- the copy of batch i + 1 overlaps the kernels of batch i, up to
//           four batches share one scan of the reference (include/sketchy_hip.h)
//   format  M threads: the rows of a finished batch as text (src/sketchy.rs:389-400)
//   write   in file order, by whichever format thread completes the next batch
// Rows are the reference's, byte for byte; what changes is who does what when.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <exception>
#include <iostream>
#include <map>
#include <mutex>
#include <optional>
#include <sched.h>
#include <thread>

#include "formats.hpp"
#include "sketchy_hip.h"

namespace sketchy {

struct PredictConfig { size_t top = 1, limit = 0; bool stream = false, consensus = false, header = false;
                       // (not in the reference: how the streaming pipeline is run, and whether it says how fast it was)
                       size_t threads = 0; bool timing = false, pin = false; };

struct SketchyError : std::runtime_error { using std::runtime_error::runtime_error; };

static void hip_check(int rc, const char* what) {
    if (rc != SKX_OK) throw SketchyError(std::string(what) + ": " + skx_last_error());
}

class Sketchy {
  public:
    explicit Sketchy(int device = 0, size_t batch_reads = 16384) : device_(device), batch_(batch_reads) {}

    void predict(const std::optional<std::string>& fastx, const std::string& reference, const std::string& genotypes,
                 const PredictConfig& config, std::ostream& out) {
        if (config.consensus && config.top % 2 != 1)  // src/sketchy.rs:74-79
            throw SketchyError("--top must be an odd number when using --consensus");
        const auto sketches = read_sketch(reference);
        if (sketches.empty()) throw SketchyError("reference sketch file holds no sketches");
        const auto geno = read_genotypes(genotypes);
        for (const auto& s : sketches)  // the reference panics on a missing name (geno_map[&name], :308/:345)
            if (!geno.map.count(s.name)) throw SketchyError("reference sketch " + s.name + " has no row in the genotype table");
        if (config.top > sketches.size()) throw SketchyError("--top exceeds the number of reference sketches");
        if (config.header) out << "reads\tsketch_id\tshared_hashes\t" << geno.header << "\n";  // :99-101
        Ref ref(sketches, device_);
        if (config.stream) sum_of_shared_hashes(fastx ? *fastx : std::string("-"), sketches, ref, geno, config, out);
        else { FastxReader reader(fastx ? *fastx : std::string("-")); shared_hashes(reader, sketches, ref, geno, config, out); }
    }

    // `sketchy shared`: every reference x query pair, "ref query common" (src/sketchy.rs:251-276)
    void shared(const std::string& reference, const std::string& query, std::ostream& out) {
        const auto refs = read_sketch(reference), qs = read_sketch(query);
        if (refs.empty() || qs.empty()) throw SketchyError("empty sketch file");
        Ref ref(refs, device_);
        uint32_t stride = 1;
        for (const auto& q : qs) stride = std::max<uint32_t>(stride, (uint32_t)q.hashes.size());
        std::vector<uint64_t> flat(qs.size() * (size_t)stride, 0);
        std::vector<uint32_t> len(qs.size());
        for (size_t i = 0; i < qs.size(); ++i) {
            std::copy(qs[i].hashes.begin(), qs[i].hashes.end(), flat.begin() + i * stride);
            len[i] = (uint32_t)qs[i].hashes.size();
        }
        std::vector<uint32_t> common(qs.size() * refs.size());
        hip_check(skx_common_hashes(ref.h, flat.data(), len.data(), (uint32_t)qs.size(), stride, common.data()), "shared");
        for (size_t r = 0; r < refs.size(); ++r)
            for (size_t q = 0; q < qs.size(); ++q) {
                if (refs[r].kmer_length != qs[q].kmer_length || refs[r].hash_seed != qs[q].hash_seed)
                    throw SketchyError("reference (" + refs[r].name + ") does not match query (" + qs[q].name + ")");
                out << refs[r].name << " " << qs[q].name << " " << common[q * refs.size() + r] << "\n";
            }
    }

    // `sketchy check` (src/sketchy.rs:212-236): the per-row identifier comparison builds an error and discards it
    // (:219-228), so only the size check has an effect -- kept that way
    void check(const std::string& reference, const std::string& genotypes, std::ostream& out) {
        const auto sk = read_sketch(reference);
        const auto geno = read_genotypes(genotypes);
        if (sk.size() != geno.rows) throw SketchyError("reference sketch and genotype table must have the same length");
        out << "ok\n";
    }

    void info(const std::string& input, bool params, std::ostream& out) {
        const auto sk = read_sketch(input);
        if (sk.empty()) throw SketchyError("empty sketch file");
        if (params) { out << "type=mash sketch_size=" << sk[0].hashes.size() << " kmer_size=" << sk[0].kmer_length << " seed=" << sk[0].hash_seed << "\n"; return; }
        for (const auto& s : sk) out << s.name << " " << s.seq_length << " " << s.hashes.size() << "\n";
    }

    // `sketchy sketch` (src/sketchy.rs:128-167; _sketch_files :465-494): one Mash sketch per input FILE over all of its
    // records (k-mers never span records), name = file name (:484).  Every record is sketched on the device
    // (skx_sketch_reads: genomes take the block-per-read + segmented-sort path); the bottom-s of a file is the
    // bottom-s of the union of its records' bottom-s sketches, merged here.
    void sketch(const std::vector<std::string>& files, const std::string& output, size_t sketch_size, uint32_t kmer, uint64_t seed) {
        const auto dot = output.rfind('.');
        const std::string ext = dot == std::string::npos ? "" : output.substr(dot + 1);
        if (ext == "fsh") throw SketchyError("Finch scaled sketches (.fsh) are outside the accelerated path");
        if (ext != "msh") throw SketchyError("output sketch file must have Mash (.msh) or Finch (.fsh) extension");  // :573-600
        if (sketch_size < 1) throw SketchyError("sketch size must be at least 1");
        if (kmer < 1 || kmer > SKX_MAX_K) throw SketchyError("k-mer size must be between 1 and " + std::to_string(SKX_MAX_K));
        if (seed > 0xFFFFFFFFull) throw SketchyError("the Mash format stores a 32-bit hash seed");
        std::vector<Sketch> out;
        for (const auto& file : files) {
            FastxReader reader(file);
            Sketch sk;
            const auto slash = file.find_last_of('/');
            sk.name = slash == std::string::npos ? file : file.substr(slash + 1);
            sk.kmer_length = kmer; sk.hash_seed = seed;
            Batch b; std::string seq;
            std::vector<uint64_t> rows, merged; std::vector<uint32_t> len;
            auto flush = [&]() {
                if (b.n() == 0) return;
                rows.assign(b.n() * sketch_size, 0); len.assign(b.n(), 0);
                hip_check(skx_sketch_reads(device_, kmer, seed, (uint32_t)sketch_size, b.bases.data(), b.offsets.data(), (uint32_t)b.n(), rows.data(), len.data()), "sketch");
                for (size_t r = 0; r < b.n(); ++r) {
                    merged.clear();
                    std::set_union(sk.hashes.begin(), sk.hashes.end(), rows.begin() + r * sketch_size, rows.begin() + r * sketch_size + len[r], std::back_inserter(merged));
                    if (merged.size() > sketch_size) merged.resize(sketch_size);
                    sk.hashes.swap(merged);
                }
                b.clear();
            };
            while (reader.next(seq)) {
                sk.seq_length += seq.size();                       // [UPSTREAM-RECALL] finch: total bases of the records
                sk.num_valid_kmers += count_valid_kmers(seq, kmer);  // [UPSTREAM-RECALL] finch: k-mers pushed to the sketcher
                if (b.n() && (b.bases.size() + seq.size() > (1ull << 30) || b.n() * sketch_size > (1ull << 24))) flush();
                if (seq.size() >= (1ull << 32)) throw SketchyError("a record of " + file + " exceeds 4 Gbases");
                b.add(seq);
            }
            flush();
            out.push_back(std::move(sk));
        }
        write_mash_file(output, out, kmer, (uint32_t)seed);
    }
    // windows of k bases that are all A/C/G/T/U (any case), as needletail's normalize + canonical_kmers see them
    static uint64_t count_valid_kmers(const std::string& seq, uint32_t k) {
        uint64_t n = 0; uint32_t run = 0;
        for (unsigned char c : seq) {
            if (c == ' ' || c == '\t' || c == '\r' || c == '\n') continue;
            const unsigned char u = c & 0xDF;
            run = (u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'U') ? run + 1 : 0;
            if (run >= k) ++n;
        }
        return n;
    }

    static std::vector<Sketch> read_sketch(const std::string& path) {  // src/sketchy.rs:497-536
        const auto dot = path.rfind('.');
        const std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
        if (ext == "msh") return read_mash_file(path);
        if (ext == "fsh") throw SketchyError("Finch scaled sketches (.fsh) are outside the accelerated path");
        throw SketchyError("reference sketch file must have Mash (.msh) or Finch (.fsh) extension");
    }

  private:
    struct Ref {  // the reference collection on the device
        skx_ref* h = nullptr; uint32_t k = 0, s = 0; uint64_t seed = 0;
        Ref(const std::vector<Sketch>& sk, int device) {
            k = sk[0].kmer_length; seed = sk[0].hash_seed;
            s = (uint32_t)sk[0].hashes.size();       // src/sketchy.rs:82 / :520-527: read sketch size := |sketch 0|
            uint32_t stride = s;
            for (const auto& x : sk) stride = std::max<uint32_t>(stride, (uint32_t)x.hashes.size());
            if (stride == 0) throw SketchyError("reference sketches are empty");
            std::vector<uint64_t> flat((size_t)stride * sk.size(), ~0ull);
            std::vector<uint32_t> len(sk.size());
            for (size_t g = 0; g < sk.size(); ++g) {
                std::copy(sk[g].hashes.begin(), sk[g].hashes.end(), flat.begin() + g * (size_t)stride);
                len[g] = (uint32_t)sk[g].hashes.size();
            }
            stride_ = stride;
            // columns may be longer than sketch 0; the read sketch size stays s (reference behaviour)
            if (s == 0) throw SketchyError("the first reference sketch is empty (it defines the read sketch size)");
            hip_check(skx_ref_create(&h, device, k, seed, s, stride, (uint32_t)sk.size(), flat.data(), len.data()), "reference upload");
        }
        ~Ref() { skx_ref_destroy(h); }
        uint32_t stride_ = 0;
    };

    struct Batch { std::vector<uint8_t> bases; std::vector<uint64_t> offsets{0}; size_t n() const { return offsets.size() - 1; }
                   void add(const std::string& seq) { bases.insert(bases.end(), seq.begin(), seq.end()); offsets.push_back(bases.size()); }
                   void clear() { bases.clear(); offsets.assign(1, 0); } };

    // ---- streaming pipeline
    // one batch travelling parse -> device -> format and back: page-locked (skx_host_alloc) packed bases, offsets (in bases),
    // rows.  Slot i % n_slots carries chunk i: a parser waits for its slot to come back from the batch n_slots chunks before --
    // which only depends on OLDER chunks, so the ring cannot deadlock.
    struct Slot {
        uint8_t* packed = nullptr; uint64_t* offsets = nullptr; uint32_t* idx = nullptr; uint64_t* sum = nullptr;
        size_t n = 0, first_read = 1;
        // what did not fit the slot (a chunk with far more / longer reads than the first records promised): heap batches the
        // device thread sends through its spill slot, one by one (rare, slow, correct)
        struct Extra { std::vector<uint8_t> packed; std::vector<uint64_t> offsets{0}; };
        std::vector<Extra> extra;
        int state = 0;           // 0 free, 1 parsed
        bool last = false;       // the input ended in (or before) this chunk
    };
    static unsigned usable_threads() {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)CPU_COUNT(&set));
        std::ifstream f("/sys/fs/cgroup/cpu.max");  // (the container's quota: hardware_concurrency reports the machine)
        std::string quota; long long period = 0;
        if (f >> quota >> period && quota != "max" && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, atoll(quota.c_str()) / period));
        return std::max(1u, n);
    }
    // --pin: keep the process on CPUs next to the device -- the first `want` of /sys/bus/pci/devices/<device>/local_cpulist.  The
    // parser threads fill page-locked buffers the device's DMA engines read, and on a two-socket box threads float over both
    // sockets.  Off by default: measured on the MI355X boxes (2 x EPYC 9575F, `predict -s` at C2) what matters more is that the
    // parsers sit next to the INPUT's pages -- 26 M reads/s with the whole job (the writer of the file included) on either
    // socket, 22 M floating, 18-22 M with only this process pinned to the device's node while the file's pages lay elsewhere.
    // Returns how many CPUs the process was confined to (0: nothing was changed).
    static unsigned pin_near_device(int device, unsigned want) {
        char bus[64] = {0};
        if (skx_device_pci_bus_id(device, bus, sizeof bus) != SKX_OK) return 0;
        std::string id(bus);
        for (auto& c : id) c = (char)tolower((unsigned char)c);
        std::ifstream f("/sys/bus/pci/devices/" + id + "/local_cpulist");
        std::string list;
        if (!(f >> list) || list.empty()) return 0;
        cpu_set_t allowed, set;
        if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return 0;
        CPU_ZERO(&set);
        unsigned n = 0;
        size_t i = 0;
        while (i < list.size() && n < want) {  // "0-63,128-191"
            size_t j = list.find(',', i);
            if (j == std::string::npos) j = list.size();
            const std::string part = list.substr(i, j - i);
            const size_t dash = part.find('-');
            const long a = atol(part.c_str()), b = dash == std::string::npos ? a : atol(part.c_str() + dash + 1);
            for (long c = a; c <= b && n < want; ++c)
                if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, &set); ++n; }
            i = j + 1;
        }
        if (n < 2 || sched_setaffinity(0, sizeof set, &set) != 0) return 0;
        return n;
    }
    static char* put_u64(char* p, uint64_t v) {  // decimal, no terminator; returns the end
        char tmp[24]; int n = 0;
        do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) *p++ = tmp[--n];
        return p;
    }

    // streaming mode
    void sum_of_shared_hashes(const std::string& path, const std::vector<Sketch>& sketches, Ref& ref, const Genotypes& geno,
                              const PredictConfig& config, std::ostream& out) {
        using clock = std::chrono::steady_clock;
        const auto t_begin = clock::now();
        MappedFile map;
        // (a BGZF file -- gzip members with their sizes in the header -- is inflated by all threads at once and then cut and parsed
        // like an uncompressed one; plain gzip is one sequential stream)
        const bool mapped = map.open(path) || map.open_bgzf(path, config.threads ? (unsigned)config.threads : std::min(usable_threads(), 22u));
        const double open_s = std::chrono::duration<double>(clock::now() - t_begin).count();  // mapping the input (BGZF: inflating all of it)
        if (mapped && map.inflated() && !map.eof_marker)
            fprintf(stderr, "sketchy-hip: warning: %s does not end with BGZF's end-of-file block (truncated at a block boundary?)\n", path.c_str());
        const char *fbegin = mapped ? map.data() : nullptr, *fend = mapped ? map.data() + map.size() : nullptr;
        bool fastq = false;
        size_t chunk_bytes = 0;
        std::vector<size_t> cuts;  // chunk i = [cuts[i], cuts[i + 1]) of the mapped file
        const size_t want_reads = std::max<size_t>(1, batch_);
        // Threads: every parser / formatter thread owns a page-locked slot (~100 MB at the default batch of 65 536 x 1.5 kb FASTQ), and the
        // front-end saturates at 12-16 parsers (tools/frontend_rate.sh) -- an unrestricted 256-thread host must not page-lock 27 GB before
        // its first read.  Without -j: at most kAutoThreads; with -j: what was asked for, the parsers still capped at kMaxParsers.
        constexpr unsigned kAutoThreads = 22, kMaxParsers = 32;
        const unsigned hw = config.threads ? (unsigned)config.threads : std::min(usable_threads(), kAutoThreads);
        const unsigned n_format = std::max(1u, std::min(4u, hw / 4));
        const unsigned n_parse = mapped ? std::min(kMaxParsers, std::max(1u, hw > n_format + 2 ? hw - n_format - 2 : 1u)) : 1u;
        const unsigned pinned = config.pin ? pin_near_device(device_, std::max(hw, 2u)) : 0u;
        if (mapped) {
            const char* p = fbegin;
            while (p < fend && (*p == '\n' || *p == '\r')) ++p;
            if (p < fend && *p != '@' && *p != '>') throw SketchyError("input is neither FASTA nor FASTQ");
            fastq = p < fend && *p == '@';
            // bytes per record from the first records; chunks of ~batch records, cut at record boundaries
            size_t n0 = 0; const char* q = p;
            while (q < fend && n0 < 64) { const char* nx = next_record_start(fbegin, q + 1, fend, fastq); ++n0; q = nx; }
            const size_t per = n0 ? std::max<size_t>(4, (size_t)(q - p) / n0) : 4;
            chunk_bytes = std::max<size_t>(64, per * want_reads);
            cuts.push_back((size_t)(p - fbegin));
            while (cuts.back() < map.size()) {
                // (the first chunks grow from 1 / n_parse of a batch to a whole one: the parser threads all start at once, and with
                // equal chunks they would all hand over their first batch at the same moment, a whole chunk's parse time in)
                const size_t k = cuts.size() - 1;
                const size_t this_chunk = k < n_parse ? std::max<size_t>(64, chunk_bytes * (k + 1) / n_parse) : chunk_bytes;
                const size_t nominal = cuts.back() + this_chunk;
                const size_t nx = nominal >= map.size() ? map.size() : (size_t)(next_record_start(fbegin, fbegin + nominal, fend, fastq) - fbegin);
                cuts.push_back(std::max(nx, cuts.back() + 1));
            }
        } else {
            chunk_bytes = std::max<size_t>(64, (size_t)3200 * want_reads);  // (what a batch may hold: reads x a long-read-ish record)
        }
        const size_t cap_reads = 2 * want_reads + 64;            // a slot's reads ...
        size_t cap_bases = std::max<size_t>(chunk_bytes, 1u << 16);  // ... and bases: a chunk holds fewer bases than bytes, so a mapped
        for (size_t i = 0; i + 1 < cuts.size(); ++i) cap_bases = std::max(cap_bases, cuts[i + 1] - cuts[i]);  // chunk always fits
        if (cap_bases >= (1ull << 32) - 64) throw SketchyError("a record of the input exceeds what one batch can hold (4 Gbases)");
        const size_t n_chunks = mapped ? cuts.size() - 1 : (size_t)-1;

        constexpr size_t kInFlight = 9;  // batches the library holds between submit and rows (its staging slots: 2 x 4 + 1)
        const size_t n_slots = kInFlight + n_parse + n_format + 2;

        skx_stream* st = nullptr;
        hip_check(skx_stream_create(&st, ref.h, (uint32_t)config.top, (uint32_t)cap_reads, cap_bases), "stream");
        struct Guard { skx_stream* s; ~Guard() { skx_stream_destroy(s); } } guard{st};
        hip_check(skx_stream_set_packed_input(st, 1), "packed input");

        std::vector<Slot> slots(n_slots + 1);  // (+ the device thread's spill slot)
        struct SlotGuard { std::vector<Slot>& s; int dev; ~SlotGuard() { for (auto& x : s) { for (void* p : {(void*)x.packed, (void*)x.offsets, (void*)x.idx, (void*)x.sum}) if (p) skx_host_free(dev, p); } } } sguard{slots, device_};
        const size_t rows_cap = cap_reads * config.top;
        for (auto& sl : slots) {
            void* p = nullptr;
            hip_check(skx_host_alloc(device_, &p, cap_bases / 2 + 64), "batch buffer"); sl.packed = static_cast<uint8_t*>(p);
            hip_check(skx_host_alloc(device_, &p, (cap_reads + 1) * 8), "batch buffer"); sl.offsets = static_cast<uint64_t*>(p);
            hip_check(skx_host_alloc(device_, &p, std::max<size_t>(rows_cap, 1) * 4), "row buffer"); sl.idx = static_cast<uint32_t*>(p);
            hip_check(skx_host_alloc(device_, &p, std::max<size_t>(rows_cap, 1) * 8), "row buffer"); sl.sum = static_cast<uint64_t*>(p);
        }
        Slot& spill = slots[n_slots];

        std::mutex mu;  // slot states, queues, the error
        std::condition_variable cv_parsed, cv_free, cv_format;
        std::exception_ptr error;
        bool stop = false;  // an error, or the limit was reached: nobody starts anything new
        auto fail_with = [&](std::exception_ptr e) { std::lock_guard<std::mutex> l(mu); if (!error) error = e; stop = true; cv_parsed.notify_all(); cv_free.notify_all(); cv_format.notify_all(); };

        // ---- stage 1: parse + pack
        // record sink of one chunk: the slot while it has room, heap batches behind it
        auto parse_into = [&](Slot& sl, auto&& walk) {
            sl.n = 0; sl.extra.clear(); sl.offsets[0] = 0;
            uint64_t pos = 0;              // nibbles written to the current target
            bool in_extra = false;
            auto cur_packed = [&]() -> uint8_t* { return in_extra ? sl.extra.back().packed.data() : sl.packed; };
            size_t open_len = 0;           // bases of the record being assembled (multi-line FASTA)
            walk([&](const char* piece, size_t len, bool last_piece) {
                // room for the record's next piece?  (a record that alone exceeds a batch cannot be scored by this stream)
                const size_t n_now = in_extra ? sl.extra.back().offsets.size() - 1 : sl.n;
                if (open_len + len > cap_bases) throw SketchyError("a record exceeds the batch buffer (" + std::to_string(cap_bases) + " bases): raise --batch");
                if (pos + len > cap_bases || (open_len == 0 && n_now >= cap_reads)) {
                    // move on to a fresh heap batch; the open record's pieces so far move with it
                    Slot::Extra e; e.packed.assign(cap_bases / 2 + 64, 0);
                    const uint64_t rec0 = pos - open_len;
                    if (open_len) {  // re-pack the open record's nibbles at the front of the new batch
                        const uint8_t* src = cur_packed();
                        for (uint64_t i = 0; i < open_len; ++i) {
                            const uint8_t c = (uint8_t)((src[(rec0 + i) >> 1] >> (4 * ((rec0 + i) & 1))) & 0xF);
                            uint8_t& b = e.packed[i >> 1];
                            b = (i & 1) ? (uint8_t)((b & 0x0F) | (c << 4)) : c;
                        }
                    }
                    sl.extra.push_back(std::move(e));
                    in_extra = true; pos = open_len;
                }
                pos = skx_pack_bases(reinterpret_cast<const uint8_t*>(piece), len, cur_packed(), pos);
                open_len += len;
                if (last_piece) {
                    // (whitespace inside a piece is dropped by the packer: the offset is what was really written)
                    if (in_extra) sl.extra.back().offsets.push_back(pos);
                    else sl.offsets[++sl.n] = pos;
                    open_len = 0;
                }
            });
        };
        // a chunk of the mapped file [a, b): one pass per sequence line -- skx_pack_line finds its end and packs it.  The slot
        // cannot run out of BASES (it holds as many as the longest chunk has bytes); when it runs out of READS the rest of the
        // chunk goes to heap batches, each with the same room.  Errors are FastxReader's, at the same records.
        auto parse_mapped = [&](Slot& sl, const char* a, const char* b) {
            sl.n = 0; sl.extra.clear(); sl.offsets[0] = 0;
            uint8_t* target = sl.packed;
            uint64_t pos = 0;
            size_t n_here = 0;
            auto begin_record = [&] {
                if (n_here < cap_reads) return;
                Slot::Extra e; e.packed.assign(cap_bases / 2 + 64, 0);
                sl.extra.push_back(std::move(e));
                target = sl.extra.back().packed.data(); pos = 0; n_here = 0;
            };
            auto end_record = [&] {
                if (sl.extra.empty()) sl.offsets[++sl.n] = pos; else sl.extra.back().offsets.push_back(pos);
                ++n_here;
            };
            auto lf = [&](const char* q) { return static_cast<const char*>(memchr(q, '\n', (size_t)(b - q))); };
            const char* p = a;
            size_t stride = 0;  // length of the previous record
            static const bool guess_next = !getenv("SKETCHY_HIP_NO_PREFETCH");  // (measurement aid, like SKETCHY_HIP_NO_POPULATE)
            while (p < b) {
                if (*p == '\n' || *p == '\r') { ++p; continue; }
                if (fastq) {
                    if (*p != '@') throw SketchyError("input is neither FASTA nor FASTQ");
                    // The quality line is skipped, so every record starts with cache misses the hardware prefetcher cannot see
                    // coming: the byte that ends this record, the next header, the first lines of its sequence.  Records of one
                    // file mostly have one length (short-read instruments: exactly), so the previous record's length says where
                    // they are: requested now, they arrive while this record's sequence is packed.  A wrong guess costs five
                    // useless prefetches.
                    if (stride && guess_next && (size_t)(b - p) > stride + 320) {
                        const char* nx = p + stride;
                        for (int i = -64; i < 320; i += 64) __builtin_prefetch(nx + i, 0, 1);
                    }
                    const char* const rec0 = p;
                    const char* e0 = lf(p);
                    if (!e0 || e0 + 1 >= b) throw SketchyError("truncated FASTQ record");
                    const char* s0 = e0 + 1;
                    begin_record();
                    uint64_t used = 0;
                    const uint64_t pos1 = skx_pack_line(reinterpret_cast<const uint8_t*>(s0), (uint64_t)(b - s0), target, pos, &used);
                    if (used == 0 || s0[used - 1] != '\n') throw SketchyError("malformed FASTQ record (no '+' line)");
                    const char* plus = s0 + used;
                    if (plus >= b || *plus != '+') throw SketchyError("malformed FASTQ record (no '+' line)");
                    const char* e2 = (plus + 1 < b && plus[1] == '\n') ? plus + 1 : lf(plus);
                    if (!e2 || e2 + 1 >= b) throw SketchyError("truncated FASTQ record");
                    const char* q0 = e2 + 1;
                    const size_t line = (size_t)used - 1;  // (the quality line is as long as the sequence line: a byte test, not a scan)
                    const char* e3 = ((size_t)(b - q0) > line && q0[line] == '\n') ? q0 + line : lf(q0);
                    pos = pos1;
                    end_record();
                    p = e3 ? e3 + 1 : b;
                    stride = (size_t)(p - rec0);
                } else {
                    if (*p != '>') throw SketchyError("input is neither FASTA nor FASTQ");
                    const char* e0 = lf(p);
                    const char* q = e0 ? e0 + 1 : b;
                    begin_record();
                    while (q < b && *q != '>') {
                        uint64_t used = 0;
                        pos = skx_pack_line(reinterpret_cast<const uint8_t*>(q), (uint64_t)(b - q), target, pos, &used);
                        q += used;
                    }
                    end_record();
                    p = q;
                }
            }
        };
        std::atomic<size_t> next_chunk{0};
        std::vector<std::thread> parsers;
        std::vector<size_t> slot_gen(n_slots, 0);  // chunks a slot has carried so far: slot s is free for chunk c when slot_gen[s] == c / n_slots
        auto wait_slot = [&](size_t c) -> Slot* {
            const size_t s = c % n_slots;
            std::unique_lock<std::mutex> l(mu);
            cv_free.wait(l, [&] { return stop || (slots[s].state == 0 && slot_gen[s] == c / n_slots); });
            return stop ? nullptr : &slots[s];
        };
        auto publish = [&](Slot& sl, bool last) { { std::lock_guard<std::mutex> l(mu); sl.state = 1; sl.last = last; } cv_parsed.notify_all(); };
        if (mapped) {
            for (unsigned t = 0; t < n_parse; ++t)
                parsers.emplace_back([&] {
                    try {
                        for (;;) {
                            const size_t c = next_chunk.fetch_add(1);
                            if (c >= n_chunks) {
                                if (c == n_chunks) { Slot* sl = wait_slot(c); if (sl) { sl->n = 0; sl->extra.clear(); publish(*sl, true); } }  // the end marker
                                return;
                            }
                            Slot* sl = wait_slot(c);
                            if (!sl) return;
                            const char *a = fbegin + cuts[c], *b = fbegin + cuts[c + 1];
                            if (!getenv("SKETCHY_HIP_NO_POPULATE")) map.prefetch(cuts[c], cuts[c + 1]);  // (the chunk's pages in one call instead of a fault every 4 KB)
                            parse_mapped(*sl, a, b);
                            publish(*sl, false);
                        }
                    } catch (...) { fail_with(std::current_exception()); }
                });
        } else {
            parsers.emplace_back([&] {  // gzip / stdin: one sequential reader, batches of `batch_` reads
                try {
                    FastxReader reader(path);
                    std::string seq;
                    bool more = true;
                    for (size_t c = 0; more; ++c) {
                        Slot* sl = wait_slot(c);
                        if (!sl) return;
                        size_t bases = 0, n = 0;
                        parse_into(*sl, [&](auto&& emit) {
                            while (n < want_reads && bases < cap_bases / 2 && (more = reader.next(seq))) { emit(seq.data(), seq.size(), true); ++n; bases += seq.size(); }
                        });
                        if (!more && sl->n == 0 && sl->extra.empty()) { publish(*sl, true); return; }
                        publish(*sl, false);
                        if (!more) { Slot* e = wait_slot(c + 1); if (e) { e->n = 0; e->extra.clear(); publish(*e, true); } return; }
                    }
                } catch (...) { fail_with(std::current_exception()); }
            });
        }

        // ---- stage 3 + 4: rows as text, written in order
        struct Job { Slot* sl = nullptr; size_t seq = 0, n = 0, first_read = 1; uint64_t ticket = 0; std::vector<uint32_t> idx; std::vector<uint64_t> sum; };
        std::deque<Job> format_q;
        bool format_closed = false;
        std::map<size_t, std::string> done_text;  // seq -> text, waiting for its turn
        size_t next_write = 0;
        std::mutex write_mu;
        std::vector<std::string> mid, tail;  // per reference sketch: "\t<name>\t" and "\t<genotype columns>\n"
        if (!config.consensus)
            for (const auto& sk : sketches) { mid.push_back("\t" + sk.name + "\t"); tail.push_back("\t" + join_tab(geno.map.at(sk.name)) + "\n"); }
        auto release_slot = [&](Slot* sl) { { std::lock_guard<std::mutex> l(mu); sl->state = 0; slot_gen[(size_t)(sl - slots.data())] += 1; } cv_free.notify_all(); };
        std::vector<std::thread> formatters;
        for (unsigned t = 0; t < n_format; ++t)
            formatters.emplace_back([&] {
                try {
                    for (;;) {
                        Job job;
                        {
                            std::unique_lock<std::mutex> l(mu);
                            cv_format.wait(l, [&] { return !format_q.empty() || format_closed || (bool)error; });
                            if (format_q.empty()) return;
                            job = std::move(format_q.front()); format_q.pop_front();
                        }
                        const uint32_t* idx = job.sl ? job.sl->idx : job.idx.data();
                        const uint64_t* sum = job.sl ? job.sl->sum : job.sum.data();
                        std::string text;
                        if (config.consensus) {
                            std::ostringstream os;
                            for (size_t r = 0; r < job.n; ++r) print_results(sketches, geno, idx + r * config.top, sum + r * config.top, job.first_read + r, config, os);
                            text = os.str();
                        } else {
                            size_t need = 0;
                            for (size_t i = 0; i < job.n * config.top; ++i) need += 42 + mid[idx[i]].size() + tail[idx[i]].size();
                            text.resize(need);
                            char* p = text.data();
                            for (size_t r = 0; r < job.n; ++r)
                                for (size_t t2 = 0; t2 < config.top; ++t2) {
                                    const uint32_t g = idx[r * config.top + t2];
                                    p = put_u64(p, job.first_read + r);
                                    memcpy(p, mid[g].data(), mid[g].size()); p += mid[g].size();
                                    p = put_u64(p, sum[r * config.top + t2]);
                                    memcpy(p, tail[g].data(), tail[g].size()); p += tail[g].size();
                                }
                            text.resize((size_t)(p - text.data()));
                        }
                        if (job.sl) release_slot(job.sl);  // the rows are text now: the slot goes back to the parsers
                        std::lock_guard<std::mutex> w(write_mu);
                        done_text.emplace(job.seq, std::move(text));
                        for (auto it = done_text.find(next_write); it != done_text.end(); it = done_text.find(next_write)) {
                            out.write(it->second.data(), (std::streamsize)it->second.size());
                            done_text.erase(it); ++next_write;
                        }
                    }
                } catch (...) { fail_with(std::current_exception()); }
            });

        // ---- stage 2: the device, in file order (this thread)
        size_t fed = 0, n_batches = 0, seq_no = 0;
        const auto t_first = clock::now();  // (the parsers have just been started: everything before was set-up)
        double s_wait_parse = 0, s_submit = 0, s_retire = 0;  // where this thread's time goes (--timing)
        auto since = [](clock::time_point t) { return std::chrono::duration<double>(clock::now() - t).count(); };
        try {
            std::deque<std::pair<Slot*, Job>> in_flight;  // submitted, rows not yet known to be on the host
            auto hand_to_format = [&](Job&& j) { { std::lock_guard<std::mutex> l(mu); format_q.push_back(std::move(j)); } cv_format.notify_one(); };
            // the library holds up to kInFlight batches between submit and rows (a submit first finishes the batch that last used
            // its staging slot): everything older is through, and skx_stream_wait on it returns at once -- waiting on a YOUNGER
            // ticket would cut its group short
            auto retire = [&](size_t keep) {
                while (in_flight.size() > keep) {
                    hip_check(skx_stream_wait(st, in_flight.front().second.ticket), "wait");
                    hand_to_format(std::move(in_flight.front().second)); in_flight.pop_front();
                }
            };
            bool limit_hit = false;
            for (size_t c = 0; !limit_hit; ++c) {
                Slot* sl = &slots[c % n_slots];
                {
                    const auto tw = clock::now();
                    std::unique_lock<std::mutex> l(mu);
                    cv_parsed.wait(l, [&] { return stop || (sl->state == 1 && slot_gen[c % n_slots] == c / n_slots); });
                    s_wait_parse += since(tw);
                    if (stop) break;
                }
                const bool last = sl->last;
                auto submit = [&](const uint8_t* packed, const uint64_t* offsets, size_t n, uint32_t* idx, uint64_t* sum, Slot* owner) {
                    if (config.limit && fed + n >= config.limit) { n = config.limit - fed; limit_hit = true; }  // src/sketchy.rs:350-353
                    if (n == 0) { if (owner) release_slot(owner); return; }
                    uint64_t ticket = 0;
                    const auto ts = clock::now();
                    hip_check(skx_stream_submit(st, packed, offsets, (uint32_t)n, idx, sum, &ticket), "submit");
                    s_submit += since(ts);
                    Job j; j.sl = owner; j.seq = seq_no++; j.n = n; j.first_read = fed + 1; j.ticket = ticket;
                    fed += n; ++n_batches;
                    in_flight.emplace_back(owner, std::move(j));
                    const auto tr = clock::now();
                    retire(kInFlight);
                    s_retire += since(tr);
                };
                std::vector<Slot::Extra> extra;
                extra.swap(sl->extra);
                submit(sl->packed, sl->offsets, sl->n, sl->idx, sl->sum, sl);
                for (auto& e : extra) {  // the chunk's overflow, through the spill slot: drained each time (rare path)
                    if (limit_hit) break;
                    const size_t n = e.offsets.size() - 1;
                    memcpy(spill.packed, e.packed.data(), std::min(e.packed.size(), cap_bases / 2 + 64));
                    memcpy(spill.offsets, e.offsets.data(), e.offsets.size() * 8);
                    submit(spill.packed, spill.offsets, n, spill.idx, spill.sum, nullptr);
                    hip_check(skx_stream_drain(st), "drain");
                    Job& j = in_flight.back().second;  // its rows leave the spill slot before the next overflow batch uses it
                    j.idx.assign(spill.idx, spill.idx + j.n * config.top); j.sum.assign(spill.sum, spill.sum + j.n * config.top);
                    retire(0);
                }
                if (last) break;
            }
            const auto td = clock::now();
            hip_check(skx_stream_drain(st), "drain");
            retire(0);
            s_retire += since(td);
        } catch (...) { fail_with(std::current_exception()); }
        { std::lock_guard<std::mutex> l(mu); format_closed = true; if (!error) stop = true; }
        cv_format.notify_all(); cv_free.notify_all(); cv_parsed.notify_all();
        for (auto& t : formatters) t.join();
        for (auto& t : parsers) t.join();
        out.flush();
        if (error) std::rethrow_exception(error);
        if (config.timing) {
            const double all_s = std::chrono::duration<double>(clock::now() - t_begin).count();
            const double run_s = std::chrono::duration<double>(clock::now() - t_first).count();
            // seconds_parse_start_to_last_row: from the moment the parser threads start (stream and page-locked slots exist) until
            // the last row is written; seconds_stream adds that set-up.  device thread: waiting for parsed chunks / inside
            // skx_stream_submit (it blocks when the library's staging slots are all in flight) / final drain + hand-over
            std::fprintf(stderr, "{\"sketchy_hip_timing\": {\"reads\": %zu, \"batches\": %zu, \"seconds_stream\": %.6f, \"seconds_open_input\": %.6f, \"seconds_parse_start_to_last_row\": %.6f, "
                                 "\"reads_per_s\": %.1f, \"parse_threads\": %u, \"format_threads\": %u, \"cpus_pinned_near_device\": %u, \"input\": \"%s\", \"batch_reads\": %zu, "
                                 "\"device_thread_s\": {\"wait_for_parsers\": %.6f, \"submit\": %.6f, \"drain_and_hand_over\": %.6f}}}\n",
                         fed, n_batches, all_s, open_s, run_s, fed / std::max(run_s, 1e-9), n_parse, n_format, pinned, mapped ? (map.inflated() ? (fastq ? "bgzf fastq" : "bgzf fasta") : (fastq ? "mapped fastq" : "mapped fasta")) : "streamed", want_reads,
                         s_wait_parse, s_submit, s_retire);
        }
    }

    // offline mode: one sketcher over all reads == bottom-s of the union of the per-read bottom-s sketches
    void shared_hashes(FastxReader& reader, const std::vector<Sketch>& sketches, Ref& ref, const Genotypes& geno,
                       const PredictConfig& config, std::ostream& out) {
        Batch b; std::string seq; size_t read = 0;
        std::vector<uint64_t> pooled, sk, merged; std::vector<uint32_t> len;
        auto flush = [&]() {
            if (b.n() == 0) return;
            sk.assign(b.n() * (size_t)ref.s, 0); len.assign(b.n(), 0);
            hip_check(skx_sketch_reads(device_, ref.k, ref.seed, ref.s, b.bases.data(), b.offsets.data(), (uint32_t)b.n(), sk.data(), len.data()), "sketch");
            for (size_t r = 0; r < b.n(); ++r) {
                merged.clear();
                std::set_union(pooled.begin(), pooled.end(), sk.begin() + r * ref.s, sk.begin() + r * ref.s + len[r], std::back_inserter(merged));
                if (merged.size() > ref.s) merged.resize(ref.s);
                pooled.swap(merged);
            }
            b.clear();
        };
        while (reader.next(seq)) {
            b.add(seq); ++read;
            if (b.n() == batch_ || b.bases.size() > (1ull << 29)) flush();
            if (read == config.limit) break;  // :296-299
        }
        flush();
        std::vector<uint32_t> common(sketches.size()), plen{(uint32_t)pooled.size()};
        if (pooled.empty()) pooled.push_back(0);
        hip_check(skx_common_hashes(ref.h, pooled.data(), plen.data(), 1, (uint32_t)std::max<size_t>(pooled.size(), 1), common.data()), "common");
        std::vector<uint32_t> order(sketches.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b2) { return common[a] > common[b2]; });  // :310
        std::vector<uint32_t> idx(order.begin(), order.begin() + config.top); std::vector<uint64_t> sum(config.top);
        for (size_t j = 0; j < config.top; ++j) sum[j] = common[idx[j]];
        print_results(sketches, geno, idx.data(), sum.data(), read, config, out);
    }

    // src/sketchy.rs:358-402
    void print_results(const std::vector<Sketch>& sketches, const Genotypes& geno, const uint32_t* idx, const uint64_t* sum,
                       size_t read, const PredictConfig& config, std::ostream& out) {
        if (config.consensus) {
            const size_t nfeat = geno.map.at(sketches[idx[0]].name).size();
            std::vector<std::string> call;
            for (size_t j = 0; j < nfeat; ++j) {
                std::map<std::string, size_t> counts;  // ties: the reference's HashMap order is unspecified; here: smallest value
                for (size_t t = 0; t < config.top; ++t) counts[geno.map.at(sketches[idx[t]].name)[j]]++;
                auto best = std::max_element(counts.begin(), counts.end(), [](const auto& a, const auto& b) { return a.second < b.second; });
                call.push_back(best->first);
            }
            out << read << "\t-\t-\t" << join_tab(call) << "\n";
        } else {
            for (size_t t = 0; t < config.top; ++t) {
                const auto& name = sketches[idx[t]].name;
                out << read << "\t" << name << "\t" << sum[t] << "\t" << join_tab(geno.map.at(name)) << "\n";
            }
        }
    }

    int device_; size_t batch_;
};

}  // namespace sketchy

// ------------------------------------------------------------------ command line (src/cli.rs flag names)
static void usage() {
    std::fprintf(stderr,
                 "sketchy-hip sketch  -o OUT.msh [-i GENOME.fa[.gz] ...] [-s SIZE=1000] [-k K=16] [-e SEED=0]   (paths on stdin without -i)\n"
                 "sketchy-hip predict -r REF.msh -g GENO.tsv [-i READS.fx[.gz]] [-t TOP] [-l LIMIT] [-s] [-c] [-H] [-b BATCH_READS] [-j THREADS] [--timing] [--pin]\n"
                 "sketchy-hip shared  -r REF.msh -q QUERY.msh\n"
                 "sketchy-hip info    -i SKETCH.msh [-p]\n"
                 "sketchy-hip check   -r REF.msh -g GENO.tsv\n");
}

int main(int argc, char** argv) {
    if (argc < 2) { usage(); return 2; }
    const std::string cmd = argv[1];
    std::map<std::string, std::string> opt; std::map<std::string, bool> flag;
    std::vector<std::string> inputs;  // `sketch -i` takes several paths (src/cli.rs:27-28: multiple = true)
    const std::map<std::string, std::string> longnames = {{"--input", "-i"}, {"--reference", "-r"}, {"--genotypes", "-g"}, {"--top", "-t"}, {"--limit", "-l"},
                                                          {"--stream", "-s"}, {"--consensus", "-c"}, {"--header", "-H"}, {"--query", "-q"}, {"--params", "-p"},
                                                          {"--device", "-d"}, {"--batch", "-b"}, {"--output", "-o"}, {"--sketch-size", "-s"},
                                                          {"--kmer-size", "-k"}, {"--seed", "-e"}, {"--threads", "-j"}, {"--timing", "-T"}, {"--pin", "-P"}};
    const bool is_sketch = cmd == "sketch";  // there -s takes a value (sketch size), elsewhere it is --stream
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        if (is_sketch && a == "--stream") { usage(); return 2; }
        if (longnames.count(a)) a = longnames.at(a);
        if (is_sketch && a == "-i") { while (i + 1 < argc && argv[i + 1][0] != '-') inputs.push_back(argv[++i]); continue; }
        if ((!is_sketch && a == "-s") || a == "-c" || a == "-H" || a == "-p" || a == "-T" || a == "-P") flag[a] = true;
        else if (i + 1 < argc) opt[a] = argv[++i];
        else { usage(); return 2; }
    }
    try {
        // (-b: reads per device batch.  Streaming: 65 536 by default -- up to four batches share one scan of the reference, and the
        // copy + kernels of a host-fed batch cost ~0.7 ms whatever its size below that (measured at C2 on an MI355X box, 1.57 M reads of
        // 1.5 kb from /dev/shm: 21 M reads/s with 32 768-read batches, 26-27 M with 65 536); it costs ~100 MB of page-locked memory per
        // slot, 27 slots with 16 threads)
        const bool streaming = cmd == "predict" && flag["-s"];
        sketchy::Sketchy app(opt.count("-d") ? std::atoi(opt["-d"].c_str()) : 0,
                             opt.count("-b") ? (size_t)std::max(1L, std::atol(opt["-b"].c_str())) : (streaming ? 65536 : 16384));
        if (cmd == "sketch") {
            if (!opt.count("-o")) { usage(); return 2; }
            if (inputs.empty()) { std::string line; while (std::getline(std::cin, line)) if (!line.empty()) inputs.push_back(line); }  // src/sketchy.rs:137-146
            app.sketch(inputs, opt["-o"], opt.count("-s") ? (size_t)std::atol(opt["-s"].c_str()) : 1000,
                       opt.count("-k") ? (uint32_t)std::atoi(opt["-k"].c_str()) : 16u, opt.count("-e") ? std::strtoull(opt["-e"].c_str(), nullptr, 10) : 0ull);
        } else if (cmd == "predict") {
            if (!opt.count("-r") || !opt.count("-g")) { usage(); return 2; }
            sketchy::PredictConfig cfg;
            cfg.top = opt.count("-t") ? (size_t)std::atol(opt["-t"].c_str()) : 1;
            cfg.limit = opt.count("-l") ? (size_t)std::atol(opt["-l"].c_str()) : 0;
            cfg.stream = flag["-s"]; cfg.consensus = flag["-c"]; cfg.header = flag["-H"];
            cfg.threads = opt.count("-j") ? (size_t)std::max(1L, std::atol(opt["-j"].c_str())) : 0; cfg.timing = flag["-T"]; cfg.pin = flag["-P"];
            app.predict(opt.count("-i") ? std::optional<std::string>(opt["-i"]) : std::nullopt, opt["-r"], opt["-g"], cfg, std::cout);
        } else if (cmd == "shared") {
            if (!opt.count("-r") || !opt.count("-q")) { usage(); return 2; }
            app.shared(opt["-r"], opt["-q"], std::cout);
        } else if (cmd == "check") {
            if (!opt.count("-r") || !opt.count("-g")) { usage(); return 2; }
            app.check(opt["-r"], opt["-g"], std::cout);
        } else if (cmd == "info") {
            if (!opt.count("-i")) { usage(); return 2; }
            app.info(opt["-i"], flag["-p"], std::cout);
        } else { usage(); return 2; }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
    return 0;
}
