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
// Streaming runs as a three-stage pipeline (the reference reads, scores and prints one record at a time on one
// thread): a reader thread parses FASTX into page-locked batch buffers, this thread pushes batches through the C ABI,
// a writer thread formats the rows -- so parsing, the H2D copy + device work, and printing overlap.
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <exception>
#include <iostream>
#include <map>
#include <mutex>
#include <optional>
#include <thread>

#include "formats.hpp"
#include "sketchy_hip.h"

namespace sketchy {

struct PredictConfig { size_t top = 1, limit = 0; bool stream = false, consensus = false, header = false; };

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
        FastxReader reader(fastx ? *fastx : std::string("-"));
        const auto geno = read_genotypes(genotypes);
        for (const auto& s : sketches)  // the reference panics on a missing name (geno_map[&name], :308/:345)
            if (!geno.map.count(s.name)) throw SketchyError("reference sketch " + s.name + " has no row in the genotype table");
        if (config.top > sketches.size()) throw SketchyError("--top exceeds the number of reference sketches");
        if (config.header) out << "reads\tsketch_id\tshared_hashes\t" << geno.header << "\n";  // :99-101
        Ref ref(sketches, device_);
        if (config.stream) sum_of_shared_hashes(reader, sketches, ref, geno, config, out);
        else shared_hashes(reader, sketches, ref, geno, config, out);
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

    // ---- streaming pipeline plumbing
    template <class T>
    class Channel {  // unbounded FIFO between two threads; close() wakes the consumer for good
      public:
        void put(T v) { { std::lock_guard<std::mutex> l(m); q.push_back(std::move(v)); } cv.notify_one(); }
        bool get(T& v) {
            std::unique_lock<std::mutex> l(m);
            cv.wait(l, [&] { return !q.empty() || closed; });
            if (q.empty()) return false;
            v = std::move(q.front()); q.pop_front();
            return true;
        }
        void close() { { std::lock_guard<std::mutex> l(m); closed = true; } cv.notify_all(); }
      private:
        std::mutex m; std::condition_variable cv; std::deque<T> q; bool closed = false;
    };
    struct Slot {  // one batch travelling reader -> device -> writer and back
        uint8_t* bases = nullptr; size_t cap = 0, len = 0;   // page-locked (skx_host_alloc)
        std::vector<uint64_t> offsets{0};
        size_t first_read = 1;
        std::vector<uint32_t> idx; std::vector<uint64_t> sum;
        size_t n() const { return offsets.size() - 1; }
        void clear() { len = 0; offsets.assign(1, 0); }
    };

    // streaming mode
    void sum_of_shared_hashes(FastxReader& reader, const std::vector<Sketch>& sketches, Ref& ref, const Genotypes& geno,
                              const PredictConfig& config, std::ostream& out) {
        const size_t cap = 256ull << 20;  // bases per batch buffer (a single record must fit)
        skx_stream* st = nullptr;
        hip_check(skx_stream_create(&st, ref.h, (uint32_t)config.top, (uint32_t)batch_, cap), "stream");
        struct Guard { skx_stream* s; ~Guard() { skx_stream_destroy(s); } } guard{st};
        constexpr int kSlots = 3;
        Slot slots[kSlots];
        struct SlotGuard { Slot* s; int n, dev; ~SlotGuard() { for (int i = 0; i < n; ++i) if (s[i].bases) skx_host_free(dev, s[i].bases); } } sguard{slots, kSlots, device_};
        Channel<Slot*> free_slots, to_device, to_writer;
        for (auto& sl : slots) {
            void* p = nullptr;
            hip_check(skx_host_alloc(device_, &p, cap), "batch buffer");
            sl.bases = static_cast<uint8_t*>(p); sl.cap = cap;
            free_slots.put(&sl);
        }
        std::exception_ptr reader_err, writer_err;

        // stage 1: parse (src/sketchy.rs:328-333: one record at a time; --limit at :350-353)
        std::thread reader_thread([&] {
            try {
                std::string seq; size_t fed = 0; Slot* sl = nullptr;
                auto hand_over = [&] { if (sl && sl->n()) { to_device.put(sl); sl = nullptr; } };
                bool stop = false;
                while (!stop && reader.next(seq)) {
                    if (seq.size() > cap) throw SketchyError("a record exceeds the batch buffer");
                    if (sl && sl->len + seq.size() > sl->cap) hand_over();
                    if (!sl) { if (!free_slots.get(sl)) break; sl->clear(); sl->first_read = fed + 1; }
                    memcpy(sl->bases + sl->len, seq.data(), seq.size());
                    sl->len += seq.size(); sl->offsets.push_back(sl->len); ++fed;
                    stop = config.limit > 0 && fed == config.limit;
                    if (sl->n() == batch_ || stop) hand_over();
                }
                hand_over();
            } catch (...) { reader_err = std::current_exception(); }
            to_device.close();
        });
        // stage 3: rows (src/sketchy.rs:389-400)
        std::thread writer_thread([&] {
            try {
                std::vector<std::string> tail;  // per reference sketch: "\t<name>\t" and "\t<genotype columns>\n"
                std::vector<std::string> geno_tail;
                if (!config.consensus)
                    for (const auto& sk : sketches) { tail.push_back("\t" + sk.name + "\t"); geno_tail.push_back("\t" + join_tab(geno.map.at(sk.name)) + "\n"); }
                std::string text;
                Slot* sl = nullptr;
                while (to_writer.get(sl)) {
                    if (config.consensus) {
                        for (size_t r = 0; r < sl->n(); ++r)
                            print_results(sketches, geno, &sl->idx[r * config.top], &sl->sum[r * config.top], sl->first_read + r, config, out);
                    } else {
                        text.clear();
                        for (size_t r = 0; r < sl->n(); ++r) {
                            const std::string rd = std::to_string(sl->first_read + r);
                            for (size_t t = 0; t < config.top; ++t) {
                                const uint32_t g = sl->idx[r * config.top + t];
                                text += rd; text += tail[g]; text += std::to_string(sl->sum[r * config.top + t]); text += geno_tail[g];
                            }
                        }
                        out.write(text.data(), (std::streamsize)text.size());
                    }
                    free_slots.put(sl);
                }
            } catch (...) { writer_err = std::current_exception(); free_slots.close(); }
        });
        // stage 2: the device path
        std::exception_ptr device_err;
        try {
            Slot* sl = nullptr;
            while (to_device.get(sl)) {
                sl->idx.assign(sl->n() * config.top, 0); sl->sum.assign(sl->n() * config.top, 0);
                hip_check(skx_stream_push(st, sl->bases, sl->offsets.data(), (uint32_t)sl->n(), sl->idx.data(), sl->sum.data(), nullptr, nullptr, nullptr), "push");
                to_writer.put(sl);
            }
        } catch (...) { device_err = std::current_exception(); free_slots.close(); }
        to_writer.close();
        if (device_err) {  // unblock the reader (it may wait for a free slot) and drain what it still hands over
            Slot* sl = nullptr;
            while (to_device.get(sl)) {}
        }
        reader_thread.join();
        writer_thread.join();
        if (device_err) std::rethrow_exception(device_err);
        if (reader_err) std::rethrow_exception(reader_err);
        if (writer_err) std::rethrow_exception(writer_err);
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
                 "sketchy-hip predict -r REF.msh -g GENO.tsv [-i READS.fx[.gz]] [-t TOP] [-l LIMIT] [-s] [-c] [-H]\n"
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
                                                          {"--kmer-size", "-k"}, {"--seed", "-e"}};
    const bool is_sketch = cmd == "sketch";  // there -s takes a value (sketch size), elsewhere it is --stream
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        if (is_sketch && a == "--stream") { usage(); return 2; }
        if (longnames.count(a)) a = longnames.at(a);
        if (is_sketch && a == "-i") { while (i + 1 < argc && argv[i + 1][0] != '-') inputs.push_back(argv[++i]); continue; }
        if ((!is_sketch && a == "-s") || a == "-c" || a == "-H" || a == "-p") flag[a] = true;
        else if (i + 1 < argc) opt[a] = argv[++i];
        else { usage(); return 2; }
    }
    try {
        sketchy::Sketchy app(opt.count("-d") ? std::atoi(opt["-d"].c_str()) : 0, opt.count("-b") ? (size_t)std::atol(opt["-b"].c_str()) : 16384);
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
