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
// Command line: the reference's flag names and defaults (src/cli.rs:51-132).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <optional>

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
    explicit Sketchy(int device = 0, size_t batch_reads = 4096) : device_(device), batch_(batch_reads) {}

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

    void info(const std::string& input, bool params, std::ostream& out) {
        const auto sk = read_sketch(input);
        if (sk.empty()) throw SketchyError("empty sketch file");
        if (params) { out << "type=mash sketch_size=" << sk[0].hashes.size() << " kmer_size=" << sk[0].kmer_length << " seed=" << sk[0].hash_seed << "\n"; return; }
        for (const auto& s : sk) out << s.name << " " << s.seq_length << " " << s.hashes.size() << "\n";
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
            hip_check(skx_ref_create(&h, device, k, seed, stride, (uint32_t)sk.size(), flat.data(), len.data()), "reference upload");
        }
        ~Ref() { skx_ref_destroy(h); }
        uint32_t stride_ = 0;
    };

    struct Batch { std::vector<uint8_t> bases; std::vector<uint64_t> offsets{0}; size_t n() const { return offsets.size() - 1; }
                   void add(const std::string& seq) { bases.insert(bases.end(), seq.begin(), seq.end()); offsets.push_back(bases.size()); }
                   void clear() { bases.clear(); offsets.assign(1, 0); } };

    // streaming mode
    void sum_of_shared_hashes(FastxReader& reader, const std::vector<Sketch>& sketches, Ref& ref, const Genotypes& geno,
                              const PredictConfig& config, std::ostream& out) {
        if (ref.stride_ != ref.s) throw SketchyError("reference sketches of unequal size are not supported in stream mode");
        skx_stream* st = nullptr;
        hip_check(skx_stream_create(&st, ref.h, (uint32_t)config.top, (uint32_t)batch_, 1ull << 30), "stream");
        struct Guard { skx_stream* s; ~Guard() { skx_stream_destroy(s); } } guard{st};
        Batch b; std::string seq;
        size_t read = 1;  // :327
        std::vector<uint32_t> idx; std::vector<uint64_t> sum;
        auto flush = [&]() {
            if (b.n() == 0) return;
            idx.assign(b.n() * config.top, 0); sum.assign(b.n() * config.top, 0);
            hip_check(skx_stream_push(st, b.bases.data(), b.offsets.data(), (uint32_t)b.n(), idx.data(), sum.data(), nullptr, nullptr, nullptr), "push");
            for (size_t r = 0; r < b.n(); ++r, ++read) print_results(sketches, geno, &idx[r * config.top], &sum[r * config.top], read, config, out);
            b.clear();
        };
        size_t fed = 0;
        while (reader.next(seq)) {
            b.add(seq); ++fed;
            const bool last = config.limit > 0 && fed == config.limit;  // :350-353
            if (b.n() == batch_ || b.bases.size() > (1ull << 29) || last) flush();
            if (last) break;
        }
        flush();
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
                 "sketchy-hip predict -r REF.msh -g GENO.tsv [-i READS.fx[.gz]] [-t TOP] [-l LIMIT] [-s] [-c] [-H]\n"
                 "sketchy-hip shared  -r REF.msh -q QUERY.msh\n"
                 "sketchy-hip info    -i SKETCH.msh [-p]\n");
}

int main(int argc, char** argv) {
    if (argc < 2) { usage(); return 2; }
    const std::string cmd = argv[1];
    std::map<std::string, std::string> opt; std::map<std::string, bool> flag;
    const std::map<std::string, std::string> longnames = {{"--input", "-i"}, {"--reference", "-r"}, {"--genotypes", "-g"}, {"--top", "-t"}, {"--limit", "-l"},
                                                          {"--stream", "-s"}, {"--consensus", "-c"}, {"--header", "-H"}, {"--query", "-q"}, {"--params", "-p"},
                                                          {"--device", "-d"}, {"--batch", "-b"}};
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        if (longnames.count(a)) a = longnames.at(a);
        if (a == "-s" || a == "-c" || a == "-H" || a == "-p") flag[a] = true;
        else if (i + 1 < argc) opt[a] = argv[++i];
        else { usage(); return 2; }
    }
    try {
        sketchy::Sketchy app(opt.count("-d") ? std::atoi(opt["-d"].c_str()) : 0, opt.count("-b") ? (size_t)std::atol(opt["-b"].c_str()) : 4096);
        if (cmd == "predict") {
            if (!opt.count("-r") || !opt.count("-g")) { usage(); return 2; }
            sketchy::PredictConfig cfg;
            cfg.top = opt.count("-t") ? (size_t)std::atol(opt["-t"].c_str()) : 1;
            cfg.limit = opt.count("-l") ? (size_t)std::atol(opt["-l"].c_str()) : 0;
            cfg.stream = flag["-s"]; cfg.consensus = flag["-c"]; cfg.header = flag["-H"];
            app.predict(opt.count("-i") ? std::optional<std::string>(opt["-i"]) : std::nullopt, opt["-r"], opt["-g"], cfg, std::cout);
        } else if (cmd == "shared") {
            if (!opt.count("-r") || !opt.count("-q")) { usage(); return 2; }
            app.shared(opt["-r"], opt["-q"], std::cout);
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
