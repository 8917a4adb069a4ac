// tools/bgzf_rate.cpp <file.fastq.gz (BGZF)> -- MappedFile::open_bgzf alone: seconds and GB/s of inflated text with 1 / 4 / 8 / 16 threads,
// this repo's decoder against zlib's inflate (g++ -O2 -std=c++17 -pthread -I sketchy_amd/host tools/bgzf_rate.cpp -lz)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "formats.hpp"
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    for (int zl = 0; zl < 2; ++zl)
        for (unsigned t : {1u, 4u, 8u, 16u, 22u}) {
            sketchy::bgzf_force_zlib() = zl != 0;
            double best = 1e9; size_t bytes = 0;
            for (int rep = 0; rep < 2; ++rep) {
                sketchy::MappedFile m;
                const auto t0 = std::chrono::steady_clock::now();
                if (!m.open_bgzf(argv[1], t)) { fprintf(stderr, "not BGZF\n"); return 1; }
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                bytes = m.size();
            }
            printf("%-14s %2u threads: %.3f s  %.2f GB/s of inflated text (%.2f GB)\n", zl ? "zlib inflate" : "fast_inflate", t, best, bytes / best / 1e9, bytes / 1e9);
        }
    return 0;
}
