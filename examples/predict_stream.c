/*
 * examples/predict_stream.c -- the reference's streaming loop (src/sketchy.rs:317-356) written against the C ABI, in plain C:
 * what a host in any language does through its FFI (INTEGRATION.md shows the same calls from Rust).
 *
 *   reference side  : Vec<Sketch> from a .msh file (src/sketchy.rs:497-536)  ->  skx_ref_create
 *   per read        : create_sketcher + process + to_vec (:331-335), _common_hashes against every genome (:337-341),
 *                     sum_of_shared_hashes[i] += shared (:341), stable sort + first `top` rows (:348, :391)
 *                                                                              ->  ONE skx_stream_push for a batch of reads
 *
 * Self-contained: the "genomes" are pseudo-random sequences (a small LCG, so a test can rebuild them), their sketches come
 * from skx_sketch_reads (finch's sketcher as the reference calls it), the reads are error-free pieces of the genomes taken
 * round robin.  Prints one row per read: read number, best genome, its running sum of shared hashes -- the first three
 * columns of `sketchy predict --stream` (:391-398).
 *
 * build:  gcc -O2 -I include examples/predict_stream.c -L sketchy_amd -lsketchy_hip -Wl,-rpath,$PWD/sketchy_amd -o predict_stream
 * exit code 2 = no usable gfx950 device (the library has no CPU path).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sketchy_hip.h"

#define N_GENOMES 6u
#define GENOME_LEN 30000u
#define SKETCH_SIZE 200u
#define KMER 16u
#define N_READS 240u
#define READ_LEN 600u

static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static uint32_t lcg(void) {  /* (the test in tests/ rebuilds the same data) */
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(lcg_state >> 33);
}

#define CHECK(call)                                                                          \
    do {                                                                                     \
        const int rc_ = (call);                                                              \
        if (rc_ != SKX_OK) {                                                                 \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, skx_last_error());            \
            return rc_ == SKX_ERR_NO_DEVICE ? 2 : 1;                                         \
        }                                                                                    \
    } while (0)

int main(void) {
    if (skx_device_count() < 1) {
        fprintf(stderr, "no HIP device: libsketchy_hip has no CPU path (%s)\n", skx_version());
        return 2;
    }
    /* ---- the reference collection: six genomes that share a common ancestor (every tenth base re-drawn per genome) */
    static uint8_t genomes[N_GENOMES * GENOME_LEN];
    static uint64_t g_off[N_GENOMES + 1];
    for (uint32_t i = 0; i < GENOME_LEN; ++i) genomes[i] = (uint8_t)"ACGT"[lcg() & 3u];
    for (uint32_t g = 1; g < N_GENOMES; ++g)
        for (uint32_t i = 0; i < GENOME_LEN; ++i)
            genomes[g * GENOME_LEN + i] = (lcg() % 10u == 0u) ? (uint8_t)"ACGT"[lcg() & 3u] : genomes[i];
    for (uint32_t g = 0; g <= N_GENOMES; ++g) g_off[g] = (uint64_t)g * GENOME_LEN;
    static uint64_t ref_hashes[N_GENOMES * SKETCH_SIZE];
    static uint32_t ref_len[N_GENOMES];
    CHECK(skx_sketch_reads(0, KMER, 0, SKETCH_SIZE, genomes, g_off, N_GENOMES, ref_hashes, ref_len));  /* `sketchy sketch` */
    skx_ref *ref = NULL;
    CHECK(skx_ref_create(&ref, 0, KMER, 0, ref_len[0], SKETCH_SIZE, N_GENOMES, ref_hashes, ref_len));   /* s := |sketch 0| (:82) */

    /* ---- the read stream: pieces of the genomes, round robin */
    static uint8_t reads[N_READS * READ_LEN];
    static uint64_t r_off[N_READS + 1];
    for (uint32_t r = 0; r < N_READS; ++r) {
        const uint32_t g = r % N_GENOMES, at = lcg() % (GENOME_LEN - READ_LEN);
        memcpy(reads + (size_t)r * READ_LEN, genomes + (size_t)g * GENOME_LEN + at, READ_LEN);
        r_off[r] = (uint64_t)r * READ_LEN;
    }
    r_off[N_READS] = (uint64_t)N_READS * READ_LEN;

    /* ---- the loop of src/sketchy.rs:328-354, two batches to show that the table carries over */
    skx_stream *st = NULL;
    CHECK(skx_stream_create(&st, ref, 1, N_READS, (uint64_t)N_READS * READ_LEN));
    static uint32_t best[N_READS];
    static uint64_t sum[N_READS];
    const uint32_t half = N_READS / 2u;
    CHECK(skx_stream_push(st, reads, r_off, half, best, sum, NULL, NULL, NULL));
    CHECK(skx_stream_push(st, reads, r_off + half, N_READS - half, best + half, sum + half, NULL, NULL, NULL));
    for (uint32_t r = 0; r < N_READS; ++r) printf("%u\t%u\t%llu\n", r + 1u, best[r], (unsigned long long)sum[r]);

    uint64_t table[N_GENOMES];
    CHECK(skx_stream_table(st, table));  /* sum_of_shared_hashes (:326) */
    fprintf(stderr, "sum_of_shared_hashes:");
    for (uint32_t g = 0; g < N_GENOMES; ++g) fprintf(stderr, " %llu", (unsigned long long)table[g]);
    fprintf(stderr, "\n");
    skx_stream_destroy(st);
    skx_ref_destroy(ref);
    return 0;
}
