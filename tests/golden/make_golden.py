"""Generates tests/golden/*.npz|json -- small committed input/expected-output vectors.

The reference (Rust; no cargo/rustc/crates here) cannot be run to produce vectors and holds
none of its own, so:
  * kats.json holds PUBLIC known answers (SMHasher's MurmurHash3_x64_128 verification value,
    mmh3.hash64("foo"), ...) and the derived self-consistency vectors listed in SURVEY.md 8(c);
    it is written by hand below, not computed from the oracle;
  * stream_small.npz holds a seeded workload and the oracle's outputs for it, frozen so that a
    later change to oracle/ or to the generator cannot silently move the target.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

KATS = {
    "smhasher_verification_x64_128": "6384BA69",
    "murmur3_x64_128": [
        {"seed": 0, "key": "hello", "h1": "cbd8a7b341bd9b02", "h2": "5b1e906a48ae1d19"},
        {"seed": 0, "key": "foo", "h1": "e271865701f54561", "h2": "7eaf87e42bba7d87"},
        {"seed": 0, "key": "The quick brown fox jumps over the lazy dog", "h1": "e34bbc7bbc071b6c", "h2": "7a433ca9c49a9347"},
        {"seed": 0, "key": "ACGTACGTACGTACGT", "h1": "e183b34678e6d5b6"},
        {"seed": 42, "key": "ACGTACGTACGTACGT", "h1": "4152541eac055887"},
        {"seed": 0, "key": "AAAAAAAAAAAAAAAA", "h1": "37bd7653d0d19d9a"},
        {"seed": 42, "key": "AAAAAAAAAAAAAAAA", "h1": "bb475554b20a1d07"},
        {"seed": 0, "key": "ACGTTGCAAGGCTTAC", "h1": "4dd801fbf3fe95e1"},
        {"seed": 42, "key": "ACGTTGCAAGGCTTAC", "h1": "a432b4b4c7468131"},
    ],
    "mmh3_hash64_foo_signed": [-2129773440516405919, 9128664383759220103],
    "canonical_kmers": {
        "read": "ACGTTGCAAGGCTTACGGATCCAT", "k": 16, "seed": 0,
        "kmers": [
            [0, "ACGTTGCAAGGCTTAC", 0, "4dd801fbf3fe95e1"], [1, "CGTAAGCCTTGCAACG", 1, "a0e27dfb23a9803c"],
            [2, "CCGTAAGCCTTGCAAC", 1, "b52f66abbd6d9f3d"], [3, "TCCGTAAGCCTTGCAA", 1, "944aa26b0a5f0fd8"],
            [4, "ATCCGTAAGCCTTGCA", 1, "cf15585527d04e84"], [5, "GATCCGTAAGCCTTGC", 1, "6e4a8487c99775b3"],
            [6, "CAAGGCTTACGGATCC", 0, "31727c0d8bfc6ddd"], [7, "AAGGCTTACGGATCCA", 0, "0a0302820a001b81"],
            [8, "AGGCTTACGGATCCAT", 0, "9a99cc6612d9481e"]],
        "bottom4": ["0a0302820a001b81", "31727c0d8bfc6ddd", "4dd801fbf3fe95e1", "6e4a8487c99775b3"],
    },
    "normalise": {
        "read": "acgttgcaaggcttacNGGATCCATACGTTGCAAGGCTTAC\n",
        "normalised": "ACGTTGCAAGGCTTACNGGATCCATACGTTGCAAGGCTTAC",
        "valid_starts": [0, 17, 18, 19, 20, 21, 22, 23, 24, 25],
        "sketch": ["066680b00fd88c1c", "0c187bb04b425bb8", "4dd801fbf3fe95e1", "6a4036c3f21bd80b", "7ee6f3353f54f905",
                   "8341b15c9455f35d", "a5851d22d27da7f9", "c6ac3bb4507ed2fa", "ed78b1da05c6da23"],
    },
    "palindrome": {"kmer": "ACGTACGTACGTACGT", "is_rc": 1},
    "rank": {"sums": [3, 5, 5, 1], "order": [1, 2, 0, 3]},
}


def main():
    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump(KATS, f, indent=1)
    from helpers import workload
    from oracle import oracle as orc
    ref, bases, offsets = workload(48, 96, 40, read_len=600, genome_len=30000, rng_seed=1234)
    hashes = ref["ref"].copy()
    hashes[11] = hashes[2]          # a tie pair
    col_len = ref["col_len"].copy()
    col_len[5] = 17
    col_len[6] = 0
    exp = orc.stream(16, 0, 96, hashes, col_len, bases, offsets, top_k=4, want_shared=True, want_sketches=True)
    np.savez_compressed(os.path.join(HERE, "stream_small.npz"), k=16, seed=0, s=96, hashes=hashes, col_len=col_len,
                        bases=bases, offsets=offsets, cum=exp["cum"], topk_idx=exp["topk_idx"], topk_sum=exp["topk_sum"],
                        shared=exp["shared"], sketches=exp["sketches"], sketch_len=exp["sketch_len"])
    print("wrote kats.json, stream_small.npz")


if __name__ == "__main__":
    main()
