#!/bin/bash
# tools/pin_from_reference.sh -- turns "PARITY UNPINNED" into a pinned oracle, on a machine that HAS a Rust toolchain
# and the reference's crates (this image has neither: no cargo/rustc, no network; DESIGN.md "Oracle").
#
# What it does, in one command:
#   1. builds the reference (esteinig/sketchy, Rust) with cargo from $REF (default /root/reference) into oracle/_ref/
#      (outputs only; no reference source is copied into this repo);
#   2. writes the golden inputs as REAL files: genome FASTAs, a read FASTQ (edge cases of tests/test_gpu_parity.py
#      included), a genotype TSV -- tools/pin_inputs.py, seeded, no reference code involved;
#   3. runs the reference:      sketchy sketch   (FASTA -> .msh, s = 1000 / 10000, k = 16, seeds 0 and 42)
#                               sketchy info -p  (names, sketch parameters)
#                               sketchy shared   (all-pairs intersections)
#                               sketchy predict -s -t 5 [-H]   (the hot path: rows after every read)
#      and the same four with this repo's host (sketchy_amd/sketchy-hip) on the SAME files -- including
#      sketchy-hip reading the .msh the REFERENCE wrote and the reference reading the .msh sketchy-hip wrote
#      (pins the Cap'n Proto layout, which so far has only been round-tripped against our own writer);
#   4. diffs every pair of outputs byte for byte, and freezes the reference's outputs under tests/golden/ref_pinned/
#      (data: inputs and expected outputs) so that tests/test_oracle.py::test_reference_pinned_vectors -- skipped while
#      the directory is absent -- pins oracle/oracle.c against them from then on.
#
# Exit code 0 = every diff empty (parity pinned); non-zero = the first difference is printed.
set -euo pipefail
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/oracle/_ref
W=${WORK:-$(mktemp -d /tmp/skx_pin.XXXXXX)}
HIP=$ROOT/sketchy_amd/sketchy-hip

command -v cargo >/dev/null || { echo "cargo not found: this recipe needs a Rust toolchain (and the crates of $REF/Cargo.lock)"; exit 3; }
[ -x "$HIP" ] || python3 -m sketchy_amd.build
mkdir -p "$OUT" "$W"

echo "== 1. build the reference ($REF) -> $OUT"
( cd "$REF" && cargo build --release --locked --target-dir "$OUT/target" )
SK=$OUT/target/release/sketchy
"$SK" --version || true

echo "== 2. golden inputs -> $W"
python3 "$ROOT/tools/pin_inputs.py" "$W"     # genomes/*.fa, genomes.txt, reads.fq, reads_edge.fq, genotypes.tsv

fail=0
cmp_out() {  # cmp_out <label> <reference output> <our output>
    if cmp -s "$2" "$3"; then echo "   same  $1"; else echo "   DIFF  $1"; diff "$2" "$3" | head -5; fail=1; fi
}

for S in 1000 10000; do for SEED in 0 42; do
    T=s${S}_e${SEED}
    echo "== 3. $T"
    "$SK"  sketch -i $(cat "$W/genomes.txt") -o "$W/ref_$T.msh" -s $S -k 16 -e $SEED
    "$HIP" sketch -i $(cat "$W/genomes.txt") -o "$W/hip_$T.msh" -s $S -k 16 -e $SEED
    # sketch parameters + genome order, each binary on each file
    for F in ref hip; do
        "$SK"  info -i "$W/${F}_$T.msh" -p > "$W/info_ref_on_$F.$T.txt"
        "$HIP" info -i "$W/${F}_$T.msh" -p > "$W/info_hip_on_$F.$T.txt"
    done
    cmp_out "info -p on the reference's .msh ($T)" "$W/info_ref_on_ref.$T.txt" "$W/info_hip_on_ref.$T.txt"
    cmp_out "info -p on our .msh ($T)"             "$W/info_ref_on_hip.$T.txt" "$W/info_hip_on_hip.$T.txt"
    # all-pairs shared hashes (src/sketchy.rs:238-279)
    "$SK"  shared -r "$W/ref_$T.msh" -q "$W/ref_$T.msh" > "$W/shared_ref.$T.txt"
    "$HIP" shared -r "$W/ref_$T.msh" -q "$W/ref_$T.msh" > "$W/shared_hip.$T.txt"
    "$SK"  shared -r "$W/hip_$T.msh" -q "$W/ref_$T.msh" > "$W/shared_ref_x.$T.txt"
    cmp_out "shared ($T)"                                   "$W/shared_ref.$T.txt" "$W/shared_hip.$T.txt"
    cmp_out "shared, our .msh read by the reference ($T)"   "$W/shared_ref.$T.txt" "$W/shared_ref_x.$T.txt"
    # the hot path: rows after every read (src/sketchy.rs:317-356, :391-398)
    for FQ in reads reads_edge; do for TOP in 1 5; do
        "$SK"  predict -i "$W/$FQ.fq" -r "$W/ref_$T.msh" -g "$W/genotypes.tsv" -t $TOP -s -H > "$W/pred_ref.$T.$FQ.$TOP.tsv"
        "$HIP" predict -i "$W/$FQ.fq" -r "$W/ref_$T.msh" -g "$W/genotypes.tsv" -t $TOP -s -H > "$W/pred_hip.$T.$FQ.$TOP.tsv"
        cmp_out "predict -s -t $TOP $FQ ($T)" "$W/pred_ref.$T.$FQ.$TOP.tsv" "$W/pred_hip.$T.$FQ.$TOP.tsv"
    done; done
    # offline mode (pooled sketch, src/sketchy.rs:281-315)
    "$SK"  predict -i "$W/reads.fq" -r "$W/ref_$T.msh" -g "$W/genotypes.tsv" -t 5 > "$W/off_ref.$T.tsv"
    "$HIP" predict -i "$W/reads.fq" -r "$W/ref_$T.msh" -g "$W/genotypes.tsv" -t 5 > "$W/off_hip.$T.tsv"
    cmp_out "predict (offline) -t 5 ($T)" "$W/off_ref.$T.tsv" "$W/off_hip.$T.tsv"
done; done

echo "== 4. freeze the reference's outputs as golden vectors"
G=$ROOT/tests/golden/ref_pinned
mkdir -p "$G"
cp "$W"/genotypes.tsv "$W"/reads.fq "$W"/reads_edge.fq "$G/"
cp "$W"/ref_s1000_e0.msh "$W"/ref_s1000_e42.msh "$G/"                 # small enough to commit (s = 1000)
cp "$W"/pred_ref.s1000_e*.tsv "$W"/shared_ref.s1000_e*.txt "$W"/info_ref_on_ref.s1000_e*.txt "$W"/off_ref.s1000_e*.tsv "$G/"
( cd "$REF" && git rev-parse HEAD 2>/dev/null || true; cargo --version; rustc --version ) > "$G/PROVENANCE.txt"
echo "frozen under $G (commit it; tests/test_oracle.py::test_reference_pinned_vectors then pins oracle/oracle.c)"
[ $fail -eq 0 ] && echo "PARITY PINNED: every output identical" || echo "DIFFERENCES FOUND (see above)"
exit $fail
