#!/usr/bin/env python3
"""tools/sketch_stalls.py <pmc csv of tools/pmc_sketch.sh>  -- the main sketch kernel's SQ counters (per launch means) as fractions:
where its waves' cycles and the SIMDs' issue slots go.  Counter semantics as in MI355X_MICROARCH.md (SQ_* are summed over the chip's
SIMDs / waves; *_CYCLES in units of 4 clocks on gfx9 = one quad-cycle issue slot)."""
import csv
import sys

rows = {}
for r in csv.reader(open(sys.argv[1])):
    if len(r) >= 4 and "sketch_wave_kernel<16, 128, true>" in r[0]:
        rows[r[1]] = float(r[3])
g = rows.get
wc, busy = g("SQ_WAVE_CYCLES"), g("SQ_BUSY_CYCLES")
print("main sketch kernel, per launch (one C2 batch of 98 304 reads), alone on the chip:")
for k in sorted(rows):
    print(f"  {k:28s} {rows[k]:16.0f}")
if wc and busy:
    n_simd = 1024
    print(f"waves {g('SQ_WAVES'):.0f}; wave-cycles per busy cycle = {wc / busy:.1f} waves resident per SQ on average")
    for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC",
              "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM_RD"):
        if g(k) is not None:
            print(f"  {k:24s} / SQ_WAVE_CYCLES = {g(k) / wc:6.3f}")
    if g("SQ_ACTIVE_INST_VALU"):
        print(f"  VALU-active quad-cycles / (busy cycles x 4 SIMDs per SQ-busy unit): {g('SQ_ACTIVE_INST_VALU') / busy / 4:6.3f}  (issue-slot utilisation of the VALU)")
    if g("SQ_INSTS_VALU"):
        print(f"  instructions per wave: VALU {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.0f}, SALU {g('SQ_INSTS_SALU', 0) / g('SQ_WAVES'):.0f}, LDS {g('SQ_INSTS_LDS', 0) / g('SQ_WAVES'):.0f}, "
              f"VMEM_RD {g('SQ_INSTS_VMEM_RD', 0) / g('SQ_WAVES'):.0f}")
        if g("SQ_ACTIVE_INST_VALU"):
            print(f"  quad-cycles per VALU instruction: {g('SQ_ACTIVE_INST_VALU') / g('SQ_INSTS_VALU'):.2f}")
    for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL"):
        if g(k) is not None and g("SQ_LDS_IDX_ACTIVE"):
            print(f"  {k:24s} / SQ_LDS_IDX_ACTIVE = {g(k) / g('SQ_LDS_IDX_ACTIVE'):6.3f}")
