#!/bin/bash
# wave-residency per kernel (rocprofv3 serialises kernels under --pmc: these are stand-alone figures)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $P/pmc_w -o pmc -- python3 bench.py --steps 6 --warmup 2 --reps 1 --cpu-seconds 0 --no-extra-legs --no-check "$@" > /dev/null 2> $P/pmc_w.err
python3 profiles/summarize_pmc.py $(find $P/pmc_w -name "*counter_collection.csv") > $P/pmc_waves_summary.csv
rm -rf $P/pmc_w
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof/pmc_waves_summary.csv')))
t={}
for r in rows:
    if 'skx::' not in r['kernel'] or float(r['dispatches'])<8: continue
    n=r['kernel'].split('skx::')[1].split('<')[0].strip('"')
    t.setdefault(n,{}); t[n][r['counter']]=t[n].get(r['counter'],0)+float(r['total_value'])/8
tot={}
for n in sorted(t,key=lambda k:-t[k].get('SQ_WAVE_CYCLES',0)):
    d=t[n]
    print(f"{n:26s} waves {d.get('SQ_WAVES',0)/1e3:8.1f}k wave_cycles {d.get('SQ_WAVE_CYCLES',0)/1e9:7.3f}G busy {d.get('SQ_BUSY_CYCLES',0)/1e6:8.1f}M wait_inst {d.get('SQ_WAIT_INST_ANY',0)/1e9:7.3f}G active_valu {d.get('SQ_ACTIVE_INST_VALU',0)/1e9:7.3f}G active_any {d.get('SQ_ACTIVE_INST_ANY',0)/1e9:7.3f}G")
    for k,v in d.items(): tot[k]=tot.get(k,0)+v
print("sum", {k:round(v/1e9,3) for k,v in tot.items()})
PY
