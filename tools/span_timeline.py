#!/usr/bin/env python3
"""usage: span_timeline.py spans.txt [bin_ms]  -- the stages' device timeline out of the experiments build's SKX_SPAN_DUMP=1 output
(bench.py --profile-all): one row per stage, one column per bin, | = a span starts"""
import re
import sys
rows, blocks, cur = [], [], []
for l in open(sys.argv[1]):
    m = re.match(r'\[skx span\] (\S+)\s+([\d.]+)\s+([\d.]+)', l)
    if m:
        cur.append((m.group(1), float(m.group(2)), float(m.group(3))))
    elif '----' in l and cur:
        blocks.append(cur); cur = []
if cur:
    blocks.append(cur)
b = max(blocks, key=len)
w = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
t0, end = min(r[1] for r in b), max(r[2] for r in b)
bins = int((end - t0) / w) + 1
sym = {'sketch': 'S', 'dictionary': 'D', 'scan': 'C', 'transpose': 'T', 'rank': 'R'}
for st in sym:
    lanes = []  # overlapping spans of one stage go to separate rows
    for n, a, e in sorted(r for r in b if r[0] == st):
        for ln in lanes:
            if ln[-1][2] <= a:
                ln.append((n, a, e)); break
        else:
            lanes.append([(n, a, e)])
    for i, ln in enumerate(lanes):
        line = ['.'] * bins
        for n, a, e in ln:
            for j in range(int((a - t0) / w), int((e - t0) / w) + 1):
                line[j] = sym[st]
            line[int((a - t0) / w)] = '|'
        print(f'{st[:9]:9s}{i} ', ''.join(line))
print('total ms', round(end - t0, 3), ' sketch spans:', ' '.join(f'{e - a:.2f}' for n, a, e in b if n == 'sketch'))
