#!/bin/bash
# usage (GPU box): tools/hostfed.sh B...  -- the host-fed legs of bench.py (skx_stream_submit from page-locked memory) at other batch sizes
for B in "$@"; do
python3 bench.py --cpu-seconds 0 --no-large-batch --no-end-to-end --steps 8 --reps 2 --batch $B 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=$B value=%d packed=%d hostfed=%d steady=%d' % (d['value'], d.get('value_host_fed_packed', {}).get('value', 0), d.get('value_host_fed', {}).get('value', 0), d.get('value_steady_state', {}).get('value', 0)))"
done
