#!/bin/bash
# usage (GPU box): tools/e2e.sh <bench args...>  -- bench.py's end-to-end leg only-ish (short timed region), printing value_end_to_end
python3 bench.py --cpu-seconds 0 --no-large-batch --steps 13 --reps 1 "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
e = d.get('value_end_to_end', {})
print('packed=%d hostfed=%d' % (d.get('value_host_fed_packed', {}).get('value', 0), d.get('value_host_fed', {}).get('value', 0)), {k: v for k, v in e.items() if k not in ('what', 'input')})"
