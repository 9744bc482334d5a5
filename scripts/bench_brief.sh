#!/bin/bash
# prints the headline numbers of one bench.py run (for scripts/ab_run.sh)
python bench.py --no-cpu-baseline --no-pmc "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
r = d['roofline']
print('ms_per_step', d['ms_per_step'], 'value', d['value'], '| dominant', r.get('kernel'), r['achieved'], 'frac', r['frac'], '| conv_stack', d.get('roofline_conv_stack', {}).get('achieved'))
"
