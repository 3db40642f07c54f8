#!/bin/bash
# generic A/B inside the 4 x 3 bench (quick legs): every argument is one configuration = space-separated overrides for
# profiles/bench_with.py ("-" = none)
run() {
  echo "== $*"
  timeout 400 python profiles/bench_with.py "$@" -- --quick --steps 20 2>gpurun_out/bw.err | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d.get('value'), d['value_blocks']['median'], d['value_blocks']['min'], d['value_blocks']['max'], d['one_pair_in_flight']['value'])
"
  grep -i "error\|Traceback" gpurun_out/bw.err | head -3
}
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then run; else run $cfg; fi
done
