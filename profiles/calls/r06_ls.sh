#!/bin/bash
# lanes x stack sweep (quick legs): "L S" pairs as arguments
for cfg in "$@"; do
  set -- $cfg
  echo "== lanes $1 stack $2"
  timeout 400 python bench.py --quick --steps 20 --lanes $1 --stack $2 2>gpurun_out/bw.err | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d.get('value'), d['ms_per_step'], d['value_blocks']['median'], d.get('host_enqueue_ms_per_step'))
"
  grep -i "error\|Traceback" gpurun_out/bw.err | head -3
done
