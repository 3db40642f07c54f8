#!/bin/bash
# A/B of the own-GEMM shape rules inside the 4 x 3 bench (quick legs)
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
run "ops.OWN_GEMM_KINDS=set()"
run "ops.OWN_GEMM_KINDS={'kpconv_fwd'}" "ops.OWN_GEMM_RULES=False"
run "ops.OWN_GEMM_KINDS={'kpconv_fwd'}"
run "ops.OWN_GEMM_KINDS={'kpconv_fwd','unary_fwd'}"
run "ops.OWN_GEMM_KINDS={'kpconv_fwd','unary_dx'}"
run "ops.OWN_GEMM_KINDS={'kpconv_fwd','unary_fwd','unary_dx'}"
run "ops.OWN_GEMM_KINDS=set()"
