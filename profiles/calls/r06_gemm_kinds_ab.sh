#!/bin/bash
# A/B of the own GEMM (csrc/gemm_epilogue.hip) per kind of product inside the 4 x 3 bench (quick legs)
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
for k in kpconv_fwd kpconv_dx kpconv_gw unary_fwd unary_dx decoder; do run "ops.OWN_GEMM_KINDS={'$k'}"; done
run "ops.OWN_GEMM_KINDS=set()"
run "ops.OWN_GEMM_KINDS=set()" "models.architectures.UPSAMPLED_HEAD=False"
