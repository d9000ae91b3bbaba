#!/bin/bash
# usage (on the GPU box): tools/ab_lib.sh <libA.so> <libB.so> [rounds]  -- same-call A/B of two builds of liba3r on the bench headline
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for L in $A $B; do
    A3R_LIB=$L python bench.py --steps 2 --no-cpu-baseline --no-align --no-cache-run --no-clip-run --no-bf16-run 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
k=r['kernels']
print('$L', r['value'], 'gemm', k['gemm_fh2_kernel (linear, split-fp16 MFMA)']['avg_us'], 'conv', k['gemm_fh2_kernel<1> (conv3x3, split-fp16 MFMA)']['avg_us'], 'attn', k['attn_fh2_kernel']['avg_us'], 'ln', k['layernorm_kernel']['avg_us'], 'elt', k['elementwise (patchify/upsample/head_final/pack)']['avg_us'])"
  done
done
