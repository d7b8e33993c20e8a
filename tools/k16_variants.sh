#!/bin/bash
# GPU box: bit-identity (tools/k16_ab.py) + wall time of the bench launch for tagged variant builds of K16: tools/k16_variants.sh "a b c" <out>
out=gpurun_out/${2:-r06}/k16_variants.txt; mkdir -p $(dirname $out); : > $out
for t in $1; do
  L=pod_compare_amd/lib/s16$t/libpod_mi355x.so
  echo "== variant $t" | tee -a $out
  POD_MI355X_LIB=$L timeout 300 python tools/k16_ab.py 20 2>&1 | grep -E "form|IDENTICAL|DIFFER" | grep -v "^.*BIT-IDENTICAL  max" | tee -a $out
done
