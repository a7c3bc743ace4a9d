#!/bin/bash
# A/B of experiment builds (make variant ...): tools/x_variants.sh "<conv_bench args>" name1 name2 ...   ("" = the product library)
ARGS=$1; shift
for rep in 1 2; do
for n in "$@"; do
  L=pytorch-yolov3_amd/lib/libyolov3_hip${n:+_$n}.so
  echo "== ${n:-product}"
  Y3_HIP_LIB=$L timeout 300 python tools/conv_bench.py $ARGS 2>&1 | grep -v amdgpu | sed 's/ GF |/|/' | awk -F'|' '{printf "%s |", $1; for (i=2;i<=NF;i++) { split($i,a," "); if (a[1]!="") printf " %s %s us %s TF d=%s |", a[1], a[2]*1000, a[4], a[8] } printf "\n"}'
done
done
