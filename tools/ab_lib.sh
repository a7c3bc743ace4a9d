# Same-box A/B of two builds of the library: tools/ab_lib.sh <libA.so> <libB.so> <conv_bench args...>
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in $A $B; do
    echo "== $L"
    Y3_HIP_LIB=$L timeout 200 python tools/conv_bench.py "$@" 2>&1 | grep -v amdgpu | sed 's/ GF |/|/' | awk -F'|' '{printf "%s |", $1; for (i=2;i<=NF;i++) { split($i,a," "); if (a[1]!="") printf " %s %s TF |", a[1], a[4] } printf "\n"}'
  done
done
