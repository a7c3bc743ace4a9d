# tools-style multi-variant A/B of the strip kernel: base + variants, interleaved, 2 reps
L=pytorch-yolov3_amd/lib
for rep in 1 2; do
  for n in "" _xw2 _xw16 _xw17 _xw1 _xw18 _xh2 _xh16 _xw2h2 _xw2h16; do
    echo "== lib$n"
    Y3_HIP_LIB=$L/libyolov3_hip$n.so timeout 200 python tools/conv_bench.py --only s76_128-256_k3,s38_256-512_k3,s19_512 --variants halo_ws_256 --rounds 3 2>&1 | grep -v amdgpu | awk '{printf "%-24s %s TF\n", $1, $8}'
  done
done
