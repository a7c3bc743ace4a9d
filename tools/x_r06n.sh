mkdir -p gpurun_out/r06n
L="s76_128-256_k3,s38_256-512_k3,s19_512-1024_k3,s38_512-256_k1,s19_1024-512_k1,s38_768,s76_256-128_k1"
for v in "" d16; do
  if [ -n "$v" ]; then export Y3_HIP_LIB=$PWD/pytorch-yolov3_amd/lib/libyolov3_hip_$v.so; else unset Y3_HIP_LIB; fi
  for b in 1 2 4; do
  python tools/conv_bench.py --batch $b --only $L --variants igemm_v3_ns3,dw48 > gpurun_out/r06n/cb_${v:-d8}_b$b.txt 2>&1
  done
done
