mkdir -p gpurun_out/r06r
python tools/conv_bench.py --only k1 --variants igemm_v2,dw_1x1,wres_1x1,dw48_always > gpurun_out/r06r/cb_k1.txt 2>&1
python tools/conv_bench.py --only s19_512-1024_k3,s38_256-512_k3,s76_128-256_k3 --variants halo_ws_256,halo_dw,dw48_always > gpurun_out/r06r/cb_k3.txt 2>&1
