mkdir -p gpurun_out/r06a
python tools/smi_probe.py > gpurun_out/r06a/smi_idle.txt 2>&1
# probe under load
python bench.py --no-extras --no-cpu-baseline --steps 1500 --warmup 10 > gpurun_out/r06a/bench_long.json 2>gpurun_out/r06a/bench_long.err &
BP=$!
sleep 25
python tools/smi_probe.py > gpurun_out/r06a/smi_load.txt 2>&1
wait $BP
python tools/conv_bench.py --only s76_128-256_k3,s38_256-512_k3,s19_512,s76_256-128_k1,s38_512-256_k1,s19_1024 --variants igemm_v2,halo_ws,halo_ws_256,halo_dw,wres_1x1,igemm_v3_ns3 > gpurun_out/r06a/conv_bench.txt 2>&1
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --dump-ops gpurun_out/r06a/ops.txt > gpurun_out/r06a/bench_default.json 2>gpurun_out/r06a/bench_default.err
