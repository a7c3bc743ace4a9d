mkdir -p gpurun_out/r06i
python tools/conv_bench.py --only k3s2 --variants igemm_v2,igemm_v3_ns3,igemm_v3_ns4,igemm_v2_bn128 > gpurun_out/r06i/cb_s2.txt 2>&1
python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r06i/pipeline_tests.log 2>&1
