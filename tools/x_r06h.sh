mkdir -p gpurun_out/r06h
for rep in 1 2; do
for s in 3 2 4 5; do
python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-extras --streams $s > gpurun_out/r06h/bench_s${s}_$rep.json 2> gpurun_out/r06h/err.txt
done
done
