export TMPDIR=/tmp; mkdir -p gpurun_out/r4t
BF="--no-cpu-baseline --no-sac-step --no-c5 --no-overlap-ab --no-small-batch"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r4t/pf -- python3 bench.py --steps 3 --warmup 1 $BF > /dev/null 2> gpurun_out/r4t/pf.err && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r4t/pw -- python3 bench.py --steps 3 --warmup 1 $BF > /dev/null 2> gpurun_out/r4t/pw.err && \
head -1 $(ls gpurun_out/r4t/pf/*/*counter_collection.csv | head -1) > gpurun_out/r4t/header.txt && \
python tools/pmc_traffic_by_launch.py gpurun_out/r4t/pf gpurun_out/r4t/pw > gpurun_out/r4t/by_launch.txt
rm -rf gpurun_out/r4t/pf gpurun_out/r4t/pw
cat gpurun_out/r4t/header.txt; cat gpurun_out/r4t/by_launch.txt
