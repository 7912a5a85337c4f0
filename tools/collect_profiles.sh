# bash tools/collect_profiles.sh [pmc|bench|all]  (default all).  bench.py quotes roofline.traffic from profiles/*hbm_traffic.json only when
# that file was measured on the same GEMM sources: collect `pmc` first, copy hbm_traffic.json into profiles/ with tools/stamp_profile.py
# (in the build container: the GPU box has no .git), then run `bench`.
STAGE=${1:-all}
mkdir -p gpurun_out/r4p && export TMPDIR=/tmp
rc=0
if [ "$STAGE" != bench ]; then
BF="--no-cpu-baseline --no-sac-step --no-c5 --no-overlap-ab --no-small-batch"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4p/ks -- python3 bench.py --steps 10 --warmup 2 $BF > gpurun_out/r4p/bench_under_rocprof.json 2> gpurun_out/r4p/ks.err && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r4p/pf -- python3 bench.py --steps 3 --warmup 1 $BF > /dev/null 2> gpurun_out/r4p/pf.err && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r4p/pw -- python3 bench.py --steps 3 --warmup 1 $BF > /dev/null 2> gpurun_out/r4p/pw.err && \
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r4p/pm -- python3 bench.py --steps 3 --warmup 1 $BF > /dev/null 2> gpurun_out/r4p/pm.err && \
python tools/pmc_traffic.py gpurun_out/r4p/pf gpurun_out/r4p/pw gpurun_out/r4p/hbm_traffic.json > gpurun_out/r4p/hbm_traffic.txt && \
python tools/pmc_mfma.py gpurun_out/r4p/pm gpurun_out/r4p/mfma_c3.json > gpurun_out/r4p/mfma_c3.txt && \
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4p/c5 -- python3 tools/c5_step.py fwd 440 7 > gpurun_out/r4p/c5.log 2>&1 && \
cp $(ls gpurun_out/r4p/ks/*/*kernel_stats.csv | head -1) gpurun_out/r4p/kernel_stats.csv && cp $(ls gpurun_out/r4p/c5/*/*kernel_stats.csv | head -1) gpurun_out/r4p/c5_kernel_stats.csv && \
python tools/step_timeline.py gpurun_out/r4p/ks > gpurun_out/r4p/step_timeline.txt && \
rm -rf gpurun_out/r4p/ks gpurun_out/r4p/pf gpurun_out/r4p/pw gpurun_out/r4p/pm gpurun_out/r4p/c5
rc=$?
fi
if [ "$STAGE" != pmc ] && [ $rc -eq 0 ]; then
python bench.py > gpurun_out/r4p/bench.json 2> gpurun_out/r4p/bench.err && \
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 20 --warmup 5 --force-collective --no-cpu-baseline --no-sac-step --no-c5 --no-overlap-ab --no-small-batch > gpurun_out/r4p/bench_rccl_1rank.json 2> gpurun_out/r4p/bench_rccl.err
rc=$?
fi
echo "rc=$rc"; tail -c 600 gpurun_out/r4p/bench.json; cat gpurun_out/r4p/hbm_traffic.txt gpurun_out/r4p/mfma_c3.txt | head -20; tail -c 400 gpurun_out/r4p/bench_rccl_1rank.json
