# config 5 (bf16) forward at B = 440: kernel stats, MFMA utilisation, L2 hit rate and HBM traffic per launch (separate --pmc passes)
mkdir -p gpurun_out/r4c5 && export TMPDIR=/tmp
O=gpurun_out/r4c5
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 tools/c5_step.py fwd 440 7 > $O/c5.log 2>&1 && \
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pm -- python3 tools/c5_step.py fwd 440 2 > /dev/null 2> $O/pm.err && \
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pl -- python3 tools/c5_step.py fwd 440 2 > /dev/null 2> $O/pl.err && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 tools/c5_step.py fwd 440 2 > /dev/null 2> $O/pf.err && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python3 tools/c5_step.py fwd 440 2 > /dev/null 2> $O/pw.err && \
cp $(ls $O/ks/*/*kernel_stats.csv | head -1) $O/c5_kernel_stats.csv && \
python tools/pmc_mfma.py $O/pm $O/mfma_c5.json > $O/mfma_c5.txt && \
python tools/pmc_l2.py $O/pl $O/l2_c5.json > $O/l2_c5.txt && \
python tools/pmc_traffic.py $O/pf $O/pw $O/hbm_traffic_c5.json > $O/hbm_traffic_c5.txt && \
rm -rf $O/ks $O/pm $O/pl $O/pf $O/pw
echo "rc=$?"; cat $O/mfma_c5.txt $O/l2_c5.txt $O/hbm_traffic_c5.txt; tail -2 $O/c5.log
