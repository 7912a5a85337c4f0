# stream GEMM tile walk: timing sweep + per-shape L2 hit rate / bytes beyond the L2s for the old walk (budget 0) and the shipped budget
mkdir -p gpurun_out/r4a && export TMPDIR=/tmp
O=gpurun_out/r4a
python tools/bf16_walk_budget.py 440 > $O/walk_budget_sweep.txt 2>&1 || exit 1
for BUD in 0 2048; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pl$BUD -- python3 tools/bf16_walk_budget.py pmc $BUD 440 3 > /dev/null 2> $O/pl.err && \
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf$BUD -- python3 tools/bf16_walk_budget.py pmc $BUD 440 3 > /dev/null 2> $O/pf.err && \
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw$BUD -- python3 tools/bf16_walk_budget.py pmc $BUD 440 3 > /dev/null 2> $O/pw.err && \
  python tools/pmc_walk.py $O/pl$BUD $O/pf$BUD $O/pw$BUD 3 440 $O/walk_l2_budget_$BUD.json > $O/walk_l2_budget_$BUD.txt 2>&1
  rm -rf $O/pl$BUD $O/pf$BUD $O/pw$BUD
done
cat $O/walk_budget_sweep.txt $O/walk_l2_budget_0.txt $O/walk_l2_budget_2048.txt
