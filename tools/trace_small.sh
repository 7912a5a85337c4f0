# per-kernel durations of a single-frame / small-batch policy.sample() on the product library
mkdir -p gpurun_out/r4e && export TMPDIR=/tmp
O=gpurun_out/r4e
for B in 1 2 32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$B -- python3 tools/small_batch_ab.py trace $B > /dev/null 2> $O/t$B.err
  cp $(ls $O/t$B/*/*kernel_stats.csv | head -1) $O/small_batch_B${B}_kernel_stats.csv
  rm -rf $O/t$B
done
for B in 1 2 32; do echo "== B=$B"; cut -d, -f1-4 $O/small_batch_B${B}_kernel_stats.csv | head -14; done
