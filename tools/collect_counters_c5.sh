mkdir -p gpurun_out/r4q && export TMPDIR=/tmp
O=gpurun_out/r4q
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/p1 -- python3 tools/c5_step.py fwd 440 2 > /dev/null 2> $O/p1.err && \
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/p2 -- python3 tools/c5_step.py fwd 440 2 > /dev/null 2> $O/p2.err
python tools/pmc_raw.py $O/p1 attn_fwd_bf16_stream gemm_bf16_stream_kernel layernorm_fwd_bf16 > $O/counters_a.txt 2>&1
python tools/pmc_raw.py $O/p2 attn_fwd_bf16_stream gemm_bf16_stream_kernel layernorm_fwd_bf16 > $O/counters_b.txt 2>&1
rm -rf $O/p1 $O/p2
cat $O/counters_a.txt $O/counters_b.txt
for B in 220 256 512; do python tools/bf16_bench.py --batch $B --tiles 256257 2>&1 | grep -E "^forward_c5|^fwd_bwd|^qkv|^fc1|^fc2|^out|^attention" | sed "s/^/B=$B /"; done > $O/batch_sweep.txt 2>&1; cat $O/batch_sweep.txt
