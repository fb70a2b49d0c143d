#!/bin/bash
# Regenerate profiles/r<NN>_* on an MI355X box: the bench lines, the rocprofv3 kernel trace of bench.py (whole-run
# stats + per-step summary of the timed steps), the PMC HBM-traffic table (bench.py's own workload: the real network
# with the folded batch norms), the SQ instruction counters of the XNOR conv, the micro-benchmarks the DESIGN.md
# ceilings rest on (VALU issue rates, launch overhead) and the per-kernel micro-benchmark.
# usage (through gpurun): scripts/capture_profiles.sh r02   -> files under gpurun_out/, copy to profiles/
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=${1:-r03}; o=gpurun_out
mkdir -p $o
python bench.py --detail $o/${tag}_bench_n1_ls2_detail.json > $o/${tag}_bench_n1_ls2.json 2> $o/bench_ls2.err
python bench.py --act fp --no-configs --detail $o/${tag}_bench_n1_fpact_detail.json > $o/${tag}_bench_n1_fpact.json 2> $o/bench_fp.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 \
  --steps 100 --cpu-sample 0 --no-configs > $o/${tag}_bench_torchrun_n1_ls2.json 2> $o/bench_torchrun.err
rm -rf $o/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_final -- python bench.py --steps 30 --warmup 5 --min-seconds 0 --cpu-sample 0 --no-configs --streams 1 > $o/prof_final.log 2>&1
f=$(ls -t $o/prof_final/*/*kernel_trace.csv | head -1)
python scripts/trace_summary.py $f 10 > $o/${tag}_rocprofv3_per_step_summary.csv
cp $(ls -t $o/prof_final/*/*kernel_stats.csv | head -1) $o/${tag}_rocprofv3_kernel_stats_incl_warmup.csv
rm -rf $o/prof_fp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_fp -- python bench.py --act fp --steps 30 --warmup 5 --min-seconds 0 --cpu-sample 0 --no-configs --streams 1 > $o/prof_fp.log 2>&1
python scripts/trace_summary.py $(ls -t $o/prof_fp/*/*kernel_trace.csv | head -1) 10 > $o/${tag}_rocprofv3_per_step_summary_fpact.csv
scripts/pmc_traffic.sh $o/${tag}_pmc_hbm_traffic.json > $o/pmc.log 2>&1
scripts/pmc_traffic.sh $o/${tag}_pmc_hbm_traffic_fpact.json --act fp > $o/pmc_fp.log 2>&1
# SQ counters of the matrix-core kernels, one layer shape at a time (MFMA busy, VALU / LDS per MFMA, LDS conflicts, waits)
specs=""
for s in "64 56 64 1 4" "64 56 128 2 1" "128 28 128 1 3" "128 28 256 2 1" "256 14 256 1 3" "256 14 512 2 1" "512 7 512 1 3"; do
  set -- $s; t="C$1_H$2_s$4"
  scripts/pmc_kernel.sh xnor_mfma x_$t python scripts/xnor_one.py $1 $2 $3 $4 > $o/pmc_x_$t.txt 2>&1
  scripts/pmc_kernel.sh signw_conv_lean s_$t python scripts/signw_one.py $1 $2 $3 $4 > $o/pmc_s_$t.txt 2>&1
  specs="$specs lsq_xnor_conv2d:$t:$5:xnor_mfma:$o/pmc_x_$t lsq_signw_conv2d:$t:$5:signw_conv_lean:$o/pmc_s_$t"
done
scripts/pmc_kernel.sh stem_conv_pool stem python scripts/stem_one.py > $o/pmc_stem.txt 2>&1
python scripts/pmc_sq_table.py $o/${tag}_pmc_sq.json $specs lsq_stem_conv_pool:224x224:1:stem_conv_pool:$o/pmc_stem > $o/${tag}_pmc_sq.txt 2>&1
for u in valu_rates launch_overhead lds_atomics l2_retention; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/$u.hip -o /tmp/$u 2> $o/$u.build.log && /tmp/$u > $o/${tag}_ubench_$u.txt 2>&1
done
python scripts/kbench.py > $o/${tag}_kbench.txt 2>&1
python scripts/ls1_chain.py > $o/${tag}_ls1_chain.txt 2>&1
python scripts/xnor_variants.py > $o/${tag}_xnor_variants.txt 2>&1
python scripts/kbench.py --fold > $o/${tag}_kbench_bnfold.txt 2>&1
# the same launches as a network sees them: caches emptied in front of every launch, the launch's input read back in
python scripts/kbench.py --evict 600 --retouch > $o/${tag}_kbench_cold.txt 2>&1
# round 5: what runs under what when consecutive steps alternate between two HIP streams (the headline's issue order)
rm -rf $o/prof_ov
rocprofv3 --kernel-trace --output-format csv -d $o/prof_ov -- python bench.py --steps 30 --warmup 5 --min-seconds 0 --cpu-sample 0 --no-configs --no-roofline --streams 2 > $o/prof_ov.log 2>&1
python scripts/overlap_trace.py $(ls -t $o/prof_ov/*/*kernel_trace.csv | head -1) 10 110 > $o/${tag}_two_stream_overlap.txt 2>&1
python scripts/queue_gaps.py $(ls -t $o/prof_ov/*/*kernel_trace.csv | head -1) 150 0 > $o/${tag}_queue_gaps.txt 2>&1
rm -rf $o/prof_ov
# what the free-running parity assertions observe on this box, next to their derived limits
LSQ_RECORD_PARITY=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -q -m gpu > $o/parity_tests.log 2>&1
cp $o/free_running_observed.json $o/${tag}_free_running_parity.json
scripts/pmc_icache.sh $o/${tag}_pmc_icache.txt > $o/pmc_icache.log 2>&1
python scripts/sched_variants.py 40 > $o/${tag}_sched_variants.txt 2>&1
# round 6: the convolution's three implementations side by side (fp4 / int8 matrix cores, popcount), the known-answer test of the
# fp4 instruction, and the three-stream row layout (conv operand layouts in isolation, the network with it on and off)
for impl in 0 2 1; do python scripts/xnor_shapes.py $impl; done > $o/${tag}_xnor_shapes.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/fp4_mfma_check.hip -o /tmp/fp4_check 2> $o/fp4_check.build.log && /tmp/fp4_check > $o/${tag}_ubench_fp4_mfma_check.txt 2>&1
python scripts/xnor_s3.py > $o/${tag}_split3_conv_variants.txt 2>&1
python scripts/split3_ab.py > $o/${tag}_split3_ab.txt 2>&1
head -3 $o/${tag}_rocprofv3_per_step_summary.csv; cut -c1-300 $o/${tag}_bench_n1_ls2.json; cut -c1-160 $o/${tag}_bench_n1_fpact.json
