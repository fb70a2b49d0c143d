#!/bin/bash
# Regenerate profiles/r<NN>_* on an MI355X box: both bench lines, the rocprofv3 kernel trace of bench.py
# (whole-run stats + per-step summary of the timed steps) and the PMC HBM-traffic table.
# usage (through gpurun): scripts/capture_profiles.sh r01   -> files under gpurun_out/, copy to profiles/
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=${1:-r01}; o=gpurun_out
mkdir -p $o
python bench.py > $o/${tag}_bench_n1_ls2.json 2> $o/bench_ls2.err
python bench.py --act fp > $o/${tag}_bench_n1_fpact.json 2> $o/bench_fp.err
rm -rf $o/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_final -- python bench.py --cpu-sample 0 > $o/prof_final.log 2>&1
f=$(ls -t $o/prof_final/*/*kernel_trace.csv | head -1)
python scripts/trace_summary.py $f 10 > $o/${tag}_rocprofv3_per_step_summary.csv
cp $(ls -t $o/prof_final/*/*kernel_stats.csv | head -1) $o/${tag}_rocprofv3_kernel_stats_incl_warmup.csv
rm -rf $o/prof_fp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_fp -- python bench.py --act fp --cpu-sample 0 > $o/prof_fp.log 2>&1
python scripts/trace_summary.py $(ls -t $o/prof_fp/*/*kernel_trace.csv | head -1) 10 > $o/${tag}_rocprofv3_per_step_summary_fpact.csv
scripts/pmc_traffic.sh $o/${tag}_pmc_hbm_traffic.json > $o/pmc.log 2>&1
python scripts/kbench.py > $o/${tag}_kbench.txt 2>&1
head -3 $o/${tag}_rocprofv3_per_step_summary.csv; cut -c1-160 $o/${tag}_bench_n1_ls2.json; cut -c1-160 $o/${tag}_bench_n1_fpact.json
