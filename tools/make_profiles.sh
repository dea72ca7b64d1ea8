#!/bin/bash
# tools/make_profiles.sh <tag> <commit> : every judged measurement of a round in one go, on the GPU box (through gpurun), from the repo root.
#   profiles/<tag>_bench_{bf16,f32}.json        bench.py lines (bf16 = the headline, with cpu_baseline)
#   profiles/<tag>_kernel_shapes_{bf16,f32}.tsv per-(kernel, shape) HIP-event table of the same runs
#   profiles/<tag>_kernel_stats_bf16.csv        rocprofv3 --kernel-trace --stats summary of bench.py (whole process)
#   profiles/<tag>_steady_state_bf16.txt        tools/kstats.py: last 3 steps of that trace, per kernel
#   profiles/<tag>_small_grids_bf16.txt         tools/kclass.py: launches with <= 64 workgroups
#   profiles/pmc_traffic.json               tools/pmc_traffic.py from two --pmc passes (FETCH_SIZE, WRITE_SIZE)
#   profiles/<tag>_bf16_parity.txt              tools/bf16_error.py: measured error of the bf16 mode on the whole model
# The PMC passes run BEFORE the judged bench line so that the line can carry `traffic` from the same kernel sources.
set -o pipefail
tag=${1:-r04}
out=gpurun_out/$tag
mkdir -p $out
root=$(pwd)
commit=${2:-unknown}
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/$out/pmc_fetch -o f -- python3 $root/bench.py --steps 2 --warmup 2 --prof-steps 0 --no-cpu-baseline > $root/$out/pmc_fetch.log 2>&1 ) || { echo "PMC fetch pass failed"; tail -n 5 $out/pmc_fetch.log; }
( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $root/$out/pmc_write -o w -- python3 $root/bench.py --steps 2 --warmup 2 --prof-steps 0 --no-cpu-baseline > $root/$out/pmc_write.log 2>&1 ) || { echo "PMC write pass failed"; tail -n 5 $out/pmc_write.log; }
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json $commit > $out/pmc_traffic.txt 2>&1 && cp $out/pmc_traffic.json profiles/pmc_traffic.json
tail -n 3 $out/pmc_traffic.txt
python3 bench.py --dump-json $out/bench_tables_bf16.json --dump-prof $out/kernel_shapes_bf16.tsv > $out/bench_bf16.json 2> $out/bench_bf16.err || { echo "bench bf16 failed"; tail -n 5 $out/bench_bf16.err; exit 1; }
python3 bench.py --dtype f32 --no-cpu-baseline --dump-json $out/bench_tables_f32.json --dump-prof $out/kernel_shapes_f32.tsv > $out/bench_f32.json 2> $out/bench_f32.err || { echo "bench f32 failed"; exit 1; }
python3 bench.py --dtype fp8 --no-cpu-baseline --dump-json $out/bench_tables_fp8.json --dump-prof $out/kernel_shapes_fp8.tsv > $out/bench_fp8.json 2> $out/bench_fp8.err || { echo "bench fp8 failed"; exit 1; }
python3 tools/fp8_error.py 2>&1 | grep -v amdgpu.ids > $out/fp8_error.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -o t -- python3 $root/bench.py --steps 20 --warmup 10 --prof-steps 0 --no-cpu-baseline > $root/$out/trace.log 2>&1 ) || { echo "trace pass failed"; tail -n 5 $out/trace.log; exit 1; }
python3 tools/bf16_error.py > $out/bf16_parity.txt 2>&1
python3 tools/kbench_ts.py 2>&1 | grep -v amdgpu.ids > $out/tsgemm.txt
python3 tools/kbench_dw.py 2>&1 | grep -v amdgpu.ids > $out/dwconv_storage.txt
bash tools/k1_counters.sh $out/k1_counters.txt > /dev/null 2>&1
python3 tools/kbench_twostream.py 2>&1 | grep -v amdgpu.ids > $out/two_streams.txt
python3 tools/kbench_gemm.py 2>&1 | grep -v amdgpu.ids > $out/gemm_bench.txt
python3 tools/kstats.py $out/trace 80 3 80 > $out/steady_state_bf16.txt 2>&1
python3 tools/kclass.py $out/trace 64 > $out/small_grids_bf16.txt 2>&1
find $out/trace -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} $out/kernel_stats_bf16.csv
rm -rf $out/trace/*/*_kernel_trace.csv $out/pmc_fetch $out/pmc_write $out/trace
cut -c1-260 $out/bench_bf16.json
cut -c1-260 $out/bench_f32.json
cut -c1-260 $out/bench_fp8.json
head -n 12 $out/steady_state_bf16.txt
