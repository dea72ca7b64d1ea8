#!/bin/bash
# tools/install_profiles.sh <tag> : copy what tools/make_profiles.sh <tag> left under gpurun_out/<tag>/ (merged back from the GPU box) into
# profiles/ under the judged names (gpurun_out/ is scratch, profiles/ is tracked).
set -e
tag=${1:-r04}
src=gpurun_out/$tag
for f in bench_bf16.json bench_f32.json bench_fp8.json bench_tables_bf16.json bench_tables_f32.json bench_tables_fp8.json fp8_error.txt kernel_shapes_bf16.tsv kernel_shapes_f32.tsv kernel_shapes_fp8.tsv kernel_stats_bf16.csv steady_state_bf16.txt small_grids_bf16.txt \
         tsgemm.txt gemm_bench.txt pmc_traffic.txt dwconv_storage.txt two_streams.txt ab_switches.txt k1_counters.txt; do
  [ -f $src/$f ] && cp $src/$f profiles/${tag}_$f || echo "(missing $f)"
done
if [ -f $src/pmc_traffic.json ]; then cp $src/pmc_traffic.json profiles/pmc_traffic.json; cp $src/pmc_traffic.json profiles/${tag}_pmc_traffic.json; fi
grep -v "amdgpu.ids\|UserWarning\|Consider using\|res\[prec\]" $src/bf16_parity.txt > profiles/${tag}_bf16_parity.txt
ls -la profiles | grep ${tag}_
