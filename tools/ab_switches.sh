#!/bin/bash
# tools/ab_switches.sh <tag> : same-session A/Bs of the execution switches (bench.py twice per switch; median window + the five windows):
# side lane, staged form (5 stage graphs + tail graph) on one GPU, bf16 storage of the full-resolution internals, narrow shadow weights.
# Writes gpurun_out/<tag>/ab_switches.txt; a separate GPU session from make_profiles.sh (eight bench runs).
tag=${1:-r04}
out=gpurun_out/$tag
mkdir -p $out
( for s in 0 1; do python3 bench.py --no-cpu-baseline --prof-steps 0 --side-stream $s 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('side_stream=$s', d['ms_per_step'], d['windows_ms_per_step'])"; done
for o in 0 1; do python3 bench.py --no-cpu-baseline --prof-steps 0 --overlap $o 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('staged(5 graphs + tail graph)=$o', d['ms_per_step'], d['windows_ms_per_step'])"; done
for s in 0 1; do ADNM_BF16_STORAGE=$s python3 bench.py --no-cpu-baseline --prof-steps 0 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('ADNM_BF16_STORAGE=$s', d['ms_per_step'], d['windows_ms_per_step'])"; done
for s in 0 1; do ADNM_NARROW_WEIGHTS=$s python3 bench.py --no-cpu-baseline --prof-steps 0 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('ADNM_NARROW_WEIGHTS=$s', d['ms_per_step'], d['windows_ms_per_step'])"; done ) > $out/ab_switches.txt 2>&1
cat $out/ab_switches.txt
