#!/bin/bash
# kernel resource report: tools/kres.sh file.hip  -> name, VGPRs, scratch, occupancy, LDS
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 \
 | grep -E "error|Function Name|  VGPRs:|ScratchSize|Occupancy|LDS Size" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(l)print l; l=$3} /VGPRs/{l=l" vgpr="$2} /Scratch/{l=l" scratch="$3} /Occupancy/{l=l" occ="$4} /LDS/{l=l" lds="$4} /error/{print} END{print l}' \
 | while read name rest; do echo "$(echo $name | c++filt | sed -E 's/\(anonymous namespace\):://; s/^void //; s/\(.*//') $rest"; done
