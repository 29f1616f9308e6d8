#!/bin/bash
# compact per-kernel resource usage of one HIP source: tools/kres.sh csrc/file.hip [filter]
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
  grep -E "Function Name|VGPRs:|AGPRs|Occupancy|ScratchSize|SGPRs:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/Function Name: //' | \
  paste - - - - - - | while IFS=$'\t' read -r name rest; do echo "$(echo "$name" | c++filt | sed -E 's/\(.*//; s/void pds:://') | $(echo "$rest" | tr '\t' ' ')"; done | grep -E "${2:-.}"
