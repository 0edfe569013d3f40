#!/bin/bash
# Summarise VGPR / spill / occupancy per kernel from `make asm` remarks.
cd "$(dirname "$0")/../nbmf_mm_amd/csrc" && make asm 2>&1 | awk '
/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
/ VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)}
/AGPRs:/ {a=$0; sub(/.*AGPRs: /,"",a); sub(/ \[.*/,"",a)}
/ScratchSize/ {s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s)}
/Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)}
/LDS Size/ {printf "%-70s vgpr=%s agpr=%s scratch=%s occ=%s\n", name, v, a, s, o}'
