#!/bin/bash
# usage: tools/kasm.sh <mangled-substring> > out.s   -- extract one kernel's ISA from build/nbmf_hip.s
f="$(dirname "$0")/../build/nbmf_hip.s"
awk -v pat="$1" '
  $0 ~ "^_ZN.*" pat ".*:" && !on {on=1}
  on {print}
  on && /^\.Lfunc_end/ {exit}' "$f"
