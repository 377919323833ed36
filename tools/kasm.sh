#!/bin/bash
# one kernel's ISA out of /tmp/<unit>.s (made by tools/asm.sh): tools/kasm.sh <unit> <mangled-name-fragment> > file
U=$1; K=$2
awk -v k="$K" '$0 ~ "^_ZN.*" k ".*:" {p=1} p {print} p && /^\.Lfunc_end/ {exit}' /tmp/$U.s
