#!/bin/bash
# One kernel's text out of a `hipcc -S` listing:  tools/isa_extract.sh file.s <mangled-name substring> > kernel.s
awk -v key="$2" '$0 ~ "^_Z" && index($0, key) && /:/ && !f {f=1} f{print} f && /^\.Lfunc_end/ {exit}' "$1"
