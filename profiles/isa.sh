#!/bin/bash
# profiles/isa.sh <mangled-prefix> <out.s> -- (re)build the gfx950 assembly of vsx_device.hip and cut out one kernel
set -e
make -C /root/repo/vsearch_amd/csrc asm 2>&1 | grep -E "error" -A5 || true
awk -v pat="^$1" '$0 ~ pat {p=1} p {print} p && /s_endpgm/ {exit}' /root/repo/build/asm/*gfx950*.s > "$2"
wc -l "$2"
