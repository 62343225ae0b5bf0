#!/bin/bash
# device assembly of the library's kernels into /tmp/isa/nkml.s, then the static instruction mix of the kernels matching $1
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I"$(dirname "$0")/../include" -w -S --cuda-device-only -o /tmp/isa/nkml.s "$(dirname "$0")/../nomalise_kmers_multi_large_amd/csrc/nkml_kernels.hip" 2>&1 | grep -E "error" 
python3 "$(dirname "$0")/isa_count.py" /tmp/isa/nkml.s "$1"
