#!/bin/bash
# list_kernels.sh <lib.so> : the kernels of the gfx950 code object inside a built library, one demangled name per line
set -e
lib=$(readlink -f "$1"); tmp=$(mktemp -d); cp "$lib" "$tmp/lib.so"
( cd "$tmp" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null 2>&1 )
for co in $(ls "$tmp" | grep gfx950); do      # one code object per translation unit
/opt/rocm/lib/llvm/bin/llvm-readelf -s -W "$tmp/$co" | awk '$4=="FUNC" && $5=="GLOBAL" && $6=="PROTECTED"{print $8}' | c++filt | sed 's/^void //'
done
rm -rf "$tmp"
