#!/bin/bash
# disasm_kernel.sh <lib.so> <substring of the demangled kernel name> [out.s] : disassemble ONE kernel of the gfx950 code object
set -e
lib=$(readlink -f "$1"); tmp=$(mktemp -d); cp "$lib" "$tmp/lib.so"
( cd "$tmp" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null 2>&1 )
sym=""
for c in $(ls "$tmp" | grep gfx950); do         # one code object per translation unit
co=$tmp/$c
sym=$(/opt/rocm/lib/llvm/bin/llvm-readelf -s -W "$co" | awk '$4=="FUNC" && $5=="GLOBAL"{print $8}' | while read m; do d=$(c++filt "$m"); case "$d" in *"$2"*) echo "$m"; break;; esac; done)
[ -n "$sym" ] && break
done
[ -n "$sym" ] || { echo "no kernel matches $2" >&2; exit 1; }
echo "# $(c++filt $sym)" > "${3:-/dev/stdout}"
/opt/rocm/lib/llvm/bin/llvm-objdump -d --disassemble-symbols="$sym" "$co" >> "${3:-/dev/stdout}"
rm -rf "$tmp"
