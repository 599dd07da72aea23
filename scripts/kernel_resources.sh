#!/bin/bash
# Register / LDS use of every kernel in one translation unit of libmdx (no GPU needed):
#   scripts/kernel_resources.sh mdx_rdf [pattern]
# Recompiles the unit with --save-temps in a scratch directory and reads the code-object notes.
set -e
unit=${1:-mdx_rdf}
pat=${2:-.}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
cd "$tmp"
extra=""
[ "$unit" = "mdx_rdf" ] && extra="-ffp-contract=off"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$root/include" $extra --save-temps \
    -c "$root/mdhelper_amd/csrc/$unit.hip" -o out.o >/dev/null 2>&1
grep -E "^\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size):" ./*gfx950*.s \
    | awk '{k=$1; v=$2; if (k==".name:") n=v; else r[n]=r[n] " " k v} END {for (n in r) print n r[n]}' \
    | grep -E "$pat" | sort
rm -rf "$tmp"
