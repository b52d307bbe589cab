#!/bin/bash
# tools/isa_summary.sh <file.hip> [extra hipcc flags]: device-only assembly of one kernel file -> /tmp/isa/<file>.s and a per-kernel
# summary (VGPRs, scratch, LDS bytes, occupancy, s_barrier sites) — what DESIGN.md quotes register counts from.
# REUSE=1 skips the compilation and summarises the .s of the last run.
set -e
f=$1; shift
mkdir -p /tmp/isa
out=/tmp/isa/$(basename "$f" .hip).s
cd "$(dirname "$0")/../image_matching_amd/csrc"
[ -n "$REUSE" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only "$@" -o "$out" "$(basename "$f")" 2>/dev/null
awk '
/^_Z[A-Za-z0-9_]*:/ {name=$1; sub(/:.*/, "", name)}
/s_barrier/ {bar[name]++}
/; NumVgprs:/ {v[name]=$3}
/; ScratchSize:/ {sc[name]=$3}
/; LDSByteSize:/ {l[name]=$3}
/; Occupancy:/ {o[name]=$3; order[++n]=name}
END {for (i=1;i<=n;i++){k=order[i]; printf "vgpr %-4s scratch %-5s lds %-6s occ %-2s barriers %-2s %s\n", v[k], sc[k], l[k], o[k], bar[k]+0, k}}' "$out" | c++filt | sed 's/void //' | cut -c1-200
