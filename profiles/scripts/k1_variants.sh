#!/bin/bash
# Builds sp_hla.hip with each given set of -D flags in a /tmp copy of the tree and runs the bench step on it.
#   bash profiles/scripts/k1_variants.sh "-DK1_UNIT=16" "-DK1_UNIT=64 -DK1_MIN_WAVES=8"
set -e
rm -rf /tmp/k1var && cp -r "${GRAFT_REPO_ROOT:-/root/repo}" /tmp/k1var
cd /tmp/k1var
for defs in "$@"; do
    (cd pb-starphase_amd/csrc && rm -f sp_hla.o && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $defs -c sp_hla.hip -o sp_hla.o 2>/dev/null && make -s 2>/dev/null)
    echo "== $defs"
    python bench.py --no-end-to-end --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items() if k.startswith('k1')}, d['concordance'])"
done
