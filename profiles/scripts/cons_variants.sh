#!/bin/bash
# Like k1_variants.sh for sp_consensus.hip; prints the end-to-end block of the bench.  bash profiles/scripts/cons_variants.sh "-DX" ""
set -e
rm -rf /tmp/consvar && cp -r "${GRAFT_REPO_ROOT:-/root/repo}" /tmp/consvar
cd /tmp/consvar
for defs in "$@"; do
    (cd pb-starphase_amd/csrc && rm -f sp_consensus.o && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $defs -c sp_consensus.hip -o sp_consensus.o 2>/dev/null && make -s 2>/dev/null)
    echo "== [$defs]"
    python bench.py --no-cpu-baseline --steps 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['end_to_end']; print(round(d['ms_per_step'],1), round(d['kernel_ms']['cons_steps'],1), d['diplotypes_equal_truth'])"
done
