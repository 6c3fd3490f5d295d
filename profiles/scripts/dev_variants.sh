#!/bin/bash
# Like k1_variants.sh for sp_device.hip (anchor / generic cells):  bash profiles/scripts/dev_variants.sh "-DSP_ANCHOR_ILP=8" ...
set -e
rm -rf /tmp/devvar && cp -r "${GRAFT_REPO_ROOT:-/root/repo}" /tmp/devvar
cd /tmp/devvar
for defs in "$@"; do
    (cd pb-starphase_amd/csrc && rm -f sp_device.o && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $defs -c sp_device.hip -o sp_device.o 2>/dev/null && make -s 2>/dev/null)
    echo "== $defs"
    python bench.py --no-end-to-end --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()}, d['concordance'])"
done
