#!/bin/bash
# Kernel-variant experiments on sp_consensus.hip: rebuilds the library in a scratch copy with the given defines and prints the headline, the chain and the CYP2D6 scenario table of
# the bench.   bash profiles/scripts/cons_variants.sh "-DX" ""
set -e; trap "tail -5 /tmp/consvar_err.txt" ERR
rm -rf /tmp/consvar && cp -r "${GRAFT_REPO_ROOT:-/root/repo}" /tmp/consvar
cd /tmp/consvar
for defs in "$@"; do
    (cd pb-starphase_amd/csrc && rm -f sp_consensus.o && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $defs -c sp_consensus.hip -o sp_consensus.o 2>/dev/null && make -s 2>/dev/null)
    echo "== [$defs]"
    python bench.py --no-cpu-baseline 2>/tmp/consvar_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['critical_path']['cyp2d6']
print('headline', round(d['value']), round(d['ms_per_step'],1), c['mode'][:12], 'chain', round(c['chain_ms'],1), {k: round(v,1) for k,v in c['per_step_us'].items()}, {k: round(v,1) for k,v in c['control_parts_us_per_step'].items()})
h=d['critical_path']['hla']
print('hla chain', round(h['chain_ms'],1), {k: round(v,1) for k,v in h['per_step_us'].items()}, 'lanes', [round(x['work'],1) for x in d['host_wall_ms']['lanes_hla_cyp2d6']])
print('other mode', [round(v['value']) for k, v in d['legs'].items() if k.startswith('headline_with') and 'value' in v], 'hla_resident', round(d['legs']['hla_resident']['ms_per_step'],1), 'cyp', {k: round(v['ms'],1) for k,v in d['legs']['cyp2d6']['scenarios'].items()}, 'lanes', {k: round(v['value']) for k,v in d['legs']['cyp2d6_lanes'].items() if isinstance(v, dict)}, 'cohort', round(d['legs']['cohort']['samples_per_s'],1), {k: round(v['seconds'],3) for k,v in d['legs']['cohort']['by_share_size'].items()})
"
done; true
