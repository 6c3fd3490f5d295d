set -e
rm -rf /tmp/seedvar && cp -r "${GRAFT_REPO_ROOT:-/root/repo}" /tmp/seedvar
cd /tmp/seedvar
for defs in "$@"; do
    (cd pb-starphase_amd/csrc && rm -f sp_hla_seed.o && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $defs -c sp_hla_seed.hip -o sp_hla_seed.o 2>/dev/null && make -s 2>/dev/null)
    echo "== [$defs]"
    python bench.py --no-cpu-baseline --steps 6 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['critical_path']['cyp2d6']; k=d['legs']['k1_modes']['seeded_best_n_5']
print('headline', round(d['value']), round(d['ms_per_step'],1), {a: round(b,1) for a,b in c['per_step_us'].items()}, 'k1', round(k['ms_per_call'],2), k['kernel_ms'], 'hla_resident', round(d['legs']['hla_resident']['ms_per_step'],1), d['concordance']['hla_diplotypes_equal_truth'])
"
done
