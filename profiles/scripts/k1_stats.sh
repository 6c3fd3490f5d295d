#!/bin/bash
# Builds a profiling copy of the library in /tmp (only sp_hla.hip gets -DSP_K1_STATS) and prints the k1 event counters.
set -e
rm -rf /tmp/k1stats && cp -r "${GRAFT_REPO_ROOT:-/root/repo}" /tmp/k1stats
cd /tmp/k1stats/pb-starphase_amd/csrc
rm -f sp_hla.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DSP_K1_STATS -c sp_hla.hip -o sp_hla.o
make -s
cd /tmp/k1stats && python profiles/scripts/k1_stats.py "$@"
