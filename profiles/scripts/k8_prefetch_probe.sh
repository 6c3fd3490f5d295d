# probe build of the consensus (counters only, results unchanged), the six scenarios, then the shipped build again
set -e
cd pb-starphase_amd/csrc
cp ../libstarphase_hip.so /tmp/libstarphase_hip.keep.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DSP_K8_PF_PROBE -c sp_consensus.hip -o /tmp/sp_consensus_probe.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libstarphase_hip.so $(ls *.o | grep -v sp_consensus.o) /tmp/sp_consensus_probe.o -lz -ldl
cd ../..
python profiles/scripts/k8_prefetch_probe.py 2000 2>&1 | tee gpurun_out/k8_prefetch_probe.txt
cp /tmp/libstarphase_hip.keep.so pb-starphase_amd/libstarphase_hip.so
