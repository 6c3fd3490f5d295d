# (an experiment of round 6 -- profiles/r06/boundaries.txt (4); the switch / build variants it uses were taken out again)
# experiment: a control workgroup that fits on a CU beside a step workgroup.  Variants built with -DSP_K8_WIDE_WAVES=<2|3> -DSP_K8_CTL_THREADS=256 as libstarphase_hip_v<w>_256.so:
#   wide step kernel at 2 waves per SIMD (198 registers: 112 left per SIMD) or 3 (168 registers, 21 spilled: 176 left); control kernel 256 threads = one wave per SIMD at 127 registers
mkdir -p gpurun_out/r06q
for v in default v3_256 v2_256 default v3_256; do
  if [ $v = default ]; then unset SP_LIB_PATH; else export SP_LIB_PATH=$PWD/pb-starphase_amd/libstarphase_hip_$v.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06q/full.json > /dev/null 2> gpurun_out/r06q/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06q/full.json"))
for k in ("cyp2d6","hla"):
    cp=d["critical_path"][k]
    print("$v %s: value %.0f ms/step %.2f | chain_ms %.1f per_step %s boundary %s" % (k, d["value"], d["ms_per_step"], cp["chain_ms"], {a: round(v,1) for a,v in cp["per_step_us"].items()}, {a: round(v,1) for a,v in cp["boundary_us"].items()}))
PY
done
unset SP_LIB_PATH
for v in default v3_256; do
  if [ $v = default ]; then unset SP_LIB_PATH; else export SP_LIB_PATH=$PWD/pb-starphase_amd/libstarphase_hip_$v.so; fi
  python bench.py --steps 12 --warmup 3 --hla-lanes 1 --cyp-lanes 1 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06q/full.json > /dev/null 2> gpurun_out/r06q/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06q/full.json"))
cp=d["critical_path"]["cyp2d6"]
print("$v one lane each: value %.0f | chain_ms %.1f per_step %s boundary %s" % (d["value"], cp["chain_ms"], {a: round(v,1) for a,v in cp["per_step_us"].items()}, {a: round(v,1) for a,v in cp["boundary_us"].items()}))
PY
done
