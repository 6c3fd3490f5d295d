# (an experiment of round 6 -- profiles/r06/lanes_matrix.txt (c): the variants were built with -DSP_K8_WAVES=<N> as libstarphase_hip_w<N>.so and are not kept)
# experiment: waves per workgroup of the K8 step kernel (compile-time SP_K8_WAVES; variants built as libstarphase_hip_w<N>.so)
mkdir -p gpurun_out/r06h
for w in 8 4 2 1; do
  if [ $w = 8 ]; then unset SP_LIB_PATH; else export SP_LIB_PATH=$PWD/pb-starphase_amd/libstarphase_hip_w$w.so; fi
  echo "== waves per workgroup $w"
  timeout 900 python -m pytest tests/test_gpu_consensus.py -x -q -m gpu 2>&1 | tail -2
  for rep in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06h/full_w$w.json > /dev/null 2> gpurun_out/r06h/err_w$w.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06h/full_w$w.json"))
cp=d["critical_path"]["cyp2d6"]
print("W=$w: value %.0f ms/step %.2f | cyp chain_ms %.1f per_step %s | hla k8 %.1f k1 %.1f | lanes work %s" % (d["value"], d["ms_per_step"], cp["chain_ms"], {k: round(v,1) for k,v in cp["per_step_us"].items()},
   d["host_wall_ms"]["hla"]["k8_loop"], d["host_wall_ms"]["hla"]["k1_total"], [round(x["work"],1) for x in d["host_wall_ms"]["lanes_hla_cyp2d6"]]))
PY
  done
  timeout 600 python bench.py --steps 12 --warmup 3 --cyp-lanes 1 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06h/full1_w$w.json > /dev/null 2> gpurun_out/r06h/err1_w$w.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06h/full1_w$w.json"))
cp=d["critical_path"]["cyp2d6"]
print("W=$w one lane: value %.0f ms/step %.2f | cyp chain_ms %.1f per_step %s" % (d["value"], d["ms_per_step"], cp["chain_ms"], {k: round(v,1) for k,v in cp["per_step_us"].items()}))
PY
done
