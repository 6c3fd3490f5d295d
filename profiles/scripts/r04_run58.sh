# the headline with the CYP2D6 context's consensus as persistent kernels (bench default from here on), the launch-pair headline as a leg: three whole lines; the bench test
timeout 1200 python -m pytest tests/test_gpu_bench.py tests/test_gpu_consensus.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2 3; do
python bench.py --no-cpu-baseline > gpurun_out/r04_hp_$i.json 2> gpurun_out/r04_hp_$i.err; echo "run $i rc $?"; tail -2 gpurun_out/r04_hp_$i.err | cut -c1-300
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_hp_$i.json").read().strip().splitlines()[-1])
L=d["legs"]; o=L.get("headline_with_launch_pairs") or L.get("headline_with_persistent_consensus")
print(round(d["value"]), round(d["ms_per_step"],2), d["config"]["cyp2d6_consensus"][:40], "| other mode", round(o.get("value",0)), "| cons_steps", round(d["kernel_ms"]["cyp2d6"]["cons_steps"],2), {k:round(v,2) for k,v in d["host_wall_ms"]["cyp2d6"].items() if not isinstance(v, dict)}, d["host_wall_ms"]["lanes_hla_cyp2d6"], round(L["cohort"]["samples_per_s"],1), round(L["samples_in_flight"]["value"]), round(L["cyp2d6"]["value"]), round(L["hla_resident"]["value"]), d["concordance"])
c=d["critical_path"]["cyp2d6"]; print(c["mode"][:30], c["dependent_steps"], round(c["chain_ms"],2), {a:round(b,1) for a,b in c["per_step_us"].items()})
PY
done
