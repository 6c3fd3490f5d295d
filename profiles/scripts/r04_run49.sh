# where the persistent leg's time goes: host wall per stage of the CYP2D6 call in both modes
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r04_pw_$i.json 2> gpurun_out/r04_pw_$i.err; echo "run $i rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_pw_$i.json").read().strip().splitlines()[-1])
print("launch pairs", round(d["value"]), round(d["ms_per_step"],2), d["host_wall_ms"]["cyp2d6"])
L=(d["legs"].get("headline_with_persistent_consensus") or d["legs"].get("headline_with_launch_pairs"))
print("persistent  ", round(L["value"]), round(L["ms_per_step"],2), L["host_wall_ms_cyp2d6"], L["host_wall_ms_k8"], round(L["cyp2d6_cons_steps_ms"],2))
PY
done
