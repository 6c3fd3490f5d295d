# -DSP_K8_SEARCH_TICKS build: the control step's search split into picking the node + consuming its tape / the decision block / loop iterations
# (control_parts_us_per_step then reads: reduce = 100 x iterations per step [in 10-ns ticks -> "us" = iterations], result = pick + consume, search = the whole search, tail = decision block)
SP_LIB_PATH=$PWD/build/variants/lib_st.so python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r04_st.json 2> gpurun_out/r04_st.err; echo "rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_st.json").read().strip().splitlines()[-1])
for k in ("cyp2d6","hla"):
    c=d["critical_path"][k]; print(k, c["dependent_steps"], {a:round(b,2) for a,b in c["control_parts_us_per_step"].items()})
PY
