for i in 1 2 3 4 5 6; do
SP_BENCH_CYP_PERSISTENT=1 python bench.py --no-cpu-baseline > gpurun_out/r04_p_$i.json 2> gpurun_out/r04_p_$i.err; echo "run $i rc $?"; tail -3 gpurun_out/r04_p_$i.err | cut -c1-400
done
