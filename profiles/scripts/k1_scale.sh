# K1 at the shape of the reference's current HLA database: launch time and HBM traffic of the cells kernel (is the database still served from L2?)
export TMPDIR=/tmp
OUT=gpurun_out/k1_scale
rm -rf $OUT; mkdir -p $OUT
python3 profiles/scripts/k1_scale.py 40 2>&1 | tail -7
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 profiles/scripts/k1_scale.py 40 > $OUT/pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"gpurun_out/k1_scale/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k1_cells_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c: v.append(float(row["Counter_Value"]))
    big = [x for x in v if x * 10 >= max(v)]
    print(c, "KB per main launch", round(sum(big) / len(big)), "launches", len(big), "(bytes: x2 for FETCH_SIZE on gfx950)")
PY
rm -rf $OUT/pmc_*
