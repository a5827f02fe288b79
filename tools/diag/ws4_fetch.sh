cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 33 20; do
  for n in 256 128; do
    rm -rf gpurun_out/pmc_f
    timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -- python3 tools/diag/ws4_fetch.py $t $n > gpurun_out/pmc_f.log 2>&1 || exit 1
    python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/pmc_f/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and "FETCH" in r["Counter_Name"]]
for r in rows[-3:]:
    print("tile $t N=$n", r["Kernel_Name"][:60], "FETCH_SIZE(KB)", r["Counter_Value"], "-> x2 =", round(float(r["Counter_Value"]) * 2 / 1024, 1), "MB (A is 88 MB)")
PY
  done
done
rm -rf gpurun_out/pmc_f
