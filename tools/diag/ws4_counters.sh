# wave-state counters of the K = 1024 streaming kernel (data-gradient form, tile 33) against the tiled 128 x 128 kernel (tile 20)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 33 20; do
  for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE; do
    rm -rf gpurun_out/pmc_c
    timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_c -- python3 tools/diag/ws4_fetch.py $t 256 > gpurun_out/pmc_c.log 2>&1 || { echo "tile $t $c: failed"; continue; }
    python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/pmc_c/*/*_counter_collection.csv")
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "gemm" in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "$c"]
    print("tile $t", "$c", vals[-1] if vals else "n/a", flush=True)
PY
  done
done
rm -rf gpurun_out/pmc_c
