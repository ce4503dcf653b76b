# Development helper, run ON the GPU box (gpurun -- 'bash tools/rns_gm_sweep.sh'): step time, GEMM launch time and L2 fetch
# bytes (FETCH_SIZE) of the residue GEMM for several tile-group heights (MPCX_RNS_GM).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for GM in 1 2 3 4 6 11; do
  echo "GM=$GM"
  MPCX_RNS_GM=$GM timeout -k 10 120 python3 -u bench.py --no-bitwise --no-cpu --steps 6 --warmup 2 > gpurun_out/probe.log 2>&1 || exit 1
  python3 - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/probe.log") if l.startswith("{")][-1])
print(" ms/step %.3f  gemm avg launch ms %.4f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
P
  rm -rf gpurun_out/pmc_tmp
  MPCX_RNS_GM=$GM timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_tmp -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-bitwise > gpurun_out/pmc_tmp.log 2>&1 || exit 1
  python3 - <<'P'
import csv,glob
for f in glob.glob("gpurun_out/pmc_tmp/*counter_collection.csv"):
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "rns_gemm_wide" in r["Kernel_Name"]]
    print(" FETCH GB per launch %.2f (n=%d)" % (sum(v)/len(v)*1024*2/1e9, len(v)))
P
done
