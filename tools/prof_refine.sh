timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "refine or plane" > gpurun_out/t_ref.log 2>&1; tail -2 gpurun_out/t_ref.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nz in 0 0.01; do rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rf -- python3 tools/prof_refine.py 2048 $nz > gpurun_out/rf.log 2>&1; python - <<PY
import csv,glob
f=glob.glob("gpurun_out/rf/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)/3e3
print("noise $nz: total %.0f us per frame" % tot)
for r in rows[:4]: print("   %-40s calls %4s avg %8.1f us total/frame %7.1f" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/3e3))
PY
rm -rf gpurun_out/rf; done
python tools/refine_perf.py 2048 2>&1 | tail -2 | cut -c1-120
