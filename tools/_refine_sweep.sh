cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "64 512" "32 4096"; do
  set -- $cfg
  for nz in 0 0.01; do
    RGBD360_REFINE_QUIET=$1 RGBD360_REFINE_POLLS=$2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rf -- python3 tools/_refine_prof.py 2048 $nz > gpurun_out/rf.log 2>&1
    python - <<PY
import csv,glob
f=glob.glob("gpurun_out/rf/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
rt=[(r["Kernel_Name"].split("(")[0][-22:], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows if "refine_tile" in r["Kernel_Name"]]
n=len(rt)//3
print("QUIET $1 POLLS $2 noise $nz: last frame launches:", " ".join("%s %.0f" % (a[-3:], b) for a,b in rt[-n:]))
PY
    rm -rf gpurun_out/rf
  done
done
