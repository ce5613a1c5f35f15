"""VALU instruction histogram of the steady-state loop of a kernel in a device-ISA dump (the innermost loop holding two source-record
loads: the software-pipelined pixel loop of eval_span):
   hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o api.s rgbd360_amd/csrc/rgbd360_api.hip
   python tools/isa_loop_stats.py api.s [mangled-name-prefix] [--dump]"""
import re
import sys
from collections import Counter
path = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "_ZN4r3606k_evalILi0ELb1EEE"
L = open(path).read().split("\n")
st = [i for i, l in enumerate(L) if l.startswith(name) and l.rstrip().split(";")[0].strip().endswith(":")][0]
end = [i for i in range(st, len(L)) if L[i].startswith(".Lfunc_end")][0]
body = L[st:end]
loops = []
for hdr in [i for i, l in enumerate(body) if "Loop Header" in l]:
    lab = body[hdr].split(":")[0]
    backs = [i for i in range(hdr, len(body)) if re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", body[i])]
    if backs:
        loops.append(body[hdr:backs[-1] + 1])
cands = [lp for lp in loops if sum("buffer_load_dwordx4" in l or "buffer_load_dwordx2" in l for l in lp) == 2] or loops
loop = min(cands, key=len)
v = [l.split()[0] for l in loop if l.strip().startswith("v_")]
print(path, "loop lines", len(loop), "VALU in loop", len(v), "| LDS", sum(l.strip().startswith("ds_") for l in loop), "SALU", sum(l.strip().startswith("s_") for l in loop))
print("waitcnt:", [l.strip().split(";")[0].strip() for l in loop if "s_waitcnt" in l])
print("loads:", [l.split()[0] for l in loop if "_load_" in l or l.strip().startswith("ds_read")])
print(sorted(Counter(v).items(), key=lambda kv: -kv[1]))
for l in L[end:end + 120]:
    if any(k in l for k in ("NumVgprs", "ScratchSize", "Occupancy", "NumSgprs")):
        print(l.strip())
if "--dump" in sys.argv:
    print("\n".join(l.split(";")[0].rstrip() for l in loop if l.split(";")[0].strip()))
