"""VALU instruction histogram of the steady-state loop of a kernel in a device-ISA dump:
   hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o api.s rgbd360_amd/csrc/rgbd360_api.hip
   python tools/isa_loop_stats.py api.s [mangled-name-prefix]"""
import sys
from collections import Counter
path = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "_ZN4r3606k_evalILi0ELb1EEE"
L = open(path).read().split("\n")
st = [i for i, l in enumerate(L) if l.startswith(name) and l.rstrip().split(";")[0].strip().endswith(":")][0]
end = [i for i in range(st, len(L)) if L[i].startswith(".Lfunc_end")][0]
body = L[st:end]
hdr = [i for i, l in enumerate(body) if "Loop Header" in l][0]
lab = body[hdr].split(":")[0]
back = [i for i in range(hdr, len(body)) if lab in body[i] and ("s_branch" in body[i] or "s_cbranch" in body[i])][-1]
loop = body[hdr:back + 1]
v = [l.split()[0] for l in loop if l.strip().startswith("v_")]
print(path, "loop lines", len(loop), "VALU in loop", len(v))
print("waitcnt:", [l.strip() for l in loop if "s_waitcnt" in l])
print("loads:", [l.split()[0] for l in loop if "_load_" in l])
print(sorted(Counter(v).items(), key=lambda kv: -kv[1]))
for l in L[end:end + 120]:
    if any(k in l for k in ("NumVgprs", "ScratchSize", "Occupancy", "NumSgprs")):
        print(l.strip())
