import os, sys, subprocess
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.environ.get("GRAFT_REPO_ROOT", ".")
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from rgbd360_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
pair = synth.make_pair(2048, 1024, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
out = []
for m in (0, 2):
    fb = [reg.forced_iters_batch(16, pair[0], pair[1], 0, np.eye(4), m, 4) for _ in range(4)]
    out.append("m%%d %%s" %% (m, " ".join("%%.1f" %% f["pass_avg_us"] for f in fb)))
print(" | ".join(out))
''' % ROOT
for name in sys.argv[1:]:
    lib, _, envs = name.partition("@")
    env = dict(os.environ)
    for kv in envs.split("@"):
        if "=" in kv: env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
    path = os.path.join(ROOT, "rgbd360_amd/lib", "librgbd360_hip.so" if lib == "default" else "librgbd360_hip_%s.so" % lib)
    r = subprocess.run([sys.executable, "-c", CHILD, path], capture_output=True, text=True, env=env)
    print("%-28s| %s" % (name, r.stdout.strip() or r.stderr.strip()[-400:]), flush=True)
