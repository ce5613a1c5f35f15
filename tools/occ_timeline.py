"""Kernel timeline of one occlusion-aware alignment.
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/occ_timeline.py run [occ] [method]
   python tools/occ_timeline.py show DIR        (the last alignment of the trace: kernel, level-free name, duration, gap before)"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import numpy as np
    from rgbd360_amd import synth
    from rgbd360_amd.register import RegisterPhotoICP
    occ = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    method = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    (rgbA, dA), (rgbB, dB), T = synth.add_occluder(synth.make_pair(2048, 1024, seed=5))
    reg = RegisterPhotoICP()
    reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB); reg.sync()
    for _ in range(6): reg.alignFrames360(np.eye(4), method, occ)
    print(reg.num_iterations)
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last alignment = everything behind the last gap longer than 50 us
    cut = 0
    for i in range(1, len(rows)):
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 50000: cut = i
    rows = rows[cut:]
    t0 = int(rows[0]["Start_Timestamp"]); prev = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void r360::", "").replace("r360::", "")
        print("%8.2f  %-34s %7.2f us  gap %5.2f  grid %s" % ((s - t0) / 1e3, name[:34], (e - s) / 1e3, (s - prev) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
        prev = e
    print("span %.1f us, %d launches" % ((prev - t0) / 1e3, len(rows)))
