#!/usr/bin/env python3
"""bench.py -- Gauss-Newton iterations/sec of the dense spherical alignment hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches this file
with torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

Step      = one level-0 Gauss-Newton iteration of RegisterPhotoICP::alignFrames360 on one 2048x1024 synthetic
            spherical pair (BASELINE.json configs[1]: photometric-only), in the forced schedule of BASELINE.md §2
            (accept rule evaluated, step applied regardless): one fused warp+residual+Jacobian pass over every
            source pixel + the 6x6 solve / pose update.  Since round 3 that is ONE launch per iteration (k_eval_fs: the solve of
            the previous pass is the prologue of the next pass's launch; the two-launch schedule {k_eval, k_solve} is a test hook, rgbd360_debug_set_schedule).
            Frames are resident in HBM before the timed region.
value     = N * K / t, t = the MEDIAN over `--repeats` timed regions of exactly K steps each (every region bracketed by a
            barrier + device synchronisation on both sides, the maximum over the ranks taken per region): with the driver's
            K = 20 a single region is 0.4 ms of wall time, far too short to be one sample.
N > 1     = weak scaling of `value` (every rank iterates on its own pair, the solved poses are all-gathered over RCCL inside
            every timed region), AND the `sequence` block: BASELINE.json configs[3] itself -- 256 consecutive pairs of an
            odometry sequence in contiguous shards over the ranks, every rank aligning its shard with the library's sequence
            entry, one all-gather of poses / status / iteration counts -- as alignments/s of the whole job.
roofline  = algorithmic bytes of the dominant kernel (28 B/px photo, 40 B/px photo+depth, LUT variant)
            / its average launch duration measured with HIP events on the library's own stream.  The dominant kernel of `value` is
            k_eval_fs (solve prologue + warp/residual/Jacobian pass: a whole Gauss-Newton iteration per launch); `roofline.pass_only`
            is the per-pixel pass without the prologue (k_eval, the same eval_span body).  At 2048x1024 the kernel's
            working set (84 MB) stays in the 256 MiB Infinity Cache between back-to-back launches: `roofline.resident` says so,
            `roofline_hbm_rotating` repeats the measurement rotating over 5 copies of the pair (420 MB: every launch HBM-fed),
            and `roofline_4096x2048` (335 MB photo+depth) is the single-pair HBM figure.
iteration = the same bytes over the whole step as the host sees it (launch + gaps + the K-step call's fixed cost / K).
sustained = after the timed regions, one >= 2 s burst of the same forced schedule (not part of `value`): its rate is printed next to
            `value` as a cross-check, and it is long enough for a utilisation sampler to catch the GPU busy.
cpu_baseline = the CPU oracle (restatement of the reference algorithm, OpenMP) on the same workload, bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Next to PyTorch's and RCCL's own
# streams that leaves the library's two sequence engines sharing queues (the `sequence` block: 8.5 k alignments/s instead of the
# 10.9 k of a process that only hosts the library).  Read at runtime initialisation: it has to be in the environment before the
# first HIP call.  This process never keeps more than four streams busy at once, which is what makes 8 safe here (INTEGRATION.md §3:
# with more queues than the four the device runs side by side, a fifth busy stream costs a factor of three).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # measured float4 copy (same guide)
LLC_BYTES = 256 * 1024 * 1024  # Infinity Cache
BYTES_PER_PX = {0: 28, 1: 28, 2: 40}   # LUT variant: 16 B source record + 12 B (24 B) gathered target records
BYTES_PER_PX_RECOMPUTE = {0: 20, 1: 20, 2: 32}   # recompute variant: 8 B of depth + intensity per source pixel, the point re-formed in the pass
WORKING_SET_PER_PX = {0: 28, 1: 28, 2: 40}
# which levels run which form (the library's rules, csrc/rgbd360_api.hip recompute_min_px / sequence_engine.h): the single-pair pass from
# 4 Mpx up (levels that cannot stay in the Infinity Cache), the lock-step engine (always HBM-fed) from 256 Kpx up
RECOMPUTE_MIN_PX = int(os.environ.get("RGBD360_RECOMPUTE_MIN_PX", 4 * 1024 * 1024))
SEQ_RECOMPUTE_MIN_PX = int(os.environ.get("RGBD360_SEQ_RECOMPUTE_MIN_PX", 256 * 1024))


def pass_bytes_per_px(method, n_px, engine=False):
    """Algorithmic bytes per source pixel of one fused pass in the form the library runs at this level size."""
    rc = n_px >= (SEQ_RECOMPUTE_MIN_PX if engine else RECOMPUTE_MIN_PX)
    return (BYTES_PER_PX_RECOMPUTE if rc else BYTES_PER_PX)[method]
METHOD_NAMES = {0: "PHOTO_CONSISTENCY", 1: "DEPTH_CONSISTENCY", 2: "PHOTO_DEPTH"}
SEQ_UNIQUE_FRAMES = 9          # frames rendered per rank for the sequence block (walked back and forth)
SEQ_INFLIGHT = 32              # pairs in flight per GPU: slots of the lock-step sequence engine (2 engines x 16)


def live_traffic(W, H, method):
    """roofline.traffic of THIS run: fabric-side bytes per launch of the dominant kernel from two rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE, each in its own child process with nothing but --pmc, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) over
    tools/prof_eval.py, which launches the same kernel on the same synthetic pair; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB, the guide's
    gfx950 correction for 16 B / lane streams.  Returns (bytes per launch, source note), or (None, None) when rocprofv3 is unavailable,
    the passes fail, or bench.py itself runs under a profiler."""
    import csv, glob, shutil, subprocess, tempfile
    if "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "not collected live: bench.py itself runs under a profiler"
    if shutil.which("rocprofv3") is None:
        return None, "not collected live: rocprofv3 is not on PATH"
    # the full instantiation name of the level's source form (", 0>" = 16-byte records), so that e.g. the recompute instantiation "<0, 1>" is not averaged in
    want = "k_eval_fs<%d, 0>" % method
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rgbd360_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                            os.path.join(ROOT, "tools", "prof_eval.py"), str(W), str(H), "6"],
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=90, check=True)
            v = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == counter and want in r.get("Kernel_Name", ""):
                        v.append(float(r["Counter_Value"]))
            if not v:
                return None, "not collected live: the %s pass produced no row for %s" % (counter, want)
            vals[counter] = sum(v) / len(v)
        except Exception as e:      # (why, into the line: the figure then comes from the tracked profiles/traffic_latest.json)
            return None, "not collected live: the %s pass failed (%s: %s)" % (counter, type(e).__name__, str(e)[:160])
        finally:
            shutil.rmtree(d, ignore_errors=True)
    nbytes = int(round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0))
    note = ("collected in THIS run: two rocprofv3 --pmc passes (FETCH_SIZE %.1f KB, WRITE_SIZE %.1f KB per launch of %s, each counter in its own child "
            "process over tools/prof_eval.py on the same synthetic pair); bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of "
            "MI355X_MICROARCH.md for 16 B / lane streams (the 12-byte gathers are an uncalibrated width: +-10 %%)" % (vals["FETCH_SIZE"], vals["WRITE_SIZE"], want))
    return nbytes, note


def avg_kernel_us(fn, batches=5):
    """Median of `batches` averages (each over back-to-back launches between two HIP events on the library's stream), so
    that one disturbed batch (another process touching the device) cannot skew it; all batches are reported."""
    vals = sorted(fn() for _ in range(batches))
    return vals[len(vals) // 2], vals


def roofline_entry(us, batches, n_px, method, kernel=None, engine=False, **extra):
    bpp = pass_bytes_per_px(method, n_px, engine)
    alg = bpp * n_px
    ach = alg / (us * 1e-6) / 1e9
    slow = max(batches)
    d = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "kernel": kernel or "k_eval<%d,true>" % method, "kernel_avg_us": us, "kernel_avg_us_batches": batches,
         "frac_slowest_batch": alg / (slow * 1e-6) / 1e9 / HBM_PEAK_GBS,
         "algorithmic_bytes_per_launch": alg, "bytes_per_pixel": bpp,
         "source_form": ("recompute: 8 B {depth, intensity} per source pixel + angle tables, point re-formed per pixel" if bpp != BYTES_PER_PX[method]
                         else "records: 16 B {x, y, z, I} per source pixel (LUT_xyz_sphere precomputed)"),
         "frac_of_measured_copy_peak": ach / HBM_COPY_GBS}
    d.update(extra)
    return d


# algorithmic bytes per pixel of the Frame360 stages (one frame, every pixel counted)
F360_STAGE_BYTES = {"a13_sphere_cloud": 2 + 12,        # u16 range in, xyz out
                    "a14_normal_map": 12 + 12,         # xyz in, normals out
                    "a15_plane_stage": 12 + 4}         # xyz + label per pixel of the moment pass


def frame360_roofline(torch, Frame360Stages, RegisterPhotoICP, device, depth_u16, reps=12):
    """HIP-event times of the three Frame360 stages of ONE rgbd360_frame_planes_dev call (depth already in HBM, events on the
    library's stream at the stage boundaries: rgbd360_frame_planes_stage_timing), median over `reps` calls, against SURVEY.md 8d's bytes."""
    H, W = depth_u16.shape
    st = Frame360Stages(RegisterPhotoICP(device=device))
    d_dev = torch.from_numpy(depth_u16.astype(np.int16)).to("cuda:%d" % device)       # (same bits as the uint16 image)
    kw = dict(depth_type=0, convention=2, angular_threshold=0.03, min_inliers=40, max_curvature=0.0013, max_planes=4096)      # tools/prof_frame360.py's call
    out = st.frame_planes_dev(d_dev.data_ptr(), H, W, **kw)                             # warm: allocations, tables
    st.stage_timing(True)
    rows = []
    for _ in range(reps):
        out = st.frame_planes_dev(d_dev.data_ptr(), H, W, **kw)
        rows.append(st.stage_times())
    st.stage_timing(False)
    med = [sorted(r[k] for r in rows)[len(rows) // 2] for k in range(3)]
    n = W * H
    stages = {}
    for (name, bpp), us in zip(F360_STAGE_BYTES.items(), med):
        ach = bpp * n / (us * 1e-6) / 1e9
        stages[name] = {"us": us, "bytes_per_pixel": bpp, "algorithmic_bytes": bpp * n, "achieved": ach, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    tot_b = sum(F360_STAGE_BYTES.values()) * n
    tot_us = sum(med)
    return {"width": W, "height": H, "planes": len(out["planes"]), "stages": stages,
            "chain": {"us": tot_us, "algorithmic_bytes": tot_b, "achieved": tot_b / (tot_us * 1e-6) / 1e9, "unit": "GB/s",
                      "frac": tot_b / (tot_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS},
            "kernels": {"a13_sphere_cloud": "k_f360_edge_bits<true> (forms the cloud, writes it and the depth-change mask)",
                        "a14_normal_map": "k_f360_distmap, k_f360_normals_sweep<8>, k_f360_normals_tiled",
                        "a15_plane_stage": "k_f360_link_flags ... k_f360_hull_pack (labels, counts, slots, moments, hull extremes; refinement off)"}}


def pingpong(n_pairs, n_unique):
    """Frame indices 0,1,..,n_unique-1,n_unique-2,..,0,1,.. : n_pairs + 1 entries, consecutive entries always neighbours."""
    idx, k, step = [], 0, 1
    for _ in range(n_pairs + 1):
        idx.append(k)
        if k + step < 0 or k + step >= n_unique:
            step = -step
        k += step
    return idx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=25, help="timed regions of exactly --steps steps; the median is reported")
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--method", type=int, default=0, help="0 photo (configs[1]), 2 photo+depth (configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-4k", action="store_true", help="skip the extra 4096x2048 (configs[4]) kernel measurement")
    ap.add_argument("--no-sequence", action="store_true", help="skip the configs[3] sequence block")
    ap.add_argument("--no-rotating", action="store_true", help="skip the HBM-fed (rotating) kernel measurement, e.g. under rocprofv3 "
                    "--stats, whose per-kernel average would otherwise mix both regimes")
    ap.add_argument("--no-native-multi", action="store_true", help="skip the single-process multi-GPU entry (child process)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not collect roofline.traffic with rocprofv3 PMC passes in child "
                    "processes (then the tracked profiles/traffic_latest.json is quoted); implied when bench.py itself runs under rocprofv3")
    ap.add_argument("--seq-pairs", type=int, default=256, help="pairs of the configs[3] sequence (whole job)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = max(args.gpus, 1)
    if world != n_gpus and world > 1:
        n_gpus = world

    import torch
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback in the product path)"
    # Development aid for 1-GPU boxes: BENCH_SHARE_DEVICE=1 puts every rank on device 0 and uses gloo for the exchange
    # (RCCL refuses two ranks on one GPU).  The driver never sets it: one rank per GPU, backend nccl (= RCCL).
    share = os.environ.get("BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    xdev = "cpu" if share else "cuda"       # where the exchanged tensors live

    from rgbd360_amd import synth
    from rgbd360_amd.batch import gather_poses, shard_range
    from rgbd360_amd.register import RegisterPhotoICP

    W, H, method = args.width, args.height, args.method
    n_px = W * H
    t_gen = time.time()
    (rgbA, dA), (rgbB, dB), T_gt = synth.make_pair(W, H, seed=1234 + rank)
    t_gen = time.time() - t_gen

    reg = RegisterPhotoICP(device=local_rank)
    reg.setNumPyr(4)
    reg.setTargetFrame(rgbA, dA)
    reg.setSourceFrame(rgbB, dB)
    reg.sync()

    # full coarse-to-fine alignment (natural accept/reject schedule): pose + per-level iteration counts
    t_align = []
    for _ in range(5):
        t0 = time.perf_counter()
        rc = reg.alignFrames360(np.eye(4), method)
        t_align.append(time.perf_counter() - t0)
    pose_gpu = reg.getOptimalPose()
    iters_nat = list(reg.num_iterations)
    start_pose = np.eye(4)     # forced schedule starts from the identity guess at level 0

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup, then `repeats` timed regions of EXACTLY K steps ---------------------------------------------
    reg.forced_iters(0, start_pose, method, max(args.warmup, 1))
    if dist is not None:
        # warm the exchange too: the first RCCL all-gather builds channels / loads kernels, which must not be timed
        w_in = torch.zeros(16, dtype=torch.float32, device=xdev)
        w_out = torch.empty(world * 16, dtype=torch.float32, device=xdev)
        dist.all_gather_into_tensor(w_out, w_in)
        w_t = torch.zeros(1, dtype=torch.float64, device=xdev)
        dist.all_reduce(w_t, op=dist.ReduceOp.MAX)
    repeats = max(args.repeats, 1)
    elapsed_all = []
    poses = None
    # the K steps as ONE C call with its arguments converted beforehand (the Python / numpy conversions of a call cost as much as
    # a step and are not the path); no HIP events inside: the wall clock around it is the measurement
    run_k_steps = reg.forced_iters_call(0, start_pose, method, args.steps)
    pose_host = torch.from_numpy(run_k_steps.pose_cm)        # shares the buffer the call writes
    gathered = torch.empty(world * 16, dtype=torch.float32, device=xdev) if dist is not None else None
    for _ in range(repeats):
        sync_all()
        t0 = time.perf_counter()
        rc_forced = run_k_steps()
        if dist is not None:
            dist.all_gather_into_tensor(gathered, pose_host.to(xdev))      # RCCL over xGMI: the path's one exchange step
            poses = gathered
        sync_all()
        elapsed_all.append(time.perf_counter() - t0)
    from rgbd360_amd.register import pose_from_cm
    out = {"pose": pose_from_cm(run_k_steps.pose_cm), "status": rc_forced}
    if dist is not None:
        tmax = torch.tensor(elapsed_all, dtype=torch.float64, device=xdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)          # per region: the slowest rank
        elapsed_all = [float(x) for x in tmax.cpu()]
        assert poses is not None and bool(torch.isfinite(poses).all())
        rccl_ranks_seen = int(poses.numel() // 16)       # rows of the gathered tensor: one 4x4 pose per rank that took part
    else:
        rccl_ranks_seen = 1
    srt = sorted(elapsed_all)
    elapsed = srt[len(srt) // 2]

    # ---- sustained burst (NOT part of `value`): the same forced schedule for >= 2 s in few large C calls, so that a utilisation
    #      sampler sees the GPU busy and the rate cross-checks `value` (which is the median of 25 regions of 0.3 ms) ----
    burst_steps = max(2000, int(2.2 / max(elapsed / args.steps, 1e-7) / 8))
    run_burst = reg.forced_iters_call(0, start_pose, method, burst_steps)
    sync_all()
    t0 = time.perf_counter()
    n_burst = 0
    while time.perf_counter() - t0 < 2.0 or n_burst < 2:
        run_burst()
        n_burst += 1
    torch.cuda.synchronize()
    burst_s = time.perf_counter() - t0
    sustained = {"gn_iterations_per_s": n_gpus * n_burst * burst_steps / burst_s, "seconds": burst_s, "steps": n_burst * burst_steps,
                 "calls": n_burst, "note": "same forced schedule as `value`, one rank's own clock (no barrier inside), not part of `value`"}

    # "must print the salient fractions" -- the share of the level-0 source pixels that pass every gate of the pass
    # (valid source point, warped inside the image, target gradient above the saliency threshold, RPI.h:2690 / 2706) and so
    # contribute a photometric / a depth residual, counted by the device's own pass at the solved pose.  Bytes are counted for
    # every pixel regardless (the record streams are contiguous).
    e_pd = reg.eval(0, pose_gpu, 2)
    e_ph = reg.eval(0, pose_gpu, 0) if method == 0 else e_pd
    salient_fraction = {"photo": float(e_ph["n_split"][0]) / n_px, "depth": float(e_pd["n_split"][1]) / n_px,
                        "photo_in_photo_depth_mode": float(e_pd["n_split"][0]) / n_px, "visible": float(e_pd["n_visible"]) / n_px,
                        "residuals_per_pixel_photo_depth": float(e_pd["n_valid"]) / n_px,
                        "note": "level 0, at the solved pose; photo / depth = pixels contributing a photometric / depth residual over all %d source pixels" % n_px}

    value = n_gpus * args.steps / elapsed
    result = {
        "metric": "Gauss-Newton iters/sec on 2048x1024 spherical pair; SE(3) err vs CPU ref",
        "value": value,
        "unit": "GN iterations/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "env": {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "RGBD360_HOST_SPIN_US", "RGBD360_RECOMPUTE_MIN_PX", "RGBD360_SEQ_RECOMPUTE_MIN_PX",
                                                "RGBD360_SEQ_ENGINES", "RGBD360_FORCE_RCCL")},      # the six runtime knobs of a product build (csrc/knobs.h)
        "sustained": sustained,
        "rccl_ranks_seen": rccl_ranks_seen,
        "salient_fraction": salient_fraction,
        "config": {
            "workload": ("configs[1]: single %dx%d synthetic sphere pair per GPU, %s RegisterPhotoICP, level-0 forced "
                         "Gauss-Newton iterations (one launch each: solve of the previous pass + fused warp/residual/Jacobian pass)" % (W, H, METHOD_NAMES[method])),
            "width": W, "height": H, "method": METHOD_NAMES[method], "n_pyr": 4,
            "pairs_per_gpu": 1, "parallelism": "independent pairs per GPU, RCCL all-gather of poses" if world > 1 else "1 GPU",
        },
        "timed_regions": {"repeats": repeats, "steps_each": args.steps, "statistic": "median of per-region max over ranks",
                          "elapsed_s_min": srt[0], "elapsed_s_median": elapsed, "elapsed_s_max": srt[-1],
                          "value_from_fastest_region": n_gpus * args.steps / srt[0],
                          "value_from_slowest_region": n_gpus * args.steps / srt[-1]},
    }

    # ---- BASELINE.json configs[3]: the odometry sequence, contiguous shards of pairs over the ranks ----------
    seq_block = None
    if not args.no_sequence:
        seq_block = run_sequence_block(args, torch, dist, synth, reg, rank, world, local_rank, xdev, W, H, gather_poses, shard_range, sync_all)
        if rank == 0:
            result["sequence"] = seq_block

    # every collective is behind us: the other ranks leave, rank 0 goes on alone (kernel timers, the child process of the
    # single-process multi-GPU entry, the CPU baseline) with no rank waiting on it
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        dist = None

    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events on the library's stream -----------------------------
        ws = WORKING_SET_PER_PX[method] * n_px
        fused = True          # (the two-launch schedule is a test hook since round 6: rgbd360_debug_set_schedule)
        kernel_us, kernel_batches = avg_kernel_us(lambda: reg.time_eval_kernel(0, pose_gpu, method, True, 50))       # the pass alone
        fused_us, fused_batches = (avg_kernel_us(lambda: reg.time_eval_kernel(0, pose_gpu, method, 2, 50)) if fused else (None, None))
        traffic, traffic_source = None, None
        live_failed = None
        if not args.no_live_traffic:
            traffic, traffic_source = live_traffic(W, H, method)
            if traffic is None:
                live_failed, traffic_source = traffic_source, None
        tr_path = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if traffic is None and os.path.exists(tr_path):
            try:
                tr = json.load(open(tr_path))
                key = "%dx%d_%s" % (W, H, METHOD_NAMES[method])
                traffic = tr.get(key, {}).get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_source = ("tracked file profiles/traffic_latest.json (%s): rocprofv3 PMC passes of an earlier run of this workload, "
                                      "2 x FETCH_SIZE + WRITE_SIZE per launch of %s; NOT collected in this run%s" % (tr.get("collected", "undated"), tr.get(key, {}).get("kernel", "the per-pixel pass"),
                                                                                                                    (" (" + live_failed + ")") if live_failed else ""))
            except Exception:
                traffic = None
        res_note = ("back-to-back launches over one pair: the %.0f MB working set stays in the 256 MiB Infinity Cache, which is the "
                    "regime of the product path (every iteration of a level re-reads the same pair); peak is still the HBM spec "
                    "figure.  roofline_hbm_rotating / roofline_4096x2048 are the HBM-fed measurements." % (ws / 1e6)) if ws < LLC_BYTES else ""
        pass_only = roofline_entry(kernel_us, kernel_batches, n_px, method, traffic=traffic, traffic_source=traffic_source,
                                   resident=("infinity_cache" if ws < LLC_BYTES else "hbm"),
                                   what="the warp + residual + Jacobian + normal-equation pass alone (k_eval: eval_span without the solve prologue)")
        if fused:
            result["roofline"] = roofline_entry(
                fused_us, fused_batches, n_px, method, kernel="k_eval_fs<%d>" % method, traffic=traffic, traffic_source=traffic_source,
                resident=("infinity_cache" if ws < LLC_BYTES else "hbm"), note=res_note,
                what=("the dominant kernel of `value`: ONE launch per Gauss-Newton iteration = solve of the previous pass (reduction of the "
                      "partial rows, 6x6 inverse, SE(3) exponential; redundantly per block) + the warp/residual/Jacobian pass over every "
                      "source pixel; back-to-back launches in the forced schedule"),
                pass_only=pass_only)
            # what else bounds the launch (round 5, profiles/r05_batched_form_experiment.txt): not part of the roofline contract, a reader's note
            result["roofline"]["issue_bound"] = {
                "vector_instructions_per_pixel": 135, "ns_of_a_simd_per_wave_step_4_waves": 178.1,
                "pure_issue_us_per_launch_2048x1024": 6.5, "serial_us_per_launch": 6.7, "ceiling_frac_of_hbm_roof": 0.56,
                "note": ("tools/ubench/front_rate.hip on MI355X: the two halves of a pixel cost a SIMD 178 ns per wave-step with four waves always ready; "
                         "32 wave-steps per SIMD and launch + the block reduction = 6.5 us of pure vector issue, around it a serial chain (kernel boundary, "
                         "partial rows, 6x6 inverse, SE(3) exponential, block tail) during which the vector units idle: 13.2 us = 0.56 of the HBM roof is the "
                         "ceiling of this kernel on this instruction stream, whatever the memory system does")}
        else:
            result["roofline"] = dict(pass_only, note=res_note)
        # the same kernel with every launch HBM-fed: rotate over enough copies of the pair to exceed the Infinity Cache
        n_rot = max(2, int(np.ceil(1.6 * LLC_BYTES / ws)))
        rot = [reg]
        for _ in range(0 if args.no_rotating else n_rot - 1):
            r2 = RegisterPhotoICP(device=local_rank)
            r2.setNumPyr(4)
            r2.setTargetFrame(rgbA, dA)
            r2.setSourceFrame(rgbB, dB)
            rot.append(r2)
        result["roofline_hbm_rotating"] = {}
        for m in ([] if args.no_rotating else sorted({method, 2})):
            us, bt = avg_kernel_us(lambda: RegisterPhotoICP.time_eval_kernel_rotating(rot, 0, pose_gpu, m, True, 10 * n_rot))
            ent = roofline_entry(us, bt, n_px, m, resident="hbm", copies=n_rot, rotating_working_set_bytes=n_rot * WORKING_SET_PER_PX[m] * n_px)
            if fused:
                us_f, bt_f = avg_kernel_us(lambda: RegisterPhotoICP.time_eval_kernel_rotating(rot, 0, pose_gpu, m, 2, 10 * n_rot))
                ent = dict(roofline_entry(us_f, bt_f, n_px, m, kernel="k_eval_fs<%d>" % m, resident="hbm", copies=n_rot,
                                          rotating_working_set_bytes=n_rot * WORKING_SET_PER_PX[m] * n_px), pass_only=ent)
            result["roofline_hbm_rotating"][METHOD_NAMES[m]] = ent
        for r2 in rot[1:]:
            r2.close()
        # same kernel in photo+depth mode (configs[2]) for reference
        other = 2 if method == 0 else 0
        k2, b2 = avg_kernel_us(lambda: reg.time_eval_kernel(0, pose_gpu, other, True, 50))
        result["roofline_other_method"] = dict(roofline_entry(k2, b2, n_px, other, resident="infinity_cache"), method=METHOD_NAMES[other])
        # whole-iteration efficiency: the bytes of one pass over the duration of one step as the host clock sees it
        solve_us = reg.time_solve_kernel(0, 0, 50)
        it_s = elapsed / args.steps
        result["iteration"] = {
            "bytes_per_iteration": pass_bytes_per_px(method, n_px) * n_px, "us_per_iteration": it_s * 1e6,
            "launch_us": fused_us if fused else kernel_us + solve_us, "pass_only_us": kernel_us, "separate_solve_launch_us": solve_us,
            "gap_and_host_us": it_s * 1e6 - (fused_us if fused else kernel_us + solve_us),
            "achieved": pass_bytes_per_px(method, n_px) * n_px / it_s / 1e9, "unit": "GB/s",
            "frac": pass_bytes_per_px(method, n_px) * n_px / it_s / 1e9 / HBM_PEAK_GBS,
            "frac_sustained": pass_bytes_per_px(method, n_px) * n_px * sustained["gn_iterations_per_s"] / n_gpus / 1e9 / HBM_PEAK_GBS,
            "note": ("value's own step: one k_eval_fs launch (+ the K-step call's fixed cost / K: the tail solve launch, the host wake-up)" if fused else
                     "value's own step: one k_eval pass + one k_solve launch + launch gaps (+ the K-step call's fixed cost / K)")}
        # the same forced schedule in the sequence engine's regime: 16 pairs iterate in lock step, one {pass, solve} launch pair
        # serving all of them (their records are separate allocations: 16 working sets, past the Infinity Cache)
        if not args.no_sequence:
            P = 16
            best = None
            for _ in range(5):
                fb = reg.forced_iters_batch(P, (rgbA, dA), (rgbB, dB), 0, start_pose, method, args.steps)
                best = fb if best is None or fb["elapsed_ms"] < best["elapsed_ms"] else best
            same = bool(all(np.array_equal(best["poses"][k], out["pose"]) for k in range(P)))
            t_it = best["elapsed_ms"] * 1e-3 / (P * args.steps)
            bpp_e = pass_bytes_per_px(method, n_px, engine=True)
            result["iteration_lockstep"] = {
                "pairs_in_flight": P, "gn_iterations_per_s": 1.0 / t_it, "us_per_pair_iteration": t_it * 1e6,
                "achieved": bpp_e * n_px / t_it / 1e9, "unit": "GB/s", "bytes_per_pixel": bpp_e,
                "frac": bpp_e * n_px / t_it / 1e9 / HBM_PEAK_GBS,
                "resident": "hbm" if P * ws >= LLC_BYTES else "infinity_cache",
                "poses_bit_identical_to_single_pair": same,
                # the batch pass alone (k_eval_b, all P slots per launch): HIP events over ten back-to-back launches
                "pass": {"kernel": "k_eval_b<%d,true>" % method, "pairs_per_launch": P, "avg_launch_us": best["pass_avg_us"],
                         "algorithmic_bytes_per_launch": P * bpp_e * n_px, "bytes_per_pixel": bpp_e,
                         "achieved": P * bpp_e * n_px / (best["pass_avg_us"] * 1e-6) / 1e9, "unit": "GB/s",
                         "frac": P * bpp_e * n_px / (best["pass_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "resident": "hbm" if P * ws >= LLC_BYTES else "infinity_cache"},
                "pass_photo_depth": None,
                "note": "rgbd360_forced_iters_batch: the step of `value` with 16 pairs per launch (k_eval_b / k_solve_b), HIP events, "
                        "best of 5; not `value` (configs[1] is a single pair)"}
        if not args.no_sequence and method != 2:
            # the same batch pass in the mode of configs[3] (photo + depth, 40 B/px): what the sequence engine's level-0 launches do
            fb2 = min((reg.forced_iters_batch(P, (rgbA, dA), (rgbB, dB), 0, start_pose, 2, 4) for _ in range(3)), key=lambda r: r["pass_avg_us"])
            by2 = P * pass_bytes_per_px(2, n_px, engine=True) * n_px
            result["iteration_lockstep"]["pass_photo_depth"] = {
                "kernel": "k_eval_b<2,true>", "pairs_per_launch": P, "avg_launch_us": fb2["pass_avg_us"], "algorithmic_bytes_per_launch": by2,
                "bytes_per_pixel": pass_bytes_per_px(2, n_px, engine=True),
                "achieved": by2 / (fb2["pass_avg_us"] * 1e-6) / 1e9, "unit": "GB/s", "frac": by2 / (fb2["pass_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "resident": "hbm"}
        srt_a = sorted(t_align)
        result["alignment"] = {"full_pyramid_ms": srt_a[len(srt_a) // 2] * 1e3, "full_pyramid_ms_min": srt_a[0] * 1e3,
                               "iters_per_level": iters_nat, "status": rc,
                               "pose_err_vs_ground_truth": dict(zip(("rot_rad", "trans_m"), synth.pose_error(pose_gpu, T_gt)))}
        result["setup_s"] = {"render_pair": t_gen}
        # ---- the same step with the warp in the REFERENCE's own arithmetic (rgbd360_set_index_arithmetic(ctx, 1): asinf / atan2f / roundf as
        #      glibc computes them, target indices bit-equal to a CPU build of the reference).  Information, not `value`: the default is the
        #      device definition. ----
        if n_gpus == 1:
            try:
                reg.set_index_arithmetic(1)
                itl = min((reg.forced_iters(0, start_pose, method, 200) for _ in range(3)), key=lambda r: r["elapsed_ms"])
                pl, _bt = avg_kernel_us(lambda: reg.time_eval_kernel(0, start_pose, method, True, 30))
                reg.alignFrames360(np.eye(4), method)
                t0 = time.perf_counter()
                for _ in range(10):
                    reg.alignFrames360(np.eye(4), method)
                al = (time.perf_counter() - t0) / 10
                result["reference_arithmetic"] = {
                    "gn_iterations_per_s": 200 / (itl["elapsed_ms"] * 1e-3), "ms_per_step": itl["elapsed_ms"] / 200, "pass_avg_us": pl,
                    "full_pyramid_ms": al * 1e3, "iters_per_level": list(reg.num_iterations),
                    "note": ("rgbd360_set_index_arithmetic(ctx, 1): the spherical warp as the reference computes it (csrc/libm_f32.h: glibc's asinf / "
                             "atan2f / roundf operation for operation) -- target indices bit-exact against the libm oracle "
                             "(tests: *_in_the_reference_arithmetic_*); opt-in, not the configuration `value` is measured on")}
            except Exception as e:      # (a measurement block must not cost the line)
                result["reference_arithmetic"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            finally:
                reg.set_index_arithmetic(0)

        # ---- the same kernel at 4096x2048 (BASELINE.json configs[4]); PHOTO_DEPTH (335 MB) exceeds the Infinity Cache,
        #      PHOTO_CONSISTENCY (235 MB) does not ----
        if n_gpus == 1 and not args.no_4k:
            (a4, d4), (b4, e4), _ = synth.make_pair(4096, 2048, seed=1234)
            reg4 = RegisterPhotoICP(device=local_rank)
            reg4.setNumPyr(5)
            reg4.setTargetFrame(a4, d4)
            reg4.setSourceFrame(b4, e4)
            reg4.alignFrames360(np.eye(4), 2)
            p4 = reg4.getOptimalPose()
            result["roofline_4096x2048"] = {}
            for m in (0, 2):
                us, bt = avg_kernel_us(lambda: reg4.time_eval_kernel(0, p4, m, True, 30))
                it = reg4.forced_iters(0, np.eye(4), m, 100)
                ws4 = WORKING_SET_PER_PX[m] * 4096 * 2048
                result["roofline_4096x2048"][METHOD_NAMES[m]] = dict(
                    roofline_entry(us, bt, 4096 * 2048, m, resident="infinity_cache" if ws4 < LLC_BYTES else "hbm", working_set_bytes=ws4),
                    gn_iterations_per_s=100 / (it["elapsed_ms"] * 1e-3))
            reg4.close()
            # ---- rows a13-a15 (Frame360 cloud / normal map / plane stage; the other half of configs[4]) at both sizes ----
            from rgbd360_amd.register import Frame360Stages
            try:
                result["roofline_frame360"] = {
                    "2048x1024": frame360_roofline(torch, Frame360Stages, RegisterPhotoICP, local_rank, dA),
                    "4096x2048": frame360_roofline(torch, Frame360Stages, RegisterPhotoICP, local_rank, d4),
                    "note": ("one rgbd360_frame_planes_dev call per repetition on a range panorama resident in HBM (the synthetic room); per stage: "
                             "SURVEY.md 8d's bytes per pixel x pixels / the stage's time between HIP events on the library's stream / 8 TB/s; "
                             "kernel-by-kernel times and counters: profiles/r06_frame360_*")}
            except Exception as e:      # (a measurement block must not cost the line)
                result["roofline_frame360"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

        # ---- the single-process multi-GPU entry (rgbd360_multi_*: host thread per device + ncclAllGather), in a child process
        #      so that nothing it does can cost the bench line ----
        frames_path = (seq_block or {}).pop("_frames_path", None)
        if not args.no_native_multi and not args.no_sequence and not share:
            result["native_multi"] = run_native_multi_child(world, args, W, H, frames_path)
        elif frames_path:
            try:
                os.remove(frames_path)
            except OSError:
                pass

        # ---- CPU baseline: the oracle on this host's cores, bounded sample (rank 0, N = 1 only) ---------------
        if n_gpus == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            ora = O.Oracle(n_pyr=4, math_mode=0, reduce_mode=0)      # reference-faithful modes
            ora.set_target(rgbA, dA)
            ora.set_source(rgbB, dB)
            st, pose_cpu = ora.align360(np.eye(4), method)
            rot_e, trans_e = synth.pose_error(pose_gpu, pose_cpu)
            result["alignment"]["pose_err_vs_cpu_ref"] = {"rot_rad": rot_e, "trans_m": trans_e}
            result["alignment"]["cpu_iters_per_level"] = list(ora.result.iters)[:4]
            # thread count: the reference uses every OpenMP thread the process may run on.  What this job may use is its CPU AFFINITY
            # (a 1-GPU lease gets 16 of the node's 256 hardware threads), not os.cpu_count(): the sweep is capped there, and the
            # baseline is quoted at the best count of the sweep (more threads than cores only lose on this memory-bound loop)
            allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            max_thr = max(1, min(O.num_threads(), allowed))
            cands = sorted({t for t in (1, 2, 4, 8, 16, 32, 64, 128, max_thr) if t <= max_thr})
            per_it_by_thr = {}
            for thr in cands:
                O.set_num_threads(thr)
                ora.forced_iters(0, np.eye(4), method, 1)       # warm
                n_s = 3 if thr > 1 else 2
                t0 = time.perf_counter()
                ora.forced_iters(0, np.eye(4), method, n_s)
                per_it_by_thr[thr] = (time.perf_counter() - t0) / n_s
            best_thr = min(per_it_by_thr, key=per_it_by_thr.get)
            best_per_it = per_it_by_thr[best_thr]
            O.set_num_threads(best_thr)
            n_cpu = int(max(8, min(5000, args.cpu_seconds / max(best_per_it, 1e-6))))
            t0 = time.perf_counter()
            ora.forced_iters(0, np.eye(4), method, n_cpu)
            dt = time.perf_counter() - t0
            result["cpu_baseline"] = {
                "value": n_cpu / dt, "unit": "GN iterations/s", "cores": best_thr, "kind": "port",
                "sample": "%d level-0 forced GN iterations (H,g pass + solve + error pass, reference structure with its "
                          "per-call Jacobian arrays) on the same %dx%d %s pair, %.1f s, best of OpenMP thread counts %s"
                          % (n_cpu, W, H, METHOD_NAMES[method], dt, cands),
                "cores_allowed": allowed, "host_cpus": os.cpu_count(),
                "single_thread_value": 1.0 / per_it_by_thr[1] if 1 in per_it_by_thr else None,
                "gn_iterations_per_s_by_threads": {str(k): 1.0 / v for k, v in sorted(per_it_by_thr.items())},
                "note": "cores = OpenMP threads of the quoted figure; cores_allowed = this process's CPU affinity (what the job may use), "
                        "the sweep never exceeds it; single_thread_value = the reference-faithful single-thread figure (SURVEY.md 8d)",
            }
        print(json.dumps(result), flush=True)


def run_sequence_block(args, torch, dist, synth, reg0, rank, world, local_rank, xdev, W, H, gather_poses, shard_range, sync_all):
    """configs[3]: args.seq_pairs consecutive pairs, contiguous shards over the ranks (rgbd360_amd.batch.shard_range), every rank
    aligning its shard through ONE call of the library's sequence entry, then one all-gather of {pose, status, iters}.
    Each rank renders SEQ_UNIQUE_FRAMES frames of the trajectory (its own stretch) and walks them back and forth -- consecutive
    entries are always neighbouring frames, i.e. genuine 6 cm / 2 degree odometry pairs -- because rendering 257 full-size frames
    on the host would take minutes.  Timed: (a) frames resident in HBM, (b) host frames (H2D inside the call)."""
    from rgbd360_amd.register import RegisterPhotoICP
    n_total = args.seq_pairs
    lo, hi = shard_range(n_total, rank, world)
    n_loc = hi - lo
    t0 = time.time()
    uniq = [synth.render(synth.trajectory_pose(rank * (SEQ_UNIQUE_FRAMES - 1) + k, 7), W, H, 7) for k in range(SEQ_UNIQUE_FRAMES)]
    t_render = time.time() - t0
    order = pingpong(n_loc, SEQ_UNIQUE_FRAMES)
    dev = torch.device("cuda", local_rank)
    # every frame of the walk is its OWN copy in HBM (n_loc + 1 distinct buffers, 10.5 MB each), as a recorded sequence would be:
    # with only the SEQ_UNIQUE_FRAMES rendered images resident (round 2) the raw inputs of the set-up kernels stayed in the Infinity
    # Cache and `resident` read 6 % high -- the whole of the gap to `native_multi`, whose loader always made distinct copies
    rgb_t = [torch.from_numpy(uniq[k][0]).to(dev) for k in order]
    dep_t = [torch.from_numpy(uniq[k][1].view(np.int16)).to(dev) for k in order]       # uint16 bits; torch has no uint16 kernels, none needed
    torch.cuda.synchronize()
    rgb_ptrs = [t.data_ptr() for t in rgb_t]
    dep_ptrs = [t.data_ptr() for t in dep_t]
    host_frames = [uniq[k] for k in order]
    reg = RegisterPhotoICP(device=local_rank)
    reg.setNumPyr(4)
    method = 2          # PHOTO_DEPTH, what OdometryRGBD360.cpp:192 runs
    out = {}
    for variant in ("resident", "host_frames"):
        def run():
            if variant == "resident":
                return reg.alignSequenceDev(rgb_ptrs, dep_ptrs, H, W, 0, method=method, n_inflight=SEQ_INFLIGHT)
            return reg.alignSequence(host_frames, method=method, n_inflight=SEQ_INFLIGHT)
        if n_loc > 0:
            if variant == "resident":
                reg.alignSequenceDev(rgb_ptrs[:2 * SEQ_INFLIGHT + 1], dep_ptrs[:2 * SEQ_INFLIGHT + 1], H, W, 0, method=method, n_inflight=SEQ_INFLIGHT)      # warm (engines, buffers)
            else:
                reg.alignSequence(host_frames[:2 * SEQ_INFLIGHT + 1], method=method, n_inflight=SEQ_INFLIGHT)
        times, res = [], None
        for _ in range(3):
            sync_all()
            t0 = time.perf_counter()
            if n_loc > 0:
                res = run()
            else:
                res = (np.zeros((0, 4, 4), np.float32), np.zeros(0, np.int32), np.zeros((0, 4), np.int32))
            full, st, it = gather_poses(res[0], n_total, dist, device=(None if xdev == "cpu" else dev), status=res[1], iters=res[2])
            sync_all()
            times.append(time.perf_counter() - t0)
        if dist is not None:
            tm = torch.tensor(times, dtype=torch.float64, device=xdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            times = [float(x) for x in tm.cpu()]
        srt = sorted(times)
        med = srt[len(srt) // 2]
        # size-independent checks: every pair converged; a pair that recurs in the walk gives bit-identical poses every time
        # (fixed-order reductions, no atomics); forward and backward pass over one pair of frames are inverse motions
        my = full[lo:hi]
        first_seen, repeats_equal = {}, True
        for j in range(n_loc):
            key = (order[j], order[j + 1])
            if key in first_seen:
                repeats_equal &= bool(np.array_equal(my[first_seen[key]], my[j]))
            else:
                first_seen[key] = j
        inv_err = 0.0
        for (a, b), j in first_seen.items():
            if (b, a) in first_seen:
                e = synth.pose_error(my[j].astype(np.float64) @ my[first_seen[(b, a)]].astype(np.float64), np.eye(4))
                inv_err = max(inv_err, e[0], e[1])
        gt_err = [0.0, 0.0]
        for (a, b), j in first_seen.items():
            base = rank * (SEQ_UNIQUE_FRAMES - 1)
            T = np.linalg.inv(synth.trajectory_pose(base + a, 7)) @ synth.trajectory_pose(base + b, 7)
            e = synth.pose_error(my[j], T)
            gt_err = [max(gt_err[0], e[0]), max(gt_err[1], e[1])]
        # compulsory bytes of one alignment of the sequence (SURVEY.md 8d figures): the set-up of the ONE new frame a pair brings
        # (level 0 reads 5 B/px of raw colour + depth, levels >= 1 read their 8 B/px of float planes; every level writes 16 + 12 + 12 B/px
        # of source / target records and the next level's 8 B/px of planes), and per level (accepted iterations + the first pass + the
        # pass whose rejection ends the level) fused passes of 40 B/px
        n_l = [(H >> l) * (W >> l) for l in range(4)]
        rec_b = [(8 if n_l[l] >= SEQ_RECOMPUTE_MIN_PX else 16) + 24 for l in range(4)]      # source record (compact on the large levels) + both target records
        setup_b = sum((5 if l == 0 else 8) * n_l[l] + rec_b[l] * n_l[l] + (8 * n_l[l + 1] if l + 1 < 4 else 0) for l in range(4))
        mean_it = it.mean(0) if len(it) else np.zeros(4)
        pass_b = float(sum(pass_bytes_per_px(method, n_l[l], engine=True) * n_l[l] * (mean_it[l] + 1 + (1 if mean_it[l] < 10 else 0)) for l in range(4)))
        pair_b = setup_b + pass_b
        ach = pair_b * n_total / med / 1e9 / world         # per GPU
        out[variant] = {
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "per": "GPU",
                         "algorithmic_bytes_per_alignment": pair_b, "setup_bytes": setup_b, "pass_bytes": pass_b,
                         "model": "frame set-up of one new frame (5 | 8 B/px in, 32 | 40 B/px records + 8 B/px next-level planes out, 4 levels) + "
                                  "sum over levels of (mean accepted iterations + 2) passes x 32 | 40 B/px (levels of 256 Kpx and more carry 8-byte {depth, I} source "
                                  "records and re-form the point in the pass, the small ones the 16-byte record); PCIe bytes of host frames not counted"},
            "alignments_per_s": n_total / med, "elapsed_ms_median": med * 1e3, "elapsed_ms_all": [t * 1e3 for t in times],
            "ms_per_pair_per_gpu": med * 1e3 / max(1, -(-n_total // world)),
            "all_status_ok": bool((st == 0).all()), "repeated_pairs_bit_identical": repeats_equal,
            "max_forward_backward_residual": inv_err, "max_pose_err_vs_ground_truth": {"rot_rad": gt_err[0], "trans_m": gt_err[1]},
            "mean_iters_per_level": np.round(it.mean(0), 3).tolist() if len(it) else [],
        }
    reg.close()
    # The host-frame leg is bound by PCIe, not by HBM: every alignment brings ONE new frame over the link (rows x cols x (3 + 2) bytes).
    # Its roofline is the host-to-device copy rate of this box, measured here with a pinned buffer (the frames themselves are pageable
    # numpy arrays the library copies with hipMemcpy2DAsync, as a caller's cv::Mat data would be).
    if "host_frames" in out and n_loc > 0:
        try:
            nb = 64 << 20
            h_pin = torch.empty(nb, dtype=torch.uint8).pin_memory()
            d_buf = torch.empty(nb, dtype=torch.uint8, device=dev)
            d_buf.copy_(h_pin, non_blocking=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                d_buf.copy_(h_pin, non_blocking=True)
            torch.cuda.synchronize()
            h2d = 8 * nb / (time.perf_counter() - t0) / 1e9
            frame_b = W * H * 5
            rate = out["host_frames"]["alignments_per_s"] / world * frame_b / 1e9
            out["host_frames"]["pcie"] = {
                "bound": "pcie", "bytes_per_alignment": frame_b, "achieved": rate, "unit": "GB/s", "peak": h2d, "frac": rate / h2d,
                "peak_source": "8 x 64 MiB pinned host -> device copies on this box, in this run",
                "note": ("one new %d x %d frame (8UC3 + 16UC1 = %.1f MB) per alignment and GPU; at the measured copy rate the link carries at most %.0f alignments/s per GPU, "
                         "whatever the kernels do -- PCIe 5.0 x16 is 63 GB/s raw, i.e. 6.0 k/s even on paper" % (W, H, frame_b / 1e6, h2d * 1e9 / frame_b))}
        except Exception as e:
            out["host_frames"]["pcie"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
    out.update({"workload": "configs[3]: %d consecutive %dx%d pairs (PHOTO_DEPTH, 4 levels), contiguous shards over %d rank(s), "
                            "rgbd360_align360_batch[_dev] per rank (lock-step engine, %d pairs in flight), one all-gather of pose/status/iters"
                            % (n_total, W, H, world, SEQ_INFLIGHT),
                "pairs_total": n_total, "pairs_per_rank": -(-n_total // world), "unique_frames_per_rank": SEQ_UNIQUE_FRAMES,
                "resident_frames_are_distinct_copies": True,
                "render_s": t_render, "exchange": "gloo (shared device)" if xdev == "cpu" and world > 1 else ("rccl" if world > 1 else "none")})
    # the child process of the native multi-GPU entry re-uses rank 0's frames instead of rendering again
    if rank == 0:
        try:
            path = "/tmp/rgbd360_bench_frames_%d.npz" % os.getpid()
            np.savez(path, rgb=np.stack([f[0] for f in uniq]), depth=np.stack([f[1] for f in uniq]))
            out["_frames_path"] = path
        except Exception:
            pass
    return out


def run_native_multi_child(world, args, W, H, path):
    """tools/native_multi_bench.py in a child process: the C entry rgbd360_multi_* with n_gpus = the job's GPU count, the same
    sequence, resident frames.  Run after the ranks' own measurements; a crash or a hang there cannot touch this process."""
    if path is None:
        return {"error": "no frames"}
    import torch
    n_dev = min(world, torch.cuda.device_count())
    cmd = [sys.executable, os.path.join(ROOT, "tools", "native_multi_bench.py"), str(n_dev), str(args.seq_pairs), str(W), str(H), path]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
        line = [l for l in p.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": "child rc %d: %s" % (p.returncode, p.stderr.decode(errors="replace")[-400:])}
        return json.loads(line[-1])
    except subprocess.TimeoutExpired:
        return {"error": "timeout"}
    except Exception as e:       # the bench line must come out whatever happens here
        return {"error": repr(e)}
    finally:
        try:
            os.remove(path)
        except OSError:
            pass


if __name__ == "__main__":
    main()
