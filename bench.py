#!/usr/bin/env python3
"""bench.py -- Gauss-Newton iterations/sec of the dense spherical alignment hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches this file
with torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

Step      = one level-0 Gauss-Newton iteration of RegisterPhotoICP::alignFrames360 on one 2048x1024 synthetic
            spherical pair (BASELINE.json configs[1]: photometric-only), in the forced schedule of BASELINE.md §2
            (accept rule evaluated, step applied regardless): one fused warp+residual+Jacobian pass over every
            source pixel + the 6x6 solve / pose update launch.  Frames are resident in HBM before the timed region.
N > 1     = every rank aligns its own independent pair (weak scaling) and the solved poses are all-gathered
            (RCCL) inside the timed region.
roofline  = algorithmic bytes of the fused kernel (28 B/px photo, 40 B/px photo+depth, LUT variant)
            / its average launch duration measured with HIP events on the library's own stream.
cpu_baseline = the CPU oracle (restatement of the reference algorithm, OpenMP) on the same workload, bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # measured float4 copy (same guide)
BYTES_PER_PX = {0: 28, 1: 28, 2: 40}   # LUT variant: 16 B source record + 12 B (24 B) gathered target records
METHOD_NAMES = {0: "PHOTO_CONSISTENCY", 1: "DEPTH_CONSISTENCY", 2: "PHOTO_DEPTH"}


def avg_kernel_us(reg, level, pose, method, reps, batches=5):
    """Average launch duration of the fused kernel (HIP events on the library's stream around `reps` back-to-back launches);
    the median of `batches` such averages, so that one disturbed batch (another process touching the device) cannot skew it."""
    vals = sorted(reg.time_eval_kernel(level, pose, method, True, reps) for _ in range(batches))
    return vals[len(vals) // 2], vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--method", type=int, default=0, help="0 photo (configs[1]), 2 photo+depth (configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-4k", action="store_true", help="skip the extra 4096x2048 (configs[4]) kernel measurement")
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

    # full coarse-to-fine alignment once (natural accept/reject schedule): pose + the level-0 starting pose
    t0 = time.time()
    rc = reg.alignFrames360(np.eye(4), method)
    t_align = time.time() - t0
    t0 = time.time()
    rc = reg.alignFrames360(np.eye(4), method)
    t_align = min(t_align, time.time() - t0)
    pose_gpu = reg.getOptimalPose()
    iters_nat = list(reg.num_iterations)
    start_pose = np.eye(4)     # forced schedule starts from the identity guess at level 0

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup, then EXACTLY K steps ------------------------------------------------------------------------
    reg.forced_iters(0, start_pose, method, max(args.warmup, 1))
    if dist is not None:
        # warm the exchange too: the first RCCL all-gather builds channels / loads kernels, which must not be timed
        w_in = torch.zeros(16, dtype=torch.float32, device=xdev)
        w_out = torch.empty(world * 16, dtype=torch.float32, device=xdev)
        dist.all_gather_into_tensor(w_out, w_in)
        w_t = torch.zeros(1, dtype=torch.float64, device=xdev)
        dist.all_reduce(w_t, op=dist.ReduceOp.MAX)
    sync_all()
    t0 = time.perf_counter()
    out = reg.forced_iters(0, start_pose, method, args.steps)
    poses = None
    if dist is not None:
        mine = torch.from_numpy(out["pose"].reshape(16).copy()).to(xdev)
        gathered = torch.empty(world * 16, dtype=torch.float32, device=xdev)
        dist.all_gather_into_tensor(gathered, mine)      # RCCL over xGMI: the path's one exchange step
        poses = gathered
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        assert poses is not None and bool(torch.isfinite(poses).all())

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
        "config": {
            "workload": ("configs[1]: single %dx%d synthetic sphere pair per GPU, %s RegisterPhotoICP, level-0 forced "
                         "Gauss-Newton iterations (1 fused pass + 1 solve launch each)" % (W, H, METHOD_NAMES[method])),
            "width": W, "height": H, "method": METHOD_NAMES[method], "n_pyr": 4,
            "pairs_per_gpu": 1, "parallelism": "independent pairs per GPU, RCCL all-gather of poses" if world > 1 else "1 GPU",
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events on the library's stream -----------------------------
        kernel_us, kernel_batches = avg_kernel_us(reg, 0, pose_gpu, method, 50)
        alg_bytes = BYTES_PER_PX[method] * n_px
        achieved = alg_bytes / (kernel_us * 1e-6) / 1e9
        traffic = None
        tr_path = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tr_path):
            try:
                tr = json.load(open(tr_path))
                key = "%dx%d_%s" % (W, H, METHOD_NAMES[method])
                traffic = tr.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "kernel": "k_eval<%d,true>" % method, "kernel_avg_us": kernel_us, "kernel_avg_us_batches": kernel_batches,
            "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_pixel": BYTES_PER_PX[method],
            "frac_of_measured_copy_peak": achieved / HBM_COPY_GBS,
        }
        # same kernel in photo+depth mode (configs[2]) for reference
        other = 2 if method == 0 else 0
        k2, _ = avg_kernel_us(reg, 0, pose_gpu, other, 50)
        result["roofline_other_method"] = {
            "method": METHOD_NAMES[other], "kernel_avg_us": k2,
            "achieved": BYTES_PER_PX[other] * n_px / (k2 * 1e-6) / 1e9,
            "frac": BYTES_PER_PX[other] * n_px / (k2 * 1e-6) / 1e9 / HBM_PEAK_GBS,
        }
        result["alignment"] = {"full_pyramid_ms": t_align * 1e3, "iters_per_level": iters_nat, "status": rc,
                               "pose_err_vs_ground_truth": dict(zip(("rot_rad", "trans_m"), synth.pose_error(pose_gpu, T_gt)))}
        result["setup_s"] = {"render_pair": t_gen}

        # ---- the same kernel at 4096x2048 (BASELINE.json configs[4]: working set beyond the 256 MiB Infinity Cache) ----
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
                us, _ = avg_kernel_us(reg4, 0, p4, m, 30)
                ach = BYTES_PER_PX[m] * 4096 * 2048 / (us * 1e-6) / 1e9
                it = reg4.forced_iters(0, np.eye(4), m, 100)
                result["roofline_4096x2048"][METHOD_NAMES[m]] = {
                    "kernel_avg_us": us, "achieved": ach, "frac": ach / HBM_PEAK_GBS, "frac_of_measured_copy_peak": ach / HBM_COPY_GBS,
                    "gn_iterations_per_s": 100 / (it["elapsed_ms"] * 1e-3)}
            reg4.close()

        # ---- CPU baseline: the oracle on this host's cores, bounded sample (rank 0, N = 1 only) ---------------
        if n_gpus == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            ora = O.Oracle(n_pyr=4, math_mode=0, reduce_mode=0)      # reference-faithful modes
            ora.set_target(rgbA, dA)
            ora.set_source(rgbB, dB)
            st, pose_cpu = ora.align360(np.eye(4), method)
            rot, trans = synth.pose_error(pose_gpu, pose_cpu)
            result["alignment"]["pose_err_vs_cpu_ref"] = {"rot_rad": rot, "trans_m": trans}
            result["alignment"]["cpu_iters_per_level"] = list(ora.result.iters)[:4]
            # thread count: the reference uses every OpenMP thread; on many-core hosts that is not the fastest
            # setting for this memory-bound loop, so the baseline is quoted at the best of a short sweep
            max_thr = O.num_threads()
            cands = sorted({t for t in (8, 16, 32, 64, 128, max_thr) if t <= max_thr})
            best_thr, best_per_it = max_thr, None
            for thr in cands:
                O.set_num_threads(thr)
                ora.forced_iters(0, np.eye(4), method, 1)       # warm
                t0 = time.perf_counter()
                ora.forced_iters(0, np.eye(4), method, 3)
                per_it = (time.perf_counter() - t0) / 3
                if best_per_it is None or per_it < best_per_it:
                    best_thr, best_per_it = thr, per_it
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
                "host_cpus": os.cpu_count(),
            }
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
