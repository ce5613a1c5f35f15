// knobs.h -- every environment variable the library reads, in one place.
//
// PRODUCT KNOBS (read by every build; INTEGRATION.md section "Runtime knobs"):
//   RGBD360_HOST_SPIN_US          host_wait.h: how long the host spins on a published tag before it starts yielding
//   GPU_MAX_HW_QUEUES             the HIP runtime's own variable; read to size the per-context route (more busy streams than hardware queues
//                                 are time-sliced by the command processor: rgbd360_api.hip ctx_route_cap)
//   RGBD360_RECOMPUTE_MIN_PX      single-pair levels of this many pixels and more run the pass in its recompute form (8 B less per source pixel)
//   RGBD360_SEQ_RECOMPUTE_MIN_PX  the same bound for the lock-step sequence engine
//   RGBD360_SEQ_ENGINES           lock-step engines (host thread + stream each) of rgbd360_align360_batch, 1..4 (default 2)
//   RGBD360_FORCE_RCCL            rgbd360_multi_*: run the ncclAllGather also with one device (tests: the rows travel through RCCL)
//
// DEBUG KNOBS: the A/B levers of the measurements in docs/HISTORY.md and profiles/.  A product build ignores them (debug() returns
// nullptr: the defaults are compiled in); `python -m rgbd360_amd.build --debug-knobs` (-DRGBD360_DEBUG_KNOBS) builds a library that reads
// them.  Settled levers whose losing path was deleted in round 6: RGBD360_ARENA, RGBD360_NORMALS_SWEEP, RGBD360_CCL_LISTS,
// RGBD360_CLOUD_X4, RGBD360_FUSE_CLOUD.  The schedules the parity tests compare (fused solve / fused occlusion build / per-context
// sequence route) are switched through rgbd360_hip_diag.h (rgbd360_debug_set_schedule, rgbd360_debug_set_sequence_route), not the environment.
#pragma once
#include <cstdlib>

namespace knobs {
inline const char* product(const char* name) { return std::getenv(name); }
inline const char* debug(const char* name) {
#ifdef RGBD360_DEBUG_KNOBS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
inline bool debug_build() {
#ifdef RGBD360_DEBUG_KNOBS
    return true;
#else
    return false;
#endif
}
}  // namespace knobs
