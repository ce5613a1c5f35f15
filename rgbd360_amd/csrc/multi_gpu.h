// multi_gpu.h -- one process, several MI355X: the odometry sequence of BASELINE.json configs[3] sharded over the devices of a
// node.  Included at the end of rgbd360_api.hip (uses its align360_batch_impl).
//
// The path shards by independent frame pairs: device d owns the contiguous pairs [lo_d, hi_d) and therefore the frames
// lo_d .. hi_d (the boundary frame hi_d = lo_{d+1} lives on both neighbours).  One host thread per device drives that device's
// contexts exactly as the single-GPU sequence entry does; there is NO data-path collective.  The one exchange step is an
// ncclAllGather (RCCL over xGMI) of the per-pair result rows {pose[16], rgbd360_result} at the end, after which every device
// -- and the host, which reads device 0's copy -- holds the whole trajectory; the caller composes it like
// OdometryRGBD360.cpp:257 (currentPose *= rel).  With one device there is no RCCL call.
#pragma once
#include <rccl/rccl.h>

struct rgbd360_multi {
    int n_gpus = 0;
    std::vector<int> dev;
    std::vector<rgbd360_ctx*> ctx;               // one primary context per device (siblings hang off it)
    std::vector<ncclComm_t> comm;                // empty when n_gpus == 1 and RCCL is not forced
    std::vector<unsigned char*> d_send, d_recv;  // per device: own rows / everybody's rows
    size_t send_cap = 0, recv_cap = 0;           // bytes
    // resident sequence (rgbd360_multi_load_sequence): per device the frames lo_d .. hi_d in HBM
    std::vector<uint8_t*> seq_rgb;
    std::vector<uint8_t*> seq_depth;
    std::vector<size_t> seq_rgb_bytes, seq_depth_bytes;
    std::vector<int> seq_lo, seq_hi;             // ... the pair span [lo_d, hi_d) each device's resident frames were cut for (fixed at load time)
    int seq_frames = 0, seq_rows = 0, seq_cols = 0, seq_depth_type = 0;      // seq_frames == 0: nothing (valid) is resident
    bool use_rccl = false;
    std::string err;
};

namespace {

constexpr size_t kRowBytes = 16 * sizeof(float) + sizeof(rgbd360_result);      // 64 + 232 = 296
static_assert(kRowBytes % 8 == 0, "result rows are exchanged as 8-byte words");

int mfail(rgbd360_multi* m, int code, const std::string& msg) {
    m->err = msg;
    return code;
}

void shard_range(int n_items, int rank, int world, int* lo, int* hi) {      // contiguous, balanced: the first n % world ranks get one more
    const int base = n_items / world, extra = n_items % world;
    *lo = rank * base + std::min(rank, extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
}

// The exchange buffer's layout (pure index arithmetic, exported as rgbd360_gather_slot for the CPU tests): every rank contributes
// max_chunk = ceil(n_pairs / world) rows -- ncclAllGather wants equal counts -- of which the first hi - lo are its pairs; global pair
// j therefore sits in row rank_of(j) * max_chunk + (j - lo_rank) of the gathered table, and the rows behind a rank's last pair are
// padding.
int gather_chunk(int n_pairs, int world) { return (n_pairs + world - 1) / world; }
void gather_slot(int n_pairs, int world, int pair, int* rank, int* row) {
    const int base = n_pairs / world, extra = n_pairs % world;
    // the first `extra` ranks own base + 1 pairs each: pairs below extra * (base + 1) belong to them
    int r;
    if (pair < extra * (base + 1)) r = pair / (base + 1);
    else r = base > 0 ? extra + (pair - extra * (base + 1)) / base : world - 1;
    int lo, hi;
    shard_range(n_pairs, r, world, &lo, &hi);
    *rank = r;
    *row = r * gather_chunk(n_pairs, world) + (pair - lo);
}

int multi_ensure_exchange(rgbd360_multi* m, int n_pairs) {
    const int max_chunk = gather_chunk(n_pairs, m->n_gpus);
    const size_t need_send = (size_t)std::max(max_chunk, 1) * kRowBytes, need_recv = need_send * m->n_gpus;
    if (m->send_cap >= need_send && m->recv_cap >= need_recv) return 0;
    for (int d = 0; d < m->n_gpus; ++d) {
        if (hipSetDevice(m->dev[d]) != hipSuccess) return mfail(m, -102, "hipSetDevice failed");
        hipFree(m->d_send[d]); hipFree(m->d_recv[d]);
        m->d_send[d] = m->d_recv[d] = nullptr;
        if (hipMalloc(&m->d_send[d], need_send) != hipSuccess || hipMalloc(&m->d_recv[d], need_recv) != hipSuccess)
            return mfail(m, -103, "cannot allocate the exchange buffers");
    }
    m->send_cap = need_send;
    m->recv_cap = need_recv;
    return 0;
}

// Runs the shards (one host thread per device), then the exchange.  frames_of(d, k) hands out the pointers of global frame k for
// device d (host images, or that device's resident copies).
template <class FrameOf>
int multi_run(rgbd360_multi* m, int n_frames, FrameOf frames_of, size_t rgb_step, size_t depth_step, int depth_type, int rows, int cols,
              const float guess[16], int method, int occlusion, int n_inflight, bool on_device, float* poses_out,
              rgbd360_result* results_out) {
    const int n_pairs = n_frames - 1;
    if (n_pairs <= 0) return 0;
    int rc = multi_ensure_exchange(m, n_pairs);
    if (rc) return rc;
    const int G = m->n_gpus;
    const int max_chunk = gather_chunk(n_pairs, G);
    std::vector<int> lo(G), hi(G), rcs(G, 0);
    for (int d = 0; d < G; ++d) shard_range(n_pairs, d, G, &lo[d], &hi[d]);
    std::vector<std::vector<unsigned char>> rows_host(G, std::vector<unsigned char>((size_t)max_chunk * kRowBytes, 0));
    auto run_device = [&](int d) {
        const int n_loc = hi[d] - lo[d];
        if (n_loc <= 0) return;
        hipSetDevice(m->dev[d]);
        std::vector<const uint8_t*> rp(n_loc + 1);
        std::vector<const void*> dp(n_loc + 1);
        for (int k = 0; k <= n_loc; ++k) frames_of(d, lo[d] + k, &rp[k], &dp[k]);
        std::vector<float> poses((size_t)n_loc * 16);
        std::vector<rgbd360_result> res(n_loc);
        rcs[d] = align360_batch_impl(m->ctx[d], n_loc + 1, rp.data(), rgb_step, dp.data(), depth_step, depth_type, rows, cols, guess, method,
                                     occlusion, n_inflight, poses.data(), res.data(), on_device);
        if (rcs[d]) return;
        for (int j = 0; j < n_loc; ++j) {
            unsigned char* row = rows_host[d].data() + (size_t)j * kRowBytes;
            memcpy(row, &poses[(size_t)j * 16], 16 * sizeof(float));
            memcpy(row + 16 * sizeof(float), &res[j], sizeof(rgbd360_result));
        }
    };
    {
        std::vector<std::thread> workers;
        std::vector<int> inline_devs;
        for (int d = 1; d < G; ++d) {
            try {
                workers.emplace_back(run_device, d);
            } catch (const std::exception&) {
                inline_devs.push_back(d);
            }
        }
        run_device(0);
        for (int d : inline_devs) run_device(d);
        for (std::thread& w : workers) w.join();
    }
    for (int d = 0; d < G; ++d)
        if (rcs[d]) return mfail(m, rcs[d], std::string("device ") + std::to_string(m->dev[d]) + ": " + m->ctx[d]->err);

    std::vector<unsigned char> all((size_t)G * max_chunk * kRowBytes);
    if (m->use_rccl) {
        // the path's one exchange step: every device contributes its rows, every device receives all of them
        for (int d = 0; d < G; ++d) {
            if (hipSetDevice(m->dev[d]) != hipSuccess) return mfail(m, -102, "hipSetDevice failed");
            if (hipMemcpyAsync(m->d_send[d], rows_host[d].data(), (size_t)max_chunk * kRowBytes, hipMemcpyHostToDevice, m->ctx[d]->stream) != hipSuccess)
                return mfail(m, -104, "upload of the result rows failed");
        }
        ncclResult_t nr = ncclGroupStart();
        for (int d = 0; d < G && nr == ncclSuccess; ++d) {
            hipSetDevice(m->dev[d]);
            nr = ncclAllGather(m->d_send[d], m->d_recv[d], (size_t)max_chunk * kRowBytes / 8, ncclUint64, m->comm[d], m->ctx[d]->stream);
        }
        const ncclResult_t ne = ncclGroupEnd();
        if (nr != ncclSuccess || ne != ncclSuccess)
            return mfail(m, -105, std::string("ncclAllGather: ") + ncclGetErrorString(nr != ncclSuccess ? nr : ne));
        for (int d = 0; d < G; ++d) {
            hipSetDevice(m->dev[d]);
            if (hipStreamSynchronize(m->ctx[d]->stream) != hipSuccess) return mfail(m, -106, "the all-gather did not complete");
        }
        hipSetDevice(m->dev[0]);
        if (hipMemcpy(all.data(), m->d_recv[0], all.size(), hipMemcpyDeviceToHost) != hipSuccess) return mfail(m, -104, "download of the gathered rows failed");
        // what came over the links must be what the shards produced (cheap, and the only check a 1-device box cannot fake)
        for (int d = 0; d < G; ++d)
            if (memcmp(all.data() + (size_t)d * max_chunk * kRowBytes, rows_host[d].data(), (size_t)(hi[d] - lo[d]) * kRowBytes) != 0)
                return mfail(m, -107, "gathered rows differ from the rows the shards produced");
    } else {
        for (int d = 0; d < G; ++d) memcpy(all.data() + (size_t)d * max_chunk * kRowBytes, rows_host[d].data(), (size_t)max_chunk * kRowBytes);
    }
    for (int j = 0; j < n_pairs; ++j) {
        int rank, slot;
        gather_slot(n_pairs, G, j, &rank, &slot);
        const unsigned char* row = all.data() + (size_t)slot * kRowBytes;
        memcpy(poses_out + (size_t)16 * j, row, 16 * sizeof(float));
        if (results_out) memcpy(&results_out[j], row + 16 * sizeof(float), sizeof(rgbd360_result));
    }
    return 0;
}

void multi_free_sequence(rgbd360_multi* m) {
    for (int d = 0; d < m->n_gpus; ++d) {
        hipSetDevice(m->dev[d]);
        hipFree(m->seq_rgb[d]); hipFree(m->seq_depth[d]);
        m->seq_rgb[d] = m->seq_depth[d] = nullptr;
        m->seq_rgb_bytes[d] = m->seq_depth_bytes[d] = 0;
    }
    m->seq_frames = 0;
}

}  // namespace

extern "C" {

void rgbd360_shard_range(int n_items, int rank, int world, int* lo, int* hi) {
    int a = 0, b = 0;
    if (world > 0 && rank >= 0 && rank < world && n_items >= 0) shard_range(n_items, rank, world, &a, &b);
    if (lo) *lo = a;
    if (hi) *hi = b;
}

void rgbd360_gather_slot(int n_pairs, int world, int pair, int* rank, int* row, int* rows_per_rank) {
    int r = -1, s = -1;
    if (world > 0 && n_pairs > 0 && pair >= 0 && pair < n_pairs) gather_slot(n_pairs, world, pair, &r, &s);
    if (rank) *rank = r;
    if (row) *row = s;
    if (rows_per_rank) *rows_per_rank = world > 0 && n_pairs >= 0 ? gather_chunk(n_pairs, world) : 0;
}

void rgbd360_multi_destroy(rgbd360_multi* m) {
    if (!m) return;
    for (size_t d = 0; d < m->comm.size(); ++d)
        if (m->comm[d]) ncclCommDestroy(m->comm[d]);
    if (!m->seq_rgb.empty()) multi_free_sequence(m);
    for (int d = 0; d < (int)m->ctx.size(); ++d) {
        if (d < (int)m->dev.size()) hipSetDevice(m->dev[d]);
        if (d < (int)m->d_send.size()) { hipFree(m->d_send[d]); hipFree(m->d_recv[d]); }
        rgbd360_destroy(m->ctx[d]);
    }
    delete m;
}

int rgbd360_multi_create(const rgbd360_params* p, int n_gpus, const int* device_ids, rgbd360_multi** out) {
    if (!p || !out || n_gpus < 1 || n_gpus > 64) return -1;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -100;      // no HIP device: no fallback
    rgbd360_multi* m = new rgbd360_multi();
    m->n_gpus = n_gpus;
    m->dev.resize(n_gpus);
    for (int d = 0; d < n_gpus; ++d) {
        m->dev[d] = device_ids ? device_ids[d] : d;
        if (m->dev[d] < 0 || m->dev[d] >= ndev) { delete m; return -101; }
        for (int e = 0; e < d; ++e)
            if (m->dev[e] == m->dev[d]) { delete m; return -101; }           // RCCL refuses two ranks on one device
    }
    m->d_send.assign(n_gpus, nullptr); m->d_recv.assign(n_gpus, nullptr);
    m->seq_rgb.assign(n_gpus, nullptr); m->seq_depth.assign(n_gpus, nullptr);
    m->seq_rgb_bytes.assign(n_gpus, 0); m->seq_depth_bytes.assign(n_gpus, 0);
    m->seq_lo.assign(n_gpus, 0); m->seq_hi.assign(n_gpus, 0);
    for (int d = 0; d < n_gpus; ++d) {
        rgbd360_params pd = *p;
        pd.device = m->dev[d];
        rgbd360_ctx* c = nullptr;
        const int rc = rgbd360_create(&pd, &c);
        if (rc) { rgbd360_multi_destroy(m); return rc; }
        m->ctx.push_back(c);
    }
    const char* force = knobs::product("RGBD360_FORCE_RCCL");          // exercise the exchange on a 1-GPU box
    m->use_rccl = n_gpus > 1 || (force && atoi(force) != 0);
    if (m->use_rccl) {
        m->comm.assign(n_gpus, nullptr);
        const ncclResult_t nr = ncclCommInitAll(m->comm.data(), n_gpus, m->dev.data());
        if (nr != ncclSuccess) {
            fprintf(stderr, "rgbd360_multi_create: ncclCommInitAll: %s\n", ncclGetErrorString(nr));
            m->comm.clear();
            rgbd360_multi_destroy(m);
            return -105;
        }
    }
    *out = m;
    return 0;
}

const char* rgbd360_multi_last_error(rgbd360_multi* m) { return m ? m->err.c_str() : "null handle"; }
int rgbd360_multi_n_gpus(rgbd360_multi* m) { return m ? m->n_gpus : 0; }
int rgbd360_multi_uses_rccl(rgbd360_multi* m) { return m && m->use_rccl ? 1 : 0; }
int rgbd360_multi_set_index_arithmetic(rgbd360_multi* m, int mode) {      // every device's context (and its engines / siblings) follows
    if (!m) return -1;
    for (rgbd360_ctx* c : m->ctx) {
        const int rc = rgbd360_set_index_arithmetic(c, mode);
        if (rc) return mfail(m, rc, rgbd360_last_error(c));
    }
    return 0;
}

int rgbd360_multi_align_sequence(rgbd360_multi* m, int n_frames, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
                                 size_t depth_step, int depth_type, int rows, int cols, const float guess[16], int method, int occlusion,
                                 int n_inflight, float* poses_out, rgbd360_result* results_out) {
    if (!m) return -1;
    if (!rgb || !depth || !poses_out || n_frames < 1) return mfail(m, -1, "bad arguments");
    for (int k = 0; k < n_frames; ++k)
        if (!rgb[k] || !depth[k]) return mfail(m, -1, "null frame pointer");
    auto frames_of = [&](int, int k, const uint8_t** r, const void** dpt) { *r = rgb[k]; *dpt = depth[k]; };
    return multi_run(m, n_frames, frames_of, rgb_step, depth_step, depth_type, rows, cols, guess, method, occlusion, n_inflight, false,
                     poses_out, results_out);
}

int rgbd360_multi_load_sequence(rgbd360_multi* m, int n_frames, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
                                size_t depth_step, int depth_type, int rows, int cols) {
    if (!m) return -1;
    if (!rgb || !depth || n_frames < 2 || rows < 1 || cols < 1 || (depth_type != 0 && depth_type != 1)) return mfail(m, -1, "bad arguments");
    for (int k = 0; k < n_frames; ++k)
        if (!rgb[k] || !depth[k]) return mfail(m, -1, "null frame pointer");      // before anything resident is touched
    const size_t dpx = depth_type == 0 ? 2 : 4;
    const size_t fr = (size_t)rows * cols * 3, fd = (size_t)rows * cols * dpx;
    const int n_pairs = n_frames - 1;
    // Not failure-atomic by construction (buffers are re-used and overwritten device by device), so the handle says "nothing
    // resident" from the first touch until EVERY device has its shard: a failed reload can never leave rgbd360_multi_align_resident
    // with the previous call's geometry over freed or half-overwritten frames.
    m->seq_frames = 0;
    auto failed = [&](int code, const char* msg) {
        multi_free_sequence(m);
        return mfail(m, code, msg);
    };
    for (int d = 0; d < m->n_gpus; ++d) {
        int lo, hi;
        shard_range(n_pairs, d, m->n_gpus, &lo, &hi);
        const int nf = hi > lo ? hi - lo + 1 : 0;
        if (hipSetDevice(m->dev[d]) != hipSuccess) return failed(-102, "hipSetDevice failed");
        if (m->seq_rgb_bytes[d] < nf * fr || m->seq_depth_bytes[d] < nf * fd) {
            hipFree(m->seq_rgb[d]); hipFree(m->seq_depth[d]);
            m->seq_rgb[d] = m->seq_depth[d] = nullptr;
            m->seq_rgb_bytes[d] = m->seq_depth_bytes[d] = 0;
            if (nf > 0 && (hipMalloc(&m->seq_rgb[d], nf * fr) != hipSuccess || hipMalloc(&m->seq_depth[d], nf * fd) != hipSuccess))
                return failed(-103, "cannot allocate the resident sequence");
            m->seq_rgb_bytes[d] = nf * fr; m->seq_depth_bytes[d] = nf * fd;
        }
        for (int k = 0; k < nf; ++k) {
            if (hipMemcpy2D(m->seq_rgb[d] + k * fr, (size_t)cols * 3, rgb[lo + k], rgb_step, (size_t)cols * 3, rows, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy2D(m->seq_depth[d] + k * fd, (size_t)cols * dpx, depth[lo + k], depth_step, (size_t)cols * dpx, rows, hipMemcpyHostToDevice) != hipSuccess)
                return failed(-104, "upload of the sequence failed");
        }
        m->seq_lo[d] = lo; m->seq_hi[d] = hi;
    }
    m->seq_rows = rows; m->seq_cols = cols; m->seq_depth_type = depth_type;
    m->seq_frames = n_frames;          // published last
    return 0;
}

int rgbd360_multi_align_resident(rgbd360_multi* m, const float guess[16], int method, int occlusion, int n_inflight, float* poses_out,
                                 rgbd360_result* results_out) {
    if (!m) return -1;
    if (m->seq_frames < 2) return mfail(m, -2, "rgbd360_multi_load_sequence was not called");
    if (!poses_out) return mfail(m, -1, "null pointer");
    const size_t dpx = m->seq_depth_type == 0 ? 2 : 4;
    const size_t fr = (size_t)m->seq_rows * m->seq_cols * 3, fd = (size_t)m->seq_rows * m->seq_cols * dpx;
    const int n_pairs = m->seq_frames - 1;
    for (int d = 0; d < m->n_gpus; ++d) {       // the shards multi_run will cut must be the ones the frames were loaded for
        int lo, hi;
        shard_range(n_pairs, d, m->n_gpus, &lo, &hi);
        if (lo != m->seq_lo[d] || hi != m->seq_hi[d]) return mfail(m, -2, "resident sequence does not match the shard layout");
        if (hi > lo && (!m->seq_rgb[d] || !m->seq_depth[d])) return mfail(m, -2, "resident sequence is incomplete");
    }
    auto frames_of = [&](int d, int k, const uint8_t** r, const void** dpt) {
        *r = m->seq_rgb[d] + (size_t)(k - m->seq_lo[d]) * fr;
        *dpt = m->seq_depth[d] + (size_t)(k - m->seq_lo[d]) * fd;
    };
    return multi_run(m, m->seq_frames, frames_of, (size_t)m->seq_cols * 3, (size_t)m->seq_cols * dpx, m->seq_depth_type, m->seq_rows,
                     m->seq_cols, guess, method, occlusion, n_inflight, true, poses_out, results_out);
}

int rgbd360_align360_batch_multi(const rgbd360_params* p, int n_frames, const uint8_t* const* rgb, size_t rgb_step,
                                 const void* const* depth, size_t depth_step, int depth_type, int rows, int cols, const float guess[16],
                                 int method, int occlusion, int n_inflight, int n_gpus, const int* device_ids, float* poses_out,
                                 rgbd360_result* results_out) {
    rgbd360_multi* m = nullptr;
    int rc = rgbd360_multi_create(p, n_gpus, device_ids, &m);
    if (rc) return rc;
    rc = rgbd360_multi_align_sequence(m, n_frames, rgb, rgb_step, depth, depth_step, depth_type, rows, cols, guess, method, occlusion,
                                      n_inflight, poses_out, results_out);
    if (rc) fprintf(stderr, "rgbd360_align360_batch_multi: %s\n", m->err.c_str());
    rgbd360_multi_destroy(m);
    return rc;
}

}  // extern "C"
