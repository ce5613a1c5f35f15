// frame360_kernels.h -- gfx950 kernels for the per-pixel Frame360 stages next to the alignment path
// (SURVEY.md rows a14 / a15): the normal map and the planar-region segmentation + inlier moments that
// Frame360::getPlanesSensor / Frame360_stereo::getPlanesStereo obtain from PCL
// (Frame360.h:949-977, Frame360_stereo.h:854-882):
//   pcl::IntegralImageNormalEstimation (AVERAGE_3D_GRADIENT, depth-dependent smoothing)
//   pcl::OrganizedMultiPlaneSegmentation (PlaneCoefficientComparator + organised connected components + PCA fit)
// PCL is third-party and unpinned (not in the reference tree); the algorithms follow PCL 1.7's published sources and
// the CPU restatement in oracle/frame360_ref.cpp documents the shared, deliberate differences (direct window sums,
// double moments, no boundary refinement, optional range-as-depth mode for full spheres).
//
//   k_f360_edge_bits  depth-change map (computeFeature) as a bit mask        1 bit/px out
//   k_f360_distmap    per-row distance to the nearest depth-change pixel + chamfer (1 / 1.4) distance map, truncated at
//                     kF360R                                                  4 B/px out
//   k_f360_normals_tiled  central differences + per-tile integral images in LDS -> window-averaged gradients -> normal
//   k_f360_ccl_rows / _merge / _compress   connected components: row runs by scan, vertical joins by union-find (atomicMin)
//   k_f360_count / _assign / _moments      region sizes, compaction of the large regions, 9 moments per region
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"      // r360::sphere_point for the fused cloud stage of k_f360_edge_bits

namespace r360 {
// Consecutive lanes own consecutive pixels and every lane stores its 12-byte point with ONE instruction (768 contiguous bytes per
// wave instruction); a thread covers four pixels 256 apart.  (Round 1's form -- a thread owning four consecutive pixels, three 16-byte
// stores with a 48-byte lane stride, 15 us at 2048 x 1024 = 1.9 TB/s -- and the one-pixel-per-thread form were deleted in round 6.)
typedef float float3s __attribute__((ext_vector_type(3)));
__global__ __launch_bounds__(256) void k_sphere_cloud_s4(const void* __restrict__ depth, size_t step, int depth_type, int rows, int cols,
                                                         int convention, const float* __restrict__ sin_theta,
                                                         const float* __restrict__ cos_theta, const float* __restrict__ sin_phi,
                                                         const float* __restrict__ cos_phi, float* __restrict__ xyz) {
    const int r = blockIdx.y;
    const uint8_t* row = (const uint8_t*)depth + (size_t)r * step;
    const float sp = sin_phi[r], cp = cos_phi[r];
    const int cbase = blockIdx.x * 1024 + threadIdx.x;
    float d[4], st[4], ct[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {          // all loads first
        const int c = cbase + 256 * k;
        const int cc = c < cols ? c : cols - 1;
        d[k] = depth_type == 0 ? 0.001f * (float)((const uint16_t*)row)[cc] : ((const float*)row)[cc];
        st[k] = sin_theta[cc];
        ct[k] = cos_theta[cc];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = cbase + 256 * k;
        if (c >= cols) continue;
        float x, y, z;
        sphere_point(convention, d[k], sp, cp, st[k], ct[k], x, y, z);
        float3s o = {x, y, z};
        *reinterpret_cast<float3s*>(xyz + 3 * ((size_t)r * cols + c)) = o;
    }
}
}  // namespace r360

namespace f360 {

constexpr int kF360R = 12;          // truncation radius of the distance map (>= smoothing_size + max depth / 10)

__device__ __forceinline__ bool finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }
// sqrtf, bit for bit: r360::sqrt_rn (5 instructions, exhaustively equal to the IEEE result on [2^-60, 2^60] and at 0) where it is
// proven, the compiler's sequence (~15 instructions with its scaling branches) elsewhere -- a branch no real cloud takes
__device__ __forceinline__ float sqrt_ieee(float x) {
    if ((x >= 0x1p-60f && x <= 0x1p60f) || x == 0.f) return r360::sqrt_rn(x);
    return sqrtf(x);
}
__device__ __forceinline__ float depth_of(const float* p, int depth_mode) {
    return depth_mode == 0 ? p[2] : sqrt_ieee(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
}

// pair test of computeFeature: `a` is the pixel the loop visits, `b` its right / lower neighbour
__device__ __forceinline__ bool depth_break(float da, float db, float factor) {
    const float ddc = factor * (fabsf(da) + 1.0f) * 2.0f;
    return (fabsf(da - db) > ddc) || !isfinite(da) || !isfinite(db);
}

// Depth-change map as a BIT mask (bit c & 63 of word [r * pitch + c / 64] = pixel (r, c) lies on a depth change; bits past
// the last column are 0).  One block owns 256 columns x 8 rows; the depths of the tile + a one-pixel ring are computed once
// into LDS (1.26 per pixel instead of 5), the 64 decisions of a wave row leave through one ballot.
// History at 2048x1024: byte map + per-row distance + chamfer map as three per-pixel kernels 19.5 + 21 + 27 us.
#ifndef F360_EDGE_TH
#define F360_EDGE_TH 8      // (round 6 sweep, 4096 x 2048 / 2048 x 1024: 4 rows 33.2 / 14.2 us, 8: 32.5 / 14.2, 12: 34.7 / 15.8, 16: 36.4 / 16.9 -- the ring's
#endif                      // share of the point arithmetic does not matter: with the cloud's 12 B/px of stores the kernel is a write stream)
constexpr int kEdgeTW = 256, kEdgeTH = F360_EDGE_TH;
// CLOUD = true (rgbd360_frame_planes): the kernel is also the sphere-cloud stage -- it forms the points of its tile + ring from the
// depth image and the angle tables (r360::sphere_point, the arithmetic of k_sphere_cloud) instead of loading them, and writes the
// tile's own points to xyz.  The cloud as a kernel of its own is a 29 MB write-dominated stream with 5 us of fixed cost (13-14 us at
// 2048 x 1024, whatever the block shape); here its 2-byte loads replace 12-byte ones and its stores ride along.
struct EdgeCloudSrc {
    const void* depth;
    size_t step;
    int depth_type, convention;
    const float *sin_theta, *cos_theta, *sin_phi, *cos_phi;
};
// SPEC >= 0 (CLOUD only): convention, depth type and depth mode as compile-time constants, SPEC = convention * 4 + depth_type * 2 +
// depth_mode -- the three are tested per row and pixel, 100 scalar branches in the generic kernel's straight-line code
template <bool CLOUD, int SPEC = -1>
__global__ __launch_bounds__(kEdgeTW) void k_f360_edge_bits(const float* __restrict__ xyz, int rows, int cols, float factor, int depth_mode_arg,
                                                           int pitch_words, unsigned long long* __restrict__ bits, EdgeCloudSrc src_arg,
                                                           float* __restrict__ xyz_out) {
    const int depth_mode = SPEC >= 0 ? (SPEC & 1) : depth_mode_arg;
    EdgeCloudSrc src = src_arg;
    if (SPEC >= 0) {
        src.depth_type = (SPEC >> 1) & 1;
        src.convention = SPEC >> 2;
    }
    __shared__ float dep[kEdgeTH + 2][kEdgeTW + 2];
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * kEdgeTW, r0 = blockIdx.y * kEdgeTH;
    {   // all loads of the thread are issued before the first depth is formed (one memory round trip per block, not eleven)
        constexpr int kN = (kEdgeTH + 2) * (kEdgeTW + 2), kTrips = (kN + kEdgeTW - 1) / kEdgeTW;
        float px[kTrips], py[kTrips], pz[kTrips];
        bool inb[kTrips];
        if (CLOUD) {
            // Row by row: thread t owns column c0 + t of the tile's ten rows (its two angle-table entries are loaded ONCE, a row's
            // sin / cos of the polar angle are wave-uniform: scalar loads), threads 0-19 own one pixel of the two ring columns
            // besides.  The flat walk over the 258 x 10 elements (eleven per thread, each with a division by 258, five loads behind
            // 64-bit addresses and all bounds tests per lane) was 150 vector instructions per output pixel.
            constexpr int kRows = kEdgeTH + 2;
            const int c = c0 + t;
            const bool c_ok = c < cols;
            const int cc = c_ok ? c : 0;
            const float st = src.sin_theta[cc], ct = src.cos_theta[cc];
            const int j_ey = t >> 1, j_c = (t & 1) ? c0 + kEdgeTW : c0 - 1;      // ring pixel of thread t < 2 kRows
            const int j_r = r0 - 1 + j_ey;
            const bool j_ok = t < 2 * kRows && j_r >= 0 && j_r < rows && j_c >= 0 && j_c < cols;
            const int jr = j_ok ? j_r : 0, jc = j_ok ? j_c : 0;
            auto load_depth = [&](int rr, int col) {
                const unsigned char* row = (const unsigned char*)src.depth + (size_t)rr * src.step;
                return src.depth_type == 0 ? 0.001f * (float)((const unsigned short*)row)[col] : ((const float*)row)[col];
            };
            float d[kRows], sp[kRows], cp[kRows];
#pragma unroll
            for (int ey = 0; ey < kRows; ++ey) {
                const int r = r0 - 1 + ey;                   // wave-uniform
                const int rr = (r >= 0 && r < rows) ? r : 0;
                d[ey] = load_depth(rr, cc);
                sp[ey] = src.sin_phi[rr]; cp[ey] = src.cos_phi[rr];
            }
            const float dj = load_depth(jr, jc), stj = src.sin_theta[jc], ctj = src.cos_theta[jc], spj = src.sin_phi[jr], cpj = src.cos_phi[jr];
#pragma unroll
            for (int ey = 0; ey < kRows; ++ey) {
                const int r = r0 - 1 + ey;
                const bool ok = c_ok && r >= 0 && r < rows;
                float x, y, z;
                r360::sphere_point(src.convention, d[ey], sp[ey], cp[ey], st, ct, x, y, z);
                if (ok && ey >= 1 && ey <= kEdgeTH) {        // the tile's own pixels (not the ring) leave as the cloud
                    float* o = xyz_out + 3 * ((size_t)r * cols + c);
                    o[0] = x; o[1] = y; o[2] = z;
                }
                const float pt[3] = {x, y, z};
                dep[ey][t + 1] = ok ? depth_of(pt, depth_mode) : 0.f;
            }
            if (t < 2 * kRows) {
                float x, y, z;
                r360::sphere_point(src.convention, dj, spj, cpj, stj, ctj, x, y, z);
                const float pt[3] = {x, y, z};
                dep[j_ey][(t & 1) ? kEdgeTW + 1 : 0] = j_ok ? depth_of(pt, depth_mode) : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kTrips; ++k) {
                const int e = t + k * kEdgeTW;
                const int ey = e / (kEdgeTW + 2), ex = e - ey * (kEdgeTW + 2);
                const int r = r0 - 1 + ey, c = c0 - 1 + ex;
                inb[k] = e < kN && r >= 0 && r < rows && c >= 0 && c < cols;
                const float* p = xyz + 3 * (inb[k] ? (size_t)r * cols + c : (size_t)0);
                px[k] = p[0]; py[k] = p[1]; pz[k] = p[2];
            }
        }
        if (!CLOUD) {
            float* flat = &dep[0][0];
#pragma unroll
            for (int k = 0; k < kTrips; ++k) {
                const int e = t + k * kEdgeTW;
                const float pt[3] = {px[k], py[k], pz[k]};
                if (e < kN) flat[e] = inb[k] ? depth_of(pt, depth_mode) : 0.f;
            }
        }
    }
    __syncthreads();
    const int c = c0 + t;
#ifdef F360_EDGE_BRANCHY
#pragma unroll
    for (int y = 0; y < kEdgeTH; ++y) {
        const int r = r0 + y;
        bool edge = false;
        if (c < cols && r < rows) {
            const float d = dep[y + 1][t + 1];
            if (r < rows - 1 && c < cols - 1) {       // visited as `index`
                edge |= depth_break(d, dep[y + 1][t + 2], factor);
                edge |= depth_break(d, dep[y + 2][t + 1], factor);
            }
            if (c >= 1 && r < rows - 1)               // right neighbour of (r, c-1)   (c-1 < cols-1 always)
                edge |= depth_break(dep[y + 1][t], d, factor);
            if (r >= 1 && c < cols - 1)               // lower neighbour of (r-1, c)
                edge |= depth_break(dep[y][t + 1], d, factor);
        }
        const unsigned long long m = __ballot(edge);
        const int w = (c0 >> 6) + (t >> 6);
        if ((t & 63) == 0 && r < rows && w < pitch_words) bits[(size_t)r * pitch_words + w] = m;
    }
#else
    // Straight-line form (round 6): a pixel's pair tests are those it makes as the visited pixel -- with its right and its lower
    // neighbour -- and those its left and upper neighbours made with it.  The lower-neighbour test of row y is the upper-neighbour
    // test of row y + 1 (carried in a register), the right-neighbour test of column c the left-neighbour test of column c + 1 (the
    // lane to the left, through the ballot word: bit l of `right` shifted up by one; lane 0 of a wave makes its own).  Every operand
    // is in the tile's LDS ring, so nothing needs a bounds branch: the image-border conditions mask the results.  (The nested form:
    // four tests per pixel behind 66 exec branches, 820 scalar instructions beside 846 vector ones.)
    const bool c_in = c < cols, c_vis = c < cols - 1;                       // column inside / visited as `index`
    const int lane = t & 63;
    // the test the row above the tile's first row made with it (row r0 - 1 visited, its lower neighbour r0)
    bool down_prev = depth_break(dep[0][t + 1], dep[1][t + 1], factor) & (r0 >= 1) & c_vis & (r0 - 1 < rows - 1);
#pragma unroll
    for (int y = 0; y < kEdgeTH; ++y) {
        const int r = r0 + y;                                             // block-uniform
        const bool r_vis = r < rows - 1;
        const float d = dep[y + 1][t + 1];
        const bool right = depth_break(d, dep[y + 1][t + 2], factor) & c_vis & r_vis;      // (r, c) visited, against (r, c + 1)
        const bool down = depth_break(d, dep[y + 2][t + 1], factor) & c_vis & r_vis;       // ... against (r + 1, c)
        // the left neighbour's `right`: pixel (r, c - 1) is visited when c - 1 < cols - 1, i.e. always for c < cols
        const bool left0 = depth_break(dep[y + 1][t], d, factor) & (c >= 1) & r_vis;       // lane 0's own (its left neighbour sits in another wave)
        const unsigned long long rm = __ballot(right);
        const unsigned long long from_left = (rm << 1) | (__ballot(left0) & 1ull);
        const bool edge = (right | down | down_prev | (((from_left >> lane) & 1ull) != 0)) & c_in & (r < rows);
        down_prev = down;
        const unsigned long long m = __ballot(edge);
        const int w = (c0 >> 6) + (t >> 6);
        if (lane == 0 && r < rows && w < pitch_words) bits[(size_t)r * pitch_words + w] = m;
    }
#endif
}

// Chamfer (1 / 1.4) distance to the nearest depth-change pixel, truncated at kF360R, from the bit mask: one block owns
// 64 columns x 32 rows.  (1) the three mask words around the tile's word column of the 56 rows within the radius go to
// LDS; (2) the per-row distance to the nearest set bit within +-kF360R comes from a 25-bit window (count-leading /
// count-trailing zeros instead of a 13-step search); (3) every pixel takes max + 0.4 min over the 25 rows around it (for
// a fixed row offset the metric grows with |dx|, so the nearest change pixel of each row suffices); the 13 x 14 possible
// values come from a table in LDS (one read + one min per candidate instead of nine VALU operations: the kernel was
// VALU-bound on the 25 candidates per pixel).
constexpr int kDistTW = 64, kDistTH = 32, kDistThreads = 256;
constexpr int kDistRows = kDistTH + 2 * kF360R;
__global__ __launch_bounds__(kDistThreads) void k_f360_distmap(const unsigned long long* __restrict__ bits, int rows, int cols, int pitch_words,
                                                              float* __restrict__ dist, unsigned* __restrict__ clear_words, int n_clear) {
    // (the normal-map sweep's claim flags + tile count, cleared here: as a memset of their own they were two 5 us fill launches)
    {
        const int g = (blockIdx.y * gridDim.x + blockIdx.x) * kDistThreads + threadIdx.x;
        if (g < n_clear) clear_words[g] = 0u;
    }
    __shared__ unsigned long long words[kDistRows][3];
    __shared__ uint8_t hd[kDistRows][kDistTW];              // 4 * min(row distance, kF360R + 1): a byte offset into a LUT row
    __shared__ float lut[kF360R + 1][16];                    // lut[|dy|][dx]: chamfer value; dx = kF360R + 1: no change pixel in that row
    const int t = threadIdx.x;
    const int w = blockIdx.x, r0 = blockIdx.y * kDistTH;
    const float far = (float)(cols + rows);
    bool any_change = false;                               // a depth-change pixel within the radius of the tile
    if (t < kDistRows * 3) {
        const int y = t / 3, k = t - 3 * y;
        const int r = r0 - kF360R + y, ww = w - 1 + k;
        const unsigned long long word = (r >= 0 && r < rows && ww >= 0 && ww < pitch_words) ? bits[(size_t)r * pitch_words + ww] : 0ull;
        words[y][k] = word;
        // the windows of the tile's columns reach kF360R bits into the neighbouring words
        const unsigned long long reach = k == 0 ? ~0ull << (64 - kF360R) : (k == 2 ? (1ull << kF360R) - 1ull : ~0ull);
        any_change = (word & reach) != 0ull;
    }
    // Most tiles of a frame see no depth change within the radius (walls, floor): every pixel of such a tile is `far` away (round 4: the
    // two LDS phases below -- 25 table reads per pixel -- were the whole cost of the kernel, 31 us at 4096 x 2048, for a constant).
    if (!__syncthreads_or(any_change ? 1 : 0)) {           // uniform
        const int c = w * 64 + (t & 63);
        constexpr int kPerE = kDistTH / (kDistThreads / 64);
#pragma unroll
        for (int j = 0; j < kPerE; ++j) {
            const int r = r0 + (t >> 6) * kPerE + j;
            if (r < rows && c < cols) dist[(size_t)r * cols + c] = far;
        }
        return;
    }
    if (t < (kF360R + 1) * 16) {
        const int ady = t >> 4, dx = t & 15;
        const int mn = dx < ady ? dx : ady, mx = dx < ady ? ady : dx;
        const float v = (float)mn * 1.4f + (float)(mx - mn);
        lut[ady][dx] = (dx <= kF360R && v < far) ? v : far;
    }
    __syncthreads();
    const int b = t & 63;
    for (int y = t >> 6; y < kDistRows; y += kDistThreads / 64) {
        const unsigned long long lo = words[y][0], mid = words[y][1], hi = words[y][2];
        const int s = 64 - kF360R + b;                      // first bit of the window in the 192-bit row lo | mid | hi
        unsigned long long x;
        if (s < 64) x = (lo >> s) | (mid << (64 - s));      // s >= 52
        else x = (s == 64) ? mid : ((mid >> (s - 64)) | (hi << (128 - s)));
        const unsigned win = (unsigned)x & ((1u << (2 * kF360R + 1)) - 1u);     // bit kF360R = the pixel itself
        const unsigned left = win & ((1u << (kF360R + 1)) - 1u), right = win >> kF360R;
        int best = kF360R + 1;
        if (left) best = kF360R - (31 - __clz((int)left));
        if (right) {
            const int dr = __ffs((int)right) - 1;
            best = dr < best ? dr : best;
        }
        hd[y][b] = (uint8_t)(best * 4);
    }
    __syncthreads();
    const int c = w * 64 + b;
    const int g = t >> 6;                                  // rows g*8 .. g*8+7 of the tile
    constexpr int kPer = kDistTH / (kDistThreads / 64);
    int col[kPer + 2 * kF360R];
#pragma unroll
    for (int k = 0; k < kPer + 2 * kF360R; ++k) col[k] = hd[g * kPer + k][b];
    const char* lut_bytes = reinterpret_cast<const char*>(&lut[0][0]);
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int r = r0 + g * kPer + j;
        float best = far;
#pragma unroll
        for (int dy = -kF360R; dy <= kF360R; ++dy) {       // rows outside the image hold "none" (their mask words are 0)
            const int ady = dy < 0 ? -dy : dy;
            const float v = *reinterpret_cast<const float*>(lut_bytes + ady * 64 + col[j + kF360R + dy]);
            best = v < best ? v : best;
        }
        if (r < rows && c < cols) dist[(size_t)r * cols + c] = best;
    }
}

// Normal map from per-tile integral images in LDS (what pcl::IntegralImageNormalEstimation does globally, in double):
// one block owns a 32 x 16 tile of normals.
//   1. the xyz tile (halo 7) is staged in LDS;
//   2. the central differences DX = p(r, c+1) - p(r, c-1), DY = p(r+1, c) - p(r-1, c) (initAverage3DGradientMethod; zero on
//      the image border) of the tile + halo 6 are formed from it ONCE (registers);
//   3. they overwrite the point tile as six float64 planes + one int32 plane of packed validity counts (non-finite differences
//      enter as 0 and are not counted), laid out with a zero row / column in front;
//   4. inclusive prefix sums along the rows, then along the columns, turn the seven planes into summed-area tables;
//   5. every pixel obtains its rect x rect window sums from four corners per plane (28 LDS reads whatever the window size).
// float64 tables keep the corner differences exact to ~1e-16 of the tile sum, like PCL's double integral images (the
// oracle sums its windows in double too): the normals are BIT-IDENTICAL to the oracle's.  History at 2048x1024: two kernels with
// per-pixel gathers through L1 270 us -> direct float32 window sums out of an LDS tile 112 us (1.8e-7 off the oracle) ->
// integral images 88 us.
constexpr int kNT_W = 32, kNT_H = 16, kNT_HALO = 6;       // 32 x 16: the tables take 71 KB, two blocks share a CU's 160 KB
constexpr int kNT_EW = kNT_W + 2 * kNT_HALO, kNT_EH = kNT_H + 2 * kNT_HALO;            // 44 x 28 entries of differences
constexpr int kNT_XW = kNT_EW + 2, kNT_XH = kNT_EH + 2;                                // 46 x 30 points (one more ring)
constexpr int kNT_SW = kNT_EW + 3, kNT_SH = kNT_EH + 1;                                // table: zero column / row in front; odd pitch 47
                                                                                       // (row scans: lane stride 47 doubles = all banks)
constexpr int kNT_SPLANE = kNT_SW * kNT_SH;                                            // 47 x 29 entries per plane
#ifndef RGBD360_NT_THREADS
#define RGBD360_NT_THREADS 1024
#endif
constexpr int kNT_THREADS = RGBD360_NT_THREADS;                                                         // 16 waves per tile (52 VGPRs: two tiles = 32 waves fit a CU): every phase is a chain of LDS round trips; 512 -> 1024 threads: 71.9 -> 69.5 us, 276 -> 266 us at 4096 x 2048 (same-box A/B);
                                                                                       // 256 threads (2 x 4 waves per CU) left them exposed: 87 -> 72 us
constexpr int kNT_BX = 11, kNT_BY = 14;                                                // scan batch lengths
static_assert(kNT_EW % kNT_BX == 0 && kNT_EH % kNT_BY == 0, "scan batches");
static_assert(kNT_HALO * 2 >= kF360R, "window offsets span -rect/2 .. rect-rect/2-1 with rect <= kF360R");
static_assert(6 * kNT_SPLANE * 8 + kNT_SPLANE * 4 <= 160 * 1024, "tables must fit the CU's LDS");
static_assert(3 * kNT_XW * kNT_XH * 4 <= 6 * kNT_SPLANE * 8, "the point tile is staged inside the table storage");

__global__ __launch_bounds__(kNT_THREADS) void k_f360_normals_tiled(const float* __restrict__ xyz, const float* __restrict__ dist,
                                                                   int rows, int cols, float smoothing_size, int depth_mode,
                                                                   float* __restrict__ normals, int* __restrict__ window,
                                                                   const unsigned* __restrict__ tile_list, int tiles_x, int tiles_total) {
    __shared__ double sat[6 * kNT_SPLANE];            // phase 1 also holds the xyz tile (3 x 46 x 30 floats)
    __shared__ int satc[kNT_SPLANE];
    float* pts = reinterpret_cast<float*>(sat);
    const int tid = threadIdx.x;
    // Behind k_f360_normals_sweep only the tiles it listed (pixels with another window size) are left to do: tile_list = {count,
    // tile ids ...}; the grid is a fixed few hundred blocks that walk the list (an empty list costs one load per block, not a wave
    // launch per tile of the frame).  Without a list (tile_list == nullptr) the blocks walk every tile.
    const int n_tiles = tile_list ? (int)min(tile_list[0], (unsigned)tiles_total) : tiles_total;
    for (int ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
    const int tile = tile_list ? (int)tile_list[1 + ti] : ti;
    const int c0 = (tile % tiles_x) * kNT_W, r0 = (tile / tiles_x) * kNT_H;
    // ---- phase 1: points of the tile + halo 7 (outside the image: NaN, never used by a pixel that produces a normal) ----
    const float qnan = __builtin_nanf("");
    {   // float by float (a tile row is 138 contiguous floats of the cloud), every load of the thread issued before the first
        // LDS store: the block waits for ONE memory round trip here, not one per sweep of the loop
        constexpr int kRowF = 3 * kNT_XW, kNF = kRowF * kNT_XH, kTrips = (kNF + kNT_THREADS - 1) / kNT_THREADS;
        float v[kTrips];
#pragma unroll
        for (int k = 0; k < kTrips; ++k) {
            const int e = tid + k * kNT_THREADS;
            const int ey = e / kRowF, ef = e - ey * kRowF;
            const int r = r0 - kNT_HALO - 1 + ey, c = c0 - kNT_HALO - 1 + ef / 3;
            const bool in = e < kNF && r >= 0 && r < rows && c >= 0 && c < cols;
            const float t = xyz[in ? 3 * ((size_t)r * cols + (c0 - kNT_HALO - 1)) + ef : (size_t)0];
            v[k] = in ? t : qnan;
        }
#pragma unroll
        for (int k = 0; k < kTrips; ++k) {
            const int e = tid + k * kNT_THREADS;
            if (e < kNF) pts[e] = v[k];
        }
    }
    __syncthreads();
    // ---- phase 2: differences of the tile + halo 6 into registers ----
    constexpr int kPer = (kNT_EW * kNT_EH + kNT_THREADS - 1) / kNT_THREADS;
    float d[kPer][6];
    int fl[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int e = tid + j * kNT_THREADS;
        float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f, gy0 = 0.f, gy1 = 0.f, gy2 = 0.f;
        int f = 257;                                  // border of the image: valid zeros for both differences
        if (e < kNT_EW * kNT_EH) {
            const int ey = e / kNT_EW, ex = e - ey * kNT_EW;
            const int r = r0 - kNT_HALO + ey, c = c0 - kNT_HALO + ex;
            if (r >= 1 && r < rows - 1 && c >= 1 && c < cols - 1) {
                const float* pc = pts + 3 * ((ey + 1) * kNT_XW + (ex + 1));
                const float *pl = pc - 3, *pr = pc + 3, *pu = pc - 3 * kNT_XW, *pd = pc + 3 * kNT_XW;
                gx0 = pr[0] - pl[0]; gx1 = pr[1] - pl[1]; gx2 = pr[2] - pl[2];
                gy0 = pd[0] - pu[0]; gy1 = pd[1] - pu[1]; gy2 = pd[2] - pu[2];
                const bool vx = finite3(gx0, gx1, gx2), vy = finite3(gy0, gy1, gy2);
                if (!vx) gx0 = gx1 = gx2 = 0.f;
                if (!vy) gy0 = gy1 = gy2 = 0.f;
                f = (vx ? 1 : 0) + (vy ? 256 : 0);
            }
        }
        d[j][0] = gx0; d[j][1] = gx1; d[j][2] = gx2; d[j][3] = gy0; d[j][4] = gy1; d[j][5] = gy2;
        fl[j] = f;
    }
    __syncthreads();
    // ---- phase 3: the planes overwrite the point tile; entry (ex, ey) lives at table position (ex + 1, ey + 1) ----
    for (int e = tid; e < kNT_SPLANE; e += kNT_THREADS) {         // zero row 0 and column 0 (and the unused pad columns)
        const int sy = e / kNT_SW, sx = e - sy * kNT_SW;
        if (sy == 0 || sx == 0 || sx > kNT_EW) {
#pragma unroll
            for (int k = 0; k < 6; ++k) sat[k * kNT_SPLANE + e] = 0.0;
            satc[e] = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int e = tid + j * kNT_THREADS;
        if (e < kNT_EW * kNT_EH) {
            const int ey = e / kNT_EW, ex = e - ey * kNT_EW;
            const int t = (ey + 1) * kNT_SW + (ex + 1);
#pragma unroll
            for (int k = 0; k < 6; ++k) sat[k * kNT_SPLANE + t] = (double)d[j][k];
            satc[t] = fl[j];
        }
    }
    __syncthreads();
    // this thread's output pixels: their point and distance-map value are fetched now, so that the global round trip runs
    // under the scans
    const int tx = tid % kNT_W, ty = tid / kNT_W;
    constexpr int kRowsPerThread = (kNT_H * kNT_W + kNT_THREADS - 1) / kNT_THREADS;      // rows ty, ty + kNT_THREADS / kNT_W, ... below kNT_H
    float ppx[kRowsPerThread], ppy[kRowsPerThread], ppz[kRowsPerThread], pdist[kRowsPerThread];
#pragma unroll
    for (int q = 0; q < kRowsPerThread; ++q) {
        const int ci = c0 + tx, ri = r0 + ty + q * (kNT_THREADS / kNT_W);
        ppx[q] = ppy[q] = ppz[q] = qnan;
        pdist[q] = 0.f;
        if (ci < cols && ri < rows && ty + q * (kNT_THREADS / kNT_W) < kNT_H) {
            const size_t index = (size_t)ri * cols + ci;
            ppx[q] = xyz[3 * index]; ppy[q] = xyz[3 * index + 1]; ppz[q] = xyz[3 * index + 2];
            pdist[q] = dist[index];
        }
    }
    // ---- phase 4: summed-area tables.  Rows: one thread per (plane, row); columns: one thread per (plane, column).  The
    //      scans run in batches (all reads of a batch issued, then the adds, then the writes): with one block per CU there
    //      is little else to hide an LDS round trip behind. ----
    for (int t = tid; t < 7 * kNT_EH; t += kNT_THREADS) {
        const int pl = t / kNT_EH, row = t - pl * kNT_EH + 1;
        if (pl < 6) {
            double* p = sat + pl * kNT_SPLANE + row * kNT_SW;
            double acc = 0.0;
            for (int x0 = 1; x0 <= kNT_EW; x0 += kNT_BX) {
                double v[kNT_BX];
#pragma unroll
                for (int k = 0; k < kNT_BX; ++k) v[k] = p[x0 + k];
#pragma unroll
                for (int k = 0; k < kNT_BX; ++k) { acc += v[k]; v[k] = acc; }
#pragma unroll
                for (int k = 0; k < kNT_BX; ++k) p[x0 + k] = v[k];
            }
        } else {
            int* p = satc + row * kNT_SW;
            int acc = 0;
            for (int x0 = 1; x0 <= kNT_EW; x0 += kNT_BX) {
                int v[kNT_BX];
#pragma unroll
                for (int k = 0; k < kNT_BX; ++k) v[k] = p[x0 + k];
#pragma unroll
                for (int k = 0; k < kNT_BX; ++k) { acc += v[k]; v[k] = acc; }
#pragma unroll
                for (int k = 0; k < kNT_BX; ++k) p[x0 + k] = v[k];
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < 7 * kNT_EW; t += kNT_THREADS) {
        const int pl = t / kNT_EW, col = t - pl * kNT_EW + 1;
        if (pl < 6) {
            double* p = sat + pl * kNT_SPLANE + col;
            double acc = 0.0;
            for (int y0 = 1; y0 <= kNT_EH; y0 += kNT_BY) {
                double v[kNT_BY];
#pragma unroll
                for (int k = 0; k < kNT_BY; ++k) v[k] = p[(y0 + k) * kNT_SW];
#pragma unroll
                for (int k = 0; k < kNT_BY; ++k) { acc += v[k]; v[k] = acc; }
#pragma unroll
                for (int k = 0; k < kNT_BY; ++k) p[(y0 + k) * kNT_SW] = v[k];
            }
        } else {
            int* p = satc + col;
            int acc = 0;
            for (int y0 = 1; y0 <= kNT_EH; y0 += kNT_BY) {
                int v[kNT_BY];
#pragma unroll
                for (int k = 0; k < kNT_BY; ++k) v[k] = p[(y0 + k) * kNT_SW];
#pragma unroll
                for (int k = 0; k < kNT_BY; ++k) { acc += v[k]; v[k] = acc; }
#pragma unroll
                for (int k = 0; k < kNT_BY; ++k) p[(y0 + k) * kNT_SW] = v[k];
            }
        }
    }
    __syncthreads();
    // ---- phase 5: four corners per plane; thread (tx, ty) handles column tx of rows ty, ty + 8 ----
    const int border = (int)smoothing_size;
#pragma unroll
    for (int q = 0; q < kRowsPerThread; ++q) {
        const int ly = ty + q * (kNT_THREADS / kNT_W);
        const int ci = c0 + tx, ri = r0 + ly;
        if (ci >= cols || ri >= rows || ly >= kNT_H) continue;
        const size_t index = (size_t)ri * cols + ci;
        float nx = qnan, ny = qnan, nz = qnan;
        int rect = 0;
        if (ri >= border && ri < rows - border && ci >= border && ci < cols - border) {
            const float px = ppx[q], py = ppy[q], pz = ppz[q];
            const float depth = depth_mode == 0 ? pz : sqrtf(px * px + py * py + pz * pz);
            if (isfinite(depth)) {
                const float smoothing = fminf(pdist[q], smoothing_size + depth / 10.0f);
                if (smoothing > 2.0f) {
                    rect = (int)smoothing;
                    // window = entries [ex0, ex0 + rect) x [ey0, ey0 + rect); table corners are one position further (zero front)
                    const int ex0 = tx + kNT_HALO - rect / 2, ey0 = ly + kNT_HALO - rect / 2;
                    const int a = ey0 * kNT_SW + ex0, b = a + rect, c = a + rect * kNT_SW, dd = c + rect;
                    double g[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double* t = sat + k * kNT_SPLANE;
                        g[k] = (t[dd] - t[b]) - (t[c] - t[a]);
                    }
                    const int icnt = (satc[dd] - satc[b]) - (satc[c] - satc[a]);
                    if ((icnt & 255) > 0 && (icnt >> 8) > 0) {
                        // gradient_y x gradient_x in double, like the sums
                        const double v0 = g[4] * g[2] - g[5] * g[1], v1 = g[5] * g[0] - g[3] * g[2], v2 = g[3] * g[1] - g[4] * g[0];
                        const double len2 = v0 * v0 + v1 * v1 + v2 * v2;
                        if (len2 != 0.0) {
                            const double inv = 1.0 / sqrt(len2);
                            nx = (float)(v0 * inv); ny = (float)(v1 * inv); nz = (float)(v2 * inv);
                            if ((-px) * nx + (-py) * ny + (-pz) * nz < 0.f) {
                                nx = -nx; ny = -ny; nz = -nz;
                            }
                        }
                    }
                }
            }
        }
        normals[3 * index] = nx; normals[3 * index + 1] = ny; normals[3 * index + 2] = nz;
        if (window) window[index] = rect;
    }
    __syncthreads();                                  // the tables are rebuilt for the block's next tile
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_f360_normals_sweep<R>: the same normal map for the pixels whose window is R x R -- almost every pixel of a frame (rect =
// int(min(distance map, smoothing_size + depth / 10)) = int(smoothing_size) away from depth edges and below 10 m) -- without any
// table: a WAVE owns a strip of 64 point columns and sweeps down the rows with everything in registers.
//   * lane l holds point column (col0 - 1 - R/2 + l); per row one coalesced 12-byte load per lane (three rows prefetched);
//   * DX(r, c) = p(r, c+1) - p(r, c-1) comes from the neighbouring LANES (DPP wave_shl / wave_shr), DY(r, c) = p(r+1, c) - p(r-1, c)
//     from the lane's own three-row window;
//   * the vertical R-sums slide: the row that enters is added, the row that left R steps ago (kept in a register ring, the loop is
//     unrolled R times so that every ring index is static) is subtracted -- in float64 like the integral images, exact for the same
//     reason (a handful of float32 terms of similar magnitude);
//   * the horizontal R-sums of the vertical sums are formed by doubling (S2 = S1 + S1<<1, S4 = S2 + S2<<2, S8 = S4 + S4<<4, then
//     the binary digits of R): the shift by one is a DPP move, the longer ones go through the LDS crossbar (ds_bpermute), which
//     runs beside the VALU -- 3 float64 adds per channel instead of R - 1 (the first version shifted R - 1 times by one lane:
//     165 of its 620 instructions per row);
//   * lane q = 1 .. 63 - R holds the sums of the window that STARTS at its column, i.e. it finishes the pixel R/2 columns to its
//     right (cross product in double, flip towards the viewpoint).
// 1.3x the pixels are read (column overlap (R+1)/64, row overlap (R+1)/segment) instead of 2.7x, nothing goes through LDS, and the
// per-pixel work is a few dozen instructions.  Pixels whose window is NOT R x R are left to k_f360_normals_tiled: the sweep marks
// their 32 x 16 tile, the tiled kernel then runs on marked tiles only (and rewrites them whole, so a tile is always the product of
// one kernel).  Exactness is that of the tiled kernel (bit-identical to the oracle up to the rare inexact float64 sum).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wshl1(float v) {      // lane l <- lane l + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}
__device__ __forceinline__ float wshr1(float v) {      // lane l <- lane l - 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ int wshl1i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ int wshr1i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }
__device__ __forceinline__ double wshl1d(double v) {
    return __hiloint2double(wshl1i(__double2hiint(v)), wshl1i(__double2loint(v)));
}
__device__ __forceinline__ double wshr1d(double v) {
    return __hiloint2double(wshr1i(__double2hiint(v)), wshr1i(__double2loint(v)));
}
// lane l <- lane l + K for K >= 2: the LDS crossbar (no LDS memory), which runs beside the VALU
template <int K>
__device__ __forceinline__ int wshlk_i(int v, int lane) {
    return K == 1 ? wshl1i(v) : __builtin_amdgcn_ds_bpermute(((lane + K) & 63) << 2, v);
}
template <int K>
__device__ __forceinline__ double wshlk_d(double v, int lane) {
    return __hiloint2double(wshlk_i<K>(__double2hiint(v), lane), wshlk_i<K>(__double2loint(v), lane));
}
// Sum of the R lanes l .. l + R - 1 by doubling: S2 = S1 + S1<<1, S4 = S2 + S2<<2, S8 = S4 + S4<<4, then the binary digits of R.
template <int R>
__device__ __forceinline__ double window_sum_d(double a, int lane) {
    const double s1 = a;
    const double s2 = s1 + wshlk_d<1>(s1, lane);
    if (R == 2) return s2;
    if (R == 3) return s2 + wshlk_d<2>(s1, lane);
    const double s4 = s2 + wshlk_d<2>(s2, lane);
    if (R == 4) return s4;
    if (R == 5) return s4 + wshlk_d<4>(s1, lane);
    if (R == 6) return s4 + wshlk_d<4>(s2, lane);
    if (R == 7) return (s4 + wshlk_d<4>(s2, lane)) + wshlk_d<6>(s1, lane);
    const double s8 = s4 + wshlk_d<4>(s4, lane);
    if (R == 8) return s8;
    if (R == 9) return s8 + wshlk_d<8>(s1, lane);
    return s8 + wshlk_d<8>(s2, lane);      // R == 10
}
template <int R>
__device__ __forceinline__ int window_sum_i(int a, int lane) {
    const int s1 = a;
    const int s2 = s1 + wshlk_i<1>(s1, lane);
    if (R == 2) return s2;
    if (R == 3) return s2 + wshlk_i<2>(s1, lane);
    const int s4 = s2 + wshlk_i<2>(s2, lane);
    if (R == 4) return s4;
    if (R == 5) return s4 + wshlk_i<4>(s1, lane);
    if (R == 6) return s4 + wshlk_i<4>(s2, lane);
    if (R == 7) return (s4 + wshlk_i<4>(s2, lane)) + wshlk_i<6>(s1, lane);
    const int s8 = s4 + wshlk_i<4>(s4, lane);
    if (R == 8) return s8;
    if (R == 9) return s8 + wshlk_i<8>(s1, lane);
    return s8 + wshlk_i<8>(s2, lane);
}
struct P3 {
    float x, y, z;
};
typedef float f3v __attribute__((ext_vector_type(3)));
typedef unsigned u3v __attribute__((ext_vector_type(3)));
constexpr int kSweepWaves = 4;
template <int R>
__global__ __launch_bounds__(64 * kSweepWaves) void k_f360_normals_sweep(const float* __restrict__ xyz, const float* __restrict__ dist, int rows,
                                                                          int cols, float smoothing_size, int depth_mode, int seg_rows,
                                                                          float* __restrict__ normals, int* __restrict__ window,
                                                                          unsigned* __restrict__ tile_flags, unsigned* __restrict__ tile_list,
                                                                          int tiles_x) {
    constexpr int OW = 63 - R;                          // output columns of a strip
    constexpr int LO = 1 + R / 2;                       // first output lane
    constexpr int kPastEnd = (int)0x80000000u;          // a buffer offset beyond any plane (the launcher keeps planes below 2 GiB)
    const int lane = threadIdx.x & 63;
    // everything about rows is wave-uniform: keep it in SGPRs (scalar branches, scalar row pointers, 32-bit lane offsets)
    const int unit = __builtin_amdgcn_readfirstlane(blockIdx.x * kSweepWaves + (threadIdx.x >> 6));
    const int strips = (cols + OW - 1) / OW;
    const int strip = unit % strips, seg = unit / strips;
    const int y0 = seg * seg_rows;
    if (y0 >= rows) return;
    const int y1 = min(rows, y0 + seg_rows);
    const int col0 = strip * OW;
    const int c = col0 - LO + lane;                     // this lane's point column
    const bool col_inner = c >= 1 && c < cols - 1;
    // Columns and rows outside the image are loaded CLAMPED, never tested: a difference is formed only where all four neighbours are
    // inside (col_inner, 1 <= e < rows - 1), and no window of a pixel that gets a normal reaches past the image (border >= R / 2).
    const int cb = 12 * min(max(c, 0), cols - 1);       // byte offset of this lane's point in a row
    // buffer addressing: descriptor + row offset in SGPRs, one 32-bit lane offset in a VGPR (no 64-bit per-lane address arithmetic);
    // the launcher guarantees rows * cols * 12 < 2^31
    const __amdgpu_buffer_rsrc_t r_xyz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xyz), 0, rows * cols * 12, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dist = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dist), 0, rows * cols * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_nrm = __builtin_amdgcn_make_buffer_rsrc(normals, 0, rows * cols * 12, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_win = __builtin_amdgcn_make_buffer_rsrc(window, 0, window ? rows * cols * 4 : 0, 0x00020000);
    const float qnan = __builtin_nanf("");
    const int e0 = y0 - R / 2;                          // first difference row the segment needs
    auto load_row = [&](int r) {
        const f3v w = (f3v)__builtin_amdgcn_raw_buffer_load_b96(r_xyz, cb, min(max(r, 0), rows - 1) * cols * 12, 0);      // row: wave-uniform
        P3 p = {w.x, w.y, w.z};
        return p;
    };
    P3 pm = load_row(e0 - 1), pc = load_row(e0), pn = load_row(e0 + 1);
    P3 qa = load_row(e0 + 2), qb = load_row(e0 + 3), qc = load_row(e0 + 4);      // in flight
    float ring[R][6];
    unsigned vbits = 0;                                  // validity of the R rows in the window: bit k = DX of ring slot k, bit 16 + k = DY
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
        for (int j = 0; j < 6; ++j) ring[k][j] = 0.f;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    int cnt = 0;                                         // count_x | count_y << 16 of the vertical window
    const int border = (int)smoothing_size;
    const int co = c + R / 2;                            // the pixel this lane finishes: its window starts at this lane's column
    const bool out_lane = lane >= 1 && lane <= OW && co < cols;
    const int co_l = min(max(co, 0), cols - 1), cob = 12 * co_l, cow = 4 * co_l;
    const bool co_inside = co >= border && co < cols - border;
    const int lane_past = out_lane ? 0 : kPastEnd;

    // row e enters the vertical sliding sums, row e - R (ring slot `slot`, a compile-time constant at every call) leaves
    auto accumulate = [&](int e, int slot) {
        // differences of row e (initAverage3DGradientMethod: zero, and valid, on the image border)
        const float lx = wshr1(pc.x), ly = wshr1(pc.y), lz = wshr1(pc.z);      // p(e, c - 1)
        const float rx = wshl1(pc.x), ry = wshl1(pc.y), rz = wshl1(pc.z);      // p(e, c + 1)
        const bool inner = col_inner && e >= 1 && e < rows - 1;
        float d[6];
        d[0] = rx - lx; d[1] = ry - ly; d[2] = rz - lz;
        d[3] = pn.x - pm.x; d[4] = pn.y - pm.y; d[5] = pn.z - pm.z;
        const bool fx = finite3(d[0], d[1], d[2]), fy = finite3(d[3], d[4], d[5]);
        const bool kx = inner && fx, ky = inner && fy;       // a difference that counts
        const bool vx = !inner || fx, vy = !inner || fy;     // "valid" (the border's zero differences are)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            d[j] = kx ? d[j] : 0.f;
            d[3 + j] = ky ? d[3 + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            acc[j] += (double)d[j];
            acc[j] -= (double)ring[slot][j];
            ring[slot][j] = d[j];
        }
        const unsigned vnew = (vx ? 1u : 0u) | (vy ? 65536u : 0u);
        cnt += (int)vnew - (int)((vbits >> slot) & 0x10001u);
        vbits = (vbits & ~(0x10001u << slot)) | (vnew << slot);
        // next point row
        pm = pc; pc = pn; pn = qa; qa = qb; qb = qc;
        qc = load_row(e + 5);
    };
    // the output pixel's own point and distance-map value, fetched one row ahead of their use
    auto load_own = [&](int y, float& ox, float& oy, float& oz, float& od) {
        const int yc = min(y, rows - 1);
        const f3v w = (f3v)__builtin_amdgcn_raw_buffer_load_b96(r_xyz, cob, yc * cols * 12, 0);
        ox = w.x; oy = w.y; oz = w.z;
        od = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_dist, cow, yc * cols * 4, 0));
    };

    // warm-up: the R - 1 rows above the first output row only enter the sums
#pragma unroll
    for (int k = 0; k < R - 1; ++k) accumulate(e0 + k, k);
    float ox, oy, oz, odist;
    load_own(y0, ox, oy, oz, odist);
    // main sweep: every row completes one output row.  Passes of R rows of straight-line code (ring slots static; the finish of
    // one row overlaps the loads and sums of the next in the scheduler's hands); rows past y1 load clamped and store nothing.
    for (int yb = y0; yb < y1; yb += R) {
        bool flagged = false;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int y = yb + k;                        // output row; the difference row that completes it is y + R - 1 - R / 2
            accumulate(y - R / 2 + R - 1, (k + R - 1) % R);
            const float px = ox, py = oy, pz = oz, pdist = odist;
            load_own(y + 1, ox, oy, oz, odist);
            // horizontal R-sums of the window starting at this lane
            double s[6];
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) s[jj] = window_sum_d<R>(acc[jj], lane);
            const int sc = window_sum_i<R>(cnt, lane);
            // straight-line finish: every lane does the arithmetic, predicates select at the end (the nested tests of the tiled
            // kernel cost a mask save / branch / three NaN moves per level here)
            // sqrtf and the division by 10, bit for bit where they can matter, in 5 + 3 instructions instead of the IEEE sequences' ~22:
            // r360::sqrt_rn equals sqrtf on [2^-60, 2^60] and the fma form below equals x / 10.0f on [2^-20, 2^60] (exhaustive:
            // rgbd360_selftest_math, tools/ubench/div10.hip).  Outside: a non-finite depth gives no normal whatever `smoothing` is; a
            // depth under 2^-20 adds under 2^-23 to a smoothing size that only counts above 2; one over 2^30 leaves `pdist` the minimum.
            const float depth = depth_mode == 0 ? pz : r360::sqrt_rn(px * px + py * py + pz * pz);
            const float q10 = depth * 0.1f;
            const float smoothing = fminf(pdist, smoothing_size + fmaf(fmaf(-q10, 10.0f, depth), 0.1f, q10));
            const bool have = co_inside & (y >= border) & (y < rows - border) & (y < y1) & isfinite(depth) & (smoothing > 2.0f);
            const int rect = have ? (int)smoothing : 0;
            const double v0 = s[4] * s[2] - s[5] * s[1], v1 = s[5] * s[0] - s[3] * s[2], v2 = s[3] * s[1] - s[4] * s[0];
            const double len2 = v0 * v0 + v1 * v1 + v2 * v2;
            // 1 / sqrt(len2): hardware estimate + two Newton steps (within an ulp or two of the quotient the tiled kernel and the checker
            // form with an IEEE sqrt and an IEEE division -- ~25 float64 instructions of the row's ~190; the normal is the product's
            // float32 rounding, which a last-bit difference of the double changes once in ~1e8 values)
            double inv = __builtin_amdgcn_rsq(len2);
            {
                const double hx = 0.5 * len2;
                inv = inv * __builtin_fma(-hx * inv, inv, 1.5);
                inv = inv * __builtin_fma(-hx * inv, inv, 1.5);
            }
            float nx = (float)(v0 * inv), ny = (float)(v1 * inv), nz = (float)(v2 * inv);
            const bool flip = (-px) * nx + (-py) * ny + (-pz) * nz < 0.f;
            const bool okn = have & (rect == R) & ((sc & 65535) > 0) & ((sc >> 16) > 0) & (len2 != 0.0);
            nx = okn ? (flip ? -nx : nx) : qnan;
            ny = okn ? (flip ? -ny : ny) : qnan;
            nz = okn ? (flip ? -nz : nz) : qnan;
            flagged = flagged | (have & (rect != R));    // another window size: the tiled kernel owns this pixel's tile
            // stores without a branch (the pass stays one basic block): a lane with nothing to write, and every lane of a row past
            // y1, offers an offset past the buffer -- the hardware range check drops the store (offsets < 2^31: no wrap-around);
            // without a window plane r_win holds no records at all
            const int past = lane_past | (y < y1 ? 0 : kPastEnd);      // (an OR, not a select: selects on the offset come back as branches)
            f3v o;
            o.x = nx; o.y = ny; o.z = nz;
            __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u3v, o), r_nrm, (y * cols * 12 + cob) | past, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)rect, r_win, (y * cols * 4 + cow) | past, 0, 0);
        }
        if (flagged && out_lane) {
            // the pass's rows lie in at most two tile rows (R <= 16); listing a tile that needs no rework only costs time.  The first
            // lane to claim a tile appends it to the list (tile_list[0] = count).
            const int ta = (yb >> 4) * tiles_x + (co >> 5), tb = ((min(yb + R, y1) - 1) >> 4) * tiles_x + (co >> 5);
            if (atomicExch(&tile_flags[ta], 1u) == 0u) tile_list[1 + atomicAdd(&tile_list[0], 1u)] = (unsigned)ta;
            if (tb != ta && atomicExch(&tile_flags[tb], 1u) == 0u) tile_list[1 + atomicAdd(&tile_list[0], 1u)] = (unsigned)tb;
        }
    }
}

// ---- organised connected components with PlaneCoefficientComparator -------------------------------------------
// PlaneCoefficientComparator::compare(i, j) as the segmentation calls it: i = the pixel visited, j = its left / upper
// neighbour; the distance threshold scales with the depth of i.  plane_d = p . n in the reference's float order.
__device__ __forceinline__ float plane_dot(const float* __restrict__ xyz, const float* __restrict__ normals, int i) {
    const float *p = xyz + 3 * (size_t)i, *q = normals + 3 * (size_t)i;
    return p[0] * q[0] + p[1] * q[1] + p[2] * q[2];
}
__device__ __forceinline__ bool plane_link(const float* __restrict__ xyz, const float* __restrict__ normals, int i, int j, float cos_thr,
                                           float dist_thr, int depth_mode) {
    const float *p = xyz + 3 * (size_t)i, *q = xyz + 3 * (size_t)j;
    if (!finite3(p[0], p[1], p[2]) || !finite3(q[0], q[1], q[2])) return false;
    const float* n = normals + 3 * (size_t)i;
    const float* m = normals + 3 * (size_t)j;
    const float z = depth_of(p, depth_mode);
    const float thr = dist_thr * z * z;                          // depth-dependent distance threshold
    const float dot = n[0] * m[0] + n[1] * m[1] + n[2] * m[2];
    return (fabsf(plane_dot(xyz, normals, i) - plane_dot(xyz, normals, j)) < thr) && (dot > cos_thr);
}

// Pass 1, per pixel: the two comparisons the segmentation makes at pixel i (against its left and its upper neighbour) as
// flag bits -- bit 0: the point is finite, bit 1: linked to the left neighbour, bit 2: linked to the upper neighbour.
// One block owns 256 columns x 4 rows; the per-pixel plane record (normal, p . n, distance threshold) of the tile + the
// row above + the column to the left is formed once in LDS (1.25 per pixel; the per-pixel kernels before it evaluated
// plane_link's 2 x 24 B gathers three to five times per pixel), all loads of a thread are issued up front.
constexpr int kLinkTW = 256, kLinkTH = 4;
__global__ __launch_bounds__(kLinkTW) void k_f360_link_flags(const float* __restrict__ xyz, const float* __restrict__ normals, int rows,
                                                            int cols, float cos_thr, float dist_thr, int depth_mode,
                                                            uint8_t* __restrict__ flags) {
    __shared__ float4 rec[kLinkTH + 1][kLinkTW + 1];       // normal.xyz, p . n
    __shared__ float thr[kLinkTH + 1][kLinkTW + 1];        // dist_thr * depth^2 (the threshold when this pixel is the visiting one)
    __shared__ uint8_t ok[kLinkTH + 1][kLinkTW + 1];       // finite point
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * kLinkTW, r0 = blockIdx.y * kLinkTH;
    {
        constexpr int kN = (kLinkTH + 1) * (kLinkTW + 1), kTrips = (kN + kLinkTW - 1) / kLinkTW;
        float p[kTrips][3], q[kTrips][3];
        bool inb[kTrips];
#pragma unroll
        for (int k = 0; k < kTrips; ++k) {
            const int e = t + k * kLinkTW;
            const int ey = e / (kLinkTW + 1), ex = e - ey * (kLinkTW + 1);
            const int r = r0 - 1 + ey, c = c0 - 1 + ex;
            inb[k] = e < kN && r >= 0 && r < rows && c >= 0 && c < cols;
            const size_t i = inb[k] ? (size_t)r * cols + c : (size_t)0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                p[k][d] = xyz[3 * i + d];
                q[k][d] = normals[3 * i + d];
            }
        }
#pragma unroll
        for (int k = 0; k < kTrips; ++k) {
            const int e = t + k * kLinkTW;
            if (e < kN) {
                const int ey = e / (kLinkTW + 1), ex = e - ey * (kLinkTW + 1);
                const bool fin = inb[k] && finite3(p[k][0], p[k][1], p[k][2]);
                const float z = depth_of(p[k], depth_mode);
                rec[ey][ex] = make_float4(q[k][0], q[k][1], q[k][2], p[k][0] * q[k][0] + p[k][1] * q[k][1] + p[k][2] * q[k][2]);
                thr[ey][ex] = dist_thr * z * z;
                ok[ey][ex] = fin ? 1 : 0;
            }
        }
    }
    __syncthreads();
    const int c = c0 + t;
    if (c >= cols) return;
#pragma unroll
    for (int y = 0; y < kLinkTH; ++y) {
        const int r = r0 + y;
        if (r >= rows) break;
        const float4 me = rec[y + 1][t + 1];
        const float th = thr[y + 1][t + 1];
        const bool fin = ok[y + 1][t + 1] != 0;
        auto linked = [&](int yy, int xx) {
            const float4 o = rec[yy][xx];
            const float dot = me.x * o.x + me.y * o.y + me.z * o.z;
            return fin && ok[yy][xx] != 0 && (fabsf(me.w - o.w) < th) && (dot > cos_thr);
        };
        const bool left = c > 0 && linked(y + 1, t), up = r > 0 && linked(y, t + 1);
        flags[(size_t)r * cols + c] = (uint8_t)((fin ? 1 : 0) | (left ? 2 : 0) | (up ? 4 : 0));
    }
}

// Pass 2, one wave per image row, flags only: every pixel is labelled with the first pixel of its horizontal run of linked
// pixels (an inclusive max-scan of "column where a run starts"; a lane owns 4 consecutive pixels), so pass 3 only has to
// join runs vertically.  Invalid pixels get -1.
constexpr int kRunRowsPerBlock = 4;
// Inclusive max-scan over the 64 lanes of a wave for values >= -1, on the VALU's data-parallel-primitive paths: shifts by 1, 2, 4, 8 inside
// the rows of 16 lanes, then lane 15 of a row to the next row and lane 31 to the upper half (lanes without a source take -1).  Twelve vector
// instructions; six __shfl_up steps are six trips through the LDS crossbar (~100 cycles each for a wave that has the SIMD to itself).
__device__ __forceinline__ int wave_scan_max(int x) {
    int t;
    t = __builtin_amdgcn_update_dpp(-1, x, 0x111, 0xF, 0xF, false); x = t > x ? t : x;      // row_shr:1
    t = __builtin_amdgcn_update_dpp(-1, x, 0x112, 0xF, 0xF, false); x = t > x ? t : x;      // row_shr:2
    t = __builtin_amdgcn_update_dpp(-1, x, 0x114, 0xF, 0xF, false); x = t > x ? t : x;      // row_shr:4
    t = __builtin_amdgcn_update_dpp(-1, x, 0x118, 0xF, 0xF, false); x = t > x ? t : x;      // row_shr:8
    t = __builtin_amdgcn_update_dpp(-1, x, 0x142, 0xA, 0xF, false); x = t > x ? t : x;      // row_bcast:15 -> rows 1, 3
    t = __builtin_amdgcn_update_dpp(-1, x, 0x143, 0xC, 0xF, false); x = t > x ? t : x;      // row_bcast:31 -> rows 2, 3
    return x;
}
// inclusive add-scan over the wave for small non-negative counts, same paths (lanes without a source add 0)
__device__ __forceinline__ int wave_scan_add(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
    return x;
}
// starts / nstarts (optional): the row's run starts as a compact list (row r's k-th start at starts[r cols + k]) for k_f360_ccl_roots_list
__global__ __launch_bounds__(64 * kRunRowsPerBlock) void k_f360_ccl_runs(const uint8_t* __restrict__ flags, int rows, int cols,
                                                                        int* __restrict__ label, int* __restrict__ starts,
                                                                        int* __restrict__ nstarts) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * kRunRowsPerBlock + (threadIdx.x >> 6);
    if (r >= rows) return;                                   // whole waves leave; no block-level synchronisation below
    const uint8_t* f = flags + (size_t)r * cols;
    int* L = label + (size_t)r * cols;
    int carry = -1;
    int n_starts = 0;                                        // wave-uniform
    // The flags of eight 256-pixel steps are requested before the first step is worked on: one step after the other, every step
    // waited for its own four bytes -- eight dependent memory round trips per 2048-pixel row, most of the kernel's 9.4 us.
    constexpr int kBatch = 8;
    const bool words = ((reinterpret_cast<size_t>(f) | (size_t)cols) & 3) == 0;      // wave-uniform: rows of whole, aligned dwords
    for (int cb = 0; cb < cols; cb += 256 * kBatch) {
        unsigned w[kBatch];
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
            const int c = cb + 256 * q + 4 * lane;
            w[q] = 0u;
            if (words) {
                if (c < cols) w[q] = *reinterpret_cast<const unsigned*>(f + c);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c + k < cols) w[q] |= (unsigned)f[c + k] << (8 * k);
            }
        }
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
            const int c0 = cb + 256 * q;
            if (c0 >= cols) break;                           // wave-uniform
            const int c = c0 + 4 * lane;
            int v[4];
            bool valid[4];
            int m = -1;
            unsigned root_bits = 0;                          // pixels that can be roots: valid and not linked to the left
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cc = c + k;
                const int fl = cc < cols ? (int)((w[q] >> (8 * k)) & 0xffu) : 0;
                valid[k] = (fl & 1) != 0;
                if (cc < cols && !(fl & 2)) m = cc;          // a run starts here
                if (cc < cols && (fl & 1) && !(fl & 2)) root_bits |= 1u << k;
                v[k] = m;
            }
            if (starts) {                                    // wave-uniform
                const int mine = __builtin_popcount(root_bits);
                const int incl = wave_scan_add(mine);
                int at = r * cols + n_starts + incl - mine;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (root_bits & (1u << k)) starts[at++] = r * cols + c + k;
                n_starts += __builtin_amdgcn_readlane(incl, 63);
            }
            const int s = wave_scan_max(m);
            int prev = __builtin_amdgcn_update_dpp(-1, s, 0x138, 0xF, 0xF, false);      // wave_shr:1 (lane 0: -1)
            prev = prev > carry ? prev : carry;
            int lab[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int start = v[k] > prev ? v[k] : prev;
                lab[k] = valid[k] ? r * cols + start : -1;
            }
            if (words) {                                     // rows of whole dwords: a lane's four labels leave as ONE 16-byte store (round 4:
                if (c < cols) *reinterpret_cast<int4*>(L + c) = make_int4(lab[0], lab[1], lab[2], lab[3]);      // four dword stores at a 16-byte stride each filled a quarter of every line they touched: 21 us at 4096 x 2048)
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c + k < cols) L[c + k] = lab[k];
            }
            const int last = __builtin_amdgcn_readlane(s, 63);
            carry = last > carry ? last : carry;
        }
    }
    if (nstarts && lane == 0) nstarts[r] = n_starts;
}

// The same with FOUR waves per row (round 4; rows of whole dwords, 1024 <= cols <= 8192): a wave owns a quarter of the columns, holds
// its flags in registers, and first reports the column of its last run start and its count of run starts; behind one barrier every
// wave knows the run that reaches into its segment (the latest start to its left) and where its run starts go in the row's list.
// One wave per row left 2048 waves on 1024 SIMDs at 4096 x 2048: two memory round trips and a serial scan per row, 15-21 us.
constexpr int kRunSegs = 4, kRunSegSteps = 8;
__global__ __launch_bounds__(64 * kRunSegs) void k_f360_ccl_runs_seg(const uint8_t* __restrict__ flags, int rows, int cols,
                                                                   int* __restrict__ label, int* __restrict__ starts,
                                                                   int* __restrict__ nstarts) {
    __shared__ int seg_last[kRunSegs], seg_roots[kRunSegs];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6, r = blockIdx.x;
    const uint8_t* f = flags + (size_t)r * cols;
    int* L = label + (size_t)r * cols;
    const int seg_px = ((cols + kRunSegs * 256 - 1) / (kRunSegs * 256)) * 256;      // <= 256 kRunSegSteps (the launcher checks)
    const int c_begin = seg * seg_px, c_end = min(cols, c_begin + seg_px);
    unsigned w[kRunSegSteps];
#pragma unroll
    for (int q = 0; q < kRunSegSteps; ++q) {
        const int c = c_begin + 256 * q + 4 * lane;
        w[q] = c < c_end ? *reinterpret_cast<const unsigned*>(f + c) : 0u;          // (c_end is a multiple of 4)
    }
    {
        int m_last = -1, roots = 0;
#pragma unroll
        for (int q = 0; q < kRunSegSteps; ++q) {
            const int c = c_begin + 256 * q + 4 * lane;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int fl = (int)((w[q] >> (8 * k)) & 0xffu);
                if (c + k < c_end && !(fl & 2)) {
                    m_last = c + k;
                    roots += fl & 1;
                }
            }
        }
        const int last = __builtin_amdgcn_readlane(wave_scan_max(m_last), 63), total = __builtin_amdgcn_readlane(wave_scan_add(roots), 63);
        if (lane == 0) {
            seg_last[seg] = last;
            seg_roots[seg] = total;
        }
    }
    __syncthreads();
    int carry = -1, n_starts = 0;
#pragma unroll
    for (int sg = 0; sg < kRunSegs; ++sg)
        if (sg < seg) {                                      // wave-uniform
            carry = max(carry, seg_last[sg]);
            n_starts += seg_roots[sg];
        }
#pragma unroll
    for (int q = 0; q < kRunSegSteps; ++q) {
        const int c0 = c_begin + 256 * q;
        if (c0 >= c_end) break;                              // wave-uniform
        const int c = c0 + 4 * lane;
        int v[4];
        bool valid[4];
        int m = -1;
        unsigned root_bits = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cc = c + k;
            const int fl = cc < c_end ? (int)((w[q] >> (8 * k)) & 0xffu) : 0;
            valid[k] = (fl & 1) != 0;
            if (cc < c_end && !(fl & 2)) m = cc;
            if (cc < c_end && (fl & 1) && !(fl & 2)) root_bits |= 1u << k;
            v[k] = m;
        }
        if (starts) {                                        // uniform
            const int mine = __builtin_popcount(root_bits);
            const int incl = wave_scan_add(mine);
            int at = r * cols + n_starts + incl - mine;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (root_bits & (1u << k)) starts[at++] = r * cols + c + k;
            n_starts += __builtin_amdgcn_readlane(incl, 63);
        }
        const int sc = wave_scan_max(m);
        int prev = __builtin_amdgcn_update_dpp(-1, sc, 0x138, 0xF, 0xF, false);      // wave_shr:1 (lane 0: -1)
        prev = prev > carry ? prev : carry;
        int lab[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int start = v[k] > prev ? v[k] : prev;
            lab[k] = valid[k] ? r * cols + start : -1;
        }
        if (c < c_end) *reinterpret_cast<int4*>(L + c) = make_int4(lab[0], lab[1], lab[2], lab[3]);
        const int last = __builtin_amdgcn_readlane(sc, 63);
        carry = last > carry ? last : carry;
    }
    if (nstarts && seg == kRunSegs - 1 && lane == 0) nstarts[r] = n_starts;
}

// Parents always have the smaller index (uf_union hangs the larger root under the smaller), so a cell only ever decreases
// and any ancestor is a valid parent: the walk re-points every node it passes at its grandparent (path splitting, by
// atomicMin so that it commutes with the unions).  Without it a wall's runs form a chain as long as the wall is tall.
#ifdef F360_DEBUG_COUNTERS
__device__ unsigned long long g_dbg[8];
#endif
__device__ __forceinline__ int uf_find(int* label, int x) {
    int p = label[x];
#ifdef F360_DEBUG_COUNTERS
    unsigned long long hops = 0;
#endif
    while (p != x) {
#ifdef F360_DEBUG_COUNTERS
        atomicAdd(&g_dbg[1], 1ull);
        atomicMax(&g_dbg[3], ++hops);
#endif
        const int g = label[p];
        if (g != p) atomicMin(&label[x], g);
        x = p;
        p = g;
    }
    return x;
}
__device__ __forceinline__ void uf_union(int* label, int a, int b) {
    for (;;) {
        // both walks advance together: the two loads of a step are in flight at the same time (a join is a chain of dependent
        // L2 round trips and nothing else)
        int pa = label[a], pb = label[b];
        while (pa != a || pb != b) {
            int ga = pa, gb = pb;
            if (pa != a) ga = label[pa];
            if (pb != b) gb = label[pb];
            if (pa != a && ga != pa) atomicMin(&label[a], ga);
            if (pb != b && gb != pb) atomicMin(&label[b], gb);
            a = pa; pa = ga;
            b = pb; pb = gb;
        }
        if (a == b) return;
        if (a < b) {
            const int t = a; a = b; b = t;
        }
        const int old = atomicMin(&label[a], b);     // a > b: hang the larger root under the smaller
#ifdef F360_DEBUG_COUNTERS
        atomicAdd(&g_dbg[2], 1ull);
#endif
        if (old == a) return;
        a = old;                                     // somebody re-rooted `a` meanwhile: retry from there
    }
}

// Pass 3: join a pixel's run with the run of its upper neighbour (flags only).  The join is skipped where the pixel to the
// left already made it (same two runs), which leaves about one union per pair of overlapping runs.
//
// The joins are made bottom-up in a binary hierarchy over the rows: the boundary above row r belongs to level ctz(r), and a
// level only starts when the levels below it are complete, so a join always connects two finished blocks of 2^level rows.
// Launched all at once instead, a wall's thousand runs hook onto each other in one step (run r under run r-1 under ...),
// and whoever comes late walks the whole chain one dependent L2 round trip at a time: 8 k joins cost 70-90 us at
// 2048x1024 through a single 75-86 hop walk.  Levels 0..5 (64 x 64 tiles) share one launch, the sparse rest (every 64th row) one more.
// fl: flags of (r, c); fl_up: of (r-1, c); fl_left: of (r, c-1)
__device__ __forceinline__ void ccl_join_up(int cols, int* __restrict__ label, int r, int c, int fl, int fl_up, int fl_left) {
    if (!(fl & 4)) return;
    if (c > 0 && (fl & 2) && (fl_up & 2) && (fl_left & 4)) return;
    const int i = r * cols + c;
    uf_union(label, i, i - cols);
}

#ifndef F360_BAND_ROWS
#define F360_BAND_ROWS 64
#endif
// (round 4: 64 x 64 tiles, six levels per launch -- until then 16-row bands of 256 columns and levels 4 and 5 as launches of their own,
// 5.6-5.8 us each for a few thousand joins; sixteen row groups keep a thread's joins per level at two)
constexpr int kBandRows = F360_BAND_ROWS, kBandLevels = kBandRows == 64 ? 6 : 4, kBandCols = kBandRows == 64 ? 64 : 256;
// row groups per tile (threads = kBandCols x groups): sixteen keep a thread's joins per level at two; with 2048 tiles and more
// (4096 x 2048) eight are faster -- twice the tiles resident, and the launch is a chain of dependent joins per tile: 41 -> 35 us there,
// but 17 -> 20 us at 2048 x 1024 and 30 -> 34 us on its fragmented frame, so the launcher picks by the tile count
constexpr int kBandGroupsMax = 1024 / kBandCols;
template <int kBandGroups>
__global__ __launch_bounds__(kBandCols * kBandGroups) void k_f360_ccl_merge_band(const uint8_t* __restrict__ flags, int rows, int cols,
                                                                                int* __restrict__ label) {
    __shared__ uint8_t tile[kBandRows][kBandCols + 4];        // flags of the band; column 0 = the column left of the block
    const int tx = threadIdx.x & (kBandCols - 1), g = threadIdx.x / kBandCols;
    const int c0 = blockIdx.x * kBandCols, r0 = blockIdx.y * kBandRows;
    for (int e = threadIdx.x; e < kBandRows * (kBandCols + 1); e += kBandCols * kBandGroups) {
        const int y = e / (kBandCols + 1), x = e - y * (kBandCols + 1);
        const int r = r0 + y, c = c0 - 1 + x;
        tile[y][x] = (r < rows && c >= 0 && c < cols) ? flags[(size_t)r * cols + c] : 0;
    }
    __syncthreads();
    const int c = c0 + tx;
#pragma unroll
    for (int level = 0; level < kBandLevels; ++level) {
        // the rows of this level: y = 2^level (2 m + 1); row group g takes every kBandGroups-th of them
        for (int m = g; (1 << level) * (2 * m + 1) < kBandRows; m += kBandGroups) {
            const int y = (1 << level) * (2 * m + 1);
            if (c < cols && r0 + y < rows) ccl_join_up(cols, label, r0 + y, c, tile[y][tx + 1], tile[y - 1][tx + 1], tile[y][tx]);
        }
        __syncthreads();        // the block's own joins of this level are issued before the next level starts (other column
                                // blocks of the band may run ahead: that costs depth, not correctness)
    }
}

// one level >= kBandLevels: rows r = 2^level * (2 m + 1); all_above: every multiple of 2^level (the sparse top of the hierarchy
// in one launch: at most rows / 2^level boundaries hook at once)
__global__ void k_f360_ccl_merge_level(const uint8_t* __restrict__ flags, int rows, int cols, int* __restrict__ label, int level, int all_above) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = all_above ? (1 << level) * ((int)blockIdx.y + 1) : (1 << level) * (2 * (int)blockIdx.y + 1);
    if (c >= cols || r >= rows) return;
    const size_t i = (size_t)r * cols + c;
    const int fl = flags[i];
    if (!(fl & 4)) return;
    ccl_join_up(cols, label, r, c, fl, flags[i - cols], c > 0 ? flags[i - 1] : 0);
}

// Pass 4: only the first pixel of a run can be (or become) a root, and every other pixel still points at the first pixel of
// its run: the pointer chase runs over the run starts only (a few per cent of the pixels); their region counters are
// cleared on the way (no memset of the 8 B/px table).
// Over the compact lists k_f360_ccl_runs left: one thread per run start, a block per image row.  (One thread per PIXEL reading
// its flag byte made 32 k waves whose only load was a 64-byte line, and the chases -- up to nine dependent hops -- started behind it:
// 17-20 us at 2048 x 1024.)
constexpr int kRootsThreads = 1024;     // (256 until round 4: a border row without normals is one run start per pixel, i.e. cols / 256 dependent rounds of pointer chases)
__global__ __launch_bounds__(kRootsThreads) void k_f360_ccl_roots_list(const int* __restrict__ starts, const int* __restrict__ nstarts, int cols,
                                                                         int* __restrict__ label, unsigned long long* __restrict__ count) {
    const int r = blockIdx.x;
    const int ns = nstarts[r];
    for (int k = threadIdx.x; k < ns; k += kRootsThreads) {
        const int i = starts[(size_t)r * cols + k];
#ifdef F360_ROOTS_WALK
        label[i] = uf_find(label, i);
#else
        // Pointer jumping (round 4): every node of the forest is a run start, i.e. has a thread in this launch, and no union runs any
        // more -- so a thread keeps re-pointing ITS OWN node at its grandparent (device-scope loads: another XCD's compression is only
        // useful if it is seen) and the reach of every pointer doubles per trip: log2(depth) trips of two loads instead of a hop per
        // level (the fragmented frame: chains of 30-60 hops, 19-45 us for this launch).  Any value read is an ancestor, so the
        // result does not depend on the interleaving.
        int p = __hip_atomic_load(&label[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            const int g = __hip_atomic_load(&label[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g == p) break;                               // p is a root
            __hip_atomic_store(&label[i], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p = g;
        }
        label[i] = p;
#endif
        count[i] = 0ull;
    }
}

// ---- region sizes and moments: block-aggregated, integer, order-independent ------------------------------------------
// A planar wall owns hundreds of thousands of pixels, so adding per pixel (or even per wave) into one global counter
// serialises on that address.  Each 1024-thread block therefore sweeps 4096 consecutive pixels, aggregates per label
// in a small LDS hash (wave-uniform labels -- the common case -- are first reduced inside the wave), and only then
// issues one global atomic per (block, label, value).  All sums are integers (counts; moments in 2^-28 fixed point):
// integer addition is associative, so the results are bitwise reproducible whatever the arrival order.
constexpr int kAggThreads = 1024;
constexpr int kMomPerThread = 8;
// 2^28 units per m (linear terms) / per m^2 (quadratic terms).  2^24 until round 6: every term is rounded to the unit once, and in
// C = sum(x x) / N - (sum(x) / N)^2 that 6e-8 m^2 stands against a smallest eigenvalue of ~1e-7 m^2 for a region of a dozen pixels --
// curvatures 2 % off the float64 checker's, planes on the other side of max_curvature (tests/tools/planes_soak.py: 2 of 5 k planes).
// The sums stay inside 64 bits while N r^2 < 2^63 / 2^28 = 3.4e10 m^2: a full 4096 x 2048 frame that is ONE region at 64 m, the whole
// range of a 16-bit millimetre image; the host checks the decoded sums (f360_planes_dev) and refuses a frame beyond that.
constexpr double kMomScale = 268435456.0;
constexpr int kMomReplicas = 16;              // copies of the global moment table (block b adds into copy b % 16): a wall is hit by
                                              // every block it spans, and same-address global atomics serialise; the host adds the copies

// sum over the wave, the same in every lane: an add-scan on the DPP paths (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31) leaves the total in lane 63.
// (Six __shfl_xor steps of a 64-bit value are twelve trips through the LDS crossbar; the moment kernel sums nine values per wave.)
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#define F360_SUM_STEP(ctrl_, rows_)                                                                                    \
    {                                                                                                                  \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(unsigned long long)v, ctrl_, rows_, 0xF, false);          \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)v >> 32), ctrl_, rows_, 0xF, false);  \
        v += (long long)(((unsigned long long)hi << 32) | lo);                                                         \
    }
    F360_SUM_STEP(0x111, 0xF)
    F360_SUM_STEP(0x112, 0xF)
    F360_SUM_STEP(0x114, 0xF)
    F360_SUM_STEP(0x118, 0xF)
    F360_SUM_STEP(0x142, 0xA)
    F360_SUM_STEP(0x143, 0xC)
#undef F360_SUM_STEP
    const unsigned tlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, 63);
    const unsigned thi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), 63);
    return (long long)(((unsigned long long)thi << 32) | tlo);
}

// Pass 5: every remaining pixel takes the root of its run start (exactly one hop), and the region sizes are counted on the
// way: count[root] = number of pixels of the region (unsigned 64-bit).  A block sweeps kCntPerThread x 1024 consecutive
// pixels; wave-uniform labels (the common case) are accumulated in a wave-scalar (key, count) pair that is only flushed to
// the block's LDS hash when the key changes, the hash goes to global memory once per (block, label).
// (round 6 sweep at 4096 x 2048 / 2048 x 1024, tools/f360_ab.sh: 1024 threads x 8 pixels 26.2 / 10.7 us; x 4: 39.0 / 12.1; x 12: 31.8 / 13.9;
// x 16: 32.1 / 17.0; x 32: 33.8 / 29.8; 512 threads x 8: 34.6 / 12.9; 256 x 8: 59.3 / 19.0; 256 x 16: 40.8 / 16.4 -- fewer pixels per block
// mean more flushes into the same few counters, more mean a longer chain per block with two blocks resident per CU)
#ifndef F360_CNT_PER_THREAD
#define F360_CNT_PER_THREAD 8
#endif
#ifndef F360_CNT_THREADS
#define F360_CNT_THREADS 1024
#endif
constexpr int kCntPerThread = F360_CNT_PER_THREAD, kCntThreads = F360_CNT_THREADS;
constexpr int kCntHashBits = 8, kCntHash = 1 << kCntHashBits;
__device__ __forceinline__ int cnt_slot(int* keys, int key) {
    int h = (int)(((unsigned)key * 2654435761u) >> (32 - kCntHashBits));
    for (int probe = 0; probe < 8; ++probe) {           // bounded: where every pixel is its own region (no normal) the table is
                                                        // full at once and the adds go straight to their own global counters
        const int old = atomicCAS(&keys[h], -1, key);
        if (old == -1 || old == key) return h;
        h = (h + 1) & (kCntHash - 1);
    }
    return -1;
}
__global__ __launch_bounds__(kCntThreads) void k_f360_finish_count(const uint8_t* __restrict__ flags, int n, int* __restrict__ label,
                                                                  unsigned long long* __restrict__ count, int* __restrict__ n_slots) {
    __shared__ int keys[kCntHash];
    __shared__ unsigned int vals[kCntHash];
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_slots = 0;        // the slot counter of k_f360_assign, the next launch (no memset)
    if (threadIdx.x < kCntHash) {
        keys[threadIdx.x] = -1;
        vals[threadIdx.x] = 0u;
    }
    __syncthreads();
    auto add = [&](int key, unsigned int cnt) {
        const int e = cnt_slot(keys, key);
        if (e >= 0) atomicAdd(&vals[e], cnt);
        else atomicAdd(&count[key], (unsigned long long)cnt);
    };
    const int base = blockIdx.x * kCntThreads * kCntPerThread;
    const bool lane0 = (threadIdx.x & 63) == 0;
    // two batched rounds of loads (flags and labels together -- a label is read whether or not its flag says it means anything -- then the
    // run starts' labels) instead of a dependent triple per pixel
    int fl[kCntPerThread], lab[kCntPerThread];
#pragma unroll
    for (int j = 0; j < kCntPerThread; ++j) {
        const int i = base + j * kCntThreads + (int)threadIdx.x;
        fl[j] = i < n ? flags[i] : 0;
        lab[j] = label[i < n ? i : n - 1];
    }
#pragma unroll
    for (int j = 0; j < kCntPerThread; ++j) lab[j] = (fl[j] & 1) ? lab[j] : -1;
#pragma unroll
    for (int j = 0; j < kCntPerThread; ++j)
        if ((fl[j] & 3) == 3) lab[j] = label[lab[j]];      // not a run start: its label is its run start, whose label is the root by now
#pragma unroll
    for (int j = 0; j < kCntPerThread; ++j) {
        const int i = base + j * kCntThreads + (int)threadIdx.x;
        if ((fl[j] & 3) == 3) label[i] = lab[j];
    }
    int pend_key = -1;
    unsigned int pend_cnt = 0;
#pragma unroll
    for (int j = 0; j < kCntPerThread; ++j) {
        const int l = lab[j];
        const int first = __builtin_amdgcn_readfirstlane(l);
        const unsigned long long same = __ballot(l == first);
        if (same == __ballot(true)) {
            if (first >= 0) {
                const unsigned int c64 = (unsigned int)__popcll(same);
                if (first == pend_key) pend_cnt += c64;
                else {
                    if (pend_cnt && lane0) add(pend_key, pend_cnt);
                    pend_key = first;
                    pend_cnt = c64;
                }
            }
        } else {
            // several regions among the wave's 64 consecutive pixels: one add per RUN of equal labels (its first lane adds the
            // run's length, found from the ballot of run starts) instead of one LDS atomic per pixel
            const int lane = (int)threadIdx.x & 63;
            const int prev = __shfl_up(l, 1);
            const bool head = lane == 0 || prev != l;
            const unsigned long long heads = __ballot(head);
            if (head && l >= 0) {
                const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
                add(l, (unsigned int)(above ? __ffsll((long long)above) : 64 - lane));
            }
        }
    }
    if (pend_cnt && lane0) add(pend_key, pend_cnt);
    __syncthreads();
    if (threadIdx.x < kCntHash && keys[threadIdx.x] >= 0) atomicAdd(&count[keys[threadIdx.x]], (unsigned long long)vals[threadIdx.x]);
}

// compaction: roots of regions with more than min_inliers points get a slot (order fixed later on the host).
// Over the compact run-start lists of k_f360_ccl_runs (a root is a run start): one thread per run start, a block per image row,
// instead of one thread per PIXEL reading its label to find out that it is not a root (14.5 us at 4096 x 2048 for ~10 k roots).
constexpr int kAssignThreads = 1024;      // a border row without normals is one run start per pixel: its list is walked in cols / 1024 dependent rounds
__global__ __launch_bounds__(kAssignThreads) void k_f360_assign_list(const int* __restrict__ starts, const int* __restrict__ nstarts, int cols,
                                                                      const int* __restrict__ label, const unsigned long long* __restrict__ count,
                                                                      int min_inliers, int max_slots, int* __restrict__ slot_of_root,
                                                                      int* __restrict__ root_of_slot, int* __restrict__ count_of_slot,
                                                                      int* __restrict__ n_slots, unsigned long long* __restrict__ mom, int mom_replicas) {
    const int r = blockIdx.x;
    const int ns = nstarts[r];
    const int lane = threadIdx.x & 63;
    for (int k0 = 0; k0 < ns; k0 += kAssignThreads) {            // uniform trip count: the waves clear the new slots' moment rows together
        const int k = k0 + (int)threadIdx.x;
        int slot = -1, i = -1;
        if (k < ns) {
            i = starts[(size_t)r * cols + k];
            if (label[i] != i) i = -1;
        }
        if (i >= 0 && count[i] > (unsigned long long)min_inliers) {
            const int s = atomicAdd(n_slots, 1);
            if (s < max_slots) {
                slot = s;
                root_of_slot[s] = i;
                count_of_slot[s] = (int)count[i];
            }
        }
        if (i >= 0) slot_of_root[i] = slot;
        // the moment rows of a new slot start at zero (no memset of the 16 x 4096 x 9 table): 144 words, written by the whole wave (one
        // thread writing them one after the other is 19 us of a lone wave's issue rate)
        unsigned long long fresh = __ballot(slot >= 0);
        while (fresh) {
            const int src = __builtin_ctzll(fresh);
            fresh &= fresh - 1;
            const int s = __builtin_amdgcn_readlane(slot, src);
            for (int q = lane; q < mom_replicas * 9; q += 64) mom[((size_t)(q / 9) * max_slots + s) * 9 + q % 9] = 0ull;
        }
    }
}

// 9 raw moments per selected region (sum x, y, z, xx, xy, xz, yy, yz, zz) in 2^-28 fixed point, two's complement in u64, at a
// cost that does not depend on how fragmented the label image is.  A lane owns 8 CONSECUTIVE pixels, so the slots a wave sees
// form runs along the lanes: every lane sums its pixels in registers (flushing to the LDS hash only where the slot changes
// inside its 8 pixels), then ONE segmented scan over the lanes (6 shuffle steps for the nine 64-bit sums, whatever the number
// of runs) leaves every run's total in its last lane, which adds it to the block's hash; a wave inside one region takes the
// plain butterfly sum instead.  (The first version summed lane-interleaved pixels with one wave reduction per region a wave
// met: 16 us at 2048 x 1024 with the room as one region, but 165 us with 342 regions, where most lanes saw several regions and
// fell back to single-lane LDS atomics; this one: 16.5 / 17.6 us, and 39 instead of 55 us at 4096 x 2048.)
// Integer sums: bitwise reproducible whatever the order.
constexpr int kMomRunHashBits = 8, kMomRunHash = 1 << kMomRunHashBits;
__device__ __forceinline__ int mom_run_slot(int* keys, int key) {
    int h = (int)(((unsigned)key * 2654435761u) >> (32 - kMomRunHashBits));
    for (int probe = 0; probe < 16; ++probe) {
        const int old = atomicCAS(&keys[h], -1, key);
        if (old == -1 || old == key) return h;
        h = (h + 1) & (kMomRunHash - 1);
    }
    return -1;
}
__global__ __launch_bounds__(kAggThreads) void k_f360_moments(const float* __restrict__ xyz, const int* __restrict__ label,
                                                                  const int* __restrict__ slot_of_root, int n,
                                                                  unsigned long long* __restrict__ mom, int max_slots) {
    __shared__ int keys[kMomRunHash];
    __shared__ unsigned long long vals[kMomRunHash][9];
    unsigned long long* mom_rep = mom + (size_t)(blockIdx.x % kMomReplicas) * max_slots * 9;
    if (threadIdx.x < kMomRunHash) {
        keys[threadIdx.x] = -1;
#pragma unroll
        for (int k = 0; k < 9; ++k) vals[threadIdx.x][k] = 0ull;
    }
    __syncthreads();
    const int lane = (int)threadIdx.x & 63;
    const int p0 = ((blockIdx.x * (kAggThreads / 64) + ((int)threadIdx.x >> 6)) * 64 + lane) * kMomPerThread;
    static_assert(kMomPerThread == 8, "two int4 label loads, six float4 point loads per lane");
    int sl[kMomPerThread];
    float pt[3 * kMomPerThread];
    const bool full = p0 + kMomPerThread <= n;
    if (full) {      // p0 is a multiple of 8: 32-byte aligned labels, 96-byte aligned points
        const int4 a = *reinterpret_cast<const int4*>(label + p0), b = *reinterpret_cast<const int4*>(label + p0 + 4);
        sl[0] = a.x; sl[1] = a.y; sl[2] = a.z; sl[3] = a.w; sl[4] = b.x; sl[5] = b.y; sl[6] = b.z; sl[7] = b.w;
        const float4* q = reinterpret_cast<const float4*>(xyz + 3 * (size_t)p0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float4 t = q[k];
            pt[4 * k] = t.x; pt[4 * k + 1] = t.y; pt[4 * k + 2] = t.z; pt[4 * k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < kMomPerThread; ++j) {
            const int i = p0 + j;
            sl[j] = i < n ? label[i] : -1;
            pt[3 * j] = i < n ? xyz[3 * (size_t)i] : 0.f;
            pt[3 * j + 1] = i < n ? xyz[3 * (size_t)i + 1] : 0.f;
            pt[3 * j + 2] = i < n ? xyz[3 * (size_t)i + 2] : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < kMomPerThread; ++j)
        if (sl[j] >= 0) sl[j] = slot_of_root[sl[j]];
    auto d2ll = [](double v) -> long long { return __double_as_longlong(v + 6755399441055744.0) - 0x4338000000000000LL; };
    auto flush = [&](int k, const long long t[9]) {
        const int e = mom_run_slot(keys, k);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (e >= 0) atomicAdd(&vals[e][q], (unsigned long long)t[q]);
            else atomicAdd(&mom_rep[(size_t)k * 9 + q], (unsigned long long)t[q]);
        }
    };
    long long v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int key = -1;
#pragma unroll
    for (int j = 0; j < kMomPerThread; ++j) {
        if (sl[j] < 0) continue;
        if (sl[j] != key) {
            if (key >= 0) {          // a second run inside the lane's 8 pixels: the first one goes out on its own
                flush(key, v);
#pragma unroll
                for (int q = 0; q < 9; ++q) v[q] = 0;
            }
            key = sl[j];
        }
        const double x = pt[3 * j], y = pt[3 * j + 1], z = pt[3 * j + 2];
        v[0] += d2ll(x * kMomScale); v[1] += d2ll(y * kMomScale); v[2] += d2ll(z * kMomScale);
        v[3] += d2ll(x * x * kMomScale); v[4] += d2ll(x * y * kMomScale); v[5] += d2ll(x * z * kMomScale);
        v[6] += d2ll(y * y * kMomScale); v[7] += d2ll(y * z * kMomScale); v[8] += d2ll(z * z * kMomScale);
    }
    // runs of equal keys along the lanes (key of a lane = the slot of its LAST run; -1 = nothing pending)
    const int first = __builtin_amdgcn_readfirstlane(key);
    if (__ballot(key == first) == __ballot(true)) {          // one region (or nothing) in the whole wave
        if (first >= 0) {
            long long t[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) t[q] = wave_sum_ll(v[q]);
            if (lane == 0) flush(first, t);
        }
    } else {
        const int prev = __shfl_up(key, 1);
        const bool head = lane == 0 || prev != key;
        const unsigned long long heads = __ballot(head);
        const int seg = __popcll(heads & (~0ull >> (63 - lane)));          // number of heads at or below this lane
        // segmented add-scan along the runs, on the DPP paths like wave_sum_ll (a source lane counts when it lies in the same run;
        // runs are contiguous, so lane 15 / lane 31 of the rows below stand for everything of the run below them)
#define F360_SEG_STEP(ctrl_, rows_)                                                                                        \
        {                                                                                                                  \
            const bool take = __builtin_amdgcn_update_dpp(-1, seg, ctrl_, rows_, 0xF, false) == seg;                       \
            _Pragma("unroll") for (int q = 0; q < 9; ++q) {                                                                \
                const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(unsigned long long)v[q], ctrl_, rows_, 0xF, false);          \
                const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)v[q] >> 32), ctrl_, rows_, 0xF, false);  \
                v[q] += take ? (long long)(((unsigned long long)hi << 32) | lo) : 0ll;                                     \
            }                                                                                                              \
        }
        F360_SEG_STEP(0x111, 0xF)
        F360_SEG_STEP(0x112, 0xF)
        F360_SEG_STEP(0x114, 0xF)
        F360_SEG_STEP(0x118, 0xF)
        F360_SEG_STEP(0x142, 0xA)
        F360_SEG_STEP(0x143, 0xC)
#undef F360_SEG_STEP
        const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
        if (tail && key >= 0) flush(key, v);
    }
    __syncthreads();
    if (threadIdx.x < kMomRunHash && keys[threadIdx.x] >= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) atomicAdd(&mom_rep[(size_t)keys[threadIdx.x] * 9 + k], vals[threadIdx.x][k]);
    }
}

// the 16 copies of the moment table -> copy 0 (integer sums: any order), so that the host fetches n_slots x 9 values once
// One record per selected region, as the host reads it back: ONE device-to-host copy of header + records instead of four.
struct F360SlotRecord {
    int root, count;
    unsigned long long mom[9];
};
constexpr int kF360PackHeader = 16;            // bytes: int n_slots + padding, the records follow (8-byte aligned)
__global__ void k_f360_mom_reduce(const unsigned long long* __restrict__ mom, const int* __restrict__ n_slots, int max_slots,
                                  const int* __restrict__ root_of_slot, const int* __restrict__ count_of_slot,
                                  unsigned char* __restrict__ pack, const int* __restrict__ relabelled, int* __restrict__ relabelled_host) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ns_all = *n_slots;
    if (i == 0) {
        *reinterpret_cast<int*>(pack) = ns_all;
        if (relabelled_host) *relabelled_host = *relabelled;      // the commit's counter travels with the records (no copy command of its own)
    }
    const int ns = ns_all < max_slots ? ns_all : max_slots;
    if (i >= ns * 9) return;
    unsigned long long acc = mom[i];
#pragma unroll
    for (int r = 1; r < kMomReplicas; ++r) acc += mom[(size_t)r * max_slots * 9 + i];
    const int slot = i / 9, q = i - slot * 9;
    F360SlotRecord* rec = reinterpret_cast<F360SlotRecord*>(pack + kF360PackHeader) + slot;
    rec->mom[q] = acc;
    if (q == 0) {
        rec->root = root_of_slot[slot];
        rec->count = count_of_slot[slot];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Colour descriptors of the planar regions (Frame360.h:1045-1046 / Frame360_stereo.h:949-950: plane.calcPlaneHistH();
// plane.calcMainColor2(); -- mrpt::pbmap::Plane, third-party, not in the reference tree; consumed by the PbMap matcher's unary colour
// constraint, config_files/configLocaliser_spherical.ini:19-21).  Per region, over its inlier pixels:
//   v3colorNrgb / dominantIntensity   mean and standard deviation of the normalised colour (R, G, B) / (R + G + B) and the mean of
//                                     R + G + B (the roles calcMainColor[2] fills; the mean stands in for MRPT's mean-shift mode)
//   hist_H                            normalised histogram of the SATURATED hue: 72 bins of 5 degrees + a bin for dark pixels
//                                     (V <= 0.2) + a bin for unsaturated ones (S <= 0.2) -- MRPT's 74-bin layout
// All per-pixel arithmetic is INTEGER (normalised colour in 2^-16 fixed point by integer division, hue bin by integer division of
// 12 (x - y) by max - min), so the per-region sums are bitwise reproducible and the CPU checker (oracle/frame360_ref.cpp) repeats them
// exactly; the host turns the sums into floats.  One pass over {label, colour}: 4 + 3 B per pixel.
// A lane owns 8 pixels 1024 apart (coalesced labels); its sums stay in registers while its region does not change, a wave inside
// one region adds them up once (the common case), the hue bins go to the block's LDS table by atomics (one table entry per region the
// block meets, 16 at most -- a pixel of a 17th region adds straight into the global table), the block's table goes out once.
// ---------------------------------------------------------------------------------------------------------
constexpr int kColSums = 8;          // sum qR, qG, qB | sum qR^2, qG^2, qB^2 | sum (R + G + B) | pixels with R + G + B > 0
constexpr int kColBins = 74;
constexpr int kColMode = 8;          // the dominant colour (k_f360_colour_mode): samples N | kept | mode qR, qG, qB | sum S over the kept | iterations | threshold^2
constexpr int kColWords = kColSums + kColBins + kColMode;      // 64-bit words per region in the global table
// calcMainColor2 (mrpt::pbmap::Plane, third-party; restated from the published source: the plane's points are thinned to about 2000 --
// `stepColor = size / 2000` -- and getMultiDimMeanShift_color shrinks the sample set around its running mean until half is gone or the
// mean stops moving).  Here the thinning is a regular grid over the region's pixels, rows r % sr == 0 and columns c % sc == 0 with
// sr = isqrt(step), sc = step / sr, step = max(count / kModeTarget, 1): a SET of samples that does not depend on any order.
constexpr int kModeTarget = 2000;
constexpr int kModeCap = 4096;       // samples a region can hold (a grid over a connected region of `count` pixels yields about count / (sr sc) <= 1.25 kModeTarget; a region with more gets no dominant colour: k_f360_colour_mode)
__host__ __device__ inline void mode_grid(int count, int& sr, int& sc) {
    const int step = count / kModeTarget > 1 ? count / kModeTarget : 1;
    sr = 1;
    while ((sr + 1) * (sr + 1) <= step) ++sr;
    sc = step / sr;
}
constexpr int kColPerThread = 8, kColHash = 16, kColListCap = 4096;
struct ColourImage {
    const uint8_t* rgb;      // device, 3 bytes per pixel
    size_t step;             // bytes per row
    int sub;                 // cloud pixel (r, c) takes image pixel (r sub + sub / 2, c sub + sub / 2): DownsampleRGBD.h:240, 285-287 (1: the image itself)
};
struct ColourPx {
    unsigned q[3], S;
    int bin;
};
// n / d for 0 <= n < 2^24, 0 < d, quotients below 2^17 (the colour stage's: (C << 16) / S <= 65536, hue bins, grid steps of rows and columns): the
// hardware reciprocal (1 ulp) puts the float quotient within 0.02 of the true one, the remainder test settles the last unit -- exact,
// at a third of the instructions of an IEEE division per quotient (the colour pass makes up to six per pixel)
__device__ __forceinline__ void divmod_small(int n, float rcp_d, int d, int& q, int& rem) {
    q = (int)((float)n * rcp_d);
    rem = n - q * d;
    if (rem < 0) { rem += d; --q; }
    else if (rem >= d) { rem -= d; ++q; }
}
__device__ __forceinline__ void divmod_small(int n, int d, int& q, int& rem) { divmod_small(n, __builtin_amdgcn_rcpf((float)d), d, q, rem); }
__device__ __forceinline__ ColourPx colour_px(unsigned R, unsigned G, unsigned B) {
    ColourPx o;
    o.S = R + G + B;
    int rem;
    int q0 = 0, q1 = 0, q2 = 0;
    if (o.S) {          // (C << 16) / S, exact: C << 16 < 2^24
        const float rS = __builtin_amdgcn_rcpf((float)o.S);
        divmod_small((int)(R << 16), rS, (int)o.S, q0, rem);
        divmod_small((int)(G << 16), rS, (int)o.S, q1, rem);
        divmod_small((int)(B << 16), rS, (int)o.S, q2, rem);
    }
    o.q[0] = (unsigned)q0; o.q[1] = (unsigned)q1; o.q[2] = (unsigned)q2;
    const unsigned mx = max(R, max(G, B)), mn = min(R, min(G, B)), delta = mx - mn;
    if (mx * 5u <= 255u) o.bin = 72;                     // V = max / 255 <= 0.2: dark
    else if (delta * 5u <= mx) o.bin = 73;               // S = (max - min) / max <= 0.2: unsaturated
    else {
        // hue in units of 5 degrees = 12 h6, h6 = sector + (x - y) / delta in [0, 6): floor((24 k delta + 12 (x - y)) / delta), wrapped
        int num;
        if (mx == R) num = 12 * ((int)G - (int)B);
        else if (mx == G) num = 24 * (int)delta + 12 * ((int)B - (int)R);
        else num = 48 * (int)delta + 12 * ((int)R - (int)G);
        if (num < 0) num += 72 * (int)delta;
        int b;
        divmod_small(num, (int)delta, b, rem);
        o.bin = b >= 72 ? b - 72 : b;
    }
    return o;
}
// (one block) where a region's samples start in the pool: exclusive prefix of min(count, kModeCap) over the slots; clears the sample
// counters and the slots' rows of the colour table, and leaves each slot's sample grid (mode_grid: the same for every pixel of a region)
__global__ __launch_bounds__(1024) void k_f360_colour_offsets(const int* __restrict__ n_slots, int max_slots, const int* __restrict__ count_of_slot,
                                                               int* __restrict__ samp_off, int* __restrict__ samp_n, int2* __restrict__ samp_grid,
                                                               unsigned long long* __restrict__ col) {
    __shared__ int part[1024];
    const int ns = min(*n_slots, max_slots);
    const int per = (ns + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, ns);
    int sum = 0;
    for (int s2 = lo; s2 < hi; ++s2) sum += min(count_of_slot[s2], kModeCap);
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int used = per > 0 ? min(1024, (ns + per - 1) / per) : 0;      // threads that own slots (a frame has tens of regions, not thousands)
        int run = 0;
        for (int t = 0; t < used; ++t) { const int v = part[t]; part[t] = run; run += v; }
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int s2 = lo; s2 < hi; ++s2) {
        const int cnt = count_of_slot[s2];
        samp_off[s2] = run;
        samp_n[s2] = 0;
        int sr, sc;
        mode_grid(cnt, sr, sc);
        samp_grid[s2] = make_int2(sr, sc);
        run += min(cnt, kModeCap);
    }
    for (int i = threadIdx.x; i < ns * kColWords; i += 1024) col[i] = 0ull;
}
struct ColourSamples {          // nullptr pool: no dominant colour is sought
    const int* count_of_slot;
    const int* samp_off;
    int* samp_n;
    const int2* grid;           // {sr, sc} of the slot's sample grid
    unsigned* pool;             // R | G << 8 | B << 16 of the sampled pixels, region by region
};
__global__ __launch_bounds__(kAggThreads) void k_f360_colour(const int* __restrict__ label, const int* __restrict__ slot_of_root, int rows, int cols,
                                                              ColourImage img, unsigned long long* __restrict__ col, ColourSamples smp) {
    __shared__ int keys[kColHash];
    __shared__ unsigned long long sums[kColHash][kColSums];
    __shared__ unsigned bins[kColHash][kColBins];
    __shared__ unsigned slist[kColListCap];              // the block's samples: R | G << 8 | B << 16 | table entry << 24
    __shared__ int slist_n, scount[kColHash], sbase[kColHash];
    if (threadIdx.x == 0) slist_n = 0;
    if (threadIdx.x < kColHash) scount[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < kColHash * kColBins; i += kAggThreads) (&bins[0][0])[i] = 0u;
    if (threadIdx.x < kColHash * kColSums) (&sums[0][0])[threadIdx.x] = 0ull;
    if (threadIdx.x < kColHash) keys[threadIdx.x] = -1;
    __syncthreads();
    const int n = rows * cols;
    auto entry_of = [&](int slot) -> int {               // the block's table entry of a region (-1: table full)
        int h = slot & (kColHash - 1);
        for (int probe = 0; probe < kColHash; ++probe) {
            const int old = atomicCAS(&keys[h], -1, slot);
            if (old == -1 || old == slot) return h;
            h = (h + 1) & (kColHash - 1);
        }
        return -1;
    };
    int key = -1, ent = -1;
    unsigned a32[5] = {0, 0, 0, 0, 0};                    // sum q (3), sum S, pixels with colour
    unsigned long long a64[3] = {0, 0, 0};                // sum q^2
    auto flush_lane = [&]() {                             // this lane's sums to its region's entry (or past a full table)
        if (key < 0) return;
        unsigned long long* dst = ent >= 0 ? sums[ent] : col + (size_t)key * kColWords;
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(&dst[k], (unsigned long long)a32[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(&dst[3 + k], a64[k]);
        atomicAdd(&dst[6], (unsigned long long)a32[3]);
        atomicAdd(&dst[7], (unsigned long long)a32[4]);
    };
    const int lane = (int)threadIdx.x & 63;
    // three batched rounds of loads (labels, their slots, the colour bytes of every pixel of the thread) in front of the arithmetic: as
    // a dependent triple per pixel the kernel was a chain of 8 x 3 memory round trips per thread (81 -> 5x us at 4096 x 2048)
    int slot_j[kColPerThread], row_j[kColPerThread], col_j[kColPerThread];
    unsigned rgb_j[kColPerThread];
#pragma unroll
    for (int j = 0; j < kColPerThread; ++j) {
        const int p = (blockIdx.x * kColPerThread + j) * kAggThreads + (int)threadIdx.x;
        slot_j[j] = p < n ? label[p] : -1;
    }
#pragma unroll
    for (int j = 0; j < kColPerThread; ++j) slot_j[j] = slot_j[j] >= 0 ? slot_of_root[slot_j[j]] : -1;
#pragma unroll
    for (int j = 0; j < kColPerThread; ++j) {
        const int p = (blockIdx.x * kColPerThread + j) * kAggThreads + (int)threadIdx.x;
        r360::divmod24(p < n ? p : 0, cols, row_j[j], col_j[j]);      // (IEEE: the row index is not a small quotient for every geometry)
        const uint8_t* px = img.rgb + (size_t)(row_j[j] * img.sub + img.sub / 2) * img.step + 3 * (size_t)(col_j[j] * img.sub + img.sub / 2);
        rgb_j[j] = (unsigned)px[0] | ((unsigned)px[1] << 8) | ((unsigned)px[2] << 16);
    }
#pragma unroll
    for (int j = 0; j < kColPerThread; ++j) {
        const int slot = slot_j[j];
        const bool on = slot >= 0;
        if (__ballot(on) == 0ull) continue;                  // uniform: nothing of a region in these 64 pixels
        const int r = row_j[j], c = col_j[j];
        const unsigned pr = rgb_j[j] & 255u, pg = (rgb_j[j] >> 8) & 255u, pb = rgb_j[j] >> 16;
        const ColourPx v = colour_px(pr, pg, pb);
        if (on && slot != key) {
            flush_lane();
            key = slot;
            ent = entry_of(slot);
#pragma unroll
            for (int k = 0; k < 5; ++k) a32[k] = 0;
            a64[0] = a64[1] = a64[2] = 0;
        }
        // The dominant colour's sample grid (kModeTarget above).  A sample needs a place in its region's stretch of the pool, i.e. a
        // counter per region -- bumped once per sample in global memory, the ~2500 samples of a wall queue up on one address (250 of this
        // kernel's 320 us at 4096 x 2048).  The block collects its samples in LDS and bumps each region's counter once, by its number of
        // samples in the block (below); only what does not fit the block's table or list goes to the global counter directly.
        if (smp.pool) {                                       // uniform
            bool samp = false;
            if (on && v.S) {
                const int2 g = smp.grid[slot];
                int qq, rr = 0, rc = 0;
                if (g.x > 1) divmod_small(r, g.x, qq, rr);
                if (g.y > 1 && rr == 0) divmod_small(c, g.y, qq, rc);
                samp = rr == 0 && rc == 0;
            }
            const unsigned rgb = pr | (pg << 8) | (pb << 16);
            const bool listed = samp && ent >= 0;
            const unsigned long long m = __ballot(listed);
            int idx = kColListCap;
            if (m) {                                          // uniform
                const int lead = __builtin_ctzll(m);
                int base = 0;
                if (lane == lead) base = atomicAdd(&slist_n, (int)__popcll(m));
                base = __builtin_amdgcn_readlane(base, lead);
                idx = base + (int)__popcll(m & ((1ull << lane) - 1ull));
                if (listed && idx < kColListCap) slist[idx] = rgb | ((unsigned)ent << 24);
            }
            if (samp && (ent < 0 || idx >= kColListCap)) {
                const int pos = atomicAdd(&smp.samp_n[slot], 1);
                if (pos < kModeCap) smp.pool[smp.samp_off[slot] + pos] = rgb;
            }
        }
        // hue bins: one LDS add per (region, bin) the wave's 64 pixels hold, by the group's first lane, with the group's size -- a wall
        // is one or two bins wide, and 64 lanes adding 1 to the same LDS word are 64 serial read-modify-writes (this was 5/6 of the
        // kernel: 316 -> 60 us at 4096 x 2048)
        {
            const bool in_table = on && ent >= 0;
            const int kb = (ent << 8) | v.bin;
            unsigned long long todo = __ballot(in_table);
            while (todo) {                                    // uniform
                const int lead = __builtin_ctzll(todo);
                const int kbl = __builtin_amdgcn_readlane(kb, lead);
                const unsigned long long same = __ballot(in_table && kb == kbl);
                if (lane == lead) atomicAdd(&bins[kbl >> 8][kbl & 255], (unsigned)__popcll(same));
                todo &= ~same;
            }
            if (on && ent < 0) atomicAdd(&col[(size_t)slot * kColWords + kColSums + v.bin], 1ull);
        }
        if (on && v.S) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a32[k] += v.q[k];
                a64[k] += (unsigned long long)v.q[k] * v.q[k];
            }
            a32[3] += v.S;
            a32[4] += 1u;
        }
    }
    // a wave that ended inside ONE region (entry in the block's table) adds its lanes up first; anything else goes lane by lane
    const int first = __builtin_amdgcn_readfirstlane(key);
    const int fent = __builtin_amdgcn_readfirstlane(ent);
    if (__ballot(key == first && ent == fent) == __ballot(true) && first >= 0 && fent >= 0) {
        unsigned t32[5];
        long long t64[3];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            unsigned x = a32[k];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
            t32[k] = x;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) t64[k] = wave_sum_ll((long long)a64[k]);
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* dst = sums[fent];
#pragma unroll
            for (int k = 0; k < 3; ++k) atomicAdd(&dst[k], (unsigned long long)t32[k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) atomicAdd(&dst[3 + k], (unsigned long long)t64[k]);
            atomicAdd(&dst[6], (unsigned long long)t32[3]);
            atomicAdd(&dst[7], (unsigned long long)t32[4]);
        }
    } else {
        flush_lane();
    }
    __syncthreads();
    if (smp.pool) {                                       // uniform: the block's samples move to their regions' stretches of the pool
        const int nl = min(slist_n, kColListCap);
        for (int i = threadIdx.x; i < nl; i += kAggThreads) atomicAdd(&scount[slist[i] >> 24], 1);
        __syncthreads();
        if (threadIdx.x < kColHash) {
            const int cnt = scount[threadIdx.x];
            sbase[threadIdx.x] = cnt > 0 ? atomicAdd(&smp.samp_n[keys[threadIdx.x]], cnt) : 0;      // one bump per (block, region)
            scount[threadIdx.x] = 0;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nl; i += kAggThreads) {
            const unsigned w = slist[i];
            const int e = (int)(w >> 24);
            const int pos = sbase[e] + atomicAdd(&scount[e], 1);
            if (pos < kModeCap) smp.pool[smp.samp_off[keys[e]] + pos] = w & 0xFFFFFFu;
        }
    }
    constexpr int kColAcc = kColSums + kColBins;          // (the words this pass accumulates; the dominant colour's follow behind them)
    for (int i = threadIdx.x; i < kColHash * kColAcc; i += kAggThreads) {
        const int e = i / kColAcc, w = i - e * kColAcc;
        if (keys[e] < 0) continue;
        const unsigned long long v = w < kColSums ? sums[e][w] : (unsigned long long)bins[e][w - kColSums];
        if (v) atomicAdd(&col[(size_t)keys[e] * kColWords + w], v);
    }
}
// The dominant colour of a region: getMultiDimMeanShift_color (MRPT, restated) on its samples, in INTEGER arithmetic so that the CPU checker
// repeats it exactly whatever the order the samples arrived in: normalised colour q = (C << 16) / S as in colour_px,
//   mean   = floor(sum q / n) per channel,   threshold^2 = sum over channels of (floor(sum q^2 / N) - mean^2)   (the norm of the std. dev.)
//   while 2 n > N and |shift|^2 > (0.001 * 65536)^2:  drop the samples farther than the threshold from the mean (for good);
//                                                     mean = that of the n left; shift = its move
// One block per region slot; samples in LDS (16 bytes each).  Results into the region's row of the colour table (kColMode words).
constexpr int kModeThreads = 256;
// Also the stage's last kernel: a block copies the finished row of each of its slots into pinned host memory (behind the plane list's
// other records: one wait ends the call).
__global__ __launch_bounds__(kModeThreads) void k_f360_colour_mode(const int* __restrict__ n_slots, int max_slots, ColourSamples smp,
                                                                    unsigned long long* __restrict__ col, unsigned long long* __restrict__ host_out) {
    __shared__ uint4 sm[kModeCap];                 // {qR, qG, qB, S | alive << 31}
    __shared__ unsigned long long acc[8];
    const int ns = min(*n_slots, max_slots);
    const int lane = (int)threadIdx.x & 63;
    // a block's five sums of a step: added up inside the waves first (256 threads adding to five LDS words are 1280 serial
    // read-modify-writes per step, and a region takes up to 64 steps)
    auto block_add = [&](int k, unsigned long long v) {
        const unsigned long long t = (unsigned long long)wave_sum_ll((long long)v);
        if (lane == 0 && t) atomicAdd(&acc[k], t);
    };
    for (int slot = blockIdx.x; slot < ns; slot += gridDim.x) {
        // A region whose grid yields more samples than a slot holds (samp_n is the exact, order-free COUNT of its grid samples; thin
        // row-aligned regions can exceed count / (sr sc)) gets NO dominant colour: which samples would have found a place depends on
        // their arrival order, and a mode of an arbitrary subset is not repeatable.  color_mode_count = 0 then: the matcher compares
        // the means, the CPU checker says the same.
        const int N = smp.samp_n[slot] > kModeCap ? 0 : smp.samp_n[slot];
        unsigned long long* row = col + (size_t)slot * kColWords;
        unsigned long long* out = row + kColSums + kColBins;
        __syncthreads();                               // (the previous slot's state is no longer read)
        if (N <= 0) {
            if (threadIdx.x < kColMode) out[threadIdx.x] = 0ull;
        } else {
            if (threadIdx.x < 8) acc[threadIdx.x] = 0ull;
            __syncthreads();
            unsigned long long s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
            {
                // every sample of the thread requested before the first is used (a load per trip of a rolled loop was a round trip per trip)
                constexpr int kPer = kModeCap / kModeThreads;
                const int off = smp.samp_off[slot];
                unsigned wv[kPer];
#pragma unroll
                for (int u = 0; u < kPer; ++u) {
                    const int i = (int)threadIdx.x + u * kModeThreads;
                    wv[u] = i < N ? smp.pool[off + i] : 0u;
                }
#pragma unroll
                for (int u = 0; u < kPer; ++u) {
                    const int i = (int)threadIdx.x + u * kModeThreads;
                    if (i < N) {
                        const unsigned w = wv[u];
                        const ColourPx v = colour_px(w & 255u, (w >> 8) & 255u, (w >> 16) & 255u);
                        sm[i] = make_uint4(v.q[0], v.q[1], v.q[2], v.S | 0x80000000u);
#pragma unroll
                        for (int k = 0; k < 3; ++k) { s1[k] += v.q[k]; s2[k] += (unsigned long long)v.q[k] * v.q[k]; }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) { block_add(k, s1[k]); block_add(3 + k, s2[k]); }
            __syncthreads();
            long long m[3];
            unsigned long long thr2 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                m[k] = (long long)(acc[k] / (unsigned long long)N);
                const long long var = (long long)(acc[3 + k] / (unsigned long long)N) - m[k] * m[k];
                thr2 += (unsigned long long)(var > 0 ? var : 0);
            }
            int n_alive = N, iters = 0;
            unsigned long long shift2 = ~0ull, sumS = 0;
            const unsigned long long conv2 = 4294ull;          // floor((0.001 * 65536)^2)
            for (;;) {
                __syncthreads();                               // everybody has read acc / the loop state is uniform
                if (!(2 * (long long)n_alive > N && shift2 > conv2) || iters >= 64) break;
                if (threadIdx.x < 8) acc[threadIdx.x] = 0ull;
                __syncthreads();
                unsigned long long t1[3] = {0, 0, 0}, tS = 0, tn = 0;
                for (int i = threadIdx.x; i < N; i += kModeThreads) {
                    uint4 e = sm[i];
                    if (!(e.w & 0x80000000u)) continue;
                    const long long d0 = (long long)e.x - m[0], d1 = (long long)e.y - m[1], d2 = (long long)e.z - m[2];
                    if ((unsigned long long)(d0 * d0 + d1 * d1 + d2 * d2) > thr2) {
                        sm[i].w = e.w & 0x7fffffffu;           // erased for good
                        continue;
                    }
                    t1[0] += e.x; t1[1] += e.y; t1[2] += e.z; tS += e.w & 0x7fffffffu; ++tn;
                }
                // (a block's sums stay below 2^28 -- 4096 samples of 16-bit values -- so two travel in one 64-bit sum)
                block_add(0, t1[0] | (t1[1] << 32));
                block_add(1, t1[2] | (tS << 32));
                block_add(2, tn);
                __syncthreads();
                ++iters;
                const unsigned long long p0 = acc[0], p1 = acc[1];
                const unsigned long long tot[3] = {p0 & 0xFFFFFFFFull, p0 >> 32, p1 & 0xFFFFFFFFull};
                const long long left = (long long)acc[2];
                if (left == 0) { n_alive = 0; break; }         // (every sample beyond the threshold: the mean stands)
                shift2 = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const long long mk = (long long)((unsigned)tot[k] / (unsigned)left);      // (both below 2^32: a 32-bit division, a quarter of the 64-bit one's instructions)
                    shift2 += (unsigned long long)((mk - m[k]) * (mk - m[k]));
                    m[k] = mk;
                }
                n_alive = (int)left;
                sumS = p1 >> 32;
            }
            if (iters == 0 || n_alive == 0) {                  // no shrink step ran (N == 1 ...) or it emptied the set: the plain mean over all samples
                unsigned long long tS = 0;
                for (int i = 0; i < N; ++i) tS += sm[i].w & 0x7fffffffu;      // (uniform, rare)
                sumS = tS;
                if (n_alive == 0) n_alive = N;
            }
            if (threadIdx.x == 0) {
                out[0] = (unsigned long long)N; out[1] = (unsigned long long)n_alive;
                out[2] = (unsigned long long)m[0]; out[3] = (unsigned long long)m[1]; out[4] = (unsigned long long)m[2];
                out[5] = sumS; out[6] = (unsigned long long)iters; out[7] = thr2;
            }
        }
        __syncthreads();                                       // the row is whole (its sums and bins are the previous launch's)
        if (host_out)
            for (int i = threadIdx.x; i < kColWords; i += kModeThreads) host_out[(size_t)slot * kColWords + i] = row[i];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Convex hull of a planar region (Frame360.h:1009-1031: regions[i].getContour() -> mrpt::pbmap::Plane::calcConvexHull ->
// computeMassCenterAndArea; MRPT is not in the reference tree: the hull of the region's contour projected onto its plane, the area
// and mass centre of that polygon).  The hull of a region's contour is the hull of its pixels, and only pixels on the region's
// boundary can be hull vertices.  On the device the boundary pixels are reduced to the region's EXTREME point in each of
// kHullDirs = 128 in-plane directions per block (a wave holds two directions per lane -- an angle in the first half turn and the
// opposite one, which share one dot product -- and walks the boundary pixels of its 64-pixel stretch through v_readlane; one 64-bit
// atomicMax per direction and run of equal labels, key = {order-preserving dot product, pixel}), kHullPhases = 8 interleaved sets of
// them, 1024 directions in all: an inscribed polygon that contains every hull vertex whose exterior angle exceeds a few direction
// steps -- exact for sharp-cornered polygons, 0.01 % low for a disc; what it can lose is a long, slightly bowed edge whose normals all
// fall between two directions of the set that happens to see it (the margin of a wall's region widens with the range).  The extremes
// arrive in direction order, i.e. already in hull order: the host drops the non-left turns in one linear pass (no sort) and takes
// the shoelace sums.
//   k_f360_slot_frames   per region slot: centroid + in-plane basis (eigenvectors of the inlier covariance, float64 Jacobi)
//   k_f360_hull_extremes per 64-pixel stretch: boundary test, in-plane coordinates, directional maxima
//   k_f360_hull_pack     winners' coordinates + the frame into the pinned record the host reads
// ---------------------------------------------------------------------------------------------------------
constexpr int kHullDirs = 128;       // per block, 2 per lane: a lane's direction and the opposite one share one dot product (round 4; 256 = quarter turns until then)
// Eight direction sets, an eighth of the set's angular step apart: block b works with set b mod 8.  Thanks to the scattered stretches every
// block sees a sample of every edge of a region, so the union of the sets' winners is a polygon of up to 1024 region pixels, still
// inscribed in the hull.  (Rounds 2-3: four sets of 256 directions, four per lane: 0.9988 x the exact hull at worst over 30 random scenes;
// eight sets of 128: 0.9979 -- each direction sees an eighth of the boundary instead of a quarter -- for half the walk, 1 KB table rows
// and 64 of them per block, tests/tools/hull_soak.py.  Round 6, 72 more draws with other seeds: 0.9968 once -- a 6.7 m wall-floor edge along
// an image row, bowed by one pixel in the plane: its whole bend is 0.7 degrees, i.e. two of the 1024 directions, and along a ROW the stretches
// of one set lie 512 pixels apart, more than the edge is long, so neither direction had a pixel near the apex; the strip lost is a pixel
// wide.  tests/tools/hull_case.py prints such a case vertex by vertex.)
constexpr int kHullPhases = 8;         // direction sets, block b works on set b mod 8 (round 4: 4 sets of 256 -- a table row was 2 KB and a block's
                                       // 32 rows overflowed into global atomics on frames with many planes: 250 us at 4096 x 2048 with 444 planes)
constexpr int kHullRecPts = kHullPhases * kHullDirs;     // points a record can hold; a point that wins several directions of a set is sent once
struct SlotFrame {
    float c[3], e1[3], e2[3], nrm[3];
};
struct F360HullRecord {
    float c[3], e1[3], e2[3];
    int n;                              // points that follow, roughly in direction order (the host sorts them anyway)
    float uv[kHullRecPts][2];           // in-plane coordinates of the extreme pixels
};
// One WAVE per slot.  Lanes 0-8 sum the moment replicas, lane 0 fits the plane, all lanes clear the slot's row of extremes.  Only the
// NORMAL has to be accurate (the hull's area does not depend on the in-plane axes): it is the eigenvector of the covariance's smallest
// eigenvalue l0, found without an iterative diagonalisation -- Newton's method on the (monic) characteristic polynomial started at 0
// converges to l0 monotonically from the left (the polynomial is increasing and concave below its first root), and an eigenvector of
// l0 is the largest cross product of two rows of C - l0 I.  (A cyclic Jacobi in one thread per slot took 77 us per frame.)
__global__ __launch_bounds__(64) void k_f360_slot_frames(const unsigned long long* __restrict__ mom, const int* __restrict__ n_slots, int max_slots,
                                                         const int* __restrict__ count_of_slot, SlotFrame* __restrict__ frames,
                                                         unsigned long long* __restrict__ ext, const int* __restrict__ root_of_slot,
                                                         unsigned char* __restrict__ pack) {
    const int lane = threadIdx.x;
    const int ns = min(*n_slots, max_slots);
    // pack != nullptr: this kernel also does k_f360_mom_reduce's job (the moment sums it forms anyway, root and count, into the pinned
    // record the host reads) -- when no refinement will change the sums afterwards
    if (pack && blockIdx.x == 0 && lane == 0) *reinterpret_cast<int*>(pack) = *n_slots;
    for (int slot = blockIdx.x; slot < ns; slot += gridDim.x) {
#pragma unroll
    for (int k = 0; k < kHullPhases * kHullDirs / 64; ++k) ext[(size_t)slot * kHullPhases * kHullDirs + 64 * k + lane] = 0ull;
    double mine = 0.0;
    if (lane < 9) {
        unsigned long long acc = 0ull;
        for (int r = 0; r < kMomReplicas; ++r) acc += mom[((size_t)r * max_slots + slot) * 9 + lane];
        mine = (double)(long long)acc / kMomScale;
        if (pack) {
            F360SlotRecord* rec = reinterpret_cast<F360SlotRecord*>(pack + kF360PackHeader) + slot;
            rec->mom[lane] = acc;
            if (lane == 0) {
                rec->root = root_of_slot[slot];
                rec->count = count_of_slot[slot];
            }
        }
    }
    double m[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) m[q] = __shfl(mine, q);
    if (lane != 0) continue;
    const double N = (double)count_of_slot[slot];
    const double cx = m[0] / N, cy = m[1] / N, cz = m[2] / N;
    const double a00 = m[3] / N - cx * cx, a01 = m[4] / N - cx * cy, a02 = m[5] / N - cx * cz;
    const double a11 = m[6] / N - cy * cy, a12 = m[7] / N - cy * cz, a22 = m[8] / N - cz * cz;
    // f(l) = l^3 - c2 l^2 + c1 l - c0
    const double c2 = a00 + a11 + a22;
    const double c1 = (a00 * a11 - a01 * a01) + (a00 * a22 - a02 * a02) + (a11 * a22 - a12 * a12);
    const double c0 = a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02);
    double l = 0.0;
    for (int it = 0; it < 12; ++it) {
        const double f = ((l - c2) * l + c1) * l - c0, df = (3.0 * l - 2.0 * c2) * l + c1;
        if (!(df > 0.0)) break;
        const double step = f / df;
        l -= step;
        if (fabs(step) <= 1e-15 * c2) break;
    }
    const double r0[3] = {a00 - l, a01, a02}, r1[3] = {a01, a11 - l, a12}, r2[3] = {a02, a12, a22 - l};
    auto cross3 = [](const double* u, const double* v, double* o) {
        o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0];
    };
    double n01[3], n02[3], n12[3];
    cross3(r0, r1, n01); cross3(r0, r2, n02); cross3(r1, r2, n12);
    const double q01 = n01[0] * n01[0] + n01[1] * n01[1] + n01[2] * n01[2], q02 = n02[0] * n02[0] + n02[1] * n02[1] + n02[2] * n02[2];
    const double q12 = n12[0] * n12[0] + n12[1] * n12[1] + n12[2] * n12[2];
    double nn[3], qq = q01;
    nn[0] = n01[0]; nn[1] = n01[1]; nn[2] = n01[2];
    if (q02 > qq) { qq = q02; nn[0] = n02[0]; nn[1] = n02[1]; nn[2] = n02[2]; }
    if (q12 > qq) { qq = q12; nn[0] = n12[0]; nn[1] = n12[1]; nn[2] = n12[2]; }
    if (!(qq > 0.0)) { nn[0] = 0.0; nn[1] = 0.0; nn[2] = 1.0; qq = 1.0; }      // isotropic blob: any frame
    const double inv = 1.0 / sqrt(qq);
    nn[0] *= inv; nn[1] *= inv; nn[2] *= inv;
    // any orthonormal pair across the normal
    double ax[3] = {1.0, 0.0, 0.0};
    if (fabs(nn[0]) > 0.9) { ax[0] = 0.0; ax[1] = 1.0; }
    double e1[3], e2[3];
    cross3(nn, ax, e1);
    const double i1 = 1.0 / sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    e1[0] *= i1; e1[1] *= i1; e1[2] *= i1;
    cross3(nn, e1, e2);
    SlotFrame F;
    F.c[0] = (float)cx; F.c[1] = (float)cy; F.c[2] = (float)cz;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        F.nrm[k] = (float)nn[k];
        F.e1[k] = (float)e1[k];
        F.e2[k] = (float)e2[k];
    }
    frames[slot] = F;
    }
}
__global__ void k_f360_hull_clear(const int* __restrict__ n_slots, int max_slots, unsigned long long* __restrict__ ext) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < min(*n_slots, max_slots) * kHullPhases * kHullDirs) ext[i] = 0ull;
}
__device__ __forceinline__ unsigned hull_f2ord(float f) {          // order-preserving float -> unsigned
    const unsigned b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
// A block of 16 waves covers 16 x kHullChunks stretches of 64 pixels, scattered over the image (see hull_stretch below), and keeps the
// extremes of the regions it meets in LDS (kHullHash rows; a region beyond that goes to memory directly); its table is written to
// memory as rows of its own at the end.  The walk over a wave's boundary pixels is serial (every lane updates its four directions per
// pixel), so the work sits where the long, nearly horizontal edges are.  (First version: eight consecutive 1024-pixel stretches per
// block and 64-bit keys in the inner loop, 77 us, the straggler blocks being those along the floor's and ceiling's edges.)
constexpr int kHullBlock = 1024, kHullChunks = 8, kHullHash = 64, kHullFramesLds = 128;
constexpr int kHullRot = kHullDirs / 64;      // directions per lane: the lane's own and its rotation(s)
static_assert(kHullRot == 2, "the walk below pairs a direction with its opposite");
#ifdef F360_HULL_TWO
constexpr int kHullList = 512;
#else
constexpr int kHullList = 640;
#endif        // boundary pixels a block shares out among its waves per round
struct HullEntry {
    int slot;
    float u, v;
    int pix;
};
#ifdef RGBD360_HULL_DBG          // diagnostic build (tools/hull_stamps.py): per-wave clock accounting of k_f360_hull_extremes
__device__ unsigned long long g_hull_dbg[4096][8];
#define HDBG_T() __builtin_readcyclecounter()
#else
#define HDBG_T() 0ull
#endif
__device__ __forceinline__ void hull_extremes_block(const float* __restrict__ xyz, const int* __restrict__ label,
                                                                    const int* __restrict__ slot_of_root, int rows, int cols,
                                                                    const SlotFrame* __restrict__ frames, unsigned long long* __restrict__ ext,
                                                                    int* __restrict__ part_keys, unsigned long long* __restrict__ part_vals, int n_frames_lds) {
    __shared__ int keys[kHullHash];
    __shared__ unsigned long long vals[kHullHash][kHullDirs];
    __shared__ float fr[kHullFramesLds][9];                // centroid + in-plane axes of the first slots: no dependent gather in front of the walk
    __shared__ HullEntry list[kHullList];
    __shared__ int cnt[(kHullBlock / 64) * kHullChunks];   // boundary pixels per (wave, stretch)
    const int lane = threadIdx.x & 63;
    const unsigned long long hd_t0 = HDBG_T();
    unsigned long long hd_walk = 0, hd_flush = 0, hd_uv = 0, hd_entries = 0, hd_runs = 0;
    const int n = rows * cols;
    // Stretch `chunk` of wave w in block b starts at pixel 64 ((chunk 16 + w) gridDim + b): the 64-pixel stretches of an image row go
    // to CONSECUTIVE BLOCKS.  The walk below is serial per wave (~30 instructions per boundary pixel), and a long horizontal edge
    // fills every stretch along its row; while a block owned 1024 consecutive pixels those were the 16 waves of two blocks, four
    // walking waves per SIMD sharing its issue slots: 300 cycles per pixel, 20 k cycles per stretch, on two CUs, with the other 254 idle
    // (tools/hull_stamps.py).  Spread over 32 (64 at 4096 x 2048) CUs, each such stretch has a SIMD almost to itself.
    // Shift by the slot index k: with plain interleaving a VERTICAL edge (the same 64-pixel column in every row) would meet the same
    // gridDim / (stretches per row) blocks in every row; this way all blocks take turns, and a block's share of such an edge falls on
    // one of its waves.
    const int wave_u = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // first pixel and first column of the wave's stretches, all wave-uniform, with TWO integer divisions per wave: from one stretch to the
    // next k grows by 16, the stretch index by 16 G + 16 (minus G where (b + k) mod G wraps), the column by that many pixels modulo the width
    const int G = (int)gridDim.x;
    int first_of[kHullChunks], col_of[kHullChunks];
    {
        int j = ((int)blockIdx.x + wave_u) % G;
        int first = (wave_u * G + j) * 64;
        int c0 = first % cols;
        const int step_px = ((kHullBlock / 64) * G + (kHullBlock / 64)) * 64;
        int step_c = step_px, wrap_c = G * 64;
        if (step_c >= cols) step_c %= cols;
        if (wrap_c >= cols) wrap_c %= cols;
#pragma unroll
        for (int chunk = 0; chunk < kHullChunks; ++chunk) {
            first_of[chunk] = first;
            col_of[chunk] = c0;
            j += kHullBlock / 64;
            first += step_px;
            c0 += step_c;
            while (j >= G) {                               // (more than once only for grids under 16 blocks)
                j -= G;
                first -= G * 64;
                c0 -= wrap_c;
            }
            while (c0 < 0) c0 += cols;
            while (c0 >= cols) c0 -= cols;
        }
    }
    auto hull_stretch = [&](int chunk) { return first_of[chunk]; };
    // The labels of all the wave's stretches are requested first (before the block's tables are set up: the first memory round trip
    // of a launch is the long one), then all the slot look-ups: two memory round trips for the eight stretches instead of two per
    // stretch (one after the other they were most of the kernel's 34 us; the serial walk only runs where there is a boundary).
    // Range-checked buffer loads with one 32-bit lane offset (a pixel outside the image reads 0 and is masked): with 64-bit addresses
    // and a division per lane and stretch this part was 770 vector instructions per wave, 5.7 us of issue with four waves per SIMD.
    const __amdgpu_buffer_rsrc_t r_lab = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(label), 0, n * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_xyz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xyz), 0, n * 12, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_sor = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(slot_of_root), 0, n * 4, 0x00020000);
    int Lc[kHullChunks];
    bool differs[kHullChunks];
    int Lraw[kHullChunks][5];
    bool edge_of[kHullChunks];
#pragma unroll
    for (int chunk = 0; chunk < kHullChunks; ++chunk) {
        const int i = first_of[chunk] + lane;
        int c = col_of[chunk] + lane;
        while (__ballot(c >= cols) != 0ull) c = c >= cols ? c - cols : c;      // once for images at least 64 wide
        edge_of[chunk] = i < cols || i >= n - cols || c == 0 || c == cols - 1;
        // a neighbour outside the image reads 0: only border pixels have one, and they count as boundary anyway
        const int vo = i * 4;
        Lraw[chunk][0] = __builtin_amdgcn_raw_buffer_load_b32(r_lab, vo, 0, 0);
        Lraw[chunk][1] = __builtin_amdgcn_raw_buffer_load_b32(r_lab, vo - 4, 0, 0);
        Lraw[chunk][2] = __builtin_amdgcn_raw_buffer_load_b32(r_lab, vo + 4, 0, 0);
        Lraw[chunk][3] = __builtin_amdgcn_raw_buffer_load_b32(r_lab, vo - cols * 4, 0, 0);
        Lraw[chunk][4] = __builtin_amdgcn_raw_buffer_load_b32(r_lab, vo + cols * 4, 0, 0);
    }
    for (int k = threadIdx.x; k < n_frames_lds * 9; k += kHullBlock) {
        const int sl = k / 9, q = k - sl * 9;
        fr[sl][q] = reinterpret_cast<const float*>(frames + sl)[q];      // SlotFrame starts with c[3], e1[3], e2[3]
    }
    for (int k = threadIdx.x; k < kHullHash * kHullDirs; k += kHullBlock) (&vals[0][0])[k] = 0ull;
    if (threadIdx.x < kHullHash) keys[threadIdx.x] = -1;
    // this lane's direction (cos, sin) in the first half turn and the opposite one: directions lane + 64 m, m = 0, 1, of the block's set
    // have the dot products d0 = u c + v s and -d0.  (The hardware's sine / cosine take revolutions; any 1024 directions spread over the
    // circle do, they need not be exact.)
    const int phase = (int)blockIdx.x & (kHullPhases - 1);
    const float rev = (float)(lane * kHullPhases + phase) * (1.f / (kHullDirs * kHullPhases));
    const float sk = __builtin_amdgcn_sinf(rev), ck = __builtin_amdgcn_cosf(rev);
#pragma unroll
    for (int chunk = 0; chunk < kHullChunks; ++chunk) {
        const int L = Lraw[chunk][0];
        const bool inside = hull_stretch(chunk) + lane < n;
        Lc[chunk] = inside ? L : -1;
        differs[chunk] = inside && L >= 0 && (edge_of[chunk] || Lraw[chunk][1] != L || Lraw[chunk][2] != L || Lraw[chunk][3] != L || Lraw[chunk][4] != L);
    }
    // second round trip, again for all stretches at once: the slot of every pixel whose label differs from a neighbour's, and its point
    int slots[kHullChunks];
    float px[kHullChunks], py[kHullChunks], pz[kHullChunks];
#pragma unroll
    for (int chunk = 0; chunk < kHullChunks; ++chunk) {
        const int i = hull_stretch(chunk) + lane;
        slots[chunk] = -1;
        px[chunk] = py[chunk] = pz[chunk] = 0.f;
        if (differs[chunk]) {
            slots[chunk] = __builtin_amdgcn_raw_buffer_load_b32(r_sor, Lc[chunk] * 4, 0, 0);
            const f3v w = (f3v)__builtin_amdgcn_raw_buffer_load_b96(r_xyz, i * 12, 0, 0);
            px[chunk] = w.x; py[chunk] = w.y; pz[chunk] = w.z;
        }
    }
    __syncthreads();                                    // the block's tables (set up while the labels were in flight)
    const unsigned long long hd_t1 = HDBG_T();
    // Walks a wave's worth of boundary pixels -- lane l holds one (slot < 0: none) -- slot by slot (a ballot per distinct slot): every
    // lane updates the maxima of its four directions per pixel (largest dot product so far and the pixel that has it; ascending, ties
    // keep the first), then the run's maxima go into the slot's row of the block's table (wave-uniform linear probe, four LDS atomics
    // nothing waits for; a full table sends them straight to memory).
    auto walk = [&](int slot, float u, float v, int pix) {
        unsigned long long remaining = __ballot(slot >= 0);
        const int ui = __builtin_bit_cast(int, u), vi = __builtin_bit_cast(int, v);
        while (remaining != 0ull) {
            const int cur = __builtin_amdgcn_readlane(slot, __builtin_ctzll(remaining));
            unsigned long long todo = __ballot(slot == cur);
            remaining &= ~todo;
            const unsigned long long hd_b = HDBG_T();
            hd_entries += __builtin_popcountll(todo); hd_runs += 1;
            float bd[kHullRot];
            int bj[kHullRot];
#pragma unroll
            for (int m = 0; m < kHullRot; ++m) { bd[m] = -__builtin_inff(); bj[m] = 0; }
            while (todo != 0ull) {                         // wave-uniform
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const float uj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ui, j));
                const float vj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, j));
                const int pj = __builtin_amdgcn_readlane(pix, j);
                const float d0 = fmaf(uj, ck, vj * sk);
                if (d0 > bd[0]) { bd[0] = d0; bj[0] = pj; }
                if (-d0 > bd[1]) { bd[1] = -d0; bj[1] = pj; }
            }
#ifdef RGBD360_HULL_DBG
            asm volatile("" :: "v"(bd[0]), "v"(bd[1]));
#endif
            const unsigned long long hd_c = HDBG_T();
            hd_walk += hd_c - hd_b;
            int h = -1;
            for (int q = 0; q < kHullHash && h < 0; ++q) {
                const int k = (cur + q) & (kHullHash - 1);
                int seen = keys[k];
                if (seen == -1) {
                    if (lane == 0) seen = atomicCAS(&keys[k], -1, cur);
                    seen = __builtin_amdgcn_readfirstlane(seen);
                    if (seen == -1) seen = cur;
                }
                if (seen == cur) h = k;
            }
            unsigned long long key[kHullRot];
#pragma unroll
            for (int m = 0; m < kHullRot; ++m) key[m] = ((unsigned long long)hull_f2ord(bd[m]) << 32) | (unsigned)bj[m];
            if (h >= 0) {
                unsigned long long* row = &vals[h][lane];
#pragma unroll
                for (int m = 0; m < kHullRot; ++m) __hip_atomic_fetch_max(row + 64 * m, key[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
#pragma unroll
                for (int m = 0; m < kHullRot; ++m) atomicMax(&ext[((size_t)cur * kHullPhases + phase) * kHullDirs + 64 * m + lane], key[m]);
            }
            hd_flush += HDBG_T() - hd_c;
        }
    };
    // in-plane coordinates of the boundary pixels of the wave's stretches, and how many there are
    float u[kHullChunks], v[kHullChunks];
    const unsigned long long hd_a = HDBG_T();
#pragma unroll
    for (int chunk = 0; chunk < kHullChunks; ++chunk) {
        const int slot = slots[chunk];
        const bool bnd = slot >= 0;
        const unsigned long long mask = __ballot(bnd);
        u[chunk] = v[chunk] = 0.f;
        if (lane == 0) cnt[wave_u * kHullChunks + chunk] = __builtin_popcountll(mask);
        if (mask == 0ull) continue;
        if (bnd) {
            float f9[9];
            if (slot < n_frames_lds) {
#pragma unroll
                for (int q = 0; q < 9; ++q) f9[q] = fr[slot][q];
            } else {
#pragma unroll
                for (int q = 0; q < 9; ++q) f9[q] = reinterpret_cast<const float*>(frames + slot)[q];
            }
            const float dx = px[chunk] - f9[0], dy = py[chunk] - f9[1], dz = pz[chunk] - f9[2];
            u[chunk] = dx * f9[3] + dy * f9[4] + dz * f9[5];
            v[chunk] = dx * f9[6] + dy * f9[7] + dz * f9[8];
        }
    }
#ifdef RGBD360_HULL_DBG
#pragma unroll
    for (int chunk = 0; chunk < kHullChunks; ++chunk) asm volatile("" :: "v"(u[chunk]), "v"(v[chunk]));
    hd_uv += HDBG_T() - hd_a;
#endif
    __syncthreads();
    // The walk is serial per wave, and the boundary pixels are where the edges are: one wave of a block met a fragmented corner (18 regions
    // for 20 pixels, ~1500 cycles per change of region for a LONE wave) or a 64-pixel stretch of a horizontal edge while its fifteen
    // neighbours waited at the barrier (tools/hull_stamps.py: 31 k cycles against 5 k).  So the block pools its boundary pixels in LDS --
    // in the fixed order (wave, stretch, lane): offsets from a prefix sum over the 128 counts, no arrival order anywhere, results stay
    // reproducible bit for bit -- and every wave walks an equal share of the list (in rounds of kHullList entries).
    static_assert((kHullBlock / 64) * kHullChunks == 128, "two counts per lane");
    int c_lo = cnt[lane], c_hi = cnt[64 + lane];
    const int s_lo = wave_scan_add(c_lo), s_hi = wave_scan_add(c_hi);
    const int tot_lo = __builtin_amdgcn_readlane(s_lo, 63);
    const int total = tot_lo + __builtin_amdgcn_readlane(s_hi, 63);
    const int ex_lo = s_lo - c_lo, ex_hi = s_hi - c_hi + tot_lo;       // exclusive prefix of flat index lane / 64 + lane
    int idx_of[kHullChunks];                               // this lane's place in the block's list (boundary pixels only)
#pragma unroll
    for (int chunk = 0; chunk < kHullChunks; ++chunk) {
        const int f = wave_u * kHullChunks + chunk;        // wave-uniform
        const int base = f < 64 ? __builtin_amdgcn_readlane(ex_lo, f & 63) : __builtin_amdgcn_readlane(ex_hi, f & 63);
        const unsigned long long mask = __ballot(slots[chunk] >= 0);
        idx_of[chunk] = base + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
    }
    // (a list of kHullList entries per round: one round, unless more than 8 % of the block's 8192 pixels are boundary pixels)
    for (int lo = 0; lo < total; lo += kHullList) {        // block-uniform
        if (lo > 0) __syncthreads();                       // the previous round's list has been walked
#pragma unroll
        for (int chunk = 0; chunk < kHullChunks; ++chunk) {
            const int idx = idx_of[chunk] - lo;
            if (slots[chunk] >= 0 && idx >= 0 && idx < kHullList) {
                HullEntry e;
                e.slot = slots[chunk]; e.u = u[chunk]; e.v = v[chunk]; e.pix = first_of[chunk] + lane;
                list[idx] = e;
            }
        }
        __syncthreads();
        const int E = min(total - lo, kHullList);
        const int per = (E + kHullBlock / 64 - 1) / (kHullBlock / 64);
        const int begin = wave_u * per, end = min(E, begin + per);
        for (int b0 = begin; b0 < end; b0 += 64) {          // wave-uniform
            const int e = b0 + lane;
            HullEntry en;
            en.slot = -1; en.u = 0.f; en.v = 0.f; en.pix = 0;
            if (e < end) en = list[e];
            walk(en.slot, en.u, en.v, en.pix);
        }
    }
    const unsigned long long hd_t2 = HDBG_T();
    __syncthreads();
    // The block's table goes to memory as it is -- rows of its own, no atomics: 256 blocks x 6 walls x 256 directions of global
    // atomicMax on 33 rows were 23 of the kernel's 27 us (same-address contention at the memory side).  k_f360_hull_merge takes the
    // maximum over the blocks' rows of a slot.  (Runs that found the block's table full went to `ext` directly: also read there.)
    if (threadIdx.x < kHullHash) part_keys[blockIdx.x * kHullHash + threadIdx.x] = keys[threadIdx.x];
    for (int k = threadIdx.x; k < kHullHash * kHullDirs; k += kHullBlock) {
        const int h = k / kHullDirs;
        if (keys[h] >= 0) part_vals[(size_t)blockIdx.x * kHullHash * kHullDirs + k] = vals[h][k - h * kHullDirs];
    }
#ifdef RGBD360_HULL_DBG
    if (lane == 0) {
        unsigned long long* o = g_hull_dbg[(blockIdx.x * (kHullBlock / 64) + (threadIdx.x >> 6)) & 4095];
        o[0] = hd_t1 - hd_t0; o[1] = hd_t2 - hd_t1; o[2] = HDBG_T() - hd_t2; o[3] = hd_uv; o[4] = hd_walk; o[5] = hd_flush; o[6] = hd_entries; o[7] = hd_runs;
    }
#endif
}
// one block per CU (81 KB of LDS, 95 VGPRs) ...
__global__ __launch_bounds__(kHullBlock) void k_f360_hull_extremes(const float* __restrict__ xyz, const int* __restrict__ label,
                                                                    const int* __restrict__ slot_of_root, int rows, int cols,
                                                                    const SlotFrame* __restrict__ frames, unsigned long long* __restrict__ ext,
                                                                    int* __restrict__ part_keys, unsigned long long* __restrict__ part_vals, int n_frames_lds) {
    hull_extremes_block(xyz, label, slot_of_root, rows, cols, frames, ext, part_keys, part_vals, n_frames_lds);
}
// ... or two (64 VGPRs, 21 of them spilled): where a frame has more blocks than the chip has CUs (4096 x 2048: four per CU) the second
// block's round trips run under the first one's walk, 50 -> 40 us; with one block per CU the spills only cost (16.3 -> 17.6 us)
__global__ __launch_bounds__(kHullBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_f360_hull_extremes_two(const float* __restrict__ xyz, const int* __restrict__ label,
                                                                    const int* __restrict__ slot_of_root, int rows, int cols,
                                                                    const SlotFrame* __restrict__ frames, unsigned long long* __restrict__ ext,
                                                                    int* __restrict__ part_keys, unsigned long long* __restrict__ part_vals, int n_frames_lds) {
    hull_extremes_block(xyz, label, slot_of_root, rows, cols, frames, ext, part_keys, part_vals, n_frames_lds);
}
// The blocks' tables -> the global rows of extremes.  A wall met by every block has a 2 KB row in each table (256 at 2048 x 1024, 1024 at
// 4096 x 2048): ONE block streaming them is bound by what a single block reads (~30-60 GB/s: 19 / 75 us per frame when the packing
// kernel did it, and anything that put a dependent look-up in front of each row load made every row a full memory round trip:
// 160 us).  Here kHullMergeSplit blocks share a slot's rows -- block (slot, m) takes the tables of the blocks b = m mod 8 -- and meet
// in the global row with one atomicMax per direction (8 adders per word).
constexpr int kHullMergeSplit = 16;     // (8 until the tables had 64 rows: the key scan and the row loop per merge block doubled)
__global__ __launch_bounds__(kHullDirs) void k_f360_hull_merge(const int* __restrict__ n_slots, int max_slots, const int* __restrict__ part_keys,
                                                                const unsigned long long* __restrict__ part_vals, int n_blocks,
                                                                unsigned long long* __restrict__ ext) {
    const int k = threadIdx.x, m = blockIdx.y;
    const int ns = min(*n_slots, max_slots);
    __shared__ int n_match;
    __shared__ int match[1024];                            // table rows of this block's share that belong to the slot
    for (int slot = blockIdx.x; slot < ns; slot += gridDim.x) {
        if (k == 0) n_match = 0;
        __syncthreads();
        // keys of the tables b = m, m + 8, ...: 16 per table
        const int n_mine = (n_blocks - m + kHullMergeSplit - 1) / kHullMergeSplit;
        for (int e = k; e < n_mine * kHullHash; e += kHullDirs) {
            const int at = (m + (e / kHullHash) * kHullMergeSplit) * kHullHash + (e & (kHullHash - 1));
            const bool hit = part_keys[at] == slot;
            const unsigned long long hits = __ballot(hit);
            if (hits != 0ull) {                            // one LDS atomic per wave with a hit
                int base = 0;
                if ((k & 63) == (int)__builtin_ctzll(hits)) base = atomicAdd(&n_match, __builtin_popcountll(hits));
                base = __builtin_amdgcn_readlane(base, __builtin_ctzll(hits));
                if (hit) {
                    const int q = base + __builtin_popcountll(hits & ((1ull << (k & 63)) - 1ull));
                    if (q < 1024) match[q] = at;
                }
            }
        }
        __syncthreads();
        const int nm = min(n_match, 1024);
        unsigned long long key = 0ull;
        for (int q = 0; q < nm; ++q) {                     // (plain loop: the row loads pipeline)
            const unsigned long long cand = part_vals[(size_t)match[q] * kHullDirs + k];
            key = cand > key ? cand : key;
        }
        if (key != 0ull) atomicMax(&ext[((size_t)slot * kHullPhases + (m & (kHullPhases - 1))) * kHullDirs + k], key);      // (the tables of the blocks b = m mod 8 are those of direction set m)
        __syncthreads();
    }
}
// (Until round 5 the last block of this kernel -- a device ticket -- could publish the host tag itself instead of a tag launch behind it:
// 256 blocks each running a system-scope fence cost 13 us where the tag kernel costs 4; measured again at both sizes, removed.)
__global__ __launch_bounds__(kHullDirs) void k_f360_hull_pack(const float* __restrict__ xyz, const SlotFrame* __restrict__ frames,
                                                               const unsigned long long* __restrict__ ext, const int* __restrict__ n_slots,
                                                               int max_slots, F360HullRecord* __restrict__ out) {
    static_assert(kHullMergeSplit % kHullPhases == 0, "the merge blocks of a slot pair up with the direction sets");
    const int k = threadIdx.x;
    const int ns = min(*n_slots, max_slots);
    __shared__ int pix_sh[kHullDirs * kHullPhases];
    __shared__ int wave_cnt[kHullDirs / 64];
    for (int slot = blockIdx.x; slot < ns; slot += gridDim.x) {      // (a block per slot of the 4096 possible ones cost 9 us of empty launches)
        // thread k owns the directions 4 k + p, p = 0 .. 3 (set p's direction k): consecutive in angle
        int pix[kHullPhases];
#pragma unroll
        for (int p = 0; p < kHullPhases; ++p) {
            const unsigned long long key = ext[((size_t)slot * kHullPhases + p) * kHullDirs + k];
            pix[p] = key != 0ull ? (int)(unsigned)(key & 0xFFFFFFFFull) : -1;
            pix_sh[k * kHullPhases + p] = pix[p];
        }
        __syncthreads();
        // a point is sent once per set: dropped when the set's previous direction found the same pixel (a corner wins hundreds; between
        // the sets duplicates are rare -- they saw different pixels -- and the host's hull removes them)
        bool keep[kHullPhases];
        int mine = 0;
#pragma unroll
        for (int p = 0; p < kHullPhases; ++p) {
            const int prev = k > 0 ? pix_sh[(k - 1) * kHullPhases + p] : -1;
            keep[p] = pix[p] >= 0 && pix[p] != prev;
            mine += keep[p] ? 1 : 0;
        }
        // exclusive prefix over the block: wave scan + wave totals
        const int incl = wave_scan_add(mine);
        if ((k & 63) == 63) wave_cnt[k >> 6] = incl;
        __syncthreads();
        int base = incl - mine, total = 0;
#pragma unroll
        for (int w = 0; w < kHullDirs / 64; ++w) {
            if (w < (k >> 6)) base += wave_cnt[w];
            total += wave_cnt[w];
        }
        const SlotFrame F = frames[slot];
#pragma unroll
        for (int p = 0; p < kHullPhases; ++p) {
            if (!keep[p]) continue;
            const size_t px = (size_t)pix[p];
            const float dx = xyz[3 * px] - F.c[0], dy = xyz[3 * px + 1] - F.c[1], dz = xyz[3 * px + 2] - F.c[2];
            out[slot].uv[base][0] = dx * F.e1[0] + dy * F.e1[1] + dz * F.e1[2];
            out[slot].uv[base][1] = dx * F.e2[0] + dy * F.e2[1] + dz * F.e2[2];
            ++base;
        }
        if (k < 3) {
            out[slot].c[k] = F.c[k];
            out[slot].e1[k] = F.e1[k];
            out[slot].e2[k] = F.e2[k];
        }
        if (k == 0) out[slot].n = total;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// The `refine` half of pcl::OrganizedMultiPlaneSegmentation::segmentAndRefine (Frame360.h:977): plane regions grow into
// neighbouring non-plane pixels that lie within the refinement comparator's distance of the plane -- in PCL two raster passes
// with in-place label updates (oracle/frame360_ref.cpp restates them literally).  A raster pass is a recurrence over a DAG: the
// label a pixel ends a pass with is a function of its own label before the pass and of the END-OF-PASS labels of the two
// neighbours the raster visits before it (pass 1: upper, then left; pass 2: lower, then right -- the order in which the
// sequential loop offers them; an accepted offer makes the pixel a plane pixel, which refuses later offers).  The device solves
// the recurrence by fixed-point iteration: a sweep recomputes every free pixel from the previous sweep's labels of its two
// predecessors; a pixel is final once its predecessors are, so the iteration ends in the unique fixed point = the labels of the
// sequential pass, whatever the execution order.  (First version: full-image Jacobi sweeps, one per pixel of the longest growth chain
// -- 804 sweeps = 17 ms on a 2048 x 1024 scene with noisy patches; now block-Jacobi over tiles with local iteration in LDS.)
// models[slot] = {a, b, c, d} of the planes `segment` produced, x = NaN for regions that are no plane (too few inliers, curvature).
// ---------------------------------------------------------------------------------------------------------
// Work labels of the refinement: the plane's slot (>= 0) for pixels of a region that became a plane, -1 for non-finite points,
// -2 for valid pixels of regions that did not ("free" pixels: the only ones that can change).
constexpr int kRefInvalid = -1, kRefFree = -2;
constexpr int kRefTH = 16, kRefWaves = 4;
#ifndef F360_REFINE_SLEEP
#define F360_REFINE_SLEEP 8
#endif
// W2: the iterate the tile launches relax in place starts as a copy of the pass labels; counters: {relabelled pixels, activity}, cleared here
// (the copy and the two clears were commands of their own: ~5 us each on the stream of a call that takes 0.3 ms on a sensor image)
__global__ void k_f360_refine_init(const int* __restrict__ label, const int* __restrict__ slot_of_root, const float4* __restrict__ models, int n,
                                   int cols, int tiles_x, int* __restrict__ W, int* __restrict__ W2, unsigned char* __restrict__ tile_free,
                                   int* __restrict__ counters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) counters[0] = counters[1] = 0;
    if (i >= n) return;
    const int l = label[i];
    int w = kRefInvalid;
    if (l >= 0) {
        const int slot = slot_of_root[l];
        w = kRefFree;
        if (slot >= 0) {
            const float4 m = models[slot];
            if (m.x == m.x) w = slot;
        }
    }
    W[i] = w;
    W2[i] = w;
    if (w == kRefFree) {                       // only tiles with free pixels have anything to do
        const int r = i / cols, c = i - r * cols;
        tile_free[(r / kRefTH) * tiles_x + (c >> 6)] = 1;
    }
}
// One block-Jacobi step of a raster pass over 64 x 16 tiles, ONE WAVE per tile, exact inside the tile in a single sweep: lane =
// column, the rows are walked in the pass's order with the previous row's final labels in a register.  Per row the offer from the
// previous row is a lane-local test; the propagation ALONG the row (in PCL a pixel-by-pixel chain) is closed-form per wave: a free
// pixel can only take the label L of the nearest non-free pixel s on the side the raster comes from (a plane pixel -- or the tile's
// ring pixel when there is none), and it does take it iff every free pixel between s and itself lies within the threshold of plane L --
// one ballot of "within the threshold of my candidate" and a mask of the lanes between s and the pixel decide it.  All of the tile's
// inputs (pass labels, points) are loaded into registers up front; the ring (the neighbouring tiles' labels) comes from the previous
// global iterate, so a growth chain costs one global step per TILE it crosses, not one per pixel (804 full-image sweeps = 17 ms on
// the 2048 x 1024 test scene with the first version).  The fixed point of the scheme is the unique solution of the raster recurrence,
// i.e. PCL's sequential pass.
__device__ __forceinline__ int wave_read_lane(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
// labels other tiles may be rewriting while this kernel runs: device-coherent accesses (the L2s of the eight XCDs are not coherent
// with each other for plain loads and stores inside a kernel)
__device__ __forceinline__ int coherent_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// X is updated IN PLACE: a neighbouring tile's ring pixel may be read while that tile rewrites it -- either value is a valid iterate
// of this asynchronous relaxation, whose only fixed point is the recurrence's solution.  A wave does not leave after its step: it
// POLLS its ring (16 + 64 labels) up to n_polls times and repeats the step whenever the ring has changed, so a growth chain crosses
// many tiles inside one launch (a launch per tile crossed: 56 launches = 1.3 ms on the 2048 x 1024 noisy scene).  Polling is bounded
// (no wave waits for another: nothing depends on residency) and ends early once `activity` -- a counter every changing step bumps
// -- has stood still for a few polls.  A launch in which no step changed anything has read nothing but final values (its first step
// sees everything earlier launches wrote): that is the host's convergence test.  Tiles without free pixels (tile_free) do nothing.
// LDS_MODELS: the plane models come out of LDS (n_lds_models = all slots) -- a template argument, not a run-time choice: with
// `n_lds_models > 0 ? s_models[l] : models[l]` the compiler formed ONE pointer and a FLAT load, whose wait is vmcnt(0) + lgkmcnt(0),
// i.e. every row of a step also waited for the device-scope stores of the rows before it to be acknowledged by memory: 2.7 us per
// row that changed a label, 44-90 us per step (per-wave clocks) -- the whole cost of the refinement's launches.
template <int PASS, bool LDS_MODELS>
__global__ __launch_bounds__(64 * kRefWaves) void k_f360_refine_tile(const float* __restrict__ xyz, const int* __restrict__ W0, int* X,
                                                                    const float4* __restrict__ models, float thr, int rows, int cols,
                                                                    int tiles_y, const unsigned char* __restrict__ tile_free,
                                                                    int* __restrict__ changed, int* activity, int n_polls, int n_quiet,
                                                                    int n_lds_models) {
    // the plane models in LDS (when they fit): a step looks two of them up per row, each row depends on the one before, and a chain
    // of 32 L1/L2 round trips per step was what a tile hop cost (~20 us)
    extern __shared__ float4 s_models[];
    for (int k = threadIdx.x; k < n_lds_models; k += blockDim.x) s_models[k] = models[k];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int tyi = blockIdx.y * kRefWaves + (threadIdx.x >> 6);
    if (tyi >= tiles_y) return;
    if (!tile_free[tyi * gridDim.x + blockIdx.x]) return;
    const int c0 = blockIdx.x * 64, r0 = tyi * kRefTH;
    const int c = c0 + lane;
    const bool col_in = c < cols;
    // the whole tile column of this lane into registers (one memory round trip)
    int w0[kRefTH], cur[kRefTH];
    float px[kRefTH], py[kRefTH], pz[kRefTH];
#pragma unroll
    for (int k = 0; k < kRefTH; ++k) {
        const int r = r0 + k;
        const bool in = col_in && r < rows;
        const size_t i = (size_t)(in ? r : 0) * cols + (in ? c : 0);
        w0[k] = in ? W0[i] : kRefInvalid;
        cur[k] = in ? X[i] : kRefInvalid;
        px[k] = xyz[3 * i]; py[k] = xyz[3 * i + 1]; pz[k] = xyz[3 * i + 2];
    }
    // the ring: lane k < 16 holds the label beside row k on the side the row propagation comes from, every lane the label of its
    // column in the row the pass comes from
    const int side_c = PASS == 1 ? c0 - 1 : c0 + 64;
    const int rprev = PASS == 1 ? r0 - 1 : r0 + kRefTH;
    const bool side_in = lane < kRefTH && r0 + lane < rows && side_c >= 0 && side_c < cols;
    const bool prev_in = col_in && rprev >= 0 && rprev < rows;
    const int* side_p = X + (size_t)(side_in ? r0 + lane : 0) * cols + (side_in ? side_c : 0);
    const int* prev_p = X + (size_t)(prev_in ? rprev : 0) * cols + (prev_in ? c : 0);
    // validity of the previous row's pixel one column further (the coupling of the two checks in PCL's loop body): for the ring row
    // from memory, afterwards from the lanes -- pass labels, static
    const int cc = PASS == 1 ? c + 1 : c - 1;
    const bool ring_side_valid = (cc >= 0 && cc < cols && rprev >= 0 && rprev < rows) ? (W0[(size_t)rprev * cols + cc] != kRefInvalid) : false;
    bool edge_valid[kRefTH];                   // the tile-edge lane's neighbour in the next tile (pass labels, static)
    {
        // lane k loads the label beside row k, the rows read it with v_readlane: ONE load (as sixteen conditional loads of a uniform
        // address the compiler issued them one after the other, each behind its own wait: 15 memory round trips, ~20 us, before a
        // wave's first step)
        const int edge_c = PASS == 1 ? c0 + 64 : c0 - 1;
        const bool edge_in = lane < kRefTH && edge_c >= 0 && edge_c < cols && r0 + lane < rows;
        const int edge_label = W0[(size_t)(edge_in ? r0 + lane : 0) * cols + (edge_in ? edge_c : 0)];
        const int edge_ok = (edge_in && edge_label != kRefInvalid) ? 1 : 0;
#pragma unroll
        for (int k = 0; k < kRefTH; ++k) edge_valid[k] = __builtin_amdgcn_readlane(edge_ok, k) != 0;
    }
    int ring_side = side_in ? coherent_load(side_p) : kRefInvalid;
    int ring_prev = prev_in ? coherent_load(prev_p) : kRefInvalid;
    bool any_change = false;

    auto step = [&]() {                        // one exact sweep of the tile against the ring held in ring_side / ring_prev
        // everything the rows read has ARRIVED before the first row: a row that met the ring's loaded value first inside its
        // conditional part got a vmcnt(0) there -- which also waits for the label store of the row before (a device-scope store is
        // acknowledged by memory: ~1 us) -- in every row, since the rows before may have skipped theirs
        asm volatile("" ::"v"(ring_side), "v"(ring_prev));
        int prev = ring_prev;
        bool prev_side_valid = ring_side_valid;
        bool chg = false;
#pragma unroll
        for (int kk = 0; kk < kRefTH; ++kk) {
            const int k = PASS == 1 ? kk : kRefTH - 1 - kk;
            const int r = r0 + k;
            int state = w0[k];
            const bool is_free = state == kRefFree;
            auto within = [&](int label) {        // PlaneRefinementComparator::compare's distance test of this pixel against plane `label`
                const int l = label >= 0 ? label : 0;
                float4 m;
                if (LDS_MODELS) m = s_models[l];
                else m = models[l];
                const double ptp_dist = fabs(m.x * px[k] + m.y * py[k] + m.z * pz[k] + m.w);
                return label >= 0 && ptp_dist < (double)thr;
            };
            // offer from the previous row
            if (PASS == 1) {
                if (is_free && r >= 1 && c <= cols - 2 && prev_side_valid && within(prev)) state = prev;
            } else {
                if (is_free && r + 1 <= rows - 1 && (c == 0 || prev_side_valid) && within(prev)) state = prev;
            }
            // propagation along the row
            const bool still_free = state == kRefFree;
            const unsigned long long nonfree = __ballot(!still_free);
            // (a row without a free pixel left -- most rows of most tiles, and every row a flood from the previous row has filled -- has
            // nothing to propagate: the search, the crossbar read and the second test are skipped; wave-uniform)
            if (nonfree != ~0ull) {
            int s;                                 // lane of the nearest non-free pixel on the incoming side, -1 / 64: the ring pixel
            if (PASS == 1) {
                const unsigned long long m = nonfree & ((1ull << lane) - 1ull);
                s = m ? 63 - __builtin_clzll(m) : -1;
            } else {
                const unsigned long long m = lane == 63 ? 0ull : (nonfree & ~((2ull << lane) - 1ull));
                s = m ? __builtin_ctzll(m) : 64;
            }
            const int from_lane = wave_read_lane(state, s & 63);
            const int ringside_k = __builtin_amdgcn_readlane(ring_side, k);
            const int cand = (s < 0 || s > 63) ? ringside_k : from_lane;
            const bool row_ok = PASS == 1 ? (c >= 1 && r <= rows - 2) : (c + 1 <= cols - 1 && r >= 1);
            const bool ok = still_free && col_in && within(cand);
            const unsigned long long okm = __ballot(ok);
            if (still_free && row_ok && cand >= 0) {
                unsigned long long range;          // the lanes strictly between s and this one, and this one
                if (PASS == 1) range = ((2ull << lane) - 1ull) & ~(s < 0 ? 0ull : ((2ull << s) - 1ull));
                else range = (s > 63 ? ~0ull : ((1ull << s) - 1ull)) & ~((1ull << lane) - 1ull);
                // every pixel of the chain must be allowed to hand the label on: the chain runs inside rows 0 .. H-2 (pass 1) / 1 .. H-1 (pass 2)
                // for all of them alike, and column limits only bind at the image border, where the chain starts
                if ((range & ~okm) == 0ull) state = cand;
            }
            }
            if (col_in && r < rows && state != cur[k]) {
                chg = true;
                cur[k] = state;
                coherent_store(X + (size_t)r * cols + c, state);
            }
            // this row is the next one's "previous row"
            const int nb = PASS == 1 ? wshl1i(w0[k]) : wshr1i(w0[k]);          // pass labels of the neighbouring lane: validity is static
            bool side_valid = nb != kRefInvalid;
            if ((PASS == 1 && lane == 63) || (PASS == 2 && lane == 0)) side_valid = edge_valid[k];
            prev_side_valid = side_valid;
            prev = state;
        }
        return __ballot(chg) != 0ull;
    };

    if (step()) {
        any_change = true;
        if (lane == 0) atomicAdd(activity, 1);
    }
    int seen = coherent_load(activity), quiet = 0;
    const int first_seen = seen;
    for (int p = 0; p < n_polls; ++p) {
        __builtin_amdgcn_s_sleep(F360_REFINE_SLEEP);
        const int rs = side_in ? coherent_load(side_p) : kRefInvalid;
        const int rp = prev_in ? coherent_load(prev_p) : kRefInvalid;
        const int act = coherent_load(activity);
        if (__ballot(rs != ring_side || rp != ring_prev) != 0ull) {
            ring_side = rs; ring_prev = rp;
            if (step()) {
                any_change = true;
                if (lane == 0) atomicAdd(activity, 1);
#ifdef F360_REFINE_TILE_DBG
                if (lane == 0) { unsigned char* tf = const_cast<unsigned char*>(tile_free) + tyi * gridDim.x + blockIdx.x; if (*tf < 250) *tf += 1; }
#endif
            }
            quiet = 0;
            seen = coherent_load(activity);
            continue;
        }
        if (act == seen) {
            // nobody has changed anything for a while: the launch is over for this tile (a launch in which nothing has moved at
            // all -- the host's verification launch, a frame with nothing to grow -- is given up four times sooner)
            if (++quiet >= (seen == first_seen && !any_change ? max(n_quiet / 4, 1) : n_quiet)) break;
        } else {
            seen = act;
            quiet = 0;
        }
    }
    if (any_change && lane == 0) *changed = 1;
}
// The grown inliers join their plane's sums: count and the nine integer moments (order independent), labels updated.  The sums of a
// block are collected in an LDS hash first (a noisy patch sends thousands of pixels to ONE plane: 1.7 M same-address global atomics
// took 5 ms), one global atomic per block, slot and sum follows.
constexpr int kCommitThreads = 1024, kCommitPerThread = 8;      // a block owns 8192 consecutive pixels (round 4: 256-pixel blocks flushed their table
                                                                 // with 11 global atomics per plane each -- 8192 blocks on a handful of words: 50-190 us)
__global__ __launch_bounds__(kCommitThreads) void k_f360_refine_commit(const float* __restrict__ xyz, int* __restrict__ label, const int* __restrict__ Winit,
                                                            const int* __restrict__ Wfinal, const int* __restrict__ root_of_slot, int n,
                                                            int* __restrict__ count_of_slot, unsigned long long* __restrict__ mom, int max_slots,
                                                            int* __restrict__ n_changed) {
    unsigned long long* mom_rep = mom + (size_t)(blockIdx.x % kMomReplicas) * max_slots * 9;      // any copy will do: k_f360_mom_reduce sums them
    __shared__ int keys[kMomRunHash];
    __shared__ unsigned long long vals[kMomRunHash][10];
    const int lane = threadIdx.x & 63;
    const int base = blockIdx.x * kCommitThreads * kCommitPerThread;
    int slot[kCommitPerThread];
    bool grown[kCommitPerThread];
    bool any = false;
#pragma unroll
    for (int j = 0; j < kCommitPerThread; ++j) {
        const int i = base + j * kCommitThreads + (int)threadIdx.x;
        slot[j] = i < n ? Wfinal[i] : kRefInvalid;
        grown[j] = i < n && slot[j] != Winit[i];           // only free pixels change, and only into a plane's slot
        any |= grown[j];
    }
    if (!__syncthreads_or(any ? 1 : 0)) return;            // most blocks of a clean frame hold no grown pixel
    if (threadIdx.x < kMomRunHash) {
        keys[threadIdx.x] = -1;
#pragma unroll
        for (int q = 0; q < 10; ++q) vals[threadIdx.x][q] = 0ull;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kCommitPerThread; ++j) {
        const unsigned long long gm = __ballot(grown[j]);
        if (gm == 0ull) continue;                          // wave-uniform
        const int i = base + j * kCommitThreads + (int)threadIdx.x;
        unsigned long long v[10] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
        if (grown[j]) {
            label[i] = root_of_slot[slot[j]];
            auto d2ll = [](double v) -> long long { return __double_as_longlong(v + 6755399441055744.0) - 0x4338000000000000LL; };
            const double x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
            v[0] = (unsigned long long)d2ll(x * kMomScale); v[1] = (unsigned long long)d2ll(y * kMomScale); v[2] = (unsigned long long)d2ll(z * kMomScale);
            v[3] = (unsigned long long)d2ll(x * x * kMomScale); v[4] = (unsigned long long)d2ll(x * y * kMomScale);
            v[5] = (unsigned long long)d2ll(x * z * kMomScale); v[6] = (unsigned long long)d2ll(y * y * kMomScale);
            v[7] = (unsigned long long)d2ll(y * z * kMomScale); v[8] = (unsigned long long)d2ll(z * z * kMomScale);
            v[9] = 1ull;
        }
        // a noisy patch sends whole waves to ONE plane: those are summed across the wave first (sums of integers: order free) and
        // enter the hash once; 64 lanes x 10 atomics on one LDS address each took most of the first version's 118 us
        const int first = __builtin_ctzll(gm);
        const int s0 = __builtin_amdgcn_readlane(slot[j], first);
        const bool one_plane = __ballot(grown[j] && slot[j] != s0) == 0ull;
        if (one_plane) {
#pragma unroll
            for (int q = 0; q < 10; ++q) v[q] = wave_sum_ll(v[q]);
        }
        if (one_plane ? lane == first : grown[j]) {
            const int h = mom_run_slot(keys, slot[j]);
            if (h >= 0) {
#pragma unroll
                for (int q = 0; q < 10; ++q) atomicAdd(&vals[h][q], v[q]);
            } else {                                       // hash full (more than 256 planes meet in one block): straight to memory
#pragma unroll
                for (int q = 0; q < 9; ++q) atomicAdd(&mom_rep[(size_t)slot[j] * 9 + q], v[q]);
                atomicAdd(&count_of_slot[slot[j]], (int)v[9]);
                atomicAdd(n_changed, (int)v[9]);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < kMomRunHash) {
        const int hs = keys[threadIdx.x];
        if (hs >= 0) {
#pragma unroll
            for (int q = 0; q < 9; ++q) atomicAdd(&mom_rep[(size_t)hs * 9 + q], vals[threadIdx.x][q]);
            atomicAdd(&count_of_slot[hs], (int)vals[threadIdx.x][9]);
            atomicAdd(n_changed, (int)vals[threadIdx.x][9]);
        }
    }
}

// Frame360::stitchImage (Frame360.h:1099-1148): one thread per panorama pixel; the sensor is fixed by the column band.
// sin/cos tables of the row / column angles come from the host's libm (the reference evaluates them per row / pixel).
struct StitchArgs {
    float Rt_inv[8][16];     // column-major 4x4 per sensor
    float fx, fy, cx, cy;
    int   sensor_rows, sensor_cols, W, H;
};
__global__ void k_stitch_sphere(StitchArgs a, const uint8_t* __restrict__ rgb /*[8][rows][cols][3]*/,
                                const uint16_t* __restrict__ depth /*[8][rows][cols]*/, const float* __restrict__ sin_phi,
                                const float* __restrict__ cos_phi, const float* __restrict__ sin_theta,
                                const float* __restrict__ cos_theta, uint8_t* __restrict__ sphereRGB, uint16_t* __restrict__ sphereDepth) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (col >= a.W || row >= a.H) return;
    const int sensor = 7 - col / a.sensor_rows;
    const float* M = a.Rt_inv[sensor];
    const float v0 = sin_phi[row];
    const float v1 = cos_phi[row] * sin_theta[col];
    const float v2 = cos_phi[row] * cos_theta[col];
    float p[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = ((M[0 * 4 + i] * v0 + M[1 * 4 + i] * v1) + M[2 * 4 + i] * v2) + M[3 * 4 + i];
    const float u = a.fx * p[0] / p[2] + a.cx;
    const float v = a.fy * p[1] / p[2] + a.cy;
    const size_t o = (size_t)row * a.W + col;
    uint8_t r = 0, g = 0, b = 0;
    uint16_t d = 0;
    if (u >= 0 && u < a.sensor_cols && v >= 0 && v < a.sensor_rows) {
        const int ui = (int)u, vi = (int)v;
        const size_t in = ((size_t)sensor * a.sensor_rows + vi) * a.sensor_cols + ui;
        r = rgb[3 * in]; g = rgb[3 * in + 1]; b = rgb[3 * in + 2];
        // range = depth * sqrt(1 + x^2 + y^2) in double, truncated to unsigned short (Frame360.h:1141)
        const double xn = (double)((u - a.cx) / a.fx), yn = (double)((v - a.cy) / a.fy);
        d = (uint16_t)((double)depth[in] * sqrt(1 + xn * xn + yn * yn));
    }
    sphereRGB[3 * o] = r; sphereRGB[3 * o + 1] = g; sphereRGB[3 * o + 2] = b;
    sphereDepth[o] = d;
}

// ---- pcl::FastBilateralFilter (bilateral grid) -----------------------------------------------------------------------
// The smoothing Frame360 applies to every sensor cloud before the planes are segmented (Frame360.h:40, 493-499: sigma_s 10 px,
// sigma_r 0.05 m).  Third-party algorithm (PCL filters/impl/fast_bilateral.hpp), restated in oracle/frame360_ref.cpp; the
// kernels repeat the oracle's float operations one for one and accumulate the cell sums as integers (2^-20 m), so the filtered
// cloud is bit-identical to the oracle's whatever the order of the atomics.  Small data (a 160 x 120 sensor cloud has a grid of
// ~20 x 16 x 100 cells): every kernel here is latency-bound; the point of running them on the device is that the cloud stays there.
struct BilatGrid {
    int nx, ny, nz;
    float sigma_s, sigma_r, base_min, base_max;
};
constexpr int kBilatPadXY = 2, kBilatPadZ = 2;
constexpr double kBilatFixed = 1048576.0;

__device__ __forceinline__ unsigned bilat_encode(float v) {          // order-preserving float -> unsigned
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// {min, max} of the finite z (encoded) per block into blk[2 b], blk[2 b + 1] ({0xffffffff, 0}: none); k_bilat_minmax_publish folds the
// blocks and hands the pair to the host.  (Until round 5 every wave ran an atomicMin and an atomicMax on one pair of words behind two
// memsets: 300 same-address atomics = 25 us for a 160 x 120 cloud, then a copy and a stream synchronisation.)
constexpr int kBilatMmThreads = 1024;
__global__ __launch_bounds__(kBilatMmThreads) void k_bilat_minmax(const float* __restrict__ xyz, int n, unsigned* __restrict__ blk) {
    __shared__ unsigned red[kBilatMmThreads / 64][2];
    unsigned lo = 0xffffffffu, hi = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float z = xyz[3 * (size_t)i + 2];
        if (isfinite(z)) {
            const unsigned e = bilat_encode(z);
            lo = min(lo, e);
            hi = max(hi, e);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, m));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, m));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = lo; red[threadIdx.x >> 6][1] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBilatMmThreads / 64; ++w) { lo = min(lo, red[w][0]); hi = max(hi, red[w][1]); }
        blk[2 * blockIdx.x] = lo;
        blk[2 * blockIdx.x + 1] = hi;
    }
}
// one wave: the blocks' pairs -> the pair in pinned host memory, then the host's tag (host_wait.h)
__global__ __launch_bounds__(64) void k_bilat_minmax_publish(const unsigned* __restrict__ blk, int nblk, unsigned* __restrict__ host_mm, unsigned* tag, unsigned seq) {
    unsigned lo = 0xffffffffu, hi = 0u;
    for (int b = threadIdx.x; b < nblk; b += 64) { lo = min(lo, blk[2 * b]); hi = max(hi, blk[2 * b + 1]); }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, m));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, m));
    }
    if (threadIdx.x == 0) { host_mm[0] = lo; host_mm[1] = hi; }
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store(tag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ size_t bilat_idx(const BilatGrid& g, int x, int y, int z) { return (((size_t)x * g.ny) + y) * g.nz + z; }

__global__ void k_bilat_scatter(const float* __restrict__ xyz, int rows, int cols, BilatGrid g, unsigned long long* __restrict__ sum,
                                int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = i < rows * cols;
    const int ic = on ? i : 0;
    const int y = ic / cols, x = ic - y * cols;
    float pz = xyz[3 * (size_t)ic + 2];
    if (!isfinite(pz)) pz = g.base_max;
    const float z = pz - g.base_min;
    const int sx = (int)((float)x / g.sigma_s + 0.5f) + kBilatPadXY, sy = (int)((float)y / g.sigma_s + 0.5f) + kBilatPadXY;
    const int sz = (int)(z / g.sigma_r + 0.5f) + kBilatPadZ;
    const int c = (int)bilat_idx(g, sx, sy, sz);                   // (cells < 64e6, checked by the launcher)
    const long long v = __double2ll_rn((double)pz * kBilatFixed);
    // a wave's 64 consecutive pixels fall into a handful of cells (sigma_s pixels wide, one or two depth slices on a wall): one pair of
    // atomics per cell and wave, carrying the cell's sum and count, instead of one per pixel (integer sums: the same totals)
    const int lane = (int)threadIdx.x & 63;
    unsigned long long todo = __ballot(on);
    while (todo) {                                                  // uniform
        const int lead = __builtin_ctzll(todo);
        const int cl = __builtin_amdgcn_readlane(c, lead);
        const bool mine = on && c == cl;
        const unsigned long long same = __ballot(mine);
        const long long tot = wave_sum_ll(mine ? v : 0ll);
        if (lane == lead) {
            atomicAdd(&sum[cl], (unsigned long long)tot);
            atomicAdd(&cnt[cl], (int)__popcll(same));
        }
        todo &= ~same;
    }
}
// fixed point -> {sum z, count} floats in `a`; `b` (the other ping-pong array) starts at zero
__global__ void k_bilat_init(const unsigned long long* __restrict__ sum, const int* __restrict__ cnt, size_t cells, float2* __restrict__ a,
                             float2* __restrict__ b) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cells) return;
    a[c] = make_float2((float)((double)(long long)sum[c] / kBilatFixed), (float)cnt[c]);
    b[c] = make_float2(0.f, 0.f);
}
// one [1 2 1] / 4 pass along one axis over the interior cells: data <- buffer (border cells of `data` keep what they held)
__global__ void k_bilat_blur(const float2* __restrict__ buffer, float2* __restrict__ data, BilatGrid g, int off) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t cells = (size_t)g.nx * g.ny * g.nz;
    if (c >= cells) return;
    const int z = (int)(c % g.nz), y = (int)((c / g.nz) % g.ny), x = (int)(c / ((size_t)g.nz * g.ny));
    if (x < 1 || x >= g.nx - 1 || y < 1 || y >= g.ny - 1 || z < 1 || z >= g.nz - 1) return;
    const float2 m = buffer[c - off], p = buffer[c + off], q = buffer[c];
    data[c] = make_float2((m.x + p.x + 2.f * q.x) / 4.f, (m.y + p.y + 2.f * q.y) / 4.f);
}
__global__ void k_bilat_interp(float* __restrict__ xyz, int rows, int cols, BilatGrid g, const float2* __restrict__ data) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int y = i / cols, x = i - y * cols;
    float pz = xyz[3 * (size_t)i + 2];
    if (!isfinite(pz)) pz = g.base_max;
    const float fx = (float)x / g.sigma_s + (float)kBilatPadXY, fy = (float)y / g.sigma_s + (float)kBilatPadXY;
    const float fz = (pz - g.base_min) / g.sigma_r + (float)kBilatPadZ;
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    const int x0 = clampi((int)fx, g.nx - 1), x1 = clampi(x0 + 1, g.nx - 1);
    const int y0 = clampi((int)fy, g.ny - 1), y1 = clampi(y0 + 1, g.ny - 1);
    const int z0 = clampi((int)fz, g.nz - 1), z1 = clampi(z0 + 1, g.nz - 1);
    const float xa = fx - (float)x0, ya = fy - (float)y0, za = fz - (float)z0;
    const float2 c000 = data[bilat_idx(g, x0, y0, z0)], c100 = data[bilat_idx(g, x1, y0, z0)], c010 = data[bilat_idx(g, x0, y1, z0)],
                 c110 = data[bilat_idx(g, x1, y1, z0)], c001 = data[bilat_idx(g, x0, y0, z1)], c101 = data[bilat_idx(g, x1, y0, z1)],
                 c011 = data[bilat_idx(g, x0, y1, z1)], c111 = data[bilat_idx(g, x1, y1, z1)];
    const float w000 = (1.f - xa) * (1.f - ya) * (1.f - za), w100 = xa * (1.f - ya) * (1.f - za), w010 = (1.f - xa) * ya * (1.f - za),
                w110 = xa * ya * (1.f - za), w001 = (1.f - xa) * (1.f - ya) * za, w101 = xa * (1.f - ya) * za, w011 = (1.f - xa) * ya * za,
                w111 = xa * ya * za;
    const float d0 = w000 * c000.x + w100 * c100.x + w010 * c010.x + w110 * c110.x + w001 * c001.x + w101 * c101.x + w011 * c011.x + w111 * c111.x;
    const float d1 = w000 * c000.y + w100 * c100.y + w010 * c010.y + w110 * c110.y + w001 * c001.y + w101 * c101.y + w011 * c011.y + w111 * c111.y;
    xyz[3 * (size_t)i + 2] = d0 / d1;
}

// ---- one sensor's organised cloud: CloudRGBD::getPointCloud + DownsampleRGBD::downsamplePointCloud ----------------------------
// (OpenNI2_Grabber/FrameRGBD/CloudRGBD.h:107-166, DownsampleRGBD.h:209-300; restated in oracle/frame360_ref.cpp, whose float
// operations this kernel repeats: the cloud is bit-identical).  One thread per output point: the step x step block of depths, the
// pinhole back-projection of its valid pixels, per coordinate the element n/2 of the sorted values.
struct SensorCloudArgs {
    int rows, cols, step;
    float inv_fx, inv_fy, ox, oy, min_depth, max_depth;
    int depth_f32;      // 0: uint16 millimetres (the sensor's image); 1: float32 metres (the image Frame360::undistort left: CloudRGBD_Ext.h:116-118)
};
__device__ __forceinline__ float sorted_pick(float* v, int n, int k) {      // k-th smallest of v[0..n), n <= 16 (insertion sort)
    for (int i = 1; i < n; ++i) {
        const float x = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > x) {
            v[j + 1] = v[j];
            --j;
        }
        v[j + 1] = x;
    }
    return v[k];
}
__global__ void k_sensor_cloud(const uint8_t* __restrict__ depth, size_t depth_step, SensorCloudArgs a, float* __restrict__ out) {
    const int orows = a.rows / a.step, ocols = a.cols / a.step;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= orows * ocols) return;
    const int r = i / ocols, c = i - r * ocols;
    float xs[16], ys[16], zs[16];
    int n = 0;
    for (int r2 = r * a.step; r2 < (r + 1) * a.step; ++r2)
        for (int c2 = c * a.step; c2 < (c + 1) * a.step; ++c2) {
            float z;
            bool valid;
            if (a.depth_f32) {                                // getPointCloudUndist: z > 0 && z >= minDepth && z <= maxDepth
                z = *reinterpret_cast<const float*>(depth + (size_t)r2 * depth_step + 4 * (size_t)c2);
                valid = z > 0.f && z >= a.min_depth && z <= a.max_depth;
            } else {
                const unsigned short d = *reinterpret_cast<const unsigned short*>(depth + (size_t)r2 * depth_step + 2 * (size_t)c2);
                z = (float)(0.001 * (double)d);               // double product rounded to float, CloudRGBD.h:147
                valid = d > 0 && a.min_depth < z && z < a.max_depth;
            }
            if (valid) {
                xs[n] = (c2 - a.ox) * z * a.inv_fx;
                ys[n] = (r2 - a.oy) * z * a.inv_fy;
                zs[n] = z;
                ++n;
            }
        }
    float* o = out + 3 * (size_t)i;
    if (n == 0) {
        o[0] = o[1] = o[2] = __builtin_nanf("");
        return;
    }
    o[0] = sorted_pick(xs, n, n / 2);
    o[1] = sorted_pick(ys, n, n / 2);
    o[2] = sorted_pick(zs, n, n / 2);
}

}  // namespace f360
