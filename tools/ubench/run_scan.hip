// Checks the run detection + segmented max-scan of k_occ_build (csrc/occlusion_kernels.h) against a scalar loop.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/run_scan tools/ubench/run_scan.hip && /tmp/run_scan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k(const unsigned* ti, const float* dinv, const int* cnd, int n, unsigned char* runinfo, int4* nodes, int* istail) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in = i < n;
    const int ic = in ? i : n - 1;
    const bool cand = in && cnd[ic] != 0;
    const float di = dinv[ic];
    const unsigned tkey = cand ? ti[ic] : (0xFF000000u | (unsigned)lane);
    const unsigned t_before = __shfl_up(tkey, 1), t_after = __shfl_down(tkey, 1);      // (outside the ||: a shuffle reads active lanes only)
    const bool run_head = lane == 0 || t_before != tkey;
    const bool run_tail = lane == 63 || t_after != tkey;
    int lead = run_head ? lane : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(lead, off);
        if (lane >= off) lead = max(lead, o);
    }
    unsigned long long key = ((unsigned long long)__float_as_uint(di) << 32) | (unsigned)ic;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long ok = __shfl_up(key, off);
        const int ol = __shfl_up(lead, off);
        if (lane >= off && ol == lead && ok > key) key = ok;
    }
    const unsigned long long kprev = __shfl_up(key, 1);
    const bool has_prev = lane > lead;
    const bool pm_run = !(has_prev && __uint_as_float((unsigned)(kprev >> 32)) > di);
    if (in) {
        runinfo[i] = cand ? (unsigned char)(0x40u | (pm_run ? 0x80u : 0u) | (unsigned)(lane - lead)) : (unsigned char)0;
        istail[i] = cand && run_tail;
        nodes[i] = make_int4((int)(unsigned)key, (int)(unsigned)(key >> 32), i - (lane - lead), 0);
    }
}
int main() {
    const int n = 64 * 1000 + 17;
    std::vector<unsigned> ti(n); std::vector<float> di(n); std::vector<int> c(n);
    srand(3);
    unsigned t = 0;
    for (int i = 0; i < n; ++i) { if (rand() % 3 == 0) t = rand() % 5000; ti[i] = t; di[i] = 0.25f * (1 + rand() % 7); c[i] = rand() % 10 != 0; }
    unsigned* dt; float* dd; int* dc; unsigned char* dr; int4* dn; int* dtl;
    hipMalloc(&dt, n * 4); hipMalloc(&dd, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dr, n); hipMalloc(&dn, n * 16); hipMalloc(&dtl, n * 4);
    hipMemcpy(dt, ti.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dd, di.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dt, dd, dc, n, dr, dn, dtl);
    std::vector<unsigned char> r(n); std::vector<int4> nd(n); std::vector<int> tl(n);
    hipMemcpy(r.data(), dr, n, hipMemcpyDeviceToHost); hipMemcpy(nd.data(), dn, n * 16, hipMemcpyDeviceToHost); hipMemcpy(tl.data(), dtl, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w * 64 < n; ++w) {
        int lead = 0; float mx = 0; int amx = 0;
        for (int l = 0; l < 64 && w * 64 + l < n; ++l) {
            const int i = w * 64 + l;
            const bool head = l == 0 || !c[i] || !c[i - 1] || ti[i - 1] != ti[i];
            float before = 0; bool hasprev = false;
            if (head) { lead = l; mx = di[i]; amx = i; } else { before = mx; hasprev = true; if (di[i] >= mx) { mx = di[i]; amx = i; } }
            const bool tail = l == 63 || i == n - 1 || !c[i] || !c[i + 1] || ti[i + 1] != ti[i];
            unsigned char e = c[i] ? (0x40 | ((hasprev && before > di[i]) ? 0 : 0x80) | (l - lead)) : 0;
            if (e != r[i]) { if (bad < 10) printf("runinfo %d: gpu %02x cpu %02x\n", i, r[i], e); ++bad; }
            if (c[i] && tail != (tl[i] != 0)) { if (bad < 10) printf("tail %d: gpu %d cpu %d\n", i, tl[i], (int)tail); ++bad; }
            if (c[i] && tail) {
                float g; memcpy(&g, &nd[i].y, 4);
                if (nd[i].x != amx || g != mx || nd[i].z != w * 64 + lead) { if (bad < 10) printf("node %d: gpu (%d %g %d) cpu (%d %g %d)\n", i, nd[i].x, g, nd[i].z, amx, mx, w * 64 + lead); ++bad; }
            }
        }
    }
    printf("%s: %d mismatches of %d\n", bad ? "FAIL" : "OK", bad, n);
    return bad != 0;
}
