"""Host->device copy rates on this box for one 2048x1024 frame (rgb 6 MiB + depth 8 MiB / 4 MiB): 1-D vs 2-D copies, pageable vs
pinned.  Pure HIP runtime through ctypes; decides how rgbd360_set_* should upload.  python tools/h2d_bw.py"""
import ctypes as C, time
import numpy as np
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
s = C.c_void_p(); assert hip.hipStreamCreate(C.byref(s)) == 0
W, H = 2048, 1024
for name, bpp in (("rgb u8x3", 3), ("depth f32", 4), ("depth u16", 2)):
    nbytes = W * H * bpp
    d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), nbytes) == 0
    pageable = np.random.randint(0, 255, nbytes, dtype=np.uint8)
    p = C.c_void_p(); assert hip.hipHostMalloc(C.byref(p), nbytes, 0) == 0
    C.memmove(p, pageable.ctypes.data, nbytes)
    for hname, hp in (("pageable", pageable.ctypes.data), ("pinned", p.value)):
        for kind in ("1d", "2d"):
            def go():
                if kind == "1d":
                    return hip.hipMemcpyAsync(d, hp, nbytes, 1, s)
                return hip.hipMemcpy2DAsync(d, W * bpp, hp, W * bpp, W * bpp, H, 1, s)
            for _ in range(3):
                assert go() == 0
            hip.hipStreamSynchronize(s)
            t0 = time.perf_counter()
            for _ in range(20):
                assert go() == 0
            t_issue = time.perf_counter() - t0
            hip.hipStreamSynchronize(s)
            dt = (time.perf_counter() - t0) / 20
            print("%-10s %-8s %s: %.3f ms/copy  %.1f GB/s  (host issue time %.3f ms/copy)" % (name, hname, kind, dt * 1e3, nbytes / dt / 1e9, t_issue / 20 * 1e3))
