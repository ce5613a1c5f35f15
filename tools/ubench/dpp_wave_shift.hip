#include <hip/hip_runtime.h>
__global__ void k(int* out) {
    int v = threadIdx.x;
    int a = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, false);   // wave_shl:1
    int b = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);   // wave_shr:1
    out[threadIdx.x] = a * 1000 + b;
}
int main() { int* d; hipMalloc(&d, 256); k<<<1,64>>>(d); int h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); for (int i = 0; i < 64; i += 9) printf("%d:%d ", i, h[i]); printf("\n%d %d\n", h[0], h[63]); return 0; }
