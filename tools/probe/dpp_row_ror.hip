// Probe: which lane does row_ror:n read from? (MI355X; prints the source lane for lanes 0..15)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ int dpp(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
__global__ void k(int *out) {
    const int l = threadIdx.x;
    out[l] = dpp<0x124>(l);          // row_ror:4
    out[64 + l] = dpp<0x128>(l);     // row_ror:8
    out[128 + l] = dpp<0x12c>(l);    // row_ror:12
    out[192 + l] = __builtin_amdgcn_ds_bpermute(((l & ~15) + 8) << 2, l);
}
int main() {
    int *d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; ++r) { printf("%s:", r == 0 ? "ror4" : r == 1 ? "ror8" : r == 2 ? "ror12" : "bperm"); for (int l = 0; l < 20; ++l) printf(" %d", h[64 * r + l]); printf("\n"); }
    return 0;
}
