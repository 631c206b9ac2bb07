// Dev probe: semantics of __builtin_amdgcn_global_load_lds (global_load_lds_dwordx4) on gfx950:
// per-lane global address, LDS destination = uniform base + lane * 16.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float *src, float *out, const int *perm) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
    const int lane = threadIdx.x;
    const float *p = src + perm[lane] * 4;                 // lane reads chunk perm[lane]
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                     (__attribute__((address_space(3))) void *)(lds + 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    float h[256]; int perm[64]; float o[512];
    for (int i = 0; i < 256; ++i) h[i] = i;
    for (int i = 0; i < 64; ++i) perm[i] = (i * 7 + 3) % 64;
    float *d, *dout; int *dp;
    hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o)); hipMalloc(&dp, sizeof(perm));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(dp, perm, sizeof(perm), hipMemcpyHostToDevice);
    hipMemset(dout, 0, sizeof(o));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout, dp);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e)
            if (o[256 + l * 4 + e] != (float)(perm[l] * 4 + e)) ++bad;
    printf("lds_dma: %s (bad %d); lane 1 got %.0f %.0f %.0f %.0f, expected chunk %d\n", bad ? "MISMATCH" : "OK: dst = base + lane*16, per-lane source",
           bad, o[256 + 4], o[256 + 5], o[256 + 6], o[256 + 7], perm[1]);
    return bad != 0;
}
