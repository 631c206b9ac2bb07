// Dev tool: sustained fp32-MFMA rate and the shader clock actually held under that load.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_clock_probe.hip -o /tmp/probe && /tmp/probe
// Each wave runs a pure v_mfma_f32_32x32x2_f32 stream (4 independent accumulators, operands
// in registers); s_memtime counts shader cycles, s_memrealtime a 100 MHz reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void probe(float *out, unsigned long long *stamps, int iters,
                                                float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = seed + threadIdx.x * 0.001f, b = 1.0f - threadIdx.x * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-7f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

int main(int argc, char **argv) {
    const int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 2;
    const int blocks = 256 * blocks_per_cu, iters = 2000, reps = 300;
    float *out; unsigned long long *st;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&st, blocks * 4 * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, st, iters, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, st, iters, 0.5f + r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * 4 /*waves*/ * iters * 64 /*mfma*/ * (2.0 * 32 * 32 * 2);
    unsigned long long *h = (unsigned long long *)malloc(blocks * 4 * 2 * sizeof(unsigned long long));
    hipMemcpy(h, st, blocks * 4 * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double ratio = 0; double cyc = 0;
    for (int w = 0; w < blocks * 4; ++w) { ratio += (double)h[2 * w] / (double)h[2 * w + 1]; cyc += (double)h[2 * w]; }
    ratio /= blocks * 4; cyc /= blocks * 4;
    printf("blocks/CU %d: %.1f TFLOP/s over %.1f ms; shader clock %.3f GHz; cycles per MFMA per wave %.1f\n",
           blocks_per_cu, flops / (ms * 1e-3) / 1e12, ms, ratio * 0.1, cyc / ((double)iters * 64));
    return 0;
}
