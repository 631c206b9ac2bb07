// Dev tool: which ingredient of the GEMM main loop costs MFMA issue slots?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256, 2) void probe(float *out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 36];
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    for (int i = threadIdx.x; i < 2 * 128 * 36; i += 256) lds[i] = seed + i * 1e-4f;
    __syncthreads();
    float4 af[2], bf[2];
    af[0] = make_float4(seed, seed + 1, seed + 2, seed + 3); af[1] = af[0]; bf[0] = af[0]; bf[1] = af[0];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (V >= 2) {   // LDS fragment reads like the GEMM (k-contiguous images)
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4 *>(lds + (wm * 64 + i * 32 + r) * 36 + 8 * q + 4 * hh);
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const float4 *>(lds + 128 * 36 + (wn * 64 + j * 32 + r) * 36 + 8 * q + 4 * hh);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float a = V == 0 ? af[0].x : (s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w);
                        float b = V == 0 ? bf[0].x : (s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                    }
        }
        if (V >= 3) __syncthreads();
        if (V >= 4) {   // LDS stores of the next tile (8 x b128 per thread) + second half barrier-free
            float4 v = make_float4(seed + it, 1.f, 2.f, 3.f);
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<float4 *>(lds + ((threadIdx.x >> 3) + 32 * (i & 3)) * 36 + (threadIdx.x & 7) * 4 + (i >> 2) * 128 * 36) = v;
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][9];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// V5/V6: the real data path -- global loads of the next A/B k tile into registers at the top
// of the iteration (row-clamped addressing like gemm.hip), MFMAs from the current LDS buffer,
// registers -> other LDS buffer, one barrier.  V6 = same with loads issued 2 tiles ahead.
template <int V>
__global__ __launch_bounds__(256, 2) void probe_g(const float *__restrict__ A, const float *__restrict__ B,
                                                  int lda, int rows, int ktiles, float *out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int T = 128 * 36;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, hh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int bm = blockIdx.x % 16, bn = blockIdx.x / 16;
    float4 stA[4], stB[4];
    auto gload = [&](int kt) {
        const int kk = kt * 32 + (t & 7) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ra = min(bm * 128 + (t >> 3) + 32 * i, rows - 1);
            const int rb = min(bn * 128 + (t >> 3) + 32 * i, rows - 1);
            stA[i] = *reinterpret_cast<const float4 *>(A + (long)ra * lda + kk);
            stB[i] = *reinterpret_cast<const float4 *>(B + (long)rb * lda + kk);
        }
    };
    auto sstore = [&](int buf) {
        float *sa = sm + buf * 2 * T, *sb = sa + T;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4 *>(sa + ((t >> 3) + 32 * i) * 36 + (t & 7) * 4) = stA[i];
            *reinterpret_cast<float4 *>(sb + ((t >> 3) + 32 * i) * 36 + (t & 7) * 4) = stB[i];
        }
    };
    gload(0); sstore(0); __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) gload(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        const float *a_s = sm + cur * 2 * T, *b_s = a_s + T;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4 *>(a_s + (wm * 64 + i * 32 + r) * 36 + 8 * q + 4 * hh);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const float4 *>(b_s + (wn * 64 + j * 32 + r) * 36 + 8 * q + 4 * hh);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float a = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
                        float b = s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < ktiles) sstore(cur ^ 1);
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][9];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

void run_g(const char *name, int K) {
    const int rows = 4096, blocks = 16 * 32, reps = 10;   // C = 2048 x 4096 tiles over A[2048.., K], B[4096, K]
    float *A, *B, *out;
    hipMalloc(&A, (size_t)rows * K * 4); hipMalloc(&B, (size_t)rows * K * 4); hipMalloc(&out, blocks * 256 * 4);
    hipMemset(A, 0, (size_t)rows * K * 4); hipMemset(B, 0, (size_t)rows * K * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_g<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 36 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe_g<5>, dim3(blocks), dim3(256), 4 * 128 * 36 * 4, 0, A, B, K, rows, K / 32, out);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe_g<5>, dim3(blocks), dim3(256), 4 * 128 * 36 * 4, 0, A, B, K, rows, K / 32, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %.1f TFLOP/s (K=%d, %.3f ms/launch)\n", name, (double)reps * 2.0 * 2048 * 4096 * K / (ms * 1e-3) / 1e12, K, ms / reps);
    hipFree(A); hipFree(B); hipFree(out);
}

template <int V> void run(const char *name) {
    const int blocks = 512, iters = 4000, reps = 40;
    float *out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f + r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * 4 * iters * 64 * (2.0 * 32 * 32 * 2);
    printf("%-58s %.1f TFLOP/s\n", name, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<0>("V0 same operand registers");
    run<1>("V1 distinct operand registers (no LDS)");
    run<2>("V2 + ds_read_b128 fragments each k block");
    run<3>("V3 + one barrier per 64 MFMAs");
    run<4>("V4 + 8 ds_write_b128 per thread + second barrier");
    run_g("V5 real data path: global->reg->LDS, 1 barrier", 8192);
    run_g("V5 real data path: global->reg->LDS, 1 barrier", 32768);
    return 0;
}
