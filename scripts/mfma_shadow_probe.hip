// Dev probe: marginal cost of one extra instruction of each class issued by the SAME wave
// between two of its fp32 MFMAs (one wave per SIMD, one block per CU).  Instruction streams
// are inline asm so nothing is reordered or consumed:
//   16 x { v_mfma_f32_32x32x2_f32 ; K x <instr> }  per iteration
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shadow_probe mfma_shadow_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP1(x) x
#define REP2(x) x x
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

template <int KIND, int K>
__global__ __launch_bounds__(256) void probe(int iters, const float *gsrc, unsigned long long *out,
                                             float *sink) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)i;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f;
    const unsigned laddr = (threadIdx.x & 63) * 16;
    const unsigned voff = threadIdx.x * 16;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (KIND == 0)
                    asm volatile("v_fma_f32 v200, v200, v200, v200" ::: "v200");
                else if constexpr (KIND == 1)
                    asm volatile("ds_read_b128 v[200:203], %0" ::"v"(laddr) : "v200", "v201", "v202", "v203");
                else if constexpr (KIND == 2)
                    asm volatile("global_load_dwordx4 v[200:203], %0, %1" ::"v"(voff), "s"(gsrc)
                                 : "v200", "v201", "v202", "v203");
                else if constexpr (KIND == 3)
                    asm volatile("ds_write_b128 %0, v[204:207]" ::"v"(laddr) : "memory");
                else if constexpr (KIND == 4)
                    asm volatile("s_add_u32 s40, s40, 1" ::: "s40", "scc");
                else if constexpr (KIND == 5)
                    asm volatile("v_cndmask_b32 v200, v200, v201, vcc" ::: "v200");
                else if constexpr (KIND == 6)
                    asm volatile("v_mov_b32 v200, v201" ::: "v200");
                else if constexpr (KIND == 7)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(gsrc + threadIdx.x * 4),
                        (__attribute__((address_space(3))) void *)(lds + 1024 * (threadIdx.x >> 6)), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int KIND, int K>
static double run(const float *gsrc, unsigned long long *out, float *sink) {
    const int nb = 256, iters = 2000;
    unsigned long long h[nb];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<KIND, K>), dim3(nb), dim3(256), 0, 0, iters, gsrc, out, sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < nb; ++i) m += h[i];
    return m / nb / (iters * 16.0);
}

template <int KIND>
static void sweep(const char *name, const float *gsrc, unsigned long long *out, float *sink) {
    const double base = run<KIND, 0>(gsrc, out, sink);
    const double k1 = run<KIND, 1>(gsrc, out, sink), k2 = run<KIND, 2>(gsrc, out, sink);
    const double k4 = run<KIND, 4>(gsrc, out, sink), k8 = run<KIND, 8>(gsrc, out, sink);
    printf("%-22s ticks per MFMA: K=0 %.1f | K=1 %.1f | K=2 %.1f | K=4 %.1f | K=8 %.1f  (marginal %.1f per instr at K=8)\n",
           name, base, k1, k2, k4, k8, (k8 - base) / 8.0);
}

int main() {
    unsigned long long *out; float *sink, *gsrc;
    hipMalloc(&out, 256 * 8); hipMalloc(&sink, 64); hipMalloc(&gsrc, 1 << 22);
    hipMemset(gsrc, 0, 1 << 22);
    sweep<0>("v_fma_f32", gsrc, out, sink);
    sweep<5>("v_cndmask_b32", gsrc, out, sink);
    sweep<6>("v_mov_b32", gsrc, out, sink);
    sweep<4>("s_add_u32", gsrc, out, sink);
    sweep<1>("ds_read_b128", gsrc, out, sink);
    sweep<3>("ds_write_b128", gsrc, out, sink);
    sweep<2>("global_load_dwordx4", gsrc, out, sink);
    sweep<7>("global_load_lds_dwordx4", gsrc, out, sink);
    return 0;
}
