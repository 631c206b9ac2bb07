// Dev probe: can a second wave on the same SIMD issue VALU / VMEM / LDS work while the first
// one streams fp32 MFMAs?  Block = 512 threads (8 waves, 2 per SIMD): waves 0-3 run MFMAs
// (or idle), waves 4-7 run a fixed batch of `kind` instructions and report their duration.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void probe(int mfma_iters, int kind, int work_iters, int prio,
                                             const float *gsrc, unsigned long long *out,
                                             float *sink) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = (float)i;
    __syncthreads();
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        float a = threadIdx.x * 0.001f, b = 1.0f;
        if (prio == 2) __builtin_amdgcn_s_setprio(3);
        unsigned long long t0 = clock64();
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        unsigned long long t1 = clock64();
        float s = 0.f;
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) s += acc[i][e];
        if (s == 123.456f) sink[0] = s;
        if (threadIdx.x == 0) out[2 * blockIdx.x] = t1 - t0;
    } else {
        // let the MFMA waves get going
        __builtin_amdgcn_s_sleep(20);
        if (prio == 1) __builtin_amdgcn_s_setprio(3);
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
        unsigned long long t0 = clock64();
        if (kind == 0) {            // independent VALU
            for (int it = 0; it < work_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    x0 = x0 * 1.0001f + 0.5f; x1 = x1 * 1.0001f + 0.5f;
                    x2 = x2 * 1.0001f + 0.5f; x3 = x3 * 1.0001f + 0.5f;
                }
            }
        } else if (kind == 1) {     // global loads (L2 hits), 4 in flight
            const float4 *p = reinterpret_cast<const float4 *>(gsrc) + (threadIdx.x & 63);
            for (int it = 0; it < work_iters; ++it) {
                float4 v0 = p[(it * 4 + 0) * 64 & 0xffff], v1 = p[(it * 4 + 1) * 64 & 0xffff];
                float4 v2 = p[(it * 4 + 2) * 64 & 0xffff], v3 = p[(it * 4 + 3) * 64 & 0xffff];
                x0 += v0.x + v1.y; x1 += v2.z + v3.w;
            }
        } else if (kind == 3) {     // VALU-free: 4 global loads -> 4 ds_write_b128, SALU loop control
            const unsigned voff = (threadIdx.x & 63) * 16;
            const unsigned laddr = (threadIdx.x & 63) * 16;
            asm volatile(
                "s_mov_b32 s20, %[iters]\n"
                "1:\n"
                "global_load_dwordx4 v[20:23], %[voff], %[sbase]\n"
                "global_load_dwordx4 v[24:27], %[voff], %[sbase] offset:1024\n"
                "global_load_dwordx4 v[28:31], %[voff], %[sbase] offset:2048\n"
                "global_load_dwordx4 v[32:35], %[voff], %[sbase] offset:3072\n"
                "s_waitcnt vmcnt(0)\n"
                "ds_write_b128 %[laddr], v[20:23]\n"
                "ds_write_b128 %[laddr], v[24:27] offset:1024\n"
                "ds_write_b128 %[laddr], v[28:31] offset:2048\n"
                "ds_write_b128 %[laddr], v[32:35] offset:3072\n"
                "s_waitcnt lgkmcnt(0)\n"
                "s_sub_u32 s20, s20, 1\n"
                "s_cmp_lg_u32 s20, 0\n"
                "s_cbranch_scc1 1b\n"
                :
                : [iters] "s"(work_iters), [voff] "v"(voff), [sbase] "s"(gsrc), [laddr] "v"(laddr)
                : "s20", "scc", "memory", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27",
                  "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35");
        } else if (kind == 4) {     // SALU only
            asm volatile(
                "s_mov_b32 s20, %[iters]\n"
                "s_mov_b32 s21, 0\n"
                "1:\n"
                "s_add_u32 s21, s21, 3\n s_add_u32 s21, s21, 3\n s_add_u32 s21, s21, 3\n s_add_u32 s21, s21, 3\n"
                "s_add_u32 s21, s21, 3\n s_add_u32 s21, s21, 3\n s_add_u32 s21, s21, 3\n s_add_u32 s21, s21, 3\n"
                "s_sub_u32 s20, s20, 1\n"
                "s_cmp_lg_u32 s20, 0\n"
                "s_cbranch_scc1 1b\n"
                :
                : [iters] "s"(work_iters)
                : "s20", "s21", "scc");
        } else {                    // LDS b128 reads
            const float4 *p = reinterpret_cast<const float4 *>(lds) + (threadIdx.x & 63);
            for (int it = 0; it < work_iters; ++it) {
                float4 v0 = p[(it * 4 + 0) * 64 & 2047], v1 = p[(it * 4 + 1) * 64 & 2047];
                float4 v2 = p[(it * 4 + 2) * 64 & 2047], v3 = p[(it * 4 + 3) * 64 & 2047];
                x0 += v0.x + v1.y; x1 += v2.z + v3.w;
            }
        }
        unsigned long long t1 = clock64();
        if (x0 + x1 + x2 + x3 == 123.456f) sink[1] = x0;
        if ((threadIdx.x & 63) == 0 && wave == 4) out[2 * blockIdx.x + 1] = t1 - t0;
    }
}

int main() {
    unsigned long long *out; float *sink, *gsrc;
    const int nb = 256;
    hipMalloc(&out, nb * 2 * 8); hipMalloc(&sink, 64); hipMalloc(&gsrc, 1 << 22);
    hipMemset(gsrc, 0, 1 << 22);
    unsigned long long h[2 * nb];
    const char *names[5] = {"VALU fma x32/iter", "global_load_dwordx4 x4/iter", "ds_read_b128 x4/iter",
                            "VALU-free 4 loads->4 ds_write", "SALU x8/iter"};
    for (int prio = 0; prio < 2; ++prio)
    for (int kind = 0; kind < 5; ++kind)
        for (int mf = (prio ? 1 : 0); mf < 2; ++mf) {
            const int work = kind == 0 ? 400 : 200;
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 0, 0, mf ? 4000 : 0, kind, work, prio, gsrc, out, sink);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0, w = 0;
            for (int i = 0; i < nb; ++i) { m += h[2 * i]; w += h[2 * i + 1]; }
            printf("prio[%s] %-30s other wave %-12s: worker %8.0f ticks (%.1f per iter), mfma wave %9.0f ticks (%.1f per MFMA)\n",
                   prio == 0 ? "equal" : (prio == 1 ? "worker hi" : "mfma hi"), names[kind], mf ? "MFMA stream" : "idle", w / nb, w / nb / work, m / nb,
                   mf ? m / nb / (4000.0 * 16) : 0.0);
        }
    return 0;
}
