// Dev probe: what bounds a "gather rows from an LDS tile and sum them" loop on gfx950?
// One workgroup per CU (WAVES waves), each wave sums N_NB pseudo-random 1-KiB rows of a 128-KiB
// LDS tile; variants switch parts of the loop off.  Prints cycles per row-read per CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_gather_probe.hip -o scripts/_build/lds_gather_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(1024) void probe(const int *idx, float *out, long long *cyc, int n_nb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 128 * 256; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = (float)(i & 7);
    __syncthreads();
    const unsigned char *tb = smem + lane * 16;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float acc2[4] = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int base = 0; base < n_nb; base += 64) {
        const int off = (idx[(wave * 64 + base + lane) & 4095] & 127) << 10;   // like one coalesced col read
        float4 v[INFLIGHT];
#pragma unroll
        for (int g = 0; g < 64; g += INFLIGHT) {
#pragma unroll
            for (int t = 0; t < INFLIGHT; ++t) {
                int o;
                if (MODE == 3) o = ((g + t) * 37 & 127) << 10;                  // constant addresses: no readlane
                else o = __builtin_amdgcn_readlane(off, g + t);
                if (MODE == 2) v[t] = make_float4(__builtin_bit_cast(float, o), 0.f, 0.f, 0.f);   // no LDS
                else v[t] = *reinterpret_cast<const float4 *>(tb + o);
            }
#pragma unroll
            for (int t = 0; t < INFLIGHT; ++t) {
                if (MODE == 1) { acc[0] += v[t].x; }                              // 1 add instead of 4
                else if (MODE == 4 && (t & 1)) { acc2[0] += v[t].x; acc2[1] += v[t].y; acc2[2] += v[t].z; acc2[3] += v[t].w; }
                else { acc[0] += v[t].x; acc[1] += v[t].y; acc[2] += v[t].z; acc[3] += v[t].w; }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + acc2[0] + acc2[1] + acc2[2] + acc2[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int INFLIGHT>
static void run(const char *name, int waves, int n_nb, const int *d_idx, float *d_out, long long *d_cyc) {
    const int blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<MODE, INFLIGHT>), hipFuncAttributeMaxDynamicSharedMemorySize, 129 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<MODE, INFLIGHT>), dim3(blocks), dim3(waves * 64), 129 * 1024, 0, d_idx, d_out, d_cyc, n_nb);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    long long cyc[256];
    hipMemcpy(cyc, d_cyc, sizeof(cyc), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks; ++i) mean += (double)cyc[i];
    mean /= blocks;
    const double reads_per_cu = (double)waves * n_nb;
    printf("%-44s waves=%2d inflight=%2d: %8.1f us, %7.0f cycles in-kernel, %5.2f cycles per 1-KiB row read per CU (LDS peak: 4)\n",
           name, waves, INFLIGHT, ms * 1e3, mean, mean / reads_per_cu);
}

int main() {
    int *d_idx; float *d_out; long long *d_cyc;
    int h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = rand();
    hipMalloc(&d_idx, sizeof(h)); hipMemcpy(d_idx, h, sizeof(h), hipMemcpyHostToDevice);
    hipMalloc(&d_out, 256 * 1024 * 4); hipMalloc(&d_cyc, 256 * 8);
    const int n = 64 * 40;
    for (int waves : {4, 8, 16}) {
        run<0, 8>("full loop (readlane + add + ds_read + 4 adds)", waves, n, d_idx, d_out, d_cyc);
        run<0, 16>("full loop", waves, n, d_idx, d_out, d_cyc);
        run<1, 8>("1 add per row instead of 4", waves, n, d_idx, d_out, d_cyc);
        run<2, 8>("no LDS reads", waves, n, d_idx, d_out, d_cyc);
        run<3, 8>("constant addresses (no readlane/add)", waves, n, d_idx, d_out, d_cyc);
        run<4, 8>("two accumulator sets", waves, n, d_idx, d_out, d_cyc);
    }
    return 0;
}
