// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, one
// rounding per product -- the 1e-4 parity bar rules out bf16/fp8 inputs).
//
// Three operand layouts, all row-major with explicit leading dimensions:
//   NT  Y[m,n] = A[m,k] . W[n,k]^T + bias   forward of nn.Linear   (modules.py:233)
//   NN  Z[m,n] = G[m,k] . W[k,n]            dZ = dY . W
//   TN  D[m,n] = G[k,m]^T . A[k,n]          dW = dY^T . Z
//
// Block tile 128x128x32, 256 threads = 4 waves in 2x2, each wave 64x64 = 2x2 MFMA
// tiles of 32x32 (64 accumulator VGPRs).  Global -> registers -> LDS staging,
// double buffered (the next tile's loads are issued before the MFMA block and
// written to the other LDS buffer after it; one barrier per k tile).
//
// LDS images (chosen so that global loads stay 16-B coalesced and fragment reads
// are bank-conflict free, MI355X_MICROARCH.md section LDS):
//   k-contiguous operand  : [128 rows][32 k + 4 pad]; a lane reads ONE ds_read_b128
//                           = 4 consecutive k of its row.  Lanes 0-31 take k = 8q..8q+3,
//                           lanes 32-63 k = 8q+4..8q+7, so MFMA step j of block q
//                           multiplies k = 8q + 4*(lane>>5) + j -- a permutation of k
//                           applied to BOTH operands, which a dot product allows.
//   m/n-contiguous operand: [32 k][128]; a lane reads 4 ds_read_b32 at the same
//                           permuted k; 32 consecutive lanes hit 32 consecutive banks.
//
// Ragged m, n, k are zero-filled in the loader; the store is predicated.
// Shapes with few output tiles (n = 41 logits, dW of the last layer) are split along
// k across blockIdx.z into a workspace and reduced deterministically.
// Blocks are dealt to XCDs in 8x8 super-tiles so that the 64 blocks sharing an L2
// touch 8 A panels + 8 B panels.
#include "common.h"

namespace gist {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_KC = BK + 4;   // k-contiguous image: row stride (floats)
constexpr int LDS_MC = BM;       // m/n-contiguous image: k-row stride (floats)
constexpr int TILE_KC = BM * LDS_KC;
constexpr int TILE_MC = BK * LDS_MC;

struct GemmArgs {
    const float *a; int64_t lda;
    const float *b; int64_t ldb;
    const float *bias;
    float *c; int64_t ldc;
    int m, n, k;
    int tiles_m, tiles_n;
    int k_per_split;       // multiple of BK
    int64_t split_stride;  // elements between split slabs (0 = write C directly)
};

// ---- global -> register staging --------------------------------------------
// Two loaders per image.  The ALIGNED one (16-B aligned base, ld % 4 == 0) is branch
// free: hipcc turns a per-load `if (in range) load` into a branch plus an
// s_waitcnt per load, which serialises the eight loads of a k tile.  Instead
//   * the NON-reduction coordinate (row of a k-contiguous image, column of an
//     m/n-contiguous image) is CLAMPED to a readable address: whatever is loaded
//     there only ever reaches output rows/columns >= m/n, which are never stored;
//   * the REDUCTION coordinate k is masked with selects (0 beyond k_end), so nothing
//     out of range enters a dot product.
// A 16-B chunk that straddles the end of a row is still inside the row's pitch
// because ld >= round_up(extent, 4) whenever ld % 4 == 0.
// The generic loader (any alignment) keeps per-element guards.

// k-contiguous operand: element (r, kk) at p[r*ld + kk].  256 threads move
// 128 rows x 32 k = 1024 float4: thread t -> rows t/8 + 32 i, k chunk t%8.
__device__ __forceinline__ void load_kc_aligned(const float *__restrict__ p, int64_t ld, int rows,
                                                int kdim, int row0, int k0, float4 (&st)[4]) {
    const int t = threadIdx.x;
    const int kk = k0 + (t & 7) * 4;
    const int kc = min(kk, (int)ld - 4);           // stays inside the row pitch
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = min(row0 + (t >> 3) + 32 * i, rows - 1);
        st[i] = *reinterpret_cast<const float4 *>(p + (int64_t)r * ld + kc);
    }
}

// Reduction-dimension mask of the aligned loaders, applied when the registers are
// written to LDS -- i.e. AFTER the MFMA block -- so that the loads stay in flight
// under the MFMAs instead of being waited for right after issue.
__device__ __forceinline__ void mask_kc(float4 (&st)[4], int kdim, int k0) {
    const int kk = k0 + (threadIdx.x & 7) * 4;
    if (kk + 3 < kdim) return;
    const bool k0ok = kk + 0 < kdim, k1ok = kk + 1 < kdim, k2ok = kk + 2 < kdim;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        st[i].x = k0ok ? st[i].x : 0.f;
        st[i].y = k1ok ? st[i].y : 0.f;
        st[i].z = k2ok ? st[i].z : 0.f;
        st[i].w = 0.f;
    }
}

__device__ __forceinline__ void mask_mc(float4 (&st)[4], int kdim, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = k0 + (threadIdx.x >> 5) + 8 * i;
        if (kk >= kdim) st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__device__ __forceinline__ void load_kc_generic(const float *__restrict__ p, int64_t ld, int rows,
                                                int kdim, int row0, int k0, float4 (&st)[4]) {
    const int t = threadIdx.x;
    const int kk = k0 + (t & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + (t >> 3) + 32 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows) {
            const float *src = p + (int64_t)r * ld + kk;
            if (kk + 0 < kdim) v.x = src[0];
            if (kk + 1 < kdim) v.y = src[1];
            if (kk + 2 < kdim) v.z = src[2];
            if (kk + 3 < kdim) v.w = src[3];
        }
        st[i] = v;
    }
}

template <bool ALIGNED>
__device__ __forceinline__ void load_kc(const float *__restrict__ p, int64_t ld, int rows, int kdim,
                                        int row0, int k0, float4 (&st)[4]) {
    if constexpr (ALIGNED) load_kc_aligned(p, ld, rows, kdim, row0, k0, st);
    else load_kc_generic(p, ld, rows, kdim, row0, k0, st);
}

__device__ __forceinline__ void store_kc(float *__restrict__ s, const float4 (&st)[4]) {
    const int t = threadIdx.x;
    const int kq = (t & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (t >> 3) + 32 * i;
        *reinterpret_cast<float4 *>(s + r * LDS_KC + kq) = st[i];
    }
}

// m/n-contiguous operand: element (kk, c) at p[kk*ld + c].  32 k x 128 cols:
// thread t -> k rows t/32 + 8 i, column chunk t%32.
__device__ __forceinline__ void load_mc_aligned(const float *__restrict__ p, int64_t ld, int cols,
                                                int kdim, int col0, int k0, float4 (&st)[4]) {
    const int t = threadIdx.x;
    const int cq = min(col0 + (t & 31) * 4, (int)ld - 4);   // clamp inside the row pitch
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = k0 + (t >> 5) + 8 * i;
        st[i] = *reinterpret_cast<const float4 *>(p + (int64_t)min(kk, kdim - 1) * ld + cq);
    }
}

__device__ __forceinline__ void load_mc_generic(const float *__restrict__ p, int64_t ld, int cols,
                                                int kdim, int col0, int k0, float4 (&st)[4]) {
    const int t = threadIdx.x;
    const int cq = col0 + (t & 31) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = k0 + (t >> 5) + 8 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kk < kdim) {
            const float *src = p + (int64_t)kk * ld + cq;
            if (cq + 0 < cols) v.x = src[0];
            if (cq + 1 < cols) v.y = src[1];
            if (cq + 2 < cols) v.z = src[2];
            if (cq + 3 < cols) v.w = src[3];
        }
        st[i] = v;
    }
}

template <bool ALIGNED>
__device__ __forceinline__ void load_mc(const float *__restrict__ p, int64_t ld, int cols, int kdim,
                                        int col0, int k0, float4 (&st)[4]) {
    if constexpr (ALIGNED) load_mc_aligned(p, ld, cols, kdim, col0, k0, st);
    else load_mc_generic(p, ld, cols, kdim, col0, k0, st);
}

__device__ __forceinline__ void store_mc(float *__restrict__ s, const float4 (&st)[4]) {
    const int t = threadIdx.x;
    const int cq = (t & 31) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = (t >> 5) + 8 * i;
        *reinterpret_cast<float4 *>(s + kk * LDS_MC + cq) = st[i];
    }
}

// ---- LDS -> fragment ---------------------------------------------------------
// Returns the 4 values a lane feeds to MFMA steps j = 0..3 of k block q for the
// 32-row (or 32-column) slab starting at `base`.
template <bool KC>
__device__ __forceinline__ float4 read_frag(const float *__restrict__ s, int base, int q, int r,
                                            int hh) {
    if constexpr (KC) {
        return *reinterpret_cast<const float4 *>(s + (base + r) * LDS_KC + 8 * q + 4 * hh);
    } else {
        const float *p = s + (8 * q + 4 * hh) * LDS_MC + base + r;
        return make_float4(p[0], p[LDS_MC], p[2 * LDS_MC], p[3 * LDS_MC]);
    }
}

__device__ __forceinline__ float f4(const float4 &v, int j) {
    return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
}

// ALIGNED: both operands have 16-B aligned bases and leading dimensions % 4 == 0
// (every buffer the engine allocates); otherwise the generic guarded loader runs.
template <bool A_KC, bool B_KC, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TA = A_KC ? TILE_KC : TILE_MC;
    constexpr int TB = B_KC ? TILE_KC : TILE_MC;
    // buffer b: A image at smem + b*(TA+TB), B image right behind it

    // ---- block -> output tile, 8x8 super-tiles per XCD ------------------------
    const int nwg = g.tiles_m * g.tiles_n;
    const int orig = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = orig & 7;
    const int L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
    constexpr int GM = 8;
    const int width = GM * g.tiles_n;
    const int group = L / width;
    const int first_m = group * GM;
    const int gsz = min(g.tiles_m - first_m, GM);
    const int bm = first_m + (L % width) % gsz;
    const int bn = (L % width) / gsz;
    const int row0 = bm * BM, col0 = bn * BN;

    const int k_begin = blockIdx.z * g.k_per_split;
    const int k_end = min(g.k, k_begin + g.k_per_split);
    const int n_kt = (k_end - k_begin + BK - 1) / BK;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hh = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 stA[4], stB[4];
    auto gload = [&](int kt) {
        const int k0 = k_begin + kt * BK;
        if constexpr (A_KC) load_kc<ALIGNED>(g.a, g.lda, g.m, k_end, row0, k0, stA);
        else load_mc<ALIGNED>(g.a, g.lda, g.m, k_end, row0, k0, stA);
        if constexpr (B_KC) load_kc<ALIGNED>(g.b, g.ldb, g.n, k_end, col0, k0, stB);
        else load_mc<ALIGNED>(g.b, g.ldb, g.n, k_end, col0, k0, stB);
    };
    auto sstore = [&](int buf, int kt) {
        float *sa = smem + buf * (TA + TB);
        float *sb = sa + TA;
        if constexpr (ALIGNED) {
            const int k0 = k_begin + kt * BK;
            if (k0 + BK > k_end) {            // only the last k tile of a split can be ragged
                if constexpr (A_KC) mask_kc(stA, k_end, k0); else mask_mc(stA, k_end, k0);
                if constexpr (B_KC) mask_kc(stB, k_end, k0); else mask_mc(stB, k_end, k0);
            }
        }
        if constexpr (A_KC) store_kc(sa, stA); else store_mc(sa, stA);
        if constexpr (B_KC) store_kc(sb, stB); else store_mc(sb, stB);
    };

    if (n_kt > 0) {
        gload(0);
        sstore(0, 0);
    }
    __syncthreads();

    for (int kt = 0; kt < n_kt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < n_kt;
        if (more) gload(kt + 1);
        __builtin_amdgcn_sched_barrier(0);      // loads are issued; keep their consumers below
        const float *a_s = smem + cur * (TA + TB);
        const float *b_s = a_s + TA;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = read_frag<A_KC>(a_s, wm * 64 + i * 32, q, r, hh);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = read_frag<B_KC>(b_s, wn * 64 + j * 32, q, r, hh);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            f4(af[i], s), f4(bf[j], s), acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);      // nothing of the store phase moves above the MFMAs
        if (more) sstore(cur ^ 1, kt + 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
    float *cbase = g.c + (int64_t)blockIdx.z * g.split_stride;
    const bool add_bias = g.bias != nullptr && g.split_stride == 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cc = col0 + wn * 64 + j * 32 + r;
        if (cc >= g.n) continue;
        const float bv = add_bias ? g.bias[cc] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = row0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (rr < g.m) cbase[(int64_t)rr * g.ldc + cc] = acc[i][j][e] + bv;
            }
        }
    }
}

// C[m,n] = sum_s slab_s[m,n] + bias[n]; slabs are dense [m][n].
__global__ void splitk_reduce_kernel(const float *__restrict__ ws, int64_t slab, int splits,
                                     const float *__restrict__ bias, float *__restrict__ c,
                                     int64_t ldc, int m, int n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)m * n) return;
    const int rr = (int)(i / n), cc = (int)(i % n);
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[k * slab + i];
    if (bias) s += bias[cc];
    c[(int64_t)rr * ldc + cc] = s;
}

// Split k only when the output grid cannot fill the chip.
static int choose_splits(int64_t m, int64_t n, int64_t k) {
    const int64_t tiles = ceil_div(m, BM) * ceil_div(n, BN);
    const int64_t kt = ceil_div(k, BK);
    if (tiles >= 192 || kt < 8) return 1;
    int64_t s = 512 / tiles;           // aim at ~2 blocks per CU
    s = s < kt / 4 ? s : kt / 4;       // keep >= 4 k tiles per split
    if (s > 32) s = 32;
    return s < 1 ? 1 : (int)s;
}

template <bool A_KC, bool B_KC>
static int launch_gemm(const char *name, const float *a, int64_t lda, const float *b, int64_t ldb,
                       const float *bias, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
                       void *ws, int64_t ws_bytes, hipStream_t st) {
    if (m < 0 || n < 0 || k < 0) { set_error("%s: negative size", name); return GIST_EINVAL; }
    if (m == 0 || n == 0) return GIST_OK;
    if (!a || !b || !c) { set_error("%s: null pointer", name); return GIST_EINVAL; }
    if (m >= (1LL << 31) || n >= (1LL << 31) || k >= (1LL << 31)) {
        set_error("%s: size >= 2^31", name); return GIST_EINVAL;
    }
    GemmArgs g;
    g.a = a; g.lda = lda; g.b = b; g.ldb = ldb; g.bias = bias; g.c = c; g.ldc = ldc;
    g.m = (int)m; g.n = (int)n; g.k = (int)k;
    const bool aligned = aligned16(a) && (lda % 4 == 0) && lda >= 4 && aligned16(b) &&
                         (ldb % 4 == 0) && ldb >= 4 && k > 0;
    g.tiles_m = (int)ceil_div(m, BM);
    g.tiles_n = (int)ceil_div(n, BN);
    int splits = choose_splits(m, n, k);
    if (splits > 1 && (ws == nullptr || ws_bytes < (int64_t)splits * m * n * 4)) splits = 1;
    g.k_per_split = (int)(ceil_div(ceil_div(k, BK), splits) * BK);
    splits = (int)ceil_div(k, g.k_per_split > 0 ? g.k_per_split : 1);
    if (splits < 1) splits = 1;
    if (k == 0) { g.k_per_split = BK; splits = 1; }
    constexpr int TA = A_KC ? TILE_KC : TILE_MC;
    constexpr int TB = B_KC ? TILE_KC : TILE_MC;
    const size_t smem = (size_t)2 * (TA + TB) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&gemm_f32_kernel<A_KC, B_KC, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(
                reinterpret_cast<const void *>(&gemm_f32_kernel<A_KC, B_KC, false>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
            return GIST_ELAUNCH;
        }
        attr_set = true;
    }
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)splits);
    auto launch = [&]() {
        if (aligned)
            hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true>), grid, dim3(256), smem, st, g);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false>), grid, dim3(256), smem, st, g);
    };
    if (splits == 1) {
        g.split_stride = 0;
        launch();
        return launch_status(name);
    }
    g.c = static_cast<float *>(ws);
    g.ldc = n;
    g.split_stride = m * n;
    g.bias = nullptr;
    launch();
    int rc = launch_status(name);
    if (rc) return rc;
    const int64_t total = m * n;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                       st, static_cast<const float *>(ws), m * n, splits, bias, c, ldc, (int)m,
                       (int)n);
    return launch_status(name);
}

}  // namespace gist

extern "C" int64_t gist_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const int s = gist::choose_splits(m, n, k);
    return s > 1 ? (int64_t)s * m * n * 4 : 0;
}

extern "C" int gist_gemm_nt_f32(const float *a, int64_t lda, const float *w, int64_t ldw,
                                const float *bias, float *y, int64_t ldy, int64_t m, int64_t n,
                                int64_t k, void *workspace, int64_t workspace_bytes,
                                gist_stream_t stream) {
    GIST_REQUIRE(lda >= k && ldw >= k && ldy >= n, "gist_gemm_nt_f32: leading dimension too small");
    return gist::launch_gemm<true, true>("gist_gemm_nt_f32", a, lda, w, ldw, bias, y, ldy, m, n, k,
                                         workspace, workspace_bytes, gist::as_stream(stream));
}

extern "C" int gist_gemm_nn_f32(const float *g, int64_t ldg, const float *w, int64_t ldw, float *z,
                                int64_t ldz, int64_t m, int64_t n, int64_t k, void *workspace,
                                int64_t workspace_bytes, gist_stream_t stream) {
    GIST_REQUIRE(ldg >= k && ldw >= n && ldz >= n, "gist_gemm_nn_f32: leading dimension too small");
    return gist::launch_gemm<true, false>("gist_gemm_nn_f32", g, ldg, w, ldw, nullptr, z, ldz, m, n,
                                          k, workspace, workspace_bytes, gist::as_stream(stream));
}

extern "C" int gist_gemm_tn_f32(const float *g, int64_t ldg, const float *a, int64_t lda, float *d,
                                int64_t ldd, int64_t m, int64_t n, int64_t k, void *workspace,
                                int64_t workspace_bytes, gist_stream_t stream) {
    GIST_REQUIRE(ldg >= m && lda >= n && ldd >= n, "gist_gemm_tn_f32: leading dimension too small");
    return gist::launch_gemm<false, false>("gist_gemm_tn_f32", g, ldg, a, lda, nullptr, d, ldd, m,
                                           n, k, workspace, workspace_bytes,
                                           gist::as_stream(stream));
}
