// bf16x3 projection for shapes too small for the pre-split kernel (gemm_b3.hip): CONVERT ON LOAD.
//
// Same arithmetic as gemm_b3.hip -- every fp32 operand x = b1 + b2 + b3 exactly (three bf16 pieces, all 24
// significant bits), C = sum_k (a1.b1 + a1.b2 + a2.b1 + a2.b2 + a1.b3 + a3.b1) accumulated in fp32 on
// v_mfma_f32_16x16x32_bf16 -- but the split happens INSIDE the GEMM: a thread loads its 8 consecutive k of
// a row as fp32, forms the three pieces in registers and writes three 16-byte chunks to LDS.  There is no
// pre-pass, so there is no flop threshold to amortise one: the per-rank projections of the N = 4 / 8
// points (2046 x 1024 x 2048 and smaller), which the pre-split path only matched the fp32 kernel on
// (its operand split costs as much as it saves below ~9 GFLOP), run on the bf16 matrix cores too.
// Price: an operand tile is converted once per workgroup that uses it (VALU, ~6 instructions per
// element) instead of once per call; a k step of a 128 x 128 tile is 96 MFMAs (1536 matrix-pipe cycles)
// and ~176 VALU instructions per producer wave, against 4096 matrix-pipe cycles of the fp32 kernel.  Three
// LDS stages (3 x 48 KiB at 128 x 128, 3 x 24 KiB at 64 x 64), producer and consumer waves (see the kernel).
//
// Layouts as gemm.hip: NT (A [m,k], B [n,k]), NN (A [m,k], B [k,n]), TN (A [k,m], B [k,n]); k-contiguous
// operands are read 2 x 16 bytes per chunk, m/n-contiguous ones as 8 rows x (T / 64) columns (the
// transposition happens in registers: a chunk is one column's 8 consecutive k).  Operands must be 16-byte
// aligned with leading dimensions % 4 == 0 (everything the engine allocates); anything else stays on the
// fp32 kernel.
//
// LDS image of an operand: three planes (pieces) of [T rows][32 k] bf16 = 64 bytes per row; the 16-byte
// chunk kg (k = 8 kg .. 8 kg + 7) of row r sits at position kg ^ ((r >> 1) & 3) of its row, so the eight
// lanes of a ds_read_b128 group (rows r .. r + 7, one kg) hit eight different 16-byte slots of the
// 128-byte bank line, and so do the writes of eight consecutive chunks.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace gist {

typedef __bf16 c3_bf16x8 __attribute__((ext_vector_type(8)));
typedef float c3_f32x4 __attribute__((ext_vector_type(4)));

struct C3Args {
    const float *a; int64_t lda;
    const float *b; int64_t ldb;
    const float *bias;
    float *c; int64_t ldc;
    int m, n, k;
    int tiles_m, tiles_n;
    int k_per_split;       // multiple of 32
    int64_t split_stride;  // elements between split slabs (0 = write C directly)
    // Tail units (one k slice, more tiles than the chip holds at once; as gemm_b3.hip): the first dp_tiles tiles of the
    // launch order run whole; every other tile is cut into tail_splits k slices of tail_k (a multiple of 32), one
    // workgroup each (blockIdx.x = dp_tiles + tile * tail_splits + slice), which leave their TM x TN fp32 results
    // at tail[(tile * tail_splits + slice) * TM * TN]; gemm_b3c_tail_sum_kernel adds them in slice order (+ bias) into C.
    int dp_tiles, tail_splits, tail_k;
    float *tail;
};

constexpr int C3_BK = 32;

typedef float c3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 c3_bf16x2 __attribute__((ext_vector_type(2)));

// Two consecutive k of one row -> their packed bf16 pieces (the LDS words), two elements per instruction:
// v_cvt_pk_bf16_f32 for the piece, a shift / a mask to widen it back, v_pk_add_f32 for the (exact)
// remainder -- 9 vector instructions per PAIR (the first version converted element by element: 430
// instructions per k step at T = 128, more cycles than the step's 96 MFMAs).
__device__ __forceinline__ void c3_split2(c3_f32x2 x, uint32_t (&w)[3]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const c3_bf16x2 p = __builtin_convertvector(x, c3_bf16x2);
        w[q] = __builtin_bit_cast(uint32_t, p);
        if (q < 2) {
            c3_f32x2 back;
            back.x = __builtin_bit_cast(float, w[q] << 16);
            back.y = __builtin_bit_cast(float, w[q] & 0xffff0000u);
            // exact; two plain subtractions, NOT one v_pk_add_f32: beside another wave's MFMA stream a packed
            // f32 instruction costs ~13 cycles of the SIMD's vector issue on top of its slot (MI355X_MICROARCH.md,
            // 'price of one filler beside MFMAs'), and plain C++ is re-packed by the SLP vectoriser
            asm("v_sub_f32_e32 %0, %1, %2" : "=v"(x.x) : "v"(x.x), "v"(back.x));
            asm("v_sub_f32_e32 %0, %1, %2" : "=v"(x.y) : "v"(x.y), "v"(back.y));
        }
    }
}

// 8 consecutive k of one row -> the row's 16-byte chunk for each of the three planes
__device__ __forceinline__ void c3_split8(const float (&v)[8], uint4 (&out)[3]) {
    uint32_t w[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c3_f32x2 x;
        x.x = v[2 * j];
        x.y = v[2 * j + 1];
        c3_split2(x, w[j]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) out[q] = make_uint4(w[0][q], w[1][q], w[2][q], w[3][q]);
}

// (The builtin, not inline asm: the compiler must KNOW these are MFMAs.  As opaque asm statements it reused
// a fragment register for an accumulator's zero-initialisation one instruction after the MFMA that still
// read it as SrcA, and copied accumulators right behind the MFMA that wrote them -- both hazards the
// backend's recognizer covers for real MFMA instructions.  Found as a one-time error of 1e-4 in ONE
// accumulator per wave.)
__device__ __forceinline__ void c3_mfma(c3_f32x4 &acc, const c3_bf16x8 &a, const c3_bf16x8 &b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

// Staging registers of one operand for one k tile.
//   k-contiguous  : thread -> chunks q = t + 256 i (i < T / 64): row q / 4, octet q % 4; 2 float4 each
//   m/n-contiguous: thread -> unit t (T = 128: column pair t % 64, octet t / 64; T = 64: column t % 64,
//                   octet t / 64): 8 rows of (T / 64) floats
// The thread's part of every address is a 32-bit byte offset from the tile's origin at k = 0, computed once;
// the k advance is wave-uniform (scalar unit).
template <bool KC, int T> struct C3Stage {
    static constexpr int NCH = T / 64;                 // chunks this thread produces
    float v[NCH][8];
    uint32_t off[NCH];                                 // KC: byte offset of chunk i; MC: off[0] only
};

template <bool KC, int T>
__device__ __forceinline__ void c3_offsets(int64_t ld, int rows, int row0, C3Stage<KC, T> &st, int t) {
    if constexpr (KC) {
#pragma unroll
        for (int i = 0; i < T / 64; ++i) {
            const int q = t + 256 * i;
            const int r = min(q >> 2, rows - 1 - row0);              // rows past the end: never stored
            st.off[i] = (uint32_t)(((int64_t)r * ld + 8 * (q & 3)) * 4);
        }
    } else {
        constexpr int CW = T / 64;
        const int c = min(row0 + (t & 63) * CW, (int)ld - CW) - row0;    // inside the row pitch
        st.off[0] = (uint32_t)(((int64_t)(8 * (t >> 6)) * ld + c) * 4);
    }
}

// origin = the operand's tile origin at this k tile (KC: p + row0 * ld + k0; MC: p + k0 * ld + row0).
// FULL: the whole tile lies below k_end (no clamps, no masks).
template <bool KC, int T, bool FULL>
__device__ __forceinline__ void c3_load(const float *__restrict__ origin, int64_t ld, int k0, int k_end,
                                        C3Stage<KC, T> &st, int t) {
    const char *ob = reinterpret_cast<const char *>(origin);
    if constexpr (KC) {
#pragma unroll
        for (int i = 0; i < T / 64; ++i) {
            const float *src = reinterpret_cast<const float *>(ob + st.off[i]);
            float4 lo, hi;
            if constexpr (FULL) {
                lo = *reinterpret_cast<const float4 *>(src);
                hi = *reinterpret_cast<const float4 *>(src + 4);
            } else {            // addresses clamped inside the row pitch (ld % 4 == 0); values masked at the store
                const int kk = k0 + 8 * ((t + 256 * i) & 3);
                const int back0 = max(kk - ((int)ld - 4), 0), back1 = max(kk + 4 - ((int)ld - 4), 0);
                lo = *reinterpret_cast<const float4 *>(src - back0);
                hi = *reinterpret_cast<const float4 *>(src + 4 - back1);
            }
            st.v[i][0] = lo.x; st.v[i][1] = lo.y; st.v[i][2] = lo.z; st.v[i][3] = lo.w;
            st.v[i][4] = hi.x; st.v[i][5] = hi.y; st.v[i][6] = hi.z; st.v[i][7] = hi.w;
        }
    } else {
        constexpr int CW = T / 64;
        const int kk = k0 + 8 * (t >> 6);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int64_t step = (int64_t)j * ld * 4;                      // uniform
            if constexpr (!FULL) step = (int64_t)(min(kk + j, k_end - 1) - kk) * ld * 4;
            const float *src = reinterpret_cast<const float *>(ob + step + st.off[0]);
            if constexpr (CW == 2) {
                const float2 x = *reinterpret_cast<const float2 *>(src);
                st.v[0][j] = x.x; st.v[1][j] = x.y;
            } else {
                st.v[0][j] = *src;
            }
        }
    }
}

// The converted chunks of one operand for one k tile, and where this thread's chunks go in an LDS image.
template <int T> struct C3Out {
    uint4 w[T / 64][3];
};
template <bool KC, int T> __device__ __forceinline__ void c3_lds_offsets(uint32_t (&off)[T / 64], int t) {
#pragma unroll
    for (int i = 0; i < T / 64; ++i) {
        int r, kg;
        if constexpr (KC) { const int q = t + 256 * i; r = q >> 2; kg = q & 3; }
        else { r = (t & 63) * (T / 64) + i; kg = t >> 6; }
        off[i] = (uint32_t)(r * 64 + ((kg ^ ((r >> 1) & 3)) << 4));
    }
}

// (mask the k tail,) split (rows / columns clamped at load time carry garbage that only reaches outputs past
// m / n, which are never stored)
template <bool KC, int T, bool FULL>
__device__ __forceinline__ void c3_convert(C3Stage<KC, T> &st, C3Out<T> &out, int k0, int k_end, int t) {
#pragma unroll
    for (int i = 0; i < T / 64; ++i) {
        if constexpr (!FULL) {
            const int kk = k0 + 8 * (KC ? ((t + 256 * i) & 3) : (t >> 6));
#pragma unroll
            for (int j = 0; j < 8; ++j) st.v[i][j] = kk + j < k_end ? st.v[i][j] : 0.f;
        }
        c3_split8(st.v[i], out.w[i]);
    }
}
template <int T>
__device__ __forceinline__ void c3_write(const C3Out<T> &out, char *image, const uint32_t (&off)[T / 64]) {
    constexpr int PLANE = T * 64;
#pragma unroll
    for (int i = 0; i < T / 64; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<uint4 *>(image + q * PLANE + off[i]) = out.w[i][q];
}

// Workgroup = 8 waves with fixed roles (each SIMD hosts one wave of either kind, so the vector ALU work of
// the split runs under the other wave's MFMAs -- inside ONE wave a vector instruction next to an MFMA stream
// costs matrix-pipe time, profiles/r01_gemm_phase_trace.md; the first version of this kernel did
// convert -> barrier -> multiply -> barrier with every wave and only matched the fp32 kernel):
//   waves 4-7 PRODUCE: tile j's fp32 values (loaded three steps earlier) -> three bf16 pieces -> LDS stage
//             j % 3, then the global loads of tile j + 3;
//   waves 0-3 CONSUME: fragments of tile j -> registers while the 6 x NI x NJ MFMAs of tile j - 1 run
//             (2 x 2 waves over the TM x TN tile, two fragment register sets);
// one workgroup barrier per k step (P_j: tile j is in LDS).
#ifdef C3_PROBE   // dev build (scripts/b3c_probe.py): where a k step's cycles go, per workgroup
__device__ unsigned long long g_c3_probe[16 * 4096];
__device__ __forceinline__ unsigned long long c3_stamp() {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define C3_T() c3_stamp()
#endif

template <bool A_KC, bool B_KC, int TM, int TN>
__global__ __launch_bounds__(512, 1) void gemm_b3c_kernel(C3Args g) {
    extern __shared__ __attribute__((aligned(16))) char c3_smem[];
    constexpr int PLANE_A = TM * 64, PLANE_B = TN * 64;      // bytes of one piece plane of an operand
    constexpr int STAGE = 3 * (PLANE_A + PLANE_B);           // A planes, then B planes
    constexpr int NSTAGE = 3;
    constexpr int WM = TM / 2, WN = TN / 2;                  // consumer wave tile (2 x 2 waves)
    constexpr int NI = WM / 16, NJ = WN / 16;                // 16 x 16 MFMA tiles per wave

    // ---- block -> output tile, 8 x 8 super-tiles per XCD (as gemm.hip) ----
    const int nwg = g.dp_tiles;
    const int orig = blockIdx.x;
    const bool tailu = orig >= nwg;
    const int tail_unit = tailu ? orig - nwg : 0;
    const int qd = nwg / kXcds, rm = nwg % kXcds, xcd = orig % kXcds;
    const int L = tailu ? nwg + tail_unit / g.tail_splits
                        : (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + orig / kXcds;
    constexpr int GM = 8;
    const int width = GM * g.tiles_n;
    const int group = L / width;
    const int first_m = group * GM;
    const int gsz = min(g.tiles_m - first_m, GM);
    const int bm = first_m + (L % width) % gsz;
    const int bn = (L % width) / gsz;
    const int row0 = bm * TM, col0 = bn * TN;
    const int k_begin = tailu ? (tail_unit % g.tail_splits) * g.tail_k : blockIdx.z * g.k_per_split;
    const int k_end = min(g.k, k_begin + (tailu ? g.tail_k : g.k_per_split));
    const int n_kt = (k_end - k_begin + C3_BK - 1) / C3_BK;
    const int n_full = (k_end - k_begin) / C3_BK;            // tiles that need no mask
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    if (wave >= 4) {
        // ================================ producers ==================================================
        const int t = threadIdx.x - 256;
        // PD tiles of fp32 values in flight per thread: a tile's loads are issued PD k steps before it is
        // converted (one step of ~0.7 us is less than the latency of a load under load: with one tile in
        // flight the producers waited ~2 us per step for their operands and the kernel ran at their pace).
        // The steady state is straight-line code over the FULL tiles: loads past the last full tile re-read
        // it (never consumed), the ragged last tile sits in a register set of its own from the start.
#ifndef C3_PD64
#define C3_PD64 3
#endif
        constexpr int PD = TM + TN <= 128 ? C3_PD64 : 3;      // 16 values per operand and set at T = 64
        C3Stage<A_KC, TM> sa[PD], ta;
        C3Stage<B_KC, TN> sb[PD], tb;
        c3_offsets<A_KC, TM>(g.lda, g.m, row0, ta, t);
        c3_offsets<B_KC, TN>(g.ldb, g.n, col0, tb, t);
#pragma unroll
        for (int d = 0; d < PD; ++d) {
#pragma unroll
            for (int i = 0; i < C3Stage<A_KC, TM>::NCH; ++i) sa[d].off[i] = ta.off[i];
#pragma unroll
            for (int i = 0; i < C3Stage<B_KC, TN>::NCH; ++i) sb[d].off[i] = tb.off[i];
        }
        // tile origins at k = 0 (uniform); the k advance is added per tile on the scalar unit
        const float *oa = A_KC ? g.a + (int64_t)row0 * g.lda : g.a + row0;
        const float *ob = B_KC ? g.b + (int64_t)col0 * g.ldb : g.b + col0;
        const bool tail = n_kt > n_full;
        const int last_full = max(n_full - 1, 0);
        auto load_full = [&](int kt, C3Stage<A_KC, TM> &ra, C3Stage<B_KC, TN> &rb) {
            const int k0 = k_begin + min(kt, last_full) * C3_BK;
            c3_load<A_KC, TM, true>(A_KC ? oa + k0 : oa + (int64_t)k0 * g.lda, g.lda, k0, k_end, ra, t);
            c3_load<B_KC, TN, true>(B_KC ? ob + k0 : ob + (int64_t)k0 * g.ldb, g.ldb, k0, k_end, rb, t);
        };
        C3Out<TM> wa;
        C3Out<TN> wb;
        uint32_t la[TM / 64], lb[TN / 64];
        c3_lds_offsets<A_KC, TM>(la, t);
        c3_lds_offsets<B_KC, TN>(lb, t);
        auto convert_full = [&](C3Stage<A_KC, TM> &ra, C3Stage<B_KC, TN> &rb) {
            c3_convert<A_KC, TM, true>(ra, wa, 0, k_end, t);
            c3_convert<B_KC, TN, true>(rb, wb, 0, k_end, t);
        };
        auto write_tile = [&](int kt) {
            char *img = c3_smem + (kt % NSTAGE) * STAGE;
            c3_write<TM>(wa, img, la);
            c3_write<TN>(wb, img + 3 * PLANE_A, lb);
        };
        if (tail) {                                           // the ragged tile: masked, clamped
            const int k0 = k_begin + n_full * C3_BK;
            c3_load<A_KC, TM, false>(A_KC ? oa + k0 : oa + (int64_t)k0 * g.lda, g.lda, k0, k_end, ta, t);
            c3_load<B_KC, TN, false>(B_KC ? ob + k0 : ob + (int64_t)k0 * g.ldb, g.ldb, k0, k_end, tb, t);
        }
        if (n_full > 0) {
#pragma unroll
            for (int d = 0; d < PD; ++d) load_full(d, sa[d], sb[d]);
            convert_full(sa[0], sb[0]);
            load_full(PD, sa[0], sb[0]);
        }
        // Software pipeline.  Entering step j the converted chunks of tile j sit in (wa, wb): their LDS writes
        // are issued first, then tile j + 1 (register set (j + 1) % PD, loaded PD steps ago) is converted WHILE
        // those writes drain -- behind the consumers' fragment reads in the LDS queue: with convert -> write ->
        // wait -> barrier the write latency was on every step's critical path -- the set is refilled with tile
        // j + 1 + PD, and only then comes the wait for the writes and barrier P_j.
        int kt = 0;
#ifdef C3_PROBE
        unsigned long long pr_store = 0, pr_load = 0, pr_bar = 0, pr_cvt = 0, pr_ld = 0;
        const unsigned long long pr_t0 = C3_T();
#endif
        for (; kt + PD < n_full; kt += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
#ifdef C3_PROBE
                const unsigned long long p0 = C3_T();
                write_tile(kt + d);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long p1 = C3_T();
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * 2 * (TM + TN) / 64) : "memory");
                const unsigned long long p1b = C3_T();
                convert_full(sa[(d + 1) % PD], sb[(d + 1) % PD]);
                const unsigned long long p1c = C3_T();
                load_full(kt + d + 1 + PD, sa[(d + 1) % PD], sb[(d + 1) % PD]);
                const unsigned long long p2 = C3_T();
                __syncthreads();
                const unsigned long long p3 = C3_T();
                pr_store += p1 - p0; pr_load += p1b - p1; pr_bar += p3 - p2; pr_cvt += p1c - p1b; pr_ld += p2 - p1c;
                __builtin_amdgcn_sched_barrier(0);
                continue;
#endif
                write_tile(kt + d);
                __builtin_amdgcn_sched_barrier(0);            // (the writes stay in front of the conversion)
                convert_full(sa[(d + 1) % PD], sb[(d + 1) % PD]);
                load_full(kt + d + 1 + PD, sa[(d + 1) % PD], sb[(d + 1) % PD]);
                __syncthreads();
                // (left alone the scheduler hoists the conversions of the next two tiles above this barrier
                // and waits for ALL loads in flight at the top of the unrolled body: vmcnt(0) every 3 steps)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < PD; ++d) {                        // the last (<= PD) full tiles: kt % PD == 0 here
            if (kt + d < n_full) {
                write_tile(kt + d);
                __builtin_amdgcn_sched_barrier(0);
                if (kt + d + 1 < n_full) convert_full(sa[(d + 1) % PD], sb[(d + 1) % PD]);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (tail) {
            const int k0 = k_begin + n_full * C3_BK;
            c3_convert<A_KC, TM, false>(ta, wa, k0, k_end, t);
            c3_convert<B_KC, TN, false>(tb, wb, k0, k_end, t);
            write_tile(n_full);
            __syncthreads();
        }
#ifdef C3_PROBE
        if (threadIdx.x == 256 && blockIdx.x < 4096 && blockIdx.z == 0) {
            unsigned long long *o = g_c3_probe + 16 * blockIdx.x;
            o[4] = C3_T() - pr_t0; o[5] = pr_store; o[6] = pr_load; o[7] = pr_bar; o[8] = pr_cvt; o[9] = pr_ld;
        }
#endif
        return;
    }
    // ==================================== consumers ==================================================
    const int lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int rr = lane & 15, kg = lane >> 4;
    c3_f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    // fragment offsets inside a plane: row (16 i + rr) of the wave's slab, chunk kg at its swizzled position
    const int fa = (wm * WM + rr) * 64 + ((kg ^ ((rr >> 1) & 3)) << 4);
    const int fb = 3 * PLANE_A + (wn * WN + rr) * 64 + ((kg ^ ((rr >> 1) & 3)) << 4);
    constexpr int pa[6] = {0, 0, 1, 1, 0, 2};
    constexpr int pb[6] = {0, 1, 0, 1, 2, 0};
    if constexpr (NI * NJ >= 16) {
        // 64 x 64 wave tile (128 x 128 workgroup tile): 24 fragment reads feed 96 MFMAs (the 32 x 32 wave tile
        // reads 12 for 24 and is bound by LDS traffic and by the vector issue the conversion takes next to the
        // MFMAs: per MFMA it converts and reads twice as much).  Two fragment sets (192 registers) do not fit
        // beside 64 accumulators, so there is ONE set, refilled piece by piece as the terms retire it: within
        // step j (registers = tile j - 1, tile j in LDS) the terms run smallest first, as everywhere,
        //   a3.b1 -> a3 <- tile j | a1.b3 -> b3 | a2.b2, a2.b1 -> a2 | a1.b2 -> b2 | a1.b1 column by column,
        //   each b1 column refilled behind its four MFMAs -> a1
        // so every read is issued at least 8 MFMAs before its first use.  The reads of tile j issued last are
        // consumed during step j + 1, i.e. before this wave reaches P_{j+2} -- the barrier after which tile
        // j + 3 overwrites the stage -- so the consumers' barrier carries NO wait (a bare s_barrier).
        static_assert(NSTAGE == 3, "the late reads of a tile need its stage for one more step");
        c3_bf16x8 af[3][NI], bf[3][NJ];
        const char *img = c3_smem;
        auto rd_a = [&](int q, int i) {
            af[q][i] = *reinterpret_cast<const c3_bf16x8 *>(img + q * PLANE_A + fa + i * 16 * 64);
        };
        auto rd_b = [&](int q, int j) {
            bf[q][j] = *reinterpret_cast<const c3_bf16x8 *>(img + q * PLANE_B + fb + j * 16 * 64);
        };
        auto term = [&](int qa, int qb) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < NI; ++i) c3_mfma(acc[i][j], af[qa][i], bf[qb][j]);
        };
        auto bare_barrier = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        if (n_kt > 0) {
            bare_barrier();                                   // P_0
#pragma unroll
            for (int q = 2; q >= 0; --q) {
#pragma unroll
                for (int i = 0; i < NI; ++i) rd_a(q, i);
#pragma unroll
                for (int j = 0; j < NJ; ++j) rd_b(q, j);
            }
        }
#ifdef C3_PROBE
        unsigned long long co_bar = 0;
        const unsigned long long co_t0 = C3_T();
#endif
        for (int kt = 1; kt < n_kt; ++kt) {
#ifdef C3_PROBE
            const unsigned long long c0 = C3_T();
            bare_barrier();
            co_bar += C3_T() - c0;
#else
            bare_barrier();                                   // P_kt
#endif
            img = c3_smem + (kt % NSTAGE) * STAGE;
            term(2, 0);
#pragma unroll
            for (int i = 0; i < NI; ++i) rd_a(2, i);
            __builtin_amdgcn_sched_barrier(0);
            term(0, 2);
#pragma unroll
            for (int j = 0; j < NJ; ++j) rd_b(2, j);
            __builtin_amdgcn_sched_barrier(0);
            term(1, 1);
            term(1, 0);
#pragma unroll
            for (int i = 0; i < NI; ++i) rd_a(1, i);
            __builtin_amdgcn_sched_barrier(0);
            term(0, 1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) rd_b(1, j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int i = 0; i < NI; ++i) c3_mfma(acc[i][j], af[0][i], bf[0][j]);
                rd_b(0, j);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) rd_a(0, i);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef C3_PROBE
        if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.z == 0) {
            unsigned long long *o = g_c3_probe + 16 * blockIdx.x;
            o[0] = C3_T() - co_t0; o[1] = co_bar; o[2] = 0; o[3] = 0;
        }
#endif
        if (n_kt > 0) {
            term(2, 0); term(0, 2); term(1, 1); term(1, 0); term(0, 1); term(0, 0);
        }
    } else {
    // Barrier P_j: tile j is in LDS (stage j % NSTAGE).  After it the wave issues the fragment reads of tile j
    // into one register set and, while they fly, the MFMAs of tile j - 1 from the other set: the matrix pipe
    // never waits for an LDS round trip (with one set the reads of a tile sat between the barrier and its
    // first MFMA: ~1300 cycles per k step on top of the MFMAs).  Three stages: tile j + NSTAGE overwrites
    // tile j's stage after P_{j+2}, by when this wave has waited for its reads of tile j (before P_{j+1}).
    c3_bf16x8 af[2][NI][3], bf[2][NJ][3];
    auto read_frags = [&](int kt, auto set_c) {
        constexpr int set = decltype(set_c)::value;
        const char *img = c3_smem + (kt % NSTAGE) * STAGE;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                bf[set][j][q] = *reinterpret_cast<const c3_bf16x8 *>(img + q * PLANE_B + fb + j * 16 * 64);
#pragma unroll
            for (int i = 0; i < NI; ++i)
                af[set][i][q] = *reinterpret_cast<const c3_bf16x8 *>(img + q * PLANE_A + fa + i * 16 * 64);
        }
    };
    auto multiply = [&](auto set_c) {
        constexpr int set = decltype(set_c)::value;
        // smallest terms first; consecutive MFMAs go to NI * NJ different accumulators
#pragma unroll
        for (int term = 5; term >= 0; --term)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) c3_mfma(acc[i][j], af[set][i][pa[term]], bf[set][j][pb[term]]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    int kt = 0;
#ifdef C3_PROBE
    unsigned long long co_bar = 0, co_work = 0, co_wait = 0;
    const unsigned long long co_t0 = C3_T();
    for (; kt + 2 <= n_kt; kt += 2) {
        const unsigned long long c0 = C3_T();
        __syncthreads();
        const unsigned long long c1 = C3_T();
        read_frags(kt, S0{});
        if (kt > 0) multiply(S1{});
        const unsigned long long c2 = C3_T();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const unsigned long long c3 = C3_T();
        __syncthreads();
        const unsigned long long c4 = C3_T();
        read_frags(kt + 1, S1{});
        multiply(S0{});
        const unsigned long long c5 = C3_T();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const unsigned long long c6 = C3_T();
        co_bar += (c1 - c0) + (c4 - c3); co_work += (c2 - c1) + (c5 - c4); co_wait += (c3 - c2) + (c6 - c5);
    }
    if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.z == 0) {
        unsigned long long *o = g_c3_probe + 16 * blockIdx.x;
        o[0] = C3_T() - co_t0; o[1] = co_bar; o[2] = co_work; o[3] = co_wait;
    }
#endif
    for (; kt + 2 <= n_kt; kt += 2) {
        __syncthreads();                                      // P_kt
        read_frags(kt, S0{});
        if (kt > 0) multiply(S1{});
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this tile's fragments are in
        __syncthreads();                                      // P_{kt+1}
        read_frags(kt + 1, S1{});
        multiply(S0{});
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    if (kt < n_kt) {                                          // odd tile count: one more
        __syncthreads();
        read_frags(kt, S0{});
        if (kt > 0) multiply(S1{});
        __builtin_amdgcn_s_waitcnt(0xc07f);
        multiply(S0{});
    } else if (n_kt > 0) {
        multiply(S1{});
    }
    }      // (the two-set consumer of the 32 x 32 / 64 x 32 wave tiles)
    // ---- epilogue: C/D of 16x16x32: col = lane & 15, row = 4 (lane >> 4) + e ----
    if (tailu) {      // this slice's tile, dense, for gemm_b3c_tail_sum_kernel
        float *tb = g.tail + (int64_t)tail_unit * (TM * TN);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    tb[(wm * WM + i * 16 + 4 * kg + e) * TN + wn * WN + j * 16 + rr] = acc[i][j][e];
        return;
    }
    float *cbase = g.c + (int64_t)blockIdx.z * g.split_stride;
    const bool add_bias = g.bias != nullptr && g.split_stride == 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = col0 + wn * WN + j * 16 + rr;
        if (col >= g.n) continue;
        const float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = row0 + wm * WM + i * 16 + 4 * kg + e;
                if (row < g.m) cbase[(int64_t)row * g.ldc + col] = acc[i][j][e] + bv;
            }
    }
}

// C tile = sum over its k slices of the tail units' results, in slice order (+ bias).  Grid (tail tile, TM * TN / 1024).
__global__ __launch_bounds__(256) void gemm_b3c_tail_sum_kernel(C3Args g, int tm, int tn) {
    const int L = g.dp_tiles + (int)blockIdx.x;
    constexpr int GM = 8;
    const int width = GM * g.tiles_n, first_m = (L / width) * GM, gsz = min(g.tiles_m - first_m, GM);
    const int row0 = (first_m + (L % width) % gsz) * tm, col0 = ((L % width) / gsz) * tn;
    const int e4 = ((int)blockIdx.y * 256 + (int)threadIdx.x) * 4;      // four consecutive columns of one tile row
    const int r = e4 / tn, c = e4 % tn;
    if (r >= tm || row0 + r >= g.m) return;
    const float *src = g.tail + (int64_t)blockIdx.x * g.tail_splits * (tm * tn) + e4;
    float4 sum = *reinterpret_cast<const float4 *>(src);
    for (int q = 1; q < g.tail_splits; ++q) {
        const float4 t = *reinterpret_cast<const float4 *>(src + (int64_t)q * (tm * tn));
        sum.x += t.x; sum.y += t.y; sum.z += t.z; sum.w += t.w;
    }
    const float v[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int col = col0 + c + u;
        if (col < g.n) g.c[(int64_t)(row0 + r) * g.ldc + col] = v[u] + (g.bias != nullptr ? g.bias[col] : 0.f);
    }
}

// ---- host side ----------------------------------------------------------------------------------
// Which calls take this kernel by default.  MEASURED (scripts/b3c_bench.py, profiles/r03_b3c_bench.txt; us,
// standalone calls, fp32 kernel -> this kernel with 64 x 64 tiles and one k slice): NT 2046 x 1024 x 2048
// 76 -> 59, NT 2046 x 1024 x 1204 50 -> 43, NN 2046 x 2048 x 1024 73 -> 66, NT 2046 x 512 x 1204 24.9 -> 22.3;
// TN (both operands k-major: 16 scalar row loads per thread and step) 1024 x 2048 x 2046 68.5 -> 64, but
// 1024 x 1204 x 2046 54 -> 60 and 512 x 1024 x 2046 25 -> 43.  So by default the NT and NN layouts from
// 2 GFLOP with at least 256 tiles of 64 x 64 come here and TN stays on the fp32 kernel; the tuning hook
// GIST_TUNE_B3C = 2 sends every shape with m, n, k >= 64 (tests, sweeps), 1 none.  A k step still costs
// several hundred cycles on top of its MFMAs; the matrix pipe is far from saturated.
static bool b3c_geometry_ok(int64_t m, int64_t n, int64_t k) {
    return h3_mode() == 2 && (int)tune(GIST_TUNE_B3C) != 1 && m >= 64 && n >= 64 && k >= 64;
}
bool b3c_shape_ok(int64_t m, int64_t n, int64_t k) {
    if (!b3c_geometry_ok(m, n, k)) return false;
    if ((int)tune(GIST_TUNE_B3C) == 2) return true;
    return ceil_div(m, 64) * ceil_div(n, 64) >= 256 && 2.0 * (double)m * (double)n * (double)k >= 2e9;
}

template <bool A_KC, bool B_KC, int TM, int TN>
static int b3c_launch(const char *name, C3Args &g, int splits, hipStream_t st) {
    const size_t smem = (size_t)3 * 3 * (TM + TN) * 64;      // three stages of (A, B) x 3 planes
    static DeviceOnce once;       // one per template instance
    int dev;
    if (once.needed(&dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_b3c_kernel<A_KC, B_KC, TM, TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
            return GIST_ELAUNCH;
        }
        once.done(dev);
    }
    g.tiles_m = (int)ceil_div(g.m, TM);
    g.tiles_n = (int)ceil_div(g.n, TN);
    const int tiles = g.tiles_m * g.tiles_n;
    unsigned grid = (unsigned)tiles;
    if (g.tail_splits > 1) grid = (unsigned)(g.dp_tiles + (tiles - g.dp_tiles) * g.tail_splits);
    else g.dp_tiles = tiles;
    hipLaunchKernelGGL((gemm_b3c_kernel<A_KC, B_KC, TM, TN>), dim3(grid, 1, (unsigned)splits), dim3(512), smem, st, g);
    int rc = launch_status(name);
    if (rc == GIST_OK && g.tail_splits > 1) {
        hipLaunchKernelGGL(gemm_b3c_tail_sum_kernel, dim3((unsigned)(tiles - g.dp_tiles), (unsigned)(TM * TN / 1024)), dim3(256), 0, st,
                           g, TM, TN);
        rc = launch_status(name);
    }
    return rc;
}

// Tile and k slices (fp32 slabs, reduced by the call or left to the consumer).  Fitted to scripts/b3c_bench.py.
static void b3c_choice(int64_t m, int64_t n, int64_t k, int *tm, int *tn, int *splits) {
    const int t_tile = (int)tune(GIST_TUNE_GEMM_TILE), t_split = (int)tune(GIST_TUNE_GEMM_SPLITS);
    auto tiles = [&](int a, int b) { return ceil_div(m, a) * ceil_div(n, b); };
    // 64 x 64 unless the output has 256 tiles of 128 x 128 (measured: NN 2046 x 2048 x 1024 62 -> 54 us; with fewer
    // tiles the 128 x 128 grid leaves CUs idle or needs k slices and loses: NT 2046 x 1024 x 2048 59 vs 82 / 60 with
    // 1 / 2 slices; 128 x 64 is slower than 64 x 64 on every per-rank shape).  A k step of the 128 x 128 tile takes
    // ~2700 cycles for 1536 of MFMA, one of the 64 x 64 tile ~1200 for 384: the producers' step (176 / 88 vector
    // instructions, 12 / 6 ds_write_b128 at ~13 issue cycles, 8 / 4 loads) sets the pace in both, and neither deeper
    // load prefetch (PD 4-6), nor LDS writes issued before the conversion, nor cheaper instructions (v_perm for
    // v_cvt_pk, v_sub for v_pk_add), nor half as many MFMA issues (32 x 32 x 16) moved it (scripts/b3c_probe.py).
    *tm = 64; *tn = 64;
    if (tiles(128, 128) >= 256) { *tm = 128; *tn = 128; }
    if (t_tile == 64) { *tm = 64; *tn = 64; }
    if (t_tile == 128 || t_tile == 12864) { *tm = 128; *tn = 64; }
    if (t_tile == 128128) { *tm = 128; *tn = 128; }
    const int64_t wgs = tiles(*tm, *tn);
    const int64_t kt = ceil_div(k, C3_BK);
    int64_t sp = 1;
    if (wgs < 128) sp = 256 / wgs;                            // (only reachable through the tuning hook)
    if (sp > kt / 8) sp = kt / 8;                             // a slice keeps >= 8 k tiles
    if (t_split > 0) sp = t_split;
    if ((int)tune(GIST_TUNE_B3C_SPLITS) > 0) sp = (int)tune(GIST_TUNE_B3C_SPLITS);
    if (sp > kt) sp = kt;
    *splits = (int)(sp < 1 ? 1 : sp);
}

// Tail units (as gemm_b3.hip's): one k slice and T tiles on S workgroup slots (64 x 64 tiles: two workgroups per CU = 512,
// the larger tiles one = 256) with 0 < T mod S <= S / 2 -- a batch of 2049-2112 rows makes 528 tiles of 64 x 64 out of 512,
// 272 of 128 x 128 out of 256: a second round for 16 tiles -- cut the tiles of the last round into k slices of >= 4 k tiles.
struct C3Tail { int dp_tiles, splits, k_per; };
static C3Tail b3c_tail(int64_t m, int64_t n, int64_t k, int tm, int tn) {
    const int64_t tiles = ceil_div(m, tm) * ceil_div(n, tn), slots = (tm == 64 && tn == 64) ? 512 : 256;
    C3Tail t{(int)tiles, 1, 0};
    if ((int)tune(GIST_TUNE_B3_TAIL) == 1 || tiles <= slots) return t;
    const int64_t r = tiles % slots, kt = ceil_div(k, C3_BK);
    if (r == 0 || r > slots / 2) return t;
    int64_t s = std::min<int64_t>(slots / r, kt / 4);
    if (s < 2) return t;
    const int64_t per = ceil_div(kt, s);
    s = ceil_div(kt, per);
    if (s < 2) return t;
    t.dp_tiles = (int)(tiles - r); t.splits = (int)s; t.k_per = (int)(per * C3_BK);
    return t;
}
// (tail = false: a layout this path does not take by default -- TN, the weight gradients: its scratch must not make a
// slab buffer exist that the caller's routing reads as "this projection's slices are summed by its consumer")
int64_t b3c_slab_bytes(int64_t m, int64_t n, int64_t k, bool tail) {
    if (!b3c_shape_ok(m, n, k)) return -1;
    int tm, tn, sp;
    b3c_choice(m, n, k, &tm, &tn, &sp);
    if (sp > 1) return (int64_t)sp * m * n * 4;
    if (!tail) return 0;
    const C3Tail t = b3c_tail(m, n, k, tm, tn);
    return t.splits > 1 ? (ceil_div(m, tm) * ceil_div(n, tn) - t.dp_tiles) * (int64_t)t.splits * tm * tn * 4 : 0;
}

// Returns 1 if the projection was issued here, 0 if this call is not for this path (the caller falls back
// to the fp32 kernel), < 0 on error.  deferred as in gemm.hip's launch_gemm.
int b3c_gemm(const char *name, bool a_kc, bool b_kc, const float *a, int64_t lda, const float *b, int64_t ldb,
             const float *bias, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k, void *ws,
             int64_t ws_bytes, hipStream_t st, int *deferred) {
    if (!b3c_shape_ok(m, n, k)) return 0;
    if (!(aligned16(a) && aligned16(b) && lda % 4 == 0 && ldb % 4 == 0 && lda >= 4 && ldb >= 4)) return 0;
    if (!a_kc && b_kc) return 0;                                     // (no caller uses this layout)
    if ((int)tune(GIST_TUNE_B3C) != 2 && !a_kc) return 0;            // by default: NT and NN only (see above)
    int tm = 128, tn = 128, splits = 1;
    b3c_choice(m, n, k, &tm, &tn, &splits);
    if (splits > 1 && (ws == nullptr || ws_bytes < (int64_t)splits * m * n * 4)) splits = 1;
    C3Args g;
    g.a = a; g.lda = lda; g.b = b; g.ldb = ldb; g.bias = bias; g.c = c; g.ldc = ldc;
    g.m = (int)m; g.n = (int)n; g.k = (int)k;
    g.k_per_split = (int)(ceil_div(ceil_div(k, C3_BK), splits) * C3_BK);
    splits = (int)ceil_div(k, g.k_per_split);
    g.split_stride = 0;
    if (splits > 1) { g.c = static_cast<float *>(ws); g.ldc = n; g.split_stride = m * n; g.bias = nullptr; }
    g.dp_tiles = 0; g.tail_splits = 1; g.tail_k = 0; g.tail = nullptr;
    if (splits == 1 && ws != nullptr && aligned16(ws)) {
        const C3Tail t = b3c_tail(m, n, k, tm, tn);
        if (t.splits > 1 && ws_bytes >= (ceil_div(m, tm) * ceil_div(n, tn) - t.dp_tiles) * (int64_t)t.splits * tm * tn * 4) {
            g.dp_tiles = t.dp_tiles; g.tail_splits = t.splits; g.tail_k = t.k_per; g.tail = static_cast<float *>(ws);
        }
    }
    const int64_t slot = timer_begin(tl_timer, 2, m, n, k, st);      // kind 2: a bf16x3 main kernel
    int rc;
#define C3_GO(AK, BK_)                                                                        \
    rc = tm == 64 ? b3c_launch<AK, BK_, 64, 64>(name, g, splits, st)                          \
         : tn == 64 ? b3c_launch<AK, BK_, 128, 64>(name, g, splits, st)                       \
                    : b3c_launch<AK, BK_, 128, 128>(name, g, splits, st)
    if (a_kc && b_kc) { C3_GO(true, true); }
    else if (a_kc) { C3_GO(true, false); }
    else { C3_GO(false, false); }
#undef C3_GO
    if (rc == GIST_OK && splits > 1) {
        if (deferred) *deferred = splits;
        else rc = splitk_reduce(name, static_cast<const float *>(ws), m * n, splits, bias, c, ldc, m, n, st);
    }
    timer_end(tl_timer, slot, st);
    return rc == GIST_OK ? 1 : rc;
}

}  // namespace gist

#ifdef C3_PROBE
extern "C" int gist_c3_probe_read(unsigned long long *out, int64_t n_blocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gist::g_c3_probe), n_blocks * 16 * sizeof(unsigned long long)) ==
                   hipSuccess ? 0 : -1;
}
#endif
