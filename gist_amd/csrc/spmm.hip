// CSR SpMM neighbour aggregation for gfx950 (MI355X).
//
// y[v,:] = (acc ? y[v,:] : 0) + out_scale[v] * sum_e src_scale[col[e]] * x[col[e],:]
//
// Replaces g.update_all(fn.copy_src, fn.sum) * norm  (reference
// cluster_gcn/modules.py:223-226) and its autograd (reverse graph, src_scale).
//
// Mapping: one 64-lane wavefront owns one (row, column tile).  The wave reads the
// row's column indices 64 at a time with one coalesced load, then walks them:
//   * wide rows (d/VEC >= 64 lanes): every lane holds VEC consecutive floats of the
//     feature tile; the neighbour id is broadcast with v_readlane (scalar address
//     math), 4 neighbour rows in flight per lane;
//   * narrow rows: the wave is split into G = 64/LPR lane groups, group g takes
//     neighbours g, g+G, ... and the groups are combined by a butterfly
//     (wavefront segmented reduction), so no lane idles on d = 41..128.
// HBM-bound: per launch the algorithmic traffic is rowptr + col + X once + Y once
// (DESIGN.md).  Column tiles are dealt to XCDs (blockIdx % 8 shares an L2) so that
// one tile of X (n_rows x 1 KiB) is gathered out of ONE L2 instead of eight.
#include <stdlib.h>

#include "common.h"

namespace gist {

template <int VEC> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };

template <int VEC>
__device__ __forceinline__ void vload(const float *p, float (&v)[VEC]) {
    using T = typename Vec<VEC>::T;
    T t = *reinterpret_cast<const T *>(p);
    if constexpr (VEC == 1) { v[0] = t; }
    if constexpr (VEC == 2) { v[0] = t.x; v[1] = t.y; }
    if constexpr (VEC == 4) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
}

template <int VEC>
__device__ __forceinline__ void vstore(float *p, const float (&v)[VEC]) {
    using T = typename Vec<VEC>::T;
    T t;
    if constexpr (VEC == 1) { t = v[0]; }
    if constexpr (VEC == 2) { t.x = v[0]; t.y = v[1]; }
    if constexpr (VEC == 4) { t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; }
    *reinterpret_cast<T *>(p) = t;
}

constexpr int kSpmmWavesPerBlock = 4;

template <int VEC, int LPR>
__global__ __launch_bounds__(256) void spmm_csr_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ x, int64_t ldx, float *__restrict__ y, int64_t ldy,
    int n_rows, int d, const float *__restrict__ out_scale,
    const float *__restrict__ src_scale, int accumulate, int n_row_blocks,
    int n_col_tiles, int xcd_tiles) {
    constexpr int G = kWave / LPR;
    // ---- block -> (row block, column tile) -------------------------------
    int rb, ct;
    const int b = blockIdx.x;
    if (xcd_tiles) {  // tiles dealt to XCDs: blocks b, b+8, b+16.. share an L2
        const int xcd = b % kXcds, i = b / kXcds;
        ct = xcd + kXcds * (i / n_row_blocks);
        rb = i % n_row_blocks;
        if (ct >= n_col_tiles) return;
    } else {
        rb = b / n_col_tiles;
        ct = b % n_col_tiles;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = rb * kSpmmWavesPerBlock + wave;
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int c0 = (ct * LPR + sub) * VEC;
    const bool active = c0 < d;

    const int beg = __builtin_amdgcn_readfirstlane(rowptr[row]);
    const int end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);

    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;

    const float *xc = x + c0;
    for (int base = beg; base < end; base += kWave) {
        const int e = base + lane;
        const int my = (e < end) ? col[e] : 0;
        float mys = 1.f;
        if (src_scale) mys = (e < end) ? src_scale[my] : 0.f;
        const int cnt = min(kWave, end - base);
        if constexpr (G == 1) {
            int j = 0;
            for (; j + 4 <= cnt; j += 4) {
                float v[4][VEC];
                float s[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int u = __builtin_amdgcn_readlane(my, j + t);
                    s[t] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                                          __builtin_bit_cast(int, mys), j + t));
                    if (active) vload<VEC>(xc + (int64_t)u * ldx, v[t]);
                }
                if (active) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(s[t], v[t][k], acc[k]);
                }
            }
            for (; j < cnt; ++j) {
                const int u = __builtin_amdgcn_readlane(my, j);
                const float s = __builtin_bit_cast(
                    float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mys), j));
                if (active) {
                    float v[VEC];
                    vload<VEC>(xc + (int64_t)u * ldx, v);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(s, v[k], acc[k]);
                }
            }
        } else {
            // group g handles neighbours g, g+G, ...; two in flight per lane.  The trip
            // count is wave-uniform so every lane is live at each cross-lane read.
            const int iters = (cnt + 2 * G - 1) / (2 * G);
            for (int t = 0; t < iters; ++t) {
                const int j = grp + t * 2 * G, j2 = j + G;
                const bool p0 = j < cnt, p1 = j2 < cnt;
                const int u0 = __shfl(my, p0 ? j : 0), u1 = __shfl(my, p1 ? j2 : 0);
                const float t0 = __shfl(mys, p0 ? j : 0), t1 = __shfl(mys, p1 ? j2 : 0);
                const float s0 = p0 ? t0 : 0.f, s1 = p1 ? t1 : 0.f;
                if (active) {
                    float v0[VEC], v1[VEC];
                    vload<VEC>(xc + (int64_t)u0 * ldx, v0);
                    vload<VEC>(xc + (int64_t)u1 * ldx, v1);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(s0, v0[k], acc[k]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(s1, v1[k], acc[k]);
                }
            }
        }
    }
    if constexpr (G > 1) {  // butterfly across the lane groups
#pragma unroll
        for (int off = LPR; off < kWave; off <<= 1)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], off);
    }
    if (active && grp == 0) {
        const float os = out_scale ? out_scale[row] : 1.f;
        float *yp = y + (int64_t)row * ldy + c0;
        float o[VEC];
        if (accumulate) {
            vload<VEC>(yp, o);
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = fmaf(os, acc[k], o[k]);
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = os * acc[k];
        }
        vstore<VEC>(yp, o);
    }
}

// ---------------------------------------------------------------------------------------
// Wide rows (d/VEC >= 64 lanes): ONE WORKGROUP PER ROW.  A single wave walking a row is a
// chain of dependent L2 round trips (4 loads per trip): a 330-neighbour hub row of a
// power-law graph takes ~80 trips and sets the kernel's duration.  Here the row's
// neighbour list is dealt to the 4 waves in interleaved slices of 16 (wave w takes
// neighbours [16w, 16w+16) of every 64), 8 row reads in flight per wave (16 in flight cost
// 158 VGPRs and ran slower), and the four partial sums are combined through LDS in wave
// order (deterministic).  A 64-neighbour row costs 2 trips per wave instead of 16; a hub
// row 11 instead of 83.
//
// HALF (VEC = 4 only): d % 4 == 2 with 16-byte aligned source rows but an output that is only
// 8-byte aligned -- the layer-0 aggregation of F = 602 features inside Z_0 = [h | ah], whose right
// half starts 2408 bytes into a 4816-byte row.  Sources are still gathered 16 bytes per lane (the
// last lane of a row reads two floats past d, inside the row pitch, and never stores them);
// the result leaves as two 8-byte stores.
template <int VEC, bool HALF = false>
__global__ __launch_bounds__(256) void spmm_csr_rowsplit_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ x, int64_t ldx, float *__restrict__ y, int64_t ldy,
    int n_rows, int d, const float *__restrict__ out_scale,
    const float *__restrict__ src_scale, int accumulate, int n_col_tiles, int chunk_rows) {
    static_assert(!HALF || VEC == 4, "HALF is the 16-byte-load / 8-byte-store variant");
    __shared__ float part[3][kWave * VEC];
    // Row chunks (~ one METIS part: consecutive batch rows whose neighbours are mostly in the
    // same chunk) are dealt to XCDs -- workgroups b, b+8, b+16, ... share an L2 -- so the
    // rows an XCD gathers are the ones its own L2 already holds, instead of every L2
    // missing on all of X.  (Speed only: the result does not depend on placement.)
    const int b = blockIdx.x;
    const int xcd = b % kXcds, i = b / kXcds;
    const int per = chunk_rows * n_col_tiles;
    const int rem = i % per;
    const int row = (xcd + kXcds * (i / per)) * chunk_rows + rem / n_col_tiles;
    const int ct = rem % n_col_tiles;
    if (row >= n_rows) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int c0 = (ct * kWave + lane) * VEC;
    const bool active = c0 < d;
    const int beg = __builtin_amdgcn_readfirstlane(rowptr[row]);
    const int end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);

    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    const float *xc = x + c0;
    // lanes 0..15 of wave w hold the ids of neighbours base + 16w + lane
    for (int base = beg + 16 * wave; base < end; base += kWave) {
        const int e = base + (lane & 15);
        const bool ok = (lane < 16) && (e < end);
        const int my = ok ? col[e] : 0;
        float mys = 1.f;
        if (src_scale) mys = ok ? src_scale[my] : 0.f;
        const int cnt = min(16, end - base);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (8 * h < cnt) {                            // wave-uniform
                float v[8][VEC];
                float s[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int jj = min(8 * h + t, cnt - 1);
                    const int u = __builtin_amdgcn_readlane(my, jj);
                    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                                                    __builtin_bit_cast(int, mys), jj));
                    s[t] = (8 * h + t < cnt) ? s0 : 0.f;   // clamped duplicates contribute 0
                    if (active) vload<VEC>(xc + (int64_t)u * ldx, v[t]);
                }
                if (active) {
#pragma unroll
                    for (int t = 0; t < 8; ++t)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(s[t], v[t][k], acc[k]);
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) part[wave - 1][lane * VEC + k] = acc[k];
    }
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += part[w][lane * VEC + k];
        const float os = out_scale ? out_scale[row] : 1.f;
        float *yp = y + (int64_t)row * ldy + c0;
        if constexpr (HALF) {
            const bool upper = c0 + 2 < d;                 // false only for the row's last lane
            float lo[2] = {0.f, 0.f}, hi[2] = {0.f, 0.f};
            if (accumulate) {
                vload<2>(yp, lo);
                if (upper) vload<2>(yp + 2, hi);
            }
            lo[0] = fmaf(os, acc[0], lo[0]); lo[1] = fmaf(os, acc[1], lo[1]);
            hi[0] = fmaf(os, acc[2], hi[0]); hi[1] = fmaf(os, acc[3], hi[1]);
            vstore<2>(yp, lo);
            if (upper) vstore<2>(yp + 2, hi);
        } else {
            float o[VEC];
            if (accumulate) {
                vload<VEC>(yp, o);
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = fmaf(os, acc[k], o[k]);
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = os * acc[k];
            }
            vstore<VEC>(yp, o);
        }
    }
}

// ---------------------------------------------------------------------------------------
// LDS-staged kernel for cluster batches.  Rows come in LOCALITY BLOCKS (a METIS part of a
// Cluster-GCN batch: ~100 consecutive rows whose neighbours are mostly in the same part).
// A 512-thread workgroup (8 waves) owns (row block, row split, column tile) and stages, with
// coalesced loads, into LDS:
//   * the block's X tile [<=128 rows x 64*VEC floats], PRE-SCALED by src_scale (backward
//     form), plus one all-zero row;
//   * for the rows it computes: a 1-BYTE local row index per neighbour (zero row for a
//     cross-block neighbour or padding), and up to 8 (global id, scale) pairs of its
//     cross-block neighbours per row.
// Per neighbour the inner loop is then v_readlane + shift + v_add + ds_read_b128 + adds --
// no scale, no branch, no global memory; the few cross-block rows are prefetched from
// global memory BEFORE the LDS pass and consumed after it.  Every element of X leaves
// L2/HBM once per workgroup; nnz*D*4 bytes of L2 gather traffic become LDS reads.
// Rows whose lists do not fit (hubs) take a per-row generic path; results never depend on
// how rows are split into blocks.
constexpr int kLdsRB = 128;          // rows staged per block
constexpr int kLdsListCap = 8192;    // neighbour entries (1 byte each) per workgroup
constexpr int kLdsRemCap = 8;        // cross-block neighbours kept per row
constexpr int kLdsThreads = 512;

template <int VEC>
struct LdsLayout {
    static constexpr int CW = kWave * VEC;
    static constexpr size_t tile_bytes = (size_t)(kLdsRB + 1) * CW * 4;
    static constexpr size_t idx_off = tile_bytes;
    static constexpr size_t rp_off = idx_off + kLdsListCap;
    static constexpr size_t rid_off = rp_off + (kLdsRB + 4) * 4;
    static constexpr size_t rsc_off = rid_off + (size_t)kLdsRB * kLdsRemCap * 4;
    static constexpr size_t rcn_off = rsc_off + (size_t)kLdsRB * kLdsRemCap * 4;
    static constexpr size_t total = rcn_off + kLdsRB * 4;
};

template <int VEC>
__global__ __launch_bounds__(512) void spmm_csr_lds_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ x, int64_t ldx, float *__restrict__ y, int64_t ldy,
    int n_rows, int d, const float *__restrict__ out_scale,
    const float *__restrict__ src_scale, int accumulate,
    const int32_t *__restrict__ row_blocks, int n_col_tiles, int row_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    using LL = LdsLayout<VEC>;
    constexpr int CW = LL::CW;
    constexpr int NW = kLdsThreads / kWave;
    constexpr int SH = VEC == 4 ? 10 : (VEC == 2 ? 9 : 8);        // log2(CW * 4)
    float *tile = reinterpret_cast<float *>(smem_b);
    unsigned char *idx8 = smem_b + LL::idx_off;
    int32_t *rp = reinterpret_cast<int32_t *>(smem_b + LL::rp_off);
    int32_t *rid = reinterpret_cast<int32_t *>(smem_b + LL::rid_off);
    float *rsc = reinterpret_cast<float *>(smem_b + LL::rsc_off);
    int32_t *rcn = reinterpret_cast<int32_t *>(smem_b + LL::rcn_off);

    const int ct = blockIdx.x % n_col_tiles;
    const int rest = blockIdx.x / n_col_tiles;
    const int half = rest % row_split;
    const int rbk = rest / row_split;
    int r0, r1;
    if (row_blocks) { r0 = row_blocks[rbk]; r1 = row_blocks[rbk + 1]; }
    else { r0 = rbk * kLdsRB; r1 = min(n_rows, r0 + kLdsRB); }
    r1 = min(r1, n_rows);
    const int nrow = r1 - r0;
    if (nrow <= 0) return;
    const int nloc = min(nrow, kLdsRB);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = ct * CW + lane * VEC;
    const bool active = c0 < d;

    // ---- stage 1: row pointers, X tile (pre-scaled), zero row ----------------------------
    for (int i = threadIdx.x; i <= nloc; i += kLdsThreads) rp[i] = rowptr[r0 + i];
    {
        float z[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) z[k] = 0.f;
        if (wave == 0) vstore<VEC>(tile + kLdsRB * CW + lane * VEC, z);
    }
    for (int rr = wave; rr < nloc; rr += 8 * NW) {       // eight row loads in flight per wave
        float v[8][VEC];
        float sc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int r = rr + t * NW;
            const int rc = min(r, nloc - 1);
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[t][k] = 0.f;
            if (active) vload<VEC>(x + (int64_t)(r0 + rc) * ldx + c0, v[t]);
            sc[t] = src_scale ? src_scale[r0 + rc] : 1.f;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int r = rr + t * NW;
            if (r < nloc) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) v[t][k] *= sc[t];
                vstore<VEC>(tile + r * CW + lane * VEC, v[t]);
            }
        }
    }
    __syncthreads();

    // ---- stage 2: classify the neighbours of the rows THIS workgroup computes ----------------
    const int cbeg = rp[0];
    for (int lr = half + row_split * wave; lr < nloc; lr += row_split * NW) {
        const int beg = rp[lr], end = rp[lr + 1];
        int nrem = 0;
        const bool fits = (end - cbeg) <= kLdsListCap;
        for (int base = beg; base < end && fits; base += kWave) {
            const int e = base + lane;
            const bool in = e < end;
            const int u = in ? col[e] : r0;
            const unsigned lu = (unsigned)(u - r0);
            const bool local = lu < (unsigned)nloc;
            if (in) idx8[e - cbeg] = (unsigned char)(local ? lu : kLdsRB);
            const unsigned long long m = __ballot(in && !local);
            if (in && !local) {
                const int pos = nrem + __popcll(m & ((1ULL << lane) - 1ULL));
                if (pos < kLdsRemCap) {
                    rid[lr * kLdsRemCap + pos] = u;
                    rsc[lr * kLdsRemCap + pos] = src_scale ? src_scale[u] : 1.f;
                }
            }
            nrem += __popcll(m);
        }
        if (lane == 0) rcn[lr] = fits ? nrem : -1;       // -1 / > cap: generic path for the row
    }
    __syncthreads();

    // ---- compute -----------------------------------------------------------------------------
    const float *xc = x + c0;
    const unsigned char *tb = reinterpret_cast<const unsigned char *>(tile) + lane * VEC * 4;
    for (int lr = half + row_split * wave; lr < nrow; lr += row_split * NW) {
        const int row = r0 + lr;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        const int nrem = lr < nloc ? rcn[lr] : -1;
        if (nrem >= 0 && nrem <= kLdsRemCap) {
            const int beg = __builtin_amdgcn_readfirstlane(rp[lr]);
            const int end = __builtin_amdgcn_readfirstlane(rp[lr + 1]);
            // cross-block rows first: their global loads fly under the LDS pass
            float rv[kLdsRemCap][VEC];
            float rs[kLdsRemCap];
            const int rc = __builtin_amdgcn_readfirstlane(nrem);
            const int my_id = lane < kLdsRemCap ? rid[lr * kLdsRemCap + lane] : row;
            const float my_sc = lane < kLdsRemCap ? rsc[lr * kLdsRemCap + lane] : 0.f;
            if (rc > 0) {
#pragma unroll
                for (int t = 0; t < kLdsRemCap; ++t) {
                    const bool on = t < rc;                                  // wave-uniform
                    const int u = on ? __builtin_amdgcn_readlane(my_id, t) : row;
                    rs[t] = on ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                                               __builtin_bit_cast(int, my_sc), t))
                               : 0.f;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) rv[t][k] = 0.f;
                    if (on && active) vload<VEC>(xc + (int64_t)u * ldx, rv[t]);
                }
            }
            for (int base = beg; base < end; base += kWave) {
                const int e = base + lane;
                const int mine = (e < end) ? (int)idx8[e - cbeg] : kLdsRB;   // pad -> zero row
                const int cnt = min(kWave, end - base);
                // groups of 8 LDS row reads, software pipelined: group g+1 is issued before
                // group g is summed, so the LDS pipe always has work queued
                float va[8][VEC], vb[8][VEC];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int li = __builtin_amdgcn_readlane(mine, t);
                    vload<VEC>(reinterpret_cast<const float *>(tb + ((unsigned)li << SH)), va[t]);
                }
                for (int j = 0; j < cnt; j += 16) {
                    if (j + 8 < cnt) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const int li = __builtin_amdgcn_readlane(mine, (j + 8 + t) & 63);
                            vload<VEC>(reinterpret_cast<const float *>(tb + ((unsigned)li << SH)), vb[t]);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[k] += va[t][k];
                    if (j + 8 < cnt) {
                        if (j + 16 < cnt) {
#pragma unroll
                            for (int t = 0; t < 8; ++t) {
                                const int li = __builtin_amdgcn_readlane(mine, (j + 16 + t) & 63);
                                vload<VEC>(reinterpret_cast<const float *>(tb + ((unsigned)li << SH)), va[t]);
                            }
                        }
#pragma unroll
                        for (int t = 0; t < 8; ++t)
#pragma unroll
                            for (int k = 0; k < VEC; ++k) acc[k] += vb[t][k];
                    }
                }
            }
            if (rc > 0) {
#pragma unroll
                for (int t = 0; t < kLdsRemCap; ++t)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(rs[t], rv[t][k], acc[k]);
            }
        } else {
            // generic path: hub rows / oversize blocks -- gather everything from global memory
            const int beg = __builtin_amdgcn_readfirstlane(rowptr[row]);
            const int end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
            for (int base = beg; base < end; base += kWave) {
                const int e = base + lane;
                const int my = (e < end) ? col[e] : 0;
                float mys = 1.f;
                if (src_scale) mys = (e < end) ? src_scale[my] : 0.f;
                const int cnt = min(kWave, end - base);
                for (int j = 0; j < cnt; j += 4) {
                    float v[4][VEC];
                    float sc[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int jj = min(j + t, cnt - 1);
                        const int u = __builtin_amdgcn_readlane(my, jj);
                        const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                                                        __builtin_bit_cast(int, mys), jj));
                        sc[t] = (j + t < cnt) ? s0 : 0.f;
#pragma unroll
                        for (int k = 0; k < VEC; ++k) v[t][k] = 0.f;
                        if (active) vload<VEC>(xc + (int64_t)u * ldx, v[t]);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(sc[t], v[t][k], acc[k]);
                }
            }
        }
        if (active) {
            const float os = out_scale ? out_scale[row] : 1.f;
            float *yp = y + (int64_t)row * ldy + c0;
            float o[VEC];
            if (accumulate) {
                vload<VEC>(yp, o);
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = fmaf(os, acc[k], o[k]);
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = os * acc[k];
            }
            vstore<VEC>(yp, o);
        }
    }
}

__global__ void in_degree_norm_kernel(const int32_t *__restrict__ rowptr, int64_t n,
                                      float *__restrict__ norm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int deg = rowptr[i + 1] - rowptr[i];
        norm[i] = deg > 0 ? 1.f / (float)deg : 0.f;
    }
}

template <int VEC, bool HALF = false>
static int launch_spmm(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx,
                       float *y, int64_t ldy, int64_t n_rows, int64_t d, const float *out_scale,
                       const float *src_scale, int accumulate, hipStream_t st) {
    const int lanes = (int)ceil_div(d, VEC);
    int lpr = 8;
    while (lpr < 64 && lpr < lanes) lpr <<= 1;
    const int n_col_tiles = (int)ceil_div(lanes, lpr);
    const int n_row_blocks = (int)ceil_div(n_rows, kSpmmWavesPerBlock);
    const int xcd_tiles = n_col_tiles >= 8 ? 1 : 0;
    if (lpr == 64) {   // wide rows: one workgroup per (row, column tile), waves share the row
        const int chunk_rows = tune(GIST_TUNE_SPMM_CHUNK) > 0.0 ? (int)tune(GIST_TUNE_SPMM_CHUNK) : 128;
        const int64_t n_chunks = ceil_div(n_rows, chunk_rows);
        const int64_t g2 = kXcds * ceil_div(n_chunks, kXcds) * chunk_rows * n_col_tiles;
        if (g2 > 0x7fffffffLL) {
            set_error("gist_spmm_csr_f32: grid too large");
            return GIST_EINVAL;
        }
        hipLaunchKernelGGL((spmm_csr_rowsplit_kernel<VEC, HALF>), dim3((unsigned)g2), dim3(256), 0,
                           st, rowptr, col, x, ldx, y, ldy, (int)n_rows, (int)d, out_scale,
                           src_scale, accumulate, n_col_tiles, chunk_rows);
        return launch_status("gist_spmm_csr_f32");
    }
    const int64_t grid = xcd_tiles ? (int64_t)kXcds * ceil_div(n_col_tiles, kXcds) * n_row_blocks
                                   : (int64_t)n_row_blocks * n_col_tiles;
    if (grid > 0x7fffffffLL) {
        set_error("gist_spmm_csr_f32: grid too large");
        return GIST_EINVAL;
    }
#define GIST_SPMM_LAUNCH(L)                                                                   \
    hipLaunchKernelGGL((spmm_csr_kernel<VEC, L>), dim3((unsigned)grid), dim3(256), 0, st,     \
                       rowptr, col, x, ldx, y, ldy, (int)n_rows, (int)d, out_scale,           \
                       src_scale, accumulate, n_row_blocks, n_col_tiles, xcd_tiles)
    switch (lpr) {
        case 8: GIST_SPMM_LAUNCH(8); break;
        case 16: GIST_SPMM_LAUNCH(16); break;
        case 32: GIST_SPMM_LAUNCH(32); break;
        default: GIST_SPMM_LAUNCH(64); break;
    }
#undef GIST_SPMM_LAUNCH
    return launch_status("gist_spmm_csr_f32");
}

template <int VEC>
static int launch_spmm_blocked(const int32_t *rowptr, const int32_t *col, const float *x,
                               int64_t ldx, float *y, int64_t ldy, int64_t n_rows, int64_t d,
                               const float *out_scale, const float *src_scale, int accumulate,
                               const int32_t *row_blocks, int64_t n_row_blocks, hipStream_t st) {
    const int n_col_tiles = (int)ceil_div(d, kWave * VEC);
    const int64_t nb = row_blocks ? n_row_blocks : ceil_div(n_rows, kLdsRB);
    // split a block's rows over several workgroups (each stages the whole tile) until the
    // grid can keep 256 CUs busy for a couple of rounds
    int row_split = (int)ceil_div(640, nb * n_col_tiles);
    row_split = row_split < 1 ? 1 : (row_split > 4 ? 4 : row_split);
    const int64_t grid = nb * row_split * n_col_tiles;
    if (grid > 0x7fffffffLL) { set_error("gist_spmm_csr_blocked_f32: grid too large"); return GIST_EINVAL; }
    const size_t smem = LdsLayout<VEC>::total;
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        hipError_t e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&spmm_csr_lds_kernel<VEC>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            set_error("gist_spmm_csr_blocked_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return GIST_ELAUNCH;
        }
        once.done(dev);
    }
    hipLaunchKernelGGL((spmm_csr_lds_kernel<VEC>), dim3((unsigned)grid), dim3(kLdsThreads), smem, st,
                       rowptr, col, x, ldx, y, ldy, (int)n_rows, (int)d, out_scale, src_scale,
                       accumulate, row_blocks, n_col_tiles, row_split);
    return launch_status("gist_spmm_csr_blocked_f32");
}

}  // namespace gist

extern "C" int gist_in_degree_norm_f32(const int32_t *rowptr, int64_t n_rows, float *norm,
                                       gist_stream_t stream) {
    GIST_REQUIRE(n_rows >= 0, "gist_in_degree_norm_f32: n_rows < 0");
    if (n_rows == 0) return GIST_OK;
    GIST_REQUIRE(rowptr && norm, "gist_in_degree_norm_f32: null pointer");
    hipLaunchKernelGGL(gist::in_degree_norm_kernel, dim3((unsigned)gist::ceil_div(n_rows, 256)),
                       dim3(256), 0, gist::as_stream(stream), rowptr, n_rows, norm);
    return gist::launch_status("gist_in_degree_norm_f32");
}

extern "C" int gist_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *x,
                                 int64_t ldx, float *y, int64_t ldy, int64_t n_rows, int64_t d,
                                 const float *out_scale, const float *src_scale, int accumulate,
                                 gist_stream_t stream) {
    using namespace gist;
    GIST_REQUIRE(n_rows >= 0 && d >= 0, "gist_spmm_csr_f32: negative size");
    if (n_rows == 0 || d == 0) return GIST_OK;
    GIST_REQUIRE(rowptr && x && y, "gist_spmm_csr_f32: null pointer");
    GIST_REQUIRE(ldx >= d && ldy >= d, "gist_spmm_csr_f32: leading dimension < d");
    GIST_REQUIRE(n_rows < (1LL << 31) && d < (1LL << 31), "gist_spmm_csr_f32: size >= 2^31");
    hipStream_t st = as_stream(stream);
    if (d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y))
        return launch_spmm<4>(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                              accumulate, st);
    // d = 4q + 2 wide enough for the workgroup-per-row kernel, 16-byte source rows with room
    // for the last lane's over-read, 8-byte destination: gather 16 bytes per lane anyway
    if (d % 4 == 2 && d >= 4 * kWave && ldx % 4 == 0 && ldx >= d + 2 && aligned16(x) && ldy % 2 == 0 &&
        aligned8(y))
        return launch_spmm<4, true>(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                                    accumulate, st);
    if (d % 2 == 0 && ldx % 2 == 0 && ldy % 2 == 0 && aligned8(x) && aligned8(y))
        return launch_spmm<2>(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                              accumulate, st);
    return launch_spmm<1>(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                          accumulate, st);
}

extern "C" int gist_spmm_csr_blocked_f32(const int32_t *rowptr, const int32_t *col, const float *x,
                                         int64_t ldx, float *y, int64_t ldy, int64_t n_rows,
                                         int64_t d, const float *out_scale,
                                         const float *src_scale, int accumulate,
                                         const int32_t *row_blocks, int64_t n_row_blocks,
                                         gist_stream_t stream) {
    using namespace gist;
    GIST_REQUIRE(n_rows >= 0 && d >= 0, "gist_spmm_csr_blocked_f32: negative size");
    if (n_rows == 0 || d == 0) return GIST_OK;
    GIST_REQUIRE(rowptr && x && y, "gist_spmm_csr_blocked_f32: null pointer");
    GIST_REQUIRE(ldx >= d && ldy >= d, "gist_spmm_csr_blocked_f32: leading dimension < d");
    GIST_REQUIRE(n_rows < (1LL << 31) && d < (1LL << 31), "gist_spmm_csr_blocked_f32: size >= 2^31");
    GIST_REQUIRE(row_blocks == nullptr || n_row_blocks > 0,
                 "gist_spmm_csr_blocked_f32: row_blocks given but n_row_blocks <= 0");
    // narrow rows keep every lane busy only in the lane-group kernel
    if (d < 128)
        return gist_spmm_csr_f32(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                                 accumulate, stream);
    hipStream_t st = as_stream(stream);
    if (d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y))
        return launch_spmm_blocked<4>(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale,
                                      src_scale, accumulate, row_blocks, n_row_blocks, st);
    if (d % 2 == 0 && ldx % 2 == 0 && ldy % 2 == 0 && aligned8(x) && aligned8(y))
        return launch_spmm_blocked<2>(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale,
                                      src_scale, accumulate, row_blocks, n_row_blocks, st);
    return gist_spmm_csr_f32(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                             accumulate, stream);
}
