// Device-resident cluster batch extraction and IST weight-block movement (gfx950).
//
// Node-induced subgraph of a union of METIS parts (reference:
// cluster_gcn/partition_utils.py:20-25 -> DGL g.subgraph on the CPU, followed by an
// H2D copy every iteration, cluster_gcn_ist_distrib.py:409).  Here the training
// graph stays in HBM and one wavefront filters one adjacency row:
//   mark      remap[ids[i]] = i
//   rowptr    per row: count neighbours with remap >= 0 (ballot + popcount), then an
//             exclusive scan (single workgroup; a batch has ~2k rows)
//   fill      per row: compact the kept neighbours in their original order
//             (ballot prefix), relabelled through remap
// Integer work, bit exact against the oracle.  Bound: HBM/L2 reads of the selected
// adjacency rows, 2 x 4 B per full-graph edge of the batch rows.
#include "common.h"
#include "adam_body.h"

namespace gist {

__global__ void mark_kernel(const int32_t *__restrict__ ids, int64_t n, int32_t *__restrict__ remap,
                            int unmark) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) remap[ids[i]] = unmark ? -1 : (int32_t)i;
}

__global__ void fill_i32_kernel(int32_t *__restrict__ p, int64_t n, int32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

// deg[i] -> out[i + 1]
__global__ __launch_bounds__(256) void induced_count_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ ids, int n_ids, const int32_t *__restrict__ remap,
    int32_t *__restrict__ out) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_ids) return;
    const int lane = threadIdx.x & 63;
    const int v = ids[i];
    const int beg = rowptr[v], end = rowptr[v + 1];
    int cnt = 0;
    for (int base = beg; base < end; base += kWave) {
        const int e = base + lane;
        const bool keep = (e < end) && (remap[col[e]] >= 0);
        cnt += __popcll(__ballot(keep));
    }
    if (lane == 0) out[i + 1] = cnt;
}

// In-place inclusive scan of p[1..n] (p[0] = 0) by ONE workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void scan_rowptr_kernel(int32_t *__restrict__ p, int n) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) { p[0] = 0; carry_s = 0; }
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + t;
        int v = (i < n) ? p[i + 1] : 0;
        // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(v, off);
            if (lane >= off) v += o;
        }
        if (lane == 63) wsum[w] = v;
        __syncthreads();
        int pre = carry_s;
        for (int k = 0; k < w; ++k) pre += wsum[k];
        if (i < n) p[i + 1] = v + pre;
        __syncthreads();
        if (t == 1023) carry_s = v + pre;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void induced_fill_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ ids, int n_ids, const int32_t *__restrict__ remap,
    const int32_t *__restrict__ sub_rowptr, int32_t *__restrict__ sub_col, int64_t capacity) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_ids) return;
    const int lane = threadIdx.x & 63;
    const int v = ids[i];
    const int beg = rowptr[v], end = rowptr[v + 1];
    int64_t w = sub_rowptr[i];
    for (int base = beg; base < end; base += kWave) {
        const int e = base + lane;
        int r = -1;
        if (e < end) r = remap[col[e]];
        const unsigned long long m = __ballot(r >= 0);
        if (r >= 0) {
            const int64_t pos = w + __popcll(m & ((1ULL << lane) - 1ULL));
            if (pos < capacity) sub_col[pos] = r;
        }
        w += __popcll(m);
    }
}

// ---- fused batch extraction (5 launches instead of 11) -----------------------------------
// blockIdx.y selects the structure: 0 = in-edge CSR (also writes norm = 1/in-degree),
// 1 = out-edge CSR.
struct CsrPair {
    const int32_t *rowptr[2];
    const int32_t *col[2];
    int32_t *sub_rowptr[2];
    int32_t *sub_col[2];
};

__global__ __launch_bounds__(256) void induced_count2_kernel(CsrPair p,
                                                             const int32_t *__restrict__ ids,
                                                             int n_ids,
                                                             const int32_t *__restrict__ remap,
                                                             float *__restrict__ norm) {
    const int which = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_ids) return;
    const int lane = threadIdx.x & 63;
    const int32_t *rowptr = p.rowptr[which], *col = p.col[which];
    const int v = ids[i];
    const int beg = rowptr[v], end = rowptr[v + 1];
    int cnt = 0;
    for (int base = beg; base < end; base += kWave) {
        const int e = base + lane;
        const bool keep = (e < end) && (remap[col[e]] >= 0);
        cnt += __popcll(__ballot(keep));
    }
    if (lane == 0) {
        p.sub_rowptr[which][i + 1] = cnt;
        if (which == 0) norm[i] = cnt > 0 ? 1.f / (float)cnt : 0.f;
    }
}

// one 1024-thread workgroup per structure
__global__ __launch_bounds__(1024) void scan_rowptr2_kernel(int32_t *p0, int32_t *p1, int n) {
    int32_t *__restrict__ p = blockIdx.x == 0 ? p0 : p1;
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) { p[0] = 0; carry_s = 0; }
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + t;
        int v = (i < n) ? p[i + 1] : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(v, off);
            if (lane >= off) v += o;
        }
        if (lane == 63) wsum[w] = v;
        __syncthreads();
        int pre = carry_s;
        for (int k = 0; k < w; ++k) pre += wsum[k];
        if (i < n) p[i + 1] = v + pre;
        __syncthreads();
        if (t == 1023) carry_s = v + pre;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void induced_fill2_kernel(CsrPair p,
                                                            const int32_t *__restrict__ ids,
                                                            int n_ids,
                                                            const int32_t *__restrict__ remap,
                                                            int64_t capacity) {
    const int which = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_ids) return;
    const int lane = threadIdx.x & 63;
    const int32_t *rowptr = p.rowptr[which], *col = p.col[which];
    int32_t *sub_col = p.sub_col[which];
    const int v = ids[i];
    const int beg = rowptr[v], end = rowptr[v + 1];
    int64_t w = p.sub_rowptr[which][i];
    for (int base = beg; base < end; base += kWave) {
        const int e = base + lane;
        int r = -1;
        if (e < end) r = remap[col[e]];
        const unsigned long long m = __ballot(r >= 0);
        if (r >= 0) {
            const int64_t pos = w + __popcll(m & ((1ULL << lane) - 1ULL));
            if (pos < capacity) sub_col[pos] = r;
        }
        w += __popcll(m);
    }
}

// Layer 0's dropout folded into the gather (gist_extract_batch_drop): z0 receives the features under
// gist_dropout_f32's mask (element index offset + i * mask_ld + c), x0 the features themselves (the
// source of layer 0's aggregation).
struct GatherDrop {
    float *x0; int64_t ldx0;
    float p, scale;
    uint64_t sm, offset;
    int64_t mask_ld;
};
__device__ __forceinline__ float gather_keep(uint64_t idx, const GatherDrop &g) {
    const uint64_t h = splitmix64((idx >> 1) + g.sm);
    const uint32_t w = (idx & 1) ? (uint32_t)(h >> 32) : (uint32_t)h;
    return ((float)(w >> 8) * (1.0f / 16777216.0f) >= g.p) ? g.scale : 0.f;
}

// features + label of batch row i, and remap[ids[i]] back to -1 (runs after both fills)
template <int VEC, bool DROP = false>
__global__ __launch_bounds__(256) void gather_batch_kernel(const float *__restrict__ feat,
                                                           int64_t ld_feat,
                                                           const int32_t *__restrict__ ids,
                                                           int n_ids, int d,
                                                           float *__restrict__ z0, int64_t ldz0,
                                                           const int32_t *__restrict__ labels_all,
                                                           int32_t *__restrict__ labels,
                                                           int32_t *__restrict__ remap, GatherDrop gd) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_ids) return;
    const int lane = threadIdx.x & 63;
    const int v = ids[i];
    const float *s = feat + (int64_t)v * ld_feat;
    float *o = z0 + (int64_t)i * ldz0;
    for (int c = lane * VEC; c < d; c += kWave * VEC) {
        if constexpr (DROP) {
            float t[VEC];
            float *o2 = gd.x0 + (int64_t)i * gd.ldx0;
            const uint64_t i0 = gd.offset + (uint64_t)i * (uint64_t)gd.mask_ld + (uint64_t)c;
#pragma unroll
            for (int k = 0; k < VEC; ++k) t[k] = s[c + k];
#pragma unroll
            for (int k = 0; k < VEC; ++k) o2[c + k] = t[k];
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[c + k] = t[k] * gather_keep(i0 + k, gd);
            continue;
        }
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(o + c) = *reinterpret_cast<const float4 *>(s + c);
        else if constexpr (VEC == 2) *reinterpret_cast<float2 *>(o + c) = *reinterpret_cast<const float2 *>(s + c);
        else o[c] = s[c];
    }
    if (lane == 0) {
        if (labels_all) labels[i] = labels_all[v];
        remap[v] = -1;
    }
}

// ---- the whole extraction in ONE launch, for batches that are unions of parts ------------------------
// A cluster batch is the union of `batch_size` parts of a fixed partition (partition_utils.py:20-25), so
// "is node u in the batch, and at which row" needs no mark pass: with node_part[u] = (part of u, position
// of u in it) (static) and the epoch's table part_slot[p] = (which batch of the epoch part p belongs to,
// at which row of it), row(u) = slot.batch == j ? slot.row + position : -1.
// grid = (ceil(n / 16), 3) workgroups of 16 waves, one wave per batch row:
//   y = 0 / 1  induced in-edge / out-edge CSR: count the row's kept neighbours (ballot + popcount, the
//              kept ones' new ids stashed in LDS in edge order), publish the workgroup's total, sum the
//              totals of the workgroups before this one (decoupled look-back, below) -> row pointers, copy
//              the stash (rows with more than 256 kept neighbours walk their list again);
//   y = 2      feature + label gather (optionally with layer 0's dropout), independent of the others.
// A workgroup that waits longer than ~1 s for a predecessor gives up and raises the error word instead
// of hanging the queue.
// Integer work: bit exact against the 5-launch path.
struct PartsArgs {
    CsrPair p;
    const int32_t *ids;
    int n, n_max;
    const int32_t *node_part, *part_slot;   // [N][2] = (part, position), [parts][2] = (batch, first row)
    int batch;
    int64_t capacity;
    float *norm;
    unsigned long long *error;       // scratch word [1]: raised when a workgroup gave up waiting
    unsigned long long *slots;       // scratch: [2][gridDim.x] (launch epoch << 31 | kept-neighbour total)
    unsigned long long *tickets;     // scratch: [2] chunk tickets of the two CSR passes (0 between launches)
    unsigned long long epoch;        // this launch's number (never 0, never reused with this scratch)
    // gather
    const float *feat; int64_t ld_feat; int d;
    float *z0; int64_t ldz0;
    const int32_t *labels_all; int32_t *labels;
    GatherDrop gd; int drop;
    // layer 0's aggregation, formed here (NULL: not): ah[i] = norm[i] . (feat_intra[v] + sum of feat[u] over the kept
    // in-neighbours u of v = ids[i] OUTSIDE v's part), feat_intra[v] = the sum over v's in-neighbours inside its part
    const float *feat_intra; int64_t ld_intra;
    float *ah;                       // row i at ah + i * ldz0
};

constexpr int kStash = 256;
constexpr int kRemote = 64;          // outside-part neighbours of a row listed in LDS (more: the list is walked again)
constexpr int kPartsWaves = 16;

// Layer 0's aggregation of one row by one wave, rows of at most NT * 64 <= 1024 features: every load of the row's
// part-internal sum is issued before the first use, the neighbours outside the part are taken RB at a time with all their
// loads in flight (a loop of load / add per neighbour and 256-column piece was a chain of 3 x (1 + n_remote) memory
// latencies: +20 us on the kernel).  Loads are BUFFER loads: a row is a buffer of d floats (scalar descriptor), a lane's
// offset one register for the whole row (4 * lane + an immediate per piece), columns past the row's end read as zero by
// the hardware's range check -- no per-load address pairs, no clamping, straight-line code, and the kernel keeps the 64
// registers that let two of these 1024-thread workgroups share a CU (with 64-bit addresses per load: 124).  Sums in edge
// order.
// n_remote > kRemote (round 5: a row of a community cut into two parts of one batch has 30-80 neighbours in the sibling
// part, a hub among them hundreds): `refill(skip)` lists the next (at most kRemote) outside neighbours behind the first
// `skip` into `rem` and returns how many -- the same register-resident accumulation, one more walk of the row's edge list
// per 64 neighbours (a load / add loop per neighbour and 256-column piece cost 110-320 us per launch on every tenth batch
// of the power-law community graph: profiles/r05_unplanted_graph.json).
template <int NT, int RB, typename Refill>
__device__ __forceinline__ void aggregate_row_regs(const PartsArgs &a, const int i, const int v_self, const int cnt,
                                                   const int n_remote_all, const int32_t *rem, const int lane, Refill refill) {
    const float nrm = cnt > 0 ? 1.f / (float)cnt : 0.f;
    const int row_bytes = a.d * 4;
    auto row = [&](const float *base, int64_t ld, int r) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base) + (int64_t)__builtin_amdgcn_readfirstlane(r) * ld, 0,
                                                 row_bytes, 0x00020000);
    };
    const int voff = lane * 4;
    float acc[NT];
    {
        const __amdgpu_buffer_rsrc_t rs = row(a.feat_intra, a.ld_intra, v_self);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + t * 256, 0, 0));
    }
    for (int done = 0; done < n_remote_all || done == 0;) {
    const int n_remote = n_remote_all <= kRemote ? n_remote_all : refill(done);
    int k = 0;
    for (; k + RB <= n_remote; k += RB) {
        float u[RB][NT];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const __amdgpu_buffer_rsrc_t rs = row(a.feat, a.ld_feat, rem[k + r]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                u[r][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + t * 256, 0, 0));
        }
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] += u[r][t];
    }
    if (RB > 1) {
        for (; k < n_remote; ++k) {
            const __amdgpu_buffer_rsrc_t rs = row(a.feat, a.ld_feat, rem[k]);
            float u[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                u[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + t * 256, 0, 0));
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] += u[t];
        }
    }
    done += n_remote;
    if (n_remote == 0) break;
    }
    float *__restrict__ o = a.ah + (int64_t)i * a.ldz0;
    const uint64_t i0 = a.gd.offset + (uint64_t)i * (uint64_t)a.gd.mask_ld + (uint64_t)a.d;      // mask index of ah[i][0]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = t * kWave + lane;
        if (c >= a.d) continue;
        float val = acc[t] * nrm;
        if (a.drop) val *= gather_keep(i0 + (uint64_t)c, a.gd);
        o[c] = val;
    }
}

// workgroup (bx, by) of a (gx, 3) grid of 16-wave workgroups (extract_parts_kernel; adam_extract_kernel below runs
// the same workgroups in one grid with the optimiser's)
__device__ __forceinline__ void extract_parts_block(const PartsArgs &a, const int bx, const int by, const int gx) {
    __shared__ int32_t stash[kPartsWaves][kStash];
    __shared__ int32_t remote[kPartsWaves][kRemote];     // global ids of a row's kept neighbours outside its part
    __shared__ int wcnt[kPartsWaves];
    __shared__ int wbase;
    __shared__ int s_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Which 16-row chunk this workgroup filters: for the two CSR passes (y = 0 / 1) a TICKET, not blockIdx.x --
    // the look-back below waits for the chunks BEFORE this one, so chunk numbers must follow the order in which
    // workgroups actually started, and HIP promises no dispatch order.  One returning atomic per workgroup; the
    // workgroup that draws the last ticket puts the counter back to zero for the next (stream-ordered) launch.
    int tile = bx;
    if (by < 2) {
        if (threadIdx.x == 0) {
            unsigned long long *tk = a.tickets + by;
            const unsigned long long t = __hip_atomic_fetch_add(tk, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == (unsigned long long)gx)
                __hip_atomic_store(tk, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_tile = (int)t;
        }
        __syncthreads();
        tile = s_tile;
    }
    const int i = tile * kPartsWaves + wave;
    if (by == 2) {                                   // ---- features + label of row i
        if (i >= a.n) return;
        const int v = a.ids[i];
        const float *__restrict__ s = a.feat + (int64_t)v * a.ld_feat;
        float *__restrict__ o = a.z0 + (int64_t)i * a.ldz0;
        float *__restrict__ o2 = a.drop ? a.gd.x0 + (int64_t)i * a.gd.ldx0 : nullptr;
        const uint64_t i0 = a.gd.offset + (uint64_t)i * (uint64_t)a.gd.mask_ld;
        // eight loads of the row in flight before the first store (load -> store -> load through pointers the
        // compiler must assume to alias is a chain of memory latencies: -0.8 us of the kernel's 20; by parts, back
        // to back from the host: launch + gather 9.3 us, + count 2.5, + publish / fill 4.3, + look-back 3.1)
        for (int c0 = 0; c0 < a.d; c0 += 8 * kWave) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u * kWave + lane;
                t[u] = c < a.d ? s[c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u * kWave + lane;
                if (c >= a.d) continue;
                if (a.drop) {
                    o2[c] = t[u];
                    o[c] = t[u] * gather_keep(i0 + (uint64_t)c, a.gd);
                } else {
                    o[c] = t[u];
                }
            }
        }
        if (lane == 0 && a.labels_all) a.labels[i] = a.labels_all[v];
        return;
    }
    const int which = by;
    const int32_t *rowptr = a.p.rowptr[which], *col = a.p.col[which];
    const int2 *node_part = reinterpret_cast<const int2 *>(a.node_part);
    const int2 *part_slot = reinterpret_cast<const int2 *>(a.part_slot);
    const bool live = i < a.n;
    const bool agg = which == 0 && a.ah != nullptr;          // this pass also forms layer 0's aggregation of its row
    int beg = 0, end = 0, v_self = 0, my_part = -1;
    if (live) {
        v_self = a.ids[i];
        beg = rowptr[v_self];
        end = rowptr[v_self + 1];
        if (agg) my_part = node_part[v_self].x;
    }
    auto new_id = [&](int e, bool in, int &u, int &part) {   // row of neighbour e in this batch, or -1
        u = -1; part = -1;
        if (!in) return -1;
        u = col[e];
        const int2 np = node_part[u];                        // (part, position in the part)
        const int2 ps = part_slot[np.x];                     // (batch of the epoch, first row)
        part = np.x;
        return ps.x == a.batch ? ps.y + np.y : -1;
    };
    int cnt = 0, n_remote = 0;
    for (int base = beg; base < end; base += 2 * kWave) {    // two chunks' loads in flight
        const int e0 = base + lane, e1 = e0 + kWave;
        int u0, u1, p0, p1;
        const int r0 = new_id(e0, e0 < end, u0, p0), r1 = new_id(e1, e1 < end, u1, p1);
        const unsigned long long m0 = __ballot(r0 >= 0), m1 = __ballot(r1 >= 0);
        const unsigned long long below = (1ULL << lane) - 1ULL;
        if (r0 >= 0) {
            const int pos = cnt + __popcll(m0 & below);
            if (pos < kStash) stash[wave][pos] = r0;
        }
        cnt += __popcll(m0);
        if (r1 >= 0) {
            const int pos = cnt + __popcll(m1 & below);
            if (pos < kStash) stash[wave][pos] = r1;
        }
        cnt += __popcll(m1);
        if (agg) {                                           // kept neighbours outside the row's part, in edge order
            const bool x0 = r0 >= 0 && p0 != my_part, x1 = r1 >= 0 && p1 != my_part;
            const unsigned long long q0 = __ballot(x0), q1 = __ballot(x1);
            if (x0) {
                const int pos = n_remote + __popcll(q0 & below);
                if (pos < kRemote) remote[wave][pos] = u0;
            }
            n_remote += __popcll(q0);
            if (x1) {
                const int pos = n_remote + __popcll(q1 & below);
                if (pos < kRemote) remote[wave][pos] = u1;
            }
            n_remote += __popcll(q1);
        }
    }
    // layer 0's aggregation of this row: ah = norm . (the part's own contribution, summed once for the whole run, +
    // the neighbours in the batch's OTHER parts).  Runs where the wave would otherwise wait: waves 1-15 between the
    // two barriers below (while wave 0 walks the look-back), wave 0 after them
    auto aggregate_row = [&]() {
        if (!live) return;
        if (a.d <= 16 * kWave) {                            // (everything but very wide inputs)
            // the next <= kRemote outside neighbours behind the first `skip`, in edge order (another walk of the list)
            auto refill = [&](int skip) {
                int seen = 0;
                for (int base = beg; base < end && seen < skip + kRemote; base += kWave) {
                    int u, pt;
                    const int r = new_id(base + lane, base + lane < end, u, pt);
                    const bool x = r >= 0 && pt != my_part;
                    const unsigned long long q = __ballot(x);
                    const int pos = seen + __popcll(q & ((1ULL << lane) - 1ULL)) - skip;
                    if (x && pos >= 0 && pos < kRemote) remote[wave][pos] = u;
                    seen += __popcll(q);
                }
                return min(seen - skip, kRemote);
            };
            const int nt = (a.d + kWave - 1) / kWave;
            if (nt <= 2) aggregate_row_regs<2, 4>(a, i, v_self, cnt, n_remote, remote[wave], lane, refill);
            else if (nt <= 4) aggregate_row_regs<4, 4>(a, i, v_self, cnt, n_remote, remote[wave], lane, refill);
            else if (nt <= 8) aggregate_row_regs<8, 2>(a, i, v_self, cnt, n_remote, remote[wave], lane, refill);
            else if (nt <= 10) aggregate_row_regs<10, 2>(a, i, v_self, cnt, n_remote, remote[wave], lane, refill);
            else aggregate_row_regs<16, 1>(a, i, v_self, cnt, n_remote, remote[wave], lane, refill);
            return;
        }
        const float nrm = cnt > 0 ? 1.f / (float)cnt : 0.f;
        const float *__restrict__ pin = a.feat_intra + (int64_t)v_self * a.ld_intra;
        float *__restrict__ o = a.ah + (int64_t)i * a.ldz0;
        const uint64_t i0 = a.gd.offset + (uint64_t)i * (uint64_t)a.gd.mask_ld + (uint64_t)a.d;      // mask index of ah[i][0]
        for (int c0 = 0; c0 < a.d; c0 += 4 * kWave) {
            float acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = c0 + t * kWave + lane;
                acc[t] = c < a.d ? pin[c] : 0.f;
            }
            if (n_remote <= kRemote) {
                for (int k = 0; k < n_remote; ++k) {
                    const float *__restrict__ xr = a.feat + (int64_t)remote[wave][k] * a.ld_feat;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int c = c0 + t * kWave + lane;
                        if (c < a.d) acc[t] += xr[c];
                    }
                }
            } else {                                         // a hub: its list again, in edge order
                for (int base = beg; base < end; base += kWave) {
                    int u, pt;
                    const int r = new_id(base + lane, base + lane < end, u, pt);
                    unsigned long long q = __ballot(r >= 0 && pt != my_part);
                    while (q) {
                        const int b = __builtin_ctzll(q);
                        q &= q - 1;
                        const float *__restrict__ xr = a.feat + (int64_t)__builtin_amdgcn_readlane(u, b) * a.ld_feat;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int c = c0 + t * kWave + lane;
                            if (c < a.d) acc[t] += xr[c];
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = c0 + t * kWave + lane;
                if (c >= a.d) continue;
                float val = acc[t] * nrm;
                if (a.drop) val *= gather_keep(i0 + (uint64_t)c, a.gd);
                o[c] = val;
            }
        }
    };
    if (lane == 0) {
        wcnt[wave] = cnt;
        if (live && which == 0) a.norm[i] = cnt > 0 ? 1.f / (float)cnt : 0.f;
    }
    // ---- exclusive prefix over the workgroups: decoupled look-back, no barrier, no read-modify-write --
    // Each workgroup PUBLISHES (launch epoch << 31 | its kept-neighbour total) in its own slot with one
    // device-scope store and reads the slots of the workgroups BEFORE it (wave 0: up to 128 slots, two per
    // lane), re-reading the ones whose epoch is not this launch's yet.  A workgroup waits only for
    // lower-numbered ones, which STARTED earlier (chunk numbers are tickets drawn at start) and which wait for
    // nobody above them, so the chain always drains -- no co-residency and no dispatch-order condition.  (Measured first: a grid barrier on ONE ticket
    // counter.  Device-scope read-modify-writes on one address complete at ~0.1-0.3 us each on this
    // multi-die part: 260 arrivals took 26 us, 1030 took 340; device-scope FENCES write back and
    // invalidate the XCD's whole L2, which the gather workgroups keep dirty: 206 us.)
    unsigned long long *slots = a.slots + (int64_t)which * gx;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < kPartsWaves; ++k) t += wcnt[k];
        __hip_atomic_store(slots + tile, (a.epoch << 31) | (unsigned long long)(unsigned)t, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {
        const int nb = tile;                                 // slots to read: [0, nb)
        unsigned long long v0 = 0, v1 = 0;
        long spins = 0;
        for (;;) {
            const bool need0 = lane < nb && (v0 >> 31) != a.epoch;
            const bool need1 = lane + kWave < nb && (v1 >> 31) != a.epoch;
            if (need0) v0 = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (need1) v1 = __hip_atomic_load(slots + lane + kWave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool miss = (lane < nb && (v0 >> 31) != a.epoch) || (lane + kWave < nb && (v1 >> 31) != a.epoch);
            if (!__ballot(miss)) break;
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1L << 22)) {                      // ~1 s: give up loudly, never hang
                if (lane == 0) __hip_atomic_store(a.error, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        int part = 0;
        if (lane < nb) part += (int)(v0 & 0x7fffffffULL);
        if (lane + kWave < nb) part += (int)(v1 & 0x7fffffffULL);
        for (int k = lane + 2 * kWave; k < nb; k += kWave) {      // more than 128 workgroups before this one
            unsigned long long v = 0;
            long sp2 = 0;
            while (((v = __hip_atomic_load(slots + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 31) != a.epoch) {
                __builtin_amdgcn_s_sleep(8);
                if (++sp2 > (1L << 22)) {
                    __hip_atomic_store(a.error, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            part += (int)(v & 0x7fffffffULL);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0) wbase = part;
    } else if (agg) {
        aggregate_row();
    }
    __syncthreads();
    if (agg && wave == 0) aggregate_row();
    int w = wbase;
    for (int k = 0; k < wave; ++k) w += wcnt[k];
    if (!live) return;
    int32_t *sub_rowptr = a.p.sub_rowptr[which], *sub_col = a.p.sub_col[which];
    if (lane == 0) {
        sub_rowptr[i] = w;
        if (i == a.n - 1) sub_rowptr[a.n] = w + cnt;
    }
    if (cnt <= kStash) {
        for (int k = lane; k < cnt; k += kWave)
            if ((int64_t)w + k < a.capacity) sub_col[w + k] = stash[wave][k];
    } else {
        int64_t wp = w;
        for (int base = beg; base < end; base += kWave) {
            const int e = base + lane;
            int u_, p_;
            const int r = new_id(e, e < end, u_, p_);
            const unsigned long long m = __ballot(r >= 0);
            if (r >= 0) {
                const int64_t pos = wp + __popcll(m & ((1ULL << lane) - 1ULL));
                if (pos < a.capacity) sub_col[pos] = r;
            }
            wp += __popcll(m);
        }
    }
}

__global__ __launch_bounds__(64 * kPartsWaves) __attribute__((amdgpu_waves_per_eu(8, 8))) void extract_parts_kernel(PartsArgs a) {
    extract_parts_block(a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// The optimiser and the NEXT batch's extraction in one grid (gist_adam_segments_extract_f32).  Once the backward pass
// is done nothing reads the batch buffers any more, and the next batch depends on nothing the step computes: its 3 gx
// extraction workgroups (dispatched first: their look-back chain is latency, 21 us for ~2000 rows) run beside the
// optimiser's (memory), instead of in front of the next step's first aggregation.  1024-thread workgroups: an
// optimiser workgroup runs four of adam_body.h's virtual blocks; the last workgroup reduces the loss.
__global__ __launch_bounds__(64 * kPartsWaves) __attribute__((amdgpu_waves_per_eu(8, 8))) void adam_extract_kernel(PartsArgs a, int gx, AdamArgs A) {
    __shared__ float red[4];
    const int n_ex = 3 * gx;
    if ((int)blockIdx.x < n_ex) {
        extract_parts_block(a, (int)blockIdx.x % gx, (int)blockIdx.x / gx, gx);
        return;
    }
    // the optimiser's workgroups in the order of their own latency: the dedicated blocks (chunk sums walked source by
    // source: the longest chains) and the loss first, so that they start with the extraction; the arena's blocks
    // (bandwidth) fill in behind them as slots come free
    int64_t w = (int64_t)blockIdx.x - n_ex;
    const int64_t n_arena = adam_arena_blocks(A.n);
    const int64_t ded_wg = (A.segs.n_ded + 3) / 4;
    const int quarter = threadIdx.x >> 8;
    if (w < ded_wg) {
        const int64_t d = 4 * w + quarter;
        if (d < A.segs.n_ded) adam_virtual_block(A, n_arena + d, threadIdx.x & 255);
        return;
    }
    w -= ded_wg;
    if (A.row_nll != nullptr) {
        if (w == 0) {
            loss_reduce_256(A.row_nll, A.n_loss_rows, A.inv_count, A.loss, red);
            return;
        }
        --w;
    }
    const int64_t vb = 4 * w + quarter;
    if (vb < n_arena) adam_virtual_block(A, vb, threadIdx.x & 255);
}

// dst[i, :] = src[ids[i], :]; one wave per row
template <int VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src,
                                                          int64_t lds_,
                                                          const int32_t *__restrict__ ids,
                                                          int n_ids, int d,
                                                          float *__restrict__ dst, int64_t ldd) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_ids) return;
    const int lane = threadIdx.x & 63;
    const float *s = src + (int64_t)ids[i] * lds_;
    float *o = dst + (int64_t)i * ldd;
    for (int c = lane * VEC; c < d; c += kWave * VEC) {
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(o + c) = *reinterpret_cast<const float4 *>(s + c);
        else if constexpr (VEC == 2) *reinterpret_cast<float2 *>(o + c) = *reinterpret_cast<const float2 *>(s + c);
        else o[c] = s[c];
    }
}

__global__ void gather_i32_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ ids,
                                  int64_t n, int32_t *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[ids[i]];
}

// ---- IST weight blocks ---------------------------------------------------------
// One workgroup row-strip: blockIdx.y = block row, threads sweep the columns, so the
// contiguous side (dst for gather, src for scatter) is accessed with full lines.
template <bool SCATTER>
__global__ __launch_bounds__(256) void block_move_kernel(const float *__restrict__ src,
                                                         int64_t lds_,
                                                         const int32_t *__restrict__ row_idx,
                                                         const int32_t *__restrict__ col_idx,
                                                         int n_rows, int n_cols,
                                                         float *__restrict__ dst, int64_t ldd) {
    for (int i = blockIdx.y; i < n_rows; i += gridDim.y) {
        const int ri = row_idx ? row_idx[i] : i;
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_cols;
             j += gridDim.x * blockDim.x) {
            const int cj = col_idx ? col_idx[j] : j;
            if constexpr (SCATTER) dst[(int64_t)ri * ldd + cj] = src[(int64_t)i * lds_ + j];
            else dst[(int64_t)i * ldd + j] = src[(int64_t)ri * lds_ + cj];
        }
    }
}

// Pairwise (tree) summation over the sources: for identical copies and a power-of-two
// count every partial sum is exact, so "dispatch -> sync without training" leaves the shared
// bias bit-identical (a sequential sum rounds at 3x).  Deterministic for any count.
__global__ void mean_rows_kernel(const float *__restrict__ src, int64_t stride, int n_src, int64_t n,
                                 float *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    constexpr int kMax = 64;
    float v[kMax];
    float tail = 0.f;
    const int m = n_src < kMax ? n_src : kMax;
    for (int k = 0; k < kMax; ++k) v[k] = k < m ? src[(int64_t)k * stride + j] : 0.f;
    for (int k = kMax; k < n_src; ++k) tail += src[(int64_t)k * stride + j];
#pragma unroll
    for (int w = 1; w < kMax; w <<= 1)
#pragma unroll
        for (int k = 0; k + w < kMax; k += 2 * w) v[k] += v[k + w];
    out[j] = (v[0] + tail) / (float)n_src;
}

}  // namespace gist

using namespace gist;

static int mark_impl(const char *name, const int32_t *ids, int64_t n_ids, int32_t *remap, int unmark,
                     gist_stream_t stream) {
    if (n_ids < 0) { set_error("%s: n_ids < 0", name); return GIST_EINVAL; }
    if (n_ids == 0) return GIST_OK;
    if (!ids || !remap) { set_error("%s: null pointer", name); return GIST_EINVAL; }
    hipLaunchKernelGGL(mark_kernel, dim3((unsigned)ceil_div(n_ids, 256)), dim3(256), 0,
                       as_stream(stream), ids, n_ids, remap, unmark);
    return launch_status(name);
}

extern "C" int gist_induced_mark(const int32_t *ids, int64_t n_ids, int32_t *remap,
                                 gist_stream_t stream) {
    return mark_impl("gist_induced_mark", ids, n_ids, remap, 0, stream);
}

extern "C" int gist_induced_unmark(const int32_t *ids, int64_t n_ids, int32_t *remap,
                                   gist_stream_t stream) {
    return mark_impl("gist_induced_unmark", ids, n_ids, remap, 1, stream);
}

extern "C" int gist_fill_i32(int32_t *p, int64_t n, int32_t value, gist_stream_t stream) {
    GIST_REQUIRE(n >= 0, "gist_fill_i32: n < 0");
    if (n == 0) return GIST_OK;
    GIST_REQUIRE(p, "gist_fill_i32: null pointer");
    const unsigned grid = (unsigned)(ceil_div(n, 256) < 4096 ? ceil_div(n, 256) : 4096);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid), dim3(256), 0, as_stream(stream), p, n, value);
    return launch_status("gist_fill_i32");
}

// dst[i] = src[i]: a copy issued as a KERNEL.  Either pointer may be pinned host memory (hipHostMalloc: device-
// accessible at the same address): the per-epoch tables travel this way (gist_hip.h, gist_copy_i32).
__global__ void copy_i32_kernel(const int32_t *__restrict__ src, int32_t *__restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

extern "C" int gist_copy_i32(const int32_t *src, int32_t *dst, int64_t n, gist_stream_t stream) {
    GIST_REQUIRE(n >= 0, "gist_copy_i32: n < 0");
    if (n == 0) return GIST_OK;
    GIST_REQUIRE(src && dst, "gist_copy_i32: null pointer");
    const unsigned grid = (unsigned)(ceil_div(n, 256) < 1024 ? ceil_div(n, 256) : 1024);
    hipLaunchKernelGGL(copy_i32_kernel, dim3(grid), dim3(256), 0, as_stream(stream), src, dst, n);
    return launch_status("gist_copy_i32");
}

// host_word[0] = *device_word (0 if NULL), host_word[1] = tag, visible to the host when the kernel has run
__global__ void publish_i64_kernel(const int64_t *__restrict__ device_word, int64_t tag, volatile int64_t *host_word) {
    host_word[0] = device_word ? *device_word : 0;
    __threadfence_system();
    host_word[1] = tag;
    __threadfence_system();
}

extern "C" int gist_publish_i64(const int64_t *device_word, int64_t tag, int64_t *host_word, gist_stream_t stream) {
    GIST_REQUIRE(host_word, "gist_publish_i64: null pointer");
    hipLaunchKernelGGL(publish_i64_kernel, dim3(1), dim3(1), 0, as_stream(stream), device_word, tag, host_word);
    return launch_status("gist_publish_i64");
}

extern "C" int gist_induced_rowptr(const int32_t *rowptr, const int32_t *col, const int32_t *ids,
                                   int64_t n_ids, const int32_t *remap, int32_t *sub_rowptr,
                                   gist_stream_t stream) {
    GIST_REQUIRE(n_ids >= 0 && n_ids < (1LL << 31) - 8, "gist_induced_rowptr: bad n_ids");
    GIST_REQUIRE(sub_rowptr, "gist_induced_rowptr: null sub_rowptr");
    hipStream_t st = as_stream(stream);
    if (n_ids > 0) {
        GIST_REQUIRE(rowptr && col && ids && remap, "gist_induced_rowptr: null pointer");
        hipLaunchKernelGGL(induced_count_kernel, dim3((unsigned)ceil_div(n_ids, 4)), dim3(256), 0, st,
                           rowptr, col, ids, (int)n_ids, remap, sub_rowptr);
    }
    hipLaunchKernelGGL(scan_rowptr_kernel, dim3(1), dim3(1024), 0, st, sub_rowptr, (int)n_ids);
    return launch_status("gist_induced_rowptr");
}

extern "C" int gist_induced_fill(const int32_t *rowptr, const int32_t *col, const int32_t *ids,
                                 int64_t n_ids, const int32_t *remap, const int32_t *sub_rowptr,
                                 int32_t *sub_col, int64_t sub_col_capacity, gist_stream_t stream) {
    GIST_REQUIRE(n_ids >= 0 && n_ids < (1LL << 31) - 8, "gist_induced_fill: bad n_ids");
    GIST_REQUIRE(sub_col_capacity >= 0, "gist_induced_fill: negative capacity");
    if (n_ids == 0) return GIST_OK;
    GIST_REQUIRE(rowptr && col && ids && remap && sub_rowptr, "gist_induced_fill: null pointer");
    GIST_REQUIRE(sub_col || sub_col_capacity == 0, "gist_induced_fill: null sub_col");
    hipLaunchKernelGGL(induced_fill_kernel, dim3((unsigned)ceil_div(n_ids, 4)), dim3(256), 0,
                       as_stream(stream), rowptr, col, ids, (int)n_ids, remap, sub_rowptr, sub_col,
                       sub_col_capacity);
    return launch_status("gist_induced_fill");
}

extern "C" int gist_gather_rows_f32(const float *src, int64_t lds_, const int32_t *ids,
                                    int64_t n_ids, int64_t d, float *dst, int64_t ldd,
                                    gist_stream_t stream) {
    GIST_REQUIRE(n_ids >= 0 && d >= 0, "gist_gather_rows_f32: negative size");
    if (n_ids == 0 || d == 0) return GIST_OK;
    GIST_REQUIRE(src && ids && dst, "gist_gather_rows_f32: null pointer");
    GIST_REQUIRE(lds_ >= d && ldd >= d, "gist_gather_rows_f32: leading dimension < d");
    GIST_REQUIRE(n_ids < (1LL << 31) - 8 && d < (1LL << 31), "gist_gather_rows_f32: size >= 2^31");
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)ceil_div(n_ids, 4));
    if (d % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0 && aligned16(src) && aligned16(dst))
        hipLaunchKernelGGL(gather_rows_kernel<4>, grid, dim3(256), 0, st, src, lds_, ids, (int)n_ids,
                           (int)d, dst, ldd);
    else if (d % 2 == 0 && lds_ % 2 == 0 && ldd % 2 == 0 && aligned8(src) && aligned8(dst))
        hipLaunchKernelGGL(gather_rows_kernel<2>, grid, dim3(256), 0, st, src, lds_, ids, (int)n_ids,
                           (int)d, dst, ldd);
    else
        hipLaunchKernelGGL(gather_rows_kernel<1>, grid, dim3(256), 0, st, src, lds_, ids, (int)n_ids,
                           (int)d, dst, ldd);
    return launch_status("gist_gather_rows_f32");
}

extern "C" int gist_gather_i32(const int32_t *src, const int32_t *ids, int64_t n_ids, int32_t *dst,
                               gist_stream_t stream) {
    GIST_REQUIRE(n_ids >= 0, "gist_gather_i32: n_ids < 0");
    if (n_ids == 0) return GIST_OK;
    GIST_REQUIRE(src && ids && dst, "gist_gather_i32: null pointer");
    hipLaunchKernelGGL(gather_i32_kernel, dim3((unsigned)ceil_div(n_ids, 256)), dim3(256), 0,
                       as_stream(stream), src, ids, n_ids, dst);
    return launch_status("gist_gather_i32");
}

static int block_move(const char *name, bool scatter, const float *src, int64_t lds_,
                      const int32_t *row_idx, const int32_t *col_idx, int64_t n_rows,
                      int64_t n_cols, float *dst, int64_t ldd, gist_stream_t stream) {
    if (n_rows < 0 || n_cols < 0) { set_error("%s: negative size", name); return GIST_EINVAL; }
    if (n_rows == 0 || n_cols == 0) return GIST_OK;
    if (!src || !dst) { set_error("%s: null pointer", name); return GIST_EINVAL; }
    if (n_rows >= (1LL << 31) || n_cols >= (1LL << 31)) { set_error("%s: size >= 2^31", name); return GIST_EINVAL; }
    const unsigned gx = (unsigned)(ceil_div(n_cols, 256) < 64 ? ceil_div(n_cols, 256) : 64);
    const unsigned gy = (unsigned)(n_rows < 8192 ? n_rows : 8192);
    if (scatter)
        hipLaunchKernelGGL(block_move_kernel<true>, dim3(gx, gy), dim3(256), 0, as_stream(stream), src,
                           lds_, row_idx, col_idx, (int)n_rows, (int)n_cols, dst, ldd);
    else
        hipLaunchKernelGGL(block_move_kernel<false>, dim3(gx, gy), dim3(256), 0, as_stream(stream), src,
                           lds_, row_idx, col_idx, (int)n_rows, (int)n_cols, dst, ldd);
    return launch_status(name);
}

extern "C" int gist_block_gather_f32(const float *src, int64_t lds_, const int32_t *row_idx,
                                     const int32_t *col_idx, int64_t n_rows, int64_t n_cols,
                                     float *dst, int64_t ldd, gist_stream_t stream) {
    return block_move("gist_block_gather_f32", false, src, lds_, row_idx, col_idx, n_rows, n_cols,
                      dst, ldd, stream);
}

extern "C" int gist_block_scatter_f32(const float *src, int64_t lds_, const int32_t *row_idx,
                                      const int32_t *col_idx, int64_t n_rows, int64_t n_cols,
                                      float *dst, int64_t ldd, gist_stream_t stream) {
    return block_move("gist_block_scatter_f32", true, src, lds_, row_idx, col_idx, n_rows, n_cols,
                      dst, ldd, stream);
}

extern "C" int gist_mean_rows_f32(const float *src, int64_t stride, int64_t n_src, int64_t n,
                                  float *out, gist_stream_t stream) {
    GIST_REQUIRE(n_src > 0 && n >= 0, "gist_mean_rows_f32: bad size");
    if (n == 0) return GIST_OK;
    GIST_REQUIRE(src && out, "gist_mean_rows_f32: null pointer");
    hipLaunchKernelGGL(mean_rows_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       as_stream(stream), src, stride, (int)n_src, n, out);
    return launch_status("gist_mean_rows_f32");
}

static int extract_impl(const int32_t *g_rowptr, const int32_t *g_col,
                                  const int32_t *g_t_rowptr, const int32_t *g_t_col,
                                  const int32_t *ids, int64_t n, int32_t *remap, int32_t *rowptr,
                                  int32_t *col, int32_t *t_rowptr, int32_t *t_col,
                                  int64_t col_capacity, float *norm, const float *feat,
                                  int64_t ld_feat, int64_t n_feat, float *z0, int64_t ldz0,
                                  const int32_t *labels_all, int32_t *labels, const GatherDrop *drop,
                                  gist_stream_t stream) {
    GIST_REQUIRE(n > 0 && n < (1LL << 31) - 8, "gist_extract_batch: bad n");
    GIST_REQUIRE(g_rowptr && g_col && g_t_rowptr && g_t_col && ids && remap && rowptr && col &&
                     t_rowptr && t_col && norm && feat && z0,
                 "gist_extract_batch: null pointer");
    GIST_REQUIRE(n_feat > 0 && ld_feat >= n_feat && ldz0 >= n_feat && n_feat < (1LL << 31),
                 "gist_extract_batch: bad feature shape");
    GIST_REQUIRE(col_capacity >= 0, "gist_extract_batch: negative capacity");
    hipStream_t st = as_stream(stream);
    CsrPair p;
    p.rowptr[0] = g_rowptr; p.col[0] = g_col; p.sub_rowptr[0] = rowptr; p.sub_col[0] = col;
    p.rowptr[1] = g_t_rowptr; p.col[1] = g_t_col; p.sub_rowptr[1] = t_rowptr; p.sub_col[1] = t_col;
    const unsigned nb4 = (unsigned)ceil_div(n, 4);
    hipLaunchKernelGGL(mark_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, ids, n, remap, 0);
    hipLaunchKernelGGL(induced_count2_kernel, dim3(nb4, 2), dim3(256), 0, st, p, ids, (int)n, remap, norm);
    hipLaunchKernelGGL(scan_rowptr2_kernel, dim3(2), dim3(1024), 0, st, rowptr, t_rowptr, (int)n);
    hipLaunchKernelGGL(induced_fill2_kernel, dim3(nb4, 2), dim3(256), 0, st, p, ids, (int)n, remap,
                       col_capacity);
    GatherDrop gd{};
    if (drop != nullptr) {
        gd = *drop;
        GIST_REQUIRE(gd.x0 != nullptr && gd.ldx0 >= n_feat && gd.mask_ld >= n_feat && gd.p >= 0.f && gd.p < 1.f,
                     "gist_extract_batch_drop: bad dropout arguments");
        if (n_feat % 2 == 0)
            hipLaunchKernelGGL((gather_batch_kernel<2, true>), dim3(nb4), dim3(256), 0, st, feat, ld_feat, ids,
                               (int)n, (int)n_feat, z0, ldz0, labels_all, labels, remap, gd);
        else
            hipLaunchKernelGGL((gather_batch_kernel<1, true>), dim3(nb4), dim3(256), 0, st, feat, ld_feat, ids,
                               (int)n, (int)n_feat, z0, ldz0, labels_all, labels, remap, gd);
        return launch_status("gist_extract_batch_drop");
    }
    if (n_feat % 4 == 0 && ld_feat % 4 == 0 && ldz0 % 4 == 0 && aligned16(feat) && aligned16(z0))
        hipLaunchKernelGGL(gather_batch_kernel<4>, dim3(nb4), dim3(256), 0, st, feat, ld_feat, ids,
                           (int)n, (int)n_feat, z0, ldz0, labels_all, labels, remap, gd);
    else if (n_feat % 2 == 0 && ld_feat % 2 == 0 && ldz0 % 2 == 0 && aligned8(feat) && aligned8(z0))
        hipLaunchKernelGGL(gather_batch_kernel<2>, dim3(nb4), dim3(256), 0, st, feat, ld_feat, ids,
                           (int)n, (int)n_feat, z0, ldz0, labels_all, labels, remap, gd);
    else
        hipLaunchKernelGGL(gather_batch_kernel<1>, dim3(nb4), dim3(256), 0, st, feat, ld_feat, ids,
                           (int)n, (int)n_feat, z0, ldz0, labels_all, labels, remap, gd);
    return launch_status("gist_extract_batch");
}

extern "C" int gist_extract_batch(const int32_t *g_rowptr, const int32_t *g_col,
                                  const int32_t *g_t_rowptr, const int32_t *g_t_col,
                                  const int32_t *ids, int64_t n, int32_t *remap, int32_t *rowptr,
                                  int32_t *col, int32_t *t_rowptr, int32_t *t_col,
                                  int64_t col_capacity, float *norm, const float *feat,
                                  int64_t ld_feat, int64_t n_feat, float *z0, int64_t ldz0,
                                  const int32_t *labels_all, int32_t *labels,
                                  gist_stream_t stream) {
    return extract_impl(g_rowptr, g_col, g_t_rowptr, g_t_col, ids, n, remap, rowptr, col, t_rowptr, t_col,
                        col_capacity, norm, feat, ld_feat, n_feat, z0, ldz0, labels_all, labels, nullptr,
                        stream);
}

extern "C" int gist_extract_batch_drop(const int32_t *g_rowptr, const int32_t *g_col,
                                       const int32_t *g_t_rowptr, const int32_t *g_t_col,
                                       const int32_t *ids, int64_t n, int32_t *remap, int32_t *rowptr,
                                       int32_t *col, int32_t *t_rowptr, int32_t *t_col,
                                       int64_t col_capacity, float *norm, const float *feat,
                                       int64_t ld_feat, int64_t n_feat, float *z0, int64_t ldz0,
                                       const int32_t *labels_all, int32_t *labels, float *x0, int64_t ldx0,
                                       float p, uint64_t seed, uint64_t offset, int64_t mask_ld,
                                       gist_stream_t stream) {
    GatherDrop gd{};
    gd.x0 = x0; gd.ldx0 = ldx0; gd.p = p; gd.scale = (p > 0.f && p < 1.f) ? 1.0f / (1.0f - p) : 1.f;
    gd.sm = seed * 0x9E3779B97F4A7C15ULL; gd.offset = offset; gd.mask_ld = mask_ld;
    return extract_impl(g_rowptr, g_col, g_t_rowptr, g_t_col, ids, n, remap, rowptr, col, t_rowptr, t_col,
                        col_capacity, norm, feat, ld_feat, n_feat, z0, ldz0, labels_all, labels, &gd, stream);
}


// ---- one-launch extraction for batches that are unions of parts (see extract_parts_kernel) ------------
extern "C" int64_t gist_extract_parts_scratch_bytes(int64_t n_max) {
    return n_max <= 0 ? 0 : 32 + 16 * ceil_div(n_max, kPartsWaves);
}

// 1 if gist_extract_parts_batch takes a buffer set sized for n_max rows
extern "C" int gist_extract_parts_supported(int64_t n_max) {
    return n_max > 0 && n_max < (1LL << 31) - 64 ? 1 : 0;
}

// checks a descriptor and turns it into the kernel's arguments (a fresh launch epoch each time)
static int parts_args(const char *name, const gist_extract_parts_desc &x, PartsArgs *out) {
    GIST_REQUIRE(x.n > 0 && x.n <= x.n_max && x.n_max < (1LL << 31) - 8, "%s: bad n", name);
    GIST_REQUIRE(x.g_rowptr && x.g_col && x.g_t_rowptr && x.g_t_col && x.ids && x.node_part && x.part_slot && x.rowptr &&
                     x.col && x.t_rowptr && x.t_col && x.norm && x.feat && x.z0 && x.scratch,
                 "%s: null pointer", name);
    GIST_REQUIRE(aligned8(x.node_part) && aligned8(x.part_slot), "%s: tables must be 8-byte aligned", name);
    GIST_REQUIRE(x.n_feat > 0 && x.ld_feat >= x.n_feat && x.ldz0 >= x.n_feat && x.n_feat < (1LL << 31),
                 "%s: bad feature shape", name);
    GIST_REQUIRE(x.col_capacity >= 0 && x.batch >= 0, "%s: bad capacity / batch index", name);
    GIST_REQUIRE(aligned8(x.scratch), "%s: scratch must be 8-byte aligned", name);
    GIST_REQUIRE(gist_extract_parts_supported(x.n_max) == 1, "%s: bad n_max", name);
    // launch epochs: process-wide, never 0 (a zeroed scratch matches no launch), never reused
    static std::atomic<unsigned long long> g_epoch{0};
    const unsigned long long epoch = (g_epoch.fetch_add(1, std::memory_order_relaxed) + 1) & ((1ULL << 33) - 1);
    GIST_REQUIRE(epoch != 0, "%s: launch counter exhausted", name);
    PartsArgs a{};
    a.p.rowptr[0] = x.g_rowptr; a.p.col[0] = x.g_col; a.p.sub_rowptr[0] = x.rowptr; a.p.sub_col[0] = x.col;
    a.p.rowptr[1] = x.g_t_rowptr; a.p.col[1] = x.g_t_col; a.p.sub_rowptr[1] = x.t_rowptr; a.p.sub_col[1] = x.t_col;
    a.ids = x.ids; a.n = (int)x.n; a.n_max = (int)x.n_max;
    a.node_part = x.node_part; a.part_slot = x.part_slot;
    a.batch = x.batch; a.capacity = x.col_capacity; a.norm = x.norm;
    a.error = static_cast<unsigned long long *>(x.scratch) + 1;
    a.tickets = static_cast<unsigned long long *>(x.scratch) + 2;
    a.slots = static_cast<unsigned long long *>(x.scratch) + 4;
    a.epoch = epoch;
    a.feat = x.feat; a.ld_feat = x.ld_feat; a.d = (int)x.n_feat; a.z0 = x.z0; a.ldz0 = x.ldz0;
    a.labels_all = x.labels_all; a.labels = x.labels;
    a.drop = 0;
    if (x.x0 != nullptr) {
        GIST_REQUIRE(x.ldx0 >= x.n_feat && x.mask_ld >= x.n_feat && x.p >= 0.f && x.p < 1.f, "%s: bad dropout arguments", name);
        a.drop = 1;
        a.gd.x0 = x.x0; a.gd.ldx0 = x.ldx0; a.gd.p = x.p; a.gd.scale = x.p > 0.f ? 1.0f / (1.0f - x.p) : 1.f;
        a.gd.sm = x.seed * 0x9E3779B97F4A7C15ULL; a.gd.offset = x.offset; a.gd.mask_ld = x.mask_ld;
    }
    if (x.ah != nullptr) {
        GIST_REQUIRE(x.feat_intra != nullptr && x.ld_intra >= x.n_feat, "%s: bad feat_intra", name);
        GIST_REQUIRE(x.x0 == nullptr || x.mask_ld >= 2 * x.n_feat, "%s: mask pitch below 2 n_feat with ah", name);
        a.feat_intra = x.feat_intra; a.ld_intra = x.ld_intra; a.ah = x.ah;
        a.gd.offset = x.offset; a.gd.mask_ld = x.mask_ld;
    }
    *out = a;
    return GIST_OK;
}

extern "C" int gist_extract_parts_desc_batch(const gist_extract_parts_desc *desc, gist_stream_t stream) {
    GIST_REQUIRE(desc != nullptr, "gist_extract_parts_desc_batch: null descriptor");
    PartsArgs a;
    const int rc = parts_args("gist_extract_parts_desc_batch", *desc, &a);
    if (rc != GIST_OK) return rc;
    hipLaunchKernelGGL(extract_parts_kernel, dim3((unsigned)ceil_div(desc->n, kPartsWaves), 3),
                       dim3(64 * kPartsWaves), 0, as_stream(stream), a);
    return launch_status("gist_extract_parts_desc_batch");
}

extern "C" int gist_extract_parts_batch(const int32_t *g_rowptr, const int32_t *g_col,
                                        const int32_t *g_t_rowptr, const int32_t *g_t_col,
                                        const int32_t *ids, int64_t n, int64_t n_max,
                                        const int32_t *node_part, const int32_t *part_slot, int32_t batch,
                                        int32_t *rowptr, int32_t *col, int32_t *t_rowptr, int32_t *t_col,
                                        int64_t col_capacity, float *norm, const float *feat, int64_t ld_feat,
                                        int64_t n_feat, float *z0, int64_t ldz0, const int32_t *labels_all,
                                        int32_t *labels, float *x0, int64_t ldx0, float p, uint64_t seed,
                                        uint64_t offset, int64_t mask_ld, void *scratch, gist_stream_t stream) {
    gist_extract_parts_desc x{};
    x.g_rowptr = g_rowptr; x.g_col = g_col; x.g_t_rowptr = g_t_rowptr; x.g_t_col = g_t_col;
    x.ids = ids; x.n = n; x.n_max = n_max; x.node_part = node_part; x.part_slot = part_slot; x.batch = batch;
    x.rowptr = rowptr; x.col = col; x.t_rowptr = t_rowptr; x.t_col = t_col; x.col_capacity = col_capacity; x.norm = norm;
    x.feat = feat; x.ld_feat = ld_feat; x.n_feat = n_feat; x.z0 = z0; x.ldz0 = ldz0;
    x.labels_all = labels_all; x.labels = labels;
    x.x0 = x0; x.ldx0 = ldx0; x.p = p; x.seed = seed; x.offset = offset; x.mask_ld = mask_ld; x.scratch = scratch;
    PartsArgs a;
    const int rc = parts_args("gist_extract_parts_batch", x, &a);
    if (rc != GIST_OK) return rc;
    hipLaunchKernelGGL(extract_parts_kernel, dim3((unsigned)ceil_div(n, kPartsWaves), 3),
                       dim3(64 * kPartsWaves), 0, as_stream(stream), a);
    return launch_status("gist_extract_parts_batch");
}

extern "C" int gist_adam_segments_extract_f32(float *param, float *grad, float *exp_avg, float *exp_avg_sq,
                                              int64_t n, float lr, float beta1, float beta2, float eps,
                                              float weight_decay, int64_t step, const gist_grad_segment *segments,
                                              int64_t n_segments, const float *row_loss, int64_t n_loss_rows,
                                              int64_t loss_count, float *loss, const gist_extract_parts_desc *next,
                                              gist_stream_t stream) {
    GIST_REQUIRE(next != nullptr, "gist_adam_segments_extract_f32: null descriptor");
    gist::AdamArgs A;
    int rc = gist::adam_segments_args("gist_adam_segments_extract_f32", param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                                      beta2, eps, weight_decay, step, segments, n_segments, row_loss, n_loss_rows,
                                      loss_count, loss, &A);
    if (rc != GIST_OK) return rc;
    PartsArgs a;
    rc = parts_args("gist_adam_segments_extract_f32", *next, &a);
    if (rc != GIST_OK) return rc;
    const int64_t gx = ceil_div(next->n, kPartsWaves);
    const int64_t blocks = 3 * gx + ceil_div((int64_t)A.segs.n_ded, 4) + (row_loss ? 1 : 0) +
                           ceil_div(gist::adam_arena_blocks(n), 4);
    GIST_REQUIRE(blocks < (1LL << 31), "gist_adam_segments_extract_f32: grid too large");
    hipLaunchKernelGGL(adam_extract_kernel, dim3((unsigned)blocks), dim3(64 * kPartsWaves), 0, as_stream(stream), a, (int)gx, A);
    return launch_status("gist_adam_segments_extract_f32");
}
