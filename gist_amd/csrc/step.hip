// Native step driver: one C-ABI call issues the whole training iteration (batch
// extraction -> forward -> CE -> backward -> Adam) on one stream.  The host-side
// sequencing of the reference's loop body (cluster_gcn_ist_distrib.py:408-417) moves
// from ~45 interpreter round trips to ~45 back-to-back hipLaunch calls, which is what
// bounds the small-width (many-GPU) regime.  Arithmetic is unchanged: it calls the same
// entry points the op-level API exposes.
#include <math.h>

#include <new>

#include "common.h"
#include "class_dw_body.h"

using namespace gist;

#define GIST_TRY(expr)            \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != GIST_OK) return rc_; \
    } while (0)

static inline uint64_t round_up2(uint64_t x) { return x + (x & 1ULL); }

// ---- HIP-event timer ------------------------------------------------------------------
struct gist_timer {
    int64_t capacity, count;
    hipEvent_t *start, *stop;
    int32_t *kind;
    int64_t *m, *n, *k;
};

extern "C" gist_timer *gist_timer_create(int64_t capacity) {
    if (capacity <= 0) return nullptr;
    gist_timer *t = new (std::nothrow) gist_timer();
    if (!t) return nullptr;
    t->capacity = capacity;
    t->count = 0;
    t->start = new hipEvent_t[capacity];
    t->stop = new hipEvent_t[capacity];
    t->kind = new int32_t[capacity];
    t->m = new int64_t[capacity];
    t->n = new int64_t[capacity];
    t->k = new int64_t[capacity];
    for (int64_t i = 0; i < capacity; ++i) {
        if (hipEventCreate(&t->start[i]) != hipSuccess || hipEventCreate(&t->stop[i]) != hipSuccess) {
            set_error("gist_timer_create: hipEventCreate failed");
            t->capacity = i;
            break;
        }
    }
    return t;
}

extern "C" void gist_timer_destroy(gist_timer *t) {
    if (!t) return;
    for (int64_t i = 0; i < t->capacity; ++i) {
        (void)hipEventDestroy(t->start[i]);
        (void)hipEventDestroy(t->stop[i]);
    }
    delete[] t->start; delete[] t->stop; delete[] t->kind;
    delete[] t->m; delete[] t->n; delete[] t->k;
    delete t;
}

extern "C" void gist_timer_reset(gist_timer *t) { if (t) t->count = 0; }
extern "C" int64_t gist_timer_count(const gist_timer *t) { return t ? t->count : 0; }

extern "C" int gist_timer_read(gist_timer *t, int64_t i, float *ms, int32_t *kind, int64_t *m,
                               int64_t *n, int64_t *k) {
    GIST_REQUIRE(t && i >= 0 && i < t->count && ms, "gist_timer_read: bad index");
    hipError_t e = hipEventElapsedTime(ms, t->start[i], t->stop[i]);
    if (e != hipSuccess) { set_error("gist_timer_read: %s", hipGetErrorString(e)); return GIST_ELAUNCH; }
    if (kind) *kind = t->kind[i];
    if (m) *m = t->m[i];
    if (n) *n = t->n[i];
    if (k) *k = t->k[i];
    return GIST_OK;
}

namespace gist {
thread_local gist_timer *tl_timer = nullptr;

int64_t timer_begin(gist_timer *t, int kind, int64_t m, int64_t n, int64_t k, hipStream_t s) {
    if (!t || t->count >= t->capacity) return -1;
    const int64_t slot = t->count++;
    t->kind[slot] = kind; t->m[slot] = m; t->n[slot] = n; t->k[slot] = k;
    (void)hipEventRecord(t->start[slot], s);
    return slot;
}

void timer_end(gist_timer *t, int64_t slot, hipStream_t s) {
    if (t && slot >= 0) (void)hipEventRecord(t->stop[slot], s);
}
}  // namespace gist

namespace {
struct Scope {   // records start now, stop at scope exit
    gist_timer *t; int64_t slot; hipStream_t s;
    Scope(gist_timer *t_, int kind, int64_t m, int64_t n, int64_t k, hipStream_t s_)
        : t(t_), slot(timer_begin(t_, kind, m, n, k, s_)), s(s_) {}
    ~Scope() { timer_end(t, slot, s); }
};
struct PairsHint {     // the aggregations below look for sibling blocks only if the plan says the batch may have them
    bool prev;
    explicit PairsHint(bool on) : prev(tl_spmm_pairs) { tl_spmm_pairs = on; }
    ~PairsHint() { tl_spmm_pairs = prev; }
};
struct ActiveTimer {   // kernels below the entry points see the armed timer for this call only
    explicit ActiveTimer(gist_timer *t) { tl_timer = t; }
    ~ActiveTimer() { tl_timer = nullptr; }
};
}  // namespace

// ---- split projection operands kept by the step (gist_step_plan.h3_workspace) ---------------
namespace {
struct H3Layer {
    bool on;                       // this layer's three projections run on pre-split operands
    int shift;                     // fixed scale exponent of its input activations Z_k
    uint32_t *Zs, *ZsT, *Ws, *WsT; // [n][kpad(2in)], [2in][kpad(n)], [out][kpad(2in)], [2in][kpad(out)]
    float *inv_zr, *inv_zt, *inv_wr, *inv_wt;
};
struct H3Step {
    bool any;
    H3Layer layer[GIST_MAX_LAYERS];
    uint32_t *dYs, *dYsT;          // [n][kpad(out)], [out][kpad(n)] (shared by the layers)
    float *inv_dyr, *inv_dyt, *rowmax, *colmax, *pmax;
    unsigned *amax;                // [GIST_MAX_LAYERS]
    int64_t bytes;
};

inline int bound_shift(double bound) {      // 2^shift * bound <= 2^13
    int e = 0;
    (void)frexp(bound, &e);                  // bound = f * 2^e, f in [0.5, 1)
    const int s = 13 - e;
    return s < -60 ? -60 : (s > 60 ? 60 : s);
}

// Deterministic carve-up of the workspace from the plan's shapes; base may be NULL (sizing).
H3Step h3_layout(const gist_step_plan *p, char *base) {
    H3Step h{};
    const int L1 = p->n_layers;
    const int64_t n = p->n_max;
    if (gist_gemm_get_mode() != 1 || n <= 0) return h;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        char *q = base ? base + off : nullptr;
        off += ceil_div(bytes, 256) * 256;
        return q;
    };
    int64_t max_out = 0;
    const double keep = p->p_drop > 0.f ? 1.0 / (1.0 - (double)p->p_drop) : 1.0;
    for (int k = 0; k < L1; ++k) {
        const gist_layer_desc &l = p->layer[k];
        const int64_t i2 = 2 * l.n_in, o = l.n_out;
        H3Layer &hl = h.layer[k];
        const double bound = k == 0 ? (double)p->feat_absmax * keep
                                    : (p->use_layernorm ? sqrt((double)(l.n_in > 1 ? l.n_in - 1 : 1)) * keep : 0.0);
        // the class layer (k == L1-1) stays on the per-call path: its dY = dlogits comes from the CE
        // kernel, which writes no row maxima for the gradient split
        hl.on = k < L1 - 1 && bound > 0.0 && h3_eligible_kept(n, o, i2) && h3_eligible_kept(o, i2, n) &&
                (k == 0 || h3_eligible_kept(n, i2, o)) && l.ldz % 4 == 0 && l.ldy % 4 == 0 &&
                aligned16(l.W) && aligned16(l.Z) && aligned16(l.Y) && aligned16(l.dW);
        if (!hl.on) continue;
        h.any = true;
        hl.shift = bound_shift(bound);
        hl.Zs = reinterpret_cast<uint32_t *>(take(n * h3_kpad(i2) * 4));
        hl.ZsT = reinterpret_cast<uint32_t *>(take(i2 * h3_kpad(n) * 4));
        hl.Ws = reinterpret_cast<uint32_t *>(take(o * h3_kpad(i2) * 4));
        hl.WsT = k > 0 ? reinterpret_cast<uint32_t *>(take(i2 * h3_kpad(o) * 4)) : nullptr;
        hl.inv_zr = reinterpret_cast<float *>(take(n * 4));
        hl.inv_zt = reinterpret_cast<float *>(take(i2 * 4));
        hl.inv_wr = reinterpret_cast<float *>(take(o * 4));
        hl.inv_wt = k > 0 ? reinterpret_cast<float *>(take(i2 * 4)) : nullptr;
        max_out = o > max_out ? o : max_out;
    }
    if (!h.any) return h;
    h.dYs = reinterpret_cast<uint32_t *>(take(n * h3_kpad(max_out) * 4));
    h.dYsT = reinterpret_cast<uint32_t *>(take(max_out * h3_kpad(n) * 4));
    h.inv_dyr = reinterpret_cast<float *>(take(n * 4));
    h.inv_dyt = reinterpret_cast<float *>(take(max_out * 4));
    h.rowmax = reinterpret_cast<float *>(take(n * 4));
    h.colmax = reinterpret_cast<float *>(take(max_out * 4));
    h.pmax = reinterpret_cast<float *>(take(gist_colsum_partials(n) * max_out * 4));
    h.amax = reinterpret_cast<unsigned *>(take(GIST_MAX_LAYERS * 4));
    h.bytes = off;
    return h;
}

// bf16x3 mode: the same idea without scales -- three bf16 pieces per element, 6 bytes each
struct B3Layer {
    bool on;
    uint16_t *Zs, *ZsT, *Ws, *WsT; // [n][kpad(2in)], [2in][kpad(n)], [out][kpad(2in)], [2in][kpad(out)]
};
struct B3Step {
    bool any;
    B3Layer layer[GIST_MAX_LAYERS];
    uint16_t *dYs, *dYsT;          // [n][kpad(out)], [out][kpad(n)] (shared by the layers)
    float *slabs; int64_t slab_bytes;     // split-K slabs of the layers' projections (shared)
    int64_t bytes;
};

B3Step b3_layout(const gist_step_plan *p, char *base) {
    B3Step h{};
    const int L1 = p->n_layers;
    const int64_t n = p->n_max;
    if (gist_gemm_get_mode() != 2 || n <= 0) return h;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        char *q = base ? base + off : nullptr;
        off += ceil_div(bytes, 256) * 256;
        return reinterpret_cast<uint16_t *>(q);
    };
    int64_t max_out = 0;
    for (int k = 0; k < L1; ++k) {
        const gist_layer_desc &l = p->layer[k];
        const int64_t i2 = 2 * l.n_in, o = l.n_out;
        B3Layer &hl = h.layer[k];
        hl.on = b3_eligible_kept(n, o, i2) && b3_eligible_kept(o, i2, n) &&
                (k == 0 || b3_eligible_kept(n, i2, o)) &&
                l.ldy < (1LL << 21) && i2 < (1LL << 21);      // 32-bit byte offsets of a 256-row C tile
        if (!hl.on) continue;
        h.any = true;
        hl.Zs = take(n * b3_kpad(i2) * 6);
        hl.ZsT = take(i2 * b3_kpad(n) * 6);
        hl.Ws = take(o * b3_kpad(i2) * 6);
        hl.WsT = k > 0 ? take(i2 * b3_kpad(o) * 6) : nullptr;
        max_out = o > max_out ? o : max_out;
        // split-K slabs: the slice count depends on the batch rows through the tile count, and a batch
        // may have fewer rows than n_max (one 256-row tile less can double the slices): the largest
        // need over every row count up to n_max (the launcher uses one slice if the slab is too small)
        // (sampled at every row count that changes a tile count AND at every row count that changes the
        // number of 64-row k pairs of the dW projection: the slice count is not monotone in either)
        int64_t sb = 0;
        for (int64_t rows = n; rows > 0; rows = (rows - 1) / 64 * 64) {
            const int64_t need[3] = {b3_slab_bytes(rows, o, i2), b3_slab_bytes(o, i2, rows),
                                     k > 0 ? b3_slab_bytes(rows, i2, o) : 0};
            for (int q = 0; q < 3; ++q) sb = need[q] > sb ? need[q] : sb;
        }
        if (sb > h.slab_bytes) h.slab_bytes = sb;
    }
    if (!h.any) return h;
    h.dYs = take(n * b3_kpad(max_out) * 6);
    h.dYsT = take(max_out * b3_kpad(n) * 6);
    h.slabs = h.slab_bytes > 0 ? reinterpret_cast<float *>(take(h.slab_bytes)) : nullptr;
    h.bytes = off;
    return h;
}
}  // namespace

// ---- fused sequence: slabs of the deferred split-K projections, bias-gradient chunk sums -------------
namespace {
constexpr int64_t kLnbMaxUnits = 256;
struct FusedLayout {
    float *dw_slabs[GIST_MAX_LAYERS]; int64_t dw_bytes[GIST_MAX_LAYERS];   // dW_k = dY_k^T . Z_k
    float *logit_slabs; int64_t logit_bytes;                               // class layer's Y = Z . W^T
    float *y_slabs; int64_t y_bytes;      // a hidden layer's Y = Z . W^T until its LayerNorm has read it (one buffer)
    float *partials[GIST_MAX_LAYERS];                                      // [partial_rows[k] >= row_chunks16(n_max)][n_out_k]
    int64_t partial_rows[GIST_MAX_LAYERS];
    int64_t bytes, partial_floats;
};

// the largest slab need over the batch sizes a plan sees (the split count depends on the reduction
// length); a batch that would need more falls back to one k slice inside the launcher
int64_t slab_need(int64_t m, int64_t n, int64_t k_max, bool k_is_rows) {
    // (called for every layer of every step: remembered per shape and GEMM mode)
    struct Memo { int64_t m, n, k; int rows, mode; int64_t need; };
    static thread_local Memo memo[32];
    static thread_local int memo_n = 0;
    const int mode = gist_gemm_get_mode();
    // (tuning overrides change tile and slice choices: neither read nor fill the memo while one is set)
    const bool tuned = tune(GIST_TUNE_GEMM_TILE) != 0.0 || tune(GIST_TUNE_GEMM_SPLITS) != 0.0 || tune(GIST_TUNE_B3C) != 0.0 ||
                       tune(GIST_TUNE_B3C_SPLITS) != 0.0;
    for (int i = 0; i < memo_n && !tuned; ++i)
        if (memo[i].m == m && memo[i].n == n && memo[i].k == k_max && memo[i].rows == (int)k_is_rows && memo[i].mode == mode)
            return memo[i].need;
    int64_t need = 0;
    // every 32 rows down to half the largest batch (the model's slice count is not monotone in the reduction
    // length: two neighbouring batch sizes can differ by a factor of two)
    for (int64_t rows = k_max; rows > 0 && rows >= k_max / 2; rows -= 32) {
        const int64_t b = k_is_rows ? gemm_f32_slab_bytes(m, n, rows, true) : gemm_f32_slab_bytes(rows, n, m);
        need = b > need ? b : need;
    }
    if (!tuned) {      // a full table overwrites its oldest entry instead of recomputing every call
        static thread_local int memo_next = 0;
        if (memo_n < 32) memo[memo_n++] = Memo{m, n, k_max, (int)k_is_rows, mode, need};
        else { memo[memo_next] = Memo{m, n, k_max, (int)k_is_rows, mode, need}; memo_next = (memo_next + 1) % 32; }
    }
    return need;
}

FusedLayout fused_layout(const gist_step_plan *p, char *base, float *partials) {
    FusedLayout f{};
    const int L1 = p->n_layers;
    int64_t off = 0, poff = 0;
    auto take = [&](int64_t bytes) {
        char *q = base ? base + off : nullptr;
        off += ceil_div(bytes, 256) * 256;
        return reinterpret_cast<float *>(q);
    };
    const int64_t chunks = gist_row_chunks16(p->n_max);
    for (int k = 0; k < L1; ++k) {
        const gist_layer_desc &l = p->layer[k];
        f.dw_bytes[k] = slab_need(l.n_out, 2 * l.n_in, p->n_max, true);
        if (k == L1 - 1 && gist_class_layer_takes(p->n_max, l.n_out, 2 * l.n_in, 2 * l.n_in, 2 * l.n_in, nullptr, nullptr)) {
            // the fused class layer leaves dW as one slab per 128 rows (gist_class_dw_slabs_f32)
            const int64_t b = gist_class_dw_slab_bytes(p->n_max, l.n_out, 2 * l.n_in);
            f.dw_bytes[k] = b > f.dw_bytes[k] ? b : f.dw_bytes[k];
        }
        f.dw_slabs[k] = f.dw_bytes[k] > 0 ? take(f.dw_bytes[k]) : nullptr;
        f.partials[k] = partials ? partials + poff : nullptr;
        // (a hidden layer of <= 256 columns may get its chunk sums from the reverse aggregation above it: one row per
        // workgroup of that launch, at most kLnbMaxUnits of them)
        const int64_t rows = (k + 1 < L1 && l.n_out <= 256 && chunks < kLnbMaxUnits) ? kLnbMaxUnits : chunks;
        f.partial_rows[k] = rows;
        poff += ceil_div(rows * l.n_out, 64) * 64;
    }
    const gist_layer_desc &last = p->layer[L1 - 1];
    f.logit_bytes = slab_need(2 * last.n_in, last.n_out, p->n_max, false);
    f.logit_slabs = f.logit_bytes > 0 ? take(f.logit_bytes) : nullptr;
    f.y_bytes = 0;
    for (int k = 0; k + 1 < L1; ++k) {
        const int64_t b = slab_need(2 * p->layer[k].n_in, p->layer[k].n_out, p->n_max, false);
        f.y_bytes = b > f.y_bytes ? b : f.y_bytes;
    }
    f.y_slabs = f.y_bytes > 0 ? take(f.y_bytes) : nullptr;
    f.bytes = off;
    f.partial_floats = poff;
    return f;
}
}  // namespace

extern "C" int64_t gist_step_fused_workspace_bytes(const gist_step_plan *plan) {
    if (!plan || plan->n_layers < 1 || plan->n_layers > GIST_MAX_LAYERS || plan->n_max <= 0) return 0;
    return fused_layout(plan, nullptr, nullptr).bytes;
}
// bytes reserved for the slabs of layer k's weight gradient (k == n_layers: the class layer's logits)
extern "C" int64_t gist_step_fused_slab_bytes(const gist_step_plan *plan, int32_t k) {
    if (!plan || plan->n_layers < 1 || plan->n_layers > GIST_MAX_LAYERS || plan->n_max <= 0) return 0;
    if (k < 0 || k > plan->n_layers + 1) return 0;
    const FusedLayout f = fused_layout(plan, nullptr, nullptr);
    if (k == plan->n_layers + 1) return f.y_bytes;      // the hidden layers' forward projections (shared)
    return k == plan->n_layers ? f.logit_bytes : f.dw_bytes[k];
}
extern "C" int64_t gist_step_col_partials_floats(const gist_step_plan *plan) {
    if (!plan || plan->n_layers < 1 || plan->n_layers > GIST_MAX_LAYERS || plan->n_max <= 0) return 0;
    return fused_layout(plan, nullptr, nullptr).partial_floats;
}

extern "C" int64_t gist_step_h3_workspace_bytes(const gist_step_plan *plan) {
    if (!plan || plan->n_layers < 1 || plan->n_layers > GIST_MAX_LAYERS) return 0;
    if (gist_gemm_get_mode() == 2) return b3_layout(plan, nullptr).bytes;
    return h3_layout(plan, nullptr).bytes;
}

// the same for a given GEMM mode, without touching the process-wide one (a caller that sizes for every
// mode it may switch to must not change the arithmetic of launches other threads issue meanwhile)
extern "C" int64_t gist_step_h3_workspace_bytes_mode(const gist_step_plan *plan, int mode) {
    if (mode < 0 || mode > 2) return 0;
    h3_mode_override(mode);
    const int64_t need = gist_step_h3_workspace_bytes(plan);
    h3_mode_override(-1);
    return need;
}

// The step's aggregations: the blocked kernels when the batch comes with its locality blocks; with the
// batch's prepared block structure (`prepared`: this orientation's) the matrix-core kernel skips its set-up.
static int step_spmm(const gist_step_plan *p, const int32_t *rowptr, const int32_t *col, const float *x,
                     int64_t ldx, float *y, int64_t ldy, int64_t n, int64_t d, const float *out_scale,
                     const float *src_scale, int accumulate, const void *prepared, gist_stream_t s) {
    if (p->row_blocks != nullptr && p->n_row_blocks > 0) {
        if (prepared != nullptr)
            return gist_spmm_csr_prepared_f32(rowptr, col, x, ldx, y, ldy, n, d, out_scale, src_scale,
                                              accumulate, p->row_blocks, p->n_row_blocks, prepared, s);
        return gist_spmm_csr_blocked_f32(rowptr, col, x, ldx, y, ldy, n, d, out_scale, src_scale,
                                         accumulate, p->row_blocks, p->n_row_blocks, s);
    }
    return gist_spmm_csr_f32(rowptr, col, x, ldx, y, ldy, n, d, out_scale, src_scale, accumulate, s);
}

constexpr int64_t kPrefetchMaxParams = 12LL << 20;
// can the next batch (plan->next_*) be extracted by gist_extract_parts_batch's kernel?
static bool next_parts_ok(const gist_step_plan *p, bool fuse) {
    return fuse && p->node_part && p->part_slot && p->extract_scratch && p->next_ids && p->next_batch_index >= 0 &&
           p->next_n > 0 && p->next_n <= p->n_max && gist_extract_parts_supported(p->n_max) == 1;
}

extern "C" int gist_sage_step_extracts_next(const gist_step_plan *p, int64_t n, int flags) {
    if (p == nullptr || !(flags & GIST_STEP_TRAIN) || n <= 0) return 0;
    const bool fuse = p->fuse != 0 && n <= p->n_max;
    if (!fuse || p->col_partials == nullptr || !aligned16(p->col_partials)) return 0;
    const FusedLayout fl = fused_layout(p, static_cast<char *>(p->fused_workspace), p->col_partials);
    const bool defer = p->fused_workspace == nullptr ? fl.bytes == 0
                                                     : (aligned16(p->fused_workspace) && fl.bytes <= p->fused_workspace_bytes);
    // (beside a LARGE optimiser pass the extraction's 1024-thread workgroups cost more than they hide: each holds half a
    // CU's wave slots for the ~20 us of its look-back chain -- 233 against 199 + 21 us at 38.8 M parameters, 32 against
    // 16 + 21 at 1.2 M.  With the aggregating extraction's loads restructured the break-even is at ~11 M: H = 2048,
    // 11.0 M parameters, 0.9788 / 0.9822 ms per step against 0.9807 / 0.9836 without; H = 4096, 38.8 M: +70 us)
    if (p->n_params > kPrefetchMaxParams) return 0;
    return defer && next_parts_ok(p, fuse) ? 1 : 0;
}

extern "C" int gist_sage_step(const gist_step_plan *p, const int32_t *ids, int64_t n,
                              uint64_t drop_offset, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int64_t adam_step, int flags,
                              gist_stream_t s) {
    GIST_REQUIRE(p != nullptr, "gist_sage_step: null plan");
    GIST_REQUIRE(p->n_layers >= 1 && p->n_layers <= GIST_MAX_LAYERS, "gist_sage_step: bad n_layers");
    GIST_REQUIRE(n > 0, "gist_sage_step: empty batch");
    GIST_REQUIRE(!((flags & GIST_STEP_EXTRACT) && (flags & GIST_STEP_PREEXTRACTED)),
                 "gist_sage_step: GIST_STEP_EXTRACT and GIST_STEP_PREEXTRACTED exclude each other");
    GIST_REQUIRE(!(flags & (GIST_STEP_EXTRACT_NEXT | GIST_STEP_PREEXTRACTED)) || (flags & GIST_STEP_TRAIN),
                 "gist_sage_step: GIST_STEP_EXTRACT_NEXT / GIST_STEP_PREEXTRACTED belong to training steps");
    const int L1 = p->n_layers;
    hipStream_t st = as_stream(s);
    ActiveTimer active(p->timer);
    PairsHint pairs_hint(p->sibling_parts != 0);
    const bool train = (flags & GIST_STEP_TRAIN) != 0;
    const bool drop = train && p->p_drop > 0.f;
    // Phases (gist_hip.h, GIST_STEP_PHASE_*): a caller whose loop is `pred = model(g); loss = f(pred); loss.backward();
    // optimizer.step()` issues the same iteration as three calls.  Every decision below is a function of (plan, n,
    // drop_offset, flags without the phase bits), so the three calls agree on it.
    const int phase_bits = flags & (GIST_STEP_PHASE_FORWARD | GIST_STEP_PHASE_BACKWARD | GIST_STEP_PHASE_OPTIMIZER);
    GIST_REQUIRE(phase_bits == 0 || train, "gist_sage_step: GIST_STEP_PHASE_* belong to training steps");
    GIST_REQUIRE(!(flags & GIST_STEP_DLOGITS_GIVEN) || phase_bits == GIST_STEP_PHASE_BACKWARD,
                 "gist_sage_step: GIST_STEP_DLOGITS_GIVEN belongs to a GIST_STEP_PHASE_BACKWARD call");
    const bool do_fwd = phase_bits == 0 || (phase_bits & GIST_STEP_PHASE_FORWARD);
    const bool do_bwd = train && (phase_bits == 0 || (phase_bits & GIST_STEP_PHASE_BACKWARD));
    const bool do_opt = train && (phase_bits == 0 || (phase_bits & GIST_STEP_PHASE_OPTIMIZER));
    const bool split_phases = phase_bits != 0 && phase_bits != (GIST_STEP_PHASE_FORWARD | GIST_STEP_PHASE_BACKWARD |
                                                                GIST_STEP_PHASE_OPTIMIZER);
    GIST_REQUIRE(!split_phases || ((do_fwd ? 1 : 0) + (do_bwd ? 1 : 0) + (do_opt ? 1 : 0)) == 1,
                 "gist_sage_step: one GIST_STEP_PHASE_* per call (or none / all three)");
    const bool dlogits_given = (flags & GIST_STEP_DLOGITS_GIVEN) != 0;
    const bool blocked = p->row_blocks != nullptr && p->n_row_blocks > 0;

    // Split operands kept by the step (see gist_step_plan.h3_workspace); off = per-call splits
    // inside gist_gemm_*.
    H3Step h3{};
    if (p->h3_workspace != nullptr && aligned16(p->h3_workspace) && n <= p->n_max) {
        h3 = h3_layout(p, static_cast<char *>(p->h3_workspace));
        if (h3.bytes > p->h3_workspace_bytes) h3 = H3Step{};
    }
    B3Step b3{};
    if (p->h3_workspace != nullptr && aligned16(p->h3_workspace) && n <= p->n_max) {
        b3 = b3_layout(p, static_cast<char *>(p->h3_workspace));
        if (b3.bytes > p->h3_workspace_bytes) b3 = B3Step{};
    }

    // ---- the fused sequence (gist_step_plan.fuse): what each layer folds ----------------------------
    const bool fuse = p->fuse != 0 && n <= p->n_max;
    FusedLayout fl{};
    bool defer = false;            // slabs + chunk sums consumed by the loss kernel / the optimiser
    if (fuse && p->col_partials != nullptr && aligned16(p->col_partials)) {
        fl = fused_layout(p, static_cast<char *>(p->fused_workspace), p->col_partials);
        defer = p->fused_workspace == nullptr ? fl.bytes == 0
                                              : (aligned16(p->fused_workspace) && fl.bytes <= p->fused_workspace_bytes);
        if (!defer) fl = FusedLayout{};
    }
    uint64_t offs[GIST_MAX_LAYERS];
    {
        uint64_t off = drop_offset;
        for (int k = 0; k < L1; ++k) {
            offs[k] = off;
            if (drop) off += round_up2((uint64_t)n * 2 * p->layer[k].n_in);
        }
    }
    bool plain[GIST_MAX_LAYERS], fwd_fold[GIST_MAX_LAYERS], bwd_fold[GIST_MAX_LAYERS];
    for (int k = 0; k < L1; ++k) {
        const gist_layer_desc &l = p->layer[k];
        plain[k] = !h3.layer[k].on && !b3.layer[k].on;
        // forward: dropout([h | ah]) written by the producers, the aggregation reads hsrc[k]
        fwd_fold[k] = fuse && drop && plain[k] && p->hsrc[k] != nullptr && (offs[k] & 1) == 0 &&
                      (k > 0 || (flags & (GIST_STEP_EXTRACT | GIST_STEP_PREEXTRACTED))) &&
                      p->ld_hsrc[k] >= l.n_in &&
                      spmm_drop_takes(1, l.n_in, p->ld_hsrc[k], l.ldz, p->hsrc[k], l.Z + l.n_in, blocked ? p->row_blocks : nullptr);
        // backward: the mask of dZ_k applied by the reverse aggregation as it reads dZ_k
        bwd_fold[k] = fuse && drop && plain[k] && k > 0 && k < L1 - 1 && (offs[k] & 1) == 0 &&
                      spmm_drop_takes(2, l.n_in, 2 * l.n_in, 2 * l.n_in, p->dZ + l.n_in, p->dZ, blocked ? p->row_blocks : nullptr);
    }
    const float keep = drop ? 1.0f / (1.0f - p->p_drop) : 1.f;
    const uint64_t sm = p->seed * 0x9E3779B97F4A7C15ULL;
    // the class layer as gist_class_layer_f32 + gist_class_dw_slabs_f32 (projection, CE, dZ with its mask and the
    // bias gradient's chunk sums in one launch, dW as slabs for the optimiser) instead of four launches
    bool cls_fused = false;
    {
        const gist_layer_desc &l = p->layer[L1 - 1];
        cls_fused = train && defer && plain[L1 - 1] && p->ldc <= 64 && (offs[L1 - 1] & 1) == 0 &&
                    (int)tune(GIST_TUNE_CLASS_FUSED) != 1 &&
                    gist_class_layer_takes(n, l.n_out, 2 * l.n_in, l.ldz, 2 * l.n_in, l.Z, l.W) == 1 &&
                    fl.dw_slabs[L1 - 1] != nullptr &&
                    fl.dw_bytes[L1 - 1] >= gist_class_dw_slab_bytes(n, l.n_out, 2 * l.n_in);
    }
    // layers on split operands (their dZ comes from a split projection, unmasked): can the reverse aggregation
    // carry dZ_k's mask instead of a dropout pass over dZ_k?
    bool bwd_fold_split[GIST_MAX_LAYERS];
    for (int k = 0; k < L1; ++k) {
        const gist_layer_desc &l = p->layer[k];
        bwd_fold_split[k] = fuse && drop && !plain[k] && k > 0 && k < L1 - 1 && (offs[k] & 1) == 0 &&
                            spmm_drop_takes(2, l.n_in, 2 * l.n_in, 2 * l.n_in, p->dZ + l.n_in, p->dZ,
                                            blocked ? p->row_blocks : nullptr);
    }

    if (b3.any && do_fwd) {      // this step's weights, one read each
        Scope sc(p->timer, 3, 0, 0, 0, st);
        for (int k = 0; k < L1; ++k) {
            const B3Layer &hl = b3.layer[k];
            if (!hl.on) continue;
            const gist_layer_desc &l = p->layer[k];
            B3Dual d{};
            d.src = l.W; d.ld = 2 * l.n_in; d.rows = l.n_out; d.cols = 2 * l.n_in;
            d.dst_r = hl.Ws;
            d.dst_t = train ? hl.WsT : nullptr;
            GIST_TRY(b3_dual_split(d, st));
        }
    }
    if (h3.any && do_fwd) {      // this step's weights: rows split for Y = Z.W^T, transposed for dZ = dY.W
        Scope sc(p->timer, 3, 0, 0, 0, st);
        bool zeroed = false;
        for (int k = 0; k < L1; ++k) {
            const H3Layer &hl = h3.layer[k];
            if (!hl.on) continue;
            const gist_layer_desc &l = p->layer[k];
            if (k > 0 && train) {      // one read of W_k, one scale for the tensor, both layouts
                if (!zeroed) {
                    if (hipMemsetAsync(h3.amax, 0, GIST_MAX_LAYERS * 4, st) != hipSuccess) {
                        set_error("gist_sage_step: hipMemsetAsync failed");
                        return GIST_ELAUNCH;
                    }
                    zeroed = true;
                }
                GIST_TRY(h3_absmax(l.W, 2 * l.n_in, l.n_out, 2 * l.n_in, h3.amax + k, st));
                H3Dual d{};
                d.src = l.W; d.ld = 2 * l.n_in; d.rows = l.n_out; d.cols = 2 * l.n_in;
                d.amax = h3.amax + k;
                d.dst_r = hl.Ws; d.inv_r = hl.inv_wr; d.dst_t = hl.WsT; d.inv_t = hl.inv_wt;
                GIST_TRY(h3_dual_split(d, st));
            } else {
                GIST_TRY(h3_split_rows(l.W, 2 * l.n_in, l.n_out, 2 * l.n_in, hl.Ws, hl.inv_wr, st));
            }
        }
    }

    // layer 0's aggregation formed by the extraction (this call's, or the previous call's GIST_STEP_EXTRACT_NEXT)?
    bool pre_ah = (flags & GIST_STEP_PREEXTRACTED) && p->feat_intra != nullptr;
    if ((flags & GIST_STEP_EXTRACT) && do_fwd) {      // (pre_ah only matters to the forward loop)
        GIST_REQUIRE(ids != nullptr, "gist_sage_step: null ids");
        const gist_layer_desc &l0 = p->layer[0];
        const bool by_parts = fuse && p->node_part && p->part_slot && p->extract_scratch && p->batch_index >= 0 && n <= p->n_max &&
                              gist_extract_parts_supported(p->n_max) == 1;
        if (by_parts) {
            gist_extract_parts_desc x{};
            x.g_rowptr = p->g_rowptr; x.g_col = p->g_col; x.g_t_rowptr = p->g_t_rowptr; x.g_t_col = p->g_t_col;
            x.ids = ids; x.n = n; x.n_max = p->n_max;
            x.node_part = p->node_part; x.part_slot = p->part_slot; x.batch = p->batch_index;
            x.rowptr = p->rowptr; x.col = p->col; x.t_rowptr = p->t_rowptr; x.t_col = p->t_col;
            x.col_capacity = p->col_capacity; x.norm = p->norm;
            x.feat = p->feat; x.ld_feat = p->ld_feat; x.n_feat = l0.n_in; x.z0 = l0.Z; x.ldz0 = l0.ldz;
            x.labels_all = p->labels_all; x.labels = p->labels;
            x.x0 = fwd_fold[0] ? p->hsrc[0] : nullptr; x.ldx0 = p->ld_hsrc[0]; x.p = p->p_drop; x.seed = p->seed;
            x.offset = offs[0]; x.mask_ld = 2 * l0.n_in; x.scratch = p->extract_scratch;
            if (p->feat_intra != nullptr) {      // layer 0's aggregation comes with the extraction
                x.feat_intra = p->feat_intra; x.ld_intra = p->ld_feat_intra; x.ah = l0.Z + l0.n_in;
                pre_ah = true;
            }
            GIST_TRY(gist_extract_parts_desc_batch(&x, s));
        } else if (fwd_fold[0])
            GIST_TRY(gist_extract_batch_drop(p->g_rowptr, p->g_col, p->g_t_rowptr, p->g_t_col, ids, n,
                                             p->remap, p->rowptr, p->col, p->t_rowptr, p->t_col,
                                             p->col_capacity, p->norm, p->feat, p->ld_feat, l0.n_in, l0.Z,
                                             l0.ldz, p->labels_all, p->labels, p->hsrc[0], p->ld_hsrc[0],
                                             p->p_drop, p->seed, offs[0], 2 * l0.n_in, s));
        else
            GIST_TRY(gist_extract_batch(p->g_rowptr, p->g_col, p->g_t_rowptr, p->g_t_col, ids, n,
                                        p->remap, p->rowptr, p->col, p->t_rowptr, p->t_col,
                                        p->col_capacity, p->norm, p->feat, p->ld_feat,
                                        l0.n_in, l0.Z, l0.ldz, p->labels_all, p->labels, s));
    }

    // ---- block structure of the batch, once for all its aggregations ----------------
    const void *prep_fwd = nullptr, *prep_bwd = nullptr;
    if (blocked && p->spmm_prepared != nullptr && aligned16(p->spmm_prepared)) {
        const int64_t one = gist_spmm_blocks_bytes(p->n_row_blocks);
        bool wide = false;      // is there an aggregation the prepared kernel takes?
        for (int k = 0; k < L1; ++k)
            wide = wide || spmm_prepared_takes(p->layer[k].n_in, p->layer[k].ldz, p->layer[k].ldz,
                                               p->layer[k].Z, p->layer[k].Z + p->layer[k].n_in);
        if (wide && p->spmm_prepared_bytes >= (train ? 2 : 1) * one) {
            char *base = static_cast<char *>(p->spmm_prepared);
            if (do_fwd)
            GIST_TRY(launch_spmm_blocks_prepare(p->rowptr, p->col, train ? p->t_rowptr : nullptr,
                                                train ? p->t_col : nullptr, n, p->row_blocks, p->n_row_blocks,
                                                base, train ? base + one : nullptr, st));
            prep_fwd = base;
            prep_bwd = train ? base + one : nullptr;
        }
    }

    // ---- forward (modules.py:310-314 / :218-237) ---------------------------------
    int logit_slabs = 0;           // > 1: the class layer's logits are still split-K slabs
    for (int k = 0; k < L1 && do_fwd; ++k) {
        const gist_layer_desc &l = p->layer[k];
        int y_slabs_n = 1;                   // > 1: this layer's pre-norm output is still split-K slabs
        const float *y_slabs = nullptr;
        if (!(k == 0 && pre_ah)) {
            Scope sc(p->timer, 0, n, n, l.n_in, st);
            if (fwd_fold[k]) {      // source = the undropped input, store = dropout(ah)
                SpmmDrop dr{};
                dr.mode = 1; dr.p = p->p_drop; dr.scale = keep; dr.sm = sm;
                dr.y_base = offs[k] + (uint64_t)l.n_in; dr.src_base = 0; dr.ld = 2 * l.n_in;
                GIST_TRY(spmm_drop(p->rowptr, p->col, p->hsrc[k], p->ld_hsrc[k], l.Z + l.n_in, l.ldz, n, l.n_in,
                                   p->norm, nullptr, 0, blocked ? p->row_blocks : nullptr, p->n_row_blocks, dr, st,
                                   prep_fwd));
            } else {
                GIST_TRY(step_spmm(p, p->rowptr, p->col, l.Z, l.ldz, l.Z + l.n_in, l.ldz, n, l.n_in,
                                   p->norm, nullptr, 0, prep_fwd, s));
            }
        }
        if (h3.layer[k].on) {
            // dropout + split of Z_k in one pass (both layouts when training); the dropped
            // fp32 Z_k is never written: the backward only needs its transposed split
            const H3Layer &hl = h3.layer[k];
            Scope sc(p->timer, 1, n, l.n_out, 2 * l.n_in, st);
            H3Dual d{};
            d.src = l.Z; d.ld = l.ldz; d.rows = n; d.cols = 2 * l.n_in;
            d.p = drop ? p->p_drop : 0.f; d.seed = p->seed; d.offset = offs[k];
            d.fixed_shift = hl.shift;
            d.dst_r = hl.Zs; d.inv_r = hl.inv_zr;
            d.dst_t = train ? hl.ZsT : nullptr; d.inv_t = hl.inv_zt;
            GIST_TRY(h3_dual_split(d, st));
            GIST_TRY(h3_gemm_presplit("gist_sage_step", hl.Zs, hl.inv_zr, hl.Ws, hl.inv_wr, l.b, l.Y,
                                      l.ldy, n, l.n_out, 2 * l.n_in, st));
        } else if (b3.layer[k].on) {
            const B3Layer &hl = b3.layer[k];
            Scope sc(p->timer, 1, n, l.n_out, 2 * l.n_in, st);
            B3Dual d{};
            d.src = l.Z; d.ld = l.ldz; d.rows = n; d.cols = 2 * l.n_in;
            d.p = drop ? p->p_drop : 0.f; d.seed = p->seed; d.offset = offs[k];
            d.dst_r = hl.Zs;
            d.dst_t = train ? hl.ZsT : nullptr;
            GIST_TRY(b3_dual_split(d, st));
            // (a hidden layer's k slices stay slabs: its LayerNorm, the next launch, sums them as it reads)
            // (with one slice the kernel adds the bias itself; slabs get it from the LayerNorm)
#ifdef STEP_NO_YSLABS
            const bool to_ln = false;
#else
            const bool to_ln = defer && k + 1 < L1;
#endif
            GIST_TRY(b3_gemm_presplit("gist_sage_step", hl.Zs, hl.Ws, l.b, l.Y, l.ldy, n, l.n_out,
                                      2 * l.n_in, b3.slabs, b3.slab_bytes, st, to_ln ? &y_slabs_n : nullptr));
            if (to_ln) y_slabs = b3.slabs;
        } else {
            if (drop && !fwd_fold[k])
                GIST_TRY(gist_dropout_f32(l.Z, l.ldz, n, 2 * l.n_in, p->p_drop, p->seed, offs[k], s));
            if (cls_fused && k == L1 - 1) continue;      // (projection, loss and dZ follow in one launch)
            Scope sc(p->timer, 1, n, l.n_out, 2 * l.n_in, st);
            if (defer && k == L1 - 1 && fl.logit_slabs != nullptr) {      // the loss kernel sums the slabs
                GIST_TRY(gemm_slabs(0, l.Z, l.ldz, l.W, 2 * l.n_in, l.b, l.Y, l.ldy, n, l.n_out, 2 * l.n_in,
                                    fl.logit_slabs, fl.logit_bytes, &logit_slabs, st));
#ifndef STEP_NO_YSLABS      // dev A/B build flag
            } else if (defer && k + 1 < L1 && fl.y_slabs != nullptr && !h3_eligible(n, l.n_out, 2 * l.n_in) &&
                       !b3_eligible(n, l.n_out, 2 * l.n_in)) {
                // the LayerNorm sums the slabs (a projection the per-call split paths take keeps their workspace)
                GIST_TRY(gemm_slabs(0, l.Z, l.ldz, l.W, 2 * l.n_in, l.b, l.Y, l.ldy, n, l.n_out, 2 * l.n_in,
                                    fl.y_slabs, fl.y_bytes, &y_slabs_n, st));
                y_slabs = fl.y_slabs;
#endif
            } else {
                GIST_TRY(gist_gemm_nt_f32(l.Z, l.ldz, l.W, 2 * l.n_in, l.b, l.Y, l.ldy, n, l.n_out,
                                          2 * l.n_in, p->workspace, p->workspace_bytes, s));
            }
        }
        if (k + 1 < L1) {
            const gist_layer_desc &nx = p->layer[k + 1];
            if (y_slabs_n > 1)
                GIST_TRY(gist_ln_relu_fwd_slabs_f32(l.Y, l.ldy, y_slabs, n * l.n_out, y_slabs_n, l.b, nx.Z, nx.ldz,
                                                    fwd_fold[k + 1] ? p->hsrc[k + 1] : nullptr,
                                                    fwd_fold[k + 1] ? p->ld_hsrc[k + 1] : 0,
                                                    p->use_layernorm ? l.rstd : nullptr, n, l.n_out,
                                                    p->use_layernorm, 1, 1e-5f, fwd_fold[k + 1] ? p->p_drop : 0.f,
                                                    p->seed, fwd_fold[k + 1] ? offs[k + 1] : 0,
                                                    fwd_fold[k + 1] ? 2 * nx.n_in : l.n_out, s));
            else if (fwd_fold[k + 1])
                GIST_TRY(gist_ln_relu_fwd_drop_f32(l.Y, l.ldy, nx.Z, nx.ldz, p->hsrc[k + 1], p->ld_hsrc[k + 1],
                                                   p->use_layernorm ? l.rstd : nullptr, n, l.n_out,
                                                   p->use_layernorm, 1, 1e-5f, p->p_drop, p->seed,
                                                   offs[k + 1], 2 * nx.n_in, s));
            else
                GIST_TRY(gist_ln_relu_fwd_f32(l.Y, l.ldy, nx.Z, nx.ldz,
                                              p->use_layernorm ? l.rstd : nullptr, n, l.n_out,
                                              p->use_layernorm, 1, 1e-5f, s));
        }
    }
    const gist_layer_desc &last = p->layer[L1 - 1];
    // the optimiser kernel reduces the loss when it runs with deferred work anyway
    const bool loss_in_adam = train && defer;
    if (!do_fwd) {
    } else if (cls_fused) {
        Scope sc(p->timer, 1, n, (L1 > 1 ? 2 : 1) * last.n_out, 2 * last.n_in, st);
        GIST_TRY(gist_class_layer_f32(last.Z, last.ldz, last.W, 2 * last.n_in, last.b, p->labels, n, last.Y, last.ldy,
                                      p->dlogits, p->ldc, p->row_loss, L1 > 1 ? p->dZ : nullptr, 2 * last.n_in,
                                      drop ? p->p_drop : 0.f, p->seed, offs[L1 - 1], fl.partials[L1 - 1], n,
                                      last.n_out, 2 * last.n_in, s));
    } else
    GIST_TRY(softmax_xent_ex("gist_sage_step", last.Y, last.ldy, logit_slabs > 1 ? fl.logit_slabs : nullptr,
                             n * last.n_out, logit_slabs > 1 ? logit_slabs : 0, last.b, p->labels, nullptr, n,
                             p->row_loss, loss_in_adam ? nullptr : p->loss, p->dlogits, p->ldc, n,
                             last.n_out, st));
    if (!train) return GIST_OK;
    // a forward-phase call leaves the loss complete: its optimiser launch is another call
    if (do_fwd && split_phases && loss_in_adam) GIST_TRY(loss_finish(p->row_loss, n, n, p->loss, st));

    // ---- backward --------------------------------------------
    // GIST_STEP_DLOGITS_GIVEN: plan->dlogits was written by the caller (any loss on the logits): the class layer's dZ and
    // bias chunk sums of the fused forward belong to ANOTHER dlogits and are recomputed from the given one
    const bool cls_dz_done = cls_fused && !dlogits_given;
    gist_grad_segment segs[2 * GIST_MAX_LAYERS];
    int n_segs = 0;
    const int64_t chunks16 = gist_row_chunks16(n);
    int64_t lnb_rows = 0;           // > 0: the layer below got its LayerNorm backward (and that many rows of partial sums) from
                                    // the reverse aggregation just launched
    auto bias_segment = [&](int k, int64_t rows) {      // db_k = chunk sums, formed by the optimiser
        const gist_layer_desc &l = p->layer[k];
        gist_grad_segment &g = segs[n_segs++];
        g.begin = l.db - p->grads; g.end = g.begin + l.n_out;
        g.src = fl.partials[k]; g.stride = l.n_out; g.n_src = (int32_t)rows;
    };
    ClassDwArgs dw_args{};          // the class layer's weight-gradient slabs, deferred to the LayerNorm backward below it
    bool dw_pending = false;
    for (int k = L1 - 1; k >= 0 && do_bwd; --k) {
        const gist_layer_desc &l = p->layer[k];
        const float *dy;
        int64_t lddy;
        bool db_done = false;      // this layer's bias gradient is already in chunks
        const int64_t db_rows = lnb_rows > 0 ? lnb_rows : chunks16;
        if (k == L1 - 1) {
            dy = p->dlogits;
            lddy = p->ldc;
        } else {
            const int64_t i_next = p->layer[k + 1].n_in;      // == l.n_out
            if (lnb_rows > 0) {      // (came with the reverse aggregation of layer k + 1)
                db_done = true;
                lnb_rows = 0;
            } else if (defer && plain[k]) {
                if (dw_pending) {      // ... with the class layer's weight-gradient slabs in the same grid
                    GIST_TRY(ln_relu_bwd_colsum_class_dw(p->dZ, 2 * i_next, l.Y, l.ldy, p->use_layernorm ? l.rstd : nullptr,
                                                         l.Y, l.ldy, n, l.n_out, p->use_layernorm, 1, fl.partials[k], dw_args,
                                                         st));
                    dw_pending = false;
                } else {
                    GIST_TRY(ln_relu_bwd_colsum(p->dZ, 2 * i_next, l.Y, l.ldy, p->use_layernorm ? l.rstd : nullptr,
                                                l.Y, l.ldy, n, l.n_out, p->use_layernorm, 1, fl.partials[k], st));
                }
                db_done = true;
            } else {
                GIST_TRY(ln_relu_bwd_ex(p->dZ, 2 * i_next, l.Y, l.ldy,
                                        p->use_layernorm ? l.rstd : nullptr, l.Y, l.ldy, n, l.n_out,
                                        p->use_layernorm, 1, h3.layer[k].on ? h3.rowmax : nullptr, st));
            }
            dy = l.Y;
            lddy = l.ldy;
        }
        if (h3.layer[k].on) {
            // bias gradient + column maxima of dY_k, one read of dY_k -> both split layouts
            // (row maxima came from the LayerNorm backward), then dZ_k and dW_k on the splits
            const H3Layer &hl = h3.layer[k];
            GIST_TRY(colsum_ex(dy, lddy, n, l.n_out, p->partials, l.db, h3.pmax, h3.colmax, st));
            {
                Scope sc(p->timer, 3, 0, 0, 0, st);
                H3Dual d{};
                d.src = dy; d.ld = lddy; d.rows = n; d.cols = l.n_out;
                d.rowmax = h3.rowmax; d.colmax = h3.colmax;
                d.dst_r = k > 0 ? h3.dYs : nullptr; d.inv_r = h3.inv_dyr;
                d.dst_t = h3.dYsT; d.inv_t = h3.inv_dyt;
                GIST_TRY(h3_dual_split(d, st));
            }
            if (k > 0) {
                Scope sc(p->timer, 1, n, 2 * l.n_in, l.n_out, st);
                GIST_TRY(h3_gemm_presplit("gist_sage_step", h3.dYs, h3.inv_dyr, hl.WsT, hl.inv_wt,
                                          nullptr, p->dZ, 2 * l.n_in, n, 2 * l.n_in, l.n_out, st));
            }
            {
                Scope sc(p->timer, 1, l.n_out, 2 * l.n_in, n, st);
                GIST_TRY(h3_gemm_presplit("gist_sage_step", h3.dYsT, h3.inv_dyt, hl.ZsT, hl.inv_zt,
                                          nullptr, l.dW, 2 * l.n_in, l.n_out, 2 * l.n_in, n, st));
            }
            if (k > 0) {
                if (bwd_fold_split[k]) {
                    Scope sc(p->timer, 0, n, n, l.n_in, st);
                    SpmmDrop dr{};
                    dr.mode = 2; dr.p = p->p_drop; dr.scale = keep; dr.sm = sm;
                    dr.y_base = offs[k]; dr.src_base = offs[k] + (uint64_t)l.n_in; dr.ld = 2 * l.n_in;
                    GIST_TRY(spmm_drop(p->t_rowptr, p->t_col, p->dZ + l.n_in, 2 * l.n_in, p->dZ, 2 * l.n_in, n,
                                       l.n_in, nullptr, p->norm, 1, p->row_blocks, p->n_row_blocks, dr, st, prep_bwd));
                } else {
                    if (drop)
                        GIST_TRY(gist_dropout_f32(p->dZ, 2 * l.n_in, n, 2 * l.n_in, p->p_drop, p->seed,
                                                  offs[k], s));
                    Scope sc(p->timer, 0, n, n, l.n_in, st);
                    GIST_TRY(step_spmm(p, p->t_rowptr, p->t_col, p->dZ + l.n_in, 2 * l.n_in, p->dZ,
                                       2 * l.n_in, n, l.n_in, nullptr, p->norm, 1, prep_bwd, s));
                }
            }
            continue;
        }
        if (b3.layer[k].on) {
            // bias gradient, then one read of dY_k -> both split layouts, dZ_k and dW_k on the splits
            const B3Layer &hl = b3.layer[k];
            {   // (the split's 64 x 64 tiles also give the bias gradient's per-chunk column sums)
                Scope sc(p->timer, 3, 0, 0, 0, st);
                B3Dual d{};
                d.src = dy; d.ld = lddy; d.rows = n; d.cols = l.n_out;
                d.dst_r = k > 0 ? b3.dYs : nullptr;
                d.dst_t = b3.dYsT;
                // (fused step: the tiles' column sums stay in the layer's own partials and the optimiser forms db_k from them
                // -- 32 sources per element, a dedicated-block segment -- instead of a 5-us reduce launch per layer)
                d.col_partials = defer && fl.partials[k] != nullptr ? fl.partials[k] : p->partials;
                GIST_TRY(b3_dual_split(d, st));
            }
            if (defer && fl.partials[k] != nullptr) {
                gist_grad_segment &g = segs[n_segs++];
                g.begin = l.db - p->grads; g.end = g.begin + l.n_out;
                g.src = fl.partials[k]; g.stride = l.n_out; g.n_src = (int32_t)gist_colsum_partials(n);
            } else {
                GIST_TRY(colsum_finish(p->partials, gist_colsum_partials(n), l.n_out, l.db, st));
            }
            if (k > 0) {
                Scope sc(p->timer, 1, n, 2 * l.n_in, l.n_out, st);
                GIST_TRY(b3_gemm_presplit("gist_sage_step", b3.dYs, hl.WsT, nullptr, p->dZ, 2 * l.n_in, n,
                                          2 * l.n_in, l.n_out, b3.slabs, b3.slab_bytes, st));
            }
            {
                Scope sc(p->timer, 1, l.n_out, 2 * l.n_in, n, st);
                // the step's LAST projection may leave its k slices to the optimiser (the slab scratch is not
                // reused before it runs)
                int ns = 1;
                GIST_TRY(b3_gemm_presplit("gist_sage_step", b3.dYsT, hl.ZsT, nullptr, l.dW, 2 * l.n_in,
                                          l.n_out, 2 * l.n_in, n, b3.slabs, b3.slab_bytes, st,
                                          defer && k == 0 ? &ns : nullptr));
                if (ns > 1) {
                    gist_grad_segment &g = segs[n_segs++];
                    g.begin = l.dW - p->grads; g.end = g.begin + l.n_out * 2 * l.n_in;
                    g.src = b3.slabs; g.stride = l.n_out * 2 * l.n_in; g.n_src = ns;
                }
            }
            if (k > 0) {
                if (bwd_fold_split[k]) {
                    Scope sc(p->timer, 0, n, n, l.n_in, st);
                    SpmmDrop dr{};
                    dr.mode = 2; dr.p = p->p_drop; dr.scale = keep; dr.sm = sm;
                    dr.y_base = offs[k]; dr.src_base = offs[k] + (uint64_t)l.n_in; dr.ld = 2 * l.n_in;
                    GIST_TRY(spmm_drop(p->t_rowptr, p->t_col, p->dZ + l.n_in, 2 * l.n_in, p->dZ, 2 * l.n_in, n,
                                       l.n_in, nullptr, p->norm, 1, p->row_blocks, p->n_row_blocks, dr, st, prep_bwd));
                } else {
                    if (drop)
                        GIST_TRY(gist_dropout_f32(p->dZ, 2 * l.n_in, n, 2 * l.n_in, p->p_drop, p->seed,
                                                  offs[k], s));
                    Scope sc(p->timer, 0, n, n, l.n_in, st);
                    GIST_TRY(step_spmm(p, p->t_rowptr, p->t_col, p->dZ + l.n_in, 2 * l.n_in, p->dZ,
                                       2 * l.n_in, n, l.n_in, nullptr, p->norm, 1, prep_bwd, s));
                }
            }
            continue;
        }
        // dZ_k and dW_k of a narrow hidden layer read the same dY_k: one launch of the fp32 kernel's tiles for both
        // (gist_gemm_nn_tn_dual_f32) when the mask of dZ_k is the reverse aggregation's business anyway
        bool dual_done = false;
        if (defer && k > 0 && k < L1 - 1 && db_done && (!drop || bwd_fold[k]) &&
            gemm_dual_takes(n, 2 * l.n_in, l.n_out, lddy, 2 * l.n_in, l.ldz, 2 * l.n_in, dy, l.W, l.Z, p->dZ)) {
            Scope sc(p->timer, 1, n, 4 * l.n_in, l.n_out, st);
            int ns = 1;
            GIST_TRY(gemm_dual_nn_tn("gist_sage_step", dy, lddy, l.W, 2 * l.n_in, p->dZ, 2 * l.n_in, l.Z, l.ldz, l.dW,
                                     2 * l.n_in, n, 2 * l.n_in, l.n_out, fl.dw_slabs[k], fl.dw_bytes[k], &ns, st));
            if (ns > 1) {
                gist_grad_segment &g = segs[n_segs++];
                g.begin = l.dW - p->grads; g.end = g.begin + l.n_out * 2 * l.n_in;
                g.src = fl.dw_slabs[k]; g.stride = l.n_out * 2 * l.n_in; g.n_src = ns;
            }
            dual_done = true;
        }
        if (dual_done) {
        } else if (cls_dz_done && k == L1 - 1) {
            db_done = true;      // (dZ and the bias chunks came with the loss)
        } else if (k > 0) {      // dZ with its dropout mask (or the mask left to the reverse aggregation)
            Scope sc(p->timer, 1, n, 2 * l.n_in, l.n_out, st);
            const bool chunk_db = defer && k == L1 - 1;      // the class layer's dZ kernel sees dlogits in 16-row chunks
            GIST_TRY(gemm_nn_dropout_ex("gist_sage_step", dy, lddy, l.W, 2 * l.n_in, p->dZ, 2 * l.n_in, n,
                                        2 * l.n_in, l.n_out, (drop && !bwd_fold[k]) ? p->p_drop : 0.f, p->seed,
                                        offs[k], p->workspace, p->workspace_bytes,
                                        chunk_db ? fl.partials[k] : nullptr, st));
            db_done = db_done || chunk_db;
        }
        if (!dual_done) {
            // The class layer's weight-gradient slabs need dlogits and the layer's input only; the LayerNorm backward of
            // the layer below (next iteration of this loop, behind the reverse aggregation) runs 128 workgroups for ~6 us:
            // the slabs' workgroups go into ITS grid (ln_relu_bwd_cs_dw_kernel) -- one launch fewer per step
            const bool dw_with_ln = cls_fused && k == L1 - 1 && L1 >= 2 && defer && plain[L1 - 2] &&
                                    (int)tune(GIST_TUNE_CLASS_FUSED) != 2;
            Scope sc(dw_with_ln ? nullptr : p->timer, 1, l.n_out, 2 * l.n_in, n, st);
            if (cls_fused && k == L1 - 1) {
                int32_t ns = 1;
                if (dw_with_ln) {
                    GIST_TRY(class_dw_args("gist_sage_step", dy, lddy, l.Z, l.ldz, fl.dw_slabs[k], fl.dw_bytes[k], n, l.n_out,
                                           2 * l.n_in, &dw_args, &ns));
                    dw_pending = true;
                } else {
                    GIST_TRY(gist_class_dw_slabs_f32(dy, lddy, l.Z, l.ldz, fl.dw_slabs[k], fl.dw_bytes[k], &ns, n, l.n_out,
                                                     2 * l.n_in, s));
                }
                gist_grad_segment &g = segs[n_segs++];
                g.begin = l.dW - p->grads; g.end = g.begin + l.n_out * 2 * l.n_in;
                g.src = fl.dw_slabs[k]; g.stride = l.n_out * 2 * l.n_in; g.n_src = ns;
            } else if (defer && fl.dw_slabs[k] != nullptr) {      // the optimiser sums the slabs
                int ns = 1;
                GIST_TRY(gemm_slabs(2, dy, lddy, l.Z, l.ldz, nullptr, l.dW, 2 * l.n_in, l.n_out, 2 * l.n_in, n,
                                    fl.dw_slabs[k], fl.dw_bytes[k], &ns, st));
                if (ns > 1) {
                    gist_grad_segment &g = segs[n_segs++];
                    g.begin = l.dW - p->grads; g.end = g.begin + l.n_out * 2 * l.n_in;
                    g.src = fl.dw_slabs[k]; g.stride = l.n_out * 2 * l.n_in; g.n_src = ns;
                }
            } else {
                GIST_TRY(gist_gemm_tn_f32(dy, lddy, l.Z, l.ldz, l.dW, 2 * l.n_in, l.n_out,
                                          2 * l.n_in, n, p->workspace, p->workspace_bytes, s));
            }
        }
        if (defer) {
            if (!db_done)      // (a one-layer model: no dZ kernel has seen dlogits)
                GIST_TRY(colsum_rows16(dy, lddy, n, l.n_out, fl.partials[k], k == L1 - 1 ? false : true, st));
            bias_segment(k, db_rows);
        } else {
            GIST_TRY(gist_colsum_f32(dy, lddy, n, l.n_out, p->partials, l.db, s));
        }
        if (k > 0) {
            Scope sc(p->timer, 0, n, n, l.n_in, st);
            if (bwd_fold[k]) {
                SpmmDrop dr{};
                dr.mode = 2; dr.p = p->p_drop; dr.scale = keep; dr.sm = sm;
                dr.y_base = offs[k]; dr.src_base = offs[k] + (uint64_t)l.n_in; dr.ld = 2 * l.n_in;
                // the LayerNorm + ReLU backward of layer k - 1 in this launch's store (its rows are whole in one wave)
                const gist_layer_desc &lo = p->layer[k - 1];
                SpmmLnBwd ln{};
                const int64_t units = blocked ? spmm_lnb_units(p->n_row_blocks) : 0;
                const bool with_ln = defer && plain[k - 1] && !dw_pending && (int)tune(GIST_TUNE_LNB_FUSED) != 1 &&
                                     units > 0 && units <= fl.partial_rows[k - 1] && lo.ldy % 4 == 0 && aligned16(lo.Y) &&
                                     (!p->use_layernorm || lo.rstd != nullptr) &&
                                     spmm_lnb_takes(l.n_in, 2 * l.n_in, 2 * l.n_in, p->dZ + l.n_in, p->dZ, p->row_blocks, prep_bwd);
                if (with_ln) {
                    ln.yhat = lo.Y; ln.ldy = lo.ldy; ln.rstd = p->use_layernorm ? lo.rstd : nullptr;
                    ln.dy = lo.Y; ln.lddy = lo.ldy; ln.col_partials = fl.partials[k - 1]; ln.relu = 1;
                    lnb_rows = units;
                }
                GIST_TRY(spmm_drop(p->t_rowptr, p->t_col, p->dZ + l.n_in, 2 * l.n_in, p->dZ, 2 * l.n_in, n,
                                   l.n_in, nullptr, p->norm, 1, p->row_blocks, p->n_row_blocks, dr, st, prep_bwd,
                                   with_ln ? &ln : nullptr));
            } else {
                GIST_TRY(step_spmm(p, p->t_rowptr, p->t_col, p->dZ + l.n_in, 2 * l.n_in, p->dZ,
                                   2 * l.n_in, n, l.n_in, nullptr, p->norm, 1, prep_bwd, s));
            }
        }
    }
    GIST_REQUIRE(!dw_pending, "gist_sage_step: internal error (class-layer weight gradient not launched)");
    if (do_bwd && !do_opt) {
        // the gradient arena complete on return (p.grad is read by the caller's optimiser, possibly by its own code first):
        // the deferred sums in the optimiser's order, without the update
        if (defer && n_segs > 0) GIST_TRY(gist_grad_segments_finish_f32(p->grads, p->n_params, segs, n_segs, s));
        return GIST_OK;
    }
    if (!do_opt) return GIST_OK;
    if (!do_bwd) n_segs = 0;      // (an optimiser-phase call: the backward-phase call finished the gradients)
    const float *opt_row_loss = do_bwd || !split_phases ? p->row_loss : nullptr;      // (and the forward-phase call the loss)
    if (defer && (flags & GIST_STEP_EXTRACT_NEXT) && next_parts_ok(p, fuse) && p->n_params <= kPrefetchMaxParams) {
        // the optimiser and the NEXT batch's extraction in one grid: nothing reads the batch buffers any more.  Layer 0's
        // mask goes into the next batch's feature gather under the rule the next call applies to itself (fwd_fold[0])
        const gist_layer_desc &l0 = p->layer[0];
        const bool fold0 = fuse && drop && plain[0] && p->hsrc[0] != nullptr && (p->next_drop_offset & 1) == 0 &&
                           p->ld_hsrc[0] >= l0.n_in &&
                           spmm_drop_takes(1, l0.n_in, p->ld_hsrc[0], l0.ldz, p->hsrc[0], l0.Z + l0.n_in,
                                           blocked ? p->row_blocks : nullptr);
        gist_extract_parts_desc x{};
        x.g_rowptr = p->g_rowptr; x.g_col = p->g_col; x.g_t_rowptr = p->g_t_rowptr; x.g_t_col = p->g_t_col;
        x.ids = p->next_ids; x.n = p->next_n; x.n_max = p->n_max;
        x.node_part = p->node_part; x.part_slot = p->part_slot; x.batch = p->next_batch_index;
        x.rowptr = p->rowptr; x.col = p->col; x.t_rowptr = p->t_rowptr; x.t_col = p->t_col;
        x.col_capacity = p->col_capacity; x.norm = p->norm;
        x.feat = p->feat; x.ld_feat = p->ld_feat; x.n_feat = l0.n_in; x.z0 = l0.Z; x.ldz0 = l0.ldz;
        x.labels_all = p->labels_all; x.labels = p->labels;
        x.x0 = fold0 ? p->hsrc[0] : nullptr; x.ldx0 = p->ld_hsrc[0]; x.p = p->p_drop; x.seed = p->seed;
        x.offset = p->next_drop_offset; x.mask_ld = 2 * l0.n_in; x.scratch = p->extract_scratch;
        if (p->feat_intra != nullptr) { x.feat_intra = p->feat_intra; x.ld_intra = p->ld_feat_intra; x.ah = l0.Z + l0.n_in; }
        GIST_TRY(gist_adam_segments_extract_f32(p->params, p->grads, p->exp_avg, p->exp_avg_sq, p->n_params, lr, beta1,
                                                beta2, eps, weight_decay, adam_step, segs, n_segs, opt_row_loss, n, n,
                                                p->loss, &x, s));
    } else if (defer)
        GIST_TRY(gist_adam_segments_f32(p->params, p->grads, p->exp_avg, p->exp_avg_sq, p->n_params, lr, beta1,
                                        beta2, eps, weight_decay, adam_step, segs, n_segs, opt_row_loss, n, n,
                                        p->loss, s));
    else
        GIST_TRY(gist_adam_f32(p->params, p->grads, p->exp_avg, p->exp_avg_sq, p->n_params, lr, beta1,
                               beta2, eps, weight_decay, adam_step, s));
    return GIST_OK;
}
