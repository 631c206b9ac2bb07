// Native step driver: one C-ABI call issues the whole training iteration (batch
// extraction -> forward -> CE -> backward -> Adam) on one stream.  The host-side
// sequencing of the reference's loop body (cluster_gcn_ist_distrib.py:408-417) moves
// from ~45 interpreter round trips to ~45 back-to-back hipLaunch calls, which is what
// bounds the small-width (many-GPU) regime.  Arithmetic is unchanged: it calls the same
// entry points the op-level API exposes.
#include <new>

#include "common.h"

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
struct ActiveTimer {   // kernels below the entry points see the armed timer for this call only
    explicit ActiveTimer(gist_timer *t) { tl_timer = t; }
    ~ActiveTimer() { tl_timer = nullptr; }
};
}  // namespace

extern "C" int gist_sage_step(const gist_step_plan *p, const int32_t *ids, int64_t n,
                              uint64_t drop_offset, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int64_t adam_step, int flags,
                              gist_stream_t s) {
    GIST_REQUIRE(p != nullptr, "gist_sage_step: null plan");
    GIST_REQUIRE(p->n_layers >= 1 && p->n_layers <= GIST_MAX_LAYERS, "gist_sage_step: bad n_layers");
    GIST_REQUIRE(n > 0, "gist_sage_step: empty batch");
    const int L1 = p->n_layers;
    hipStream_t st = as_stream(s);
    ActiveTimer active(p->timer);
    const bool train = (flags & GIST_STEP_TRAIN) != 0;
    const bool drop = train && p->p_drop > 0.f;

    if (flags & GIST_STEP_EXTRACT) {
        GIST_REQUIRE(ids != nullptr, "gist_sage_step: null ids");
        GIST_TRY(gist_extract_batch(p->g_rowptr, p->g_col, p->g_t_rowptr, p->g_t_col, ids, n,
                                    p->remap, p->rowptr, p->col, p->t_rowptr, p->t_col,
                                    p->col_capacity, p->norm, p->feat, p->ld_feat,
                                    p->layer[0].n_in, p->layer[0].Z, p->layer[0].ldz,
                                    p->labels_all, p->labels, s));
    }

    // ---- forward (modules.py:310-314 / :218-237) ---------------------------------
    uint64_t offs[GIST_MAX_LAYERS];
    uint64_t off = drop_offset;
    for (int k = 0; k < L1; ++k) {
        const gist_layer_desc &l = p->layer[k];
        {
            Scope sc(p->timer, 0, n, n, l.n_in, st);
            GIST_TRY(gist_spmm_csr_f32(p->rowptr, p->col, l.Z, l.ldz, l.Z + l.n_in, l.ldz, n,
                                       l.n_in, p->norm, nullptr, 0, s));
        }
        offs[k] = off;
        if (drop) {
            GIST_TRY(gist_dropout_f32(l.Z, l.ldz, n, 2 * l.n_in, p->p_drop, p->seed, off, s));
            off += round_up2((uint64_t)n * 2 * l.n_in);
        }
        {
            Scope sc(p->timer, 1, n, l.n_out, 2 * l.n_in, st);
            GIST_TRY(gist_gemm_nt_f32(l.Z, l.ldz, l.W, 2 * l.n_in, l.b, l.Y, l.ldy, n, l.n_out,
                                      2 * l.n_in, p->workspace, p->workspace_bytes, s));
        }
        if (k + 1 < L1) {
            const gist_layer_desc &nx = p->layer[k + 1];
            GIST_TRY(gist_ln_relu_fwd_f32(l.Y, l.ldy, nx.Z, nx.ldz,
                                          p->use_layernorm ? l.rstd : nullptr, n, l.n_out,
                                          p->use_layernorm, 1, 1e-5f, s));
        }
    }
    const gist_layer_desc &last = p->layer[L1 - 1];
    GIST_TRY(gist_softmax_xent_f32(last.Y, last.ldy, p->labels, nullptr, n, p->row_loss, p->loss,
                                   p->dlogits, p->ldc, n, last.n_out, s));
    if (!train) return GIST_OK;

    // ---- backward --------------------------------------------
    // Adam is HBM-bound, the backward GEMMs are MFMA-bound: with GIST_STEP_OVERLAP_ADAM each
    // layer's parameter slice [W_k | b_k] is updated on a side stream as soon as dW_k, db_k
    // exist and W_k has been read for the last time (dZ = dY.W_k), concurrently with the
    // rest of the backward; the step joins the side stream before it returns.
    const bool overlap = (flags & GIST_STEP_OVERLAP_ADAM) != 0 && L1 > 1;
    const bool overlap_dw = (flags & GIST_STEP_OVERLAP_DW) != 0 && !overlap && p->workspace2 != nullptr;
    static hipStream_t side = nullptr;
    static hipEvent_t ev_ready[GIST_MAX_LAYERS], ev_done = nullptr;
    if ((overlap || overlap_dw) && side == nullptr) {
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) {
            set_error("gist_sage_step: cannot create the Adam side stream");
            return GIST_ELAUNCH;
        }
        for (int k = 0; k < GIST_MAX_LAYERS; ++k)
            (void)hipEventCreateWithFlags(&ev_ready[k], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&ev_done, hipEventDisableTiming);
    }
    for (int k = L1 - 1; k >= 0; --k) {
        const gist_layer_desc &l = p->layer[k];
        const float *dy;
        int64_t lddy;
        if (k == L1 - 1) {
            dy = p->dlogits;
            lddy = p->ldc;
        } else {
            const int64_t i_next = p->layer[k + 1].n_in;      // == l.n_out
            GIST_TRY(gist_ln_relu_bwd_f32(p->dZ, 2 * i_next, l.Y, l.ldy,
                                          p->use_layernorm ? l.rstd : nullptr, l.Y, l.ldy, n,
                                          l.n_out, p->use_layernorm, 1, s));
            dy = l.Y;
            lddy = l.ldy;
        }
        if (overlap_dw) {
            // dY_k is complete on the main stream: fork dW_k / db_k to the side stream, where
            // all weight-gradient work is serialised (one split-K scratch, one colsum scratch).
            (void)hipEventRecord(ev_ready[k], st);
            (void)hipStreamWaitEvent(side, ev_ready[k], 0);
            {
                Scope sc(p->timer, 1, l.n_out, 2 * l.n_in, n, side);
                GIST_TRY(gist_gemm_tn_f32(dy, lddy, l.Z, l.ldz, l.dW, 2 * l.n_in, l.n_out,
                                          2 * l.n_in, n, p->workspace2, p->workspace2_bytes,
                                          (gist_stream_t)side));
            }
            GIST_TRY(gist_colsum_f32(dy, lddy, n, l.n_out, p->partials, l.db, (gist_stream_t)side));
        }
        if (k > 0) {      // dZ (before dW on the main stream: it is the last reader of W_k)
            Scope sc(p->timer, 1, n, 2 * l.n_in, l.n_out, st);
            GIST_TRY(gist_gemm_nn_f32(dy, lddy, l.W, 2 * l.n_in, p->dZ, 2 * l.n_in, n, 2 * l.n_in,
                                      l.n_out, p->workspace, p->workspace_bytes, s));
        }
        if (!overlap_dw) {
            {
                Scope sc(p->timer, 1, l.n_out, 2 * l.n_in, n, st);
                GIST_TRY(gist_gemm_tn_f32(dy, lddy, l.Z, l.ldz, l.dW, 2 * l.n_in, l.n_out,
                                          2 * l.n_in, n, p->workspace, p->workspace_bytes, s));
            }
            GIST_TRY(gist_colsum_f32(dy, lddy, n, l.n_out, p->partials, l.db, s));
        }
        if (overlap) {
            const int64_t off = l.W - p->params;                 // [W_k | b_k] is contiguous
            const int64_t cnt = l.n_out * 2 * l.n_in + l.n_out;
            (void)hipEventRecord(ev_ready[k], st);
            (void)hipStreamWaitEvent(side, ev_ready[k], 0);
            GIST_TRY(gist_adam_f32(p->params + off, p->grads + off, p->exp_avg + off,
                                   p->exp_avg_sq + off, cnt, lr, beta1, beta2, eps, weight_decay,
                                   adam_step, (gist_stream_t)side));
        }
        if (k > 0) {
            if (drop)
                GIST_TRY(gist_dropout_f32(p->dZ, 2 * l.n_in, n, 2 * l.n_in, p->p_drop, p->seed,
                                          offs[k], s));
            {
                Scope sc(p->timer, 0, n, n, l.n_in, st);
                GIST_TRY(gist_spmm_csr_f32(p->t_rowptr, p->t_col, p->dZ + l.n_in, 2 * l.n_in, p->dZ,
                                           2 * l.n_in, n, l.n_in, nullptr, p->norm, 1, s));
            }
        }
    }
    if (overlap) {
        (void)hipEventRecord(ev_done, side);
        (void)hipStreamWaitEvent(st, ev_done, 0);
        return launch_status("gist_sage_step");
    }
    if (overlap_dw) {       // every dW / db must exist before Adam reads the gradient arena
        (void)hipEventRecord(ev_done, side);
        (void)hipStreamWaitEvent(st, ev_done, 0);
    }
    GIST_TRY(gist_adam_f32(p->params, p->grads, p->exp_avg, p->exp_avg_sq, p->n_params, lr, beta1,
                           beta2, eps, weight_decay, adam_step, s));
    return GIST_OK;
}
