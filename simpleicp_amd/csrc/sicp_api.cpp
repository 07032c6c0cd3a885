// sicp_api.cpp -- host side of libsimpleicp_hip.so: C ABI (include/simpleicp_hip.h), device
// memory, kernel sequencing of one ICP iteration, and the host 6x6 Levenberg-Marquardt solve
// that the fused GPU reductions feed.  No CPU compute fallback exists: without a gfx950 device
// sicp_ctx_create fails with SICP_ERR_NO_DEVICE.
#include "sicp_host.h"

namespace sicph {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace sicph

// error sink for the host-only translation units (sicp_io.cpp)
int sicp_io_fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
namespace sicph {

int sync(sicp_ctx *c)
{
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &p : c->pending) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, p.a, p.b));
        c->t_ms[p.kernel] += ms;
        c->t_n[p.kernel] += 1;
        c->pool.push_back(p);
    }
    c->pending.clear();
    return SICP_OK;
}

// non-blocking: fold in the event pairs whose kernels have finished (used on the polling path so
// that kernel timing does not add a stream synchronisation to every iteration)
void collect_ready(sicp_ctx *c)
{
    size_t w = 0;
    for (size_t i = 0; i < c->pending.size(); ++i) {
        EventPair p = c->pending[i];
        float ms = 0;
        if (hipEventQuery(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->t_ms[p.kernel] += ms; c->t_n[p.kernel] += 1; c->pool.push_back(p);
        } else {
            c->pending[w++] = p;
        }
    }
    c->pending.resize(w);
}

// Kernels that hand a few doubles to the host write them into pinned + mapped memory and then publish a
// sequence number there: polling that word sees the result a few microseconds before the end-of-kernel
// signal and spares a copy + hipStreamSynchronize per hand-over.  Falls back to a stream wait.
// An exchange that will never complete (a rank left the job, the ranks' collectives went out of step): give the communicator up,
// so that the collective -- and every chained kernel queued behind it on the ctx's stream -- ends instead of wedging each later
// hipStreamSynchronize (sicp_comm_destroy, sicp_ctx_destroy and sicp_icp_get_state all start with one).  ncclCommAbort makes the
// in-flight collective return; kernels behind it then run on garbage and finish.  Should the stream still not drain (a callback
// exchange: its collective is torch's, not ours to abort), the ctx moves to a fresh stream and the wedged one is left behind.

}  // namespace sicph

namespace sicph {

int check_slot(sicp_ctx *c, int slot, bool need_data)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (slot != SICP_FIX && slot != SICP_MOV) return fail(SICP_ERR_INVALID, "slot must be SICP_FIX or SICP_MOV");
    CHK(upload_join(c, slot));                            // (an upload still running behind the caller: wait, hand over its verdict)
    if (need_data && c->cloud[slot].n <= 0) return fail(SICP_ERR_INVALID, "cloud slot %d is empty", slot);
    return SICP_OK;
}

// rows handed over the ABI index device gathers: every one must be a row of the cloud (host-or-device pointer)
int check_rows(const int64_t *rows, int64_t m, int64_t n, const char *what)
{
    hipPointerAttribute_t at;
    std::vector<int64_t> tmp;
    if (hipPointerGetAttributes(&at, rows) == hipSuccess && at.type == hipMemoryTypeDevice) {
        tmp.resize((size_t)m);
        HIPCHK(hipMemcpy(tmp.data(), rows, (size_t)m * sizeof(int64_t), hipMemcpyDeviceToHost));
        rows = tmp.data();
    } else (void)hipGetLastError();                       // plain host memory is "invalid value" to the query: not an error
    for (int64_t i = 0; i < m; ++i)
        if (rows[i] < 0 || rows[i] >= n) return fail(SICP_ERR_INVALID, "%s[%lld] = %lld is not a row of the cloud (%lld points)", what,
                                                     (long long)i, (long long)rows[i], (long long)n);
    return SICP_OK;
}

}  // namespace sicph

namespace sicph {


double key_to_double(unsigned long long k)
{
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    double v; std::memcpy(&v, &b, sizeof v); return v;
}


}  // namespace sicph

// ------------------------------------------------------------------------------------------
SICP_EXPORT int sicp_abi_version(void) { return SICP_ABI_VERSION; }
SICP_EXPORT const char *sicp_last_error(void) { return g_err.c_str(); }

SICP_EXPORT int sicp_device_count(int *count_out)
{
    if (!count_out) return fail(SICP_ERR_INVALID, "count_out is null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count_out = n;
    return SICP_OK;
}

SICP_EXPORT int sicp_ctx_create(int device, sicp_ctx **ctx_out)
{
    if (!ctx_out) return fail(SICP_ERR_INVALID, "ctx_out is null");
    *ctx_out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SICP_ERR_NO_DEVICE, "no HIP device visible: libsimpleicp_hip has no CPU path");
    if (device < 0 || device >= n) return fail(SICP_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    sicp_ctx *c = new sicp_ctx();
    c->device = device;
    if (hipGetDeviceProperties(&c->prop, device) != hipSuccess) { delete c; return fail(SICP_ERR_HIP, "hipGetDeviceProperties failed"); }
    if (std::strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        const std::string arch = c->prop.gcnArchName;
        delete c;
        return fail(SICP_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, arch.c_str());
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(SICP_ERR_HIP, "hipStreamCreate failed"); }
    if (hipHostMalloc((void **)&c->h_small, 256 * sizeof(double), hipHostMallocMapped) != hipSuccess) { delete c; return fail(SICP_ERR_HIP, "hipHostMalloc failed"); }
    int rc = c->small.reserve(128);
    if (rc == SICP_OK) rc = c->ne_partial.reserve((size_t)NE_MAX_GRID * 64);
    if (rc == SICP_OK) rc = c->ticket.reserve(4);
    if (rc == SICP_OK) rc = c->icp_dev.reserve(1);
    if (rc == SICP_OK) rc = c->lm_dev.reserve(1);
    if (rc == SICP_OK && hipHostMalloc((void **)&c->h_lm, sizeof(LmDev), hipHostMallocDefault) != hipSuccess) rc = SICP_ERR_HIP;
    if (rc == SICP_OK) rc = c->match_work.reserve(8);
    if (rc == SICP_OK && hipMemsetAsync(c->match_work.p, 0, 8 * sizeof(unsigned long long), c->stream) != hipSuccess) rc = SICP_ERR_HIP;
    if (rc == SICP_OK && hipHostMalloc((void **)&c->h_rec, (size_t)REC_RING * REC_DOUBLES * sizeof(double), hipHostMallocMapped) != hipSuccess) rc = SICP_ERR_HIP;
    if (rc == SICP_OK && hipHostMalloc((void **)&c->h_state, sizeof(IcpDev), hipHostMallocDefault) != hipSuccess) rc = SICP_ERR_HIP;
    if (rc == SICP_OK) std::memset(c->h_rec, 0, (size_t)REC_RING * REC_DOUBLES * sizeof(double));
    if (rc == SICP_OK && hipMemsetAsync(c->ticket.p, 0, 4 * sizeof(unsigned), c->stream) != hipSuccess) rc = SICP_ERR_HIP;
    if (rc != SICP_OK) { sicp_ctx_destroy(c); return rc; }
    if (const char *e = std::getenv("SICP_ORDER_MIN_Q")) c->order_min_q = std::atol(e);
    if (const char *e = std::getenv("SICP_KNN_SWEEP")) c->knn_sweep = std::atoi(e) != 0;
    if (const char *e = std::getenv("SICP_KNN_BATCH")) c->knn_batch = std::atol(e);
    if (const char *e = std::getenv("SICP_KNN_GROUP")) c->knn_group = std::atoi(e);
    if (const char *e = std::getenv("SICP_NN16_MIN_Q")) c->nn16_min_q = std::atol(e);
    if (const char *e = std::getenv("SICP_NN16")) c->nn16_filter = !std::strcmp(e, "exact") ? 0 : !std::strcmp(e, "far") ? 1 : 2;
    if (const char *e = std::getenv("SICP_BOXES")) { c->use_boxes = std::atoi(e) != 0; c->boxes_always = std::atoi(e) >= 2; }
    if (const char *e = std::getenv("SICP_NN16F_MIN_Q")) { c->nn16f_min_q = std::atol(e); c->nn16f_min_q_forced = true; }
    if (const char *e = std::getenv("SICP_FAR_MOVE")) { const double t = std::atof(e); if (t >= 0) c->far_move = t; }
    if (const char *e = std::getenv("SICP_COARSE_MIN_N")) c->coarse_min_n = std::atol(e);
    if (const char *e = std::getenv("SICP_LM")) c->lm_one_launch = std::strcmp(e, "launches") != 0;
    if (const char *e = std::getenv("SICP_HSEL_WINDOW")) c->hsel_window = std::atoi(e) != 0;
    if (const char *e = std::getenv("SICP_TAIL_WINDOW")) c->tail_window = std::atoi(e) != 0;
    if (const char *e = std::getenv("SICP_TEST_BARRIER_FAULT")) c->test_barrier_fault = std::atoi(e);
    if (const char *e = std::getenv("SICP_NN_GROUP")) { const int v = std::atoi(e); if (v == 8 || v == 16) c->nn_group = v; }
    if (const char *e = std::getenv("SICP_XCHG_KEYS_MIN_Q")) c->xkeys_min_q = std::atol(e);
    if (const char *e = std::getenv("SICP_COMM_TIMEOUT_S")) { const double v = std::atof(e); if (v > 0) c->xchg_timeout_s = 2.0 * v; }    // (a record wait with collectives in flight: twice the rendezvous deadline)
    if (const char *e = std::getenv("SICP_FSCAN")) c->fscan_variant = !std::strcmp(e, "inline") ? 1 : 0;
    if (const char *e = std::getenv("SICP_FSCAN_CAP")) c->fscan_cap = std::atol(e);
    // SICP_SOLVE_TRACE: per-iteration traces on stderr -- any value: the tail's cycle counters; "host": the host's enqueue timings too;
    // "sel" / "eval": the fine splits of a -DSICP_SEL_FINE_TRACE / -DSICP_EVAL_FINE_TRACE build (build.build_variant)
    if (const char *e = std::getenv("SICP_SOLVE_TRACE")) {
        c->solve_trace = true;
        c->host_trace = std::strstr(e, "host") != nullptr;
        c->trace_sel = std::strstr(e, "sel") != nullptr; c->trace_eval = std::strstr(e, "eval") != nullptr;
    }
    if (const char *e = std::getenv("SICP_SOLVE")) c->solve_mode = !std::strcmp(e, "fused") ? 1 : !std::strcmp(e, "host") ? 2 : 0;
    if (const char *e = std::getenv("SICP_KNN1"))
        c->knn1_mode = !std::strcmp(e, "exact") ? 1 : !std::strcmp(e, "filter") ? 2 : !std::strcmp(e, "grid") ? 3 : 0;
    *ctx_out = c;
    return SICP_OK;
}

SICP_EXPORT int sicp_ctx_destroy(sicp_ctx *c)
{
    if (!c) return SICP_OK;
    (void)hipSetDevice(c->device);
    for (int s = 0; s < 2; ++s) (void)upload_join(c, s);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); c->copy_stream = nullptr; }
    if (c->h_bg) { (void)hipHostFree(c->h_bg); c->h_bg = nullptr; }
    c->stage_bg.release(); c->bg_small.release();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) { (void)rccl()->CommDestroy(c->comm); c->comm = nullptr; }
    if (c->h_dl) { (void)hipHostFree(c->h_dl); c->h_dl = nullptr; }
    for (auto &e : c->dl_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (auto &p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto &p : c->pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto &cl : c->cloud) { cl.xyz.release(); cl.pl.release(); cl.grid.cell_start.release(); cl.grid.rec.release();
                                 cl.sub_xyz.release(); cl.sub_grid.cell_start.release(); cl.sub_grid.rec.release(); }
    c->bound_p2.release(); c->bound_d2.release(); c->bound_idx.release(); c->q_order.release(); c->k_order.release(); c->k_sel.release(); c->k_nv.release(); c->k_pl.release(); c->k_cov.release(); c->k_redo.release(); c->g_ids.release(); c->g_counts.release(); c->g_cursor.release(); c->g_blk.release(); c->match_work.release(); c->rj_keys.release();
    c->stage.release(); c->part_d2.release(); c->part_idx.release(); c->kq.release(); c->k_d2.release();
    c->k_idx.release(); c->floor_d2.release(); c->floor_idx.release(); c->bound.release(); c->hit_cnt.release(); c->hit_list.release(); c->x_send.release(); c->x_recv.release(); c->q.release(); c->normals.release();
    c->planarity.release(); c->m_idx.release(); c->m_d2.release(); c->m_p2.release(); c->dist.release();
    c->resid.release(); c->flag.release(); c->keep.release(); c->small.release(); c->ne_partial.release();
    c->lm_bar_buf.release(); c->lm_gsum.release(); c->ticket.release(); c->icp_dev.release(); c->lm_dev.release(); c->resid2.release();
    c->corr_pl.release();
    if (c->h_lm) (void)hipHostFree(c->h_lm);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->h_rec) (void)hipHostFree(c->h_rec);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SICP_OK;
}

SICP_EXPORT int sicp_ctx_device_name(sicp_ctx *c, char *buf, int buflen)
{
    if (!c || !buf || buflen <= 0) return fail(SICP_ERR_INVALID, "bad arguments");
    std::snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", c->prop.name, c->prop.gcnArchName, c->prop.multiProcessorCount);
    return SICP_OK;
}

// ------------------------------------------------------------------------------------------
SICP_EXPORT int sicp_device_memory(sicp_ctx *c, int64_t *free_out, int64_t *total_out)
{
    if (!c || !free_out || !total_out) return fail(SICP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    *free_out = (int64_t)f; *total_out = (int64_t)t;
    return SICP_OK;
}

SICP_EXPORT int sicp_ctx_stream(sicp_ctx *c, void **stream_out)
{
    if (!c || !stream_out) return fail(SICP_ERR_INVALID, "null argument");
    *stream_out = (void *)c->stream;
    return SICP_OK;
}

SICP_EXPORT int sicp_timing_enable(sicp_ctx *c, int on)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    c->timing = on != 0;
    c->count_work = on == 2;
    return SICP_OK;
}
SICP_EXPORT int sicp_last_match_kernel(sicp_ctx *c, int *kind_out)
{
    if (!c || !kind_out) return fail(SICP_ERR_INVALID, "null argument");
    *kind_out = c->last_match_kernel;
    return SICP_OK;
}
SICP_EXPORT int sicp_match_work(sicp_ctx *c, uint64_t out3[3])
{
    if (!c || !out3) return fail(SICP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out3, c->match_work.p, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}
SICP_EXPORT int sicp_match_deferred(sicp_ctx *c, uint64_t *out)
{
    if (!c || !out) return fail(SICP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out, c->match_work.p + 3, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}
SICP_EXPORT int sicp_tail_cycles(sicp_ctx *c, double out5[5])
{
    if (!c || !out5) return fail(SICP_ERR_INVALID, "null argument");
    std::memcpy(out5, c->last_tail_cycles, sizeof c->last_tail_cycles);
    return SICP_OK;
}
SICP_EXPORT int sicp_tail_selection(sicp_ctx *c, int64_t out3[3])
{
    if (!c || !out3) return fail(SICP_ERR_INVALID, "null argument");
    out3[0] = c->last_sel_rounds[0]; out3[1] = c->last_sel_rounds[1]; out3[2] = c->sel_window_hits;
    return SICP_OK;
}
SICP_EXPORT int sicp_knn_work(sicp_ctx *c, uint64_t out4[4])
{
    if (!c || !out4) return fail(SICP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out4, c->match_work.p + 4, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}
SICP_EXPORT int sicp_timing_reset(sicp_ctx *c)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (!c->pending.empty()) CHK(sync(c));
    HIPCHK(hipMemsetAsync(c->match_work.p, 0, 8 * sizeof(unsigned long long), c->stream));
    for (int i = 0; i < SICP_K_COUNT; ++i) { c->t_ms[i] = 0; c->t_n[i] = 0; }
    return SICP_OK;
}
SICP_EXPORT int sicp_timing_get(sicp_ctx *c, int kernel, double *total_ms_out, int64_t *launches_out)
{
    if (!c || kernel < 0 || kernel >= SICP_K_COUNT) return fail(SICP_ERR_INVALID, "bad arguments");
    if (!c->pending.empty()) CHK(sync(c));          // fold in whatever is still in flight
    if (total_ms_out) *total_ms_out = c->t_ms[kernel];
    if (launches_out) *launches_out = c->t_n[kernel];
    return SICP_OK;
}

