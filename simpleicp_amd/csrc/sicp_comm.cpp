// sicp_comm.cpp -- multi-GPU plumbing: the RCCL entry points (loaded on demand), the per-iteration exchanges, communicator exports.
// Split from sicp_api.cpp (round 5).
#include "sicp_host.h"

namespace sicph {

Rccl &rccl_state() { static Rccl R; return R; }
Rccl *rccl()
{
    Rccl &R = rccl_state();
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (R.h) break;
            const char *e = dlerror();
            R.why += std::string(R.why.empty() ? "" : "; ") + (e ? e : "dlopen failed");
        }
        if (R.h) {
            R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(R.h, "ncclGetUniqueId");
            R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.h, "ncclCommInitRank");
            R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.h, "ncclCommDestroy");
            R.AllGather = (decltype(R.AllGather))dlsym(R.h, "ncclAllGather");
            R.AllReduce = (decltype(R.AllReduce))dlsym(R.h, "ncclAllReduce");
            R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.h, "ncclGetErrorString");
            R.CommCount = (decltype(R.CommCount))dlsym(R.h, "ncclCommCount");
            R.CommUserRank = (decltype(R.CommUserRank))dlsym(R.h, "ncclCommUserRank");
            R.CommAbort = (decltype(R.CommAbort))dlsym(R.h, "ncclCommAbort");
            if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather || !R.AllReduce || !R.GetErrorString ||
                !R.CommCount || !R.CommUserRank || !R.CommAbort) { R.h = nullptr; R.why = "librccl lacks an entry point this library needs"; }
        }
    }
    return R.h ? &R : nullptr;
}


}  // namespace sicph

namespace sicph {

void abandon_exchange(sicp_ctx *c)
{
    if (c->comm) { (void)rccl()->CommAbort(c->comm); c->comm = nullptr; }
    c->comm_active = false;
    c->xfn = nullptr; c->xuser = nullptr; c->xfn_u64 = false;
    c->rank = 0; c->world = 1; c->gn_shard = 0;
    const auto t0 = std::chrono::steady_clock::now();
    bool drained = false;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 5.0) {
        if (hipStreamQuery(c->stream) != hipErrorNotReady) { drained = true; break; }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    (void)hipGetLastError();
    if (!drained) {
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) == hipSuccess) c->stream = fresh;    // (the old one is leaked on purpose)
        c->pending.clear();                                   // their events sit on the abandoned stream
    }
    c->have_iter = false; c->have_corr = false; c->have_prev_match = false;
    c->slot_cnt = -1;
    c->hsel_dirty = true;                                     // whatever the interrupted launches left in the selection state
}

int wait_ticket(sicp_ctx *c, const double *flag_word, double seq)
{
    volatile const double *flag = flag_word;
    bool seen = false;
    for (long spin = 0; spin < 4000000L; ++spin) {
        if (*flag == seq) { seen = true; break; }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen && c->collective()) {
        // collectives are enqueued between the kernels: a rank that left the job (or a rank whose launches went out of step)
        // would leave this stream waiting forever -- give up with an error instead of hanging the process
        const auto t0 = std::chrono::steady_clock::now();
        while (*flag != seq) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return fail(SICP_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(q));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->xchg_timeout_s) {
                const int rank = c->rank, world = c->world;
                abandon_exchange(c);
                return fail(SICP_ERR_EXCHANGE, "no result after %.0f s with a multi-GPU exchange in flight (rank %d of %d): a rank left "
                                               "the job or the ranks' collectives are out of step (twice SICP_COMM_TIMEOUT_S); the "
                                               "communicator was aborted, the context is single-GPU again",
                            c->xchg_timeout_s, rank, world);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        seen = *flag == seq;
    }
    if (!seen) return sync(c);
    if (c->timing) collect_ready(c);
    return SICP_OK;
}


}  // namespace sicph

namespace sicph {

// job-wide winner per query: pack (d2, idx, xyz) records, all-gather through the host's callback
// (torch.distributed over RCCL), reduce lexicographically on the device -- one collective per call
// recv[world][count] <- every rank's send[count], enqueued in order on the library's stream
int all_gather_f64(sicp_ctx *c, double *send, double *recv, long count)
{
    if (c->comm && c->comm_active) {
        const ncclResult_t r = rccl()->AllGather(send, recv, (size_t)count, ncclDouble, c->comm, c->stream);
        if (r != ncclSuccess) return fail(SICP_ERR_EXCHANGE, "ncclAllGather failed: %s", rccl()->GetErrorString(r));
        return SICP_OK;
    }
    // no host wait: the callback enqueues the collective in order on this stream (or synchronises itself)
    if (c->xfn(c->xuser, SICP_XCHG_ALLGATHER_F64, send, recv, nullptr, count) != 0)
        return fail(SICP_ERR_EXCHANGE, "exchange callback (ALLGATHER_F64) failed");
    return SICP_OK;
}
int all_reduce_sum_f64(sicp_ctx *c, double *buf, long count)
{
    if (c->comm && c->comm_active) {
        const ncclResult_t r = rccl()->AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, c->comm, c->stream);
        if (r != ncclSuccess) return fail(SICP_ERR_EXCHANGE, "ncclAllReduce failed: %s", rccl()->GetErrorString(r));
        return SICP_OK;
    }
    if (c->xfn(c->xuser, SICP_XCHG_SUM_F64, buf, nullptr, nullptr, count) != 0)
        return fail(SICP_ERR_EXCHANGE, "exchange callback (SUM_F64) failed");
    return SICP_OK;
}

// cloud shards: job-wide winner per query = lexicographic minimum over the ranks' local winners
int exchange_best(sicp_ctx *c, double *d2, int64_t *idx, double *p2, long Q)
{
    if (!c->collective() || c->partition != SICP_PART_CLOUD) return SICP_OK;
    Timed t(c, SICP_K_XCHG);
    CHK(c->x_send.reserve((size_t)5 * Q));
    CHK(c->x_recv.reserve((size_t)5 * Q * c->world));
    launch_pack_best(c->stream, d2, idx, p2, Q, c->x_send.p);
    HIPCHK(hipGetLastError());
    CHK(all_gather_f64(c, c->x_send.p, c->x_recv.p, 5 * Q));
    launch_lexmin_gathered(c->stream, c->x_recv.p, c->world, Q, d2, idx, p2);
    HIPCHK(hipGetLastError());
    return SICP_OK;
}

// The same behind the match of a chained ICP iteration, in two launches fewer: the match kernel's winning lanes left the packed
// records themselves (PostMatch::pack), and ONE kernel takes the lexicographic minimum over the ranks and forms the
// point-to-plane distance + planarity verdict (k_postmatch's work) from it.
int exchange_best_chained(sicp_ctx *c, const TailArgs &A, long Q, bool packed_by_match)
{
    Timed t(c, SICP_K_XCHG);
    CHK(c->x_recv.reserve((size_t)5 * Q * c->world));
    if (!packed_by_match) {
        launch_pack_best(c->stream, c->m_d2.p, c->m_idx.p, c->m_p2.p, Q, c->x_send.p);
        HIPCHK(hipGetLastError());
    }
    CHK(all_gather_f64(c, c->x_send.p, c->x_recv.p, 5 * Q));
    launch_lexmin_postmatch(c->stream, c->x_recv.p, c->world, Q, c->q.p, c->q.p + c->qpad, c->q.p + 2 * c->qpad, c->normals.p,
                            c->planarity.p, A.min_planarity, A.pl2, A.pl2_n, c->icp_dev.p, c->m_d2.p, c->m_idx.p, c->m_p2.p,
                            c->dist.p, c->flag.p);
    HIPCHK(hipGetLastError());
    return SICP_OK;
}

// ... and for MANY queries (xkeys_min_q on; the library's own communicator, or a callback that serves SICP_XCHG_MIN_U64 / MAX_U64):
// three all-reduces on 8-byte keys instead of the all-gather of 40 bytes per query and rank (sicp_kernels.hip); then k_postmatch on
// the job-wide winners
bool exchange_by_keys(const sicp_ctx *c, long Q)
{
    const bool served = (c->comm && c->comm_active) || (c->xfn && c->xfn_u64);      // the library's communicator, or a callback that said it can
    return served && c->partition == SICP_PART_CLOUD && c->xkeys_min_q > 0 && Q >= c->xkeys_min_q;
}
// element-wise unsigned minimum / maximum over the ranks, in place, enqueued in order on the library's stream
int all_reduce_u64(sicp_ctx *c, unsigned long long *buf, long count, bool take_max)
{
    if (c->comm && c->comm_active) {
        const ncclResult_t r = rccl()->AllReduce(buf, buf, (size_t)count, ncclUint64, take_max ? ncclMax : ncclMin, c->comm, c->stream);
        if (r != ncclSuccess) return fail(SICP_ERR_EXCHANGE, "ncclAllReduce (keys) failed: %s", rccl()->GetErrorString(r));
        return SICP_OK;
    }
    if (c->xfn(c->xuser, take_max ? SICP_XCHG_MAX_U64 : SICP_XCHG_MIN_U64, buf, nullptr, nullptr, count) != 0)
        return fail(SICP_ERR_EXCHANGE, "exchange callback (%s) failed", take_max ? "MAX_U64" : "MIN_U64");
    return SICP_OK;
}
int exchange_best_keys_chained(sicp_ctx *c, const TailArgs &A, long Q)
{
    Timed t(c, SICP_K_XCHG);
    CHK(c->x_send.reserve((size_t)5 * Q));
    unsigned long long *gmin = (unsigned long long *)c->x_send.p, *gidx = gmin + Q, *xyz = gidx + Q;
    launch_xkey_d2(c->stream, c->m_d2.p, c->m_idx.p, Q, gmin);
    CHK(all_reduce_u64(c, gmin, Q, false));
    launch_xkey_idx(c->stream, c->m_d2.p, c->m_idx.p, gmin, Q, gidx);
    CHK(all_reduce_u64(c, gidx, Q, false));
    launch_xkey_xyz(c->stream, c->m_idx.p, c->m_p2.p, gidx, Q, xyz);
    CHK(all_reduce_u64(c, xyz, 3 * Q, true));
    launch_xkey_unpack(c->stream, gmin, gidx, xyz, Q, c->m_d2.p, c->m_idx.p, c->m_p2.p);
    Xf unused = {};
    launch_postmatch(c->stream, c->q.p, c->q.p + c->qpad, c->q.p + 2 * c->qpad, c->normals.p, c->planarity.p, c->m_p2.p, c->m_idx.p, Q,
                     unused, A.min_planarity, A.pl2, A.pl2_n, c->dist.p, c->flag.p, c->icp_dev.p);
    HIPCHK(hipGetLastError());
    return SICP_OK;
}

// query shards (cloud replicated): rank r matched queries [r * per, (r + 1) * per); the slices are gathered in rank
// order, which IS query order, so every rank ends up with all Q results
long query_slice(const sicp_ctx *c, long Q, long *lo)
{
    const long per = (Q + c->world - 1) / c->world;
    *lo = std::min<long>(Q, per * c->rank);
    return std::min<long>(Q, *lo + per) - *lo;
}
// ... gathered slim: 8 bytes per query (the matched index) instead of the 40-byte (d2, idx, xyz) record -- the cloud is replicated,
// so every rank looks the coordinates up itself and forms distance + verdict in the same pass (k_postmatch's work)
int exchange_query_slices_idx(sicp_ctx *c, const TailArgs &A, long Q, bool packed_by_match)
{
    const Cloud &cl = c->cloud[SICP_MOV];
    const long per = (Q + c->world - 1) / c->world;
    long lo; const long cnt = query_slice(c, Q, &lo);
    Timed t(c, SICP_K_XCHG);
    CHK(c->x_send.reserve((size_t)per));
    CHK(c->x_recv.reserve((size_t)per * c->world));
    if (packed_by_match) {
        // the match kernel's winning lanes wrote the slice's entries; the padding behind a short last slice reads "no match"
        if (per > cnt) HIPCHK(hipMemsetAsync(c->x_send.p + cnt, 0xff, (size_t)(per - cnt) * sizeof(double), c->stream));
    } else {
        launch_pack_idx(c->stream, c->m_idx.p + lo, cnt, per, c->x_send.p);
        HIPCHK(hipGetLastError());
    }
    CHK(all_gather_f64(c, c->x_send.p, c->x_recv.p, per));
    launch_unpack_idx_postmatch(c->stream, c->x_recv.p, Q, cl.x(), cl.y(), cl.z(), cl.idx_base, cl.n, c->q.p, c->q.p + c->qpad,
                                c->q.p + 2 * c->qpad, c->normals.p, c->planarity.p, A.min_planarity, A.pl2, A.pl2_n, c->icp_dev.p,
                                c->m_idx.p, c->m_p2.p, c->dist.p, c->flag.p);
    HIPCHK(hipGetLastError());
    return SICP_OK;
}


}  // namespace sicph

SICP_EXPORT int sicp_set_exchange(sicp_ctx *c, sicp_exchange_fn fn, void *user, int rank, int world, int gn_shard)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (world < 1 || rank < 0 || rank >= world) return fail(SICP_ERR_INVALID, "bad rank/world");
    if (world > 1 && !fn) return fail(SICP_ERR_INVALID, "world > 1 needs an exchange callback");
    c->comm_active = false;                                // a callback replaces the library's own communicator (which stays parked)
    c->xfn = fn; c->xuser = user; c->rank = rank; c->world = world; c->gn_shard = gn_shard ? 1 : 0;
    // ABI 6: does the callback reduce 8-byte keys?  Asked with no data (count = 0, a = NULL): nothing is exchanged, every rank's
    // callback answers for itself -- the same code on every rank, hence the same answer
    c->xfn_u64 = fn != nullptr && fn(user, SICP_XCHG_MIN_U64, nullptr, nullptr, nullptr, 0) == 0 &&
                 fn(user, SICP_XCHG_MAX_U64, nullptr, nullptr, nullptr, 0) == 0;
    return SICP_OK;
}

SICP_EXPORT int sicp_comm_unique_id(void *id128)
{
    if (!id128) return fail(SICP_ERR_INVALID, "null argument");
    Rccl *R = rccl();
    if (!R) return fail(SICP_ERR_EXCHANGE, "librccl could not be loaded: %s", rccl_state().why.c_str());
    ncclUniqueId id;
    const ncclResult_t r = R->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(SICP_ERR_EXCHANGE, "ncclGetUniqueId failed: %s", R->GetErrorString(r));
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof id);
    return SICP_OK;
}

SICP_EXPORT int sicp_comm_destroy(sicp_ctx *c)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)rccl()->CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->comm_active = false;
    if (!c->xfn) { c->rank = 0; c->world = 1; c->gn_shard = 0; }
    return SICP_OK;
}

namespace sicph {
// ncclCommInitRank is a rendezvous: it returns when EVERY rank has called it.  A rank that never does (it crashed, it took
// another code path) would leave the callers blocked for good, so the call runs on a helper thread and the caller waits for it
// with a deadline; on a timeout the helper stays behind (there is no handle to abort yet) and the caller reports an error --
// simpleicp_amd/dist.py then sends every rank to the torch.distributed callback exchange together.
struct CommInit {
    std::mutex m; std::condition_variable cv;
    bool done = false;
    ncclResult_t r = ncclSuccess;
    ncclComm_t comm = nullptr;
    bool abandoned = false;        // the caller's deadline passed: whoever gets a communicator now must give it up
};
}  // namespace sicph

SICP_EXPORT int sicp_comm_init(sicp_ctx *c, const void *id128, int rank, int world, int gn_shard)
{
    if (!c || !id128) return fail(SICP_ERR_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(SICP_ERR_INVALID, "bad rank/world");
    Rccl *R = rccl();
    if (!R) return fail(SICP_ERR_EXCHANGE, "librccl could not be loaded: %s", rccl_state().why.c_str());
    CHK(sicp_comm_destroy(c));
    HIPCHK(hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    double timeout_s = 60.0;
    if (const char *e = std::getenv("SICP_COMM_TIMEOUT_S")) { const double v = std::atof(e); if (v > 0) timeout_s = v; }
    auto job = std::make_shared<CommInit>();
    const int device = c->device;
    std::thread([job, R, id, rank, world, device] {
        (void)hipSetDevice(device);
        ncclComm_t comm = nullptr;
        const ncclResult_t r = R->CommInitRank(&comm, world, id, rank);
        std::lock_guard<std::mutex> g(job->m);
        job->r = r; job->comm = comm; job->done = true;
        if (job->abandoned && r == ncclSuccess && comm) { (void)R->CommAbort(comm); job->comm = nullptr; }    // nobody is waiting any more
        job->cv.notify_all();
    }).detach();
    {
        std::unique_lock<std::mutex> g(job->m);
        if (!job->cv.wait_for(g, std::chrono::duration<double>(timeout_s), [&] { return job->done; })) {
            job->abandoned = true;                                // (under the lock: the helper aborts what it gets, should it ever return)
            return fail(SICP_ERR_EXCHANGE, "ncclCommInitRank did not return within %.0f s (rank %d of %d): not every rank joined "
                                           "(SICP_COMM_TIMEOUT_S)", timeout_s, rank, world);
        }
        if (job->r != ncclSuccess) return fail(SICP_ERR_EXCHANGE, "ncclCommInitRank failed: %s", R->GetErrorString(job->r));
        c->comm = job->comm;
    }
    // what the communicator says about itself must be what the caller said (a mixed-up id would pair the wrong processes)
    int n = 0, r = -1;
    if (R->CommCount(c->comm, &n) != ncclSuccess || R->CommUserRank(c->comm, &r) != ncclSuccess || n != world || r != rank) {
        (void)R->CommAbort(c->comm); c->comm = nullptr;
        return fail(SICP_ERR_EXCHANGE, "RCCL communicator reports rank %d of %d, expected %d of %d", r, n, rank, world);
    }
    c->comm_rank = r; c->comm_world = n;
    // handshake: one small all-gather on the ctx's stream, awaited with a deadline -- the first collective is where a transport
    // problem (a link that does not come up, a peer that cannot be mapped) shows, and it must show here, not inside a run
    CHK(c->x_send.reserve(8)); CHK(c->x_recv.reserve((size_t)8 * world));
    std::vector<double> h((size_t)8 * world, -1.0);
    for (int j = 0; j < 8; ++j) h[j] = 1000.0 * rank + j;
    HIPCHK(hipMemcpyAsync(c->x_send.p, h.data(), 8 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->comm_active = true;
    int rc = all_gather_f64(c, c->x_send.p, c->x_recv.p, 8);
    if (rc == SICP_OK) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { rc = fail(SICP_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(q)); break; }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                rc = fail(SICP_ERR_EXCHANGE, "the first RCCL all-gather did not complete within %.0f s (rank %d of %d)", timeout_s, rank, world);
                break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    }
    if (rc == SICP_OK) {
        if (hipMemcpy(h.data(), c->x_recv.p, (size_t)8 * world * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(SICP_ERR_HIP, "hipMemcpy after the handshake failed");
        for (int k = 0; rc == SICP_OK && k < world; ++k)
            for (int j = 0; j < 8; ++j)
                if (h[(size_t)8 * k + j] != 1000.0 * k + j) { rc = fail(SICP_ERR_EXCHANGE, "RCCL handshake: slot %d holds %g, not rank %d's words", k, h[(size_t)8 * k + j], k); break; }
    }
    if (rc != SICP_OK) {
        (void)R->CommAbort(c->comm); c->comm = nullptr; c->comm_active = false;
        return rc;
    }
    c->xfn = nullptr; c->xuser = nullptr; c->xfn_u64 = false;
    c->rank = rank; c->world = world; c->gn_shard = gn_shard ? 1 : 0;
    return SICP_OK;
}

// The communicator stays with the ctx between runs; a run switches its use on, the end of the run off (a standalone
// PointCloud operator on the same ctx must not issue a collective the other ranks never join).
SICP_EXPORT int sicp_comm_activate(sicp_ctx *c, int on, int gn_shard)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (on && !c->comm) return fail(SICP_ERR_INVALID, "no communicator: call sicp_comm_init first");
    if (on) {
        c->xfn = nullptr; c->xuser = nullptr; c->xfn_u64 = false;
        c->comm_active = true; c->rank = c->comm_rank; c->world = c->comm_world; c->gn_shard = gn_shard ? 1 : 0;
    } else {
        c->comm_active = false;
        if (!c->xfn) { c->rank = 0; c->world = 1; c->gn_shard = 0; }
    }
    return SICP_OK;
}

SICP_EXPORT int sicp_comm_info(sicp_ctx *c, int out[6])
{
    if (!c || !out) return fail(SICP_ERR_INVALID, "null argument");
    out[0] = c->xfn ? 1 : (c->comm && c->comm_active) ? 2 : 0;          // 0 none, 1 host callback, 2 the library's RCCL communicator
    out[1] = c->world; out[2] = c->rank; out[3] = c->partition; out[4] = c->gn_shard;
    out[5] = c->comm ? 1 : 0;                                            // a communicator exists (active or parked)
    if (out[0] == 2) {
        // as RCCL itself counts them, not as the caller declared them
        int n = 0, r = -1;
        if (rccl()->CommCount(c->comm, &n) != ncclSuccess || rccl()->CommUserRank(c->comm, &r) != ncclSuccess)
            return fail(SICP_ERR_EXCHANGE, "ncclCommCount / ncclCommUserRank failed");
        out[1] = n; out[2] = r;
    }
    return SICP_OK;
}

SICP_EXPORT int sicp_exchange_info(sicp_ctx *c, int64_t out4[4])
{
    if (!c || !out4) return fail(SICP_ERR_INVALID, "null argument");
    out4[0] = c->last_xchg_form; out4[1] = c->xchg_count; out4[2] = c->xkeys_min_q; out4[3] = (c->xfn && c->xfn_u64) ? 1 : 0;
    return SICP_OK;
}

SICP_EXPORT int sicp_set_partition(sicp_ctx *c, int mode)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (mode != SICP_PART_CLOUD && mode != SICP_PART_QUERIES) return fail(SICP_ERR_INVALID, "mode must be SICP_PART_CLOUD or SICP_PART_QUERIES");
    c->partition = mode;
    return SICP_OK;
}

SICP_EXPORT int sicp_lexmin_gathered(sicp_ctx *c, const double *gathered, int world, int64_t Q, double *d2_out,
                                     int64_t *idx_out, double *xyz_out)
{
    if (!c || !gathered || !d2_out || !idx_out || world < 1 || Q < 1) return fail(SICP_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    DevBuf<double> g, d2, xyz; DevBuf<int64_t> idx;
    int rc = g.reserve((size_t)5 * Q * world);
    if (rc == SICP_OK) rc = d2.reserve(Q);
    if (rc == SICP_OK) rc = xyz.reserve((size_t)3 * Q);
    if (rc == SICP_OK) rc = idx.reserve(Q);
    auto body = [&]() -> int {
        HIPCHK(hipMemcpyAsync(g.p, gathered, (size_t)5 * Q * world * sizeof(double), hipMemcpyDefault, c->stream));
        launch_lexmin_gathered(c->stream, g.p, world, Q, d2.p, idx.p, xyz.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(d2_out, d2.p, (size_t)Q * sizeof(double), hipMemcpyDefault, c->stream));
        HIPCHK(hipMemcpyAsync(idx_out, idx.p, (size_t)Q * sizeof(int64_t), hipMemcpyDefault, c->stream));
        if (xyz_out) HIPCHK(hipMemcpyAsync(xyz_out, xyz.p, (size_t)3 * Q * sizeof(double), hipMemcpyDefault, c->stream));
        return sync(c);
    };
    if (rc == SICP_OK) rc = body();
    (void)hipStreamSynchronize(c->stream);
    g.release(); d2.release(); xyz.release(); idx.release();
    return rc;
}

