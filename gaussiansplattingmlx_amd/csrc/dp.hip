// dp.hip -- the data-parallel step behind the C ABI (SURVEY row e; BASELINE north-star: "camera views shard one-per-rank
// across the 8 GPUs of one node with an RCCL all-reduce of parameter gradients over xGMI").
//
// The reference is single-device (GaussianTrainer.swift:486-498 trains batch-1), so there is no reference call site to
// replace: the contract is the north-star's.  One process per GPU; every rank renders its own view with the ordinary
// gs_render_forward (no data-path collective), then calls gs_dp_step, which runs the backward, exchanges the gradients
// with RCCL collectives the LIBRARY issues on a side stream of its own, and applies Adam with grad_scale = 1 / world --
// the same update on every rank, so the replicas stay bit-identical without a broadcast.  A Swift / C++ host needs no
// RCCL binding of its own: gs_dp_unique_id + gs_dp_init are the bootstrap (or gs_dp_attach borrows a communicator).
//
// RCCL is loaded with dlopen on the first gs_dp_* call: single-GPU hosts never map it (the library is 570 MB), and in a
// process that already has it (PyTorch ships a copy with the same SONAME) the loader hands back that copy.
//
// Stream discipline of a step (mode GS_DP_SH_COMPRESSED; one side stream, so the collectives of a communicator are
// issued in one order on every rank).  Round 6:
//
//   ctx stream                                          side stream (RCCL)
//   blend backward
//   ONE kernel: colour cotangents + this rank's
//   overflow word behind them + the four geometry
//   gradients WITHOUT the SH rows (the xyz gradient
//   lacks its view-direction term; a copy of it
//   stays behind for the densify statistic)    --ev-->  all-gather colorCot + word  (12 B / Gaussian / rank + 16 B)
//                                                       all-reduce(sum) geometry    (44 B / Gaussian)
//   <--ev-- gather done
//   SH rows read ONCE: the view-direction terms of all views rebuilt from the gathered cotangents (their sum to xyzAdd,
//   this rank's own into the densify statistic), the SH gradients rebuilt + their Adam step; ORs the R gathered words
//   = the step's gate, tests it, leaves it in the gate word
//   <--ev-- reduce done
//   Adam on the geometry slice, xyz gradient = reduced + xyzAdd (tests the gate word)
//
// so the geometry all-reduce runs under the SH kernel (~0.1 ms); the all-gather has nothing left to hide under -- rounds 2-5
// forked it between a colour-cotangent kernel and a geometry backward that staged every SH row a second time (46 us), which
// cost more (two launches, a fork, 288 B per Gaussian read twice) than the overlap gave.  TWO collectives per step since
// round 5: rounds 3-4 max-reduced the overflow words in a 4-byte all-reduce of their own in front of the all-gather -- a
// full RCCL launch and ring latency per step for four bytes; now every rank's word rides behind its colour cotangents
// (GS_DP_ALLREDUCE: behind the gradient arena, summed -- any value > 0 gates).
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>

#include <rccl/rccl.h>      // types and prototypes only: every function is resolved with dlsym below

#include "gs_ctx.h"

namespace {

// GSPLAT_DP_HOST_TRACE=1: host time spent in each phase of gs_dp_step, printed by gs_dp_shutdown (diagnostic: which call
// keeps the host from running ahead of the device)
struct HostTrace {
    bool on = getenv("GSPLAT_DP_HOST_TRACE") != nullptr;
    double sum[12] = {0};
    long n = 0;
    timespec t0;
    void begin() { if (on) clock_gettime(CLOCK_MONOTONIC, &t0); }
    void lap(int i)
    {
        if (!on) return;
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        sum[i] += (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
        t0 = t1;
    }
    void report()
    {
        if (!on || !n) return;
        static const char* names[12] = {"(unused)", "(unused)", "(unused)", "blend backward / colour cot", "fork cc",
                                        "all-gather", "projection backward", "fork geom", "all-reduce", "wait gather + SH rebuild",
                                        "wait reduce", "adam"};
        for (int i = 0; i < 12; i++) fprintf(stderr, "gs_dp_step host  %-28s %8.1f us/step\n", names[i], sum[i] / n);
    }
};
HostTrace g_trace;

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;      // optional
    decltype(&ncclCommSplit) CommSplit = nullptr;        // optional (NCCL >= 2.18): a second communicator for the inline all-gather
    std::string err;
};

Rccl* rccl_load()
{
    static Rccl r;
    if (r.handle) return &r;
    const char* names[] = {getenv("GSPLAT_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
        r.err = dlerror();
    }
    if (!r.handle) return &r;
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(r.handle, name); if (!p) { ok = false; r.err = std::string("missing RCCL symbol ") + name; } return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(r.handle); r.handle = nullptr; return &r; }
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.handle, "ncclGetVersion"));
    r.CommSplit = reinterpret_cast<decltype(r.CommSplit)>(dlsym(r.handle, "ncclCommSplit"));
    return &r;
}

// gs_dp_check_replicas: (sum, sum of magnitudes) of the arena in f64, in a FIXED order -- 256 blocks of 256 threads, every
// thread its own strided subsequence, a block's threads and then the blocks summed in index order -- so that identical
// replicas give identical bits
constexpr int GS_REPL_BLOCKS = 256, GS_REPL_THREADS = 256;
__global__ __launch_bounds__(GS_REPL_THREADS) void replica_partial_kernel(const float* __restrict__ a, long long n, double* __restrict__ part)
{
    __shared__ double s0[GS_REPL_THREADS], s1[GS_REPL_THREADS];
    double x = 0.0, y = 0.0;
    for (long long i = (long long)blockIdx.x * GS_REPL_THREADS + threadIdx.x; i < n; i += (long long)GS_REPL_BLOCKS * GS_REPL_THREADS) {
        const double v = (double)a[i];
        x += v; y += fabs(v);
    }
    s0[threadIdx.x] = x; s1[threadIdx.x] = y;
    __syncthreads();
    for (int w = GS_REPL_THREADS / 2; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) { s0[threadIdx.x] += s0[threadIdx.x + w]; s1[threadIdx.x] += s1[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s0[0]; part[2 * blockIdx.x + 1] = s1[0]; }
}
// out[0..2] = (N, sum, sum of magnitudes), out[3..5] = their negatives: ONE max all-reduce gives the maxima and the minima
__global__ void replica_final_kernel(const double* __restrict__ part, double nGauss, double* __restrict__ out)
{
    double x = 0.0, y = 0.0;
    for (int b = 0; b < GS_REPL_BLOCKS; b++) { x += part[2 * b]; y += part[2 * b + 1]; }
    out[0] = nGauss; out[1] = x; out[2] = y; out[3] = -nGauss; out[4] = -x; out[5] = -y;
}

int comm_fail(gs_ctx* c, const char* what, const std::string& why)
{
    c->err = std::string(what) + ": " + why;
    return GS_ERR_COMM;
}

}  // namespace

struct GsDp {
    Rccl* lib = nullptr;
    ncclComm_t comm = nullptr;
    bool ownComm = false;
    ncclComm_t commGather = nullptr;            // ncclCommSplit of comm (same ranks): the step's all-gather runs on it ON THE CTX STREAM
                                                // while the geometry all-reduce runs on `comm` on the side stream -- two collectives
                                                // in flight at once need two communicators; nullptr: both on the side stream
    int rank = 0, world = 1;
    hipStream_t sComm = nullptr;
    hipEvent_t evFlag = nullptr, evGate = nullptr, evCc = nullptr, evGather = nullptr, evGeom = nullptr, evReduce = nullptr;
    uint32_t* words = nullptr;                  // device: [0] gate of the current step (max over ranks of the overflow words),
                                                //         [1] "some step since the last gs_dp_check_overflow was gated"
    unsigned long long* need = nullptr;         // device: pair count to agree on (gs_dp_check_overflow)
    unsigned long long* hostWords = nullptr;    // pinned: [0] need, [1] seen
    double* repl = nullptr;                     // device: [2 * GS_REPL_BLOCKS] partial sums + [6] the words gs_dp_check_replicas reduces
    double* planDev = nullptr;                  // device: [16] the words gs_dp_check_plan reduces (w, -w)
    double* planHost = nullptr;                 // pinned: [16] this rank's words, [16] the reduced ones
    float* xyzOwn = nullptr;                    // device [xyzCap]: this view's xyz gradient without the view-direction term
    float* xyzAdd = nullptr;                    // device [xyzCap]: sum over the step's views of that term (+ a zero pad)
    long long xyzCap = 0;
    hipEvent_t evRepl = nullptr;                // behind the D2H copy of a gs_dp_check_replicas_begin
    bool replPending = false;
    // exchange timing (gs_dp_exchange_timing; measurement only): per step, the events around every collective on the side
    // stream and around the ctx stream's waits for them
    bool timing = false;
    struct StepEvents { hipEvent_t e[10]; bool used[10]; };
    std::vector<StepEvents> timed;
    size_t timedUsed = 0;
};

// slots of a step's timing events
enum { XE_GATE0 = 0, XE_GATE1, XE_GATHER0, XE_GATHER1, XE_REDUCE0, XE_REDUCE1, XE_WAIT_GATHER0, XE_WAIT_GATHER1, XE_WAIT_REDUCE0,
       XE_WAIT_REDUCE1 };
constexpr size_t GS_DP_TIMED_STEPS_MAX = 512;

#define GS_NCCL_CHECK(ctx, d, expr)                                                             \
    do {                                                                                        \
        ncclResult_t _r = (expr);                                                               \
        if (_r != ncclSuccess) return comm_fail((ctx), #expr, (d)->lib->GetErrorString(_r));    \
    } while (0)

namespace {

int dp_create(gs_ctx* c, ncclComm_t comm, bool own, int rank, int world, Rccl* lib)
{
    GsDp* d = new GsDp();
    d->lib = lib; d->comm = comm; d->ownComm = own; d->rank = rank; d->world = world;
    c->dp = d;
    // Round 6: the all-gather is on the step's critical path at every world size (the SH kernel needs the gathered cotangents
    // and nothing else is left to run), so the two cross-stream hops of a side-stream launch (~25 us of its 34 on one rank:
    // event record -> side stream wakes -> RCCL kernel -> event -> ctx stream wakes) are pure loss: it is issued on the ctx
    // stream itself.  The geometry all-reduce keeps the side stream (it hides under the SH kernel), and two collectives in
    // flight at once need two communicators: a split of the same ranks (collective over `comm`; GSPLAT_DP_INLINE_GATHER=0 or a
    // library without ncclCommSplit: both collectives on the side stream as in rounds 3-5).
    const char* inl = getenv("GSPLAT_DP_INLINE_GATHER");
    if (lib->CommSplit && !(inl && inl[0] == '0')) {
        ncclComm_t g = nullptr;
        const ncclResult_t sr = lib->CommSplit(comm, 0, rank, &g, nullptr);
        if (sr == ncclSuccess) d->commGather = g;
    }
    GS_HIP_CHECK(c, hipStreamCreateWithFlags(&d->sComm, hipStreamNonBlocking));
    hipEvent_t* evs[] = {&d->evFlag, &d->evGate, &d->evCc, &d->evGather, &d->evGeom, &d->evReduce};
    for (hipEvent_t* e : evs) GS_HIP_CHECK(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    GS_HIP_CHECK(c, hipMalloc((void**)&d->words, 4 * sizeof(uint32_t)));
    GS_HIP_CHECK(c, hipMemset(d->words, 0, 4 * sizeof(uint32_t)));
    GS_HIP_CHECK(c, hipMalloc((void**)&d->need, 2 * sizeof(unsigned long long)));
    GS_HIP_CHECK(c, hipHostMalloc((void**)&d->hostWords, 16 * sizeof(unsigned long long)));
    GS_HIP_CHECK(c, hipMalloc((void**)&d->repl, (2 * GS_REPL_BLOCKS + 8) * sizeof(double)));
    GS_HIP_CHECK(c, hipMalloc((void**)&d->planDev, 16 * sizeof(double)));
    GS_HIP_CHECK(c, hipHostMalloc((void**)&d->planHost, 32 * sizeof(double)));
    GS_HIP_CHECK(c, hipEventCreateWithFlags(&d->evRepl, hipEventDisableTiming));
    c->adamGate = d->words;         // from now on every optimizer kernel of the ctx tests the step's COMMON word
    c->gateSeen = d->words + 1;     // ... and one that finds it raised says so (gs_dp_check_overflow)
    return GS_OK;
}

// exchange timing: this step's event set (nullptr: timing off, or the pool is full -- the step then goes untimed)
GsDp::StepEvents* timed_step(GsDp* d)
{
    if (!d->timing) return nullptr;
    if (d->timedUsed == d->timed.size()) {
        if (d->timed.size() >= GS_DP_TIMED_STEPS_MAX) return nullptr;
        GsDp::StepEvents s;
        for (int i = 0; i < 10; i++)
            if (hipEventCreate(&s.e[i]) != hipSuccess) { for (int k = 0; k < i; k++) (void)hipEventDestroy(s.e[k]); return nullptr; }
        d->timed.push_back(s);
    }
    GsDp::StepEvents* s = &d->timed[d->timedUsed++];
    for (bool& u : s->used) u = false;
    return s;
}
void mark(GsDp::StepEvents* s, int slot, hipStream_t st)
{
    if (s && hipEventRecord(s->e[slot], st) == hipSuccess) s->used[slot] = true;
}

// the side stream picks up after everything queued on the ctx stream so far
int fork_after(gs_ctx* c, GsDp* d, hipEvent_t ev)
{
    GS_HIP_CHECK(c, hipEventRecord(ev, c->stream));
    GS_HIP_CHECK(c, hipStreamWaitEvent(d->sComm, ev, 0));
    return GS_OK;
}
// ... and the ctx stream waits for what the side stream has been given so far
int join_before(gs_ctx* c, GsDp* d, hipEvent_t ev)
{
    GS_HIP_CHECK(c, hipEventRecord(ev, d->sComm));
    GS_HIP_CHECK(c, hipStreamWaitEvent(c->stream, ev, 0));
    return GS_OK;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int gs_dp_unique_id(void* id)
{
    if (!id) return GS_ERR_INVALID_ARG;
    Rccl* lib = rccl_load();
    if (!lib->handle) return GS_ERR_COMM;
    ncclUniqueId u;
    if (lib->GetUniqueId(&u) != ncclSuccess) return GS_ERR_COMM;
    static_assert(sizeof(u) == GS_DP_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof u);
    return GS_OK;
}

int gs_dp_init(gs_ctx* c, const void* id, int rank, int world)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!id || world < 1 || world > 16 || rank < 0 || rank >= world) { c->err = "gs_dp_init: bad id / rank / world (at most 16 ranks)"; return GS_ERR_INVALID_ARG; }
    if (c->dp) { c->err = "gs_dp_init: the ctx already has a communicator (gs_dp_shutdown first)"; return GS_ERR_INVALID_ARG; }
    Rccl* lib = rccl_load();
    if (!lib->handle) return comm_fail(c, "gs_dp_init: RCCL not loadable", lib->err);
    GS_HIP_CHECK(c, hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = lib->CommInitRank(&comm, world, u, rank);
    if (r != ncclSuccess) return comm_fail(c, "ncclCommInitRank", lib->GetErrorString(r));
    const int rc = dp_create(c, comm, true, rank, world, lib);
    if (rc) (void)gs_dp_shutdown(c);
    return rc;
}

int gs_dp_attach(gs_ctx* c, void* nccl_comm, int rank, int world)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!nccl_comm || world < 1 || world > 16 || rank < 0 || rank >= world) { c->err = "gs_dp_attach: bad communicator / rank / world (at most 16 ranks)"; return GS_ERR_INVALID_ARG; }
    if (c->dp) { c->err = "gs_dp_attach: the ctx already has a communicator (gs_dp_shutdown first)"; return GS_ERR_INVALID_ARG; }
    Rccl* lib = rccl_load();
    if (!lib->handle) return comm_fail(c, "gs_dp_attach: RCCL not loadable", lib->err);
    int n = 0, me = -1;
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    if (lib->CommCount(comm, &n) != ncclSuccess || lib->CommUserRank(comm, &me) != ncclSuccess || n != world || me != rank) {
        c->err = "gs_dp_attach: rank / world do not match the communicator";
        return GS_ERR_SIZE_MISMATCH;
    }
    GS_HIP_CHECK(c, hipSetDevice(c->device));
    const int rc = dp_create(c, comm, false, rank, world, lib);
    if (rc) (void)gs_dp_shutdown(c);
    return rc;
}

int gs_dp_shutdown(gs_ctx* c)
{
    if (!c) return GS_ERR_INVALID_ARG;
    GsDp* d = c->dp;
    if (!d) return GS_OK;
    g_trace.report();
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (d->sComm) (void)hipStreamSynchronize(d->sComm);
    if (c->adamGate == d->words) c->adamGate = c->counters + GS_CNT_OVERFLOW;
    if (c->gateSeen == d->words + 1) c->gateSeen = nullptr;
    if (d->commGather) (void)d->lib->CommDestroy(d->commGather);
    if (d->ownComm && d->comm) (void)d->lib->CommDestroy(d->comm);
    hipEvent_t evs[] = {d->evFlag, d->evGate, d->evCc, d->evGather, d->evGeom, d->evReduce};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    for (auto& st : d->timed) for (hipEvent_t e : st.e) (void)hipEventDestroy(e);
    if (d->sComm) (void)hipStreamDestroy(d->sComm);
    if (d->words) (void)hipFree(d->words);
    if (d->need) (void)hipFree(d->need);
    if (d->repl) (void)hipFree(d->repl);
    if (d->planDev) (void)hipFree(d->planDev);
    if (d->xyzOwn) (void)hipFree(d->xyzOwn);
    if (d->xyzAdd) (void)hipFree(d->xyzAdd);
    if (d->planHost) (void)hipHostFree(d->planHost);
    if (d->evRepl) (void)hipEventDestroy(d->evRepl);
    if (d->hostWords) (void)hipHostFree(d->hostWords);
    delete d;
    c->dp = nullptr;
    return GS_OK;
}

int gs_dp_info(gs_ctx* c, int* rank, int* world)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (rank) *rank = c->dp ? c->dp->rank : 0;
    if (world) *world = c->dp ? c->dp->world : 0;
    return GS_OK;
}

int gs_dp_allreduce_sum(gs_ctx* c, float* buf, long long n)
{
    if (!c) return GS_ERR_INVALID_ARG;
    GsDp* d = c->dp;
    if (!d) { c->err = "gs_dp_allreduce_sum: no communicator (gs_dp_init / gs_dp_attach)"; return GS_ERR_INVALID_ARG; }
    if (n < 0 || (n > 0 && !buf)) { c->err = "gs_dp_allreduce_sum: bad buffer"; return GS_ERR_INVALID_ARG; }
    if (n == 0) return GS_OK;
    int rc;
    if ((rc = fork_after(c, d, d->evGeom))) return rc;
    GS_NCCL_CHECK(c, d, d->lib->AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, d->comm, d->sComm));
    return join_before(c, d, d->evReduce);
}

int gs_dp_step(gs_ctx* c, int mode, const gs_dp_step_args* a)
{
    if (!c) return GS_ERR_INVALID_ARG;
    GsDp* d = c->dp;
    if (!d) { c->err = "gs_dp_step: no communicator (gs_dp_init / gs_dp_attach)"; return GS_ERR_INVALID_ARG; }
    if (!a || (mode != GS_DP_ALLREDUCE && mode != GS_DP_SH_COMPRESSED)) { c->err = "gs_dp_step: bad mode / arguments"; return GS_ERR_INVALID_ARG; }
    if (!c->fwd.valid || c->fwd.consumed) { c->err = "gs_dp_step: no gs_render_forward on this context"; return GS_ERR_NO_FORWARD; }
    const int N = c->fwd.N, K = c->fwd.K;
    if (!a->cot_color || a->n_arena < 0 || a->nseg < 1 || a->nseg > 8 ||
        (N > 0 && (!a->params_base || !a->grads_base || !a->m_base || !a->v_base))) {
        c->err = "gs_dp_step: null buffer / bad segments";
        return GS_ERR_INVALID_ARG;
    }
    if (mode == GS_DP_SH_COMPRESSED && (!a->cam_centers || (N > 0 && (!a->color_cot_local || !a->color_cot_all)) ||
                                        a->geom_numel < 0 || a->geom_numel > a->n_arena)) {
        c->err = "gs_dp_step: the sh_compressed exchange needs cam_centers, color_cot_local / _all and geom_numel";
        return GS_ERR_INVALID_ARG;
    }
    // the six tensors of the forward lie in the parameter arena; gradients and moments share its layout
    const float* lo = a->params_base;
    const float* hi = a->params_base + a->n_arena;
    auto inside = [&](const float* p, long long n) { return n == 0 || (p >= lo && p + n <= hi); };
    if (!inside(c->fwd.xyz, 3LL * N) || !inside(c->fwd.fdc, 3LL * N) || !inside(c->fwd.frest, 3LL * (K - 1) * N) ||
        !inside(c->fwd.scales, 3LL * N) || !inside(c->fwd.rot, 4LL * N) || !inside(c->fwd.opacity, N)) {
        c->err = "gs_dp_step: the forward's tensors do not lie in the arena";
        return GS_ERR_SIZE_MISMATCH;
    }
    auto grad_of = [&](const float* p) { return p ? a->grads_base + (p - a->params_base) : nullptr; };
    auto lr_at = [&](const float* p) {
        const long long off = p - a->params_base;
        for (int i = 0; i < a->nseg; i++) if (off < a->seg_end[i]) return a->seg_lr[i];
        return a->seg_lr[a->nseg - 1];
    };
    if (mode == GS_DP_SH_COMPRESSED) {
        // the geometry slice leads the arena and holds exactly the four geometry tensors' segments
        const long long g = a->geom_numel;
        auto in_geom = [&](const float* p, long long n) { return n == 0 || (p >= lo && p + n <= lo + g); };
        bool segOk = false;
        for (int i = 0; i < a->nseg; i++) segOk = segOk || a->seg_end[i] == g;
        if (!segOk || !in_geom(c->fwd.xyz, 3LL * N) || !in_geom(c->fwd.scales, 3LL * N) || !in_geom(c->fwd.rot, 4LL * N) ||
            !in_geom(c->fwd.opacity, N) || (N > 0 && (c->fwd.fdc < lo + g || (K > 1 && c->fwd.frest < lo + g)))) {
            c->err = "gs_dp_step: geom_numel must end a segment, hold xyz / scales / rotation / opacity and none of the SH tensors";
            return GS_ERR_SIZE_MISMATCH;
        }
    }
    const float scale = 1.0f / (float)d->world;
    // No rank may leave the step half-way: the others would wait in a collective.  The deferred host-side overflow
    // error is therefore off for the duration (the gate below skips the update on every rank instead; the ranks look
    // together in gs_dp_check_overflow).
    const int hostErrors = c->hostOverflowErrors;
    c->hostOverflowErrors = 0;
    struct Restore { gs_ctx* c; int v; ~Restore() { c->hostOverflowErrors = v; } } restore{c, hostErrors};
    int rc;
    GsDp::StepEvents* xt = timed_step(d);
    g_trace.begin(); g_trace.n++;
    // The step's gate -- the OR over the ranks of the forwards' overflow words -- rides in the step's first payload (round 5;
    // rounds 3-4 max-reduced it in a 4-byte all-reduce of its own: a third RCCL launch and ring latency per step).  The
    // backward's first kernel stores this rank's word as 0.0f / 1.0f behind what the collective carries anyway
    // (c->overflowRider); the counter it reads is written by the forward's binning only, and the next forward, which
    // clears it, is queued behind this step's Adam.
    struct Rider { gs_ctx* c; float* was; long long bf; int bc; uint32_t* go; const uint32_t* gate;
                   ~Rider() { c->overflowRider = was; c->ccBlockFloats = bf; c->ccBlockCount = bc; c->gatheredGateOut = go; c->adamGate = gate; } }
        rider{c, c->overflowRider, c->ccBlockFloats, c->ccBlockCount, c->gatheredGateOut, c->adamGate};
    if (N == 0) return GS_OK;      // (every rank holds the same N: none of them enters a collective)
    if (mode == GS_DP_ALLREDUCE) {
        // the word behind the gradient arena: summed with it, > 0 on every rank iff some rank's forward overflowed -- and
        // +0.0f is the all-zero word, so the sum itself is the word the optimizer kernel tests
        c->overflowRider = a->grads_base + a->n_arena;
        if ((rc = gs_render_backward(c, a->cot_color, a->cot_depth, a->cot_alpha, grad_of(c->fwd.xyz), grad_of(c->fwd.fdc),
                                     grad_of(c->fwd.frest), grad_of(c->fwd.scales), grad_of(c->fwd.rot),
                                     grad_of(c->fwd.opacity))))
            return rc;
        if ((rc = fork_after(c, d, d->evGeom))) return rc;
        mark(xt, XE_REDUCE0, d->sComm);
        GS_NCCL_CHECK(c, d, d->lib->AllReduce(a->grads_base, a->grads_base, (size_t)a->n_arena + 1, ncclFloat, ncclSum, d->comm, d->sComm));
        mark(xt, XE_REDUCE1, d->sComm);
        mark(xt, XE_WAIT_REDUCE0, c->stream);
        if ((rc = join_before(c, d, d->evReduce))) return rc;
        mark(xt, XE_WAIT_REDUCE1, c->stream);
        c->adamGate = reinterpret_cast<const uint32_t*>(a->grads_base + a->n_arena);
        return gs_adam_step(c, a->n_arena, a->params_base, a->grads_base, a->m_base, a->v_base, a->nseg, a->seg_end, a->seg_lr,
                            a->beta1, a->beta2, a->eps, scale);
    }
    // sh_compressed: a rank's gather block = its [N,3] cotangents, then its word, padded to four floats
    const long long ccFloats = gs_dp_cc_floats(N);
    c->overflowRider = a->color_cot_local + 3LL * N;
    // round 6: the geometry gradients without the SH rows (projection.hip, "the SH rows are read ONCE per step"): the xyz
    // gradient's view-direction term is rebuilt for all views by the SH kernel below, which has the rows in hand
    const long long xyzFloats = (3LL * N + 3) & ~3LL;
    if (xyzFloats > d->xyzCap) {
        GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));       // (kernels of an earlier step may still read the old buffers)
        if (d->xyzOwn) GS_HIP_CHECK(c, hipFree(d->xyzOwn));
        if (d->xyzAdd) GS_HIP_CHECK(c, hipFree(d->xyzAdd));
        d->xyzOwn = d->xyzAdd = nullptr; d->xyzCap = 0;
        const long long cap = xyzFloats + xyzFloats / 2;
        GS_HIP_CHECK(c, hipMalloc((void**)&d->xyzOwn, cap * sizeof(float)));
        GS_HIP_CHECK(c, hipMalloc((void**)&d->xyzAdd, cap * sizeof(float)));
        GS_HIP_CHECK(c, hipMemset(d->xyzAdd, 0, cap * sizeof(float)));      // (the pad behind 3 N is read by the Adam kernel's last float4)
        d->xyzCap = cap;
    } else if (xyzFloats > 3LL * N) {
        GS_HIP_CHECK(c, hipMemsetAsync(d->xyzAdd + 3LL * N, 0, (xyzFloats - 3LL * N) * sizeof(float), c->stream));     // (N moves with every densify event)
    }
    // blend backward, then ONE kernel: colour cotangents + gate word + the four geometry gradients (round 6; rounds 2-5 forked
    // the all-gather between two kernels, when the second one still staged the SH rows and took 46 us)
    if ((rc = gs_render_backward_dp_geom(c, a->cot_color, a->cot_depth, a->cot_alpha, a->color_cot_local, grad_of(c->fwd.xyz),
                                         grad_of(c->fwd.scales), grad_of(c->fwd.rot), grad_of(c->fwd.opacity), d->xyzOwn)))
        return rc;
    g_trace.lap(6);
    if ((rc = fork_after(c, d, d->evGeom))) return rc;
    g_trace.lap(7);
    const bool inlineGather = d->commGather != nullptr;
    if (!inlineGather) {
        mark(xt, XE_GATHER0, d->sComm);
        GS_NCCL_CHECK(c, d, d->lib->AllGather(a->color_cot_local, a->color_cot_all, (size_t)ccFloats, ncclFloat, d->comm, d->sComm));
        mark(xt, XE_GATHER1, d->sComm);
        GS_HIP_CHECK(c, hipEventRecord(d->evGather, d->sComm));
    }
    mark(xt, XE_REDUCE0, d->sComm);
    if (a->geom_numel > 0)
        GS_NCCL_CHECK(c, d, d->lib->AllReduce(a->grads_base, a->grads_base, (size_t)a->geom_numel, ncclFloat, ncclSum, d->comm, d->sComm));
    mark(xt, XE_REDUCE1, d->sComm);
    GS_HIP_CHECK(c, hipEventRecord(d->evReduce, d->sComm));
    g_trace.lap(8);
    // the gathered cotangents and, behind every rank's, its word: the SH rebuild ORs them into the step's gate word
    if (inlineGather) {
        // on the ctx stream itself, through the second communicator: all of it is "exposed", none of it is hops
        mark(xt, XE_GATHER0, c->stream);
        GS_NCCL_CHECK(c, d, d->lib->AllGather(a->color_cot_local, a->color_cot_all, (size_t)ccFloats, ncclFloat, d->commGather, c->stream));
        mark(xt, XE_GATHER1, c->stream);      // (gs_dp_exchange_read: with no wait events of its own, this interval is also the exposed time)
    } else {
        mark(xt, XE_WAIT_GATHER0, c->stream);
        GS_HIP_CHECK(c, hipStreamWaitEvent(c->stream, d->evGather, 0));
        mark(xt, XE_WAIT_GATHER1, c->stream);
    }
    c->ccBlockFloats = ccFloats; c->ccBlockCount = d->world; c->gatheredGateOut = d->words;
    const float* own[16] = {nullptr};
    own[d->rank] = d->xyzOwn;
    if ((rc = gs_sh_grad_from_views_adam_dir(c, N, K, d->world, c->fwd.xyz, a->color_cot_all, a->cam_centers, own,
                                             const_cast<float*>(c->fwd.fdc), const_cast<float*>(c->fwd.frest), a->params_base,
                                             a->m_base, a->v_base, a->n_arena, lr_at(c->fwd.fdc), K > 1 ? lr_at(c->fwd.frest) : 0.0f,
                                             a->beta1, a->beta2, a->eps, scale, d->xyzAdd)))
        return rc;
    g_trace.lap(9);
    mark(xt, XE_WAIT_REDUCE0, c->stream);
    GS_HIP_CHECK(c, hipStreamWaitEvent(c->stream, d->evReduce, 0));
    mark(xt, XE_WAIT_REDUCE1, c->stream);
    g_trace.lap(10);
    int nsegGeom = 0;
    while (nsegGeom < a->nseg && a->seg_end[nsegGeom] <= a->geom_numel) nsegGeom++;
    if (a->geom_numel == 0) return GS_OK;
    // (the xyz tensor leads the arena -- checked above: the geometry slice leads and xyz is its first segment when its
    // gradient pointer is grads_base; any other arena order takes the plain step on a slice the term was added to... which no
    // shipped host builds: refused)
    if (c->fwd.xyz != a->params_base) { c->err = "gs_dp_step: the xyz tensor must lead the arena (sh_compressed)"; return GS_ERR_SIZE_MISMATCH; }
    rc = gs_adam_step_add(c, a->geom_numel, a->params_base, a->grads_base, a->m_base, a->v_base, nsegGeom, a->seg_end, a->seg_lr,
                          a->beta1, a->beta2, a->eps, scale, d->xyzAdd, 3LL * N);
    g_trace.lap(11);
    return rc;
}

long long gs_dp_cc_floats(int N) { return N < 0 ? 0 : ((3LL * N + 1 + 3) & ~3LL); }

// ABI 6: the check in two halves -- _begin queues the checksum kernels on the ctx stream and the collective on the side stream
// behind them, _end waits for the reduced words and gives the verdict.  A host that calls _end where it waits for the device
// anyway (the next gs_dp_check_overflow look, the next event) keeps its queue filled through a densify event;
// gs_dp_check_replicas = _begin + _end.
int gs_dp_check_replicas_begin(gs_ctx* c, int N, const float* arena, long long n_arena)
{
    if (!c) return GS_ERR_INVALID_ARG;
    GsDp* d = c->dp;
    if (!d) { c->err = "gs_dp_check_replicas: no communicator (gs_dp_init / gs_dp_attach)"; return GS_ERR_INVALID_ARG; }
    if (N < 0 || n_arena < 0 || (n_arena > 0 && !arena)) { c->err = "gs_dp_check_replicas: bad arena"; return GS_ERR_INVALID_ARG; }
    if (d->replPending) { const int prc = gs_dp_check_replicas_end(c); if (prc) return prc; }
    hipLaunchKernelGGL(replica_partial_kernel, dim3(GS_REPL_BLOCKS), dim3(GS_REPL_THREADS), 0, c->stream, arena, n_arena, d->repl);
    double* words = d->repl + 2 * GS_REPL_BLOCKS;
    hipLaunchKernelGGL(replica_final_kernel, dim3(1), dim3(1), 0, c->stream, d->repl, (double)N, words);
    GS_HIP_CHECK(c, hipGetLastError());
    static_assert(sizeof(double) == sizeof(unsigned long long), "pinned words");
    double* mine = reinterpret_cast<double*>(d->hostWords + 2);      // pinned: [2..7] this rank's words, [8..13] the reduced ones
    double* all = reinterpret_cast<double*>(d->hostWords + 8);
    int rc;
    if ((rc = fork_after(c, d, d->evGeom))) return rc;
    GS_HIP_CHECK(c, hipMemcpyAsync(mine, words, 6 * sizeof(double), hipMemcpyDeviceToHost, d->sComm));
    GS_NCCL_CHECK(c, d, d->lib->AllReduce(words, words, 6, ncclDouble, ncclMax, d->comm, d->sComm));
    GS_HIP_CHECK(c, hipMemcpyAsync(all, words, 6 * sizeof(double), hipMemcpyDeviceToHost, d->sComm));
    GS_HIP_CHECK(c, hipEventRecord(d->evRepl, d->sComm));
    // (the next begin's kernels rewrite `words` on the ctx stream: they must not overtake this collective)
    GS_HIP_CHECK(c, hipStreamWaitEvent(c->stream, d->evRepl, 0));
    d->replPending = true;
    return GS_OK;
}

int gs_dp_check_replicas_end(gs_ctx* c)
{
    if (!c) return GS_ERR_INVALID_ARG;
    GsDp* d = c->dp;
    if (!d) { c->err = "gs_dp_check_replicas: no communicator (gs_dp_init / gs_dp_attach)"; return GS_ERR_INVALID_ARG; }
    if (!d->replPending) return GS_OK;
    d->replPending = false;
    GS_HIP_CHECK(c, hipEventSynchronize(d->evRepl));
    const double* mine = reinterpret_cast<const double*>(d->hostWords + 2);
    const double* all = reinterpret_cast<const double*>(d->hostWords + 8);
    const bool sameN = all[0] == -all[3], sameSum = all[1] == -all[4], sameAbs = all[2] == -all[5];
    if (sameN && sameSum && sameAbs) return GS_OK;
    char buf[400];
    snprintf(buf, sizeof buf, "gs_dp_check_replicas: the ranks hold different models (%s%s%s): over the ranks N in [%.0f, %.0f], sum in "
             "[%.17g, %.17g], sum of magnitudes in [%.17g, %.17g]; rank %d has N = %.0f, sum = %.17g, sum of magnitudes = %.17g",
             sameN ? "" : "N ", sameSum ? "" : "sum ", sameAbs ? "" : "magnitudes", -all[3], all[0], -all[4], all[1], -all[5], all[2],
             d->rank, mine[0], mine[1], mine[2]);
    c->err = buf;
    return GS_ERR_REPLICA_MISMATCH;
}

int gs_dp_check_replicas(gs_ctx* c, int N, const float* arena, long long n_arena)
{
    const int rc = gs_dp_check_replicas_begin(c, N, arena, n_arena);
    if (rc) return rc;
    return gs_dp_check_replicas_end(c);
}

// The ranks' densify PLANS compared (ABI 6): n <= 8 host words (gs_densify_plan_read's) are max-reduced together with their
// negatives in one fixed-size collective on the side stream -- which does NOT wait for the ctx stream: the event's gather and
// resets keep running while the ranks agree -- and a rank whose words differ from the extremes says so, as does every other
// rank (all see the same reduced words): a diverged plan stops the job before the next size-dependent collective.
int gs_dp_check_plan(gs_ctx* c, const long long* words, int n)
{
    if (!c) return GS_ERR_INVALID_ARG;
    GsDp* d = c->dp;
    if (!d) { c->err = "gs_dp_check_plan: no communicator (gs_dp_init / gs_dp_attach)"; return GS_ERR_INVALID_ARG; }
    if (!words || n < 1 || n > 8) { c->err = "gs_dp_check_plan: 1..8 words"; return GS_ERR_INVALID_ARG; }
    double* mine = d->planHost;
    double* all = d->planHost + 16;
    for (int i = 0; i < 8; i++) { const double w = i < n ? (double)words[i] : 0.0; mine[i] = w; mine[8 + i] = -w; }
    GS_HIP_CHECK(c, hipMemcpyAsync(d->planDev, mine, 16 * sizeof(double), hipMemcpyHostToDevice, d->sComm));
    GS_NCCL_CHECK(c, d, d->lib->AllReduce(d->planDev, d->planDev, 16, ncclDouble, ncclMax, d->comm, d->sComm));
    GS_HIP_CHECK(c, hipMemcpyAsync(all, d->planDev, 16 * sizeof(double), hipMemcpyDeviceToHost, d->sComm));
    GS_HIP_CHECK(c, hipStreamSynchronize(d->sComm));
    int bad = -1;
    for (int i = 0; i < n && bad < 0; i++) if (all[i] != -all[8 + i]) bad = i;
    if (bad < 0) return GS_OK;
    static const char* names[8] = {"new N", "applies", "total", "keep", "split", "clone", "prune", "N"};
    char buf[400];
    int len = snprintf(buf, sizeof buf, "gs_dp_check_plan: the ranks planned different densify events (");
    for (int i = 0; i < n && len < (int)sizeof buf - 40; i++)
        if (all[i] != -all[8 + i]) len += snprintf(buf + len, sizeof buf - len, "%s in [%.0f, %.0f] ", names[i], -all[8 + i], all[i]);
    snprintf(buf + len, sizeof buf - len, "); rank %d planned new N = %.0f from N = %.0f", d->rank, mine[0], n > 7 ? mine[7] : 0.0);
    c->err = buf;
    return GS_ERR_REPLICA_MISMATCH;
}

int gs_dp_exchange_timing(gs_ctx* c, int enable)
{
    if (!c) return GS_ERR_INVALID_ARG;
    GsDp* d = c->dp;
    if (!d) { c->err = "gs_dp_exchange_timing: no communicator (gs_dp_init / gs_dp_attach)"; return GS_ERR_INVALID_ARG; }
    GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    GS_HIP_CHECK(c, hipStreamSynchronize(d->sComm));
    d->timing = enable != 0;
    if (d->timing) d->timedUsed = 0;
    return GS_OK;
}

int gs_dp_exchange_read(gs_ctx* c, float ms[GS_DP_XT_COUNT], int* steps, int* rccl_version)
{
    if (!c || !ms) return GS_ERR_INVALID_ARG;
    GsDp* d = c->dp;
    if (!d) { c->err = "gs_dp_exchange_read: no communicator (gs_dp_init / gs_dp_attach)"; return GS_ERR_INVALID_ARG; }
    GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    GS_HIP_CHECK(c, hipStreamSynchronize(d->sComm));
    for (int i = 0; i < GS_DP_XT_COUNT; i++) ms[i] = 0.0f;
    static const int pairOf[5][3] = {{GS_DP_XT_GATE, XE_GATE0, XE_GATE1}, {GS_DP_XT_GATHER, XE_GATHER0, XE_GATHER1},
                                     {GS_DP_XT_REDUCE, XE_REDUCE0, XE_REDUCE1},
                                     {GS_DP_XT_EXPOSED_GATHER, XE_WAIT_GATHER0, XE_WAIT_GATHER1},
                                     {GS_DP_XT_EXPOSED_REDUCE, XE_WAIT_REDUCE0, XE_WAIT_REDUCE1}};
    for (size_t i = 0; i < d->timedUsed; i++)
        for (const auto& p : pairOf) {
            float t = 0.0f;
            if (d->timed[i].used[p[1]] && d->timed[i].used[p[2]] &&
                hipEventElapsedTime(&t, d->timed[i].e[p[1]], d->timed[i].e[p[2]]) == hipSuccess) {
                ms[p[0]] += t;
                // the all-gather issued on the ctx stream itself (commGather): its duration IS the render stream's wait
                if (p[0] == GS_DP_XT_GATHER && !d->timed[i].used[XE_WAIT_GATHER0]) ms[GS_DP_XT_EXPOSED_GATHER] += t;
            }
        }
    if (steps) *steps = (int)d->timedUsed;
    if (rccl_version) {
        int v = 0;
        if (d->lib->GetVersion) (void)d->lib->GetVersion(&v);
        *rccl_version = v;
    }
    return GS_OK;
}

int gs_dp_check_overflow(gs_ctx* c, int* regrown, long long* pairs_needed)
{
    if (!c) return GS_ERR_INVALID_ARG;
    GsDp* d = c->dp;
    if (!d) { c->err = "gs_dp_check_overflow: no communicator (gs_dp_init / gs_dp_attach)"; return GS_ERR_INVALID_ARG; }
    if (regrown) *regrown = 0;
    if (pairs_needed) *pairs_needed = 0;
    // "was any step since the last check gated?" is built from REDUCED words, so every rank reads the same answer and
    // the collective below is entered by all of them or by none
    GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    GS_HIP_CHECK(c, hipStreamSynchronize(d->sComm));
    uint32_t seen = 0;
    GS_HIP_CHECK(c, hipMemcpy(&seen, d->words + 1, sizeof seen, hipMemcpyDeviceToHost));
    if (!seen) return GS_OK;
    // (an overflow of the checkpoint arena needs no more pairs: ask for what is held, gs_ctx_reserve regrows the arena)
    unsigned long long mine = c->missHost[4] == 1u ? (unsigned long long)c->missHost[5] : 0ull;
    if (c->missHost[4] == 2u || c->arenaRegrowPending) {
        c->arenaRegrowPending = true;
        if (mine < (unsigned long long)c->capM) mine = (unsigned long long)c->capM;
    }
    GS_HIP_CHECK(c, hipMemcpy(d->need, &mine, sizeof mine, hipMemcpyHostToDevice));
    GS_NCCL_CHECK(c, d, d->lib->AllReduce(d->need, d->need, 1, ncclUint64, ncclMax, d->comm, d->sComm));
    GS_HIP_CHECK(c, hipStreamSynchronize(d->sComm));
    unsigned long long need = 0;
    GS_HIP_CHECK(c, hipMemcpy(&need, d->need, sizeof need, hipMemcpyDeviceToHost));
    GS_HIP_CHECK(c, hipMemset(d->words, 0, 2 * sizeof(uint32_t)));
    c->missHost[4] = 0;
    if (pairs_needed) *pairs_needed = (long long)need;
    if (need == 0) return GS_OK;
    long long want = need > (unsigned long long)c->capM ? (long long)(need + need / 2 + 65536) : c->capM;
    const int rc = gs_ctx_reserve(c, c->capN, want);
    if (rc == GS_OK && regrown) *regrown = 1;
    return rc;
}

}  // extern "C"
#pragma GCC visibility pop
