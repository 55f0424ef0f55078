// C ABI of libschpf_hip.so (include/schpf_hip.h): context management, uploads, and the
// ordering of kernel launches that makes one CAVI iteration (scHPF_.py:657-714).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/schpf_hip.h"
#include "kernels.h"
#include "plan.h"

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

struct HipError : std::runtime_error {
    hipError_t code;
    HipError(const std::string &what, hipError_t code_ = hipErrorUnknown) : std::runtime_error(what), code(code_) {}
};

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            char b_[512];                                                                    \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                     __FILE__, __LINE__);                                                    \
            throw HipError(b_, e_);                                                          \
        }                                                                                    \
    } while (0)

template <typename F> int guarded(F &&f)
{
    // status SCHPF_ERR_NO_MEMORY: the device (hipErrorOutOfMemory) or the host (std::bad_alloc while building plans) ran
    // out of memory -- the one failure a caller may answer with a smaller layout; everything else is 1
    try {
        f();
        return 0;
    } catch (const HipError &e) {
        g_err = e.what();
        if (e.code == hipErrorOutOfMemory) (void)hipGetLastError();
        return e.code == hipErrorOutOfMemory ? SCHPF_ERR_NO_MEMORY : 1;
    } catch (const schpf::DeviceNoMemory &e) {
        g_err = e.what();
        return SCHPF_ERR_NO_MEMORY;
    } catch (const std::bad_alloc &) {
        g_err = "out of host memory (std::bad_alloc)";
        return SCHPF_ERR_NO_MEMORY;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    } catch (...) {
        g_err = "unknown error";
        return 1;
    }
}

// RAII device buffer
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    void alloc(size_t n, bool zero = false, hipStream_t st = nullptr)
    {
        release();
        bytes = n ? n : 16;
        HIPCHK(hipMalloc(&p, bytes));
        if (zero) HIPCHK(hipMemsetAsync(p, 0, bytes, st));
    }
    template <typename U> U *as() const { return reinterpret_cast<U *>(p); }
};

template <typename U, typename A> void upload(DevBuf &b, const std::vector<U, A> &v, hipStream_t st)
{
    b.alloc(v.size() * sizeof(U));
    if (!v.empty()) HIPCHK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(U), hipMemcpyHostToDevice, st));
}

int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

struct PlanDev {
    schpf::SweepPlanHost host;  // entries cleared after upload; order/mptr/cptr kept
    DevBuf entries, slice_off, slice_steps, chunk_major, chunk_natid, wave_slice, cptr, partials;
    int64_t n_waves = 0, n_chunks = 0, entry_slots = 0;
};

struct TileDev {
    schpf::TilePlanHost host;   // entries/steps cleared after upload; order/mptr kept
    DevBuf entries, steps, block_rows, task_block, task_w0, task_w1, task_wave_off, pfirst, pcount, partials;
    DevBuf task_order;          // tasks by decreasing work: the slot list of a persistent single-side launch
    // The loss pass (MODE_LLH) writes no partial rows, so its tasks may be cut finer than the iteration's: sub-ranges of
    // the tasks' window ranges, enough of them for a few rounds of the device (Engine::loss_tasks)
    DevBuf llh_block, llh_w0, llh_w1, llh_stage_end, llh_wave_off, llh_order;
    int64_t n_llh_tasks = 0;
    double llh_model = 0.0;     // modelled length of the loss pass on this plan, in step units (0: unknown)
    int llh_parts = 1;          // sub-ranges per task the model chose for the loss pass
    DevBuf minor_of;            // balanced windows (plan.h): [n_blocks * n_virtual] table row staged at a window position, or empty
    int n_virtual = 0;
    DevBuf order_dev;           // device-built plans: (major, minor)-sorted position -> caller's COO position
    bool order_identity = false; //                    ... or the input was already in that order
    int64_t n_tasks = 0, entry_slots = 0, n_wave_out = 0;
    int threads = 512;
    size_t lds_bytes = 0;
    bool packed = false;
};

struct Profiler {
    bool on = false;
    struct Rec { int kind; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        return e;
    }
    ~Profiler()
    {
        for (auto &r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

struct ScopedTimer {
    Profiler &p; hipStream_t st; int kind; hipEvent_t a{}, b{};
    ScopedTimer(Profiler &p_, hipStream_t st_, int kind_) : p(p_), st(st_), kind(kind_)
    {
        if (p.on) { a = p.get(); b = p.get(); HIPCHK(hipEventRecord(a, st)); }
    }
    void stop()
    {
        if (p.on) { HIPCHK(hipEventRecord(b, st)); p.recs.push_back({kind, a, b}); }
    }
};

// ---- RCCL, bound at run time.  The library has no DT_NEEDED on RCCL for the reason it has none on
// the HIP runtime (Makefile): a process must use ONE copy, and PyTorch bundles its own.  The copy
// already in the process is taken when there is one (RTLD_NOLOAD), else $SCHPF_RCCL_PATH, else
// the system's.  Only the handful of entry points the sharded iteration needs; the types are the
// C ABI of rccl.h (ncclUniqueId = 128 opaque bytes, ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0).
struct RcclUniqueId { char internal[128]; };
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl load_rccl()
{
    Rccl r;
    const char *env = getenv("SCHPF_RCCL_PATH");
    const char *names[] = {"librccl.so", "librccl.so.1", env && *env ? env : nullptr, "librccl.so", "librccl.so.1",
                           "/opt/rocm/lib/librccl.so"};
    for (int i = 0; i < 6 && !r.handle; ++i) {
        if (!names[i]) continue;
        r.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL | (i < 2 ? RTLD_NOLOAD : 0));
    }
    if (!r.handle) throw std::runtime_error("cannot load RCCL (librccl.so): set SCHPF_RCCL_PATH");
    auto sym = [&](const char *n) {
        void *p = dlsym(r.handle, n);
        if (!p) throw std::runtime_error(std::string("RCCL lacks ") + n);
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    return r;
}
Rccl &rccl()
{
    static Rccl r = load_rccl();   // thread-safe; a failed load throws and is tried again by the next caller
    return r;
}
#define RCCLCHK(expr)                                                                                     \
    do {                                                                                                  \
        const int r_ = (expr);                                                                            \
        if (r_ != 0) throw std::runtime_error(std::string(#expr " failed: ") + rccl().GetErrorString(r_)); \
    } while (0)

}  // namespace

// ------------------------------------------------------------------------------------
struct schpf_ctx {
    int device = 0, dtype = SCHPF_F64, N = 0, G = 0, K = 0;
    int KP = 0, KL = 0, LPC = 1, NV = 1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    virtual ~schpf_ctx() { comm_destroy(); }
    virtual void upload_coo(int64_t nnz, const int32_t *row, const int32_t *col, const void *val, int kind) = 0;
    virtual void set_state(int which, const void *shape, const void *rate) = 0;
    virtual void get_state(int which, void *shape, void *rate) = 0;
    virtual void init_phi_host(const double *xphi) = 0;
    virtual void init_phi_device(uint64_t seed) = 0;
    virtual void step_local(unsigned flags) = 0;
    virtual void step_finish(unsigned flags) = 0;
    virtual void steps(unsigned flags, int n) = 0;
    virtual void hypers_changed() = 0;
    virtual void hint_sharded(int on) = 0;
    virtual void hint_transient(int on) = 0;
    virtual void keep_rows(int on) = 0;
    virtual void upload_rows(schpf_ctx *source, const int32_t *rows, int n_rows) = 0;
    virtual void steps_sharded(unsigned flags, int n) = 0;
    virtual void loss_terms_all(double *llh, double *gl, int64_t *nnz) = 0;
    // cells sharded over GPUs: this rank's RCCL communicator and the stream its collectives run on
    void *comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_reduced = nullptr;
    void comm_init(const void *id, int rank, int world)
    {
        if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("rank must be in [0, world)");
        comm_destroy();
        RcclUniqueId uid;
        std::memcpy(&uid, id, sizeof uid);
        RCCLCHK(rccl().CommInitRank(&comm, world, uid, rank));
        comm_rank = rank; comm_world = world;
        HIPCHK(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ev_packed, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_reduced, hipEventDisableTiming));
    }
    void comm_destroy()
    {
        if (comm) { (void)hipStreamSynchronize(comm_stream); (void)rccl().CommDestroy(comm); comm = nullptr; }
        if (comm_stream) { (void)hipStreamDestroy(comm_stream); comm_stream = nullptr; }
        if (ev_packed) { (void)hipEventDestroy(ev_packed); ev_packed = nullptr; }
        if (ev_reduced) { (void)hipEventDestroy(ev_reduced); ev_reduced = nullptr; }
    }
    virtual void exchange(void **p, int64_t *count) = 0;
    virtual void loss_terms(double *llh, double *gl, int64_t *nnz) = 0;
    virtual void plan_info(int64_t info[16]) = 0;
    virtual void upload_info(int64_t info[4]) = 0;
    virtual void profile_clock(double *shader_mhz, int64_t *launches) = 0;
    virtual void sweep_bytes(int64_t info[8]) = 0;
    double a = 0.3, c = 0.3, bp = 1.0, dp = 1.0;
    Profiler prof;
};

namespace {

template <typename T> struct Engine final : schpf_ctx {
    // variational parameters (C-contiguous, stride K)
    DevBuf xi_s, xi_r, th_s, th_r, eta_s, eta_r, be_s, be_r;
    // tables, stride KP, padding columns zero
    DevBuf th_exp, th_e, th_log, be_exp, be_e, be_log;
    DevBuf exchange_buf;                            // [G*K + K] of T
    DevBuf dense_cell;                              // [N*K] of T (t = 0 only)
    DevBuf s_theta, s_beta, s_beta_next;            // double[K]
    DevBuf colpart_cell, colpart_gene;              // double[UPD_BLOCKS * K]
    DevBuf wave_out, scalars;                       // llh per wave; scalars[0]=llh sum
    // the loss pass's two results land in pinned host memory that the device writes directly: the reduction kernels
    // store there, the host reads after the stream has drained -- no copy of 24 bytes out of pageable memory per check
    double *loss_host = nullptr;
    PlanDev cell, gene;                             // gather plans: major = cell / major = gene
    TileDev tcell, tgene;                           // tile plans (LDS-staged sweep)
    DevBuf dual_order;                              // merged launch order of both plans' tasks (or empty)
    DevBuf dual_queue;                              // persistent dual launch: {next slot, workgroups done}, self-zeroing
    DevBuf clock_probe;                             // 5 x u64: shader cycles, constant-rate ticks, 2 start stamps, launches (sweep_impl.h)
    int64_t dual_slots = 0;
    bool use_tile = false, want_tile = true;
    int64_t nnz = 0;
    double gammaln_sum = 0.0;
    int64_t n_rounded = 0, n_zero = 0;              // upload facts: values rounded to float32; stored zeros
    DevBuf zero_row, zero_col;                      // positions of explicitly stored zeros (loss only)
    bool have_coo = false;
    bool dirty_theta = true, dirty_beta = true;
    int pending_init = 0;  // 0 none, 1 dense accumulators, 2 chunk partials
    // n iterations captured as one hipGraph (schpf_steps): the state is device-resident and nothing on
    // the host changes between two loss checks, so a fit replays one graph per check interval
    // The sum-of-beta buffers swap roles every iteration (beta_parity counts the swaps mod 2) and a
    // capture bakes the pointers in, so a graph is keyed by (flags, n, parity at its start): one cached
    // graph per parity.  A stretch with an odd count (check_freq = 5: graph of 4 + one eager iteration)
    // starts its calls at alternating parities and alternates between the two.
    struct CachedGraph { hipGraphExec_t exec = nullptr; unsigned flags = 0; int n = 0; };
    CachedGraph graphs[2];
    int beta_parity = 0;
    bool eager_since_upload = false;   // one eager iteration has run on this plan (kernel attributes are set)
    // small problems: the update kernels sum the other side's per-block column sums themselves and the
    // two reduce launches of an iteration are skipped; s_theta / s_beta are then brought up to date
    // only when a path that reads them comes along (sums_stale)
    bool sums_stale = false;
    // Minibatch CAVI without re-uploads (scHPF_.py:643-650): an engine that was told to keep_rows() holds, beside
    // its plans, the matrix once more as a (row, col)-sorted device copy; a batch engine's upload_rows(source,
    // rows) gathers its rows from there and builds its plans from device arrays -- no host slicing, no PCIe.
    bool want_rows = false, rows_packed_ok = true;
    DevBuf rows_ptr, rows_col, rows_val;            // int64[N + 1], int32[nnz], float[nnz]; host copy of rows_ptr: tcell.host.mptr
    bool have_loss_constants = true;                // false after upload_rows (no lgamma sum / stored-zero list for a batch)
    // balanced windows (plan.h): on for uploads of a whole matrix; off for an engine that keeps a (row, col)-sorted copy
    // (the plans' own order is then the virtual one) and for batch engines, which re-plan every iteration
    bool balance_now = false;
    bool transient = false;             // schpf_hint_transient: the matrix is replaced every iteration, plan the cheapest way
    bool planning_batch_rows = false;   // inside schpf_upload_rows (gathered batch rows: no loss constants, no loss tasks)
    bool expect_sharded = false;        // schpf_hint_sharded: a rank of a sharded fit (gene-side sums leave for an all-reduce)
    static constexpr int UPD_BLOCKS = 2048;
    static constexpr size_t TABLE_PAD = 256 * 1024;

    // The COO's index arrays start their trip over PCIe on a helper thread and a copy stream of its own
    // while the calling thread is still validating / converting the values and sampling the block loads:
    // the copy does not care whether the indices are in range, only the plan kernels do (and they run after
    // the validation has passed).
    struct EarlyIndexCopy {
        DevBuf d_row, d_col;
        std::thread worker;
        std::string error;
        double seconds = 0.0;
        bool started = false;
        void start(int device, int64_t n, const int32_t *row, const int32_t *col)
        {
            d_row.alloc((size_t)n * 4); d_col.alloc((size_t)n * 4);
            started = true;
            worker = std::thread([this, device, n, row, col] {
                const double t0 = now_s();
                hipStream_t cs = nullptr;
                hipError_t e = hipSetDevice(device);
                if (e == hipSuccess) e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
                if (e == hipSuccess && n > 0) e = hipMemcpyAsync(d_col.p, col, (size_t)n * 4, hipMemcpyHostToDevice, cs);
                if (e == hipSuccess && n > 0) e = hipMemcpyAsync(d_row.p, row, (size_t)n * 4, hipMemcpyHostToDevice, cs);
                if (e == hipSuccess) e = hipStreamSynchronize(cs);
                if (cs) (void)hipStreamDestroy(cs);
                if (e != hipSuccess) error = std::string("H2D of the COO indices failed: ") + hipGetErrorString(e);
                seconds = now_s() - t0;
            });
        }
        void join() { if (worker.joinable()) worker.join(); }
        ~EarlyIndexCopy() { join(); }
    };
    Engine(int device_, void *stream_, int dtype_, int N_, int G_, int K_)
    {
        device = device_; dtype = dtype_; N = N_; G = G_; K = K_;
        HIPCHK(hipSetDevice(device));
        // NULL: a stream of our own; SCHPF_STREAM_DEFAULT: the device's null stream (what
        // torch.cuda.current_stream() is unless the caller switched streams); else the given handle
        if (stream_ == SCHPF_STREAM_DEFAULT) stream = nullptr;
        else if (stream_) stream = (hipStream_t)stream_;
        else { HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); own_stream = true; }
        {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
                cu_count = prop.multiProcessorCount;
        }
        choose_config();
        const size_t s = sizeof(T);
        dual_queue.alloc(2 * sizeof(int), true, stream);
        clock_probe.alloc(8 * sizeof(unsigned long long), true, stream);
        xi_s.alloc((size_t)N * s); xi_r.alloc((size_t)N * s);
        eta_s.alloc((size_t)G * s); eta_r.alloc((size_t)G * s);
        th_s.alloc((size_t)N * K * s); th_r.alloc((size_t)N * K * s);
        be_s.alloc((size_t)G * K * s); be_r.alloc((size_t)G * K * s);
        // + TABLE_PAD zero bytes: slack behind the last row for whole-piece copies
        for (DevBuf *b : {&th_exp, &th_e, &th_log}) b->alloc((size_t)N * KP * s + TABLE_PAD, true, stream);
        for (DevBuf *b : {&be_exp, &be_e, &be_log}) b->alloc((size_t)G * KP * s + TABLE_PAD, true, stream);
        exchange_buf.alloc(((size_t)G * K + K) * s, true, stream);
        for (DevBuf *b : {&s_theta, &s_beta, &s_beta_next}) b->alloc((size_t)K * sizeof(double), true, stream);
        colpart_cell.alloc((size_t)UPD_BLOCKS * K * sizeof(double));
        colpart_gene.alloc((size_t)UPD_BLOCKS * K * sizeof(double));
        scalars.alloc(8 * sizeof(double), true, stream);
        if (hipHostMalloc((void **)&loss_host, 8 * sizeof(double), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            loss_host = nullptr;            // falls back to the copy out of `scalars`
        } else std::memset(loss_host, 0, 8 * sizeof(double));
    }
    void hypers_changed() override { drop_graph(); }
    // the engine holds no count matrix any more: plans, row copy and captured graphs released; step / loss calls
    // raise until the next successful upload
    void forget_matrix()
    {
        have_coo = false;
        drop_graph();
        HIPCHK(hipStreamSynchronize(stream));
        cell = PlanDev(); gene = PlanDev(); tcell = TileDev(); tgene = TileDev();
        dual_order.release(); dual_slots = 0;
        rows_ptr.release(); rows_col.release(); rows_val.release();
        zero_row.release(); zero_col.release();
        pending_init = 0;
        eager_since_upload = false;
    }
    void hint_sharded(int on) override { expect_sharded = on != 0; }
    void hint_transient(int on) override { transient = on != 0; }
    void keep_rows(int on) override { want_rows = on != 0; }   // a, c, bp, dp are kernel arguments of the captured launches
    void drop_graph()
    {
        for (CachedGraph &g : graphs) {
            if (g.exec) { (void)hipStreamSynchronize(stream); (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
            g.n = 0;
        }
    }
    // the graph of `count` (even) iterations issued by `body` for the current parity: cached or captured now
    template <typename F> hipGraphExec_t graph_for(unsigned key_flags, int count, F &&body)
    {
        CachedGraph &g = graphs[beta_parity & 1];
        if (g.exec && g.flags == key_flags && g.n == count) return g.exec;
        if (g.exec) { (void)hipStreamSynchronize(stream); (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; g.n = 0; }
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        try {
            body(count);   // runs the host side of `count` iterations (an even number of swaps) without executing them
        } catch (...) {
            (void)hipStreamEndCapture(stream, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        HIPCHK(hipStreamEndCapture(stream, &graph));
        const hipError_t e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIPCHK(e);
        g.flags = key_flags;
        g.n = count;
        return g.exec;
    }

    // n iterations of schpf_step.  From the second call on with the same (flags, n) they are one
    // graph launch: launch overhead is what bounds small matrices (BASELINE C2: five launches of
    // 5-25 us each per iteration).  The sum-of-beta buffers swap roles every iteration, so a graph
    // always holds an even number of iterations; an odd one runs eagerly.
    void steps(unsigned flags_, int n) override
    {
        if (n < 0) throw std::invalid_argument("n must be >= 0");
        const bool graphable = env_int("SCHPF_GRAPH", 1) && !prof.on && stream != nullptr && pending_init == 0 &&
                               eager_since_upload && !dirty_theta && !dirty_beta && !(flags_ & SCHPF_SHARDED);
        int done = 0;
        if (graphable && n >= 2) {
            const int even = n & ~1;
            hipGraphExec_t exec = graph_for(flags_, even, [&](int count) {
                for (int i = 0; i < count; ++i) { step_local(flags_); step_finish(flags_); }
            });
            HIPCHK(hipGraphLaunch(exec, stream));
            done = even;
        }
        for (; done < n; ++done) { step_local(flags_); step_finish(flags_); }
        if (n > 0) eager_since_upload = true;
    }

    // n iterations with the cells sharded over the ranks of `comm` (sharded.py protocol, driven from
    // here): gene-side sweep + packing on the context's stream; ONE all-reduce of [G*K sums | K sums
    // of E[theta]] on the communicator's stream, ordered after the packing by an event; the
    // cell-side sweep meanwhile; the update kernels after an event on the all-reduce.  No host
    // round trip and no Python between the launches of an iteration.
    void steps_sharded(unsigned flags_, int n) override
    {
        if (!comm) throw std::logic_error("no communicator (schpf_comm_init)");
        const unsigned base = (flags_ | SCHPF_SHARDED) & ~(unsigned)(SCHPF_LOCAL_GENE | SCHPF_LOCAL_CELL);
        const bool freeze = flags_ & SCHPF_FREEZE_GENES;
        const int dt = sizeof(T) == 4 ? 7 : 8;   // ncclFloat32 / ncclFloat64
        auto iterate = [&](int count) {
        for (int i = 0; i < count; ++i) {
            if (freeze) { step_local(base); step_finish(base); continue; }   // nothing to exchange
            step_local(base | SCHPF_LOCAL_GENE);
            HIPCHK(hipEventRecord(ev_packed, stream));
            HIPCHK(hipStreamWaitEvent(comm_stream, ev_packed, 0));
            RCCLCHK(rccl().AllReduce(exchange_buf.p, exchange_buf.p, (size_t)G * K + K, dt, 0, comm, comm_stream));
            HIPCHK(hipEventRecord(ev_reduced, comm_stream));
            step_local(base | SCHPF_LOCAL_CELL);
            HIPCHK(hipStreamWaitEvent(stream, ev_reduced, 0));
            step_finish(base);
        }
        };
        // The stretch as one hipGraph -- both streams, the events between them and the RCCL all-reduce captured (RCCL
        // supports stream capture): 0.183 -> 0.168 ms per iteration of a 1/8 shard of C3.  The default for a one-rank
        // communicator, which is all this build could ever run it with; with more ranks every rank must replay the same
        // graph, so there it stays opt-in (SCHPF_GRAPH_SHARDED=1) until tests/test_multigpu.py has seen two GPUs.
        int done = 0;
        const bool graphable = env_int("SCHPF_GRAPH_SHARDED", comm_world == 1 ? 1 : 0) && !prof.on && stream != nullptr &&
                               pending_init == 0 && eager_since_upload && !dirty_theta && !dirty_beta && !freeze;
        if (graphable && n >= 2) {
            const int even = n & ~1;
            hipGraphExec_t exec = graph_for(base | 0x80000000u, even, iterate);
            HIPCHK(hipGraphLaunch(exec, stream));
            done = even;
        }
        iterate(n - done);
        if (n > 0) eager_since_upload = true;
    }

    // loss terms summed over the ranks (three doubles through the same communicator)
    void loss_terms_all(double *llh, double *gl, int64_t *nnz_out) override
    {
        if (!comm) throw std::logic_error("no communicator (schpf_comm_init)");
        double h[3];
        int64_t local_nnz = 0;
        loss_terms(&h[0], &h[1], &local_nnz);
        h[2] = (double)local_nnz;
        double *d = scalars.as<double>() + 4;
        HIPCHK(hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, stream));
        RCCLCHK(rccl().AllReduce(d, d, 3, 8, 0, comm, stream));
        HIPCHK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        *llh = h[0]; *gl = h[1]; *nnz_out = (int64_t)(h[2] + 0.5);
    }

    ~Engine() override
    {
        drop_graph();
        (void)hipStreamSynchronize(stream);
        if (loss_host) (void)hipHostFree(loss_host);
        if (own_stream) (void)hipStreamDestroy(stream);
    }

    // Row layout of the tables the sweeps read: KP = NV * LPC * VEC values (VEC = values per 16 B);
    // a group of LPC lanes shares one row, lane `sub` holding the 16-byte vectors q*LPC + sub.
    //  * tile plan (LDS-staged): VALU work per nonzero has a fixed part (reciprocal, cross-lane
    //    sum, addressing) that every lane of the group repeats, so rows are split over as FEW
    //    lanes as the register budget allows: the smallest LPC with <= 112 row bytes per lane
    //    (measured on C3: f64 K=20 -> LPC 2, f32 K=20 -> LPC 1; profiles/r01/explore*.log);
    //  * gather plan (L2): the L1 path is charged per 64-byte sector touched, so the cost model
    //    is accesses per nonzero = NV * max(1, LPC/4) and LPC = 4 usually wins.
    void choose_config()
    {
        if (K < 1 || K > 256) throw std::invalid_argument("nfactors must be in [1, 256]");
        const int vec = 16 / (int)sizeof(T);
        const int nvec = (K + vec - 1) / vec;
        static const int nv_ok[] = {1, 2, 3, 4, 5, 6, 7, 8, 10};
        const char *pk = getenv("SCHPF_PLAN");
        want_tile = (size_t)nvec * 16 <= 1024;          // at least ~150 rows per 152 KiB window
        if (pk && !strcmp(pk, "gather")) want_tile = false;
        if (pk && !strcmp(pk, "tile")) want_tile = true;
        const int force_lpc = env_int("SCHPF_LPC", 0);
        int best_lpc = 0, best_nv = 0, best_cost = 1 << 30;
        static const int order_tile[] = {1, 2, 4, 8, 16};
        static const int order_gather[] = {4, 8, 2, 16, 1};
        for (int lpc : (want_tile ? order_tile : order_gather)) {
            if (force_lpc && lpc != force_lpc) continue;
            const int need = (nvec + lpc - 1) / lpc;
            int nv = 0;
            for (int v : nv_ok) if (v >= need) { nv = v; break; }
            if (!nv) continue;
            int cost;
            if (want_tile && nv > 7) continue;                 // the tile sweeps are instantiated for <= 7 vectors
            if (want_tile) cost = (nv * 16 <= 112 || force_lpc) ? 0 : 1 << 20;   // first that fits
            else cost = nv * std::max(1, lpc / 4);
            // only instantiated pairs (kernels.h): the planner's own choices always are, a forced LPC may not be
            if (!(want_tile ? schpf::tile_combo_ok(nv, lpc) : schpf::gather_combo_ok(nv, lpc))) continue;
            if (cost < best_cost) { best_cost = cost; best_lpc = lpc; best_nv = nv; }
        }
        if (!best_lpc) throw std::invalid_argument("SCHPF_LPC must be one of 1,2,4,8,16, fit nfactors and be an "
                                                   "instantiated shape (kernels.h tile_combo_ok / gather_combo_ok)");
        LPC = best_lpc; NV = best_nv; KL = NV * vec; KP = KL * LPC;
    }

    static int pick_windows(size_t table_bytes, const char *envname)
    {
        int w = env_int(envname, 0);
        if (w > 0) return w;
        const size_t budget = (size_t)env_int("SCHPF_L2_BUDGET_KB", 2048) * 1024;
        w = 1;
        while ((table_bytes + w - 1) / w > budget && w < 4096) w *= 2;
        return w;
    }

    void build_plan(PlanDev &pd, int64_t nnz_, const int32_t *major, const int32_t *minor, const float *val,
                    int n_major, int n_minor, int windows, int chunk_len)
    {
        schpf::build_sweep_plan(nnz_, major, minor, val, n_major, n_minor, LPC, chunk_len, windows, true,
                                pd.host);
        auto &h = pd.host;
        pd.n_waves = h.n_waves;
        pd.n_chunks = h.n_chunks;
        pd.entry_slots = (int64_t)h.entries.size() / 2;
        upload(pd.entries, h.entries, stream);
        upload(pd.slice_off, h.slice_off, stream);
        upload(pd.slice_steps, h.slice_steps, stream);
        upload(pd.chunk_major, h.chunk_major, stream);
        upload(pd.chunk_natid, h.chunk_natid, stream);
        upload(pd.wave_slice, h.wave_slice, stream);
        upload(pd.cptr, h.cptr, stream);
        pd.partials.alloc((size_t)std::max<int64_t>(h.n_chunks, 1) * KP * sizeof(T), true, stream);
        HIPCHK(hipStreamSynchronize(stream));
        schpf::BigVec<uint32_t>().swap(h.entries);
        std::vector<int32_t>().swap(h.chunk_major);
        std::vector<int32_t>().swap(h.chunk_natid);
        std::vector<int32_t>().swap(h.wave_slice);
        std::vector<int64_t>().swap(h.slice_off);
        std::vector<int32_t>().swap(h.slice_steps);
    }

    // device half of a tile plan: upload the host-built arrays (td.host), allocate the partials
    void upload_tile(TileDev &td, double host_seconds)
    {
        const double t1 = now_s();
        auto &h = td.host;
        td.entry_slots = (int64_t)h.entries.size() / (h.packed ? 1 : 2);
        upload(td.entries, h.entries, stream);
        upload(td.steps, h.steps, stream);
        finish_tile(td);
        if (env_int("SCHPF_VERBOSE", 0))
            fprintf(stderr, "[schpf_hip]   tile plan %d x %d: host build %.3f s, H2D %.3f s (%.2f GB entries)\n",
                    h.n_major, h.n_minor, host_seconds, now_s() - t1, h.entries.size() * 4e-9);
        schpf::BigVec<uint32_t>().swap(h.entries);
    }

    // Tasks of the loss pass.  The iteration's task ranges are chosen for the merged launch of both orientations and for
    // few partial rows (C3 f64: ONE range per cell block = 196 tasks on 256 compute units -- a loss pass over them ran
    // 0.42 ms where half a dual launch is 0.32); the loss pass keeps no partial rows, so every task's window range may be
    // cut into `parts` sub-ranges (never below two windows / four sub-windows per sub-task: a first window costs a staging
    // and the major rows).  `parts` is the count in 1..8 with the shortest modelled pass: the sub-tasks, longest first,
    // on the resident workgroups (list schedule), a sub-task = its barrier-limited steps + two per window + a fixed cost.
    // What decides is the last round: 784 equal tasks on 256 workgroups take four rounds, not 3.06 (measured: 0.46 ms
    // against 0.41 for the gene-side plan's 640).  The modelled time also picks the plan (loss_side).  Needs the host
    // copies of steps / task_wave_off.
    void loss_tasks(TileDev &td)
    {
        auto &h = td.host;
        td.n_llh_tasks = 0;
        td.llh_model = 0.0;
        for (DevBuf *b : {&td.llh_block, &td.llh_w0, &td.llh_w1, &td.llh_stage_end, &td.llh_wave_off, &td.llh_order}) b->release();
        if (h.n_tasks <= 0 || h.steps.empty() || h.task_wave_off.empty()) return;
        // a matrix that is replaced every iteration (minibatch engines: schpf_hint_transient, schpf_upload_rows) is planned the
        // cheapest way, and batch engines never evaluate the loss themselves
        if (transient || planning_batch_rows) return;
        const int wpb = h.wpb, W = h.n_windows;
        const size_t lds = h.ring > 1 ? (size_t)h.ring * h.slot16 * 16 : (size_t)h.win_rows * KP * sizeof(T);
        const int resident = n_cu() * per_cu(lds);
        const int min_windows = h.ring > 1 ? 4 : 2;   // sub-windows of the half-window schedule are half as long
        const double task_cost = (double)env_int("SCHPF_LOSS_TASK_STEPS", 8);
        // barrier-limited steps (+ 2) of every (block, window)
        std::vector<int32_t> wwork((size_t)h.n_blocks * W);
        for (int64_t b = 0; b < h.n_blocks; ++b)
            for (int w = 0; w < W; ++w) {
                int mx = 0;
                for (int v = 0; v < wpb; ++v) mx = std::max<int>(mx, h.steps[((size_t)b * wpb + v) * W + w]);
                wwork[(size_t)b * W + w] = (int32_t)schpf::tile_stored_steps(h, mx) + 2;
            }
        auto cut_points = [&](int64_t t, int parts, std::vector<int> &cuts) {   // sub-range starts of task t, + its end
            const int a0 = h.task_w0[(size_t)t], a1 = h.task_w1[(size_t)t];
            const int n = std::max(1, std::min(parts, (a1 - a0) / min_windows));
            cuts.clear();
            for (int p = 0; p <= n; ++p) cuts.push_back(a0 + (int)((int64_t)(a1 - a0) * p / n));
        };
        std::vector<int> cuts;
        std::vector<double> dur, load;
        auto model = [&](int parts) {
            dur.clear();
            for (int64_t t = 0; t < h.n_tasks; ++t) {
                cut_points(t, parts, cuts);
                const int32_t *ww = wwork.data() + (size_t)h.task_block[(size_t)t] * W;
                for (size_t p = 0; p + 1 < cuts.size(); ++p) {
                    double d = task_cost;
                    for (int w = cuts[p]; w < cuts[p + 1]; ++w) d += ww[w];
                    dur.push_back(d);
                }
            }
            std::sort(dur.begin(), dur.end(), std::greater<double>());
            load.assign((size_t)resident, 0.0);
            std::make_heap(load.begin(), load.end(), std::greater<double>());
            for (double d : dur) {
                std::pop_heap(load.begin(), load.end(), std::greater<double>());
                load.back() += d;
                std::push_heap(load.begin(), load.end(), std::greater<double>());
            }
            return *std::max_element(load.begin(), load.end());
        };
        int best_parts = 1;
        const double uncut = model(1);
        double best = uncut;
        if (env_int("SCHPF_LOSS_SPLIT", 1))
            for (int parts = 2; parts <= 8; ++parts) {
                const double m = model(parts);
                if (m < 0.97 * best) { best = m; best_parts = parts; }   // a cut has to pay for itself
            }
        td.llh_model = best;
        if (env_int("SCHPF_VERBOSE", 0))
            fprintf(stderr, "[schpf_hip]   loss pass on the %d x %d plan: %d sub-range(s) per task, modelled %.0f step units (uncut %.0f)\n",
                    h.n_major, h.n_minor, best_parts, best, uncut);
        td.llh_parts = best_parts;
        if (best_parts <= 1) return;
        std::vector<int32_t> blk, w0s, w1s, ends, order;
        std::vector<int64_t> woff;
        std::vector<double> work;
        for (int64_t t = 0; t < h.n_tasks; ++t) {
            const int b = h.task_block[(size_t)t], a1 = h.task_w1[(size_t)t];
            cut_points(t, best_parts, cuts);
            std::vector<int64_t> off((size_t)wpb);
            for (int v = 0; v < wpb; ++v) off[(size_t)v] = h.task_wave_off[(size_t)t * wpb + v];
            for (size_t p = 0; p + 1 < cuts.size(); ++p) {
                blk.push_back(b); w0s.push_back(cuts[p]); w1s.push_back(cuts[p + 1]); ends.push_back(a1);
                for (int v = 0; v < wpb; ++v) woff.push_back(off[(size_t)v]);
                double wk = 0.0;
                for (int w = cuts[p]; w < cuts[p + 1]; ++w) {
                    for (int v = 0; v < wpb; ++v)
                        off[(size_t)v] += schpf::tile_stored_steps(h, h.steps[((size_t)b * wpb + v) * W + w]) * h.gpw;
                    wk += wwork[(size_t)b * W + w];
                }
                work.push_back(wk);
            }
        }
        order.resize(blk.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return work[(size_t)x] > work[(size_t)y]; });
        td.n_llh_tasks = (int64_t)blk.size();
        upload(td.llh_block, blk, stream); upload(td.llh_w0, w0s, stream); upload(td.llh_w1, w1s, stream);
        upload(td.llh_stage_end, ends, stream); upload(td.llh_wave_off, woff, stream); upload(td.llh_order, order, stream);
        HIPCHK(hipStreamSynchronize(stream));
    }

    // the small arrays of a tile plan (its entries and steps are on the device already)
    void finish_tile(TileDev &td)
    {
        auto &h = td.host;
        const int wpb = h.wpb;
        td.n_tasks = h.n_tasks;
        td.threads = 64 * wpb;
        td.lds_bytes = h.ring > 1 ? (size_t)h.ring * h.slot16 * 16 : (size_t)h.win_rows * KP * sizeof(T);
        td.packed = h.packed;
        loss_tasks(td);
        td.n_wave_out = std::max<int64_t>(h.n_tasks, td.n_llh_tasks) * wpb;
        upload(td.block_rows, h.block_rows, stream);
        upload(td.task_block, h.task_block, stream);
        upload(td.task_w0, h.task_w0, stream);
        upload(td.task_w1, h.task_w1, stream);
        upload(td.task_wave_off, h.task_wave_off, stream);
        upload(td.task_order, h.task_order, stream);
        upload(td.pfirst, h.pfirst, stream);
        upload(td.pcount, h.pcount, stream);
        td.partials.alloc((size_t)std::max<int64_t>(h.n_partial_rows, 1) * KP * sizeof(T), true, stream);
        HIPCHK(hipStreamSynchronize(stream));
        std::vector<uint16_t>().swap(h.steps);
        std::vector<int64_t>().swap(h.task_wave_off);
    }

    // Both tile plans built by device passes over the uploaded COO (plan_device.hip): same plans,
    // bit for bit, as build_tiles(); SCHPF_DEVICE_PLAN=0 selects the host builder.
    void build_tiles_device(const int32_t *row, const int32_t *col, const float *val, bool packed_ok,
                            EarlyIndexCopy &early)
    {
        const double t0 = now_s();
        int ranges[2] = {0, 0}, half[2] = {-1, -1};
        if (!choose_ranges(row, col, ranges, half)) { ranges[0] = ranges[1] = 0; half[0] = half[1] = -1; }
        bool rc_sorted = true, cr_sorted = true;
        schpf::coo_order_flags(nnz, row, col, rc_sorted, cr_sorted);
        DevBuf d_val;
        d_val.alloc((size_t)nnz * 4);
        if (nnz > 0) HIPCHK(hipMemcpyAsync(d_val.p, val, (size_t)nnz * 4, hipMemcpyHostToDevice, stream));
        early.join();                                  // the indices went up beside the validation pass
        if (!early.error.empty()) throw HipError(early.error);
        const double t1 = now_s();
        plans_from_device_coo(early.d_row, early.d_col, d_val, rc_sorted, cr_sorted, packed_ok, ranges, half);
        // constant term of the loss, sum lgamma(x + 1) (hpf_numba.py:49-50), while the values are still resident:
        // no second trip of the values over PCIe
        gammaln_partial_sums(d_val.as<float>());
        gammaln_on_device = true;
        if (want_rows) {   // the (row, col)-sorted copy minibatches gather their rows from
            rows_col.alloc((size_t)nnz * 4); rows_val.alloc((size_t)nnz * 4);
            HIPCHK(schpf::launch_gather_by_order(tcell.order_identity ? nullptr : tcell.order_dev.as<int>(),
                                                 static_cast<const int *>(early.d_col.p), d_val.as<float>(), nnz, rows_col.as<int>(),
                                                 rows_val.as<float>(), stream));
            upload(rows_ptr, tcell.host.mptr, stream);
            rows_packed_ok = packed_ok;
            HIPCHK(hipStreamSynchronize(stream));
        }
        if (env_int("SCHPF_VERBOSE", 0))
            fprintf(stderr, "[schpf_hip]   tile plans on the device: ranges + H2D of the values %.3f s (indices: %.3f s on the "
                    "helper thread, from the start of the upload), both plans %.3f s (%.2f GB entries)\n",
                    t1 - t0, early.seconds, now_s() - t1, (tcell.entries.bytes + tgene.entries.bytes) * 1e-9);
    }

    // both tile plans from a COO that is already in HBM
    void plans_from_device_coo(const DevBuf &d_row, const DevBuf &d_col, const DevBuf &d_val, bool rc_sorted,
                               bool cr_sorted, bool packed_ok, const int ranges[2], const int half[2])
    {
        const schpf::TileShape sh_c = tile_shape(N, G, false, ranges[0], half[0]),
                               sh_g = tile_shape(G, N, true, ranges[1], half[1]);
        // the two orientations are independent (the COO is only read): the gene side on a helper thread with a
        // stream of its own, so that the builders' host round trips (run pointers, step counts, allocations) and
        // their short kernels overlap instead of adding up (SCHPF_PLAN_THREADS=1: one after the other)
        auto build_side = [&](int side, hipStream_t st) {
            TileDev &td = side == 0 ? tcell : tgene;
            void *e = nullptr, *s = nullptr, *o = nullptr;
            size_t eb = 0;
            bool presorted = side == 0 ? rc_sorted : cr_sorted;
            const int32_t *d_major = side == 0 ? d_row.as<int32_t>() : d_col.as<int32_t>();
            const int32_t *d_minor = side == 0 ? d_col.as<int32_t>() : d_row.as<int32_t>();
            const schpf::TileShape &sh = side == 0 ? sh_c : sh_g;
            int n_minor_plan = side == 0 ? G : N;
            DevBuf vminor;
            td.minor_of.release(); td.n_virtual = 0;
            if (balance_now && sh.ring <= 1 && sh.waves_per_block >= 12) {   // the balanced kernels are 1024-thread ones
                const double tb = now_s();
                schpf::BalanceGeometry geo;
                void *mo = nullptr;
                // the balancing needs ~20 bytes per nonzero of scratch and 4 bytes per (block, minor row) for good: a matrix
                // that leaves no room for that is planned by index instead (the shape is valid for either)
                bool balanced = true;
                try {
                    vminor.alloc((size_t)nnz * 4);
                    schpf::balance_windows_device((void *)st, nnz, d_major, d_minor, side == 0 ? N : G, n_minor_plan, sh,
                                                  vminor.as<int32_t>(), &mo, geo);
                } catch (const std::invalid_argument &) {
                    throw;
                } catch (const std::exception &e) {
                    (void)hipGetLastError();
                    balanced = false;
                    if (env_int("SCHPF_VERBOSE", 0))
                        fprintf(stderr, "[schpf_hip]   balanced windows, side %d: not built (%s); windows by index\n", side, e.what());
                }
                if (balanced) {
                    td.minor_of.p = mo; td.minor_of.bytes = (size_t)geo.n_blocks * geo.n_virtual * 4;
                    td.n_virtual = geo.n_virtual;
                    d_minor = vminor.as<int32_t>();
                    n_minor_plan = geo.n_virtual;
                    presorted = false;
                } else vminor.release();
                if (env_int("SCHPF_VERBOSE", 0))
                    fprintf(stderr, "[schpf_hip]   balanced windows, side %d: %d sections of %d windows, %.3f s\n", side,
                            geo.n_sections, geo.D, now_s() - tb);
            }
            schpf::build_tile_plan_device((void *)st, nnz, d_major, d_minor, d_val.as<float>(),
                                          presorted, packed_ok, side == 0 ? N : G, n_minor_plan,
                                          sh, td.host, &e, &eb, &s, &o);
            td.entries.release(); td.entries.p = e; td.entries.bytes = eb;
            td.steps.release(); td.steps.p = s; td.steps.bytes = td.host.steps.size() * 2;
            td.order_dev.release(); td.order_dev.p = o; td.order_dev.bytes = o ? (size_t)nnz * 4 : 0;
            td.order_identity = presorted;
            td.entry_slots = (int64_t)(eb / 4) / (td.host.packed ? 1 : 2);
        };
        HIPCHK(hipStreamSynchronize(stream));   // the COO is on the device before either builder reads it
        if (env_int("SCHPF_PLAN_THREADS", 2) >= 2) {
            std::exception_ptr err;
            std::thread helper([&] {
                try {
                    HIPCHK(hipSetDevice(device));
                    hipStream_t st2 = nullptr;
                    HIPCHK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
                    try {
                        build_side(1, st2);
                        HIPCHK(hipStreamSynchronize(st2));
                    } catch (...) { (void)hipStreamSynchronize(st2); (void)hipStreamDestroy(st2); throw; }
                    (void)hipStreamDestroy(st2);
                } catch (...) { err = std::current_exception(); }
            });
            try { build_side(0, stream); } catch (...) { helper.join(); throw; }
            helper.join();
            if (err) std::rethrow_exception(err);
        } else {
            build_side(0, stream);
            build_side(1, stream);
        }
        finish_tile(tcell);
        finish_tile(tgene);
        build_dual_order();
    }

    // This engine's matrix := the rows `rows` (in that order) of `source`'s, gathered on the device
    void upload_rows(schpf_ctx *source_, const int32_t *rows, int n_rows) override
    {
        Engine<T> *src = dynamic_cast<Engine<T> *>(source_);
        if (!src) throw std::invalid_argument("the source engine must have this engine's dtype");
        if (!src->rows_ptr.p || !src->have_coo) throw std::logic_error("the source keeps no rows (schpf_keep_rows before its upload)");
        if (src == this) throw std::invalid_argument("an engine cannot gather batch rows from itself");
        if (src->device != device) throw std::invalid_argument("source and batch engine must be on one device");
        if (src->G != G || src->K != K) throw std::invalid_argument("source and batch engine differ in genes or factors");
        if (n_rows != N) throw std::invalid_argument("n_rows must be the number of cells the batch engine was created with");
        if (!want_tile) throw std::invalid_argument("upload_rows needs the tile plan");
        const std::vector<int64_t> &sp = src->tcell.host.mptr;
        std::vector<int64_t> dp((size_t)n_rows + 1, 0);
        for (int i = 0; i < n_rows; ++i) {
            if (rows[i] < 0 || rows[i] >= src->N) throw std::invalid_argument("batch row out of range");
            dp[(size_t)i + 1] = dp[(size_t)i] + (sp[(size_t)rows[i] + 1] - sp[(size_t)rows[i]]);
        }
        forget_matrix();                  // a failed plan build must not leave have_coo set over empty plans
        balance_now = false;              // a batch is planned every iteration: the cheapest build
        nnz = dp[(size_t)n_rows];
        std::vector<int32_t> rv(rows, rows + n_rows);
        DevBuf d_rows, d_dp, d_row, d_col, d_val;
        upload(d_rows, rv, stream);
        upload(d_dp, dp, stream);
        d_row.alloc((size_t)nnz * 4); d_col.alloc((size_t)nnz * 4); d_val.alloc((size_t)nnz * 4);
        HIPCHK(schpf::launch_gather_rows(d_rows.as<int>(), n_rows, src->rows_ptr.as<int64_t>(), src->rows_col.as<int>(),
                                         src->rows_val.as<float>(), d_dp.as<int64_t>(), d_row.as<int>(), d_col.as<int>(),
                                         d_val.as<float>(), stream));
        use_tile = true;
        const int ranges[2] = {0, 0}, half[2] = {-1, -1};
        // rows in batch order with their columns ascending: sorted by (row, col) already
        planning_batch_rows = true;
        try { plans_from_device_coo(d_row, d_col, d_val, true, false, src->rows_packed_ok, ranges, half); }
        catch (...) { planning_batch_rows = false; throw; }
        planning_batch_rows = false;
        wave_out.alloc((size_t)std::max<int64_t>(tcell.n_wave_out, 1) * sizeof(double), true, stream);
        HIPCHK(hipStreamSynchronize(stream));
        n_rounded = 0; n_zero = 0;
        zero_row.release(); zero_col.release();
        have_loss_constants = false;      // no lgamma sum, no stored-zero list: the loss is the source engine's business
        have_coo = true;
        pending_init = 0;
        drop_graph();
        eager_since_upload = false;
    }

    // Workgroup shape of the tile sweep.  One 1024-thread workgroup per CU with a 152 KiB window
    // (fewest stagings, longest row segments => least sliced-ELL padding) unless that leaves fewer
    // than 256 (block, window) pairs per orientation; then the workgroup is halved (64 KiB windows,
    // two or more workgroups per CU) until there are, down to 256 threads.  Measured with the graph /
    // persistent launches of round 2 (profiles/r02/explore_c2_shapes.log): C2 (10k x 5k) 256-thread
    // workgroups 24.6 k -> 26.7 k it/s in f64, -4 % per iteration in f32 (128 threads: +5 %, hence the
    // floor); a 1/8 shard of C3 keeps the large workgroup in f64 (525 pairs) and halves it in f32 (-3 %).
    int cu_count = 256;
    int n_cu() const { return cu_count; }
    // workgroups of a tile sweep that fit a compute unit at once.  Sized by the LOSS pass's LDS (window + the 1 KiB
    // logarithm table behind it, run_sweep): the PHI and LLH launches of a plan must agree on the residency
    static int per_cu(size_t window_lds_bytes) { return window_lds_bytes + 1024 > 80 * 1024 ? 1 : 2; }
    void pick_workgroup(int n_major, int n_minor, int &wpb, int &lds_kb) const
    {
        wpb = env_int("SCHPF_WPB", 0);
        // <= 158 KiB: the loss pass adds a 1 KiB table behind the window and the kernels opt in to 159 KiB
        lds_kb = std::min(env_int("SCHPF_LDS_KB", 0), 158);
        const size_t row_bytes = (size_t)KP * sizeof(T);
        if (!wpb) {
            wpb = 16;
            for (;;) {
                const int kb = lds_kb ? lds_kb : (wpb >= 12 ? 152 : 64);
                const int64_t wr = std::max<int64_t>(1, (int64_t)kb * 1024 / (int64_t)row_bytes);
                const int64_t blocks = ((int64_t)n_major + (64 / LPC) * wpb - 1) / ((64 / LPC) * wpb);
                const int64_t windows = ((int64_t)n_minor + wr - 1) / wr;
                if (blocks * windows >= env_int("SCHPF_MIN_PAIRS", 256) || wpb <= 4) break;
                wpb /= 2;
            }
        }
        if (!lds_kb) lds_kb = wpb >= 12 ? 152 : 64;
    }
    // Task ranges of both orientations of the one-launch iteration from the list-schedule model of
    // plan.h choose_task_ranges (big problems with the 1024-thread workgroup on both sides; knobs that fix
    // task counts or schedules by hand switch it off).  Constants from C3 on an MI355X: a workgroup works
    // through ~1.7e11 / (K sizeof(T)) nonzeros per second (K = 20: 1.06e9 f64, 2.1e9 f32; measured 1.07 /
    // 1.9), a partial row is written and read back at ~3.5 TB/s, a task costs 3 us beside its nonzeros
    // (SCHPF_TASK_US; swept 2-16: 2-4 pick one range per cell block and 18 per gene block at C3 f64, the
    // fastest measured).  Against the former fixed counts (profiles/r02/explore_task_ranges.log), per
    // iteration: C3 f64 (6, 13) -> (1, 18) ranges -2.4 %, C3 f32 (3, 13) -> (3, 11) -3.3 %, half of C3's
    // cells -7.5 %, a quarter -3 %, the C5 share -1..2 % (f64) / -4 % (f32).
    bool choose_ranges(const int32_t *row, const int32_t *col, int ranges[2], int half[2]) const
    {
        if (!env_int("SCHPF_RANGES", 1) || (!expect_sharded && !env_int("SCHPF_DUAL", 1))) return false;
        for (const char *knob : {"SCHPF_TASKS", "SCHPF_TASKS_CELL", "SCHPF_TASKS_GENE"})
            if (getenv(knob) && *getenv(knob)) return false;
        const int half_env = env_int("SCHPF_HALF", -1);
        if (half_env >= 2) return false;
        const size_t row_bytes = (size_t)KP * sizeof(T);
        const int n_maj[2] = {N, G}, n_min[2] = {G, N};
        int64_t blocks[2], half_windows[2];
        bool half_ok[2];
        double partial_seconds[2];
        for (int s = 0; s < 2; ++s) {
            int wpb, lds_kb;
            pick_workgroup(n_maj[s], n_min[s], wpb, lds_kb);
            if (wpb < 12) return false;
            const int64_t half_rows = ((int64_t)lds_kb * 512 - 64) / (int64_t)row_bytes;
            if (half_rows < 1) return false;
            blocks[s] = ((int64_t)n_maj[s] + (64 / LPC) * wpb - 1) / ((64 / LPC) * wpb);
            half_windows[s] = ((int64_t)n_min[s] + half_rows - 1) / half_rows;
            const double per_row = (double)nnz / std::max(1, n_maj[s]) * (double)half_rows / std::max(1, n_min[s]);
            half_ok[s] = half_env != 0 && per_row >= 16.0 && !balance_now;
            partial_seconds[s] = 2.0 * (double)n_maj[s] * (double)row_bytes / 3.5e12;
        }
        const int resident = n_cu();
        // only where a launch is several rounds of workgroups (1/8 of C3: -4 % in one launch, +-0 in two): smaller
        // problems keep the rules of tile_shape
        if (blocks[0] * half_windows[0] + blocks[1] * half_windows[1] < env_int("SCHPF_RANGES_MIN", 6) * (int64_t)resident)
            return false;
        // where the nonzeros sit: a skewed matrix has heavy blocks (the planted benchmark matrix: one range per
        // cell block -- the uniform model's choice -- doubles the iteration, its heaviest block runs last)
        std::vector<double> share[2];
        const int64_t stride = std::max<int64_t>(1, nnz / 4000000);   // ~4 M samples per orientation: a few ms
        {   // the blocks the plans will cut: rows per block follow the workgroup (a forced SCHPF_WPB=12 has 12 waves)
            int wpb, lds_kb;
            pick_workgroup(N, G, wpb, lds_kb);
            share[0] = schpf::block_shares(nnz, row, N, (64 / LPC) * wpb, stride);
            pick_workgroup(G, N, wpb, lds_kb);
            share[1] = schpf::block_shares(nnz, col, G, (64 / LPC) * wpb, stride);
        }
        const schpf::RangeChoice c = schpf::choose_task_ranges(blocks, half_windows, half_ok, share, (double)nnz, resident,
                                                               1.7e11 / ((double)K * sizeof(T)), 1e-6 * env_int("SCHPF_TASK_US", 3),
                                                               partial_seconds,
                                                               expect_sharded ? 4 : 6, balance_now ? 1.0 : 1.12, 32, expect_sharded,
                                                               env_int("SCHPF_TAPER", 30) / 100.0);
        if (c.ranges[0] <= 0 || c.ranges[1] <= 0) return false;
        for (int s = 0; s < 2; ++s) { ranges[s] = c.ranges[s]; half[s] = c.half[s] ? 1 : 0; }
        // exploration: fix the ranges by hand, keep the model's schedules (tools/explore.py)
        if (env_int("SCHPF_RANGES_CELL", 0) > 0) ranges[0] = env_int("SCHPF_RANGES_CELL", 0);
        if (env_int("SCHPF_RANGES_GENE", 0) > 0) ranges[1] = env_int("SCHPF_RANGES_GENE", 0);
        if (env_int("SCHPF_VERBOSE", 0))
            fprintf(stderr, "[schpf_hip]   task ranges from the list-schedule model: cell %d (%s), gene %d (%s), %.3f ms\n",
                    ranges[0], half[0] ? "half windows" : "windows", ranges[1], half[1] ? "half windows" : "windows",
                    c.seconds * 1e3);
        return true;
    }
    schpf::TileShape tile_shape(int n_major, int n_minor, bool gene_side = false, int ranges = 0,
                                int force_half = -1) const
    {
        int wpb, lds_kb;
        pick_workgroup(n_major, n_minor, wpb, lds_kb);
        const size_t row_bytes = (size_t)KP * sizeof(T);
        schpf::TileShape sh;
        sh.lpc = LPC;
        sh.waves_per_block = wpb;
        sh.row_slots = (int)(row_bytes / 16);
        sh.bank_order = env_int("SCHPF_BANK_ORDER", 2);   // 0 minor order, 1 per row, 2 jointly per LDS pass (plan.h)
        sh.allow_packed = env_int("SCHPF_PACK", 1) != 0;
        sh.taper = env_int("SCHPF_TAPER", 30) / 100.0;   // window ranges of unequal length (plan.h tile_range_starts), per cent
        sh.win_rows = (int)std::max<size_t>(1, (size_t)lds_kb * 1024 / row_bytes);
        // tasks per orientation: a few rounds of the 256 CUs for big problems; about one round when
        // there are few (block, window) pairs (1/8 shard of C3: 1024 -> 256 tasks is 10 % faster:
        // fewer partial rows to write and to sum, no ragged second round)
        const int64_t full_rows = sh.ring > 1 ? (int64_t)sh.win_rows * (sh.ring - 1) : sh.win_rows;
        const int64_t blocks = ((int64_t)n_major + (64 / LPC) * wpb - 1) / ((64 / LPC) * wpb);
        const int64_t windows = ((int64_t)n_minor + full_rows - 1) / full_rows;
        // ... and half as many for an orientation with few blocks (the gene side of C3: 40 blocks of 512
        // genes): 1024 tasks there are 26 window ranges per block = 26 partial rows per gene to write and
        // to sum; 512 measured -3.5 % sweep, -15 % update time (profiles/r02/explore_tasks_per_side.log)
        int dflt = blocks * windows >= 2048 ? (wpb >= 12 ? 1024 : 2048) : 256;
        if (dflt >= 1024 && blocks < 64) dflt /= 2;
        sh.target_tasks = env_int("SCHPF_TASKS", dflt);
        sh.target_tasks = env_int(gene_side ? "SCHPF_TASKS_GENE" : "SCHPF_TASKS_CELL", sh.target_tasks);
        // Half-window schedule (plan.h): the window's LDS cut into two slots, refilled at the epoch boundary
        // by the window kernel itself.  Chosen per orientation where it was measured to pay
        // (profiles/r02/explore_half_window.log, explore_half_midsize.log):
        //  * rows with many nonzeros per half window -- the lock-step loss is what it removes; with ~3 per
        //    half window (C5) the second barrier per window costs more;
        //  * the 1024-thread workgroup (64 KiB windows halved lose 5 %);
        //  * tasks long enough to work ahead in: the horizon ends with the task and a task's first epoch
        //    fills both slots.  >= 6 half windows per task in the one-launch iteration (C3 8 / 16: -2..3 %;
        //    half of C3's cells 4 / 8: the cell side +1..4 % with it; 1/8: +2 %), >= 4 in the two-launch
        //    iteration of a row shard (1/8 of C3: sweeps 2 x 70 -> 2 x 63 us).
        // SCHPF_HALF = 0 / slots overrides.
        // one-nonzero-at-a-time kernels (sweep_impl.h: rows wider than 96 bytes per lane in the 1024-thread workgroup --
        // the rolling loop in float64, the plain loop in float32) count their steps in nonzeros wherever rows do not
        // work ahead
        const bool one_at_a_time = (size_t)KL * sizeof(T) > 96 && wpb >= 12;
        sh.single = one_at_a_time && env_int("SCHPF_SINGLE", 1) != 0;
        {
            const int half_env = env_int("SCHPF_HALF", -1);
            int n_slots = half_env >= 2 ? half_env : 0;
            // balanced windows are whole windows (plan.h).  Decided for the upload, not per side: the library only
            // balances matrices with < 24 nonzeros per row and whole window, i.e. < 12 per half window, where the rule
            // below (>= 16) would not pick half windows either -- a side that then is NOT balanced (too small a
            // workgroup, no memory for the scratch) gets the same whole index-cut windows it would have got without
            // balancing.  Only a forced SCHPF_BALANCE=1 on a dense matrix can lose the half-window schedule this way.
            if (balance_now && half_env < 2) n_slots = 0;
            else if (force_half >= 0) n_slots = force_half ? 2 : 0;
            else if (half_env < 0 && sh.ring <= 1 && wpb >= 12) {
                const int64_t half_rows = ((int64_t)lds_kb * 512 - 64) / (int64_t)row_bytes;
                if (half_rows >= 1) {
                    const double per_row = (double)nnz / std::max(1, n_major) * (double)half_rows / std::max(1, n_minor);
                    const int64_t half_windows = ((int64_t)n_minor + half_rows - 1) / half_rows;
                    const int64_t per_task = half_windows * blocks / std::max(1, sh.target_tasks);   // plan.cpp: wpt
                    if (per_row >= 16.0 && per_task >= (expect_sharded ? 4 : 6)) n_slots = 2;
                }
            }
            if (n_slots >= 2 && sh.ring <= 1) {
                const int slot_bytes = (int)((size_t)lds_kb * 1024 / (size_t)n_slots / 16 * 16);
                const int64_t sub_rows = ((int64_t)slot_bytes - 64) / (int64_t)row_bytes;
                if (sub_rows >= 1) {
                    sh.ring = n_slots;
                    sh.sync_stage = 1;
                    sh.slot_bytes = slot_bytes;
                    sh.win_rows = (int)sub_rows;
                    sh.single = false;   // rows work ahead: pairs
                }
            }
        }
        // workgroups in flight: one 1024-thread (152 KiB) workgroup per CU, two of the smaller ones; both
        // orientations share a launch unless the iteration is sharded (two launches, schpf_hint_sharded)
        const int per_launch = n_cu() * (wpb >= 12 ? 1 : 2);
        sh.slots = env_int("SCHPF_TASK_ROUNDING", 1) ? (expect_sharded ? per_launch : per_launch / 2) : 0;
        sh.ranges = ranges;
        return sh;
    }

    // both orientations are built concurrently on the host (each with its own thread team),
    // then uploaded one after the other on the context's stream
    void build_tiles(const int32_t *row, const int32_t *col, const float *val)
    {
        int ranges[2] = {0, 0}, half[2] = {-1, -1};
        if (!choose_ranges(row, col, ranges, half)) { ranges[0] = ranges[1] = 0; half[0] = half[1] = -1; }
        const schpf::TileShape sh_c = tile_shape(N, G, false, ranges[0], half[0]),
                               sh_g = tile_shape(G, N, true, ranges[1], half[1]);
        std::exception_ptr err;
        double secs_gene = 0.0;
        // balanced windows: the builder runs on the block's virtual numbering of the minor rows (plan.h)
        std::vector<int32_t> mo_cell, mo_gene;
        auto build_host = [&](const int32_t *major, const int32_t *minor, int n_major, int n_minor, const schpf::TileShape &sh,
                              TileDev &td, std::vector<int32_t> &mo) {
            td.n_virtual = 0;
            if (balance_now && sh.ring <= 1 && sh.waves_per_block >= 12) {
                schpf::BigVec<int32_t> vminor;
                schpf::BalanceGeometry geo;
                schpf::balance_windows_host(nnz, major, minor, n_major, n_minor, sh, vminor, mo, geo);
                td.n_virtual = geo.n_virtual;
                schpf::build_tile_plan(nnz, major, vminor.data(), val, n_major, geo.n_virtual, sh, true, td.host);
            } else {
                schpf::build_tile_plan(nnz, major, minor, val, n_major, n_minor, sh, true, td.host);
            }
        };
        std::thread side([&] {
            try {
                const double t0 = now_s();
                build_host(col, row, G, N, sh_g, tgene, mo_gene);
                secs_gene = now_s() - t0;
            } catch (...) { err = std::current_exception(); }
        });
        double secs_cell = 0.0;
        try {
            const double t0 = now_s();
            build_host(row, col, N, G, sh_c, tcell, mo_cell);
            secs_cell = now_s() - t0;
        } catch (...) { side.join(); throw; }
        side.join();
        if (err) std::rethrow_exception(err);
        upload_tile(tcell, secs_cell);
        upload_tile(tgene, secs_gene);
        tcell.minor_of.release(); tgene.minor_of.release();
        mo_cell.resize(mo_cell.size() + 16, -1);   // a list is copied in 16-byte pieces: slack behind the last one
        mo_gene.resize(mo_gene.size() + 16, -1);
        if (tcell.n_virtual) upload(tcell.minor_of, mo_cell, stream);
        if (tgene.n_virtual) upload(tgene.minor_of, mo_gene, stream);
        HIPCHK(hipStreamSynchronize(stream));
        build_dual_order();
    }

    void build_dual_order()
    {
        // Both sweeps of an iteration in one launch (kernels.h launch_tile_sweep_dual) when the two
        // plans agree on the workgroup shape: slots = all tasks of both plans, longest first
        dual_slots = 0;
        dual_order.release();
        if (env_int("SCHPF_DUAL", 1) && tcell.threads == tgene.threads && tcell.packed == tgene.packed &&
            (tcell.n_virtual != 0) == (tgene.n_virtual != 0)) {
            const auto &hc = tcell.host, &hg = tgene.host;
            std::vector<int32_t> ord;
            const int n_xcd = env_int("SCHPF_XCD", 1);
            if (n_xcd > 1 && n_cu() % n_xcd == 0) {
                // same-range tasks on one XCD at a time (plan.h xcd_launch_order); one 152 KiB workgroup per
                // compute unit, two of the smaller ones.  Opt-in (SCHPF_XCD=8): it does what it is meant to --
                // L2 hits of the sweep 47 % -> 83 % at the C5 share, 49 % -> 54 % at C3 -- and the sweep is no
                // faster for it (C5 2.48 vs 2.42 ms, C3 f32 +10 %: coarser tail): the window copy is bound by
                // the CU's own LDS-DMA rate, not by where the rows come from (tools/micro/stage_bench.hip)
                const schpf::TilePlanHost *both[2] = {&hc, &hg};
                schpf::xcd_launch_order(both, 2, n_xcd, n_cu() / n_xcd * per_cu(tcell.lds_bytes), ord);
            } else {
                ord.reserve((size_t)(hc.n_tasks + hg.n_tasks));
                size_t i = 0, j = 0;   // merge of two lists already sorted by decreasing work
                while (i < hc.task_order.size() || j < hg.task_order.size()) {
                    const bool take_cell = j >= hg.task_order.size() ||
                        (i < hc.task_order.size() &&
                         hc.task_work[(size_t)hc.task_order[i]] >= hg.task_work[(size_t)hg.task_order[j]]);
                    if (take_cell) ord.push_back(hc.task_order[i++]);
                    else ord.push_back(~hg.task_order[j++]);
                }
            }
            dual_slots = (int64_t)ord.size();
            if (dual_slots > 0) { upload(dual_order, ord, stream); HIPCHK(hipStreamSynchronize(stream)); }
        }
    }

    bool gammaln_on_device = false;
    DevBuf gammaln_part;
    void gammaln_partial_sums(const float *d_values)
    {
        const int nb = 512;
        if (!gammaln_part.p) gammaln_part.alloc(nb * sizeof(double));
        HIPCHK(schpf::launch_gammaln_sum(d_values, nnz, gammaln_part.as<double>(), nb, stream));
        HIPCHK(schpf::launch_sum_doubles(gammaln_part.as<double>(), nb, scalars.as<double>() + 1, stream));
    }

    static double now_s()
    {
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

    void upload_coo(int64_t nnz_, const int32_t *row, const int32_t *col, const void *val, int kind) override
    {
        const bool verbose = env_int("SCHPF_VERBOSE", 0) != 0;
        const double t_start = now_s();
        if (nnz_ < 0 || nnz_ >= (int64_t)1 << 31) throw std::invalid_argument("nnz must be < 2^31");
        if (kind < SCHPF_VAL_I32 || kind > SCHPF_VAL_F64) throw std::invalid_argument("unknown value kind");
        // whatever the engine held is discarded on every path below: let go of it BEFORE anything new is allocated
        // (a re-upload onto a live engine would otherwise peak at the old plans + the new indices), and an upload
        // that fails leaves an engine without a matrix, not one with half of the old one
        forget_matrix();
        // Balanced windows where the rows are sparse in a window (on average under 24 nonzeros per row and 152 KiB window,
        // both orientations: the C5 share has 7): there the lock-step padding is 45 % of the executed step slots and the
        // balancing takes a quarter of the sweep's compute away; at C3 (49 per row and window) the half-window schedule
        // already fills 0.87-0.93 of the slots and the row-list indirection of the staging costs what the rest would
        // return (profiles/r04/ab_balanced_windows.txt).  SCHPF_BALANCE=1 / 0 forces it on / off.
        {
            const int forced = env_int("SCHPF_BALANCE", -1);
            const double win = 152.0 * 1024.0 / ((double)KP * sizeof(T));
            const double per_row_cell = (double)nnz_ / std::max(1, N) * std::min(1.0, win / std::max(1, G));
            const double per_row_gene = (double)nnz_ / std::max(1, G) * std::min(1.0, win / std::max(1, N));
            const bool sparse = per_row_cell < 24.0 && per_row_gene < 24.0 && (double)G > 2.0 * win && (double)N > 2.0 * win;
            balance_now = (forced < 0 ? sparse : forced != 0) && want_tile && !want_rows && !transient;
        }
        EarlyIndexCopy early;
        const bool device_plans = want_tile && env_int("SCHPF_DEVICE_PLAN", 1);
        if (device_plans) early.start(device, nnz_, row, col);
        schpf::BigVec<float> v((size_t)nnz_);   // no serial zero-fill: written by the threaded pass below
        bool packed_ok = true;
        n_rounded = 0;
        std::vector<int32_t> zrow, zcol;         // explicitly stored zeros (rare): see zero_rate_sum()
        {   // validate + convert, in parallel slabs (first offending entry per slab is reported)
            const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(schpf::host_threads(), nnz_ / 65536 + 1));
            std::vector<int64_t> bad_val((size_t)nth, -1), bad_idx((size_t)nth, -1), rounded((size_t)nth, 0);
            std::vector<std::vector<int32_t>> zr((size_t)nth), zc((size_t)nth);
            std::vector<char> wide((size_t)nth, 0);   // a count that does not fit the packed 16-bit entry format
            std::vector<std::thread> th;
            for (int t = 0; t < nth; ++t)
                th.emplace_back([&, t] {
                    const int64_t b = nnz_ * t / nth, e = nnz_ * (t + 1) / nth;
                    for (int64_t i = b; i < e; ++i) {
                        double d;
                        switch (kind) {
                        case SCHPF_VAL_I32: d = (double)((const int32_t *)val)[i]; break;
                        case SCHPF_VAL_I64: d = (double)((const int64_t *)val)[i]; break;
                        case SCHPF_VAL_F32: d = (double)((const float *)val)[i]; break;
                        default: d = ((const double *)val)[i]; break;
                        }
                        const float f = (float)d;
                        // the reference takes any X.data (hpf_numba.py:98-112 only multiplies by it); what
                        // cannot be a Poisson observation at all (negative, NaN, inf) is refused
                        if (!(d >= 0.0 && f <= 3.0e38f) && bad_val[(size_t)t] < 0) bad_val[(size_t)t] = i;
                        if ((row[i] < 0 || row[i] >= N || col[i] < 0 || col[i] >= G) && bad_idx[(size_t)t] < 0)
                            bad_idx[(size_t)t] = i;
                        else if (d == 0.0) { zr[(size_t)t].push_back(row[i]); zc[(size_t)t].push_back(col[i]); }
                        if ((double)f != d) ++rounded[(size_t)t];
                        v[(size_t)i] = f;
                        if (!(f <= 65535.0f) || f != (float)(uint32_t)f) wide[(size_t)t] = 1;
                    }
                });
            for (auto &x : th) x.join();
            for (int t = 0; t < nth; ++t) packed_ok = packed_ok && !wide[(size_t)t];
            for (int t = 0; t < nth; ++t) {
                if (bad_idx[(size_t)t] >= 0)
                    throw std::invalid_argument("COO index out of range at entry " + std::to_string(bad_idx[(size_t)t]));
                if (bad_val[(size_t)t] >= 0)
                    throw std::invalid_argument("X.data must be finite and >= 0; offending entry " +
                                                std::to_string(bad_val[(size_t)t]));
                n_rounded += rounded[(size_t)t];
                zrow.insert(zrow.end(), zr[(size_t)t].begin(), zr[(size_t)t].end());
                zcol.insert(zcol.end(), zc[(size_t)t].begin(), zc[(size_t)t].end());
            }
        }
        n_zero = (int64_t)zrow.size();
        upload(zero_row, zrow, stream);
        upload(zero_col, zcol, stream);
        const double t_valid = now_s();
        nnz = nnz_;
        const int cpw = 64 / LPC;
        int chunk = env_int("SCHPF_CHUNK", 0);
        if (!chunk) {
            const int64_t target_waves = 16384;
            int64_t c = nnz / (target_waves * cpw);
            chunk = 16;
            while (chunk * 2 <= c && chunk < 256) chunk *= 2;
        }
        if (chunk < 2) chunk = 2;
        chunk &= ~1;
        use_tile = want_tile;
        int64_t n_out;
        if (use_tile) {
            if (device_plans) build_tiles_device(row, col, v.data(), packed_ok, early);
            else build_tiles(row, col, v.data());
            n_out = std::max(tcell.n_wave_out, tgene.n_wave_out);   // the loss pass sweeps either plan (loss_side)
        } else {
            const int wc = pick_windows((size_t)G * KP * sizeof(T), "SCHPF_WINDOWS_CELL");
            const int wg = pick_windows((size_t)N * KP * sizeof(T), "SCHPF_WINDOWS_GENE");
            build_plan(cell, nnz, row, col, v.data(), N, G, wc, chunk);
            build_plan(gene, nnz, col, row, v.data(), G, N, wg, chunk);
            n_out = cell.n_waves;
        }
        wave_out.alloc((size_t)std::max<int64_t>(n_out, 1) * sizeof(double), true, stream);

        const double t_plans = now_s();
        // constant term of the loss: sum lgamma(x + 1)   (hpf_numba.py:49-50)
        DevBuf dv;
        if (!gammaln_on_device) {          // host-built plans: the values go up once more for it
            upload(dv, v, stream);
            gammaln_partial_sums(dv.as<float>());
        }
        gammaln_on_device = false;
        HIPCHK(hipMemcpyAsync(&gammaln_sum, scalars.as<double>() + 1, sizeof(double), hipMemcpyDeviceToHost,
                              stream));
        HIPCHK(hipStreamSynchronize(stream));
        have_coo = true;
        have_loss_constants = true;
        pending_init = 0;
        drop_graph();
        eager_since_upload = false;
        if (verbose)
            fprintf(stderr, "[schpf_hip] upload_coo nnz=%lld: validate %.3f s, plans+H2D %.3f s, gammaln %.3f s (%d host threads)\n",
                    (long long)nnz, t_valid - t_start, t_plans - t_valid, now_s() - t_plans, schpf::host_threads());
    }

    DevBuf &shape_buf(int which)
    {
        switch (which) {
        case SCHPF_XI: return xi_s;
        case SCHPF_THETA: return th_s;
        case SCHPF_ETA: return eta_s;
        case SCHPF_BETA: return be_s;
        }
        throw std::invalid_argument("which must be SCHPF_XI/THETA/ETA/BETA");
    }
    DevBuf &rate_buf(int which)
    {
        switch (which) {
        case SCHPF_XI: return xi_r;
        case SCHPF_THETA: return th_r;
        case SCHPF_ETA: return eta_r;
        case SCHPF_BETA: return be_r;
        }
        throw std::invalid_argument("which must be SCHPF_XI/THETA/ETA/BETA");
    }
    size_t state_bytes(int which) const
    {
        const size_t n = (which == SCHPF_XI || which == SCHPF_THETA) ? (size_t)N : (size_t)G;
        const size_t k = (which == SCHPF_THETA || which == SCHPF_BETA) ? (size_t)K : 1;
        return n * k * sizeof(T);
    }
    void set_state(int which, const void *shape, const void *rate) override
    {
        const size_t b = state_bytes(which);
        if (shape) HIPCHK(hipMemcpyAsync(shape_buf(which).p, shape, b, hipMemcpyHostToDevice, stream));
        if (rate) HIPCHK(hipMemcpyAsync(rate_buf(which).p, rate, b, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (which == SCHPF_THETA) dirty_theta = true;
        if (which == SCHPF_BETA) dirty_beta = true;
        // the graph reads the parameters through fixed pointers: still valid; only xi/eta shapes are constants
    }
    void get_state(int which, void *shape, void *rate) override
    {
        const size_t b = state_bytes(which);
        if (shape) HIPCHK(hipMemcpyAsync(shape, shape_buf(which).p, b, hipMemcpyDeviceToHost, stream));
        if (rate) HIPCHK(hipMemcpyAsync(rate, rate_buf(which).p, b, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
    }

    int rows_per_block() const { return schpf::update_rows_per_block(K); }
    int upd_blocks(int n) const
    {
        const int groups = (n + rows_per_block() - 1) / rows_per_block();
        return std::max(1, std::min(groups, (int)UPD_BLOCKS));
    }

    // (re)build E, E[log], exp-shifted tables and column sums from the stored parameters
    void refresh_tables()
    {
        if (dirty_theta) {
            schpf::UpdateArgs<T> u{};
            u.n = N; u.K = K; u.KP = KP; u.rows_per_block = rows_per_block();
            u.shape = th_s.as<T>(); u.rate = th_r.as<T>();
            u.tab_e = th_e.as<T>(); u.tab_log = th_log.as<T>(); u.tab_exp = th_exp.as<T>();
            u.colsum_part = colpart_cell.as<double>();
            const int nb = upd_blocks(N);
            HIPCHK(schpf::launch_gamma_update(u, schpf::SRC_NONE, nb, stream));
            HIPCHK(schpf::launch_colsum_reduce(colpart_cell.as<double>(), nb, K, s_theta.as<double>(),
                                               exchange_buf.as<T>() + (size_t)G * K, sizeof(T) == 4, stream));
            dirty_theta = false;
        }
        if (dirty_beta) {
            schpf::UpdateArgs<T> u{};
            u.n = G; u.K = K; u.KP = KP; u.rows_per_block = rows_per_block();
            u.shape = be_s.as<T>(); u.rate = be_r.as<T>();
            u.tab_e = be_e.as<T>(); u.tab_log = be_log.as<T>(); u.tab_exp = be_exp.as<T>();
            u.colsum_part = colpart_gene.as<double>();
            const int nb = upd_blocks(G);
            HIPCHK(schpf::launch_gamma_update(u, schpf::SRC_NONE, nb, stream));
            HIPCHK(schpf::launch_colsum_reduce(colpart_gene.as<double>(), nb, K, s_beta.as<double>(), nullptr, 0,
                                               stream));
            dirty_beta = false;
        }
    }

    schpf::SweepArgs<T> sweep_args(PlanDev &pd, const DevBuf &tab_major, const DevBuf &tab_minor,
                                   const DevBuf &log_major, const DevBuf &log_minor)
    {
        schpf::SweepArgs<T> a{};
        a.entries = pd.entries.as<uint4>();
        a.slice_off = pd.slice_off.as<int64_t>();
        a.slice_steps = pd.slice_steps.as<int>();
        a.chunk_major = pd.chunk_major.as<int>();
        a.chunk_natid = pd.chunk_natid.as<int>();
        a.wave_slice = pd.wave_slice.as<int>();
        a.tab_major = tab_major.as<T>();
        a.tab_minor = tab_minor.as<T>();
        a.log_major = log_major.as<T>();
        a.log_minor = log_minor.as<T>();
        a.partials = pd.partials.as<T>();
        a.wave_out = wave_out.as<double>();
        a.K = K;
        return a;
    }

    schpf::TileArgs<T> tile_args(TileDev &td, const DevBuf &tab_major, const DevBuf &tab_minor,
                                 const DevBuf &log_major, const DevBuf &log_minor, int n_minor)
    {
        schpf::TileArgs<T> a{};
        a.entries = td.entries.p;
        a.steps = td.steps.as<uint16_t>();
        a.block_rows = td.block_rows.as<int>();
        a.task_block = td.task_block.as<int>();
        a.task_w0 = td.task_w0.as<int>();
        a.task_w1 = td.task_w1.as<int>();
        a.task_wave_off = td.task_wave_off.as<int64_t>();
        a.task_order = nullptr;   // natural order (plan.cpp)
        a.tab_major = tab_major.as<T>();
        a.tab_minor = tab_minor.as<T>();
        a.log_major = log_major.as<T>();
        a.log_minor = log_minor.as<T>();
        a.partials = td.partials.as<T>();
        a.wave_out = wave_out.as<double>();
        a.K = K; a.n_minor = td.n_virtual ? td.n_virtual : n_minor; a.n_windows = td.host.n_windows; a.win_rows = td.host.win_rows;
        a.minor_of = td.n_virtual ? td.minor_of.as<int>() : nullptr;
        a.n_virtual = td.n_virtual;
        a.wpb = td.host.wpb;
        a.ring = td.host.ring; a.slot_bytes = td.host.slot16 * 16; a.sync_stage = td.host.sync_stage;
        a.single = td.host.single ? 1 : 0;
        a.clock_probe = clock_probe.as<unsigned long long>();
        return a;
    }

    // one sweep of either plan kind.  side 0: major = cell, side 1: major = gene.
    void run_sweep(int side, int mode, uint64_t seed = 0)
    {
        const bool cellside = side == 0;
        const DevBuf &tmaj = mode == schpf::MODE_LLH ? (cellside ? th_e : be_e) : (cellside ? th_exp : be_exp);
        const DevBuf &tmin = mode == schpf::MODE_LLH ? (cellside ? be_e : th_e) : (cellside ? be_exp : th_exp);
        const DevBuf &lmaj = cellside ? th_log : be_log;
        const DevBuf &lmin = cellside ? be_log : th_log;
        if (use_tile) {
            TileDev &td = cellside ? tcell : tgene;
            auto a = tile_args(td, tmaj, tmin, lmaj, lmin, cellside ? G : N);
            a.seed = seed; a.major_is_cell = cellside ? 1 : 0;
            int64_t n_tasks = td.n_tasks;
            const bool cut = mode == schpf::MODE_LLH && td.n_llh_tasks > 0;   // the loss pass's finer tasks (loss_tasks)
            if (cut) {
                a.task_block = td.llh_block.as<int>(); a.task_w0 = td.llh_w0.as<int>(); a.task_w1 = td.llh_w1.as<int>();
                a.task_stage_end = td.llh_stage_end.as<int>(); a.task_wave_off = td.llh_wave_off.as<int64_t>();
                n_tasks = td.n_llh_tasks;
            }
            if (mode != schpf::MODE_RANDOM && env_int("SCHPF_PERSISTENT", 1)) {   // see step_local
                a.queue = dual_queue.as<int>();
                a.resident = n_cu() * per_cu(td.lds_bytes);
                a.task_order = cut ? td.llh_order.as<int>() : td.task_order.as<int>();
            } else if (cut) a.task_order = td.llh_order.as<int>();
            // the loss pass keeps a 1 KiB logarithm table behind the window (sweep_impl.h LlhAccumulator)
            a.llh_tab_off = (int)((td.lds_bytes + 15) & ~(size_t)15);
            const size_t lds = mode == schpf::MODE_LLH ? (size_t)a.llh_tab_off + 1024 : td.lds_bytes;
            HIPCHK(schpf::launch_tile_sweep<T>(a, NV, LPC, mode, td.packed ? 1 : 0, n_tasks, td.threads, lds, stream));
        } else {
            PlanDev &pd = cellside ? cell : gene;
            auto a = sweep_args(pd, tmaj, tmin, lmaj, lmin);
            if (mode == schpf::MODE_RANDOM)
                HIPCHK(schpf::launch_random_phi<T>(a, NV, LPC, seed, cellside ? 1 : 0, pd.n_waves, stream));
            else
                HIPCHK(schpf::launch_sweep<T>(a, NV, LPC, mode, pd.n_waves, stream));
        }
    }

    // where the update kernel finds a side's accumulated chunk/task partials
    void partial_source(int side, schpf::UpdateArgs<T> &u, int &src)
    {
        if (use_tile) {
            TileDev &td = side == 0 ? tcell : tgene;
            src = schpf::SRC_STRIDED;
            u.partials = td.partials.as<T>(); u.pfirst = td.pfirst.as<int>(); u.pcount = td.pcount.as<int>();
            u.pstride = td.host.pstride;
        } else {
            PlanDev &pd = side == 0 ? cell : gene;
            src = schpf::SRC_PARTIALS;
            u.partials = pd.partials.as<T>(); u.cptr = pd.cptr.as<int>();
        }
    }

    void need_coo() const
    {
        if (!have_coo) throw std::logic_error("no count matrix uploaded (schpf_upload_coo)");
    }

    // (major, minor)-sorted position -> position in the caller's COO, on the device
    const int *order_of(int side, DevBuf &scratch)
    {
        if (!use_tile) { upload(scratch, side == 0 ? cell.host.order : gene.host.order, stream); return scratch.as<int>(); }
        TileDev &td = side == 0 ? tcell : tgene;
        if (td.order_dev.p) return td.order_dev.as<int>();
        if (td.order_identity) {
            std::vector<int32_t> iota((size_t)nnz);
            for (int64_t j = 0; j < nnz; ++j) iota[(size_t)j] = (int32_t)j;
            upload(scratch, iota, stream);
            HIPCHK(hipStreamSynchronize(stream));   // iota dies with this scope
            return scratch.as<int>();
        }
        upload(scratch, td.host.order, stream);
        return scratch.as<int>();
    }

    void init_phi_host(const double *xphi) override
    {
        need_coo();
        DevBuf dx, ord, mp;
        dx.alloc((size_t)nnz * K * sizeof(double));
        HIPCHK(hipMemcpyAsync(dx.p, xphi, (size_t)nnz * K * sizeof(double), hipMemcpyHostToDevice, stream));
        dense_cell.alloc((size_t)N * K * sizeof(T));
        const int *ord_c = order_of(0, ord);
        upload(mp, use_tile ? tcell.host.mptr : cell.host.mptr, stream);
        HIPCHK(schpf::launch_segment_sum<T>(dx.as<double>(), ord_c, mp.as<int64_t>(), N, K,
                                            dense_cell.as<T>(), stream));
        HIPCHK(hipStreamSynchronize(stream));
        const int *ord_g = order_of(1, ord);
        upload(mp, use_tile ? tgene.host.mptr : gene.host.mptr, stream);
        HIPCHK(schpf::launch_segment_sum<T>(dx.as<double>(), ord_g, mp.as<int64_t>(), G, K,
                                            exchange_buf.as<T>(), stream));
        HIPCHK(hipStreamSynchronize(stream));
        pending_init = 1;
    }

    void init_phi_device(uint64_t seed) override
    {
        need_coo();
        // a rank of a communicator numbers its cells from 0 like every other rank: without this, local cell i of
        // every shard would draw the same responsibilities for a gene
        if (comm && comm_world > 1) seed += 0x9E3779B97F4A7C15ull * (uint64_t)(comm_rank + 1);
        run_sweep(0, schpf::MODE_RANDOM, seed);
        run_sweep(1, schpf::MODE_RANDOM, seed);
        pending_init = 2;
    }

    void step_local(unsigned flags_) override
    {
        need_coo();
        refresh_tables();
        const bool freeze = flags_ & SCHPF_FREEZE_GENES;
        const bool sharded = flags_ & SCHPF_SHARDED;
        const bool only_gene = flags_ & SCHPF_LOCAL_GENE, only_cell = flags_ & SCHPF_LOCAL_CELL;
        const bool do_cell = !only_gene || only_cell, do_gene = !only_cell || only_gene;
        if (pending_init == 0 && use_tile && dual_slots > 0 && do_gene && do_cell && !freeze) {
            // both sweeps read the same old tables: one launch (timed as kind 0, see schpf_profile_read)
            ScopedTimer tm(prof, stream, 0);
            auto ac = tile_args(tcell, th_exp, be_exp, th_log, be_log, G);
            auto ag = tile_args(tgene, be_exp, th_exp, be_log, th_log, N);
            ac.major_is_cell = 1; ag.major_is_cell = 0;
            // persistent workgroups (SCHPF_PERSISTENT=0: one workgroup per slot): as many as the device holds at
            // once draw the slots of the longest-first list from a counter -- no workgroup teardown / launch
            // between the ~6 tasks of a compute unit and whoever is free takes the next task: C3 sweep
            // -2 % f64, -5 % f32, nothing at C2 / the C5 share (profiles/r02/explore_persistent.log)
            const size_t lds = std::max(tcell.lds_bytes, tgene.lds_bytes);
            int *queue = nullptr;
            int resident = 0;
            if (env_int("SCHPF_PERSISTENT", 1)) {
                queue = dual_queue.as<int>();
                resident = n_cu() * per_cu(lds);
            }
            HIPCHK(schpf::launch_tile_sweep_dual<T>(ac, ag, dual_order.as<int>(), NV, LPC, tcell.packed ? 1 : 0,
                                                    dual_slots, tcell.threads, lds, queue, resident, stream));
            tm.stop();
        } else if (pending_init == 0) {
            if (do_gene && !freeze) {
                ScopedTimer tm(prof, stream, 1);
                run_sweep(1, schpf::MODE_PHI);
                tm.stop();
            }
            if (do_cell) {
                ScopedTimer tm(prof, stream, 0);
                run_sweep(0, schpf::MODE_PHI);
                tm.stop();
            }
        }
        if (sharded && !freeze && pending_init != 1 && do_gene) {
            // fixed-order reduction of this rank's gene-side partials into the exchange buffer
            if (use_tile)
                HIPCHK(schpf::launch_combine_strided<T>(tgene.partials.as<T>(), tgene.pfirst.as<int>(),
                                                        tgene.pcount.as<int>(), tgene.host.pstride, G, K, KP,
                                                        exchange_buf.as<T>(), stream));
            else
                HIPCHK(schpf::launch_combine_partials<T>(gene.partials.as<T>(), gene.cptr.as<int>(), G, K, KP,
                                                         exchange_buf.as<T>(), stream));
        }
    }

    void exchange(void **p, int64_t *count) override
    {
        *p = exchange_buf.p;
        *count = (int64_t)G * K + K;
    }

    void step_finish(unsigned flags_) override
    {
        need_coo();
        const bool freeze = flags_ & SCHPF_FREEZE_GENES;
        const bool simultaneous = flags_ & SCHPF_SIMULTANEOUS;
        const bool sharded = flags_ & SCHPF_SHARDED;
        ScopedTimer tm(prof, stream, 3);
        const bool cells_first = flags_ & SCHPF_CELLS_FIRST;
        // default ordering on a small problem: no reduce launches (BASELINE C2: 2 of its 5 launches)
        const bool fuse = !sharded && !freeze && !simultaneous && !cells_first && env_int("SCHPF_FUSE_SUMS", 1) &&
                          (int64_t)upd_blocks(N) * K <= 16384 && (int64_t)upd_blocks(G) * K <= 16384;
        if (!fuse && sums_stale) {   // s_theta / s_beta from the partials the last fused iteration left
            HIPCHK(schpf::launch_colsum_reduce(colpart_cell.as<double>(), upd_blocks(N), K, s_theta.as<double>(),
                                               exchange_buf.as<T>() + (size_t)G * K, sizeof(T) == 4, stream));
            HIPCHK(schpf::launch_colsum_reduce(colpart_gene.as<double>(), upd_blocks(G), K, s_beta.as<double>(), nullptr,
                                               0, stream));
            sums_stale = false;
        }
        // sharded: the all-reduced sum_i E[theta_ik] (old theta) is the tail of the exchange buffer; the
        // gene update reads it from there (s_other_t)
        auto gene_update = [&] {
        if (!freeze) {  // gene block, scHPF_.py:697-704 (or :668-673 + :682-685)
            schpf::UpdateArgs<T> u{};
            u.n = G; u.K = K; u.KP = KP; u.rows_per_block = rows_per_block();
            int src;
            if (sharded || pending_init == 1) { src = schpf::SRC_DENSE; u.dense = exchange_buf.as<T>(); }
            else partial_source(1, u, src);
            u.prior_shape = c;
            u.cap_shape = eta_s.as<T>(); u.cap_rate = eta_r.as<T>();
            u.s_other = s_theta.as<double>();
            if (sharded) u.s_other_t = exchange_buf.as<T>() + (size_t)G * K;
            if (fuse) { u.s_other_part = colpart_cell.as<double>(); u.s_other_nb = upd_blocks(N); }
            u.cap_prior_rate = dp;
            u.shape = be_s.as<T>(); u.rate = be_r.as<T>(); u.cap_rate_out = eta_r.as<T>();
            u.tab_e = be_e.as<T>(); u.tab_log = be_log.as<T>(); u.tab_exp = be_exp.as<T>();
            u.colsum_part = colpart_gene.as<double>();
            const int nb = upd_blocks(G);
            HIPCHK(schpf::launch_gamma_update(u, src, nb, stream));
            if (!fuse)
                HIPCHK(schpf::launch_colsum_reduce(colpart_gene.as<double>(), nb, K, s_beta_next.as<double>(), nullptr,
                                                   0, stream));
        }
        };
        auto cell_update = [&] {
        {  // cell block, scHPF_.py:706-714 (or :675-680)
            schpf::UpdateArgs<T> u{};
            u.n = N; u.K = K; u.KP = KP; u.rows_per_block = rows_per_block();
            int src;
            if (pending_init == 1) { src = schpf::SRC_DENSE; u.dense = dense_cell.as<T>(); }
            else partial_source(0, u, src);
            u.prior_shape = a;
            u.cap_shape = xi_s.as<T>(); u.cap_rate = xi_r.as<T>();
            // theta.rate uses the beta just updated (scHPF_.py:711-713) unless the updates are
            // simultaneous (:677-679) or the genes are frozen
            u.s_other = (freeze || simultaneous || cells_first) ? s_beta.as<double>() : s_beta_next.as<double>();
            if (fuse) { u.s_other_part = colpart_gene.as<double>(); u.s_other_nb = upd_blocks(G); }
            u.cap_prior_rate = bp;
            u.shape = th_s.as<T>(); u.rate = th_r.as<T>(); u.cap_rate_out = xi_r.as<T>();
            u.tab_e = th_e.as<T>(); u.tab_log = th_log.as<T>(); u.tab_exp = th_exp.as<T>();
            u.colsum_part = colpart_cell.as<double>();
            const int nb = upd_blocks(N);
            HIPCHK(schpf::launch_gamma_update(u, src, nb, stream));
            if (!fuse)
                HIPCHK(schpf::launch_colsum_reduce(colpart_cell.as<double>(), nb, K, s_theta.as<double>(),
                                                   exchange_buf.as<T>() + (size_t)G * K, sizeof(T) == 4, stream));
        }
        };
        if (cells_first) { cell_update(); gene_update(); }   // minibatch order: theta first, beta from the NEW theta
        else { gene_update(); cell_update(); }
        if (!freeze) { std::swap(s_beta.p, s_beta_next.p); beta_parity ^= 1; }
        if (fuse) sums_stale = true;
        if (pending_init == 1) dense_cell.release();
        pending_init = 0;
        tm.stop();
    }


    void loss_terms(double *llh, double *gl, int64_t *nnz_out) override
    {
        need_coo();
        if (!have_loss_constants)
            throw std::logic_error("this engine holds gathered batch rows (schpf_upload_rows): evaluate the loss on the source");
        refresh_tables();
        ScopedTimer tm(prof, stream, 2);
        const int side = loss_side();
        run_sweep(side, schpf::MODE_LLH);
        double *res = loss_host ? loss_host : scalars.as<double>();   // pinned host memory is device-addressable as it is
        HIPCHK(schpf::launch_sum_doubles(wave_out.as<double>(),
                                         use_tile ? (side ? tgene.n_wave_out : tcell.n_wave_out) : cell.n_waves,
                                         res, stream));
        // explicitly stored zeros look like padding to the sweeps (weight 0, which is what they
        // contribute to the shape updates, hpf_numba.py:97-112), but the reference's loss counts
        // them: x log r - r - lgamma(x+1) = -r (hpf_numba.py:43-50)
        if (n_zero > 0)
            HIPCHK(schpf::launch_zero_rate_sum<T>(zero_row.as<int>(), zero_col.as<int>(), n_zero, th_e.as<T>(),
                                                  be_e.as<T>(), K, KP, res + 2, stream));
        tm.stop();
        double h[3] = {0.0, 0.0, 0.0};
        if (!loss_host) HIPCHK(hipMemcpyAsync(h, scalars.p, 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (loss_host) { h[0] = loss_host[0]; h[2] = loss_host[2]; }
        *llh = n_zero > 0 ? h[0] - h[2] : h[0];
        *gl = gammaln_sum;
        *nnz_out = nnz;
    }

    // The loss pass sweeps ONE plan, either will do (both hold every nonzero; r = sum_k E[theta] E[beta] is symmetric).
    // The cell-side plan unless it has too few tasks to fill the device and the gene-side plan has more: the task ranges
    // are chosen for the iteration's merged launch, where C3 f64 gets one range per cell block = 196 tasks for 256
    // compute units (loss pass 421 us on the cell plan; the gene plan's 640 tapered tasks: see DESIGN 9).
    int loss_side() const
    {
        const int forced = env_int("SCHPF_LOSS_SIDE", -1);
        if (forced == 0 || forced == 1) return use_tile ? forced : 0;
        if (!use_tile || wave_out.bytes < (size_t)tgene.n_wave_out * sizeof(double)) return 0;
        // the plan with the shorter modelled pass (loss_tasks); the gene side's steps are worth a little more: its windows
        // are shorter (more stagings per nonzero than the model's two step units per window say)
        if (tcell.llh_model > 0.0 && tgene.llh_model > 0.0) return tgene.llh_model * 1.05 < tcell.llh_model ? 1 : 0;
        const int64_t resident = (int64_t)n_cu() * per_cu(tcell.lds_bytes);
        return (tcell.n_tasks < 2 * resident && tgene.n_tasks > tcell.n_tasks) ? 1 : 0;
    }

    void upload_info(int64_t info[4]) override
    {
        info[0] = nnz; info[1] = n_rounded; info[2] = n_zero;
        info[3] = (use_tile ? (tcell.packed ? 1 : 0) : 0) | (rows_ptr.p ? 2 : 0);   // bit 1: a row-sorted copy is kept
    }

    // Shader clock the chip sustained under the sweep launches since the last read (tile plans; 0 launches: unknown).
    void profile_clock(double *shader_mhz, int64_t *launches) override
    {
        unsigned long long h[5] = {0, 0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(h, clock_probe.p, sizeof h, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemsetAsync(clock_probe.p, 0, sizeof h, stream));
        HIPCHK(hipStreamSynchronize(stream));
        int khz = 0;   // rate of s_memrealtime
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) {
            (void)hipGetLastError();
            khz = 100000;
        }
        *launches = (int64_t)h[4];
        *shader_mhz = h[1] ? (double)h[0] / (double)h[1] * (double)khz * 1e-3 : 0.0;
    }

    // LDS bytes the tasks of a tile plan stage: every (sub-)window of a task's range exactly once -- the half-window
    // schedule fills all slots at the first epoch and afterwards only the slot the last epoch owned, never beyond the
    // task's last window (sweep_impl.h, the window loop)
    int64_t staged_bytes(const TileDev &td, int n_minor) const
    {
        const schpf::TilePlanHost &P = td.host;
        const int nm = td.n_virtual ? td.n_virtual : n_minor;
        int64_t rows = 0;
        for (size_t t = 0; t < P.task_w0.size(); ++t)
            for (int w = P.task_w0[t]; w < P.task_w1[t]; ++w) rows += std::max(0, std::min(P.win_rows, nm - w * P.win_rows));
        return rows * (int64_t)KP * (int64_t)sizeof(T);
    }
    // Bytes one iteration moves through the LDS and streams from HBM, from the plans (tile plans; else zeros):
    //   [0] LDS reads of the nonzeros alone: every nonzero reads one table row of KP values per orientation
    //   [1] ... of the stored step slots (padding slots execute the same reads)
    //   [2] [3] LDS writes of the window stagings, cell / gene side
    //   [4] entry stream of both plans in HBM   [5] partial rows written
    void sweep_bytes(int64_t info[8]) override
    {
        for (int i = 0; i < 8; ++i) info[i] = 0;
        if (!use_tile || !have_coo) return;
        const int64_t row = (int64_t)KP * (int64_t)sizeof(T);
        info[0] = 2 * nnz * row;
        info[1] = (tcell.entry_slots + tgene.entry_slots) * row;
        info[2] = staged_bytes(tcell, G);
        info[3] = staged_bytes(tgene, N);
        info[4] = (tcell.entry_slots + tgene.entry_slots) * (tcell.packed ? 4 : 8);
        info[5] = (tcell.host.n_partial_rows + tgene.host.n_partial_rows) * row;
        // what the loss pass will sweep (loss_side, loss_tasks): so that a report can say which plan and cut the model chose
        const int side = loss_side();
        info[6] = side;
        info[7] = (side ? tgene : tcell).n_llh_tasks > 0 ? (side ? tgene : tcell).n_llh_tasks : (side ? tgene : tcell).n_tasks;
    }

    void plan_info(int64_t info[16]) override
    {
        info[0] = KP; info[1] = KL; info[2] = LPC;
        info[12] = use_tile ? tcell.host.ring : 0; info[13] = use_tile ? tgene.host.ring : 0;
        info[14] = use_tile ? tcell.host.slot16 * 16 : 0; info[15] = use_tile ? tcell.host.wpb : 0;
        if (use_tile) {
            info[3] = -tcell.host.win_rows;            // negative: tile plan, rows per LDS window
            info[4] = tcell.host.n_windows; info[5] = tgene.host.n_windows;
            info[6] = tcell.host.n_partial_rows; info[7] = tgene.host.n_partial_rows;
            info[8] = tcell.n_tasks; info[9] = tgene.n_tasks;
            info[10] = tcell.entry_slots; info[11] = tgene.entry_slots;
        } else {
            info[3] = cell.host.chunk_len;
            info[4] = cell.host.n_windows; info[5] = gene.host.n_windows;
            info[6] = cell.n_chunks; info[7] = gene.n_chunks;
            info[8] = cell.n_waves; info[9] = gene.n_waves;
            info[10] = cell.entry_slots; info[11] = gene.entry_slots;
        }
    }
};

// ------------------------------------------------------------- stateless helpers
struct TempStream {
    hipStream_t st = nullptr;
    TempStream() { HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); }
    ~TempStream() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } }
};

template <typename U> void h2d(DevBuf &b, const void *src, size_t count, hipStream_t st)
{
    b.alloc(count * sizeof(U));
    if (count) HIPCHK(hipMemcpyAsync(b.p, src, count * sizeof(U), hipMemcpyHostToDevice, st));
}
void d2h(void *dst, const DevBuf &b, size_t bytes, hipStream_t st)
{
    if (bytes) HIPCHK(hipMemcpyAsync(dst, b.p, bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
}

void check_indices(int64_t nnz, const int32_t *ix, int n, const char *what)
{
    for (int64_t i = 0; i < nnz; ++i)
        if (ix[i] < 0 || ix[i] >= n) throw std::invalid_argument(std::string(what) + " index out of range");
}

template <typename T>
void xphi_or_llh(bool want_llh, int64_t nnz, int N, int G, int K, const void *x, const int32_t *row,
                 const int32_t *col, const void *ths, const void *thr, const void *bes, const void *ber, void *out)
{
    check_indices(nnz, row, N, "row");
    check_indices(nnz, col, G, "col");
    TempStream ts;
    DevBuf dx, dr, dc, a, b, c, d, tt, tb, o;
    h2d<T>(dx, x, (size_t)nnz, ts.st);
    h2d<int32_t>(dr, row, (size_t)nnz, ts.st);
    h2d<int32_t>(dc, col, (size_t)nnz, ts.st);
    h2d<T>(a, ths, (size_t)N * K, ts.st); h2d<T>(b, thr, (size_t)N * K, ts.st);
    h2d<T>(c, bes, (size_t)G * K, ts.st); h2d<T>(d, ber, (size_t)G * K, ts.st);
    tt.alloc((size_t)N * K * sizeof(T)); tb.alloc((size_t)G * K * sizeof(T));
    if (want_llh) {
        HIPCHK(schpf::launch_ratio<T>(a.as<T>(), b.as<T>(), (int64_t)N * K, tt.as<T>(), ts.st));
        HIPCHK(schpf::launch_ratio<T>(c.as<T>(), d.as<T>(), (int64_t)G * K, tb.as<T>(), ts.st));
        o.alloc((size_t)nnz * sizeof(T));
        HIPCHK(schpf::launch_llh_coo<T>(dx.as<T>(), dr.as<int>(), dc.as<int>(), tt.as<T>(), tb.as<T>(), nnz, K,
                                        o.as<T>(), ts.st));
        d2h(out, o, (size_t)nnz * sizeof(T), ts.st);
    } else {
        HIPCHK(schpf::launch_elog<T>(a.as<T>(), b.as<T>(), (int64_t)N * K, tt.as<T>(), ts.st));
        HIPCHK(schpf::launch_elog<T>(c.as<T>(), d.as<T>(), (int64_t)G * K, tb.as<T>(), ts.st));
        o.alloc((size_t)nnz * K * sizeof(T));
        HIPCHK(schpf::launch_xphi_coo<T>(dx.as<T>(), dr.as<int>(), dc.as<int>(), tt.as<T>(), tb.as<T>(), nnz, K,
                                         o.as<T>(), ts.st));
        d2h(out, o, (size_t)nnz * K * sizeof(T), ts.st);
    }
}

template <typename T>
void shape_update(int64_t nnz, int K, const void *xphi, const int32_t *keep, int nkeep, double prior, void *out)
{
    check_indices(nnz, keep, nkeep, "keep");
    schpf::BigVec<int32_t> order;
    std::vector<int64_t> ptr;
    schpf::counting_sort_positions(nnz, keep, nkeep, order, ptr);
    TempStream ts;
    DevBuf dx, dord, dptr, o;
    h2d<T>(dx, xphi, (size_t)nnz * K, ts.st);
    upload(dord, order, ts.st);
    upload(dptr, ptr, ts.st);
    o.alloc((size_t)nkeep * K * sizeof(T));
    HIPCHK(schpf::launch_shape_update<T>(dx.as<T>(), dord.as<int>(), dptr.as<int64_t>(), nkeep, K, prior,
                                         o.as<T>(), ts.st));
    d2h(out, o, (size_t)nkeep * K * sizeof(T), ts.st);
}

template <typename T>
void rate_update(int n, int m, int K, const void *ps, const void *pr, const void *os, const void *orr, void *out)
{
    if (K < 1 || K > 256) throw std::invalid_argument("nfactors must be in [1, 256]");
    TempStream ts;
    DevBuf a, b, c, d, part, S, o;
    h2d<T>(a, ps, (size_t)n, ts.st); h2d<T>(b, pr, (size_t)n, ts.st);
    h2d<T>(c, os, (size_t)m * K, ts.st); h2d<T>(d, orr, (size_t)m * K, ts.st);
    const int rb = 256 / K;
    const int nb = std::max(1, std::min((m + rb - 1) / rb, 512));
    part.alloc((size_t)nb * K * sizeof(double));
    S.alloc((size_t)K * sizeof(double));
    HIPCHK(schpf::launch_ratio_colsum<T>(c.as<T>(), d.as<T>(), m, K, part.as<double>(), nb, ts.st));
    HIPCHK(schpf::launch_colsum_reduce(part.as<double>(), nb, K, S.as<double>(), nullptr, 0, ts.st));
    o.alloc((size_t)n * K * sizeof(T));
    HIPCHK(schpf::launch_rate_update<T>(a.as<T>(), b.as<T>(), S.as<double>(), n, K, o.as<T>(), ts.st));
    d2h(out, o, (size_t)n * K * sizeof(T), ts.st);
}

template <typename T> void capacity_rate(int n, int K, const void *shape, const void *rate, double prior, void *out)
{
    TempStream ts;
    DevBuf a, b, o;
    h2d<T>(a, shape, (size_t)n * K, ts.st); h2d<T>(b, rate, (size_t)n * K, ts.st);
    o.alloc((size_t)n * sizeof(T));
    HIPCHK(schpf::launch_capacity_rate<T>(a.as<T>(), b.as<T>(), n, K, prior, o.as<T>(), ts.st));
    d2h(out, o, (size_t)n * sizeof(T), ts.st);
}

void special_array(bool gammaln, int64_t n, const double *x, double *out)
{
    TempStream ts;
    DevBuf a, o;
    h2d<double>(a, x, (size_t)n, ts.st);
    o.alloc((size_t)n * sizeof(double));
    if (gammaln) HIPCHK(schpf::launch_gammaln_array(a.as<double>(), n, o.as<double>(), ts.st));
    else HIPCHK(schpf::launch_digamma_array(a.as<double>(), n, o.as<double>(), ts.st));
    d2h(out, o, (size_t)n * sizeof(double), ts.st);
}

bool bad_dtype(int dtype) { return dtype != SCHPF_F32 && dtype != SCHPF_F64; }

}  // namespace

// ------------------------------------------------------------------------- C ABI
// SCHPF_BACKTRACE=1 (debugging aid, read when the library is loaded): a SIGSEGV / SIGBUS / SIGABRT prints the native call
// stack (glibc backtrace: module + offset per frame, resolvable with addr2line against this .so) before the previous
// handler -- Python's faulthandler under pytest -- runs.  The GPU boxes have no debugger.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
namespace {
struct sigaction g_prev_segv, g_prev_bus, g_prev_abrt;
void crash_trace(int sig, siginfo_t *info, void *uctx)
{
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "[schpf_hip] fatal signal, native stack:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    struct sigaction *prev = sig == SIGSEGV ? &g_prev_segv : sig == SIGBUS ? &g_prev_bus : &g_prev_abrt;
    sigaction(sig, prev, nullptr);          // hand over: the previous handler (or the default action) sees the re-raised signal
    raise(sig);
    (void)info; (void)uctx;
}
struct CrashTraceInstaller {
    CrashTraceInstaller()
    {
        const char *e = getenv("SCHPF_BACKTRACE");
        if (!e || !*e || *e == '0') return;
        struct sigaction sa;
        std::memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = crash_trace;
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigaction(SIGSEGV, &sa, &g_prev_segv);
        sigaction(SIGBUS, &sa, &g_prev_bus);
        sigaction(SIGABRT, &sa, &g_prev_abrt);
    }
} g_crash_trace_installer;
}  // namespace

extern "C" {

const char *schpf_last_error(void) { return g_err.c_str(); }
const char *schpf_version(void)
{
    return "schpf_hip 0.3 (gfx950)";
}

int schpf_device_count(int *count)
{
    return guarded([&] {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
        *count = n;
    });
}

int schpf_digamma(int64_t n, const double *x, double *out) { return guarded([&] { special_array(false, n, x, out); }); }
int schpf_gammaln(int64_t n, const double *x, double *out) { return guarded([&] { special_array(true, n, x, out); }); }

int schpf_xphi(int dtype, int64_t nnz, int N, int G, int K, const void *x, const int32_t *row, const int32_t *col,
               const void *ths, const void *thr, const void *bes, const void *ber, void *out)
{
    if (bad_dtype(dtype)) return fail("dtype must be SCHPF_F32 or SCHPF_F64");
    return guarded([&] {
        if (dtype == SCHPF_F64) xphi_or_llh<double>(false, nnz, N, G, K, x, row, col, ths, thr, bes, ber, out);
        else xphi_or_llh<float>(false, nnz, N, G, K, x, row, col, ths, thr, bes, ber, out);
    });
}
int schpf_pois_llh_pointwise(int dtype, int64_t nnz, int N, int G, int K, const void *x, const int32_t *row,
                             const int32_t *col, const void *ths, const void *thr, const void *bes,
                             const void *ber, void *out)
{
    if (bad_dtype(dtype)) return fail("dtype must be SCHPF_F32 or SCHPF_F64");
    return guarded([&] {
        if (dtype == SCHPF_F64) xphi_or_llh<double>(true, nnz, N, G, K, x, row, col, ths, thr, bes, ber, out);
        else xphi_or_llh<float>(true, nnz, N, G, K, x, row, col, ths, thr, bes, ber, out);
    });
}
int schpf_shape_update(int dtype, int64_t nnz, int K, const void *xphi, const int32_t *keep, int nkeep,
                       double prior, void *out)
{
    if (bad_dtype(dtype)) return fail("dtype must be SCHPF_F32 or SCHPF_F64");
    return guarded([&] {
        if (dtype == SCHPF_F64) shape_update<double>(nnz, K, xphi, keep, nkeep, prior, out);
        else shape_update<float>(nnz, K, xphi, keep, nkeep, prior, out);
    });
}
int schpf_rate_update(int dtype, int n, int m, int K, const void *ps, const void *pr, const void *os,
                      const void *orr, void *out)
{
    if (bad_dtype(dtype)) return fail("dtype must be SCHPF_F32 or SCHPF_F64");
    return guarded([&] {
        if (dtype == SCHPF_F64) rate_update<double>(n, m, K, ps, pr, os, orr, out);
        else rate_update<float>(n, m, K, ps, pr, os, orr, out);
    });
}
int schpf_capacity_rate_update(int dtype, int n, int K, const void *shape, const void *rate, double prior,
                               void *out)
{
    if (bad_dtype(dtype)) return fail("dtype must be SCHPF_F32 or SCHPF_F64");
    return guarded([&] {
        if (dtype == SCHPF_F64) capacity_rate<double>(n, K, shape, rate, prior, out);
        else capacity_rate<float>(n, K, shape, rate, prior, out);
    });
}

int schpf_create(schpf_ctx **out, int device, void *stream, int dtype, int ncells, int ngenes, int nfactors)
{
    if (!out) return fail("out is NULL");
    *out = nullptr;
    if (bad_dtype(dtype)) return fail("dtype must be SCHPF_F32 or SCHPF_F64");
    if (ncells < 1 || ngenes < 1) return fail("ncells and ngenes must be positive");
    return guarded([&] {
        int n = 0;
        HIPCHK(hipGetDeviceCount(&n));
        if (device < 0 || device >= n) throw std::invalid_argument("no such HIP device");
        if (dtype == SCHPF_F64) *out = new Engine<double>(device, stream, dtype, ncells, ngenes, nfactors);
        else *out = new Engine<float>(device, stream, dtype, ncells, ngenes, nfactors);
    });
}
int schpf_destroy(schpf_ctx *ctx)
{
    return guarded([&] { delete ctx; });
}

#define CTX_CALL(body)                                       \
    if (!ctx) return fail("ctx is NULL");                    \
    return guarded([&] {                                     \
        HIPCHK(hipSetDevice(ctx->device));                   \
        body;                                                \
    })

int schpf_upload_coo(schpf_ctx *ctx, int64_t nnz, const int32_t *row, const int32_t *col, const void *val,
                     int val_kind)
{
    CTX_CALL(ctx->upload_coo(nnz, row, col, val, val_kind));
}
int schpf_set_hypers(schpf_ctx *ctx, double a, double c, double bp, double dp)
{
    if (!(a > 0 && c > 0 && bp > 0 && dp > 0)) return fail("hyperparameters must be positive");
    CTX_CALL(ctx->a = a; ctx->c = c; ctx->bp = bp; ctx->dp = dp; ctx->hypers_changed());
}
int schpf_set_state(schpf_ctx *ctx, int which, const void *shape, const void *rate)
{
    CTX_CALL(ctx->set_state(which, shape, rate));
}
int schpf_get_state(schpf_ctx *ctx, int which, void *shape, void *rate)
{
    CTX_CALL(ctx->get_state(which, shape, rate));
}
int schpf_init_phi_host(schpf_ctx *ctx, const double *xphi) { CTX_CALL(ctx->init_phi_host(xphi)); }
int schpf_init_phi_device(schpf_ctx *ctx, uint64_t seed) { CTX_CALL(ctx->init_phi_device(seed)); }
int schpf_step(schpf_ctx *ctx, unsigned flags)
{
    if (flags & SCHPF_SHARDED) return fail("schpf_step is the single-GPU form; use step_local/step_finish");
    CTX_CALL(ctx->steps(flags, 1));
}
int schpf_step_local(schpf_ctx *ctx, unsigned flags) { CTX_CALL(ctx->step_local(flags)); }
int schpf_exchange_buffer(schpf_ctx *ctx, void **device_ptr, int64_t *count)
{
    CTX_CALL(ctx->exchange(device_ptr, count));
}
int schpf_step_finish(schpf_ctx *ctx, unsigned flags) { CTX_CALL(ctx->step_finish(flags)); }
int schpf_steps(schpf_ctx *ctx, unsigned flags, int n)
{
    if (flags & SCHPF_SHARDED) return fail("schpf_steps is the single-GPU form; use step_local/step_finish");
    CTX_CALL(ctx->steps(flags, n));
}
int schpf_loss_terms(schpf_ctx *ctx, double *llh_sum, double *gammaln_sum, int64_t *nnz)
{
    CTX_CALL(ctx->loss_terms(llh_sum, gammaln_sum, nnz));
}
int schpf_synchronize(schpf_ctx *ctx) { CTX_CALL(HIPCHK(hipStreamSynchronize(ctx->stream))); }

int schpf_comm_unique_id(void *out128)
{
    if (!out128) return fail("out is NULL");
    return guarded([&] {
        RcclUniqueId id;
        RCCLCHK(rccl().GetUniqueId(&id));
        std::memcpy(out128, &id, sizeof id);
    });
}
int schpf_comm_init(schpf_ctx *ctx, const void *unique_id128, int rank, int world)
{
    if (!unique_id128) return fail("unique_id is NULL");
    // a cached hipGraph of a sharded stretch holds an all-reduce bound to the communicator it was captured with: every
    // cached graph goes before the communicator does (hypers_changed() is "drop the captured graphs")
    CTX_CALL(ctx->hypers_changed(); ctx->comm_init(unique_id128, rank, world));
}
int schpf_comm_destroy(schpf_ctx *ctx) { CTX_CALL(ctx->hypers_changed(); ctx->comm_destroy()); }
int schpf_hint_sharded(schpf_ctx *ctx, int on) { CTX_CALL(ctx->hint_sharded(on)); }
int schpf_hint_transient(schpf_ctx *ctx, int on) { CTX_CALL(ctx->hint_transient(on)); }
int schpf_keep_rows(schpf_ctx *ctx, int on) { CTX_CALL(ctx->keep_rows(on)); }
int schpf_upload_rows(schpf_ctx *ctx, schpf_ctx *source, const int32_t *rows, int n_rows)
{
    if (!source) return fail("source is NULL");
    if (!rows && n_rows > 0) return fail("rows is NULL");
    CTX_CALL(ctx->upload_rows(source, rows, n_rows));
}
int schpf_steps_sharded(schpf_ctx *ctx, unsigned flags, int n)
{
    if (n < 0) return fail("n must be >= 0");
    CTX_CALL(ctx->steps_sharded(flags, n));
}
int schpf_loss_terms_all(schpf_ctx *ctx, double *llh_sum, double *gammaln_sum, int64_t *nnz)
{
    CTX_CALL(ctx->loss_terms_all(llh_sum, gammaln_sum, nnz));
}
int schpf_stream_handle(schpf_ctx *ctx, void **stream)
{
    if (!stream) return fail("stream is NULL");
    CTX_CALL(*stream = (void *)ctx->stream);
}

int schpf_profile_enable(schpf_ctx *ctx, int enable) { CTX_CALL(ctx->prof.on = enable != 0); }
int schpf_profile_read(schpf_ctx *ctx, double ms[4], int64_t launches[4])
{
    CTX_CALL(
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < 4; ++i) { ms[i] = 0.0; launches[i] = 0; }
        for (auto &r : ctx->prof.recs) {
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, r.a, r.b));
            ms[r.kind] += (double)t;
            launches[r.kind]++;
            ctx->prof.pool.push_back(r.a);
            ctx->prof.pool.push_back(r.b);
        }
        ctx->prof.recs.clear());
}
int schpf_profile_clock(schpf_ctx *ctx, double *shader_mhz, int64_t *launches)
{
    if (!shader_mhz || !launches) return fail("output pointer is NULL");
    CTX_CALL(ctx->profile_clock(shader_mhz, launches));
}
int schpf_sweep_bytes(schpf_ctx *ctx, int64_t info[8])
{
    if (!info) return fail("output pointer is NULL");
    CTX_CALL(ctx->sweep_bytes(info));
}
int schpf_plan_info(schpf_ctx *ctx, int64_t info[16])
{
    if (!info) return fail("output pointer is NULL");
    CTX_CALL(ctx->plan_info(info));
}
int schpf_upload_info(schpf_ctx *ctx, int64_t info[4])
{
    if (!info) return fail("output pointer is NULL");
    CTX_CALL(ctx->upload_info(info));
}

int schpf_coo_marginals(int64_t nnz, const int32_t *row, const int32_t *col, const void *val, int kind,
                        int ncells, int ngenes, double *row_sums, double *col_sums)
{
    return guarded([&] {
        if (nnz < 0 || ncells < 0 || ngenes < 0) throw std::invalid_argument("negative size");
        if (kind < SCHPF_VAL_I32 || kind > SCHPF_VAL_F64) throw std::invalid_argument("unknown value kind");
        // per-thread partial sums over contiguous slabs, added up in thread order: exact for counts
        // (integers far below 2^53) and run-to-run deterministic for anything else
        const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(schpf::host_threads(), 16), nnz / 65536 + 1));
        std::vector<std::vector<double>> pr((size_t)nth), pc((size_t)nth);
        std::vector<int64_t> bad((size_t)nth, -1);
        std::vector<std::thread> th;
        for (int t = 0; t < nth; ++t)
            th.emplace_back([&, t] {
                pr[(size_t)t].assign((size_t)ncells, 0.0);
                pc[(size_t)t].assign((size_t)ngenes, 0.0);
                double *r = pr[(size_t)t].data(), *g = pc[(size_t)t].data();
                const int64_t b = nnz * t / nth, e = nnz * (t + 1) / nth;
                for (int64_t i = b; i < e; ++i) {
                    if (row[i] < 0 || row[i] >= ncells || col[i] < 0 || col[i] >= ngenes) {
                        if (bad[(size_t)t] < 0) bad[(size_t)t] = i;
                        continue;
                    }
                    double d;
                    switch (kind) {
                    case SCHPF_VAL_I32: d = (double)((const int32_t *)val)[i]; break;
                    case SCHPF_VAL_I64: d = (double)((const int64_t *)val)[i]; break;
                    case SCHPF_VAL_F32: d = (double)((const float *)val)[i]; break;
                    default: d = ((const double *)val)[i]; break;
                    }
                    r[row[i]] += d;
                    g[col[i]] += d;
                }
            });
        for (auto &x : th) x.join();
        for (int t = 0; t < nth; ++t)
            if (bad[(size_t)t] >= 0)
                throw std::invalid_argument("COO index out of range at entry " + std::to_string(bad[(size_t)t]));
        for (int i = 0; i < ncells; ++i) { double s = 0.0; for (int t = 0; t < nth; ++t) s += pr[(size_t)t][(size_t)i]; row_sums[i] = s; }
        for (int i = 0; i < ngenes; ++i) { double s = 0.0; for (int t = 0; t < nth; ++t) s += pc[(size_t)t][(size_t)i]; col_sums[i] = s; }
    });
}

int schpf_debug_plan_expand(int64_t nnz, const int32_t *major, const int32_t *minor, const float *val,
                            int n_major, int n_minor, int lpc, int chunk_len, int n_windows,
                            int32_t *out_major, int32_t *out_minor, float *out_val, int32_t *out_natid,
                            int32_t *out_wave, int32_t *out_cptr, int64_t stats[4])
{
    return guarded([&] {
        schpf::SweepPlanHost P;
        schpf::build_sweep_plan(nnz, major, minor, val, n_major, n_minor, lpc, chunk_len, n_windows, false, P);
        std::vector<int32_t> wave_of_slice((size_t)P.n_slices, -1);
        for (int64_t w = 0; w < P.n_waves; ++w)
            if (P.wave_slice[(size_t)w] >= 0) {
                if (wave_of_slice[(size_t)P.wave_slice[(size_t)w]] != -1)
                    throw std::logic_error("slice scheduled twice");
                wave_of_slice[(size_t)P.wave_slice[(size_t)w]] = (int32_t)w;
            }
        int64_t n = 0;
        for (int64_t s = 0; s < P.n_slices; ++s) {
            if (wave_of_slice[(size_t)s] < 0) throw std::logic_error("slice never scheduled");
            const uint32_t *base = P.entries.data() + (size_t)P.slice_off[(size_t)s] * 4;
            for (int step = 0; step < P.slice_steps[(size_t)s]; ++step)
                for (int slot = 0; slot < P.cpw; ++slot)
                    for (int u = 0; u < 2; ++u) {
                        const uint32_t *e = base + ((size_t)step * P.cpw + slot) * 4 + (size_t)u * 2;
                        float f;
                        std::memcpy(&f, &e[1], 4);
                        if (f == 0.0f) continue;
                        if (n >= nnz) throw std::logic_error("plan stores more nonzeros than given");
                        out_major[n] = P.chunk_major[(size_t)s * P.cpw + slot];
                        out_minor[n] = (int32_t)e[0];
                        out_val[n] = f;
                        out_natid[n] = P.chunk_natid[(size_t)s * P.cpw + slot];
                        out_wave[n] = wave_of_slice[(size_t)s];
                        ++n;
                    }
        }
        if (n != nnz) throw std::logic_error("plan lost nonzeros");
        for (int m = 0; m <= n_major; ++m) out_cptr[m] = P.cptr[(size_t)m];
        stats[0] = P.n_chunks; stats[1] = P.n_slices; stats[2] = P.n_waves;
        stats[3] = (int64_t)P.entries.size() / 2;
    });
}


int schpf_debug_tile_expand(int64_t nnz, const int32_t *major, const int32_t *minor, const float *val,
                            int n_major, int n_minor, int lpc, int waves_per_block, int win_rows,
                            int target_tasks, int ring, int slot_bytes, int32_t *out_major, int32_t *out_minor,
                            float *out_val, int32_t *out_prow, int32_t *out_task, int32_t *out_pfirst,
                            int32_t *out_pcount, int64_t stats[8])
{
    return guarded([&] {
        schpf::TilePlanHost P;
        schpf::TileShape sh;
        sh.lpc = lpc; sh.waves_per_block = waves_per_block; sh.win_rows = win_rows; sh.target_tasks = target_tasks;
        sh.row_slots = env_int("SCHPF_DEBUG_ROW_SLOTS", 10);   // 160-byte table rows
        sh.ring = ring < 0 ? -ring : ring; sh.sync_stage = sh.ring > 1 ? 1 : 0; sh.slot_bytes = slot_bytes;
        // SCHPF_DEBUG_SINGLE=1: steps count nonzeros (plan.h; window schedule only)
        sh.single = env_int("SCHPF_DEBUG_SINGLE", 0) != 0 && sh.ring <= 1;
        sh.allow_packed = getenv("SCHPF_PACK") ? atoi(getenv("SCHPF_PACK")) != 0 : true;
        sh.bank_order = env_int("SCHPF_BANK_ORDER", 2);
        sh.taper = env_int("SCHPF_TAPER", 0) / 100.0;
        if (sh.taper > 0.0) sh.slots = 1;   // tapered ranges are for orientations with more tasks than workgroups (plan.h)
        // SCHPF_DEBUG_BALANCE=1: balanced windows (plan.h) -- the plan is built on the blocks' virtual numbering of the
        // minor rows and every entry is mapped back through minor_of
        std::vector<int32_t> minor_of;
        schpf::BalanceGeometry geo;
        const bool balanced = env_int("SCHPF_DEBUG_BALANCE", 0) != 0 && sh.ring <= 1;
        if (balanced) {
            schpf::BigVec<int32_t> vminor;
            schpf::balance_windows_host(nnz, major, minor, n_major, n_minor, sh, vminor, minor_of, geo);
            schpf::build_tile_plan(nnz, major, vminor.data(), val, n_major, geo.n_virtual, sh, false, P);
        } else
        schpf::build_tile_plan(nnz, major, minor, val, n_major, n_minor, sh, false, P);
        const int W = P.n_windows, gpw = P.gpw, wpb = P.wpb, gpb = P.gpb;
        // the LDS model of plan.cpp::bank_order: the lane groups of a pass read one row each per half step; rows of
        // one class (16-byte position mod 16, / lpc) are served one after the other
        const std::vector<int> pass_of = schpf::tile_pass_of(lpc, gpw);
        const int n_classes = std::max(1, 16 / std::max(1, lpc));
        int lpc_shift = 0;
        while ((1 << lpc_shift) < lpc) ++lpc_shift;
        int64_t lds_reads = 0, lds_extra = 0;
        int64_t n = 0;
        for (int64_t t = 0; t < P.n_tasks; ++t) {
            const int b = P.task_block[(size_t)t];
            for (int v = 0; v < wpb; ++v) {
                int64_t off = P.task_wave_off[(size_t)t * wpb + v];
                for (int w = P.task_w0[(size_t)t]; w < P.task_w1[(size_t)t]; ++w) {
                    const int steps = P.steps[((size_t)b * wpb + v) * W + w];
                    // the kernel's walk: `single` plans execute `steps` nonzeros (the slot halves 0 .. steps - 1)
                    const int n_half = P.single ? steps : 2 * steps;
                    for (int p = 0; 2 * p < n_half; ++p)
                        for (int u = 0; u < 2 && 2 * p + u < n_half; ++u) {
                          int in_class[4][16] = {};
                          for (int grp = 0; grp < gpw; ++grp) {
                                float f;
                                uint32_t off16;
                                if (P.packed) {
                                    const uint32_t *e = P.entries.data() + ((size_t)off + (size_t)p * gpw + grp) * 2;
                                    off16 = (e[0] >> (16 * u)) & 0xFFFFu;
                                    f = (float)((e[1] >> (16 * u)) & 0xFFFFu);
                                } else {
                                    const uint32_t *e = P.entries.data() + ((size_t)off + (size_t)p * gpw + grp) * 4 + (size_t)u * 2;
                                    off16 = e[0];
                                    std::memcpy(&f, &e[1], 4);
                                }
                                int mn;   // the kernel's reconstruction (sweep_impl.h entry_minor)
                                if (P.ring > 1) {
                                    const int slot = (int)(off16 / (uint32_t)P.slot16);
                                    const int r = (int)((off16 - (uint32_t)slot * P.slot16) / (uint32_t)P.row_slots);
                                    const int ahead = (slot - w % P.ring + P.ring) % P.ring;
                                    // readable in epoch w: sub-windows w .. w + look, inside the task
                                    if (slot >= P.ring || ahead > P.look || w + ahead >= P.task_w1[(size_t)t])
                                        throw std::logic_error("ring plan: an entry points outside the readable slots");
                                    if (f == 0.0f && off16 != (uint32_t)(w % P.ring) * (uint32_t)P.slot16)
                                        throw std::logic_error("ring plan: padding must point at the epoch's own slot");
                                    mn = (w + ahead) * P.win_rows + r;
                                } else {
                                    mn = w * P.win_rows + (int)(off16 / (uint32_t)P.row_slots);
                                }
                                if (f == 0.0f) continue;
                                in_class[pass_of[(size_t)grp]][((off16 & 15u) >> lpc_shift) & (unsigned)(n_classes - 1)]++;
                                if (schpf::tile_off16(P, mn) != off16) throw std::logic_error("tile plan: bad LDS position");
                                if (n >= nnz) throw std::logic_error("tile plan stores more nonzeros than given");
                                const int g = v * gpw + grp;
                                out_major[n] = P.block_rows[(size_t)b * gpb + g];
                                out_minor[n] = balanced ? minor_of[(size_t)b * geo.n_virtual + (size_t)mn] : (int32_t)mn;
                                out_val[n] = f;
                                out_prow[n] = (int32_t)(t * gpb + g);
                                out_task[n] = (int32_t)t;
                                ++n;
                          }
                          for (int ps = 0; ps < 4; ++ps) {
                              int worst = 0;
                              for (int c = 0; c < 16; ++c) worst = std::max(worst, in_class[ps][c]);
                              if (worst) { lds_reads++; lds_extra += worst - 1; }
                          }
                        }
                    off += schpf::tile_stored_steps(P, steps) * gpw;
                }
            }
        }
        if (n != nnz) throw std::logic_error("tile plan lost nonzeros");
        for (int m = 0; m < n_major; ++m) { out_pfirst[m] = P.pfirst[(size_t)m]; out_pcount[m] = P.pcount[(size_t)m]; }
        stats[0] = P.n_tasks; stats[1] = P.n_blocks; stats[2] = P.n_windows; stats[3] = P.pstride;
        stats[4] = (int64_t)P.entries.size() / (P.packed ? 1 : 2); stats[5] = P.windows_per_task;
        stats[6] = lds_reads; stats[7] = lds_extra;
    });
}

}  // extern "C"
