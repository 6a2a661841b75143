// solver.hip -- device-resident AMG hierarchy, multigrid cycle, coarse-level safe CG,
// preconditioned CG driver and the C-ABI of libfasp_hip.so.
//
// Boundary (include/fasp_hip.h): fasp_solver_dcsr_krylov_amg() keeps the reference's
// signature (base/src/SolCSR.c:476).  Inside it the host builds the hierarchy
// (host_setup.cpp), uploads it once, and the whole Krylov loop runs on the GPU; only
// reduction scalars cross back, through a pinned buffer.
//
// Reference control flow restated here (paths relative to the reference tree):
//   fasp_solver_dcsr_pcg     base/src/KryPcg.c:96-362   (incl. stagnation / false-convergence)
//   fasp_precond_amg         base/src/PreCSR.c:416-435  (tol NOT forwarded: coarse tol 1e-10)
//   fasp_solver_mgcycle      base/src/PreMGCycle.c:48-274
//   fasp_coarse_itsolver     base/src/PreMGUtil.inl:37-58
//   fasp_solver_dcsr_spcg    base/src/KrySPcg.c:60-367  (pc == NULL)
//
// There is NO CPU fallback: without a usable gfx950 device every entry point that
// computes returns ERROR_MISC after printing why.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fasp_comm.h"
#include "fasp_internal.h"
#include "kernels.hip.h"
#include "kernels2.hip.h"
#include "kernels3.hip.h"
#include "small_solvers.hip.h"
#include "seq_split.hip.h"
#include "seq_chain.hip.h"

namespace fasp {

#define HIPCK(expr)                                                                        \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "### ERROR: fasp_hip: %s failed: %s [%s:%d]\n", #expr,    \
                         hipGetErrorString(e_), __FILE__, __LINE__);                       \
            return ERROR_MISC;                                                             \
        }                                                                                  \
    } while (0)

// ---------------------------------------------------------------------------
// device context (one per process: one process per GPU)
// ---------------------------------------------------------------------------
struct Ctx {
    bool        ready  = false;
    int         device = -1;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;   // halo exchanges that overlap the interior rows (hierarchy.hip.h, dist_launch)
    hipEvent_t  ev_ready = nullptr, ev_halo = nullptr;
    double*     d_partials = nullptr;  // 8 quantities x MAXGRID
    double*     d_partials2 = nullptr; // second set (consumer kernels that read the first)
    double*     h_part     = nullptr;  // pinned mirror of d_partials2
    double*     d_red      = nullptr;  // reduced scalars (device)
    double*     h_red      = nullptr;  // pinned host mirror
    int         num_cu     = 256;
};
static Ctx g_ctx;
static int g_requested_device = -1;

constexpr int RED_SLOTS = 16;

static int ctx_init()
{
    if (g_ctx.ready) return FASP_SUCCESS;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        std::fprintf(stderr, "### ERROR: fasp_hip: no HIP device available (%s). This library has "
                             "no CPU fallback.\n", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return ERROR_MISC;
    }
    // an explicit fasp_hip_set_device() wins over the environment; FASP_HIP_DEVICE only fills in when nothing was asked for
    int dev = 0;
    if (g_requested_device >= 0) dev = g_requested_device;
    else if (const char* lr = std::getenv("FASP_HIP_DEVICE")) dev = std::atoi(lr);
    if (dev < 0 || dev >= ndev) {
        // several ranks on one device is the validation set-up only (shared-memory transport): it must be asked for
        const char* wrap = std::getenv("FASP_HIP_ALLOW_DEVICE_WRAP");
        if (wrap && std::atoi(wrap) != 0 && dev >= 0) dev = dev % ndev;
        else {
            std::fprintf(stderr, "### ERROR: fasp_hip: device %d requested, %d visible (one process per GPU; "
                                 "FASP_HIP_ALLOW_DEVICE_WRAP=1 shares devices for the shared-memory validation transport)\n", dev, ndev);
            return ERROR_INPUT_PAR;
        }
    }
    HIPCK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIPCK(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        std::fprintf(stderr, "### WARNING: fasp_hip: device %d is %s; kernels are built for gfx950\n",
                     dev, prop.gcnArchName);
    g_ctx.num_cu = prop.multiProcessorCount;
    g_ctx.device = dev;
    HIPCK(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
    HIPCK(hipStreamCreateWithFlags(&g_ctx.comm_stream, hipStreamNonBlocking));
    HIPCK(hipEventCreateWithFlags(&g_ctx.ev_ready, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&g_ctx.ev_halo, hipEventDisableTiming));
    HIPCK(hipMalloc(&g_ctx.d_partials, sizeof(double) * 8 * MAXGRID));
    HIPCK(hipMalloc(&g_ctx.d_partials2, sizeof(double) * (8 * MAXGRID + 8)));
    HIPCK(hipHostMalloc(&g_ctx.h_part, sizeof(double) * (8 * MAXGRID + 8), hipHostMallocDefault));
    HIPCK(hipMalloc(&g_ctx.d_red, sizeof(double) * RED_SLOTS));
    HIPCK(hipHostMalloc(&g_ctx.h_red, sizeof(double) * RED_SLOTS, hipHostMallocDefault));
    g_ctx.ready = true;
    // development knobs from the environment: FASP_HIP_TUNE="key=value,key=value" (the keys of fasp_hip_tune)
    if (const char* tv = std::getenv("FASP_HIP_TUNE")) {
        std::string all(tv);
        size_t p0 = 0;
        while (p0 < all.size()) {
            size_t p1 = all.find(',', p0);
            if (p1 == std::string::npos) p1 = all.size();
            const std::string kv = all.substr(p0, p1 - p0);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos && fasp_hip_tune(kv.substr(0, eq).c_str(), std::atoi(kv.c_str() + eq + 1)) < 0)
                std::fprintf(stderr, "### WARNING: fasp_hip: FASP_HIP_TUNE: unknown key in '%s'\n", kv.c_str());
            p0 = p1 + 1;
        }
    }
    return FASP_SUCCESS;
}

static inline int vec_grid(int n)
{
    long long g = ((long long)n + BLOCK - 1) / BLOCK;
    if (g > MAXGRID) g = MAXGRID;
    if (g < 1) g = 1;
    return (int)g;
}

#include "device_csr.hip.h"

#include "hierarchy.hip.h"

#include "smoothers.hip.h"

#include "coarse_cg.hip.h"

#include "krylov.hip.h"

#include "cycles.hip.h"

#include "pcg.hip.h"

}  // namespace fasp

#include "bsr.hip.h"

// ===========================================================================
// C ABI
// ===========================================================================
// ---------------------------------------------------------------------------
// Threading contract of the boundary (SURVEY section 8b "Threading"; the serial reference, AuxThreads.c:29-57, lets different
// threads solve on DISJOINT data).  The device side of this library is one context per process -- one stream, one set of
// reduction buffers, the fasp_hip_tune switches -- so every computing entry point takes ONE process-wide lock for its whole
// duration: calls from different host threads are SERIALISED, never interleaved.  The lock is recursive (an entry may call
// another one, and a caller's precond / mxv_matfree callback may call back into the library on the same thread).
// Concurrency across GPUs is one PROCESS per GPU, not threads.
// ---------------------------------------------------------------------------
static std::recursive_mutex g_entry_mutex;
#define FASP_ENTRY() std::lock_guard<std::recursive_mutex> fasp_entry_lock_(g_entry_mutex)

extern "C" {

int fasp_hip_set_device(int device)
{
    FASP_ENTRY();
    if (g_ctx.ready && device != g_ctx.device) {
        std::fprintf(stderr, "### ERROR: fasp_hip: device already bound to %d\n", g_ctx.device);
        return ERROR_INPUT_PAR;
    }
    g_requested_device = device;
    return ctx_init();
}

int fasp_hip_device_count(void)
{
    FASP_ENTRY();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return ERROR_MISC;
    return n;
}

// PCI bus id ("0000:c1:00.0") of the device this process is bound to (binding it to the default one if nothing is yet): how a
// multi-rank run proves that its ranks sit on DIFFERENT devices (bench_dist.py, tests/test_gpu_dist.py).
int fasp_hip_device_identity(char* out, int cap)
{
    FASP_ENTRY();
    if (!out || cap < 16) return ERROR_INPUT_PAR;
    if (ctx_init() < 0) return ERROR_MISC;
    if (hipDeviceGetPCIBusId(out, cap, g_ctx.device) != hipSuccess) return ERROR_MISC;
    return g_ctx.device;
}

int fasp_hip_available(void)
{
    FASP_ENTRY();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    return 1;
}

int fasp_hip_amg_create_host(fasp_hip_amg** out, const dCSRmat* A, AMG_param* amgparam)
{
    FASP_ENTRY();
    if (!out || !A || !amgparam) return ERROR_INPUT_PAR;
    *out = nullptr;
    int st = check_supported(nullptr, amgparam);
    if (st < 0) return st;
    fasp_hip_amg* h = new fasp_hip_amg();
    st = (amgparam->AMG_type == SA_AMG)   ? host_setup_sa(A, amgparam, h->H)  // SolCSR.c:509-521
         : (amgparam->AMG_type == UA_AMG) ? host_setup_ua(A, amgparam, h->H)
                                          : host_setup_rs(A, amgparam, h->H);
    if (st < 0) { delete h; return st; }
    h->param = *amgparam;
    *out = h;
    return FASP_SUCCESS;
}

// ---- one host setup per node (SURVEY.md section 8e "Setup"): the rank that ran it publishes the host hierarchy
// in a POSIX shared-memory segment, the other ranks of the node map it read-only and take their rows from it.
namespace {
struct ShmHierHeader {
    unsigned long long magic;       // 'FASPHIER'
    unsigned long long total_bytes;
    int                nl;
    AMG_param          param;       // as the setup left it
    struct Lvl { int has_coarse; int dims[3][3]; unsigned long long off[3][3]; unsigned long long cf_off; unsigned long long cf_n; } lvl[MAX_AMG_LVL];
};
constexpr unsigned long long HIER_MAGIC = 0x5245494850534146ull;
inline unsigned long long align64(unsigned long long x) { return (x + 63ull) & ~63ull; }
struct ShmMapping { void* base; size_t bytes; };
std::vector<std::pair<fasp_hip_amg*, ShmMapping>> g_attached;  // unmapped when the handle is destroyed
}  // namespace

int fasp_hip_amg_publish(const fasp_hip_amg* h, const char* name)
{
    FASP_ENTRY();
    if (!h || !name || h->H.L.empty() || (int)h->H.L.size() > MAX_AMG_LVL) return ERROR_INPUT_PAR;
    const int nl = (int)h->H.L.size();
    ShmHierHeader hd;
    std::memset(&hd, 0, sizeof(hd));
    hd.magic = HIER_MAGIC; hd.nl = nl; hd.param = h->param;
    unsigned long long off = align64(sizeof(ShmHierHeader));
    for (int l = 0; l < nl; ++l) {
        const HostLevel& L = h->H.L[l];
        hd.lvl[l].has_coarse = L.has_coarse ? 1 : 0;
        const HostCSR* M[3] = {&L.A, &L.P, &L.R};
        for (int w = 0; w < 3; ++w) {
            if (w > 0 && !L.has_coarse) continue;
            hd.lvl[l].dims[w][0] = M[w]->row; hd.lvl[l].dims[w][1] = M[w]->col; hd.lvl[l].dims[w][2] = M[w]->nnz;
            hd.lvl[l].off[w][0] = off; off = align64(off + 4ull * ((unsigned long long)M[w]->row + 1));
            hd.lvl[l].off[w][1] = off; off = align64(off + 4ull * (unsigned long long)M[w]->nnz);
            hd.lvl[l].off[w][2] = off; off = align64(off + 8ull * (unsigned long long)M[w]->nnz);
        }
        hd.lvl[l].cf_n = L.cfmark.n;
        hd.lvl[l].cf_off = off; off = align64(off + 4ull * L.cfmark.n);
    }
    hd.total_bytes = off;
    const std::string nm = std::string("/") + name;
    shm_unlink(nm.c_str());
    const int fd = shm_open(nm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)off) != 0) { std::perror("fasp_hip_amg_publish"); if (fd >= 0) close(fd); return ERROR_MISC; }
    char* base = (char*)mmap(nullptr, (size_t)off, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) { std::perror("fasp_hip_amg_publish mmap"); return ERROR_ALLOC_MEM; }
    for (int l = 0; l < nl; ++l) {
        const HostLevel& L = h->H.L[l];
        const HostCSR* M[3] = {&L.A, &L.P, &L.R};
        for (int w = 0; w < 3; ++w) {
            if (w > 0 && !L.has_coarse) continue;
            std::memcpy(base + hd.lvl[l].off[w][0], M[w]->ia.data(), 4 * ((size_t)M[w]->row + 1));
            std::memcpy(base + hd.lvl[l].off[w][1], M[w]->ja.data(), 4 * (size_t)M[w]->nnz);
            std::memcpy(base + hd.lvl[l].off[w][2], M[w]->val.data(), 8 * (size_t)M[w]->nnz);
        }
        if (L.cfmark.n) std::memcpy(base + hd.lvl[l].cf_off, L.cfmark.data(), 4 * L.cfmark.n);
    }
    std::memcpy(base, &hd, sizeof(hd));  // header last: a reader that sees the magic sees everything
    munmap(base, (size_t)off);
    return FASP_SUCCESS;
}

int fasp_hip_amg_unpublish(const char* name)
{
    FASP_ENTRY();
    if (!name) return ERROR_INPUT_PAR;
    shm_unlink((std::string("/") + name).c_str());
    return FASP_SUCCESS;
}

int fasp_hip_amg_attach(fasp_hip_amg** out, const char* name)
{
    FASP_ENTRY();
    if (!out || !name) return ERROR_INPUT_PAR;
    *out = nullptr;
    const std::string nm = std::string("/") + name;
    const int fd = shm_open(nm.c_str(), O_RDONLY, 0600);
    if (fd < 0) { std::fprintf(stderr, "### ERROR: fasp_hip_amg_attach: no segment %s\n", nm.c_str()); return ERROR_OPEN_FILE; }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(ShmHierHeader)) { close(fd); return ERROR_WRONG_FILE; }
    char* base = (char*)mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) return ERROR_ALLOC_MEM;
    const ShmHierHeader* hd = reinterpret_cast<const ShmHierHeader*>(base);
    if (hd->magic != HIER_MAGIC || hd->total_bytes > (unsigned long long)sb.st_size || hd->nl <= 0 || hd->nl > MAX_AMG_LVL) {
        munmap(base, (size_t)sb.st_size);
        return ERROR_WRONG_FILE;
    }
    fasp_hip_amg* h = new fasp_hip_amg();
    h->param = hd->param;
    h->param.amli_coef = nullptr;  // a pointer of the publishing process; formed again on first use
    h->H.L.resize((size_t)hd->nl);
    for (int l = 0; l < hd->nl; ++l) {
        HostLevel& L = h->H.L[(size_t)l];
        L.has_coarse = hd->lvl[l].has_coarse != 0;
        HostCSR* M[3] = {&L.A, &L.P, &L.R};
        for (int w = 0; w < 3; ++w) {
            if (w > 0 && !L.has_coarse) continue;
            M[w]->row = hd->lvl[l].dims[w][0]; M[w]->col = hd->lvl[l].dims[w][1]; M[w]->nnz = hd->lvl[l].dims[w][2];
            M[w]->ia.view(reinterpret_cast<int*>(base + hd->lvl[l].off[w][0]), (size_t)M[w]->row + 1);
            M[w]->ja.view(reinterpret_cast<int*>(base + hd->lvl[l].off[w][1]), (size_t)M[w]->nnz);
            M[w]->val.view(reinterpret_cast<double*>(base + hd->lvl[l].off[w][2]), (size_t)M[w]->nnz);
        }
        if (hd->lvl[l].cf_n) L.cfmark.view(reinterpret_cast<int*>(base + hd->lvl[l].cf_off), (size_t)hd->lvl[l].cf_n);
    }
    g_attached.push_back({h, ShmMapping{base, (size_t)sb.st_size}});
    *out = h;
    return FASP_SUCCESS;
}

int fasp_hip_amg_upload(fasp_hip_amg* h)
{
    FASP_ENTRY();
    if (!h) return ERROR_INPUT_PAR;
    if (!h->L.empty()) return FASP_SUCCESS;
    int st = ctx_init();
    if (st < 0) return st;
    try { return upload_hierarchy(h); }
    catch (const std::bad_alloc&) { std::printf("### ERROR: fasp_hip: host allocation failed during the upload\n"); return ERROR_ALLOC_MEM; }
}

namespace {
// Overlapped upload (single rank, classical setup): a second thread sends level l to the device -- matrix coding,
// re-sorting and all -- while the host setup builds level l + 1.  256^3: the 3.5-4 s of coding + upload disappear
// behind the 10 s of host setup.
struct AheadUpload {
    fasp_hip_amg*           h = nullptr;
    std::mutex              mu;
    std::condition_variable cv;
    std::vector<int>        ready;
    bool                    done = false;
    int                     status = FASP_SUCCESS;
    int                     coarsest = -1;   // set (under mu) in front of the last level's turn: that one is not smoothed
    std::thread             th;
    static void on_matrix(int level, void* ctx)   // (on the setup thread)
    {
        AheadUpload* self = static_cast<AheadUpload*>(ctx);
        sched_jobs_start_level(self->h, level, true);
    }
    static void on_ready(int level, void* ctx)
    {
        AheadUpload* self = static_cast<AheadUpload*>(ctx);
        { std::lock_guard<std::mutex> lk(self->mu); self->ready.push_back(level); }
        self->cv.notify_one();
    }
    void run()
    {
        (void)hipSetDevice(g_ctx.device);
        HostThreads team;
        double t_first = -1.0;
        for (;;) {
            int l = -1, last = -1;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !ready.empty() || done; });
                if (ready.empty()) return;
                l = ready.front(); ready.erase(ready.begin());
                last = coarsest;
            }
            if (t_first < 0) t_first = wall_seconds();
            const double t0 = wall_seconds();
            if (status >= 0) {
                int st;
                try { st = upload_level(h, l, nullptr); }
                catch (const std::bad_alloc&) { st = ERROR_ALLOC_MEM; }   // (a host buffer of the coding passes: never out of this thread)
                if (st < 0) status = st;
                // sequential smoothers: this level's sweep schedules, on host threads of their own beside the rest of the setup (smoothers.hip.h)
                if (st >= 0 && l != last) sched_jobs_start_level(h, l);
            }
            if (std::getenv("FASP_HIP_SETUP_TIMING"))
                std::printf("  [upload ahead] level %d: %.3f s (started %.3f s after the first)\n", l, wall_seconds() - t0, t0 - t_first);
        }
    }
};
}  // namespace

int fasp_hip_amg_create(fasp_hip_amg** out, const dCSRmat* A, AMG_param* amgparam)
{
    FASP_ENTRY();
    if (!out || !A || !amgparam) return ERROR_INPUT_PAR;
    *out = nullptr;
    int st = check_supported(nullptr, amgparam);
    if (st < 0) return st;
    if ((st = ctx_init()) < 0) return st;  // fail before the (long) host setup when there is no GPU
    fasp_hip_amg* h = nullptr;
    static const bool timing = std::getenv("FASP_HIP_SETUP_TIMING") != nullptr;
    double tp = wall_seconds();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const double now = wall_seconds();
        std::printf("  [create] %-28s %8.3f s\n", what, now - tp);
        tp = now;
    };
    static const bool ahead_on = !(std::getenv("FASP_HIP_UPLOAD_AHEAD") && std::atoi(std::getenv("FASP_HIP_UPLOAD_AHEAD")) == 0);
    const bool ahead = ahead_on && comm_size() == 1 && amgparam->AMG_type != SA_AMG && amgparam->AMG_type != UA_AMG &&
                       A->nnz >= 1000000;   // worth a thread from a few hundred thousand rows on
    if (ahead) {
        h = new fasp_hip_amg();
        h->param = *amgparam;                    // (compress_enabled() etc. read globals only)
        h->L.resize(MAX_AMG_LVL + 1);            // no reallocation while the second thread fills levels
        AheadUpload up;
        up.h = h;
        up.th = std::thread([&up] { up.run(); });
        g_on_level_ready = &AheadUpload::on_ready; g_on_level_matrix = &AheadUpload::on_matrix; g_on_level_ready_ctx = &up;
        st = host_setup_rs(A, amgparam, h->H);
        g_on_level_ready = nullptr; g_on_level_matrix = nullptr; g_on_level_ready_ctx = nullptr;
        lap("host setup");
        if (st >= 0) {   // the coarsest level
            { std::lock_guard<std::mutex> lk(up.mu); up.coarsest = (int)h->H.L.size() - 1; }
            AheadUpload::on_ready((int)h->H.L.size() - 1, &up);
        }
        { std::lock_guard<std::mutex> lk(up.mu); up.done = true; }
        up.cv.notify_one();
        up.th.join();
        lap("wait for the level uploads");
        if (st >= 0) st = up.status;
        // Schedules started early (sweeps in natural order start the moment a level's matrix is final) for the level that turned out to be
        // the coarsest -- it is solved, not smoothed -- are given back now instead of holding device memory until the hierarchy goes.
        {
            const int last = (int)h->H.L.size() - 1;
            std::lock_guard<std::mutex> lk(h->sched_mu);
            for (auto& J : h->sched_jobs)
                if (J && J->level >= last) {
                    if (J->th.joinable()) J->th.join();
                    if (J->uploaded) J->S.release();
                    J.reset();
                }
        }
        if (st < 0) { h->L.resize(h->H.L.size()); fasp_hip_amg_destroy(h); return st; }
        h->param = *amgparam;
    } else {
        st = fasp_hip_amg_create_host(&h, A, amgparam);
        if (st < 0) return st;
        lap("host setup");
    }
    try { st = upload_hierarchy(h); }
    catch (const std::bad_alloc&) { std::printf("### ERROR: fasp_hip: host allocation failed during the upload\n"); st = ERROR_ALLOC_MEM; }
    lap("upload_hierarchy");
    if (st < 0) { fasp_hip_amg_destroy(h); return st; }
    *out = h;
    return FASP_SUCCESS;
}

void fasp_hip_amg_destroy(fasp_hip_amg* h)
{
    FASP_ENTRY();
    if (h) sched_jobs_join(h);
    if (!h) return;
    if (g_ctx.ready) (void)hipStreamSynchronize(g_ctx.stream);
    for (auto& D : h->L) free_level(D);
    double* v[] = {h->b, h->u, h->p, h->t, h->r, h->cp, h->cr, h->ct, h->cbest};
    for (double* q : v)
        if (q) (void)hipFree(q);
    if (h->d_lazy) (void)hipFree(h->d_lazy);
    if (h->reg_img) (void)hipFree(h->reg_img);
    if (h->h_lazy) (void)hipHostFree(h->h_lazy);
    for (auto& e : h->ev) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (int s = 0; s < 2; ++s)
        for (double* q : h->gm[s])
            if (q) (void)hipFree(q);
    if (h->gm_hh) (void)hipFree(h->gm_hh);
    if (h->spcg_state) (void)hipFree(h->spcg_state);
    if (h->spcg_fused_buf) (void)hipFree(h->spcg_fused_buf);
    {
        auto& P = h->persist;
        void* q[] = {P.vals, P.cols, P.wrow, P.wend, P.t2, P.sync};
        for (void* x : q) if (x) (void)hipFree(x);
    }
    ShmMapping map{nullptr, 0};
    for (size_t q = 0; q < g_attached.size(); ++q)
        if (g_attached[q].first == h) { map = g_attached[q].second; g_attached.erase(g_attached.begin() + (long)q); break; }
    delete h;                                     // views release nothing
    if (map.base) munmap(map.base, map.bytes);    // ... and the mapping goes after them
}

// Host-only check of the lossless matrix coding (no GPU needed): codes A the way upload_csr would,
// decodes it again and compares with A bit for bit.  *kind_out = 5 (row patterns), 4 (byte
// dictionary) or 0 (stays plain CSR).  Returns 0 when the round trip is exact.
int fasp_hip_coding_selftest(const dCSRmat* A, int* kind_out)
{
    FASP_ENTRY();
    if (!A || !kind_out) return ERROR_INPUT_PAR;
    HostCSR M;
    M.row = A->row; M.col = A->col; M.nnz = A->nnz;
    M.ia.alloc((size_t)A->row + 1); M.ja.alloc((size_t)std::max(A->nnz, 1)); M.val.alloc((size_t)std::max(A->nnz, 1));
    std::memcpy(M.ia.data(), A->IA, sizeof(int) * ((size_t)A->row + 1));
    std::memcpy(M.ja.data(), A->JA, sizeof(int) * (size_t)A->nnz);
    std::memcpy(M.val.data(), A->val, sizeof(double) * (size_t)A->nnz);
    const bool square = M.row == M.col;
    *kind_out = 0;
    if (!(M.nnz >= 4096 && (double)M.nnz <= 48.0 * M.row)) return FASP_SUCCESS;
    auto bits = [](double v) { unsigned long long b; std::memcpy(&b, &v, 8); return b; };
    {
        Buf<unsigned short> pat; Buf<int> rb;
        std::vector<int> pstart, plen, poff; std::vector<double> pval;
        if (build_rowpat(M, pat, pstart, plen, poff, pval, rb)) {
            *kind_out = 5;
            for (int r = 0; r < M.row; ++r) {
                const int id = pat[r], base = square ? r : rb[r];
                if (plen[id] != M.ia[r + 1] - M.ia[r]) return ERROR_MISC;
                for (int j = 0; j < plen[id]; ++j) {
                    const int k = M.ia[r] + j;
                    if (base + poff[pstart[id] + j] != M.ja[k] || bits(pval[pstart[id] + j]) != bits(M.val[k])) return ERROR_MISC;
                }
                for (int j = plen[id]; j < (plen[id] + 7) / 8 * 8; ++j)  // padding: offset 0, value +0.0
                    if (poff[pstart[id] + j] != 0 || bits(pval[pstart[id] + j]) != 0ull) return ERROR_MISC;
            }
            return FASP_SUCCESS;
        }
    }
    {
        std::vector<int> doff; std::vector<double> dval;
        Buf<unsigned char> code; Buf<int> rb;
        if (build_dict8(M, doff, dval, code, rb)) {
            *kind_out = 4;
            for (int r = 0; r < M.row; ++r) {
                const int base = square ? r : rb[r];
                for (int k = M.ia[r]; k < M.ia[r + 1]; ++k)
                    if (base + doff[code[k]] != M.ja[k] || bits(dval[code[k]]) != bits(M.val[k])) return ERROR_MISC;
            }
        }
    }
    return FASP_SUCCESS;
}

int fasp_hip_amg_num_levels(const fasp_hip_amg* h) { return h ? (int)h->H.L.size() : ERROR_INPUT_PAR; }

// which kernel family serves operator `which` (0 A, 1 P, 2 R) of a level, and how many bytes of
// matrix data one pass of it reads (row pointers / indices / values, or their coded form)
int fasp_hip_amg_kernel_info(const fasp_hip_amg* h, int level, int which, int* kind, double* matrix_bytes)
{
    FASP_ENTRY();
    if (!h || level < 0 || level >= (int)h->L.size() || which < 0 || which > 2) return ERROR_INPUT_PAR;
    const DevLevel& D = h->L[level];
    const DevCSR& M = which == 0 ? D.A : which == 1 ? D.P : D.R;
    if (!M.ia) return ERROR_INPUT_PAR;
    int k = M.kind;
    double bytes = 12.0 * M.nnz + 4.0 * (M.row + 1.0);
    if (M.kind == 0 && M.ja16 && g_tune.ja16) bytes = 10.0 * M.nnz + 4.0 * (M.row + 1.0) + (M.jbase ? 4.0 * M.row : 0.0);   // 16-bit indices
    if (M.code && g_tune.compress) { k = 4; bytes = 1.0 * M.nnz + 4.0 * (M.row + 1.0) + (M.rowbase ? 4.0 * M.row : 0.0); }
    if (M.pat && g_tune.compress) { k = 5; bytes = 2.0 * M.row + (M.rowbase ? 4.0 * M.row : 0.0) + 12.0 * M.npent; }
    // second-generation kernels (kernels2.hip.h), same selection as launch_csr: 6 = k_csr_rowpat4, 7 = k_csr_lstream, 8 = k_csr_wstream2, 9 = k_csr_rowpat5, 10 = k_csr_xtile
    if (k == 5 && g_tune.gen2 && M.nxrows >= 0 && !M.rowbase) k = 6;
    else if (k == 5 && g_tune.gen2 >= 2 && M.nxrows >= 0 && M.rowbase && (double)M.nnz <= 0.1 * g_tune.rp5_max * M.row) k = 9;   // k_csr_rowpat5
    if (k == 2 && g_tune.gen2 && M.wrows == 64 && M.wcap == 512 && (double)M.nnz <= 7.6 * M.row) k = 7;
    else if (k == 2 && g_tune.gen2 >= 2 && g_tune.xtile && M.lja16 && M.wrows == 64 && M.wcap == 512) {   // k_csr_xtile
        k = 10;
        bytes = 10.0 * M.nnz + 4.0 * (M.row + 1.0) + 4.0 * M.ntcols + 4.0 * ((M.row + 63) / 64 + 1.0);   // values + 16-bit positions + the tiles' column lists
    }
    else if (k == 2 && g_tune.gen2 >= 2 && M.wrows == 64 && M.wcap == 512) k = 8;   // k_csr_wstream2
    if (kind) *kind = k;
    if (matrix_bytes) *matrix_bytes = bytes;
    return FASP_SUCCESS;
}

int fasp_hip_amg_get_matrix(const fasp_hip_amg* h, int level, int which, dCSRmat* view)
{
    FASP_ENTRY();
    if (!h || !view || level < 0 || level >= (int)h->H.L.size()) return ERROR_INPUT_PAR;
    const HostLevel& L = h->H.L[level];
    if (which != 0 && !L.has_coarse) return ERROR_INPUT_PAR;
    *view = which == 0 ? L.A.view() : which == 1 ? L.P.view() : L.R.view();
    return FASP_SUCCESS;
}

int fasp_hip_amg_get_cfmark(const fasp_hip_amg* h, int level, ivector* view)
{
    FASP_ENTRY();
    if (!h || !view || level < 0 || level >= (int)h->H.L.size() || !h->H.L[level].has_coarse)
        return ERROR_INPUT_PAR;
    view->row = h->H.L[level].A.row;
    view->val = const_cast<int*>(h->H.L[level].cfmark.data());
    return FASP_SUCCESS;
}

int fasp_hip_set_rhs(fasp_hip_amg* h, const dvector* b)
{
    FASP_ENTRY();
    if (!h || !b || h->L.empty()) return ERROR_INPUT_PAR;
    const DevLevel& D0 = h->L[0];
    if (b->row != D0.nglobal) return ERROR_MAT_SIZE;  // host vectors are global; a rank uploads its rows
    HIPCK(hipMemcpyAsync(h->b, b->val + D0.row0, sizeof(double) * D0.nloc, hipMemcpyHostToDevice, g_ctx.stream));
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    return FASP_SUCCESS;
}

int fasp_hip_set_guess(fasp_hip_amg* h, const dvector* x)
{
    FASP_ENTRY();
    if (!h || h->L.empty()) return ERROR_INPUT_PAR;
    const DevLevel& D0 = h->L[0];
    if (x) {
        if (x->row != D0.nglobal) return ERROR_MAT_SIZE;
        HIPCK(hipMemcpyAsync(h->u, x->val + D0.row0, sizeof(double) * D0.nloc, hipMemcpyHostToDevice, g_ctx.stream));
    } else {
        HIPCK(hipMemsetAsync(h->u, 0, sizeof(double) * D0.nvec, g_ctx.stream));
    }
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    return FASP_SUCCESS;
}

int fasp_hip_get_solution(fasp_hip_amg* h, dvector* x)
{
    FASP_ENTRY();
    if (!h || !x || h->L.empty()) return ERROR_INPUT_PAR;
    const DevLevel& D0 = h->L[0];
    if (x->row != D0.nglobal) return ERROR_MAT_SIZE;  // a rank fills the rows it owns
    HIPCK(hipMemcpyAsync(x->val + D0.row0, h->u, sizeof(double) * D0.nloc, hipMemcpyDeviceToHost, g_ctx.stream));
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    return FASP_SUCCESS;
}

int fasp_hip_device_synchronize(void)
{
    FASP_ENTRY();
    if (!g_ctx.ready) return FASP_SUCCESS;
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    return FASP_SUCCESS;
}

int fasp_hip_solve_resident(fasp_hip_amg* h, const ITS_param* itparam, double* hist, int hist_cap,
                            fasp_hip_stats* stats)
{
    FASP_ENTRY();
    if (!h || !itparam) return ERROR_INPUT_PAR;
    if (h->L.empty()) return ERROR_INPUT_PAR;  // hierarchy not uploaded
    int st = check_supported(itparam, &h->param);
    if (st < 0) return st;
    // ITS_CHECK, KryUtil.inl:71-83
    if (itparam->tol < SMALLREAL)
        std::printf("### WARNING: Convergence tolerance is too small! [%s:%d]\n", "ITS_CHECK", 74);
    if (itparam->maxit <= 0)
        std::printf("### WARNING: Max number of iterations must be POSITIVE! [%s:%d]\n", "ITS_CHECK", 78);

    h->ev_used = 0;
    h->use_fmg = itparam->precond_type == PREC_FMG;   // SolCSR.c:537-538
    const long long ci0 = h->coarse_iters, vc0 = h->vcycles;
    Hist   H{hist, hist_cap, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    const double t0 = wall_seconds();
    // SolCSR.c:530-551: the AMG preconditioner is always installed on this path;
    // SolCSR.c:84-130: dispatch on itsolver_type (restart is narrowed to SHORT, :62)
    switch (itparam->itsolver_type) {
        case SOLVER_BiCGstab:
        {
            KOps K = csr_ops(h, 0, true);
            st = bicgstab_device(K, h->b, h->u, itparam->tol, itparam->maxit, itparam->print_level, &H, &po);
        } break;
        case SOLVER_GMRES:
        case SOLVER_VGMRES:
        case SOLVER_VFGMRES:
        {
            KOps K = csr_ops(h, 0, true);
            st = gmres_device(K, h->b, h->u, itparam->itsolver_type == SOLVER_VFGMRES ? 1 : itparam->itsolver_type == SOLVER_GMRES ? 3 : 0, itparam->tol,
                              itparam->abstol, itparam->maxit, (short)itparam->restart, itparam->stop_type,
                              itparam->print_level, &H, &po);
        } break;
        case SOLVER_MinRes:
        {
            KOps K = csr_ops(h, 0, true);
            st = minres_device(K, h->b, h->u, itparam->tol, itparam->abstol, itparam->maxit, itparam->stop_type,
                               itparam->print_level, &H, &po);
        } break;
        case SOLVER_GCG:
        {
            KOps K = csr_ops(h, 0, true);
            st = gcg_device(K, h->b, h->u, itparam->tol, itparam->abstol, itparam->maxit, itparam->stop_type,
                            itparam->print_level, &H, &po);
        } break;
        case SOLVER_GCR:
        {
            KOps K = csr_ops(h, 0, true);
            st = gcr_device(K, h->b, h->u, itparam->tol, itparam->abstol, itparam->maxit, (short)itparam->restart,
                            itparam->stop_type, itparam->print_level, &H, &po);
        } break;
        default:
        {
            KOps K = csr_ops(h, 0, true);
            PcgVecs V{h->b, h->u, h->p, h->t, h->r};
            st = pcg_device(K, V, itparam->tol, itparam->abstol, itparam->maxit, itparam->stop_type,
                            itparam->print_level, H, po);
        }
    }
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    if (seq_err_check() < 0) return ERROR_MISC;   // a broken cluster of the sequential smoothers (smoothers.hip.h)
    const double t_solve = wall_seconds() - t0;

    if (stats) {
        stats->iters = st; stats->nhist = H.n; stats->relres = po.relres; stats->absres = po.absres;
        stats->normr0 = po.normr0; stats->solve_seconds = t_solve; stats->upload_seconds = 0.0;
        double ms = 0.0;
        for (int i = 0; i < h->ev_used; ++i) {
            float e = 0.f;
            if (hipEventElapsedTime(&e, h->ev[i].a, h->ev[i].b) == hipSuccess) ms += e;
        }
        stats->spmv_launches = h->ev_used;
        stats->spmv_ms = h->ev_used ? ms / h->ev_used : 0.0;
        stats->coarse_iters = h->coarse_iters - ci0;
        stats->vcycles = h->vcycles - vc0;
    }
    if (itparam->print_level >= PRINT_SOME && st >= 0)
        std::printf("Iterative method costs %.4f seconds.\n", t_solve);
    return st;
}

int fasp_hip_solve(fasp_hip_amg* h, const dvector* b, dvector* x, const ITS_param* itparam, double* hist,
                   int hist_cap, fasp_hip_stats* stats)
{
    FASP_ENTRY();
    if (!h || !b || !x || !itparam) return ERROR_INPUT_PAR;
    double t0 = wall_seconds();
    int st = fasp_hip_set_rhs(h, b);
    if (st < 0) return st;
    if ((st = fasp_hip_set_guess(h, x)) < 0) return st;
    double t_up = wall_seconds() - t0;
    st = fasp_hip_solve_resident(h, itparam, hist, hist_cap, stats);
    t0 = wall_seconds();
    const int st2 = fasp_hip_get_solution(h, x);
    if (st2 < 0) return st2;
    t_up += wall_seconds() - t0;
    if (stats) stats->upload_seconds = t_up;
    return st;
}

// AMG as a stand-alone solver on a resident hierarchy (PreMGSolve.c:49); param == NULL: the
// parameters the hierarchy was built with
int fasp_hip_amg_solve(fasp_hip_amg* h, const dvector* b, dvector* x, const AMG_param* param, double* hist,
                       int hist_cap, fasp_hip_stats* stats)
{
    FASP_ENTRY();
    if (!h || !b || !x || h->L.empty()) return ERROR_INPUT_PAR;
    const AMG_param& p = param ? *param : h->param;
    int st = check_supported(nullptr, &p);
    if (st < 0) return st;
    double t0 = wall_seconds();
    if ((st = fasp_hip_set_rhs(h, b)) < 0) return st;
    if ((st = fasp_hip_set_guess(h, x)) < 0) return st;
    double t_up = wall_seconds() - t0;
    h->ev_used = 0;
    const long long ci0 = h->coarse_iters, vc0 = h->vcycles;
    Hist   H{hist, hist_cap, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    t0 = wall_seconds();
    st = amg_solve_device(h, p, H, po);
    const double t_solve = wall_seconds() - t0;
    t0 = wall_seconds();
    const int st2 = fasp_hip_get_solution(h, x);
    if (st2 < 0) return st2;
    t_up += wall_seconds() - t0;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->iters = st; stats->nhist = H.n; stats->relres = po.relres; stats->absres = po.absres;
        stats->normr0 = po.normr0; stats->solve_seconds = t_solve; stats->upload_seconds = t_up;
        stats->coarse_iters = h->coarse_iters - ci0;
        stats->vcycles = h->vcycles - vc0;
    }
    if (p.print_level > PRINT_NONE) std::printf("AMG solve costs %.4f seconds.\n", t_solve);
    return st;
}

// SolAMG.c:49.  A failed setup returns its error code (the reference would fall back to an
// unpreconditioned CPU GMRES there; this library has no CPU solve path).
int fasp_solver_amg(dCSRmat* A, dvector* b, dvector* x, AMG_param* param)
{
    FASP_ENTRY();
    if (!A || !b || !x || !param) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    int st = check_supported(nullptr, param);
    if (st < 0) return st;
    fasp_hip_amg* h = nullptr;
    g_oneshot_upload = std::getenv("FASP_HIP_ONESHOT_CODING") == nullptr;
    st = fasp_hip_amg_create(&h, A, param);
    g_oneshot_upload = false;
    if (st < 0) return st;
    st = fasp_hip_amg_solve(h, b, x, param, nullptr, 0, nullptr);
    if (param->print_level > PRINT_NONE) std::printf("AMG totally costs %.4f seconds.\n", wall_seconds() - t0);
    fasp_hip_amg_destroy(h);
    return st;
}

// SolFAMG.c:41 -> fasp_famg_solve (PreMGSolve.c:300): ONE full-multigrid cycle as the solver; x is the
// initial guess of the finest level and receives the result.  void in the reference; the status is an extension.
int fasp_solver_famg(const dCSRmat* A, const dvector* b, dvector* x, AMG_param* param)
{
    FASP_ENTRY();
    if (!A || !b || !x || !param) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    int st = check_supported(nullptr, param);
    if (st < 0) return st;
    fasp_hip_amg* h = nullptr;
    g_oneshot_upload = std::getenv("FASP_HIP_ONESHOT_CODING") == nullptr;
    st = fasp_hip_amg_create(&h, A, param);
    g_oneshot_upload = false;
    if (st < 0) return st;
    if ((st = fasp_hip_set_rhs(h, b)) >= 0 && (st = fasp_hip_set_guess(h, x)) >= 0) {
        DevLevel& D0 = h->L[0];
        const int m = D0.A.row;
        double red[2];
        D0.b = h->b;
        (void)hipMemcpyAsync(D0.x, h->u, sizeof(double) * m, hipMemcpyDeviceToDevice, g_ctx.stream);
        D0.x_zero = false;
        st = d_dot(m, h->b, h->b, red, false) < 0 ? ERROR_MISC : FASP_SUCCESS;
        const double sumb = std::sqrt(red[0]);
        if (st >= 0 && sumb <= SMALLREAL) (void)hipMemsetAsync(D0.x, 0, sizeof(double) * m, g_ctx.stream);
        if (st >= 0) st = fmg_cycle(h, *param);
        if (st >= 0) {
            d_resid(D0.A, D0.x, D0.b, D0.w);
            if (d_dot(m, D0.w, D0.w, red, false) < 0) st = ERROR_MISC;
            else if (param->print_level > PRINT_NONE)
                std::printf("FMG finishes with relative residual %e.\n", std::sqrt(red[0]) / std::max(SMALLREAL, sumb));
            (void)hipMemcpyAsync(h->u, D0.x, sizeof(double) * m, hipMemcpyDeviceToDevice, g_ctx.stream);
            const int st2 = fasp_hip_get_solution(h, x);
            if (st2 < 0) st = st2;
        }
    }
    if (param->print_level > PRINT_NONE) std::printf("FAMG totally costs %.4f seconds.\n", wall_seconds() - t0);
    fasp_hip_amg_destroy(h);
    return st;
}

int fasp_hip_precond_amg(fasp_hip_amg* h, const double* r, double* z)
{
    FASP_ENTRY();
    if (!h || !r || !z || h->L.empty()) return ERROR_INPUT_PAR;
    const int m = h->L[0].nloc;
    r += h->L[0].row0; z += h->L[0].row0;  // global host vectors, own rows
    HIPCK(hipMemcpyAsync(h->r, r, sizeof(double) * m, hipMemcpyHostToDevice, g_ctx.stream));
    double* dz = nullptr;
    const int st = precond_amg(h, h->r, &dz);
    if (st < 0) return st;
    HIPCK(hipMemcpyAsync(z, dz, sizeof(double) * m, hipMemcpyDeviceToHost, g_ctx.stream));
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    return seq_err_check();
}

// ---- row-partition inspection (host only; used by the CPU-side distributed tests) ----
int fasp_hip_dist_plan(fasp_hip_amg* h, int rank, int nranks, int min_rows)
{
    FASP_ENTRY();
    if (!h) return ERROR_INPUT_PAR;
    return build_dist_plan(h->H, rank, nranks, min_rows, h->dist);
}

int fasp_hip_dist_level_info(const fasp_hip_amg* h, int level, int* info)
{
    FASP_ENTRY();
    if (!h || !info || level < 0 || level >= (int)h->dist.L.size()) return ERROR_INPUT_PAR;
    const DistLevel& D = h->dist.L[level];
    info[0] = D.replicated; info[1] = D.nglobal; info[2] = D.row0; info[3] = D.nloc;
    info[4] = (int)D.ghosts.size(); info[5] = (int)D.send_idx.size();
    info[6] = h->dist.first_replicated; info[7] = h->dist.nranks;
    return FASP_SUCCESS;
}

int fasp_hip_dist_get_matrix(const fasp_hip_amg* h, int level, int which, dCSRmat* view)
{
    FASP_ENTRY();
    if (!h || !view || level < 0 || level >= (int)h->dist.L.size()) return ERROR_INPUT_PAR;
    const DistLevel& D = h->dist.L[level];
    if (D.replicated) return fasp_hip_amg_get_matrix(h, level, which, view);
    const HostCSR& M = which == 0 ? D.A : which == 1 ? D.P : D.R;
    if (!M.ia.data()) return ERROR_INPUT_PAR;
    *view = M.view();
    return FASP_SUCCESS;
}

int fasp_hip_dist_get_list(const fasp_hip_amg* h, int level, int which, ivector* view)
{
    FASP_ENTRY();
    if (!h || !view || level < 0 || level >= (int)h->dist.L.size()) return ERROR_INPUT_PAR;
    const DistLevel& D = h->dist.L[level];
    if (which >= 5 && which <= 7) {   // interior window [lo, hi) of the local A (5), P (6), R (7); hi < 0: none
        const int* w = which == 5 ? D.winA : which == 6 ? D.winP : D.winR;
        view->row = 2;
        view->val = const_cast<int*>(w);
        return FASP_SUCCESS;
    }
    const std::vector<int>* v = which == 0 ? &D.ghosts : which == 1 ? &D.recv_off : which == 2 ? &D.send_off
                              : which == 3 ? &D.send_idx : which == 4 ? &D.start : nullptr;
    if (!v) return ERROR_INPUT_PAR;
    view->row = (int)v->size();
    view->val = const_cast<int*>(v->data());
    return FASP_SUCCESS;
}

// SolCSR.c:476 -- the drop-in entry point
int fasp_solver_dcsr_krylov_amg(dCSRmat* A, dvector* b, dvector* x, ITS_param* itparam, AMG_param* amgparam)
{
    FASP_ENTRY();
    if (!A || !b || !x || !itparam || !amgparam) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    int st = check_supported(itparam, amgparam);
    if (st < 0) return st;
    fasp_hip_amg* h = nullptr;
    g_oneshot_upload = std::getenv("FASP_HIP_ONESHOT_CODING") == nullptr;  // (set the variable to keep the coding)
    st = fasp_hip_amg_create(&h, A, amgparam);
    g_oneshot_upload = false;
    if (st < 0) return st;
    st = fasp_hip_solve(h, b, x, itparam, nullptr, 0, nullptr);
    if (itparam->print_level >= PRINT_MIN)
        std::printf("AMG_Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    fasp_hip_amg_destroy(h);
    return st;
}

// ---------------------------------------------------------------------------
// BSR path: resident hierarchy + the drop-in of SolBSR.c:349
// ---------------------------------------------------------------------------
void fasp_hip_bsr_amg_destroy(fasp_hip_amg_bsr* h)
{
    FASP_ENTRY();
    if (!h) return;
    if (g_ctx.ready) (void)hipStreamSynchronize(g_ctx.stream);
    for (auto& Lv : h->L) {
        double* v[] = {Lv.dinv, Lv.b, Lv.x, Lv.x2, Lv.w};
        for (double* q : v) if (q) (void)hipFree(q);
        for (auto& sc : Lv.sched) sc.release();
        if (Lv.d_send_idx) (void)hipFree(Lv.d_send_idx);
        if (Lv.d_sendbuf) (void)hipFree(Lv.d_sendbuf);
    }
    double* v[] = {h->b, h->u, h->p, h->t, h->r};
    for (double* q : v) if (q) (void)hipFree(q);
    for (int s = 0; s < 2; ++s)
        for (double* q : h->gm[s]) if (q) (void)hipFree(q);
    if (h->gm_hh) (void)hipFree(h->gm_hh);
    if (h->small_ws) (void)hipFree(h->small_ws);
    delete h;
}

// host part only (no GPU needed): the hierarchy can be inspected, not solved with
int fasp_hip_bsr_amg_create_host(fasp_hip_amg_bsr** out, const dBSRmat* A, AMG_param* amgparam)
{
    FASP_ENTRY();
    if (!out || !A || !amgparam) return ERROR_INPUT_PAR;
    *out = nullptr;
    int st = check_supported_bsr(nullptr, amgparam, A->nb);
    if (st < 0) return st;
    fasp_hip_amg_bsr* h = new fasp_hip_amg_bsr();
    st = host_setup_ua_bsr(A, amgparam, h->H);
    if (st < 0) { delete h; return st; }
    h->param = *amgparam;
    *out = h;
    return FASP_SUCCESS;
}

int fasp_hip_bsr_amg_create(fasp_hip_amg_bsr** out, const dBSRmat* A, AMG_param* amgparam)
{
    FASP_ENTRY();
    if (!out || !A || !amgparam) return ERROR_INPUT_PAR;
    *out = nullptr;
    int st = check_supported_bsr(nullptr, amgparam, A->nb);
    if (st < 0) return st;
    if ((st = ctx_init()) < 0) return st;
    fasp_hip_amg_bsr* h = nullptr;
    if ((st = fasp_hip_bsr_amg_create_host(&h, A, amgparam)) < 0) return st;
    const int nl = (int)h->H.L.size();
    h->L.resize(nl);
    // one process per GPU: block rows partitioned over the ranks (dist_plan.cpp, build_dist_plan_bsr); levels below
    // FASP_HIP_DIST_MIN_ROWS block rows -- and always the coarsest one -- are kept whole on every rank.  Block Gauss-Seidel /
    // SOR sweeps couple all rows: such hierarchies stay whole.
    std::vector<DistLocalBSR> local;
    if (comm_size() > 1) {
        int min_rows = 200000;
        if (const char* e = std::getenv("FASP_HIP_DIST_MIN_ROWS")) min_rows = std::atoi(e);
        if (h->param.smoother != SMOOTHER_JACOBI) min_rows = 2147483647;
        min_rows = std::max(min_rows, h->H.L[(size_t)nl - 1].A.ROW + 1);
        HostThreads team;
        if ((st = build_dist_plan_bsr(h->H, comm_rank(), comm_size(), min_rows, h->dist, local)) < 0) { fasp_hip_bsr_amg_destroy(h); return st; }
        h->distributed = !h->dist.L[0].replicated;
    }
    for (int l = 0; l < nl; ++l) {
        const HostLevelBSR& HL = h->H.L[l];
        BsrLevel& Lv = h->L[l];
        const int nb = HL.A.nb, nb2 = nb * nb;
        const DistLevel* DL = (comm_size() > 1 && !h->dist.L[(size_t)l].replicated) ? &h->dist.L[(size_t)l] : nullptr;
        Lv.replicated = DL == nullptr;
        Lv.nglobal = HL.A.ROW;
        Lv.row0 = DL ? DL->row0 : 0; Lv.nloc = DL ? DL->nloc : HL.A.ROW; Lv.nghost = DL ? (int)DL->ghosts.size() : 0;
        const dBSRmat vA = DL ? local[(size_t)l].A.view() : HL.A.view();
        Lv.A.reset(new TmpBSR(&vA));
        bool ok = Lv.A->ok;
        Lv.n = Lv.nloc * nb;
        Lv.nv = (Lv.nloc + Lv.nghost) * nb;
        if (HL.has_coarse) {
            const dBSRmat vP = DL ? local[(size_t)l].P.view() : HL.P.view(), vR = DL ? local[(size_t)l].R.view() : HL.R.view();
            Lv.P.reset(new TmpBSR(&vP));
            Lv.R.reset(new TmpBSR(&vR));
            ok = ok && Lv.P->ok && Lv.R->ok;
            const size_t nd = (size_t)Lv.nloc * nb2;
            if (hipMalloc(&Lv.dinv, sizeof(double) * std::max<size_t>(nd, 1)) != hipSuccess) ok = false;
            else (void)hipMemcpy(Lv.dinv, HL.diaginv.data() + (size_t)Lv.row0 * nb2, sizeof(double) * nd, hipMemcpyHostToDevice);
        }
        if (ok && DL) {   // halo lists, expanded from blocks to scalars
            const int P = comm_size();
            Lv.send_off.assign((size_t)P + 1, 0); Lv.recv_off.assign((size_t)P + 1, 0);
            for (int q = 0; q <= P; ++q) { Lv.send_off[(size_t)q] = DL->send_off[(size_t)q] * nb; Lv.recv_off[(size_t)q] = DL->recv_off[(size_t)q] * nb; }
            std::vector<int> sidx(DL->send_idx.size() * (size_t)nb);
            for (size_t i = 0; i < DL->send_idx.size(); ++i)
                for (int c = 0; c < nb; ++c) sidx[i * nb + c] = DL->send_idx[i] * nb + c;
            if (hipMalloc(&Lv.d_send_idx, sizeof(int) * std::max<size_t>(sidx.size(), 1)) != hipSuccess ||
                hipMalloc(&Lv.d_sendbuf, sizeof(double) * std::max<size_t>(sidx.size(), 1)) != hipSuccess) ok = false;
            else if (!sidx.empty()) (void)hipMemcpy(Lv.d_send_idx, sidx.data(), sizeof(int) * sidx.size(), hipMemcpyHostToDevice);
        }
        if (!ok || dalloc(&Lv.b, Lv.nv) < 0 || dalloc(&Lv.x, Lv.nv) < 0 || dalloc(&Lv.x2, Lv.nv) < 0 ||
            dalloc(&Lv.w, Lv.nv) < 0) {
            fasp_hip_bsr_amg_destroy(h);
            return ERROR_ALLOC_MEM;
        }
    }
    const size_t n0 = (size_t)h->L[0].nv;
    if (dalloc(&h->b, n0) < 0 || dalloc(&h->u, n0) < 0 || dalloc(&h->p, n0) < 0 || dalloc(&h->t, n0) < 0 ||
        dalloc(&h->r, n0) < 0) {
        fasp_hip_bsr_amg_destroy(h);
        return ERROR_ALLOC_MEM;
    }
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    *out = h;
    return FASP_SUCCESS;
}

// block rows of level 0 this rank owns: info[0] = 1 if the level is whole on every rank, [1] first owned block row,
// [2] owned block rows, [3] ghost blocks, [4] first level kept whole, [5] block size
int fasp_hip_bsr_dist_info(const fasp_hip_amg_bsr* h, int* info)
{
    FASP_ENTRY();
    if (!h || !info || h->L.empty()) return ERROR_INPUT_PAR;
    const BsrLevel& L0 = h->L[0];
    int first_rep = 0;
    while (first_rep < (int)h->L.size() && !h->L[(size_t)first_rep].replicated) ++first_rep;
    info[0] = L0.replicated ? 1 : 0; info[1] = L0.row0; info[2] = L0.nloc; info[3] = L0.nghost; info[4] = first_rep; info[5] = L0.A->nb;
    return FASP_SUCCESS;
}

int fasp_hip_bsr_amg_num_levels(const fasp_hip_amg_bsr* h) { return h ? (int)h->H.L.size() : ERROR_INPUT_PAR; }

int fasp_hip_bsr_amg_get_matrix(const fasp_hip_amg_bsr* h, int level, int which, dBSRmat* view)
{
    FASP_ENTRY();
    if (!h || !view || level < 0 || level >= (int)h->H.L.size()) return ERROR_INPUT_PAR;
    const HostLevelBSR& L = h->H.L[level];
    if (which != 0 && !L.has_coarse) return ERROR_INPUT_PAR;
    *view = which == 0 ? L.A.view() : which == 1 ? L.P.view() : L.R.view();
    return FASP_SUCCESS;
}

const double* fasp_hip_bsr_amg_get_diaginv(const fasp_hip_amg_bsr* h, int level)
{
    FASP_ENTRY();
    if (!h || level < 0 || level >= (int)h->H.L.size() || !h->H.L[level].has_coarse) return nullptr;
    return h->H.L[level].diaginv.data();
}

int fasp_hip_bsr_solve(fasp_hip_amg_bsr* h, const dvector* b, dvector* x, const ITS_param* itparam, double* hist,
                       int hist_cap, fasp_hip_stats* stats)
{
    FASP_ENTRY();
    if (!h || !b || !x || !itparam || h->L.empty()) return ERROR_INPUT_PAR;
    const int n = h->L[0].n;                                    // owned scalar rows
    const int nglob = h->L[0].nglobal * h->H.L[0].A.nb;          // b and x are the GLOBAL vectors: a rank reads / fills its rows
    const size_t off0 = (size_t)h->L[0].row0 * h->H.L[0].A.nb;
    if (b->row != nglob || x->row != nglob) return ERROR_MAT_SIZE;
    int st = check_supported_bsr(itparam, &h->param, h->H.L[0].A.nb);
    if (st < 0) return st;
    if (itparam->tol < SMALLREAL)
        std::printf("### WARNING: Convergence tolerance is too small! [%s:%d]\n", "ITS_CHECK", 74);
    if (itparam->maxit <= 0)
        std::printf("### WARNING: Max number of iterations must be POSITIVE! [%s:%d]\n", "ITS_CHECK", 78);
    hipStream_t s = g_ctx.stream;
    double t0 = wall_seconds();
    HIPCK(hipMemcpyAsync(h->b, b->val + off0, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(h->u, x->val + off0, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIPCK(hipStreamSynchronize(s));
    double t_up = wall_seconds() - t0;

    const long long ci0 = h->coarse_iters, vc0 = h->vcycles;
    Hist   H{hist, hist_cap, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    t0 = wall_seconds();
    KOps K = bsr_ops(h, 0, 0);
    switch (itparam->itsolver_type) {  // fasp_solver_dbsr_itsolver, SolBSR.c:55-150
        case SOLVER_BiCGstab:
            st = bicgstab_device(K, h->b, h->u, itparam->tol, itparam->maxit, itparam->print_level, &H, &po);
            break;
        case SOLVER_GMRES:
        case SOLVER_VGMRES:
        case SOLVER_VFGMRES:
            st = gmres_device(K, h->b, h->u, itparam->itsolver_type == SOLVER_VFGMRES ? 1 : itparam->itsolver_type == SOLVER_GMRES ? 3 : 0, itparam->tol,
                              itparam->abstol, itparam->maxit, (short)itparam->restart, itparam->stop_type,
                              itparam->print_level, &H, &po);
            break;
        default:
        {
            PcgVecs V{h->b, h->u, h->p, h->t, h->r};
            st = pcg_device(K, V, itparam->tol, itparam->abstol, itparam->maxit, itparam->stop_type,
                            itparam->print_level, H, po);
        }
    }
    HIPCK(hipStreamSynchronize(s));
    const double t_solve = wall_seconds() - t0;
    t0 = wall_seconds();
    HIPCK(hipMemcpy(x->val + off0, h->u, sizeof(double) * n, hipMemcpyDeviceToHost));
    t_up += wall_seconds() - t0;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->iters = st; stats->nhist = H.n; stats->relres = po.relres; stats->absres = po.absres;
        stats->normr0 = po.normr0; stats->solve_seconds = t_solve; stats->upload_seconds = t_up;
        stats->coarse_iters = h->coarse_iters - ci0;
        stats->vcycles = h->vcycles - vc0;
    }
    if (itparam->print_level >= PRINT_SOME && st >= 0)
        std::printf("Iterative method costs %.4f seconds.\n", t_solve);
    return st;
}

// SolBSR.c:349: UA-AMG setup on the host, hierarchy uploaded, Krylov loop on the device
int fasp_solver_dbsr_krylov_amg(dBSRmat* A, dvector* b, dvector* x, ITS_param* itparam, AMG_param* amgparam)
{
    FASP_ENTRY();
    if (!A || !b || !x || !itparam || !amgparam) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    int st = check_supported_bsr(itparam, amgparam, A->nb);
    if (st < 0) return st;
    fasp_hip_amg_bsr* h = nullptr;
    st = fasp_hip_bsr_amg_create(&h, A, amgparam);
    if (st < 0) return st;
    st = fasp_hip_bsr_solve(h, b, x, itparam, nullptr, 0, nullptr);
    if (itparam->print_level >= PRINT_MIN)
        std::printf("AMG_Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    fasp_hip_bsr_amg_destroy(h);
    return st;
}

// ---------------------------------------------------------------------------
// kernel-level operators with host pointers (reference names, device kernels)
// ---------------------------------------------------------------------------
namespace {
struct TmpCSR {
    DevCSR D;
    bool   ok = false;
    explicit TmpCSR(const dCSRmat* A)
    {
        if (ctx_init() < 0) return;
        D.row = A->row; D.col = A->col; D.nnz = A->nnz;
        if (hipMalloc(&D.ia, sizeof(int) * ((size_t)A->row + 1)) != hipSuccess) return;
        if (hipMalloc(&D.ja, sizeof(int) * std::max(A->nnz, 1)) != hipSuccess) return;
        if (hipMalloc(&D.val, sizeof(double) * std::max(A->nnz, 1)) != hipSuccess) return;
        (void)hipMemcpy(D.ia, A->IA, sizeof(int) * ((size_t)A->row + 1), hipMemcpyHostToDevice);
        (void)hipMemcpy(D.ja, A->JA, sizeof(int) * (size_t)A->nnz, hipMemcpyHostToDevice);
        (void)hipMemcpy(D.val, A->val, sizeof(double) * (size_t)A->nnz, hipMemcpyHostToDevice);
        pick_kernel(D);
        ok = true;
    }
    ~TmpCSR() { D.release(); }
};
struct TmpVec {
    double* d = nullptr;
    size_t  n;
    TmpVec(const double* h, size_t n_) : n(n_)
    {
        if (hipMalloc(&d, sizeof(double) * std::max<size_t>(n, 1)) != hipSuccess) { d = nullptr; return; }
        if (h) (void)hipMemcpy(d, h, sizeof(double) * n, hipMemcpyHostToDevice);
    }
    void get(double* h) { (void)hipStreamSynchronize(g_ctx.stream); (void)hipMemcpy(h, d, sizeof(double) * n, hipMemcpyDeviceToHost); }
    ~TmpVec() { if (d) (void)hipFree(d); }
};
[[noreturn]] void die_no_device(const char* fn)
{
    std::fprintf(stderr, "### ERROR: %s: no usable HIP device and no CPU fallback in libfasp_hip\n", fn);
    std::exit(ERROR_MISC);
}
}  // namespace

// ---------------------------------------------------------------------------
// plug-in level of the reference: the Krylov methods with a caller
// supplied `precond` (fasp.h:1095), and the AMG preconditioner as such a plug-in
// ---------------------------------------------------------------------------
// PreCSR.c:416 signature: z = B r, host vectors; data is the fasp_hip_amg* of fasp_hip_precond_setup
void fasp_hip_precond_fct(double* r, double* z, void* data)
{
    FASP_ENTRY();
    fasp_hip_amg* h = static_cast<fasp_hip_amg*>(data);
    if (fasp_hip_precond_amg(h, r, z) < 0) {
        std::fprintf(stderr, "### ERROR: fasp_hip_precond_fct: device preconditioner failed\n");
        std::exit(ERROR_MISC);
    }
}

// PreCSR.c:46 for PREC_AMG: hierarchy built and uploaded once, handed out as a `precond`
precond* fasp_hip_precond_setup(dCSRmat* A, AMG_param* amgparam)
{
    FASP_ENTRY();
    fasp_hip_amg* h = nullptr;
    if (fasp_hip_amg_create(&h, A, amgparam) < 0) return nullptr;
    precond* pc = static_cast<precond*>(std::calloc(1, sizeof(precond)));
    pc->data = h;
    pc->fct = fasp_hip_precond_fct;
    return pc;
}

#include "precond_api.hip.h"

void fasp_hip_precond_free(precond* pc)
{
    FASP_ENTRY();
    if (!pc) return;
    if (pc->fct == fasp_hip_precond_fct) fasp_hip_amg_destroy(static_cast<fasp_hip_amg*>(pc->data));
    std::free(pc);
}

namespace {
bool same_host_matrix(const HostCSR& M, const dCSRmat* A)
{
    return M.row == A->row && M.col == A->col && M.nnz == A->nnz &&
           std::memcmp(M.ia.data(), A->IA, sizeof(int) * ((size_t)A->row + 1)) == 0 &&
           std::memcmp(M.ja.data(), A->JA, sizeof(int) * (size_t)A->nnz) == 0 &&
           std::memcmp(M.val.data(), A->val, sizeof(double) * (size_t)A->nnz) == 0;
}

// which: 0 PCG, 1 VGMRES, 2 VFGMRES, 3 BiCGstab, 4 GMRES (fixed restart), 5 MinRes, 6 GCG, 7 GCR
int krylov_plugin(const char* fn, int which, dCSRmat* A, dvector* b, dvector* u, precond* pc, double tol,
                  double abstol, int MaxIt, short restart, short StopType, short PrtLvl)
{
    if (ctx_init() < 0) die_no_device(fn);
    if (!A || !b || !u || A->row != A->col || b->row != A->row || u->row != A->row) return ERROR_INPUT_PAR;
    if (comm_size() > 1) return ERROR_INPUT_PAR;  // plug-in level: one GPU
    const int n = b->row;
    fasp_hip_amg* h = (pc && pc->fct == fasp_hip_precond_fct) ? static_cast<fasp_hip_amg*>(pc->data) : amg_handle_of_precond(pc);
    if (h && (h->L.empty() || h->L[0].A.row != n)) return ERROR_INPUT_PAR;
    std::unique_ptr<TmpCSR> own;
    const DevCSR* dA = nullptr;
    if (h && same_host_matrix(h->H.L[0].A, A)) dA = &h->L[0].A;  // the resident level-0 operator is A itself
    else {
        own.reset(new TmpCSR(A));
        if (!own->ok) return ERROR_ALLOC_MEM;
        dA = &own->D;
    }
    TmpVec db(b->val, n), du(u->val, n), dp(nullptr, n), dt(nullptr, n), dr(nullptr, n), dz(nullptr, n);
    if (!db.d || !du.d || !dp.d || !dt.d || !dr.d || !dz.d) return ERROR_ALLOC_MEM;
    std::vector<double> hr, hz;
    std::vector<double*> ws;
    size_t ws_len = 0;
    double* hh = nullptr;
    KOps K;
    K.n = n; K.nvec = (size_t)n; K.fmt = "CSR"; K.dist = false;
    K.halo = [](double*) { return 0; };
    K.mxv = [dA](const double* x, double* y) { d_mxv(*dA, x, y); };
    K.resid = [dA](const double* x, const double* bb, double* r) { d_resid(*dA, x, bb, r); };
    K.mxv_dot = [dA](const double* x, double* y) {
        CsrArgs a{}; a.x = x; a.y = y; a.dotv = x; a.partials = g_ctx.d_partials;
        return launch_csr<OP_MXV_DOT>(*dA, a);
    };
    std::unique_ptr<TmpVec> ddiag;
    if (h) {
        K.pc = [h](double* in, double** out) { return precond_amg(h, in, out); };  // stays in HBM
    } else if (pc && pc->fct == fasp_precond_diag && pc->data && static_cast<dvector*>(pc->data)->row == n) {
        // the reference's diagonal preconditioner: recognised by its function pointer, applied on the device
        ddiag.reset(new TmpVec(static_cast<dvector*>(pc->data)->val, n));
        if (!ddiag->d) return ERROR_ALLOC_MEM;
        const double* dd = ddiag->d;
        double* zz = dz.d;
        K.pc = [dd, zz, n](double* in, double** out) {
            hipLaunchKernelGGL(k_diag_precond, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, dd, (const double*)in, zz);
            *out = zz;
            return 0;
        };
    } else if (pc && pc->fct) {
        // foreign preconditioner: a host function; the residual is staged through host memory
        hr.resize((size_t)n); hz.resize((size_t)n);
        K.pc = [&, pc](double* in, double** out) {
            HIPCK(hipMemcpyAsync(hr.data(), in, sizeof(double) * n, hipMemcpyDeviceToHost, g_ctx.stream));
            HIPCK(hipStreamSynchronize(g_ctx.stream));
            pc->fct(hr.data(), hz.data(), pc->data);
            HIPCK(hipMemcpyAsync(dz.d, hz.data(), sizeof(double) * n, hipMemcpyHostToDevice, g_ctx.stream));
            *out = dz.d;
            return 0;
        };
    }
    K.ws = &ws; K.ws_len = &ws_len; K.hh = &hh;
    K.stats = nullptr;
    Hist   H{nullptr, 0, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    int st;
    if (which == 0) {
        PcgVecs V{db.d, du.d, dp.d, dt.d, dr.d};
        st = pcg_device(K, V, tol, abstol, MaxIt, StopType, PrtLvl, H, po);
    } else if (which == 3) {
        st = bicgstab_device(K, db.d, du.d, tol, MaxIt, PrtLvl, &H, &po);
    } else if (which == 5) {
        st = minres_device(K, db.d, du.d, tol, abstol, MaxIt, StopType, PrtLvl, &H, &po);
    } else if (which == 6) {
        st = gcg_device(K, db.d, du.d, tol, abstol, MaxIt, StopType, PrtLvl, &H, &po);
    } else if (which == 7) {
        st = gcr_device(K, db.d, du.d, tol, abstol, MaxIt, restart, StopType, PrtLvl, &H, &po);
    } else {
        st = gmres_device(K, db.d, du.d, which == 2 ? 1 : which == 4 ? 3 : 0, tol, abstol, MaxIt, restart, StopType, PrtLvl, &H, &po);
    }
    du.get(u->val);
    for (double* q : ws) if (q) (void)hipFree(q);
    if (hh) (void)hipFree(hh);
    return st;
}
}  // namespace

// KryPcg.c:96
int fasp_solver_dcsr_pcg(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                         const int MaxIt, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin(__func__, 0, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
// KryPgmres.c:66
int fasp_solver_dcsr_pgmres(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                            const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin(__func__, 4, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
// KryPvgmres.c:66
int fasp_solver_dcsr_pvgmres(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                             const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin(__func__, 1, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
// KryPbcgs.c:62
int fasp_solver_dcsr_pbcgs(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                           const int MaxIt, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin(__func__, 3, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
// KryPminres.c:61
int fasp_solver_dcsr_pminres(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                             const int MaxIt, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin(__func__, 5, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
// KryPgcg.c:60
int fasp_solver_dcsr_pgcg(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                          const int MaxIt, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin(__func__, 6, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
// KryPgcr.c:55
int fasp_solver_dcsr_pgcr(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                          const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin(__func__, 7, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
// KryPvfgmres.c:67
int fasp_solver_dcsr_pvfgmres(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                              const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin(__func__, 2, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}

// ---- the same plug-in level for block matrices ---------------------------------
// z = B r with the resident block hierarchy (PreBSR.c:1149 signature); data = fasp_hip_amg_bsr*
void fasp_hip_bsr_precond_fct(double* r, double* z, void* data)
{
    FASP_ENTRY();
    fasp_hip_amg_bsr* h = static_cast<fasp_hip_amg_bsr*>(data);
    if (!h || h->L.empty() || ctx_init() < 0) die_bsr(__func__);
    const int n = h->L[0].n;
    TmpVec dr(r, (size_t)n);
    double* dz = nullptr;
    if (!dr.d || precond_amg_bsr(h, dr.d, &dz) < 0) die_bsr(__func__);
    (void)hipStreamSynchronize(g_ctx.stream);
    (void)hipMemcpy(z, dz, sizeof(double) * n, hipMemcpyDeviceToHost);
}

precond* fasp_hip_bsr_precond_setup(dBSRmat* A, AMG_param* amgparam)
{
    FASP_ENTRY();
    fasp_hip_amg_bsr* h = nullptr;
    if (fasp_hip_bsr_amg_create(&h, A, amgparam) < 0) return nullptr;
    precond* pc = static_cast<precond*>(std::calloc(1, sizeof(precond)));
    pc->data = h;
    pc->fct = fasp_hip_bsr_precond_fct;
    return pc;
}

void fasp_hip_bsr_precond_free(precond* pc)
{
    FASP_ENTRY();
    if (!pc) return;
    if (pc->fct == fasp_hip_bsr_precond_fct) fasp_hip_bsr_amg_destroy(static_cast<fasp_hip_amg_bsr*>(pc->data));
    std::free(pc);
}

namespace {
int krylov_plugin_bsr(const char* fn, int which, dBSRmat* A, dvector* b, dvector* u, precond* pc, double tol,
                      double abstol, int MaxIt, short restart, short StopType, short PrtLvl)
{
    if (!A || !b || !u || A->ROW != A->COL || b->row != A->ROW * A->nb || u->row != b->row) return ERROR_INPUT_PAR;
    TmpBSR M(A);
    if (!M.ok) die_bsr(fn);
    const int n = b->row;
    fasp_hip_amg_bsr* h = (pc && pc->fct == fasp_hip_bsr_precond_fct) ? static_cast<fasp_hip_amg_bsr*>(pc->data) : nullptr;
    if (h && (h->L.empty() || h->L[0].n != n)) return ERROR_INPUT_PAR;
    TmpVec db(b->val, n), du(u->val, n), dp(nullptr, n), dt(nullptr, n), dr(nullptr, n), dz(nullptr, n);
    if (!db.d || !du.d || !dp.d || !dt.d || !dr.d || !dz.d) return ERROR_ALLOC_MEM;
    std::vector<double> hr, hz;
    std::vector<double*> ws;
    size_t ws_len = 0;
    double* hh = nullptr;
    KOps K;
    K.n = n; K.nvec = (size_t)n; K.fmt = "BSR"; K.dist = false;
    K.halo = [](double*) { return 0; };
    const TmpBSR* Mp = &M;
    K.mxv = [Mp](const double* x, double* y) { bsr_mxv(*Mp, x, y); };
    K.resid = [Mp](const double* x, const double* bb, double* r) { bsr_resid(*Mp, x, bb, r); };
    std::unique_ptr<TmpVec> ddiag;
    if (h) {
        K.pc = [h](double* in, double** out) { return precond_amg_bsr(h, in, out); };
    } else if (pc && pc->fct == fasp_precond_dbsr_diag && pc->data &&
               static_cast<precond_diag_bsr*>(pc->data)->diag.row == A->ROW * A->nb * A->nb) {
        // block-diagonal preconditioner of the reference (PreBSR.c:49): z_i = Dinv_i r_i on the device
        ddiag.reset(new TmpVec(static_cast<precond_diag_bsr*>(pc->data)->diag.val, (size_t)A->ROW * A->nb * A->nb));
        if (!ddiag->d) return ERROR_ALLOC_MEM;
        const double* dd = ddiag->d;
        double* zz = dz.d;
        const int nb = A->nb;
        K.pc = [dd, zz, n, nb](double* in, double** out) {
            hipLaunchKernelGGL(k_bsr_dinv_apply, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, nb, dd, (const double*)in, zz);
            *out = zz;
            return 0;
        };
    } else if (pc && pc->fct) {
        hr.resize((size_t)n); hz.resize((size_t)n);
        K.pc = [&, pc](double* in, double** out) {
            HIPCK(hipMemcpyAsync(hr.data(), in, sizeof(double) * n, hipMemcpyDeviceToHost, g_ctx.stream));
            HIPCK(hipStreamSynchronize(g_ctx.stream));
            pc->fct(hr.data(), hz.data(), pc->data);
            HIPCK(hipMemcpyAsync(dz.d, hz.data(), sizeof(double) * n, hipMemcpyHostToDevice, g_ctx.stream));
            *out = dz.d;
            return 0;
        };
    }
    K.ws = &ws; K.ws_len = &ws_len; K.hh = &hh;
    Hist   H{nullptr, 0, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    int st;
    if (which == 0) {
        PcgVecs V{db.d, du.d, dp.d, dt.d, dr.d};
        st = pcg_device(K, V, tol, abstol, MaxIt, StopType, PrtLvl, H, po);
    } else if (which == 3) {
        st = bicgstab_device(K, db.d, du.d, tol, MaxIt, PrtLvl, &H, &po);
    } else {
        st = gmres_device(K, db.d, du.d, which == 2 ? 1 : which == 4 ? 3 : 0, tol, abstol, MaxIt, restart, StopType, PrtLvl, &H, &po);
    }
    du.get(u->val);
    for (double* q : ws) if (q) (void)hipFree(q);
    if (hh) (void)hipFree(hh);
    return st;
}
}  // namespace

// KryPcg.c:386, KryPbcgs.c:400, KryPgmres.c:357, KryPvgmres.c:416, KryPvfgmres.c:386
int fasp_solver_dbsr_pcg(dBSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                         const int MaxIt, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin_bsr(__func__, 0, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
int fasp_solver_dbsr_pbcgs(dBSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                           const int MaxIt, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin_bsr(__func__, 3, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
int fasp_solver_dbsr_pgmres(dBSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                            const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin_bsr(__func__, 4, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
int fasp_solver_dbsr_pvgmres(dBSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                             const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin_bsr(__func__, 1, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
int fasp_solver_dbsr_pvfgmres(dBSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                              const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_plugin_bsr(__func__, 2, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}

// ---------------------------------------------------------------------------
// The other solver-level entry points of SolCSR.c / SolBSR.c: dispatch on itsolver_type with a caller's
// preconditioner, no preconditioner, or the (block-)diagonal one.
// ---------------------------------------------------------------------------
void fasp_precond_diag(double* r, double* z, void* data)  // PreCSR.c:172 (host arrays)
{
    FASP_ENTRY();
    const dvector* diag = static_cast<const dvector*>(data);
    std::memcpy(z, r, sizeof(double) * (size_t)diag->row);
    for (int i = 0; i < diag->row; ++i)
        if (std::fabs(diag->val[i]) > SMALLREAL) z[i] /= diag->val[i];
}
void fasp_precond_dbsr_diag(double* r, double* z, void* data)  // PreBSR.c:49 (host arrays): z_i = Dinv_i r_i
{
    FASP_ENTRY();
    const precond_diag_bsr* d = static_cast<const precond_diag_bsr*>(data);
    const int nb = d->nb, nb2 = nb * nb, m = d->diag.row / nb2;
    for (int i = 0; i < m; ++i)
        for (int rr = 0; rr < nb; ++rr) {
            const double* D = d->diag.val + (size_t)i * nb2 + rr * nb;
            double s = D[0] * r[(size_t)i * nb];
            for (int c = 1; c < nb; ++c) s = s + D[c] * r[(size_t)i * nb + c];
            z[(size_t)i * nb + rr] = s;
        }
}

// SolCSR.c:56
int fasp_solver_dcsr_itsolver(dCSRmat* A, dvector* b, dvector* x, precond* pc, ITS_param* itparam)
{
    FASP_ENTRY();
    if (!itparam) return ERROR_INPUT_PAR;
    const short prtlvl = itparam->print_level, stop_type = itparam->stop_type, restart = (short)itparam->restart;
    const int MaxIt = itparam->maxit;
    const double tol = itparam->tol, abstol = itparam->abstol, t0 = wall_seconds();
    int iter;
    if (tol < SMALLREAL) std::printf("### WARNING: Convergence tolerance is too small! [%s:%d]\n", "ITS_CHECK", 74);
    if (MaxIt <= 0) std::printf("### WARNING: Max number of iterations must be POSITIVE! [%s:%d]\n", "ITS_CHECK", 78);
    switch (itparam->itsolver_type) {
        case SOLVER_CG: iter = fasp_solver_dcsr_pcg(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_BiCGstab: iter = fasp_solver_dcsr_pbcgs(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_MinRes: iter = fasp_solver_dcsr_pminres(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_GMRES: iter = fasp_solver_dcsr_pgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        case SOLVER_VGMRES: iter = fasp_solver_dcsr_pvgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        case SOLVER_VFGMRES: iter = fasp_solver_dcsr_pvfgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        case SOLVER_GCG: iter = fasp_solver_dcsr_pgcg(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_GCR: iter = fasp_solver_dcsr_pgcr(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        default:
            std::printf("### ERROR: Unknown iterative solver type %d! [%s]\n", itparam->itsolver_type, __func__);
            return ERROR_SOLVER_TYPE;
    }
    if ((prtlvl >= PRINT_SOME) && (iter >= 0)) std::printf("Iterative method costs %.4f seconds.\n", wall_seconds() - t0);
    return iter;
}
// SolCSR.c:245
int fasp_solver_dcsr_krylov(dCSRmat* A, dvector* b, dvector* x, ITS_param* itparam)
{
    FASP_ENTRY();
    if (!itparam) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    const int status = fasp_solver_dcsr_itsolver(A, b, x, nullptr, itparam);
    if (itparam->print_level >= PRINT_MIN) std::printf("Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    return status;
}
// SolCSR.c:333: diagonal preconditioner from fasp_dcsr_getdiag(0, A, ..) -- the FIRST diagonal hit of each row
int fasp_solver_dcsr_krylov_diag(dCSRmat* A, dvector* b, dvector* x, ITS_param* itparam)
{
    FASP_ENTRY();
    if (!A || !itparam || !A->IA || !A->JA || !A->val) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    const int n = std::min(A->row, A->col);
    std::vector<double> dv((size_t)std::max(n, 1), 0.0);
    for (int i = 0; i < n; ++i)
        for (int k = A->IA[i]; k < A->IA[i + 1]; ++k)
            if (A->JA[k] == i) { dv[(size_t)i] = A->val[k]; break; }
    dvector diag{n, dv.data()};
    precond pc{&diag, fasp_precond_diag};
    const int status = fasp_solver_dcsr_itsolver(A, b, x, &pc, itparam);
    if (itparam->print_level >= PRINT_MIN) std::printf("Diag_Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    return status;
}
// SolBSR.c:64
int fasp_solver_dbsr_itsolver(dBSRmat* A, dvector* b, dvector* x, precond* pc, ITS_param* itparam)
{
    FASP_ENTRY();
    if (!itparam) return ERROR_INPUT_PAR;
    const short prtlvl = itparam->print_level, stop_type = itparam->stop_type, restart = (short)itparam->restart;
    const int MaxIt = itparam->maxit;
    const double tol = itparam->tol, abstol = itparam->abstol, t0 = wall_seconds();
    int iter;
    if (tol < SMALLREAL) std::printf("### WARNING: Convergence tolerance is too small! [%s:%d]\n", "ITS_CHECK", 74);
    if (MaxIt <= 0) std::printf("### WARNING: Max number of iterations must be POSITIVE! [%s:%d]\n", "ITS_CHECK", 78);
    switch (itparam->itsolver_type) {
        case SOLVER_CG: iter = fasp_solver_dbsr_pcg(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_BiCGstab: iter = fasp_solver_dbsr_pbcgs(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_GMRES: iter = fasp_solver_dbsr_pgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        case SOLVER_VGMRES: iter = fasp_solver_dbsr_pvgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        case SOLVER_VFGMRES: iter = fasp_solver_dbsr_pvfgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        default:
            std::printf("### ERROR: Unknown iterative solver type %d! [%s]\n", itparam->itsolver_type, __func__);
            return ERROR_SOLVER_TYPE;
    }
    if ((prtlvl >= PRINT_SOME) && (iter >= 0)) std::printf("Iterative method costs %.4f seconds.\n", wall_seconds() - t0);
    return iter;
}
// SolBSR.c:145
int fasp_solver_dbsr_krylov(dBSRmat* A, dvector* b, dvector* x, ITS_param* itparam)
{
    FASP_ENTRY();
    if (!itparam) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    const int status = fasp_solver_dbsr_itsolver(A, b, x, nullptr, itparam);
    if (itparam->print_level >= PRINT_MIN) std::printf("Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    return status;
}
// SolBSR.c:186: block-diagonal preconditioner, inverse blocks by fasp_smat_inv
int fasp_solver_dbsr_krylov_diag(dBSRmat* A, dvector* b, dvector* x, ITS_param* itparam)
{
    FASP_ENTRY();
    if (!A || !itparam || !A->IA || !A->JA || !A->val) return ERROR_INPUT_PAR;
    if (A->nb < 1 || A->nb > 7) {
        std::printf("### ERROR: fasp_hip: block-diagonal preconditioner needs 1 <= nb <= 7, got %d\n", A->nb);
        return ERROR_INPUT_PAR;
    }
    const double t0 = wall_seconds();
    const int nb2 = A->nb * A->nb;
    std::vector<double> dv((size_t)std::max(A->ROW, 1) * nb2, 0.0);
    const int st = bsr_diaginv(A, dv.data());
    if (st < 0) return st;
    precond_diag_bsr diag;
    diag.nb = A->nb; diag.diag.row = A->ROW * nb2; diag.diag.val = dv.data();
    precond pc{&diag, fasp_precond_dbsr_diag};
    const int status = fasp_solver_dbsr_itsolver(A, b, x, &pc, itparam);
    if (itparam->print_level > PRINT_NONE) std::printf("Diag_Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    return status;
}

// ---------------------------------------------------------------------------
// Matrix-free interface of the reference (fasp.h:1109 mxv_matfree, SolMatFree.c): Krylov methods
// that see the operator only as y = A x.  An operator installed by fasp_solver_matfree_init
// (MAT_CSR / MAT_BSR) is recognised by its function pointer: the matrix is uploaded once and the
// whole iteration stays in HBM.  Any other mf->fct is a host callback: its vectors are staged
// through host memory once per application (as for foreign preconditioners).
// ---------------------------------------------------------------------------
void fasp_hip_mxv_csr(const void* A, const double* x, double* y)  // SolMatFree.c: fasp_blas_mxv_csr
{
    FASP_ENTRY();
    fasp_blas_dcsr_mxv(static_cast<const dCSRmat*>(A), x, y);
}
void fasp_hip_mxv_bsr(const void* A, const double* x, double* y)  // SolMatFree.c: fasp_blas_mxv_bsr
{
    FASP_ENTRY();
    fasp_blas_dbsr_mxv(static_cast<const dBSRmat*>(A), x, y);
}

// SolMatFree.c:201
void fasp_solver_matfree_init(int matrix_format, mxv_matfree* mf, void* A)
{
    FASP_ENTRY();
    switch (matrix_format) {
        case MAT_CSR: mf->fct = fasp_hip_mxv_csr; break;
        case MAT_BSR: mf->fct = fasp_hip_mxv_bsr; break;
        default:  // the reference also knows STR / BLC / CSRL: formats this library does not have
            std::printf("### ERROR: Unknown matrix format %d!\n", matrix_format);
            std::exit(ERROR_DATA_STRUCTURE);
    }
    mf->data = A;
}

namespace {
// which: 0 CG, 1 VGMRES, 2 VFGMRES, 3 BiCGstab, 4 GMRES, 5 MinRes, 6 GCG
int krylov_matfree(const char* fn, int which, mxv_matfree* mf, dvector* b, dvector* u, precond* pc, double tol,
                   double abstol, int MaxIt, short restart, short StopType, short PrtLvl)
{
    if (ctx_init() < 0) die_no_device(fn);
    if (!mf || !mf->fct || !b || !u || b->row != u->row || b->row <= 0) return ERROR_INPUT_PAR;
    if (comm_size() > 1) return ERROR_INPUT_PAR;
    const int n = b->row;
    std::unique_ptr<TmpCSR> csr;
    std::unique_ptr<fasp_bsr::TmpBSR> bsr;
    KOps K;
    K.n = n; K.nvec = (size_t)n; K.fmt = "MatFree"; K.dist = false;
    K.halo = [](double*) { return 0; };
    TmpVec db(b->val, n), du(u->val, n), dz(nullptr, n), dy(nullptr, n);
    if (!db.d || !du.d || !dz.d || !dy.d) return ERROR_ALLOC_MEM;
    std::vector<double> hx, hy, hr, hz;
    if (mf->fct == fasp_hip_mxv_csr) {
        const dCSRmat* A = static_cast<const dCSRmat*>(mf->data);
        if (!A || A->row != n || A->col != n) return ERROR_INPUT_PAR;
        csr.reset(new TmpCSR(A));
        if (!csr->ok) return ERROR_ALLOC_MEM;
        const DevCSR* dA = &csr->D;
        K.mxv = [dA](const double* x, double* y) { d_mxv(*dA, x, y); };
        K.resid = [dA](const double* x, const double* bb, double* r) { d_resid(*dA, x, bb, r); };
    } else if (mf->fct == fasp_hip_mxv_bsr) {
        const dBSRmat* A = static_cast<const dBSRmat*>(mf->data);
        if (!A || A->ROW * A->nb != n || A->ROW != A->COL) return ERROR_INPUT_PAR;
        bsr.reset(new fasp_bsr::TmpBSR(A));
        if (!bsr->ok) return ERROR_ALLOC_MEM;
        const fasp_bsr::TmpBSR* Mp = bsr.get();
        K.mxv = [Mp](const double* x, double* y) { fasp_bsr::bsr_mxv(*Mp, x, y); };
        K.resid = [Mp](const double* x, const double* bb, double* r) { fasp_bsr::bsr_resid(*Mp, x, bb, r); };
    } else {
        hx.resize((size_t)n); hy.resize((size_t)n);
        auto host_mxv = [&, mf](const double* x, double* y) {
            (void)hipMemcpyAsync(hx.data(), x, sizeof(double) * n, hipMemcpyDeviceToHost, g_ctx.stream);
            (void)hipStreamSynchronize(g_ctx.stream);
            mf->fct(mf->data, hx.data(), hy.data());
            (void)hipMemcpyAsync(y, hy.data(), sizeof(double) * n, hipMemcpyHostToDevice, g_ctx.stream);
            (void)hipStreamSynchronize(g_ctx.stream);  // hy is reused by the next application
        };
        K.mxv = host_mxv;
        K.resid = [&, host_mxv](const double* x, const double* bb, double* r) {  // r = 1.0 b + (-1.0) A x
            host_mxv(x, dy.d);
            (void)hipMemcpyAsync(r, dy.d, sizeof(double) * n, hipMemcpyDeviceToDevice, g_ctx.stream);
            d_axpby(n, 1.0, bb, -1.0, r);
        };
    }
    fasp_hip_amg* h = (pc && pc->fct == fasp_hip_precond_fct) ? static_cast<fasp_hip_amg*>(pc->data) : nullptr;
    if (h && (h->L.empty() || h->L[0].A.row != n)) return ERROR_INPUT_PAR;
    if (h) {
        K.pc = [h](double* in, double** out) { return precond_amg(h, in, out); };
    } else if (pc && pc->fct) {
        hr.resize((size_t)n); hz.resize((size_t)n);
        K.pc = [&, pc](double* in, double** out) {
            HIPCK(hipMemcpyAsync(hr.data(), in, sizeof(double) * n, hipMemcpyDeviceToHost, g_ctx.stream));
            HIPCK(hipStreamSynchronize(g_ctx.stream));
            pc->fct(hr.data(), hz.data(), pc->data);
            HIPCK(hipMemcpyAsync(dz.d, hz.data(), sizeof(double) * n, hipMemcpyHostToDevice, g_ctx.stream));
            *out = dz.d;
            return 0;
        };
    }
    std::vector<double*> ws;
    size_t ws_len = 0;
    double* hh = nullptr;
    K.ws = &ws; K.ws_len = &ws_len; K.hh = &hh;
    K.stats = nullptr;
    Hist   H{nullptr, 0, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    int st;
    switch (which) {
        case 0: st = pcg_mf_device(K, db.d, du.d, tol, abstol, MaxIt, StopType, PrtLvl, &po); break;
        case 1: st = gmres_mf_device(K, true, false, db.d, du.d, tol, MaxIt, restart, StopType, PrtLvl, &po); break;
        case 2: st = gmres_mf_device(K, true, true, db.d, du.d, tol, MaxIt, restart, StopType, PrtLvl, &po); break;
        case 3: st = bicgstab_device(K, db.d, du.d, tol, MaxIt, PrtLvl, &H, &po); break;
        case 4: st = gmres_mf_device(K, false, false, db.d, du.d, tol, MaxIt, restart, StopType, PrtLvl, &po); break;
        case 5: st = minres_mf_device(K, db.d, du.d, tol, abstol, MaxIt, StopType, PrtLvl, &po); break;
        case 6: st = gcg_device(K, db.d, du.d, tol, abstol, MaxIt, StopType, PrtLvl, &H, &po); break;
        default: st = ERROR_SOLVER_TYPE;
    }
    du.get(u->val);
    for (double* q : ws) if (q) (void)hipFree(q);
    if (hh) (void)hipFree(hh);
    return st;
}
}  // namespace

// KryPcg.c:1260, KryPbcgs.c:1349, KryPgcg.c:213, KryPgmres.c:1309, KryPvgmres.c:1468, KryPvfgmres.c:1026
int fasp_solver_pcg(mxv_matfree* mf, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                    const int MaxIt, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_matfree(__func__, 0, mf, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
int fasp_solver_pbcgs(mxv_matfree* mf, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                      const int MaxIt, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_matfree(__func__, 3, mf, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
int fasp_solver_pgcg(mxv_matfree* mf, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                     const int MaxIt, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_matfree(__func__, 6, mf, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
int fasp_solver_pgmres(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                       const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_matfree(__func__, 4, mf, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
int fasp_solver_pvgmres(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                        const int MaxIt, short restart, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_matfree(__func__, 1, mf, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
int fasp_solver_pvfgmres(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                         const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_matfree(__func__, 2, mf, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
// KryPminres.c:1283: the reference's older MinRes text for mxv_matfree (minres_mf_device, krylov.hip.h)
int fasp_solver_pminres(mxv_matfree* mf, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                        const int MaxIt, const short StopType, const short PrtLvl)
{
    FASP_ENTRY();
    return krylov_matfree(__func__, 5, mf, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}

// SolMatFree.c:58: dispatch on itsolver_type
int fasp_solver_itsolver(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, ITS_param* itparam)
{
    FASP_ENTRY();
    if (!itparam) return ERROR_INPUT_PAR;
    const short prtlvl = itparam->print_level, stop_type = itparam->stop_type;
    const int restart = itparam->restart, MaxIt = itparam->maxit;
    const double tol = itparam->tol, abstol = itparam->abstol;
    const double t0 = wall_seconds();
    int iter = ERROR_SOLVER_TYPE;
    if (tol < SMALLREAL) std::printf("### WARNING: Convergence tolerance is too small! [%s:%d]\n", "ITS_CHECK", 74);
    if (MaxIt <= 0) std::printf("### WARNING: Max number of iterations must be POSITIVE! [%s:%d]\n", "ITS_CHECK", 78);
    switch (itparam->itsolver_type) {
        case SOLVER_CG: iter = fasp_solver_pcg(mf, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_BiCGstab: iter = fasp_solver_pbcgs(mf, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_MinRes: iter = fasp_solver_pminres(mf, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_GMRES: iter = fasp_solver_pgmres(mf, b, x, pc, tol, abstol, MaxIt, (short)restart, stop_type, prtlvl); break;
        case SOLVER_VGMRES: iter = fasp_solver_pvgmres(mf, b, x, pc, tol, abstol, MaxIt, (short)restart, stop_type, prtlvl); break;
        case SOLVER_VFGMRES: iter = fasp_solver_pvfgmres(mf, b, x, pc, tol, abstol, MaxIt, (short)restart, stop_type, prtlvl); break;
        case SOLVER_GCG: iter = fasp_solver_pgcg(mf, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        default:
            std::printf("### ERROR: Unknown iterative solver type %d! [%s]\n", itparam->itsolver_type, __func__);
            return ERROR_SOLVER_TYPE;
    }
    if ((prtlvl >= PRINT_SOME) && (iter >= 0)) std::printf("Iterative method costs %.4f seconds.\n", wall_seconds() - t0);
    return iter;
}

// SolMatFree.c:157: Krylov method without preconditioner
int fasp_solver_krylov(mxv_matfree* mf, dvector* b, dvector* x, ITS_param* itparam)
{
    FASP_ENTRY();
    if (!itparam) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    const int status = fasp_solver_itsolver(mf, b, x, nullptr, itparam);
    if (itparam->print_level >= PRINT_MIN) std::printf("Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    return status;
}

void fasp_blas_dcsr_mxv(const dCSRmat* A, const double* x, double* y)
{
    FASP_ENTRY();
    TmpCSR M(A);
    if (!M.ok) die_no_device(__func__);
    TmpVec dx(x, A->col), dy(nullptr, A->row);
    d_mxv(M.D, dx.d, dy.d);
    dy.get(y);
}

void fasp_blas_dcsr_aAxpy(const double alpha, const dCSRmat* A, const double* x, double* y)
{
    FASP_ENTRY();
    TmpCSR M(A);
    if (!M.ok) die_no_device(__func__);
    TmpVec dx(x, A->col), dy(y, A->row);
    d_aAxpy(alpha, M.D, dx.d, dy.d);
    dy.get(y);
}

double fasp_blas_darray_dotprod(const int n, const double* x, const double* y)
{
    FASP_ENTRY();
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n), dy(y, n);
    double out = 0.0;
    (void)d_dot(n, dx.d, dy.d, &out);
    return out;
}

double fasp_blas_darray_norm2(const int n, const double* x)
{
    FASP_ENTRY();
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n);
    double out[2] = {0, 0};
    (void)d_norms(n, dx.d, out);
    return std::sqrt(out[0]);
}

double fasp_blas_darray_norminf(const int n, const double* x)
{
    FASP_ENTRY();
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n);
    double out[2] = {0, 0};
    (void)d_norms(n, dx.d, out);
    return out[1];
}

void fasp_blas_darray_axpy(const int n, const double a, const double* x, double* y)
{
    FASP_ENTRY();
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n), dy(y, n);
    d_axpy(n, a, dx.d, dy.d);
    dy.get(y);
}

void fasp_blas_darray_axpby(const int n, const double a, const double* x, const double b, double* y)
{
    FASP_ENTRY();
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n), dy(y, n);
    d_axpby(n, a, dx.d, b, dy.d);
    dy.get(y);
}

// ---- the remaining names of SURVEY.md section 8 rows a10-a12 as stand-alone entries over the solve's own kernels ----
// alpha = y' A x (BlaSpmvCSR.c:839): the fused SpMV + dot epilogue the coarse-scaling step of the cycle uses
double fasp_blas_dcsr_vmv(const dCSRmat* A, const double* x, const double* y)
{
    FASP_ENTRY();
    TmpCSR M(A);
    if (!M.ok) die_no_device(__func__);
    TmpVec dx(x, A->col), dy(y, A->row), dt(nullptr, A->row);
    CsrArgs a{}; a.x = dx.d; a.y = dt.d; a.dotv = dy.d; a.partials = g_ctx.d_partials;
    const int G = launch_csr<OP_MXV_DOT>(M.D, a);
    d_finalize(G, 1, 0u, 0, false);
    double out = 0.0;
    (void)fetch_red(0, 1, &out);
    return out;
}
namespace {
// the matrix of the *_agg kernels: the caller's pattern with unit entries (the reference never reads val there, and
// 1.0 * x = x exactly, so the general kernels ARE the aggregation kernels: same sums, same order)
struct TmpUnitCSR {
    dCSRmat view;
    std::vector<double> ones;
    explicit TmpUnitCSR(const dCSRmat* A) : view(*A), ones((size_t)std::max(A->nnz, 1), 1.0) { view.val = ones.data(); }
};
}  // namespace
void fasp_blas_dcsr_mxv_agg(const dCSRmat* A, const double* x, double* y)  // BlaSpmvCSR.c:438
{
    FASP_ENTRY();
    TmpUnitCSR U(A);
    fasp_blas_dcsr_mxv(&U.view, x, y);
}
void fasp_blas_dcsr_aAxpy_agg(const double alpha, const dCSRmat* A, const double* x, double* y)  // BlaSpmvCSR.c:727
{
    FASP_ENTRY();
    TmpUnitCSR U(A);
    fasp_blas_dcsr_aAxpy(alpha, &U.view, x, y);
}
void fasp_blas_darray_ax(const int n, const double a, double* x)  // BlaArray.c:43
{
    FASP_ENTRY();
    if (a == 1.0) return;
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n);
    hipLaunchKernelGGL(k_scale, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, a, dx.d);
    dx.get(x);
}
void fasp_blas_darray_axpyz(const int n, const double a, const double* x, const double* y, double* z)  // BlaArray.c:403
{
    FASP_ENTRY();
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n), dy(y, n), dz(nullptr, n);
    hipLaunchKernelGGL(k_axpyz, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, a, dx.d, dy.d, dz.d);
    dz.get(z);
}
namespace {
int norm1_nan(const char* fn, int n, const double* x, double out[2])
{
    if (ctx_init() < 0) die_no_device(fn);
    TmpVec dx(x, n);
    const int G = vec_grid(n);
    hipLaunchKernelGGL(k_norm1_nan, dim3(G), dim3(BLOCK), 0, g_ctx.stream, n, dx.d, g_ctx.d_partials);
    d_finalize(G, 2, 0u, 0, false);
    return fetch_red(0, 2, out);
}
}  // namespace
double fasp_blas_darray_norm1(const int n, const double* x)  // BlaArray.c:663
{
    FASP_ENTRY();
    double out[2] = {0, 0};
    (void)norm1_nan(__func__, n, x, out);
    return out[0];
}
short fasp_dvec_isnan(const dvector* u)  // AuxVector.c:39
{
    FASP_ENTRY();
    double out[2] = {0, 0};
    (void)norm1_nan(__func__, u->row, u->val, out);
    return out[1] > 0.0 ? 1 : 0;
}
void fasp_darray_cp(const int n, const double* x, double* y)  // AuxArray.c:210: through HBM (upload, device copy, download)
{
    FASP_ENTRY();
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n), dy(nullptr, n);
    (void)hipMemcpyAsync(dy.d, dx.d, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, g_ctx.stream);
    dy.get(y);
}
void fasp_darray_set(const int n, double* x, const double val)  // AuxArray.c:41
{
    FASP_ENTRY();
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(nullptr, n);
    hipLaunchKernelGGL(k_set, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, val, dx.d);
    dx.get(x);
}

void fasp_blas_dbsr_mxv(const dBSRmat* A, const double* x, double* y)
{
    FASP_ENTRY();
    TmpBSR M(A);
    if (!M.ok) die_bsr(__func__);
    TmpVec dx(x, (size_t)A->COL * A->nb), dy(nullptr, (size_t)A->ROW * A->nb);
    BsrArgs a{}; a.x = dx.d; a.y = dy.d;
    launch_bsr<0>(M, a);
    dy.get(y);
}

void fasp_blas_dbsr_aAxpy(const double alpha, const dBSRmat* A, const double* x, double* y)
{
    FASP_ENTRY();
    if (alpha == 0.0) return;  // BlaSpmvBSR.c:548
    TmpBSR M(A);
    if (!M.ok) die_bsr(__func__);
    TmpVec dx(x, (size_t)A->COL * A->nb), dy(y, (size_t)A->ROW * A->nb);
    BsrArgs a{}; a.x = dx.d; a.y = dy.d; a.alpha = alpha;
    launch_bsr<1>(M, a);
    dy.get(y);
}

// BlaSparseBSR.c:543 (host): diagonal blocks inverted as fasp_smat_inv does (BlaSmallMatInv.c:603): closed forms for
// nb = 2, 3, 4, Gauss-Jordan with full pivoting for 5..7; nb == 1: reciprocals
dvector fasp_dbsr_getdiaginv(const dBSRmat* A)
{
    FASP_ENTRY();
    dvector out{0, nullptr};
    if (!A || A->nb < 1 || A->nb > 7) {
        std::fprintf(stderr, "### ERROR: fasp_dbsr_getdiaginv: block size %d not supported (1..7)\n", A ? A->nb : -1);
        return out;
    }
    out.row = A->ROW * A->nb * A->nb;
    out.val = (double*)std::calloc((size_t)std::max(out.row, 1), sizeof(double));
    (void)bsr_diaginv(A, out.val);
    return out;
}

void fasp_smoother_dbsr_jacobi1(dBSRmat* A, dvector* b, dvector* u, double* diaginv)
{
    FASP_ENTRY();
    TmpBSR M(A);
    if (!M.ok) die_bsr(__func__);
    const size_t n = (size_t)A->ROW * A->nb;
    TmpVec du(u->val, n), dun(nullptr, n), db(b->val, n), dd(diaginv, (size_t)A->ROW * A->nb * A->nb);
    BsrArgs a{}; a.x = du.d; a.y = dun.d; a.b = db.d; a.dinv = dd.d;
    launch_bsr<2>(M, a);
    dun.get(u->val);
}

double fasp_hip_time_bsr_mxv(const dBSRmat* A, int reps)
{
    FASP_ENTRY();
    TmpBSR M(A);
    if (!M.ok || reps <= 0) return -1.0;
    TmpVec dx(nullptr, (size_t)A->COL * A->nb), dy(nullptr, (size_t)A->ROW * A->nb);
    (void)hipMemsetAsync(dx.d, 0, sizeof(double) * dx.n, g_ctx.stream);
    BsrArgs a{}; a.x = dx.d; a.y = dy.d;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch_bsr<0>(M, a); launch_bsr<0>(M, a);
    (void)hipEventRecord(e0, g_ctx.stream);
    for (int i = 0; i < reps; ++i) launch_bsr<0>(M, a);
    (void)hipEventRecord(e1, g_ctx.stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (double)ms / reps;
}

// ItrSmootherCSR.c:98 -- rows i_1..i_n (either direction) of the square system
void fasp_smoother_dcsr_jacobi(dvector* u, const int i_1, const int i_n, const int s, dCSRmat* A, dvector* b,
                               int L, const double w)
{
    FASP_ENTRY();
    (void)s;
    TmpCSR M(A);
    if (!M.ok) die_no_device(__func__);
    const int n = A->row;
    const int lo = std::min(i_1, i_n), hi = std::max(i_1, i_n);
    if (lo != 0 || hi != n - 1) {
        std::fprintf(stderr, "### ERROR: fasp_smoother_dcsr_jacobi (device): only full sweeps 0..n-1 supported\n");
        std::exit(ERROR_INPUT_PAR);
    }
    std::vector<double> d(n, 0.0);
    bool dup = false;
    for (int i = 0; i < n; ++i) {
        int hits = 0;
        for (int k = A->IA[i]; k < A->IA[i + 1]; ++k)
            if (A->JA[k] == i) { d[i] = A->val[k]; ++hits; }
        dup = dup || hits > 1;
    }
    M.D.dup_diag = dup;   // (a diagonal stored twice: the kernels that take the diagonal's product out of a full row sum do not apply)
    TmpVec du(u->val, n), du2(nullptr, n), db(b->val, n), dd(d.data(), n);
    double *x = du.d, *xo = du2.d;
    while (L--) {
        CsrArgs a{};
        a.x = x; a.y = xo; a.b = db.d; a.diag = dd.d; a.omega = w;
        launch_csr<OP_JACOBI>(M.D, a);
        std::swap(x, xo);
    }
    (void)hipStreamSynchronize(g_ctx.stream);
    (void)hipMemcpy(u->val, x, sizeof(double) * n, hipMemcpyDeviceToHost);
}

#ifdef FLOW_TIMING
extern "C" int fasp_hip_flow_times(unsigned long long* out, int n)   // (development build only: tools/perf_gs_one.py)
{
    FASP_ENTRY();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fasp::g_flow_times), sizeof(unsigned long long) * (size_t)std::min(n, 8192)) == hipSuccess ? 0 : -1;
}
#endif

// development knob: override kernel selection / launch geometry at run time
static int g_time_cold = 0;   // fasp_hip_tune("time_cold"): fasp_hip_time_kernel times every launch on its own behind a flush of the Infinity Cache
int fasp_hip_tune(const char* key, int value)
{
    FASP_ENTRY();
    if (!key) return ERROR_INPUT_PAR;
    if (!std::strcmp(key, "maxgrid")) g_tune.maxgrid = value;
    else if (!std::strcmp(key, "xcd")) g_tune.xcd = value;
    else if (!std::strcmp(key, "gen2")) g_tune.gen2 = value;
    else if (!std::strcmp(key, "ws2_bpc")) g_tune.ws2_bpc = value;
    else if (!std::strcmp(key, "nt")) g_tune.nt = value;
    else if (!std::strcmp(key, "kind")) g_tune.kind = value;
    else if (!std::strcmp(key, "compress")) g_tune.compress = value;
    else if (!std::strcmp(key, "rpl")) g_tune.rpl = value;
    else if (!std::strcmp(key, "lds_tab")) g_tune.lds_tab = value;
    else if (!std::strcmp(key, "xcd_pat")) g_tune.xcd_pat = value;
    else if (!std::strcmp(key, "spcg_batch")) g_tune.spcg_batch = value;
    else if (!std::strcmp(key, "spcg_fused")) g_tune.spcg_fused = value;
    else if (!std::strcmp(key, "spcg_persist")) { g_tune.spcg_persist = value; if (value) g_persist_disabled = false; }
    else if (!std::strcmp(key, "spcg_test_hang")) g_tune.spcg_test_hang = value;
    else if (!std::strcmp(key, "spcg_grid")) g_tune.spcg_grid = value;
    else if (!std::strcmp(key, "small_lds")) g_tune.small_lds = value;
    else if (!std::strcmp(key, "small_onewave")) g_tune.small_onewave = value;
    else if (!std::strcmp(key, "ja16")) g_tune.ja16 = value;
    else if (!std::strcmp(key, "device_sort")) g_device_sort = value;
    else if (!std::strcmp(key, "halo_overlap")) g_halo_overlap = value;
    else if (!std::strcmp(key, "coarse_mode")) g_coarse_mode = value;            // replicated levels: 0 redundant work, 1 split rows + all-gather, -1 (default) split over peer windows only
    else if (!std::strcmp(key, "coarse_split_min")) g_coarse_split_min = value;
    else if (!std::strcmp(key, "split_rows")) g_tune.split_rows = value;
    else if (!std::strcmp(key, "gs_multicolor")) g_tune.gs_multicolor = value;
    else if (!std::strcmp(key, "seq_flow")) { g_tune.seq_flow = value; if (value) g_flow_disabled = false; }
    else if (!std::strcmp(key, "seq_strip_kb")) g_tune.seq_strip_kb = value;   // KB of lower entries per strip of the dataflow solve (0, default: 256 / 512 / 1024 by level shape, seq_sched.cpp)
    else if (!std::strcmp(key, "seq_jobs")) g_tune.seq_jobs = value;
    else if (!std::strcmp(key, "local_square")) g_tune.local_square = value;   // a rank's rows of a partitioned level are coded like a square operator (read at upload)
    else if (!std::strcmp(key, "seq_grid")) g_tune.seq_grid = value;     // workgroups of the dataflow solve at most (0: twice the strips the chain front is in at a time + 2, seq_sched.cpp; < 0: as many as are resident)
    else if (!std::strcmp(key, "seq_chain")) g_tune.seq_chain = value;   // chain form of the triangular solve (seq_chain.hip.h): 0 never, 1 on chain-bound sweeps, 2 wherever it applies; read when a schedule is built
    else if (!std::strcmp(key, "seq_chain_n1")) g_tune.seq_chain_n1 = value;       // blocks of 64 rows in tier 1 (0: two)
    else if (!std::strcmp(key, "seq_chain_grid")) g_tune.seq_chain_grid = value;   // tier-2 workgroups of the chain launch (0: default)
    else if (!std::strcmp(key, "seq_test_hang")) g_tune.seq_test_hang = value;     // tests: the next chain launches go without their tier-2 workgroups -- every waiter runs into its bounded poll (2 s)
    else if (!std::strcmp(key, "seq_rest_lanes")) g_tune.seq_rest_lanes = value;   // lanes per row of the rest pass (0: from the mean row length); read at launch
    else if (!std::strcmp(key, "seq_chain_ref")) g_tune.seq_chain_ref = value;     // 1: the plain one-wavefront form (k_tri_chain_ref: A/B, fallback)
    else if (!std::strcmp(key, "seq_spine")) g_tune.seq_spine = value;   // 0 never, 1 where the schedule chooses it, 2 wherever a row has two lanes (seq_sched.h); read when a schedule is built
    else if (!std::strcmp(key, "seq_partition")) g_seq_partition = value;
    else if (!std::strcmp(key, "fuse_zr")) g_tune.fuse_zr = value;
    else if (!std::strcmp(key, "fuse_presmooth")) g_tune.fuse_presmooth = value;
    else if (!std::strcmp(key, "seq_lanes")) g_tune.seq_lanes = value;
    else if (!std::strcmp(key, "lazy_coarse")) g_tune.lazy_coarse = value;
    else if (!std::strcmp(key, "xtile")) g_tune.xtile = value;
    else if (!std::strcmp(key, "rp5_max")) g_tune.rp5_max = value;
    else if (!std::strcmp(key, "rp_bpc")) g_tune.rp_bpc = value;
    else if (!std::strcmp(key, "rp_stream")) g_tune.rp_stream = value;
    else if (!std::strcmp(key, "renumber")) g_tune.renumber = value;             // brick renumbering of the uncoded mid levels at upload (order-independent smoothers only; hierarchy.hip.h): 1 on (default), 2 also levels behind a coded one (whose transfer operators keep their coding and take a numbering bridge), 0 off; read when a hierarchy is uploaded
    else if (!std::strcmp(key, "renumber_chunk")) g_tune.renumber_chunk = value; // rows per chunk inside which the balls grow (reorder.cpp)
    else if (!std::strcmp(key, "time_cold")) g_time_cold = value;
    else if (!std::strcmp(key, "estream")) g_tune.estream = value;   // long-row operators: the entry-parallel stream kernel where it measured faster (1, default: mean rows below 256 entries), wherever its tables exist (2), never (0: the row kernel); read at launch
    else if (!std::strcmp(key, "es_dbg")) g_tune.es_dbg = value;     // (FASP_LAB_DEBUG builds: parts of the stream kernel switched off -- timings only)
    else if (!std::strcmp(key, "pcg_dev_beta")) g_tune.pcg_dev_beta = value;   // top-level PCG: (z, r), beta and alpha stay on the device, one host wait per iteration (1, default) or two (0)
    else if (!std::strcmp(key, "seq_chain_touch")) g_tune.seq_chain_touch = value;   // chain form: blocks by which a workgroup of its own on the chain's XCD touches the band planes ahead (8; 0: the importer wave does, four ahead)
    else if (!std::strcmp(key, "seq_chain_touch_t1")) g_tune.seq_chain_touch_t1 = value;   // ... and tier 1's entries of those blocks (1, default)
    else if (!std::strcmp(key, "seq_zero_skip")) g_tune.seq_zero_skip = value;   // sequential sweeps: the parallel pass of a sweep that starts from the zero vector reads b only (1, default) or every entry (0)
    else if (!std::strcmp(key, "pcg_fold")) g_tune.pcg_fold = value;           // one rank: (t,p) and (z,r) are summed from their partials by the kernels that divide by them (1, default) or by a k_finalize launch each (0)
    else if (!std::strcmp(key, "spcg_spec")) g_tune.spcg_spec = value;         // persistent coarse CG: the true residual of Check III queued behind the kernel, one host wait per coarse solve (1, default) or two (0)
    else if (!std::strcmp(key, "ev_every")) g_tune.ev_every = value;           // the level-0 t = A p launch inside a solve is bracketed by an event pair every n-th iteration (4; 1: every one)
    else if (!std::strcmp(key, "rp_xcd")) g_tune.rp_xcd = value;
    else if (!std::strcmp(key, "rp_strip")) g_tune.rp_strip = value;   // coded operators of a 3-D grid: an XCD sweeps a strip of every plane (1: the square ones, 2: the transfer operators too, default) or a slab of planes (0)
    else if (!std::strcmp(key, "host_parallel_min")) g_parallel_min_nnz = value;
    else if (!std::strcmp(key, "lanes")) g_tune.lanes = value;
    else if (!std::strcmp(key, "wrows")) g_tune.wrows = value;
    else if (!std::strcmp(key, "wcap")) g_tune.wcap = value;
    else return ERROR_INPUT_PAR;
    return FASP_SUCCESS;
}

// CPU test entry: the tables of k_csr_estream for a matrix with these row pointers, walked on the host (device_csr.hip.h)
int fasp_hip_estream_selftest(const int* ia, int nrow, int nnz, int per_wave, int wmax, int* info)
{
    if (!ia || nrow < 1 || nnz < 1 || per_wave < 8 || wmax < 32) return ERROR_INPUT_PAR;
    return estream_selftest_host(ia, nrow, nnz, per_wave, wmax, info);
}

// Measured device ceilings beside the roofline: a 16-byte-per-lane read, copy and triad over buffers of
// `bytes` each (use >= 512 MiB: beyond the 256 MiB Infinity Cache).  out[0..2] = GB/s of read, copy (read + write
// counted), triad (two reads + one write counted).  Returns 0 or a negative error code.
int fasp_hip_measure_ceilings(double* out, size_t bytes, int reps)
{
    FASP_ENTRY();
    if (!out || bytes < (1u << 20) || reps <= 0) return ERROR_INPUT_PAR;
    if (ctx_init() != FASP_SUCCESS) return ERROR_MISC;
    f64x2_t *p = nullptr, *q = nullptr, *r = nullptr;
    HIPCK(hipMalloc(&p, bytes)); HIPCK(hipMalloc(&q, bytes)); HIPCK(hipMalloc(&r, bytes));
    HIPCK(hipMemsetAsync(p, 0, bytes, g_ctx.stream)); HIPCK(hipMemsetAsync(q, 0, bytes, g_ctx.stream));
    HIPCK(hipMemsetAsync(r, 0, bytes, g_ctx.stream));
    const size_t n16 = bytes / 16;
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    const int grid = 4 * g_ctx.num_cu;
    for (int which = 0; which < 3; ++which) {
        auto run = [&]() {
            if (which == 0) hipLaunchKernelGGL(k_read16, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, n16, p, (double*)q);
            else if (which == 1) hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, n16, p, q);
            else hipLaunchKernelGGL(k_triad16, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, n16, 0.5, p, q, r);
        };
        run(); run();
        HIPCK(hipEventRecord(e0, g_ctx.stream));
        for (int i = 0; i < reps; ++i) run();
        HIPCK(hipEventRecord(e1, g_ctx.stream));
        HIPCK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCK(hipEventElapsedTime(&ms, e0, e1));
        out[which] = (double)(which + 1) * (double)bytes * reps / ((double)ms * 1e-3) / 1e9;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(p); (void)hipFree(q); (void)hipFree(r);
    return FASP_SUCCESS;
}

// timed micro-benchmark of one kernel class on a resident level
double fasp_hip_time_kernel(fasp_hip_amg* h, int kind, int level, int reps)
{
    FASP_ENTRY();
    if (!h || level < 0 || level >= (int)h->L.size() || reps <= 0) return -1.0;
    DevLevel& D = h->L[level];
    const int n = D.A.row;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
    double* x = D.xa; double* y = D.xb; double* w = D.w;
    auto run = [&]() {
        switch (kind) {
            case 0: d_mxv(D.A, x, y); break;
            case 1: d_aAxpy(-1.0, D.A, x, y); break;
            case 2: { CsrArgs a{}; a.x = x; a.y = y; a.b = w; a.diag = D.diag; a.omega = 0.6667; launch_csr<OP_JACOBI>(D.A, a); } break;
            case 3: hipLaunchKernelGGL(k_dot, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, x, y, g_ctx.d_partials); break;
            case 4: d_axpy(n, 0.5, x, y); break;
            case 5: { CsrArgs a{}; a.x = x; a.y = y; a.dotv = x; a.partials = g_ctx.d_partials; launch_csr<OP_MXV_DOT>(D.A, a); } break;
            case 6: if (D.R.ia) d_mxv(D.R, w, h->L[level + 1].xa); break;
            case 7: if (D.P.ia) d_aAxpy(1.0, D.P, h->L[level + 1].xa, y); break;
            case 8:   // the plain stream bound of a level: its values (and as many bytes again for every 4 of index data) read 16 bytes per lane
                if (D.A.val) hipLaunchKernelGGL(k_read16, dim3(2048), dim3(BLOCK), 0, g_ctx.stream, (size_t)D.A.nnz / 2, (const f64x2_t*)D.A.val, y);
                break;
            case 10: case 11: case 12: case 13: (void)seq_sweep(h, level, kind - 10, 1, 1.0); break;   // GS sweep: ascending, descending, C rows, F rows
            default: break;
        }
    };
    (void)hipMemsetAsync(x, 0, sizeof(double) * n, g_ctx.stream);
    (void)hipMemsetAsync(y, 0, sizeof(double) * n, g_ctx.stream);
    (void)hipMemsetAsync(w, 0, sizeof(double) * n, g_ctx.stream);
    double* const keep_b = D.b; double* const keep_x = D.x;
    if (kind >= 10) { D.b = w; D.x = x; }   // the sweeps work on the level's own vectors: scratch ones while timing
    if (g_time_cold) {
        // as a solve meets the kernel: one event pair per launch, 512 MB of other data READ in between (1) -- nothing of the operator or the
        // vectors waits in the Infinity Cache; back to back, a level of up to 256 MB is served from there -- or WRITTEN in between (2): the
        // launch also pays for the write-back of what its predecessors left there
        double* big = nullptr;
        const size_t bigb = (size_t)512 << 20;
        if (hipMalloc(&big, bigb + 4096) != hipSuccess) return -1.0;
        (void)hipMemsetAsync(big, 0, bigb, g_ctx.stream);
        double tot = 0.0;
        for (int i = 0; i < reps + 1; ++i) {
            if (g_time_cold >= 2) (void)hipMemsetAsync(big, 0, bigb, g_ctx.stream);
            else hipLaunchKernelGGL(k_read16, dim3(1024), dim3(BLOCK), 0, g_ctx.stream, bigb / 16, (const f64x2_t*)big, big + bigb / 8);
            (void)hipEventRecord(e0, g_ctx.stream);
            run();
            (void)hipEventRecord(e1, g_ctx.stream);
            (void)hipEventSynchronize(e1);
            float ms1 = 0.f;
            (void)hipEventElapsedTime(&ms1, e0, e1);
            if (i >= 1) tot += ms1;
        }
        (void)hipFree(big);
        D.b = keep_b; D.x = keep_x;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        return tot / reps;
    }
    run(); run();
    (void)hipEventRecord(e0, g_ctx.stream);
    for (int i = 0; i < reps; ++i) run();
    (void)hipEventRecord(e1, g_ctx.stream);
    (void)hipEventSynchronize(e1);
    D.b = keep_b; D.x = keep_x;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (double)ms / reps;
}


// Development / test entry: one operator through the resident upload path (lossless coding, kernel selection as for a
// hierarchy level) timed with HIP events on the launch stream.  op: 0 y = A x, 1 y -= A x, 2 Jacobi sweep, 5 y = A x
// fused with (y, x).  Returns milliseconds per launch (< 0: error); *kind_out = kernel family as fasp_hip_amg_kernel_info.
double fasp_hip_time_matrix(const dCSRmat* A, int op, int reps, int* kind_out)
{
    FASP_ENTRY();
    if (!A || reps <= 0 || ctx_init() < 0) return -1.0;
    HostCSR M;
    M.row = A->row; M.col = A->col; M.nnz = A->nnz;
    M.row_aligned = A->col > A->row && std::getenv("FASP_HIP_TIME_ROW_ALIGNED") != nullptr;   // (tools/perf_local_op.py: a rank's rows of a partitioned level)
    M.ia.view(A->IA, (size_t)A->row + 1); M.ja.view(A->JA, (size_t)std::max(A->nnz, 1)); M.val.view(A->val, (size_t)std::max(A->nnz, 1));
    DevCSR D;
    {
        HostThreads team;
        if (upload_csr(M, D) < 0) { D.release(); return -1.0; }
    }
    const int n = A->row, nc = A->col;
    double *x = nullptr, *y = nullptr, *w = nullptr, *dg = nullptr;
    if (hipMalloc(&x, 8 * (size_t)std::max(nc, n)) != hipSuccess || hipMalloc(&y, 8 * (size_t)n) != hipSuccess ||
        hipMalloc(&w, 8 * (size_t)n) != hipSuccess || hipMalloc(&dg, 8 * (size_t)n) != hipSuccess) { D.release(); return -1.0; }
    (void)hipMemsetAsync(x, 0, 8 * (size_t)std::max(nc, n), g_ctx.stream);
    (void)hipMemsetAsync(y, 0, 8 * (size_t)n, g_ctx.stream);
    (void)hipMemsetAsync(w, 0, 8 * (size_t)n, g_ctx.stream);
    hipLaunchKernelGGL(k_set, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, 1.0, dg);
    auto run = [&]() {
        switch (op) {
            case 0: d_mxv(D, x, y); break;
            case 1: d_aAxpy(-1.0, D, x, y); break;
            case 2: { CsrArgs a{}; a.x = x; a.y = y; a.b = w; a.diag = dg; a.omega = 0.6667; launch_csr<OP_JACOBI>(D, a); } break;
            default: { CsrArgs a{}; a.x = x; a.y = y; a.dotv = x; a.partials = g_ctx.d_partials; launch_csr<OP_MXV_DOT>(D, a); } break;
        }
    };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    run(); run();
    (void)hipEventRecord(e0, g_ctx.stream);
    for (int i = 0; i < reps; ++i) run();
    (void)hipEventRecord(e1, g_ctx.stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (kind_out) {
        int k = D.kind;
        if (D.code && g_tune.compress) k = 4;
        if (D.pat && g_tune.compress) k = (D.nxrows >= 0 && g_tune.gen2) ? (D.rowbase ? 9 : 6) : 5;
        *kind_out = k;
    }
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(w); (void)hipFree(dg);
    D.release();
    return (double)ms / reps;
}

}  // extern "C"
