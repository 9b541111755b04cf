// sgym_hip.hip -- host side of libsgym_hip.so: the C ABI declared in include/sgym.h.
// Owns the device buffers and one HIP stream per handle; builds the BatchReplayEntity union knot
// grids on the host (sort/unique per scenario, threaded) and everything else on the device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <iterator>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#define SG_UNIT_MAIN // the setup / sensor / fix-up kernels live in this object
#include "sgym_launch.hpp"

using sg::Params;

// Device allocations of one upload, kept across uploads: a pipeline that feeds batch after batch of the same shape through
// a handle must not pay hipFree / hipMalloc each time (both wait for the WHOLE device, i.e. for the rollout another handle
// is running: the double-buffered upload of tools/upload_time.py would serialise with it).  sg_upload requests its buffers
// in a fixed order; request i reuses slot i when it is large enough.
struct ReusePool {
    std::vector<void *> ptr;
    std::vector<size_t> cap;
    size_t cursor = 0;
    void rewind() { cursor = 0; }
};

struct sg_handle {
    sg_config cfg{};
    int R = 0, E = 0, EP = 0, G = 0, WV = 1;
    bool has_ped = false;
    bool planar = false;      // every knot of the batch has z = pitch = roll = +0.0 (bit patterns): rollout_kernel_tab_planar
    bool sliceable = false;   // every entity is a replay entity / replay agent / PID or vehicle agent (or padding): the batch
                              // can be time-sliced (launch_sliced)
    int slice_mode = 1;       // sg_set_tuning / env SG_SLICE: 0 never, 1 automatic (small batches, long rollouts)
    std::vector<void *> slice_allocs; // device arrays of launch_sliced, kept between calls of the same shape
    int slice_T = -1, slice_S = 0;
    sg::SliceArgs slice_args{};
    int *d_n_final = nullptr, *d_slice_done = nullptr;
    double *d_slice_tab = nullptr; // the controller table of a whole sliced call (batches with PID / vehicle agents)
    std::vector<double> clock_t0;  // distinct scenario start times (ScenarioGym.get_start_time): one clock each
    std::vector<int> clock_of;     // [R]
    double *d_clock_t0 = nullptr;
    bool all_ped = false;     // every entity of the batch is a pedestrian agent of catalog type Pedestrian (or padding)
    bool crowd_riders = false; // a crowd (64-lane tiles) whose other lanes are replay entities / replay agents / PID / vehicle agents:
                               // rollout_kernel_crowd_riders + control_kernel_riders (env SG_CROWD_RIDERS=0: the general variant)
    int crowd_kernel = 1;     // env SG_CROWD_KERNEL=0: all-pedestrian batches take the general pedestrian variant too
    bool crowd_models = true, models_all_sf = true; // env SG_CROWD_MODELS; every model of sg_set_ped_models is a SocialForce
    sg_social_force sf{};
    int ped_behaviour = 0;        // sg_set_ped_behaviour
    int n_ped_models = 0;         // sg_set_ped_models: > 1 = the batch mixes behaviour models / parameter sets
    double *d_ped_models = nullptr;  // [n_ped_models][PM_W]
    int32_t *d_model_of = nullptr;   // [NE]
    int noise_mode = 0;           // sg_set_ped_noise
    double noise_std[2] = {0.0, 0.0};
    double *d_normals = nullptr;  // [R][noise_len]
    long long noise_len = 0;
    unsigned long long noise_seed = 0;
    double *d_gon = nullptr;
    size_t NE = 0; // padded entity count
    bool uploaded = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing_now = true;       // this call records its timing events (calls of >= 16 steps, and every table-path call)
    bool timed = false;
    Params p{};
    ReusePool static_allocs, state_allocs;
    int32_t *d_row_scen = nullptr;
    std::vector<hipEvent_t> up_ev;  // sg_upload: one event per piece of the knot copy
    std::vector<double> up_grid_t;  // host buffers of sg_upload, kept between calls
    double *up_stat = nullptr;      // (page-locked)
    size_t up_stat_cap = 0;
    std::vector<int32_t> up_row_scen;
    std::vector<std::vector<double>> up_grids;
    int64_t total_rows = 0;
    double *d_actions = nullptr;
    size_t actions_cap = 0;
    // controller pre-pass (sg::control_kernel): controlled lanes of the batch, its own stream, two table buffers
    int n_ctl = 0;
    int n_ext = 0;            // SG_KIND_AGENT_EXTERNAL slots in the batch
    double *d_ext = nullptr;  // [NE][6]
    int max_ctl_per_block = 0; // controlled lanes in the fullest 64-slot block
    hipStream_t ctl_stream = nullptr;
    bool wide = false;                           // more than 512 entities per scenario: the multi-kernel step (sgym_wide.hpp)
    std::vector<void *> wide_allocs;
    sg::WideArgs wide_args{};
    int *wide_running = nullptr;    // page-locked ring of "scenarios still running" answers (launch_wide)
    unsigned wide_check = 0;        // check points enqueued so far
    // page-locked staging of sg_read_metrics (the per-scenario state and the event table travel every time metrics are read:
    // 0.5 + up to 6 MB for 4096 scenarios; pageable copies ran at a third of the PCIe rate)
    void *pin_sd = nullptr, *pin_ev = nullptr;
    size_t pin_sd_cap = 0, pin_ev_cap = 0;
    // the table path as one persistent launch (sgym_queue.hpp, launch_queue)
    unsigned *d_qwords = nullptr;   // queue state + progress words: [Q_STATE_WORDS + n_ctl_waves + nblk + Q_MAX_CHUNKS]
    size_t qwords_cap = 0;
    double *d_qtab = nullptr;       // the table ring
    size_t qtab_bytes = 0;
    // page-locked copies of the queue state words (give-up code, items done) of the persistent launches that have not been looked
    // at yet (check_queue): a ring with one slot per launch, so that a second launch before the check cannot overwrite the
    // first one's code.  A give-up is STICKY: every later call that runs or reads the batch fails with its message until
    // sg_reset / sg_upload start the batch anew (the state is undefined in between).
    char last_kernel[96] = {0};     // sg_last_kernel
    int q_waves_per_cu[3] = {-1, -1, -1}; // occupancy of rollout_kernel_tabq / _planar / _rss_tabq (slots_of), queried once
    unsigned *q_host = nullptr;
    int q_head = 0, q_count = 0;    // next slot to use; launches not yet looked at
    bool q_failed = false;
    char q_msg[320] = {0};
    int queue_mode = 1;             // env SG_QUEUE=0: the chunk launches of rounds 1-4 instead
    int last_schedule = 0;          // 0: not the table path, 1: chunk launches, 2: the persistent queue launch
    int last_chunks = 0, last_ring = 0, last_grid = 0;
    double *d_tab[4] = {nullptr, nullptr, nullptr, nullptr}; // controller-table buffers (launch_rollout: two, four with block groups)
    int n_tab = 0;
    int n_simd = 1024;                                       // SIMDs of the device (4 per compute unit)
    size_t tab_bytes = 0;     // bytes of each table buffer
    std::vector<hipEvent_t> ev_pool;
    int tab_min = 16, chunk_steps = 1024, overlap = 1; // sg_set_tuning
    // sg_tick: the kernels of one RL tick captured once as a hipGraph and replayed with a single launch
    hipGraphExec_t tick_exec = nullptr;
    uint64_t generation = 0, tick_gen = ~0ull;         // bumped by every call that changes what the kernels are launched with
    double tick_w = 0, tick_h = 0;
    int tick_nw = 0, tick_nh = 0, tick_nl = 0;
    bool tick_rss = false;                             // the captured step runs the RSS callback (rollout_kernel_rss* + rss_lines_kernel)
    int32_t tick_layers[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t *d_rss_state = nullptr, *d_rss_code = nullptr; // [NE] sg_rss_update
    int32_t *d_rss_seen = nullptr;                         // [R]
    bool rss_fused = false;                                // this launch runs the callback inside rollout_kernel_rss
    bool rss_enabled = false;                              // sg_set_rss: RSSDistances runs after every step of sg_rollout / sg_step
    bool ego_first = true;                                 // every scenario's ego is its entity 0
    double *d_rss_safe = nullptr;                          // [NE][2]
    double *d_rssq = nullptr;                              // line-test queues of rollout_kernel_rss: [NE / 64][(rssq_steps + 1) * 64][12]
    int32_t *d_rssq_n = nullptr;                           // [NE / 64]
    int rssq_steps = 0;                                    // steps per launch the queues are sized for
    size_t rss_NE = 0, rssq_NE = 0;                        // padded entity counts the records / the queues were allocated for
    bool rss_stale = true;                                 // the records belong to a batch that is gone (sg_upload): cleared on next use
    double c_tol = 0.4;                                // CollisionMetric(c_tol): angular half-width of a box corner, metrics/collision.py:57
    unsigned char *d_reset_mask = nullptr;             // [R] sg_reset_scenarios
    uint32_t *d_term_flags = nullptr;                  // [R] sg_terminal_flags
    void *obs_buf = nullptr;                           // device scratch of the observation calls (grown on demand)
    size_t obs_cap = 0;
    std::vector<void *> road_allocs;                   // sg_set_road_networks
    sg::RoadIndex road{};                              // host copy of the device pointers (raster kernels take it by value)
    bool has_road = false;
    int ped_serial = 0;                                // env SG_PED_SERIAL: pedestrian pair loop one pedestrian per lane
    int ctl_slice = 64;                                // steps per control_kernel launch (env SG_CTL_SLICE)
    int n_launches = 0;           // rollout_kernel launches of the last call
    std::vector<int> launch_ev;   // their (start, stop) event indices into ev_pool
    std::string err;
};

// the RSSDistances records hold results of the current batch (sg_upload leaves the buffers, not their contents)
static bool rss_live(const sg_handle *h) { return h->d_rss_state && !h->rss_stale; }

// the crowd variants (rollout_kernel_crowd / _riders / _models) hold the social force model alone
// (several models: a pass of the force code per model, as the general variant does it -- up to four, SocialForce all of them)
static bool crowd_allowed(const sg_handle *h)
{
    return h->crowd_kernel && h->ped_behaviour == SG_PED_SOCIAL_FORCE && (h->n_ped_models <= 1 || (h->n_ped_models <= 4 && h->models_all_sf && h->crowd_models));
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
// ... and a road network only as the source of the boundary forces (social_force.py:86-104: a phase behind the neighbour sums,
// gated on the launch having polygons at all); ego_off_road needs the cell lookup of rollout_kernel_road / the general variants
static bool crowd_road_ok(const sg_handle *h) { return !h->has_road || (!(h->cfg.terminal_mask & SG_TERM_EGO_OFF_ROAD) && env_int("SG_CROWD_ROADS", 1) != 0); }

// Host threads for sg_upload's pass over the scenarios: the logical CPUs, capped at 64 and by the cgroup CPU quota (a box can
// show 256 CPUs under a quota of 16: more busy threads than that are throttled, not run -- ADVICE r2); SG_UPLOAD_THREADS overrides.
static unsigned host_threads()
{
    static const unsigned n = [] {
        unsigned v = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota> <period>" or "max <period>"
            char q[32] = {0};
            long period = 0;
            if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const long quota = atol(q);
                if (quota > 0) v = std::min<unsigned>(v, (unsigned)std::max(1L, (quota + period - 1) / period));
            }
            fclose(f);
        }
        const int forced = env_int("SG_UPLOAD_THREADS", 0);
        return forced > 0 ? (unsigned)forced : v;
    }();
    return n;
}

static thread_local std::string g_create_err;

static int fail(sg_handle *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(h, SG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                        __FILE__, __LINE__);                                                      \
    } while (0)

// SG_POISON=1 (tests): every device array that is handed out WITHOUT being zeroed is filled with 0xA5 bytes instead of being
// left as the allocator found it -- a read of something no kernel wrote shows in a fresh process as it would after a
// thousand other handles (tests/test_gpu_parity.py::test_poisoned_allocations_change_nothing).  Synchronous, so that a fill can never land after a copy another stream makes into the same array.
static int poison_byte() // 0: off; else the byte (SG_POISON=165: 0xA5, 255: NaNs / -1, 127: huge ints, NaN-free doubles)
{
    static const int v = env_int("SG_POISON", 0) & 0xff;
    return v;
}
static bool poison_allocs() { return poison_byte() != 0; }
static void poison(hipStream_t s, void *ptr, size_t bytes) // (a fresh allocation no kernel has written yet)
{
    if (poison_allocs() && ptr) {
        (void)hipMemsetAsync(ptr, poison_byte(), bytes, s);
        (void)hipStreamSynchronize(s);
    }
}

template <typename T>
static int dev_alloc(sg_handle *h, std::vector<void *> &pool, T **out, size_t n, bool zero = true)
{
    void *ptr = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIP_TRY(h, hipMalloc(&ptr, bytes));
    pool.push_back(ptr);
    if (zero) HIP_TRY(h, hipMemsetAsync(ptr, 0, bytes, h->stream));
    else if (poison_allocs()) { // (synchronous: the array may be filled on another stream next, e.g. the knot copy of sg_upload)
        HIP_TRY(h, hipMemsetAsync(ptr, poison_byte(), bytes, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    *out = (T *)ptr;
    return SG_OK;
}

static int ensure_rss(sg_handle *h, bool *fresh);
static int ensure_rssq(sg_handle *h);

template <typename T>
static int dev_upload(sg_handle *h, std::vector<void *> &pool, const T **out, const std::vector<T> &v)
{
    T *d = nullptr;
    int rc = dev_alloc(h, pool, &d, v.size(), false);
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(h, hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    *out = d;
    return SG_OK;
}

template <typename T>
static int dev_alloc(sg_handle *h, ReusePool &pool, T **out, size_t n, bool zero = true)
{
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    const size_t i = pool.cursor++;
    if (i == pool.ptr.size()) { pool.ptr.push_back(nullptr); pool.cap.push_back(0); }
    if (pool.cap[i] < bytes) {
        if (pool.ptr[i]) HIP_TRY(h, hipFree(pool.ptr[i]));
        pool.ptr[i] = nullptr;
        pool.cap[i] = 0;
        HIP_TRY(h, hipMalloc(&pool.ptr[i], bytes));
        pool.cap[i] = bytes;
    }
    if (zero) HIP_TRY(h, hipMemsetAsync(pool.ptr[i], 0, bytes, h->stream));
    else if (poison_allocs()) {
        HIP_TRY(h, hipMemsetAsync(pool.ptr[i], poison_byte(), bytes, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    *out = (T *)pool.ptr[i];
    return SG_OK;
}

template <typename T>
static int dev_upload(sg_handle *h, ReusePool &pool, const T **out, const std::vector<T> &v)
{
    T *d = nullptr;
    int rc = dev_alloc(h, pool, &d, v.size(), false);
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(h, hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    *out = d;
    return SG_OK;
}

static void free_pool(ReusePool &pool)
{
    for (void *ptr : pool.ptr)
        if (ptr) (void)hipFree(ptr);
    pool.ptr.clear();
    pool.cap.clear();
    pool.cursor = 0;
}

static void free_pool(std::vector<void *> &pool)
{
    for (void *ptr : pool) (void)hipFree(ptr);
    pool.clear();
}

extern "C" int sg_version(void) { return SG_ABI_VERSION; }

extern "C" const char *sg_last_error(const sg_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int sg_create(const sg_config *cfg, sg_handle **out)
{
    if (!cfg || !out) return fail(nullptr, SG_ERR_INVALID, "sg_create: null argument");
    *out = nullptr;
    if (cfg->n_scenarios <= 0 || cfg->n_entities <= 0)
        return fail(nullptr, SG_ERR_INVALID, "sg_create: n_scenarios and n_entities must be positive");
    if (cfg->n_entities > 16384)
        return fail(nullptr, SG_ERR_INVALID, "sg_create: n_entities=%d > 16384 (the event record keeps the other entity in 32 bits, "
                    "the state blocks SG_F_COLL + n_entities / 64 rows: nothing stops at 512 any more, this is a sanity bound)",
                    cfg->n_entities);
    if (!(cfg->timestep > 0.0)) return fail(nullptr, SG_ERR_INVALID, "sg_create: timestep must be > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, SG_ERR_NO_DEVICE, "sg_create: no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, SG_ERR_INVALID, "sg_create: device %d out of range (%d devices)", cfg->device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail(nullptr, SG_ERR_HIP, "sg_create: hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, SG_ERR_NO_DEVICE, "sg_create: device %d is %s; this library is built for gfx950 only",
                    cfg->device, prop.gcnArchName);
    sg_handle *h = new sg_handle();
    h->cfg = *cfg;
    h->R = cfg->n_scenarios;
    h->E = cfg->n_entities;
    int G = 4;
    while (G < h->E && G < 64) G <<= 1;
    h->G = G;
    // wavefronts per scenario.  8 (257..512 entities): the eight-wavefront instances of the general variants (plain, pedestrian,
    // RSS, road); the table path, the crowd kernels and the riders' pre-pass stop at 256.
    // More than 512: no fused kernel -- the step runs as four kernels over as many workgroups as the scenario needs (sgym_wide.hpp)
    h->WV = h->E <= 64 ? 1 : (h->E <= 128 ? 2 : (h->E <= 256 ? 4 : (h->E <= 512 ? 8 : (h->E + 63) / 64)));
    h->wide = h->WV > 8;
    if (h->wide && h->R > 65535) { // (the scenario is the y coordinate of the wide kernels' grids)
        delete h;
        return fail(nullptr, SG_ERR_INVALID, "sg_create: more than 65535 scenarios of more than 512 entities in one handle (n_scenarios=%d)", cfg->n_scenarios);
    }
    h->EP = G * h->WV;
    // SocialForceParameters defaults, pedestrian/social_force.py:16-30 (noise off)
    h->sf = sg_social_force{1.5, 1.0, 1.0, 0.0, 0.5, 1.0, std::cos(200.0 / 2 * M_PI / 180), 1.3, 0.0, 0.0, 2.0, 0.1};
    h->NE = (((size_t)h->R * h->EP + 63) / 64) * 64;
    h->tab_min = env_int("SG_TAB_MIN_STEPS", h->tab_min);
    h->chunk_steps = env_int("SG_CHUNK_STEPS", h->chunk_steps);
    h->overlap = env_int("SG_OVERLAP", h->overlap);
    h->ctl_slice = std::max(1, env_int("SG_CTL_SLICE", h->ctl_slice));
    h->ped_serial = env_int("SG_PED_SERIAL", 0) != 0;
    h->crowd_kernel = env_int("SG_CROWD_KERNEL", 1);
    h->crowd_models = env_int("SG_CROWD_MODELS", 1) != 0; // (0: batches with several pedestrian models keep to the general variant; the tests compare)
    h->slice_mode = env_int("SG_SLICE", 1);
    h->queue_mode = env_int("SG_QUEUE", 1);
    // the controller stream carries the serial chain of the table path (control_kernel_fast: 64 wavefronts that every rollout
    // launch waits for): highest stream priority, so that its launches are dispatched ahead of the rollout kernels'
    // (measured: no difference at 4096 x 64, where the launches never queue; SG_CTL_PRIO=0 creates it at the lowest)
    int prio_lo = 0, prio_hi = 0;
    if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess ||
        hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&h->ctl_stream, hipStreamNonBlocking * 0, env_int("SG_CTL_PRIO", 1) ? prio_hi : prio_lo) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        delete h;
        return fail(nullptr, SG_ERR_HIP, "sg_create: stream/event creation failed");
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) h->n_simd = 4 * cus;
        else (void)hipGetLastError();
    }
    *out = h;
    return SG_OK;
}

extern "C" int sg_destroy(sg_handle *h)
{
    if (!h) return SG_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
#ifdef SG_RSS_STATS
    {
        unsigned long long c[8];
        (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(sg::sg_rss_stats), sizeof c);
        fprintf(stderr, "rss stats: flushes %llu groups %llu items %llu passes %llu wave-updates %llu lat-lanes %llu long-lanes %llu\n", c[0], c[1], c[2], c[3], c[4], c[5], c[6]);
    }
#endif
#ifdef SG_PHASE_TIMERS
    if (h->p.phase_cycles) {
        unsigned long long c[16];
        (void)hipMemcpy(c, h->p.phase_cycles, sizeof c, hipMemcpyDeviceToHost);
        unsigned long long tot = 0;
        for (int i = 0; i < 16; ++i) tot += c[i];
        fprintf(stderr, "phase cycles (s_memtime, summed over wavefronts):");
        for (int i = 0; i < 16; ++i) fprintf(stderr, " [%d] %.1f%%", i, tot ? 100.0 * c[i] / tot : 0.0);
        fprintf(stderr, "  total %.3e\n", (double)tot);
        std::vector<unsigned long long> hw(4096);
        (void)hipMemcpy(hw.data(), h->p.phase_cycles + 16, 4096 * 8, hipMemcpyDeviceToHost);
        for (int b = 0; b < 12; ++b) {
            fprintf(stderr, "block %d:", b);
            for (int w = 0; w < 4; ++w) { unsigned v = (unsigned)hw[b * 4 + w]; fprintf(stderr, " [wave %u simd %u cu %u se %u xcc?%x]", v & 15, (v >> 4) & 3, (v >> 8) & 15, (v >> 13) & 7, v >> 16); }
            fprintf(stderr, "\n");
        }
    }
#endif
    free_pool(h->static_allocs);
    free_pool(h->state_allocs);
    free_pool(h->road_allocs);
    free_pool(h->slice_allocs);
    free_pool(h->wide_allocs);
    if (h->pin_sd) (void)hipHostFree(h->pin_sd);
    if (h->pin_ev) (void)hipHostFree(h->pin_ev);
    if (h->obs_buf) (void)hipFree(h->obs_buf);
    if (h->d_reset_mask) (void)hipFree(h->d_reset_mask);
    if (h->d_term_flags) (void)hipFree(h->d_term_flags);
    if (h->d_rss_state) (void)hipFree(h->d_rss_state);
    if (h->d_rss_seen) (void)hipFree(h->d_rss_seen);
    if (h->d_rss_code) (void)hipFree(h->d_rss_code);
    if (h->d_rss_safe) (void)hipFree(h->d_rss_safe);
    if (h->d_rssq) (void)hipFree(h->d_rssq);
    if (h->d_rssq_n) (void)hipFree(h->d_rssq_n);
    if (h->tick_exec) (void)hipGraphExecDestroy(h->tick_exec);
    if (h->ctl_stream) (void)hipStreamSynchronize(h->ctl_stream);
    if (h->d_actions) (void)hipFree(h->d_actions);
    if (h->d_gon) (void)hipFree(h->d_gon);
    if (h->d_normals) (void)hipFree(h->d_normals);
    for (int b = 0; b < 4; ++b)
        if (h->d_tab[b]) (void)hipFree(h->d_tab[b]);
    if (h->d_ped_models) (void)hipFree(h->d_ped_models);
    if (h->d_model_of) (void)hipFree(h->d_model_of);
    if (h->d_qwords) (void)hipFree(h->d_qwords);
    if (h->d_qtab) (void)hipFree(h->d_qtab);
    if (h->q_host) (void)hipHostFree(h->q_host);
    if (h->wide_running) (void)hipHostFree(h->wide_running);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->up_ev) (void)hipEventDestroy(e);
    if (h->up_stat) (void)hipHostFree(h->up_stat);
    if (h->ctl_stream) (void)hipStreamDestroy(h->ctl_stream);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return SG_OK;
}

// reference defaults: VehicleController.__init__ controller.py:64-70, PIDController.__init__ :154-161
// + PedestrianAgent / PedestrianController defaults, pedestrian/agent.py:18-27
static const double kDefaultCtrl[SG_NCTRL] = {0.7, 5.0, NAN, 0.0, 0.03054, 1.5709, 0.3753, 1.8970, 0.0204,
                                              0.0, 5.0, 0.0, 1.0, 0, 0, 0};

// one group holding every block: an ordinary launch of a table variant
static sg::TabGroups one_group(const sg_handle *h, const double *tab, int n_steps)
{
    sg::TabGroups tg{};
    tg.active = 1;
    tg.gsz = (int)std::max<size_t>(1, h->NE / 64);
    tg.n[0] = n_steps;
    tg.buf[0] = tab;
    tg.start0 = 0; tg.len0 = (unsigned)tg.gsz; tg.start1 = 0; tg.len1 = 0;
    return tg;
}

// grid of a launch of the one-wavefront-per-tile table kernels: the blocks of the active groups
static dim3 tab_grid(const sg_handle *h, const sg::TabGroups &tg)
{
    return dim3((unsigned)std::min<size_t>(h->NE / 64, (size_t)tg.len0 + tg.len1));
}

// the entry point the handle launched last for a step loop (sg_last_kernel): what a kernel trace of the call shows
static void note_kernel(sg_handle *h, const char *fmt, int a = 0, int b = 0) { snprintf(h->last_kernel, sizeof h->last_kernel, fmt, a, b); }
// the rollout kernel family of this handle's batch (launchers: sgym_launch.hpp, one object per family)
static void launch_variant(sg_handle *h, dim3 grid, int n_steps, int do_reset, int force, const double *d_actions,
                           const double *d_tab, bool use_tab, const sg::TabGroups &tg)
{
    const int G = h->G, WV = h->WV;
    const hipStream_t s = h->stream;
    const sgl::RolloutArgs a{&h->p, h->cfg.timestep, n_steps, do_reset, force, d_actions, nullptr};
    const sgl::RolloutArgs at{&h->p, h->cfg.timestep, n_steps, 0, force, nullptr, d_tab}; // table variants never reset
    if (WV == 8 && !h->has_ped && (h->cfg.terminal_mask & SG_TERM_EGO_OFF_ROAD)) { // 257..512 entities, ego_off_road
        // (RSSDistances + ego_off_road in one wide rollout: no fused variant at that width -- launch_rollout_impl and sg_tick clear
        // rss_fused for it and run the callback as a launch of its own behind every step: unfused_rss)
        note_kernel(h, "sg::rollout_kernel_road<64, 8>");
        sgl::rollout_road(64, 8, grid, s, a);
        return;
    }
    if (WV == 8 && h->rss_fused && !h->has_ped) // 257..512 entities: the RSS callback inside the kernel, eight wavefronts
        note_kernel(h, "sg::rollout_kernel_rss<64, 8>"), sgl::rollout_rss(64, 8, false, grid, s, a);
    else if (WV == 8 && h->has_ped) // 257..512 entities with pedestrian agents: the general pedestrian variant on eight wavefronts
        note_kernel(h, "sg::rollout_kernel<64, 8, true, false>"), sgl::rollout_ped(64, 8, false, grid, s, a);
    else if (WV == 8) // ... vehicles and replay only (launch_rollout never takes the table path at this width)
        note_kernel(h, "sg::rollout_kernel<64, 8, false, false>"), sgl::rollout_plain(64, 8, false, grid, s, a);
    else if (h->has_ped && h->all_ped && G == 64 && crowd_road_ok(h) && crowd_allowed(h) && !h->rss_fused)
        note_kernel(h, h->n_ped_models > 1 ? "sg::rollout_kernel_crowd_models<%d>" : "sg::rollout_kernel_crowd<%d>", WV),
            sgl::rollout_crowd(WV, false, grid, s, a, h->n_ped_models > 1);
    else if (use_tab && h->has_ped && G == 64) // (launch_rollout: a crowd with riders, their table is d_tab)
        note_kernel(h, "sg::rollout_kernel_crowd_riders<%d>", WV), sgl::rollout_crowd(WV, true, grid, s, at);
    else if (h->has_ped && h->rss_fused)
        note_kernel(h, "sg::rollout_kernel_rss_ped<%d, %d>", std::max(G, 16), WV), sgl::rollout_ped(G, WV, true, grid, s, a);
    else if (h->rss_fused && (h->cfg.terminal_mask & SG_TERM_EGO_OFF_ROAD))
        note_kernel(h, "sg::rollout_kernel_rss_road<%d, %d>", G, WV), sgl::rollout_rss(G, WV, true, grid, s, a);
    else if (h->has_ped)
        note_kernel(h, "sg::rollout_kernel<%d, %d, true, false>", std::max(G, 16), WV), sgl::rollout_ped(G, WV, false, grid, s, a);
    else if (h->rss_fused && use_tab && WV == 1) // (launch_rollout: the controlled lanes' poses come from the pre-pass table)
        note_kernel(h, "sg::rollout_kernel_rss_tab<%d>", G), sgl::rollout_rss_tab(G, tab_grid(h, tg), s, h->p, h->cfg.timestep, force, tg);
    else if (h->rss_fused)
        note_kernel(h, "sg::rollout_kernel_rss<%d, %d>", G, WV), sgl::rollout_rss(G, WV, false, grid, s, a);
    else if (h->cfg.terminal_mask & SG_TERM_EGO_OFF_ROAD)
        note_kernel(h, "sg::rollout_kernel_road<%d, %d>", G, WV), sgl::rollout_road(G, WV, grid, s, a);
    else if (use_tab && WV == 1 && h->n_ctl > 0)
        note_kernel(h, h->planar ? "sg::rollout_kernel_tab_planar<%d>" : "sg::rollout_kernel_tab<%d>", G), sgl::rollout_tab(G, h->planar, tab_grid(h, tg), s, h->p, h->cfg.timestep, force, tg);
    else if (use_tab)
        note_kernel(h, "sg::rollout_kernel<%d, %d, false, true>", G, WV), sgl::rollout_plain(G, WV, true, grid, s, at);
    else
        note_kernel(h, "sg::rollout_kernel<%d, %d, false, false>", G, WV), sgl::rollout_plain(G, WV, false, grid, s, a);
}

static int get_event(sg_handle *h, size_t idx, hipEvent_t *out)
{
    while (h->ev_pool.size() <= idx) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreate(&e));
        h->ev_pool.push_back(e);
    }
    *out = h->ev_pool[idx];
    return SG_OK;
}

// one rollout_kernel launch on the handle's stream, bracketed by its own pair of timing events
// (groups: the block groups of a grouped table-variant launch, launch_rollout; else every block runs n_steps on d_tab)
static int launch_main(sg_handle *h, int n_steps, int do_reset, int force, const double *d_actions, const double *d_tab,
                       bool use_tab, size_t *ev_next, const sg::TabGroups *groups = nullptr)
{
    const sg::TabGroups tg = groups ? *groups : one_group(h, d_tab, n_steps);
    dim3 grid(h->WV == 1 ? (unsigned)(h->NE / 64) : (unsigned)h->R);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc;
    if (h->timing_now) {
        if ((rc = get_event(h, *ev_next, &e0)) || (rc = get_event(h, *ev_next + 1, &e1))) return rc;
        HIP_TRY(h, hipEventRecord(e0, h->stream));
    }
    launch_variant(h, grid, n_steps, do_reset, force, d_actions, d_tab, use_tab, tg);
    HIP_TRY(h, hipGetLastError());
    if (use_tab && d_tab && h->n_ctl > 0 && h->p.ev_cap > 0 && n_steps > 0) {
        // a controlled ego's pose at an event of this chunk is a row of the chunk's controller table: copied into the event
        // now -- before the event below, which the pre-pass of a later chunk waits for before it reuses the buffer
        sg::event_ego_pose_kernel<<<dim3((unsigned)h->R), dim3(64), 0, h->stream>>>(h->p, tg);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->rss_fused) { // (launch_variant ran a rollout_kernel_rss* variant)
        sgl::rss_lines(tab_grid(h, tg), h->stream, h->p, tg);
        HIP_TRY(h, hipGetLastError());
    }
    if (!h->timing_now) return SG_OK;
    HIP_TRY(h, hipEventRecord(e1, h->stream));
    if (n_steps > 0) { // reset-only launches are not counted as hot-path launches
        h->launch_ev.push_back((int)*ev_next);
        ++h->n_launches;
    }
    *ev_next += 2;
    return SG_OK;
}

// ScenarioGym.rollout / n x step for the whole batch.  Scenarios without pedestrians and with at least
// SG_TAB_MIN_STEPS steps to do take the two-kernel path: control_kernel integrates the PID / vehicle agents
// for a chunk of steps on its own stream while rollout_kernel<TAB> consumes the previous chunks' tables -- large batches as
// two or three pipelines on streams of their own (below).
// Scenarios of more than 512 entities (sgym_wide.hpp): State.reset / n x ScenarioGym.step as four kernels per step.  Every
// scenario that may run steps in lockstep (a done scenario sits the step out unless `force`).  The host never waits here: every
// 64 steps a one-workgroup kernel writes the number of running scenarios into page-locked memory, and rollout() stops
// enqueuing once an EARLIER check point has answered 0 (what it enqueued in the meantime are no-ops).
constexpr unsigned WIDE_RING = 1024;
static int ensure_wide(sg_handle *h) // (the scratch of the multi-kernel step; sg_tick calls it before it starts capturing)
{
    int rc = SG_OK;
    if (!h->wide_args.scr) {
        auto &A = h->wide_allocs;
        if ((rc = dev_alloc(h, A, &h->wide_args.scr, h->NE * sg::WS_W)) || (rc = dev_alloc(h, A, &h->wide_args.cor, h->NE * 8)) ||
            (rc = dev_alloc(h, A, &h->wide_args.circ, h->NE * 4)) || (rc = dev_alloc(h, A, &h->wide_args.last_row, (size_t)h->R * h->WV)) ||
            (rc = dev_alloc(h, A, &h->wide_args.last_same, h->NE)) || (rc = dev_alloc(h, A, &h->wide_args.dup, (size_t)h->R)) ||
            (rc = dev_alloc(h, A, &h->wide_args.walkers, (size_t)h->R)))
            return rc;
    }
    if (!h->wide_running) HIP_TRY(h, hipHostMalloc((void **)&h->wide_running, WIDE_RING * sizeof(int), hipHostMallocDefault));
    return SG_OK;
}

static int launch_wide(sg_handle *h, int n_steps, int do_reset, int force, const double *d_actions)
{
    const int R = h->R, EP = h->EP;
    int rc = SG_OK;
    if ((rc = ensure_wide(h))) return rc;
    const dim3 ge((unsigned)((EP + 255) / 256), (unsigned)R), gs((unsigned)R);
    auto one = [&](int mode, const double *acts) {
        sg::WideArgs wa = h->wide_args;
        wa.mode = mode;
        wa.force = force;
        wa.actions = acts;
        wa.no_peds = h->has_ped ? 0 : 1;
        note_kernel(h, "sg::wide_move_kernel + wide_commit_kernel + wide_collide_kernel + wide_finish_kernel");
        sgl::wide_step(ge, gs, h->stream, h->p, h->cfg.timestep, wa, !h->has_ped);
    };
    // sg_set_rss at this width: RSSDistances.__call__ as a launch of its own after the reset and after every step
    auto rss = [&](int reset) {
        if (h->rss_fused)
            sg::rss_kernel<<<dim3((unsigned)R), dim3(512), 0, h->stream>>>(h->p, reset, h->d_rss_state, h->d_rss_code, h->d_rss_safe, h->d_rss_seen);
    };
    if (do_reset) {
        one(do_reset == 2 ? 2 : 1, nullptr);
        rss(do_reset == 2 ? 2 : 1);
        HIP_TRY(h, hipGetLastError());
    }
    const unsigned first_check = h->wide_check;
    for (int k = 0; k < n_steps; ++k) {
        one(0, d_actions ? d_actions + (size_t)k * R * 2 : nullptr);
        rss(0);
        if (!force && (k & 63) == 63 && k + 1 < n_steps) { // is anybody still running?
            HIP_TRY(h, hipGetLastError());
            bool nobody = false;
            for (unsigned c = first_check; c != h->wide_check && !nobody; ++c)
                nobody = __atomic_load_n(&h->wide_running[c % WIDE_RING], __ATOMIC_ACQUIRE) == 0;
            if (nobody) break;
            if (h->wide_check - first_check < WIDE_RING) { // (a call of more than 65,536 steps stops asking)
                int *word = &h->wide_running[h->wide_check++ % WIDE_RING];
                __atomic_store_n(word, -1, __ATOMIC_RELEASE);
                sgl::wide_running(h->stream, h->p, word);
            }
        }
    }
    HIP_TRY(h, hipGetLastError());
    return SG_OK;
}

enum { Q_HOST_RING = 16 };
// after a synchronisation of h->stream: did every persistent launch since the last look run to its end?  (Sticky: see q_failed.)
static int check_queue(sg_handle *h)
{
    for (; h->q_count > 0; --h->q_count) {
        const unsigned *w = h->q_host + (size_t)((h->q_head - h->q_count + Q_HOST_RING) % Q_HOST_RING) * sg::Q_STATE_WORDS;
        const unsigned code = w[sg::Q_ERR];
        if (code == 0 || h->q_failed) continue; // (the first give-up is the one reported)
        static const char *what[] = {"", "a rollout wavefront waited for the controller pre-pass", "a rollout wavefront waited for the previous chunk of its block",
                                     "the controller pre-pass waited for a buffer of the table ring"};
        h->q_failed = true;
        h->queue_mode = 0; // (whatever kept its wavefronts from meeting will do so again: the handle's later calls take the chunk launches)
        snprintf(h->q_msg, sizeof h->q_msg, "sg_rollout: the persistent table launch gave up (%s longer than SG_QUEUE_TIMEOUT_MS; %u work items had finished): "
                                            "the state of the batch is undefined -- sg_reset / sg_upload before the next call", what[code < 4 ? code : 0],
                 w[sg::Q_ITEMS_DONE]);
    }
    return h->q_failed ? fail(h, SG_ERR_HIP, "%s", h->q_msg) : SG_OK;
}
// sg_reset / sg_upload start the batch anew: whatever a launch before them gave up on is history
static void forget_queue_failure(sg_handle *h)
{
    h->q_count = 0;
    h->q_failed = false;
}

// The table path as ONE launch (sgym_queue.hpp): the controller pre-pass and the rollout of every chunk of the time axis in one
// grid of persistent wavefronts, work items (chunk, block) from a device-side counter.  `chunk` = the longest chunk.
// Returns SG_OK, an error, or SG_QUEUE_FALLBACK: the table ring could not be allocated -- the caller takes the chunk launches.
#define SG_QUEUE_FALLBACK 1
// Wavefront slots of the persistent kernel on this device: what the build's launch bounds say (wavefronts per SIMD x SIMDs), cut
// down to what the runtime's occupancy query grants when that is less (a CU mask, a partitioned device, a build whose
// registers or LDS grew) -- every wavefront of the grid has to be resident at once, the pre-pass roles never yield.
static size_t slots_of(sg_handle *h, bool rss)
{
    const size_t by_bounds = (size_t)h->n_simd * (size_t)(rss ? SG_WAVES_PER_SIMD : (h->planar ? SG_PLANAR_WAVES : SG_TAB_WAVES));
    int &cached = h->q_waves_per_cu[rss ? 2 : (h->planar ? 1 : 0)];
    if (cached < 0) cached = rss ? sgl::rss_tabq_waves_per_cu(h->G) : sgl::tabq_waves_per_cu(h->G, h->planar);
    return cached > 0 ? std::min(by_bounds, (size_t)cached * (size_t)(h->n_simd / 4)) : by_bounds;
}
static int launch_queue(sg_handle *h, int n_steps, int force, const double *d_actions, int chunk, size_t *ev_next, bool rss = false)
{
    const size_t nblk = h->NE / 64, np = (size_t)h->p.n_ctl_pad, n_ctl_waves = np / 64;
    // chunks of the time axis: short at first (the first rollout items cannot start before the pre-pass has written their chunk),
    // growing by ~1.4x -- the pre-pass is only 1.3 ... 1.9x faster per step than a rollout wavefront beside it, so chunk c + 1
    // has to be written in about the time chunk c takes to roll out: with lengths that doubled, a quarter of the wavefronts
    // waited through the first 4 ms (tools/dbg/queue_timeline.py) -- up to a plateau of SG_QUEUE_CAP = 512 steps (blocks are
    // at most one chunk apart, and the call ends when the LAST block does: 1024-step chunks measured 4 % slower, 256-step
    // ones 2 %, profiles/r05_ab_chunk_cap.txt), and shrinking again the same way at the end (SG_QUEUE_DECAY, percent) so that
    // the last items are short.  SG_QUEUE_GROW: the growth in percent.
    std::vector<int> len;
    chunk = std::min(chunk, std::max(1, env_int("SG_QUEUE_CAP", 512))); // (never above what the caller allows: the RSS line-test queue holds `chunk` steps)
    // (first chunk 96 / 128 / 160 / 192 / 256 steps: 94.3 / 95.9 / 97.4 / 97.2 / 95.4 G, means of four interleaved runs on one box,
    // profiles/r05_ab_first_chunk.txt -- the length also sets the last chunk's, the ramps mirror each other)
    const int first = std::max(1, std::min(chunk, env_int("SG_QUEUE_FIRST", 160)));
    const int grow = std::max(101, env_int("SG_QUEUE_GROW", 140));
    const int decay = env_int("SG_QUEUE_DECAY", 140); // 0: only the last chunk is halved (below)
    std::vector<int> up, down; // first, first * g, ... (< chunk); the mirror image at the end of the call
    for (long long n = first; n < chunk && n < chunk * 3ll / 4; n = std::max(n + 1, n * grow / 100)) up.push_back((int)n);
    if (decay > 100)
        for (long long n = first; n < chunk && n < chunk * 3ll / 4; n = std::max(n + 1, n * decay / 100)) down.insert(down.begin(), (int)n);
    long long ramp = 0;
    for (int n : up) ramp += n;
    for (int n : down) ramp += n;
    if (ramp + chunk <= n_steps) {
        // ramp up, a plateau of whole chunks, ramp down; what is left over (< chunk) goes where the ramp down reaches its length
        const long long mid = n_steps - ramp;
        const int whole = (int)(mid / chunk), rest = (int)(mid % chunk);
        len = up;
        len.insert(len.end(), (size_t)whole, chunk);
        if (rest > 0) down.insert(std::lower_bound(down.begin(), down.end(), rest, std::greater<int>()), rest);
        len.insert(len.end(), down.begin(), down.end());
    } else {
        for (int k0 = 0, n = 0; k0 < n_steps; k0 += n) {
            const long long want = len.empty() ? first : std::max<long long>((long long)len.back() + 1, (long long)len.back() * grow / 100);
            n = (int)std::min<long long>(std::min<long long>(chunk, want), n_steps - k0);
            if (want >= chunk / 2 + chunk / 4 && want < chunk) n = std::min(chunk, n_steps - k0); // (no odd chunk just below the cap)
            len.push_back(n);
        }
        while (len.size() > 1 && len.back() >= 2 * first && len.back() > 128) { // ..., L -> ..., L - L / 2, L / 2, repeated on the tail
            const int L = len.back(), half = L / 2;
            len.back() = L - half;
            len.push_back(half);
        }
    }
    const int C = (int)len.size();
    if (C > sg::Q_MAX_CHUNKS) return SG_QUEUE_FALLBACK;
    // the table ring: as many chunk buffers as the call has chunks when they fit a quarter of the free memory (no buffer is
    // ever reused: the pre-pass never waits), else a ring of at least three
    const size_t row = (size_t)sg::CT_PLANES * sg::CT_W * np;
    // (rows per lane = the longest chunk of THIS launch + 1, whatever table an earlier chunk-launch call of the handle needed:
    // the stride is a launch parameter -- ADVICE r5: a handle that once ran thousands of steps per chunk kept ring buffers of
    // that size for ever)
    const int ts = chunk;
    const size_t buf_bytes = (size_t)(ts + 1) * row * sizeof(double);
    int n_buf = C;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)8 << 30; }
        const size_t budget = std::min<size_t>((free_b + h->qtab_bytes) / 4, (size_t)std::max(1, env_int("SG_QUEUE_TAB_MB", 32768)) << 20);
        n_buf = (int)std::min<size_t>((size_t)C, std::max<size_t>(3, budget / buf_bytes));
        if (const int forced = env_int("SG_QUEUE_RING", 0)) n_buf = std::min(C, std::max(2, forced)); // (tests: a ring that is reused)
    }
    const size_t need = (size_t)n_buf * buf_bytes;
    if (need > h->qtab_bytes) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->d_qtab) HIP_TRY(h, hipFree(h->d_qtab));
        h->d_qtab = nullptr;
        h->qtab_bytes = 0;
        if (hipMalloc((void **)&h->d_qtab, need) != hipSuccess) {
            (void)hipGetLastError();
            h->d_qtab = nullptr;
            return SG_QUEUE_FALLBACK;
        }
        h->qtab_bytes = need;
        poison(h->stream, h->d_qtab, need);
    }
    h->p.tab_steps = ts;
    const bool q_trace = env_int("SG_QUEUE_DEBUG", 0) != 0;
    const size_t words = (size_t)sg::Q_STATE_WORDS + n_ctl_waves + nblk + (size_t)sg::Q_MAX_CHUNKS + 1 + (size_t)sg::Q_SEATS + (q_trace ? nblk : 0);
    if (words > h->qwords_cap) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->d_qwords) HIP_TRY(h, hipFree(h->d_qwords));
        h->d_qwords = nullptr;
        h->qwords_cap = 0;
        HIP_TRY(h, hipMalloc((void **)&h->d_qwords, words * sizeof(unsigned)));
        h->qwords_cap = words;
    }
    if (!h->q_host) HIP_TRY(h, hipHostMalloc((void **)&h->q_host, (size_t)Q_HOST_RING * sg::Q_STATE_WORDS * sizeof(unsigned), hipHostMallocDefault));
    if (h->q_count == Q_HOST_RING) { // every slot holds a launch nobody has looked at: look now
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        const int rcq = check_queue(h);
        if (rcq) return rcq;
    }
    HIP_TRY(h, hipMemsetAsync(h->d_qwords, 0, words * sizeof(unsigned), h->stream)); // every polled word, before every launch
    sg::TabQueue tq{};
    tq.state = h->d_qwords;
    tq.ctl_prog = tq.state + sg::Q_STATE_WORDS;
    tq.blk_prog = tq.ctl_prog + n_ctl_waves;
    tq.chunk_cnt = tq.blk_prog + nblk;
    tq.seats = tq.chunk_cnt + sg::Q_MAX_CHUNKS + 1;
    tq.trace = q_trace ? tq.seats + sg::Q_SEATS : nullptr;
    tq.defer_ticks = (long long)std::max(1, env_int("SG_QUEUE_DEFER_US", 30)) * 100ll; // 100 MHz
    tq.tab = h->d_qtab;
    tq.buf_doubles = buf_bytes / sizeof(double);
    tq.actions = d_actions;
    tq.timeout_ticks = (long long)std::max(1, env_int("SG_QUEUE_TIMEOUT_MS", 20000)) * 100000ll; // 100 MHz
    if (const int us = env_int("SG_QUEUE_TIMEOUT_US", 0)) tq.timeout_ticks = (long long)std::max(1, us) * 100ll; // (tests: a give-up on demand)
    tq.handoff = env_int("SG_QUEUE_HANDOFF", 1) != 0; // 0: a release fence per item instead (correct as well, 60 G on c3)
    tq.lag_prio = env_int("SG_QUEUE_LAGPRIO", 2);
    const char *times_path = getenv("SG_QUEUE_TIMES"); // experiment: per-item time stamps, dumped as u64 after the launch
    static unsigned long long *d_times = nullptr;
    static size_t times_cap = 0;
    const size_t n_times = (size_t)C * nblk * 4 + n_ctl_waves * (size_t)C;
    if (times_path && *times_path) {
        if (n_times > times_cap) {
            if (d_times) HIP_TRY(h, hipFree(d_times));
            HIP_TRY(h, hipMalloc((void **)&d_times, n_times * 8));
            times_cap = n_times;
        }
        HIP_TRY(h, hipMemsetAsync(d_times, 0, n_times * 8, h->stream));
        tq.times = d_times;
    }
    tq.n_chunks = C;
    tq.n_buf = n_buf;
    tq.nblk = (int)nblk;
    tq.n_ctl_waves = (int)n_ctl_waves;
    tq.k0[0] = 0;
    for (int c = 0; c < C; ++c) tq.k0[c + 1] = tq.k0[c] + len[(size_t)c];
    // as many wavefronts as the device holds at once (three per SIMD), no more than there is work for
    const size_t slots = slots_of(h, rss);
    const unsigned grid = (unsigned)std::min(slots, n_ctl_waves + nblk);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc;
    if ((rc = get_event(h, *ev_next, &e0)) || (rc = get_event(h, *ev_next + 1, &e1))) return rc;
    static hipStream_t ws = nullptr; // SG_QUEUE_DEBUG: the watcher's stream and page-locked words, made before the launch
    static unsigned *w = nullptr;
    if (q_trace && !ws) {
        HIP_TRY(h, hipStreamCreateWithFlags(&ws, hipStreamNonBlocking));
        HIP_TRY(h, hipHostMalloc((void **)&w, (sg::Q_STATE_WORDS + 16) * sizeof(unsigned), hipHostMallocDefault));
        HIP_TRY(h, hipMemcpyAsync(w, h->d_qwords, 64, hipMemcpyDeviceToHost, ws));
        HIP_TRY(h, hipStreamSynchronize(ws));
    }
    HIP_TRY(h, hipEventRecord(e0, h->stream));
    note_kernel(h, rss ? "sg::rollout_kernel_rss_tabq<%d>" : (h->planar ? "sg::rollout_kernel_tabq_planar<%d>" : "sg::rollout_kernel_tabq<%d>"), h->G);
    if (rss) sgl::rollout_rss_tabq(h->G, dim3(grid), h->stream, h->p, h->cfg.timestep, force, tq);
    else sgl::rollout_tabq(h->G, h->planar, dim3(grid), h->stream, h->p, h->cfg.timestep, force, tq);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(e1, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->q_host + (size_t)h->q_head * sg::Q_STATE_WORDS, h->d_qwords, sg::Q_STATE_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    h->q_head = (h->q_head + 1) % Q_HOST_RING;
    ++h->q_count;
    if (tq.times) {
        std::vector<unsigned long long> ht(n_times + 4);
        HIP_TRY(h, hipMemcpyAsync(ht.data() + 4, d_times, n_times * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        ht[0] = (unsigned long long)C; ht[1] = nblk; ht[2] = n_ctl_waves; ht[3] = grid;
        if (FILE *f = fopen(times_path, "wb")) {
            fwrite(ht.data(), 8, ht.size(), f);
            fwrite(tq.k0, sizeof(int), (size_t)C + 1, f);
            fclose(f);
        }
    }
    if (const int dbg = env_int("SG_QUEUE_DEBUG", 0)) { // watch the queue words from the host while the launch runs (dbg x 100 ms)
        fprintf(stderr, "queue: grid %u, %d chunks, ring %d, %zu blocks, %zu pre-pass wavefronts, first chunk %d steps\n", grid, C, n_buf, nblk, n_ctl_waves, len[0]);
        const int us = std::max(50, env_int("SG_QUEUE_DEBUG_US", 100000));
        const auto t_start = std::chrono::steady_clock::now();
        for (int i = 0; i < dbg; ++i) {
            std::this_thread::sleep_for(std::chrono::microseconds(us));
            if (hipMemcpyAsync(w, h->d_qwords, (sg::Q_STATE_WORDS + 4) * sizeof(unsigned), hipMemcpyDeviceToHost, ws) != hipSuccess || hipStreamSynchronize(ws) != hipSuccess) break;
            fprintf(stderr, "queue +%.2f ms: tickets %u head %u err %u items done %u | ctl_prog %u %u %u %u",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), w[0], w[1], w[2], w[3], w[8], w[9], w[10], w[11]);
            unsigned *tr = w + sg::Q_STATE_WORDS + 8;
            if (hipMemcpyAsync(tr, tq.trace, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, ws) == hipSuccess && hipStreamSynchronize(ws) == hipSuccess)
                fprintf(stderr, " | trace (chunk << 8 | stage) of blocks 0..7: %x %x %x %x %x %x %x %x", tr[0], tr[1], tr[2], tr[3], tr[4], tr[5], tr[6], tr[7]);
            fprintf(stderr, "\n");
            if (hipStreamQuery(h->stream) == hipSuccess) break;
        }
    }
    h->launch_ev.push_back((int)*ev_next);
    ++h->n_launches;
    *ev_next += 2;
    h->last_schedule = 2;
    h->last_chunks = C;
    h->last_ring = n_buf;
    h->last_grid = (int)grid;
    return SG_OK;
}


static int launch_rollout_impl(sg_handle *h, int n_steps, int do_reset, int force, const double *d_actions);
// Work of a failed call may still be running on the controller stream and the pipeline streams (the fan-out of the table
// path joins them into h->stream only at its end): wait for it, so that a later sg_synchronize / sg_upload / table regrow,
// which look at h->stream alone, never free a buffer a kernel is reading.
static void drain_streams(sg_handle *h)
{
    if (h->ctl_stream) (void)hipStreamSynchronize(h->ctl_stream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipGetLastError();
}
static int launch_rollout(sg_handle *h, int n_steps, int do_reset, int force, const double *d_actions)
{
    const int rc = launch_rollout_impl(h, n_steps, do_reset, force, d_actions);
    if (rc) drain_streams(h);
    return rc;
}
// what no fused rollout variant carries (launch_rollout_impl runs these step by step, sg_tick appends the same launches)
static bool unfused_off_road(const sg_handle *h) { return !h->wide && h->has_ped && (h->cfg.terminal_mask & SG_TERM_EGO_OFF_ROAD); }
static bool unfused_rss(const sg_handle *h)
{
    return !h->wide && h->rss_fused && h->WV == 8 && (h->has_ped || (h->cfg.terminal_mask & SG_TERM_EGO_OFF_ROAD));
}
static bool needs_unfused_extras(const sg_handle *h) { return unfused_off_road(h) || unfused_rss(h); }

static int launch_rollout_impl(sg_handle *h, int n_steps, int do_reset, int force, const double *d_actions)
{
    if (h->q_failed && do_reset != 1) return fail(h, SG_ERR_HIP, "%s", h->q_msg); // (a give-up is sticky until the batch starts anew)
    if (do_reset == 1) forget_queue_failure(h); // (State.reset of the whole batch)
    h->last_schedule = 0;
    if (h->wide) {
        h->n_launches = 0;
        h->launch_ev.clear();
        h->timing_now = n_steps >= 16;
        if (h->timing_now) HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
        const int rcw = launch_wide(h, n_steps, do_reset, force, d_actions);
        if (rcw) return rcw;
        if (h->timing_now) HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
        h->timed = h->timing_now;
        return SG_OK;
    }
    if (needs_unfused_extras(h)) {
        // Combinations no fused variant carries go step by step with the missing part as a launch of its own behind every step:
        // pedestrian agents AND the ego_off_road terminal condition (ego_off_road_kernel: check_terminal is the last thing a step
        // does to `done`, so a condition added afterwards is the same as one in the list); the RSS callback on scenarios of
        // 257..512 entities with pedestrian agents or ego_off_road (rss_kernel, as beyond 512).  A scenario that is done sits
        // the later launches out.  Rare enough combinations not to deserve kernel variants of their own.
        const bool off_road = unfused_off_road(h), rss = unfused_rss(h);
        const bool rss_was = h->rss_fused;
        if (rss) h->rss_fused = false;
        h->n_launches = 0;
        h->launch_ev.clear();
        h->timing_now = false;
        h->timed = false;
        size_t evn = 0;
        int rc1 = SG_OK;
        auto rss_launch = [&](int reset) {
            sg::rss_kernel<<<dim3((unsigned)h->R), dim3(512), 0, h->stream>>>(h->p, reset, h->d_rss_state, h->d_rss_code, h->d_rss_safe, h->d_rss_seen);
        };
        if (do_reset) {
            rc1 = launch_main(h, 0, do_reset, 0, nullptr, nullptr, false, &evn);
            if (!rc1 && rss) rss_launch(do_reset == 2 ? 2 : 1);
        }
        for (int k = 0; k < n_steps && !rc1; ++k) {
            rc1 = launch_main(h, 1, 0, force, d_actions ? d_actions + (size_t)k * h->R * 2 : nullptr, nullptr, false, &evn);
            if (rc1) break;
            if (off_road) sg::ego_off_road_kernel<<<dim3((unsigned)((h->R + 63) / 64)), dim3(64), 0, h->stream>>>(h->p);
            if (rss) rss_launch(0);
        }
        h->rss_fused = rss_was;
        if (rc1) return rc1;
        HIP_TRY(h, hipGetLastError());
        return SG_OK;
    }
    const int tab_min = h->tab_min, chunk_steps = std::max(1, h->chunk_steps), no_overlap = !h->overlap;
    h->n_launches = 0;
    h->launch_ev.clear();
    size_t ev_next = 0;
    // the table variant serves SG_TAB_LANES controlled lanes per wavefront; denser batches keep their controllers
    // in the rollout kernel, where they fill the wavefront anyway
    // (a crowd with riders: lanes of other kinds ride the crowd kernel on a pre-pass table; short calls -- the per-tick loop of
    // an RL driver -- keep the general pedestrian variant, like the table path keeps the in-kernel controllers)
    const bool riders = h->crowd_riders && crowd_road_ok(h) && !h->rss_fused && h->n_ctl > 0 && n_steps >= tab_min;
    // (the RSS callback inside the kernel: its controlled lanes ride the table too -- rollout_kernel_rss_tab -- which takes the
    // controller code out of the one variant that has no issue slot to spare; launches stay within the line-test queue)
    const bool rss_tab = h->rss_fused && h->WV == 1 && !h->has_ped && !(h->cfg.terminal_mask & SG_TERM_EGO_OFF_ROAD) && h->n_ext == 0 &&
                         h->n_ctl > 0 && n_steps >= tab_min && env_int("SG_RSS_TAB", 1) != 0;
    const bool use_tab = riders || rss_tab || (h->WV <= 4 && !h->has_ped && !h->rss_fused && !(h->cfg.terminal_mask & SG_TERM_EGO_OFF_ROAD) && h->n_ext == 0 && n_steps >= tab_min && h->max_ctl_per_block <= SG_TAB_LANES(h->G, h->WV));
    // short calls (the per-tick loop of an RL driver) are not timed: four event records cost more than their kernel
    h->timing_now = use_tab || n_steps >= 16;
    if (h->timing_now) HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    int rc = SG_OK;
    if (!use_tab && h->rss_fused && n_steps > h->rssq_steps) { // the line-test queues hold rssq_steps steps per launch (ensure_rss)
        for (int k0 = 0; k0 < n_steps && !rc; k0 += h->rssq_steps)
            rc = launch_main(h, std::min(h->rssq_steps, n_steps - k0), k0 == 0 ? do_reset : 0, force,
                             d_actions ? d_actions + (size_t)k0 * h->R * 2 : nullptr, nullptr, false, &ev_next);
    } else if (!use_tab) {
        rc = launch_main(h, n_steps, do_reset, force, d_actions, nullptr, false, &ev_next);
    } else {
        if (do_reset && (rc = launch_main(h, 0, do_reset, 0, nullptr, nullptr, false, &ev_next))) return rc;
        if (h->n_ctl == 0) { // nothing to integrate: the table variant reads (and ignores) one dummy row
            if (!h->d_tab[0]) {
                HIP_TRY(h, hipMalloc((void **)&h->d_tab[0], 64 * sizeof(double)));
                HIP_TRY(h, hipMalloc((void **)&h->d_tab[1], 64 * sizeof(double)));
                h->tab_bytes = 64 * sizeof(double);
                h->n_tab = 2;
                HIP_TRY(h, hipMemsetAsync(h->d_tab[0], 0, 64 * sizeof(double), h->stream));
            }
            rc = launch_main(h, n_steps, 0, force, nullptr, h->d_tab[0], true, &ev_next);
        } else {
            const size_t np = (size_t)h->p.n_ctl_pad, row = (size_t)sg::CT_PLANES * sg::CT_W * np; // doubles per step, all planes
            // chunk length: SG_CHUNK_STEPS, capped so that one table buffer stays under 1 GiB
            int ch = (int)std::min<size_t>((size_t)chunk_steps, std::max<size_t>(1, ((size_t)1 << 27) / row));
            ch = std::min(ch, n_steps);
            if (rss_tab) ch = std::min(ch, std::max(1, h->rssq_steps)); // one launch fills at most the line-test queue
            // one persistent launch (sgym_queue.hpp) where the batch is one wavefront per block and nothing rides along; the
            // pre-pass role must leave most of the wavefront slots to the rollout
            if (h->queue_mode && h->WV == 1 && !riders && !no_overlap && (size_t)h->p.n_ctl_pad / 64 <= std::min((size_t)h->n_simd, slots_of(h, rss_tab)) / 2 /* a SIMD of
                its own for every pre-pass role, and at least as many rollout wavefronts resident beside them */) {
                rc = launch_queue(h, n_steps, force, d_actions, ch, &ev_next, rss_tab);
                if (rc != SG_QUEUE_FALLBACK) {
                    if (rc) return rc;
                    if (h->timing_now) HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
                    h->timed = h->timing_now;
                    return SG_OK;
                }
                rc = SG_OK;
            }
            h->last_schedule = 1;
            // The chunk launches of rounds 1-4 (crowds with riders, the RSS table variant, tiles of several wavefronts, SG_QUEUE=0):
            // the pre-pass (its own stream) writes chunk c + 1 into the second table buffer while the rollout kernel reads chunk c
            const int NB = 2; // table buffers
            const int ctl_slice = h->ctl_slice; // the pre-pass in launches of ctl_slice steps (its load then moves between SIMDs)
            if (ch > h->p.tab_steps || (size_t)(h->p.tab_steps + 1) * row * sizeof(double) > h->tab_bytes || NB > h->n_tab) {
                // grow: tab_steps + 1 rows per lane is part of the table addressing
                const int ts = std::max(ch, h->p.tab_steps);
                const size_t need = (size_t)(ts + 1) * row * sizeof(double);
                if (need > h->tab_bytes || NB > h->n_tab) { // (the buffers outlive sg_upload: the next batch of the same shape reuses them)
                    HIP_TRY(h, hipStreamSynchronize(h->stream));
                    HIP_TRY(h, hipStreamSynchronize(h->ctl_stream));
                    for (int b = 0; b < 4; ++b) {
                        if (h->d_tab[b]) HIP_TRY(h, hipFree(h->d_tab[b]));
                        h->d_tab[b] = nullptr;
                    }
                    const size_t bytes = std::max(need, h->tab_bytes);
                    h->tab_bytes = 0;
                    h->n_tab = 0;
                    for (int b = 0; b < NB; ++b) {
                        HIP_TRY(h, hipMalloc((void **)&h->d_tab[b], bytes));
                        poison(h->stream, h->d_tab[b], bytes);
                    }
                    h->tab_bytes = bytes;
                    h->n_tab = NB;
                }
                h->p.tab_steps = ts;
            }
            hipStream_t cs = no_overlap ? h->stream : h->ctl_stream;
            hipEvent_t e;
            if (!no_overlap) { // the other streams start after everything queued so far (reset, uploads)
                if ((rc = get_event(h, ev_next++, &e))) return rc;
                HIP_TRY(h, hipEventRecord(e, h->stream));
                HIP_TRY(h, hipStreamWaitEvent(cs, e, 0));
            }
            const dim3 cgrid((unsigned)(np / 64));
            const bool rss_fast = env_int("SG_RSS_CTL_FAST", 1) != 0;
            // chunks of the time axis: lengths double from two slices up to `ch` -- the rollout kernel cannot start before
            // the table of its chunk exists, and the pre-pass of the chunks after it (about 0.4x the rollout kernel's time
            // per step) then always finishes under the rollout kernel
            std::vector<int> ck0, cn;
            for (int k0 = 0, n = 0, c = 0; k0 < n_steps; k0 += n, ++c) {
                n = std::min(std::min(ch, c < 20 ? (2 * h->ctl_slice) << c : ch), n_steps - k0);
                ck0.push_back(k0);
                cn.push_back(n);
            }
            const int C = (int)cn.size();
            std::vector<hipEvent_t> ctl_done((size_t)C, nullptr), chunk_done((size_t)C, nullptr);
            int ctl_issued = 0;
            auto issue_ctl = [&](int upto) -> int { // the pre-pass of the chunks up to `upto`, each into buffer (chunk mod NB)
                for (; ctl_issued <= upto && ctl_issued < C; ++ctl_issued) {
                    const int c = ctl_issued, k0 = ck0[(size_t)c], n = cn[(size_t)c];
                    double *tab = h->d_tab[c % NB];
                    if (!no_overlap && c >= NB) // the buffer is free once the rollout is through chunk c - NB
                        HIP_TRY(h, hipStreamWaitEvent(cs, chunk_done[(size_t)(c - NB)], 0));
                    for (int s0 = 0; s0 < n; s0 += ctl_slice) {
                        const int ns = std::min(ctl_slice, n - s0);
                        if (riders)
                            sgl::control(sgl::CTL_RIDERS, cgrid, cs, h->p, h->cfg.timestep, ns, c == 0 && s0 == 0, k0 + s0, d_actions, tab, s0, 0);
                        else if (rss_tab && rss_fast) // (the ego's metrics are the rollout kernel's, from its own velocities)
                            sgl::control(sgl::CTL_FAST, cgrid, cs, h->p, h->cfg.timestep, ns, c == 0 && s0 == 0, k0 + s0, d_actions, tab, s0, 0);
                        else
                            sgl::control(sgl::CTL_GENERAL, cgrid, cs, h->p, h->cfg.timestep, ns, c == 0 && s0 == 0, k0 + s0, d_actions, tab, s0,
                                         rss_tab ? 0 : 1);
                    }
                    HIP_TRY(h, hipGetLastError());
                    if (!no_overlap) {
                        const int rc2 = get_event(h, ev_next++, &ctl_done[(size_t)c]);
                        if (rc2) return rc2;
                        HIP_TRY(h, hipEventRecord(ctl_done[(size_t)c], cs));
                    }
                }
                return SG_OK;
            };
            for (int c = 0; c < C; ++c) {
                // (chunk c + NB - 1 goes into the buffer of chunk c - 1, whose launches were queued by the previous iteration)
                if ((rc = issue_ctl(std::min(C - 1, c + NB - 1)))) return rc;
                const int b = c % NB;
                const sg::TabGroups tg = one_group(h, h->d_tab[b], cn[(size_t)c]);
                if (!no_overlap) HIP_TRY(h, hipStreamWaitEvent(h->stream, ctl_done[(size_t)c], 0));
                if ((rc = launch_main(h, cn[(size_t)c], 0, force, nullptr, h->d_tab[b], true, &ev_next, &tg))) return rc;
                chunk_done[(size_t)c] = h->ev_pool[ev_next - 1];
            }
        }
    }
    if (rc) return rc;
    if (h->timing_now) HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    h->timed = h->timing_now;
    return SG_OK;
}

// ScenarioGym.rollout time-sliced (sgym_device.hpp, SliceArgs): the reset launch, the clock, the slices of the time axis
// side by side, the last executed step with the full state stores, the ordered sums.  For batches whose lanes are replay
// entities / replay agents -- and PID / vehicle agents: their controller pre-pass (control_kernel) then fills ONE table for
// the whole call, running ahead of the slices group by group.  Worth it when the batch alone cannot fill the chip (BASELINE
// config 2: 64 wavefronts; config 4's shards: 512); the results are bit-identical to launch_rollout's, the intermediate
// states are not written anywhere.
static void launch_slice_kernels(sg_handle *h, const Params &ps, const sg::SliceArgs &sa, dim3 grid, const double *tab)
{
    note_kernel(h, tab ? "sg::rollout_kernel_slice_tab<%d>" : "sg::rollout_kernel_slice<%d>", h->G);
    sgl::rollout_slice(h->G, grid, h->stream, ps, h->cfg.timestep, sa, tab);
}
static void launch_fixup_kernel(sg_handle *h, const Params &ps, const sg::SliceArgs &sa)
{
    const dim3 grid((unsigned)(h->NE / 64)), block(64);
    switch (h->G) {
    case 4: sg::replay_fixup_kernel<4><<<grid, block, 0, h->stream>>>(ps, sa, h->d_n_final); break;
    case 8: sg::replay_fixup_kernel<8><<<grid, block, 0, h->stream>>>(ps, sa, h->d_n_final); break;
    case 16: sg::replay_fixup_kernel<16><<<grid, block, 0, h->stream>>>(ps, sa, h->d_n_final); break;
    case 32: sg::replay_fixup_kernel<32><<<grid, block, 0, h->stream>>>(ps, sa, h->d_n_final); break;
    default: sg::replay_fixup_kernel<64><<<grid, block, 0, h->stream>>>(ps, sa, h->d_n_final); break;
    }
}

// (tests) a launch that does nothing for a while: SG_SLICE_DELAY_US puts one in front of the launch that materialises the last
// step of a time-sliced call, so that anything on the second stream that is NOT ordered behind that launch gets to run first
static __global__ void delay_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// slices, steps per slice and slices per launch group for a call of n_steps
struct SlicePlan { int S, len, SG; size_t bytes; };

static SlicePlan slice_plan(const sg_handle *h, int n_steps)
{
    const size_t nblk = h->NE / 64, R = (size_t)h->R, T1 = (size_t)n_steps + 1;
    const bool ctl = h->n_ctl > 0;
    SlicePlan pl{};
    // enough slices for ~4096 wavefronts per launch (mode 2, the tests' "always": short slices too)
    pl.SG = (int)std::max<size_t>(1, 4096 / nblk);
    if (ctl && h->slice_mode == 2) pl.SG = std::min(pl.SG, 8); // (the tests: several groups even for short calls)
    if (!ctl) { // one launch: all slices side by side, at least 64 steps each
        int S = (int)std::min<size_t>((size_t)pl.SG, (size_t)std::max(1, n_steps / (h->slice_mode == 2 ? 7 : 64)));
        pl.len = (n_steps + S - 1) / S;
        pl.S = (n_steps + pl.len - 1) / pl.len;
        pl.SG = pl.S;
    } else { // groups of SG slices of ~160 steps: a group starts when the pre-pass has passed its last step
        pl.len = std::min(n_steps, h->slice_mode == 2 ? 7 : std::max(16, env_int("SG_SLICE_LEN", 160)));
        pl.S = (n_steps + pl.len - 1) / pl.len;
    }
    // every array of the sliced path: |delta pose| rows, clocks, ego speeds, per-slice events and flags, and
    // the call-spanning controller table
    pl.bytes = nblk * T1 * 512 + h->clock_t0.size() * T1 * 8 + R * T1 * 16 +
               R * (size_t)pl.S * ((size_t)std::max(h->p.ev_cap, 1) * sizeof(sg_event) + 8) + R * 8 +
               (ctl ? (size_t)sg::CT_PLANES * sg::CT_W * (size_t)h->p.n_ctl_pad * T1 * 8 : 0);
    return pl;
}

static bool slicing_pays(const sg_handle *h, int n_steps)
{
    if (!h->slice_mode || !h->sliceable || h->WV != 1 || h->p.rec_cap > 0 || h->rss_enabled || h->n_ext > 0 ||
        (h->cfg.terminal_mask & SG_TERM_EGO_OFF_ROAD) || n_steps < (h->slice_mode == 2 ? 2 : 512))
        return false;
    if (h->n_ctl > 0 && h->max_ctl_per_block > SG_TAB_LANES(h->G, h->WV)) return false; // (as launch_rollout's table path)
    const size_t nblk = h->NE / 64;
    if (nblk > 1024 && h->slice_mode != 2) return false; // more than one wavefront per SIMD: the batch fills the chip by itself
    return slice_plan(h, n_steps).bytes <= ((size_t)std::max(1, env_int("SG_SLICE_MB", 8192)) << 20);
}

// SG_OK, an error, or SG_SLICE_FALLBACK: the arrays could not be allocated -- the caller takes the step-by-step path
#define SG_SLICE_FALLBACK 1
static int launch_sliced(sg_handle *h, int n_steps)
{
    const int R = h->R;
    const size_t nblk = h->NE / 64;
    const bool ctl = h->n_ctl > 0;
    const SlicePlan pl = slice_plan(h, n_steps);
    const int S = pl.S, len = pl.len;
    int rc;
    if (h->slice_T != n_steps || h->slice_S != S) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->ctl_stream));
        free_pool(h->slice_allocs);
        h->slice_T = -1;
        auto &A = h->slice_allocs;
        sg::SliceArgs sa{};
        double *tt = nullptr;
        const double *ct0 = nullptr;
        h->d_slice_tab = nullptr;
        rc = dev_alloc(h, A, &tt, (size_t)(n_steps + 1) * h->clock_t0.size(), false);
        sa.tt = tt;
        if (!rc) rc = dev_upload(h, A, &sa.clock_of, h->clock_of);
        if (!rc) rc = dev_upload(h, A, &ct0, h->clock_t0);
        h->d_clock_t0 = const_cast<double *>(ct0);
        if (!rc) rc = dev_alloc(h, A, &sa.dnorm, nblk * (size_t)(n_steps + 1) * 64, false);
        if (!rc) rc = dev_alloc(h, A, &sa.espeed, (size_t)R * (n_steps + 1), false);
        if (!rc) rc = dev_alloc(h, A, &sa.first_done, (size_t)R * S, false);
        if (!rc) rc = dev_alloc(h, A, &sa.ev, (size_t)R * S * std::max(h->p.ev_cap, 1), false);
        if (!rc) rc = dev_alloc(h, A, &sa.nev, (size_t)R * S, false);
        if (!rc) rc = dev_alloc(h, A, &h->d_n_final, (size_t)R, false);
        if (!rc) rc = dev_alloc(h, A, &h->d_slice_done, (size_t)R, false);
        if (!rc && ctl)
            rc = dev_alloc(h, A, &h->d_slice_tab, (size_t)sg::CT_PLANES * sg::CT_W * (size_t)h->p.n_ctl_pad * (size_t)(n_steps + 1), false);
        if (rc) { // out of device memory: not an error of the call -- the plain path needs none of these arrays
            (void)hipGetLastError();
            (void)hipStreamSynchronize(h->stream);
            free_pool(h->slice_allocs);
            h->err.clear();
            return SG_SLICE_FALLBACK;
        }
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        sa.n_slices = S; sa.len = len; sa.n_total = n_steps;
        h->slice_args = sa;
        h->slice_T = n_steps;
        h->slice_S = S;
    }
    sg::SliceArgs sa = h->slice_args;
    Params ps = h->p;      // the kernels of this path address the controller table with the call's length
    ps.tab_steps = n_steps;
    const double *tab = ctl ? h->d_slice_tab : nullptr;
    h->n_launches = 0;
    h->launch_ev.clear();
    h->timing_now = true;
    size_t ev_next = 0;
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    if ((rc = launch_main(h, 0, 1, 0, nullptr, nullptr, false, &ev_next))) return rc; // State.reset (not counted as a hot-path launch)
    hipEvent_t e_reset = nullptr;
    if (ctl) { // the pre-pass reads the reset state
        if ((rc = get_event(h, ev_next++, &e_reset))) return rc;
        HIP_TRY(h, hipEventRecord(e_reset, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->ctl_stream, e_reset, 0));
    }
    HIP_TRY(h, hipMemsetAsync(sa.first_done, 0x7f, (size_t)R * S * sizeof(int), h->stream)); // 0x7f7f7f7f: "never"
    HIP_TRY(h, hipMemsetAsync(sa.nev, 0, (size_t)R * S * sizeof(int), h->stream));
    const int n_clocks = (int)h->clock_t0.size();
    sg::clock_kernel<<<dim3((unsigned)((n_clocks + 63) / 64)), dim3(64), 0, h->stream>>>(h->d_clock_t0, n_clocks, h->cfg.timestep, n_steps,
                                                                                        const_cast<double *>(sa.tt));
    sa.mode = 0;
    const int ctl_len = std::max(1, env_int("SG_SLICE_CTL_STEPS", 2048)); // steps per control_kernel launch
    for (int s0 = 0; s0 < S; s0 += pl.SG) {
        const int ns = std::min(pl.SG, S - s0);
        if (ctl) { // rows (s0 * len, (s0 + ns) * len] of the table, then the event the group waits for
            const int k0 = s0 * len, k1 = std::min(n_steps, (s0 + ns) * len);
            const dim3 cgrid((unsigned)(ps.n_ctl_pad / 64));
            for (int k = k0; k < k1; k += ctl_len)
                sgl::control(sgl::CTL_FAST, cgrid, h->ctl_stream, ps, h->cfg.timestep, std::min(ctl_len, k1 - k), k == 0, k, nullptr, h->d_slice_tab, k, 0);
            HIP_TRY(h, hipGetLastError());
            hipEvent_t e_c = nullptr;
            if ((rc = get_event(h, ev_next++, &e_c))) return rc;
            HIP_TRY(h, hipEventRecord(e_c, h->ctl_stream));
            HIP_TRY(h, hipStreamWaitEvent(h->stream, e_c, 0));
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if ((rc = get_event(h, ev_next, &e0)) || (rc = get_event(h, ev_next + 1, &e1))) return rc;
        HIP_TRY(h, hipEventRecord(e0, h->stream));
        sa.slice0 = s0;
        launch_slice_kernels(h, ps, sa, dim3((unsigned)nblk, (unsigned)ns), tab);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(e1, h->stream));
        h->launch_ev.push_back((int)ev_next);
        ++h->n_launches;
        ev_next += 2;
    }
    sa.slice0 = 0;
    sg::slice_final_kernel<<<dim3((unsigned)((R + 63) / 64)), dim3(64), 0, h->stream>>>(ps, sa, h->d_n_final, h->d_slice_done);
    sa.mode = 1;
    sa.n_final = h->d_n_final;
    if (const int us = env_int("SG_SLICE_DELAY_US", 0)) delay_kernel<<<dim3(1), dim3(64), 0, h->stream>>>((long long)us * 100ll); // 100 MHz
    launch_slice_kernels(h, ps, sa, dim3((unsigned)nblk, 1), tab);
    {   // the per-scenario ordered pass (a serial recurrence per scenario) on the second stream, beside the per-entity pass --
        // and AFTER the launch that materialises the last step: that launch starts from the scenario records of the reset
        // (clock, `done`, step count), which this pass overwrites with the final ones.  (Until round 5 it only waited for
        // slice_final_kernel and normally lost the race by a few microseconds; when it won -- seen on the first suite run of
        // cold boxes, one run in four -- a scenario it had already marked done sat the last launch out and kept its reset
        // poses.)
        hipEvent_t e_m1 = nullptr;
        if ((rc = get_event(h, ev_next++, &e_m1))) return rc;
        HIP_TRY(h, hipEventRecord(e_m1, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->ctl_stream, e_m1, 0));
        sg::replay_scenario_fixup_kernel<<<dim3((unsigned)((R + 63) / 64)), dim3(64), 0, h->ctl_stream>>>(ps, sa, h->d_n_final, h->d_slice_done);
        if (ctl && h->p.ev_cap > 0) // the controlled egos' (and hazards') poses at the events: rows of the table
            sg::event_ego_pose_kernel<<<dim3((unsigned)R), dim3(64), 0, h->ctl_stream>>>(ps, one_group(h, tab, n_steps));
    }
    launch_fixup_kernel(h, ps, sa);
    HIP_TRY(h, hipGetLastError());
    {
        hipEvent_t e_sc = nullptr;
        if ((rc = get_event(h, ev_next++, &e_sc))) return rc;
        HIP_TRY(h, hipEventRecord(e_sc, h->ctl_stream));
        HIP_TRY(h, hipStreamWaitEvent(h->stream, e_sc, 0));
    }
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return SG_OK;
}

extern "C" int sg_set_social_force(sg_handle *h, const sg_social_force *params)
{
    if (!h || !params) return SG_ERR_INVALID;
    h->sf = *params;
    h->p.sf = *params;
    ++h->generation;
    return SG_OK;
}

extern "C" int sg_set_ped_behaviour(sg_handle *h, int32_t behaviour)
{
    if (!h) return SG_ERR_INVALID;
    if (behaviour != SG_PED_SOCIAL_FORCE && behaviour != SG_PED_RANDOM_WALK)
        return fail(h, SG_ERR_INVALID, "sg_set_ped_behaviour: unknown behaviour %d", behaviour);
    if (h->uploaded && behaviour != h->ped_behaviour)
        return fail(h, SG_ERR_STATE, "sg_set_ped_behaviour: call before sg_upload (the batch's kernels are chosen there)");
    h->ped_behaviour = behaviour;
    h->p.ped_behaviour = behaviour;
    ++h->generation;
    return SG_OK;
}

static void apply_noise(sg_handle *h);
// PedestrianAgent(..., behaviour=...) per agent (pedestrian/agent.py:18-41): the distinct models of the batch + the model of
// every entity slot.  One model: the handle-wide setters.
extern "C" int sg_set_ped_models(sg_handle *h, int32_t n_models, const sg_ped_model *models, const int32_t *model_of)
{
    if (!h) return SG_ERR_INVALID;
    if (n_models < 1 || n_models > SG_MAX_PED_MODELS || !models)
        return fail(h, SG_ERR_INVALID, "sg_set_ped_models: n_models=%d (1 .. %d) or null models", n_models, SG_MAX_PED_MODELS);
    if (h->uploaded) return fail(h, SG_ERR_STATE, "sg_set_ped_models: call before sg_upload (the batch's kernels are chosen there)");
    for (int m = 0; m < n_models; ++m) {
        if (models[m].behaviour != SG_PED_SOCIAL_FORCE && models[m].behaviour != SG_PED_RANDOM_WALK)
            return fail(h, SG_ERR_INVALID, "sg_set_ped_models: model %d: unknown behaviour %d", m, models[m].behaviour);
        if (!(models[m].std_lon >= 0.0) || !(models[m].std_lat >= 0.0))
            return fail(h, SG_ERR_INVALID, "sg_set_ped_models: model %d: std must be >= 0", m);
    }
    if (n_models > 1 && !model_of) return fail(h, SG_ERR_INVALID, "sg_set_ped_models: several models need model_of[n_scenarios * n_entities]");
    // (every refusal before anything of the handle changes: a refused call leaves the models it had)
    if (n_models > 1)
        for (int r = 0; r < h->R; ++r)
            for (int e = 0; e < h->E; ++e)
                if (model_of[(size_t)r * h->E + e] >= n_models)
                    return fail(h, SG_ERR_INVALID, "sg_set_ped_models: model_of[%d][%d] = %d >= n_models = %d", r, e, model_of[(size_t)r * h->E + e], n_models);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    // model 0 is also what the handle-wide fields say (the single-model kernels, the oracle of a one-model batch)
    h->sf = models[0].params;
    h->p.sf = h->sf;
    h->ped_behaviour = models[0].behaviour;
    h->p.ped_behaviour = h->ped_behaviour;
    if (h->noise_mode != SG_NOISE_OFF) { h->noise_std[0] = models[0].std_lon; h->noise_std[1] = models[0].std_lat; }
    h->n_ped_models = n_models;
    h->models_all_sf = true;
    for (int m = 0; m < n_models; ++m) h->models_all_sf = h->models_all_sf && models[m].behaviour == SG_PED_SOCIAL_FORCE;
    if (n_models > 1) {
        std::vector<double> rows((size_t)n_models * sg::PM_W, 0.0);
        for (int m = 0; m < n_models; ++m) {
            double *r = rows.data() + (size_t)m * sg::PM_W;
            r[sg::PM_BEHAVIOUR] = (double)models[m].behaviour;
            memcpy(r + sg::PM_SF, &models[m].params, sizeof(sg_social_force));
            r[sg::PM_STD_LON] = models[m].std_lon; // (read only when the handle's noise mode is not off)
            r[sg::PM_STD_LAT] = models[m].std_lat;
        }
        std::vector<int32_t> mo(h->NE, 0);
        for (int r = 0; r < h->R; ++r)
            for (int e = 0; e < h->E; ++e) {
                const int32_t v = model_of[(size_t)r * h->E + e];
                mo[(size_t)r * h->EP + e] = v < 0 ? 0 : v;
            }
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->d_ped_models) HIP_TRY(h, hipFree(h->d_ped_models));
        if (h->d_model_of) HIP_TRY(h, hipFree(h->d_model_of));
        h->d_ped_models = nullptr;
        h->d_model_of = nullptr;
        HIP_TRY(h, hipMalloc((void **)&h->d_ped_models, rows.size() * sizeof(double)));
        HIP_TRY(h, hipMalloc((void **)&h->d_model_of, mo.size() * sizeof(int32_t)));
        HIP_TRY(h, hipMemcpy(h->d_ped_models, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_model_of, mo.data(), mo.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    apply_noise(h);
    ++h->generation;
    return SG_OK;
}

static void apply_noise(sg_handle *h)
{
    h->p.noise_mode = h->noise_mode;
    h->p.noise_std_lon = h->noise_std[0];
    h->p.noise_std_lat = h->noise_std[1];
    h->p.noise_normals = h->d_normals;
    h->p.noise_len = h->noise_len;
    h->p.noise_seed = h->noise_seed;
}

extern "C" int sg_set_ped_noise(sg_handle *h, int32_t mode, double std_lon, double std_lat, const double *normals,
                                int64_t per_scenario, uint64_t seed)
{
    if (!h) return SG_ERR_INVALID;
    if (mode < SG_NOISE_OFF || mode > SG_NOISE_DEVICE) return fail(h, SG_ERR_INVALID, "sg_set_ped_noise: unknown mode %d", mode);
    if (!(std_lon >= 0.0) || !(std_lat >= 0.0)) return fail(h, SG_ERR_INVALID, "sg_set_ped_noise: std must be >= 0");
    if (mode == SG_NOISE_STREAM && (!normals || per_scenario < 2))
        return fail(h, SG_ERR_INVALID, "sg_set_ped_noise: SG_NOISE_STREAM needs [n_scenarios][per_scenario >= 2] variates");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->d_normals) { (void)hipFree(h->d_normals); h->d_normals = nullptr; }
    h->noise_len = 0;
    if (mode == SG_NOISE_STREAM) {
        const size_t n = (size_t)h->R * (size_t)per_scenario;
        HIP_TRY(h, hipMalloc((void **)&h->d_normals, n * sizeof(double)));
        HIP_TRY(h, hipMemcpy(h->d_normals, normals, n * sizeof(double), hipMemcpyHostToDevice));
        h->noise_len = per_scenario;
    }
    h->noise_mode = mode;
    h->noise_std[0] = mode == SG_NOISE_OFF ? 0.0 : std_lon;
    h->noise_std[1] = mode == SG_NOISE_OFF ? 0.0 : std_lat;
    h->noise_seed = seed;
    apply_noise(h);
    ++h->generation;
    return SG_OK;
}

extern "C" int sg_upload(sg_handle *h, const sg_scenarios *sc)
{
    if (!h || !sc) return SG_ERR_INVALID;
    if (!sc->kind || !sc->etype || !sc->bbox || !sc->knot_off || !sc->knots || !sc->ego || !sc->t0 || !sc->length)
        return fail(h, SG_ERR_INVALID, "sg_upload: null array in sg_scenarios");
    const auto t_entry = std::chrono::steady_clock::now();
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    forget_queue_failure(h);
    HIP_TRY(h, hipStreamSynchronize(h->ctl_stream));
    h->static_allocs.rewind(); // (buffers of the previous batch are reused where they are large enough)
    h->state_allocs.rewind();
    free_pool(h->road_allocs); // the networks belong to a batch (net_of_scenario)
    // the RSS records and the line-test queue (GiBs) belong to the handle's shape, not to the batch: they stay allocated and
    // start anew (ensure_rss / ensure_rssq on first use; a hipFree + hipMalloc of the queue per upload stalled every tenth
    // or so sg_upload of a sweep for a second)
    h->rss_stale = true;
    h->p.rss_state = nullptr; h->p.rss_code = nullptr; h->p.rss_seen = nullptr; h->p.rss_safe = nullptr;
    h->p.rssq = nullptr; h->p.rssq_n = nullptr;
    h->has_road = false;
    h->road = sg::RoadIndex{};
    h->uploaded = false; // (the controller table buffers stay: launch_rollout regrows them when the new batch needs more)
    h->ego_first = true;
    ++h->generation;
    // pedestrian agents are compiled for tiles of >= 16 lanes
    h->has_ped = false;
    h->all_ped = true;
    h->sliceable = true;
    free_pool(h->slice_allocs);
    h->slice_T = -1;
    for (size_t i = 0; i < (size_t)h->R * h->E; ++i) {
        h->sliceable = h->sliceable && (sc->kind[i] == SG_KIND_NONE || sc->kind[i] == SG_KIND_REPLAY || sc->kind[i] == SG_KIND_AGENT_REPLAY ||
                                        sc->kind[i] == SG_KIND_AGENT_PID || sc->kind[i] == SG_KIND_AGENT_VEHICLE);
        h->has_ped = h->has_ped || sc->kind[i] == SG_KIND_AGENT_PEDESTRIAN;
        h->all_ped = h->all_ped && (sc->kind[i] == SG_KIND_NONE || (sc->kind[i] == SG_KIND_AGENT_PEDESTRIAN && sc->etype[i] == 1));
    }
    if (h->has_ped && h->WV == 1 && h->G < 16) { h->G = 16; h->EP = 16; h->NE = (((size_t)h->R * h->EP + 63) / 64) * 64; }
    h->crowd_riders = false;
    if (h->has_ped && !h->all_ped && h->G == 64 && h->WV <= 4 && crowd_allowed(h) && h->n_ped_models <= 1 /* (the riders variant knows one model) */ &&
        env_int("SG_CROWD_RIDERS", 1) != 0) {
        bool ok = true; // pedestrian agents of catalog type Pedestrian, and nothing the pre-pass cannot ride for
        for (size_t i = 0; i < (size_t)h->R * h->E && ok; ++i)
            ok = sc->kind[i] == SG_KIND_AGENT_PEDESTRIAN ? sc->etype[i] == 1 : sc->kind[i] != SG_KIND_AGENT_EXTERNAL;
        h->crowd_riders = ok;
    }
    if (h->has_ped && (!sc->route_off || !sc->routes)) return fail(h, SG_ERR_INVALID, "sg_upload: pedestrian agents need route_off/routes");
    // (257..512 entities: pedestrian agents run the general pedestrian variant, rollout_kernel<64, 8, true, false>; the crowd
    // kernels, the riders' pre-pass and road networks with pedestrians stop at 256)
    const int R = h->R, E = h->E, EP = h->EP;
    const size_t NE = h->NE;
    const bool trace = env_int("SG_TRACE_UPLOAD", 0) != 0; // stage timings on stderr
    auto t_last = t_entry;
    auto stage = [&](const char *name) {
        if (!trace) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "sg_upload: %-28s %7.2f ms\n", name, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };

    stage("entry (syncs, pools, kinds)");
    // ---- the knots (by far the largest array: 1.6 GB for 4096 x 64 x 128) start crossing PCIe NOW, from a thread of their
    // own on the second stream, while the host validates the batch and builds the union grids below.  Their extent comes
    // from knot_off, which is checked first (a bad offset must not turn into an out-of-bounds read of the copy).
    const int64_t rows_total = sc->knot_off[(size_t)R * E];
    {
        bool ok = sc->knot_off[0] >= 0;
        for (size_t i = 0; i < (size_t)R * E && ok; ++i) ok = sc->knot_off[i + 1] >= sc->knot_off[i];
        if (!ok) return fail(h, SG_ERR_INVALID, "sg_upload: knot_off is not monotone");
    }
    double *d_knots = nullptr;
    {
        int rc0 = dev_alloc(h, h->static_allocs, &d_knots, (size_t)std::max<int64_t>(rows_total, 1) * 7, false);
        if (rc0) return rc0;
    }
    // The copy goes in UP_CHUNKS pieces on scenario boundaries, an event after each: the stage-1 resample of a piece's
    // scenarios (build_grid_kernel, at the end of this function) runs while the later pieces are still crossing.
    // Ordinary (pageable) host memory goes in one piece: the runtime stages it through its own buffers, and several large
    // copies in flight from such memory disturbed the host threads below (every other upload took 60 ms instead of 34).
    constexpr int UP_MAX = 4;
    int UP_CHUNKS = 1;
    {
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, sc->knots) == hipSuccess && attr.type == hipMemoryTypeHost) UP_CHUNKS = UP_MAX;
        (void)hipGetLastError(); // (an unregistered pointer is reported as an error by some runtimes)
    }
    while (h->up_ev.size() < (size_t)UP_CHUNKS) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->up_ev.push_back(e);
    }
    int chunk_r[UP_MAX + 1];
    for (int c = 0; c <= UP_CHUNKS; ++c) chunk_r[c] = (int)((int64_t)R * c / UP_CHUNKS);
    hipError_t copy_err = hipSuccess;
    std::atomic<int> issued{0};
    std::thread copier([&]() {
        if (rows_total > 0) copy_err = hipSetDevice(h->cfg.device);
        for (int c = 0; c < UP_CHUNKS && rows_total > 0 && copy_err == hipSuccess; ++c) {
            const int64_t a = sc->knot_off[(size_t)chunk_r[c] * E], b = sc->knot_off[(size_t)chunk_r[c + 1] * E];
            if (b > a)
                copy_err = hipMemcpyAsync(d_knots + a * 7, sc->knots + a * 7, (size_t)(b - a) * 7 * sizeof(double), hipMemcpyHostToDevice, h->ctl_stream);
            if (copy_err == hipSuccess) copy_err = hipEventRecord(h->up_ev[c], h->ctl_stream);
            issued.store(c + 1, std::memory_order_release);
        }
        issued.store(UP_CHUNKS, std::memory_order_release); // (also after an error: nobody waits for a piece that will not come)
        if (rows_total > 0 && copy_err == hipSuccess) copy_err = hipStreamSynchronize(h->ctl_stream);
    });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } copier_guard{copier}; // every return path waits

    // ---- validate + block re-layout + union knot grids (host, one parallel pass over the scenarios) ----
    const size_t nblk = NE / 64;
    const size_t stat_n = nblk * sg::ST_COUNT * 64;
    // (48 MB for 4096 x 64: every slot is written by the pass below.  This and the other host buffers of an upload belong to
    // the handle: mapping, faulting in and unmapping them anew took 6 ms of every call)
    // It is page-locked: the copy engine takes it from where the pass wrote it.
    if (h->up_stat_cap < stat_n) {
        if (h->up_stat) HIP_TRY(h, hipHostFree(h->up_stat));
        h->up_stat = nullptr;
        h->up_stat_cap = 0;
        HIP_TRY(h, hipHostMalloc((void **)&h->up_stat, stat_n * sizeof(double), hipHostMallocDefault));
        h->up_stat_cap = stat_n;
    }
    double *stat = h->up_stat;
    auto S = [&](size_t ent, int f) -> double & { return stat[(ent >> 6) * sg::ST_COUNT * 64 + (size_t)f * 64 + (ent & 63)]; };
    auto SI = [&](size_t ent, int f) -> int64_t & { return *reinterpret_cast<int64_t *>(&S(ent, f)); };
    auto slot_defaults = [&](size_t o) { // a padding slot: never present
        for (int f = 0; f < sg::ST_COUNT; ++f) S(o, f) = 0.0;
        for (int q = 0; q < 4; ++q) S(o, sg::ST_BW + q) = 1.0;
        for (int q = 0; q < sg::NCTRL_ROWS; ++q) S(o, sg::ST_CTRL + q) = kDefaultCtrl[q];
        SI(o, sg::ST_META) = SG_KIND_NONE | (2 << 8);
        SI(o, sg::ST_CTL) = -1;
    };
    for (size_t o = (size_t)R * EP; o < NE; ++o) slot_defaults(o); // the tail of the last block
    std::vector<int32_t> ctl_ent; // controlled lanes (PID / vehicle agents) in entity order
    std::vector<char> zpr_zero;   // [R] every knot of the scenario has z = pitch = roll = +0.0 (a planar recording: the usual case)
    int n_ext = 0;
    std::vector<sg::ScenStatic> sstat(R);
    std::vector<std::vector<double>> &grids = h->up_grids; // BatchReplayEntity union knot grid per scenario (entity/batch.py:83-95)
    grids.resize(R);
    for (auto &g : grids) g.clear(); // (capacity stays)
    {   // scenarios are validated, re-laid out and given their union grid in parallel (the strictly-increasing check walks
        // every knot: 33 M for the 4096 x 64 x 128 batch; the grid sorts them); the first error by scenario index is reported
        const unsigned nthr = host_threads();
        std::vector<std::string> errs(nthr);
        std::vector<int> err_r(nthr, R), ext_cnt(nthr, 0);
        std::vector<char> ego_nz(nthr, 0);
        zpr_zero.assign(R, 1);
        auto work = [&](unsigned w) {
            auto bad = [&](int r, const char *fmt, size_t i, int v) {
                char buf[256];
                snprintf(buf, sizeof buf, fmt, i, v);
                errs[w] = buf;
                err_r[w] = r;
            };
            std::vector<double> times, merged; // scratch of the union grid
            for (int r = (int)((int64_t)R * w / nthr); r < (int)((int64_t)R * (w + 1) / nthr); ++r) {
                if (sc->ego[r] < 0 || sc->ego[r] >= E) return bad(r, "sg_upload: ego[%zu]=%d out of range", (size_t)r, sc->ego[r]);
                sstat[r].ego = sc->ego[r];
                if (sc->ego[r] != 0) ego_nz[w] = 1;
                sstat[r].t0 = sc->t0[r];
                sstat[r].length = sc->length[r];
                for (int e = 0; e < EP; ++e) slot_defaults((size_t)r * EP + e);
                for (int e = 0; e < E; ++e) {
                    size_t i = (size_t)r * E + e, o = (size_t)r * EP + e;
                    int k = sc->kind[i];
                    if (k < SG_KIND_NONE || k > SG_KIND_AGENT_EXTERNAL) return bad(r, "sg_upload: kind[%zu]=%d unknown", i, k);
                    if (k == SG_KIND_AGENT_EXTERNAL) ++ext_cnt[w];
                    if (k == SG_KIND_AGENT_PEDESTRIAN) {
                        int64_t ra = sc->route_off[i], rb = sc->route_off[i + 1];
                        if (ra < 0 || rb <= ra) return bad(r, "sg_upload: pedestrian agent %zu has no route (%d)", i, 0);
                        SI(o, sg::ST_ROUTE) = ra | ((rb - ra) << 48);
                    }
                    int64_t a = sc->knot_off[i], b = sc->knot_off[i + 1];
                    if (a < 0 || b < a || b > rows_total) return bad(r, "sg_upload: knot_off not monotone at %zu (%d)", i, 0);
                    if (k != SG_KIND_NONE && b == a) return bad(r, "sg_upload: entity %zu has no knots (%d)", i, 0);
                    SI(o, sg::ST_META) = (int64_t)k | ((int64_t)(sc->etype[i] & 0xff) << 8) | ((int64_t)(b - a) << 32);
                    SI(o, sg::ST_KNOT_OFF) = a;
                    for (int q = 0; q < 4; ++q) S(o, sg::ST_BW + q) = sc->bbox[i * 4 + q];
                    if (sc->ctrl) for (int q = 0; q < sg::NCTRL_ROWS; ++q) S(o, sg::ST_CTRL + q) = sc->ctrl[i * SG_NCTRL + q];
                    if (b > a) {
                        S(o, sg::ST_MIN_T) = sc->knots[(size_t)a * 7];
                        S(o, sg::ST_MAX_T) = sc->knots[(size_t)(b - 1) * 7];
                        for (int64_t j = a + 1; j < b; ++j)
                            if (!(sc->knots[(size_t)j * 7] > sc->knots[(size_t)(j - 1) * 7]))
                                return bad(r, "sg_upload: knot times of entity %zu are not strictly increasing (%d)", i, 0);
                        // the row is in cache: are z, pitch and roll +0.0 in every knot (bit patterns: -0.0 and NaN are not)?
                        uint64_t any = 0;
                        for (int64_t j = a; j < b; ++j) {
                            const uint64_t *kr = reinterpret_cast<const uint64_t *>(sc->knots + (size_t)j * 7);
                            any |= kr[3] | kr[5] | kr[6];
                        }
                        if (any) zpr_zero[r] = 0;
                    }
                }
                // the union grid (np.unique of the concatenated knot times), while the scenario's knots are in cache.  Every
                // entity's times are strictly increasing (checked above), so the union grows by merging sorted lists -- and
                // an entity on the grid found so far (the usual case: one recording, one clock) costs one comparison per knot
                std::vector<double> &g = grids[r];
                for (int e = 0; e < E; ++e) {
                    size_t i = (size_t)r * E + e;
                    if (sc->kind[i] != SG_KIND_REPLAY) continue;
                    const int64_t a = sc->knot_off[i], b = sc->knot_off[i + 1];
                    const size_t n = (size_t)(b - a);
                    if (n == 1) { // batch.py:85-88: a second knot 0.1 s later
                        const double v0 = sc->knots[(size_t)a * 7], two[2] = {v0 == v0 ? v0 : 0.0 /* np.nan_to_num */, v0 + 1e-1};
                        merged.clear();
                        std::set_union(g.begin(), g.end(), two, two + 2, std::back_inserter(merged));
                        g.swap(merged);
                        continue;
                    }
                    bool same = g.size() == n;
                    for (size_t j = 0; j < n && same; ++j) same = g[j] == sc->knots[(size_t)(a + (int64_t)j) * 7];
                    if (same) continue;
                    times.resize(n);
                    for (size_t j = 0; j < n; ++j) times[j] = sc->knots[(size_t)(a + (int64_t)j) * 7];
                    merged.clear();
                    std::set_union(g.begin(), g.end(), times.begin(), times.end(), std::back_inserter(merged));
                    g.swap(merged);
                }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < nthr; ++w) pool.emplace_back(work, w);
        work(0);
        for (auto &th : pool) th.join();
        unsigned first = 0;
        for (unsigned w = 1; w < nthr; ++w)
            if (err_r[w] < err_r[first]) first = w;
        if (err_r[first] < R) return fail(h, SG_ERR_INVALID, "%s", errs[first].c_str());
        for (unsigned w = 0; w < nthr; ++w) {
            n_ext += ext_cnt[w];
            if (ego_nz[w]) h->ego_first = false;
        }
        // the controlled lanes in entity order (their index is the column of the controller table)
        for (int r = 0; r < R; ++r)
            for (int e = 0; e < E; ++e) {
                const int k = sc->kind[(size_t)r * E + e];
                if (k == SG_KIND_AGENT_PID || k == SG_KIND_AGENT_VEHICLE ||
                    (h->crowd_riders && (k == SG_KIND_REPLAY || k == SG_KIND_AGENT_REPLAY))) {
                    const size_t o = (size_t)r * EP + e;
                    SI(o, sg::ST_CTL) = (int64_t)ctl_ent.size();
                    ctl_ent.push_back((int32_t)o);
                }
            }
    }

    stage("validate + re-layout + union grids");
    std::vector<int64_t> grid_off(R + 1, 0);
    for (int r = 0; r < R; ++r) {
        sstat[r].grid_n = (int32_t)grids[r].size();
        sstat[r].grid_off = grid_off[r];
        grid_off[r + 1] = grid_off[r] + sstat[r].grid_n;
    }
    const int64_t total_rows = grid_off[R];
    std::vector<double> &grid_t = h->up_grid_t;
    std::vector<int32_t> &row_scen = h->up_row_scen;
    grid_t.resize((size_t)total_rows);
    row_scen.resize((size_t)total_rows);
    for (int r = 0; r < R; ++r) {
        std::copy(grids[r].begin(), grids[r].end(), grid_t.begin() + grid_off[r]);
        std::fill(row_scen.begin() + grid_off[r], row_scen.begin() + grid_off[r + 1], r);
    }

    {   // scenarios that start at the same time run on the same clock (launch_sliced)
        std::vector<std::pair<uint64_t, int>> key(R);
        for (int r = 0; r < R; ++r) { uint64_t b; std::memcpy(&b, &sstat[r].t0, 8); key[r] = {b, r}; }
        std::sort(key.begin(), key.end());
        h->clock_t0.clear();
        h->clock_of.assign(R, 0);
        for (int i = 0; i < R; ++i) {
            if (i == 0 || key[i].first != key[i - 1].first) h->clock_t0.push_back(sstat[key[i].second].t0);
            h->clock_of[key[i].second] = (int)h->clock_t0.size() - 1;
        }
    }
    stage("union grids");
    // ---- device copies ----
    Params &p = h->p;
    p = Params{};
    p.R = R; p.E = E; p.EP = EP;
    p.WV = h->WV; p.FROWS = SG_F_COLL + h->WV;
    p.sf = h->sf;
    p.ped_behaviour = h->ped_behaviour;
    p.n_ped_models = h->n_ped_models;
    p.ped_models = h->d_ped_models;
    p.model_of = h->d_model_of;
    apply_noise(h);
    p.ped_serial = h->ped_serial;
    p.ctl_general = env_int("SG_CTL_FAST", 1) == 0;
    p.reset_mask = h->d_reset_mask;
    p.persist = h->cfg.persist;
    p.term_mask = h->cfg.terminal_mask;
    p.rec_cap = h->cfg.record_capacity > 0 ? h->cfg.record_capacity : 0;
    p.ev_cap = h->cfg.event_capacity > 0 ? h->cfg.event_capacity : 0;
    auto &SA = h->static_allocs;
    int rc = 0;
    {
        double *d_stat = nullptr;
        if ((rc = dev_alloc(h, SA, &d_stat, stat_n, false))) return rc;
        HIP_TRY(h, hipMemcpyAsync(d_stat, stat, stat_n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        p.stat = d_stat;
    }
    if ((rc = dev_upload(h, SA, &p.sstat, sstat))) return rc;
    if ((rc = dev_upload(h, SA, &p.grid_t, grid_t))) return rc;
    {
        p.knots = d_knots; // (on its way since the top of the call)
        const int32_t *drs = nullptr;
        if ((rc = dev_upload(h, SA, &drs, row_scen))) return rc;
        h->d_row_scen = const_cast<int32_t *>(drs);
        h->total_rows = total_rows;
        if ((rc = dev_alloc(h, SA, &p.grid_y, (size_t)total_rows * 6 * EP, false))) return rc;
    }
    {   // pedestrian routes + the 64-gon table of Point.buffer (host libm, as shapely's caller sees it)
        size_t rrows = sc->route_off ? (size_t)sc->route_off[(size_t)R * E] : 0;
        std::vector<double> routes(sc->routes, sc->routes + rrows * 2);
        if (routes.empty()) routes.assign(2, 0.0);
        if ((rc = dev_upload(h, SA, &p.routes, routes))) return rc;
        std::vector<double> gon(128);
        for (int i = 0; i < 64; ++i) { double a = 2.0 * 3.141592653589793 * i / 64; gon[2 * i] = std::cos(a); gon[2 * i + 1] = std::sin(a); }
        if ((rc = dev_upload(h, SA, &p.gon, gon))) return rc;
    }
    h->n_ext = n_ext;
    {   // external poses start as "None" (NaN: all-ones bytes) for every slot
        double *d = nullptr;
        if ((rc = dev_alloc(h, SA, &d, NE * 6, false))) return rc;
        HIP_TRY(h, hipMemsetAsync(d, 0xFF, NE * 6 * sizeof(double), h->stream));
        h->d_ext = d;
        p.ext_pose = d;
    }
    h->n_ctl = (int)ctl_ent.size();
    h->max_ctl_per_block = 0;
    for (size_t i = 0, run = 0; i < ctl_ent.size(); ++i) { // ctl_ent is sorted by entity index
        run = (i > 0 && (ctl_ent[i] >> 6) == (ctl_ent[i - 1] >> 6)) ? run + 1 : 1;
        h->max_ctl_per_block = std::max(h->max_ctl_per_block, (int)run);
    }
    h->planar = n_ext == 0 && env_int("SG_PLANAR", 1) != 0; // (the table variant of the rollout kernel: rollout_kernel_tab_planar)
    for (int r = 0; r < R && h->planar; ++r) h->planar = zpr_zero[r] != 0;
    ctl_ent.resize(((ctl_ent.size() + 63) / 64) * 64, -1);
    p.n_ctl_pad = (int)ctl_ent.size();
    if ((rc = dev_upload(h, SA, &p.ctl_ent, ctl_ent))) return rc;
    auto &M = h->state_allocs;
    if ((rc = dev_alloc(h, M, &p.ctl_state, (size_t)sg::CS_COUNT * std::max(p.n_ctl_pad, 1)))) return rc;
    if ((rc = dev_alloc(h, M, &p.dyn, nblk * (size_t)p.FROWS * 64))) return rc;
    if ((rc = dev_alloc(h, M, &p.sdyn, (size_t)R))) return rc;
    if ((rc = dev_alloc(h, M, &p.events, (size_t)R * std::max(p.ev_cap, 1)))) return rc;
    if ((rc = dev_alloc(h, M, &p.ev_pose, (size_t)R * std::max(p.ev_cap, 1) * 3))) return rc;
    if ((rc = dev_alloc(h, M, &p.ev_hpose, (size_t)R * std::max(p.ev_cap, 1) * 3, false))) return rc;
    HIP_TRY(h, hipMemsetAsync(p.ev_hpose, 0xFF, (size_t)R * std::max(p.ev_cap, 1) * 3 * sizeof(double), h->stream));
    if ((rc = dev_alloc(h, M, &p.rec_t, (size_t)std::max(p.rec_cap, 1) * R))) return rc;
    if ((rc = dev_alloc(h, M, &p.rec_pose, (size_t)std::max(p.rec_cap, 0) * 6 * R * EP + 1))) return rc;

#ifdef SG_PHASE_TIMERS
    if ((rc = dev_alloc(h, M, &p.phase_cycles, 16 + 4096))) return rc;
#endif
    // stage-1 resample on the device, piece by piece behind the knot copy
    for (int c = 0; c < UP_CHUNKS; ++c) {
        while (issued.load(std::memory_order_acquire) <= c) std::this_thread::yield();
        const int64_t row0 = grid_off[chunk_r[c]], row1 = grid_off[chunk_r[c + 1]];
        if (rows_total <= 0 || row1 <= row0) continue;
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->up_ev[c], 0));
        const int64_t threads = (row1 - row0) * EP;
        sg::build_grid_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, h->stream>>>(p, h->d_row_scen, row0, row1);
        HIP_TRY(h, hipGetLastError());
    }
    copier.join();
    if (copy_err != hipSuccess) return fail(h, SG_ERR_HIP, "sg_upload: copying the knots failed: %s", hipGetErrorString(copy_err));
    HIP_TRY(h, hipStreamSynchronize(h->stream)); // host vectors go out of scope
    stage("knot copy (since the start) + stage-1 resample");
    h->uploaded = true;
    int rc_reset = sg_reset(h);
    stage("reset");
    return rc_reset;
}

extern "C" int sg_reset(sg_handle *h)
{
    if (!h) return SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_reset: no scenarios uploaded");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    int rc;
    if (h->rss_enabled && h->ego_first) {
        bool fresh = false;
        if ((rc = ensure_rss(h, &fresh)) || (rc = ensure_rssq(h))) return rc;
        h->rss_fused = true;
        rc = launch_rollout(h, 0, 1, 0, nullptr);
        h->rss_fused = false;
    } else {
        rc = launch_rollout(h, 0, 1, 0, nullptr);
        if (!rc && h->rss_enabled) rc = sg_rss_update(h, 1);
    }
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SG_OK;
}

extern "C" int sg_reset_scenarios(sg_handle *h, const uint8_t *mask)
{
    if (!h || !mask) return h ? fail(h, SG_ERR_INVALID, "sg_reset_scenarios: null mask") : SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_reset_scenarios: no scenarios uploaded");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!h->d_reset_mask) HIP_TRY(h, hipMalloc((void **)&h->d_reset_mask, (size_t)h->R));
    HIP_TRY(h, hipMemcpyAsync(h->d_reset_mask, mask, (size_t)h->R, hipMemcpyHostToDevice, h->stream));
    h->p.reset_mask = h->d_reset_mask;
    int rc;
    if (h->rss_enabled && h->ego_first && rss_live(h)) { // the flagged scenarios' RSS histories start anew as well
        bool fresh = false;
        if ((rc = ensure_rss(h, &fresh)) || (rc = ensure_rssq(h))) return rc;
        h->rss_fused = true;
        rc = launch_rollout(h, 0, 2, 0, nullptr);
        h->rss_fused = false;
    } else {
        rc = launch_rollout(h, 0, 2, 0, nullptr);
    }
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SG_OK;
}

extern "C" int sg_terminal_flags(sg_handle *h, uint32_t *out, const uint32_t **d_out)
{
    if (!h || (!out && !d_out)) return h ? fail(h, SG_ERR_INVALID, "sg_terminal_flags: no output given") : SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_terminal_flags: no scenarios uploaded");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!h->d_term_flags) HIP_TRY(h, hipMalloc((void **)&h->d_term_flags, (size_t)h->R * sizeof(uint32_t)));
    sg::terminal_flags_kernel<<<dim3((unsigned)h->R), dim3(64), 0, h->stream>>>(h->p, h->cfg.timestep, h->d_term_flags);
    HIP_TRY(h, hipGetLastError());
    if (out) {
        HIP_TRY(h, hipMemcpyAsync(out, h->d_term_flags, (size_t)h->R * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (const int rcq = check_queue(h)) return rcq; // (a persistent launch that gave up: sticky)
    }
    if (d_out) *d_out = h->d_term_flags;
    return SG_OK;
}

extern "C" int sg_set_timestep(sg_handle *h, double timestep)
{
    if (!h || !(timestep > 0.0)) return h ? fail(h, SG_ERR_INVALID, "sg_set_timestep: timestep must be > 0") : SG_ERR_INVALID;
    h->cfg.timestep = timestep;
    ++h->generation;
    return SG_OK;
}

extern "C" int sg_step(sg_handle *h, int32_t n_steps, const double *actions, int32_t actions_device)
{
    if (!h) return SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_step: no scenarios uploaded");
    if (n_steps < 0) return fail(h, SG_ERR_INVALID, "sg_step: n_steps < 0");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const double *d_act = nullptr;
    if (actions && actions_device) {
        d_act = actions;
    } else if (actions) {
        size_t n = (size_t)n_steps * h->R * 2;
        if (n > h->actions_cap) {
            if (h->d_actions) HIP_TRY(h, hipFree(h->d_actions));
            h->d_actions = nullptr;
            HIP_TRY(h, hipMalloc((void **)&h->d_actions, std::max<size_t>(n, 2) * sizeof(double)));
            h->actions_cap = n;
        }
        if (n) HIP_TRY(h, hipMemcpyAsync(h->d_actions, actions, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        d_act = h->d_actions;
    } else {
        // no actions: SG_KIND_AGENT_VEHICLE slots get (0, 0)
        size_t n = (size_t)n_steps * h->R * 2;
        if (n > h->actions_cap) {
            if (h->d_actions) HIP_TRY(h, hipFree(h->d_actions));
            h->d_actions = nullptr;
            HIP_TRY(h, hipMalloc((void **)&h->d_actions, std::max<size_t>(n, 2) * sizeof(double)));
            h->actions_cap = n;
        }
        if (n) HIP_TRY(h, hipMemsetAsync(h->d_actions, 0, n * sizeof(double), h->stream));
        d_act = h->d_actions;
    }
    int rc = SG_OK;
    if (h->rss_enabled && h->ego_first && rss_live(h)) {
        bool fresh = false;
        if (!(rc = ensure_rss(h, &fresh))) rc = ensure_rssq(h);
        h->rss_fused = true;
        if (!rc) rc = launch_rollout(h, n_steps, 0, 1, d_act);
        h->rss_fused = false;
    } else if (h->rss_enabled) {
        for (int k = 0; k < n_steps && !rc; ++k)
            if (!(rc = launch_rollout(h, 1, 0, 1, d_act + (size_t)k * h->R * 2))) rc = sg_rss_update(h, 0);
    } else {
        rc = launch_rollout(h, n_steps, 0, 1, d_act);
    }
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return check_queue(h);
}

// device scratch shared by the observation entry points: a tick of an RL loop calls them once per step, a hipMalloc /
// hipFree pair per call would cost more than the kernels
static int obs_scratch(sg_handle *h, size_t bytes, unsigned char **out)
{
    if (bytes > h->obs_cap) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->obs_buf) HIP_TRY(h, hipFree(h->obs_buf));
        h->obs_buf = nullptr;
        h->obs_cap = 0;
        HIP_TRY(h, hipMalloc(&h->obs_buf, bytes));
        poison(h->stream, h->obs_buf, bytes);
        h->obs_cap = bytes;
        ++h->generation;
    }
    *out = (unsigned char *)h->obs_buf;
    return SG_OK;
}

// One tick of the RL loop (integrations/openaigym.py:171-226) as ONE graph launch: the step with the policy's actions, the
// terminal conditions of the new state, the map observation.  Four short kernels whose launch and synchronisation
// overheads exceed their run time when issued one by one; captured once per (batch, observation geometry) and replayed.
extern "C" int sg_tick(sg_handle *h, const double *actions, int32_t actions_device, double width, double height, int32_t nw,
                       int32_t nh, int32_t n_layers, const int32_t *layers, const uint8_t **d_obs, const uint32_t **d_flags)
{
    if (!h) return SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_tick: no scenarios uploaded");
    if (!layers || n_layers < 1 || n_layers > 8 || nw < 1 || nh < 1 || !(width >= 0.0) || !(height >= 0.0))
        return fail(h, SG_ERR_INVALID, "sg_tick: bad observation geometry (1..8 layers)");
    if (h->n_ext > 0) return fail(h, SG_ERR_STATE, "sg_tick: batches with caller-run agents are driven through sg_set_external_poses + sg_step");
    bool any_surface = false;
    for (int k = 0; k < n_layers; ++k) {
        const uint32_t L = (uint32_t)layers[k];
        if (layers[k] < 0 || L > 255u || (L & (L - 1))) return fail(h, SG_ERR_INVALID, "sg_tick: layers[%d]=%d is not 0 or one SG_LAYER_* bit", k, layers[k]);
        any_surface = any_surface || L != 0;
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    // fixed device addresses for everything the graph's kernels read or write
    const size_t n_act = (size_t)h->R * 2;
    if (n_act > h->actions_cap) {
        if (h->d_actions) HIP_TRY(h, hipFree(h->d_actions));
        h->d_actions = nullptr;
        HIP_TRY(h, hipMalloc((void **)&h->d_actions, std::max<size_t>(n_act, 2) * sizeof(double)));
        h->actions_cap = n_act;
        ++h->generation;
    }
    if (!h->d_term_flags) { HIP_TRY(h, hipMalloc((void **)&h->d_term_flags, (size_t)h->R * sizeof(uint32_t))); ++h->generation; }
    const size_t plane = (size_t)nw * nh, bytes = (size_t)h->R * n_layers * plane, lay_off = (bytes + 15) & ~(size_t)15;
    unsigned char *d = nullptr;
    int rc = obs_scratch(h, lay_off + 8 * sizeof(int32_t), &d);
    if (rc) return rc;
    int32_t *dl = reinterpret_cast<int32_t *>(d + lay_off);
    // sg_set_rss: the callback runs after the step, inside the captured launch (like sg_step; without records of a reset --
    // the callback was switched on after sg_upload -- through sg_rss_update after the graph)
    if (h->wide && (rc = ensure_wide(h))) return rc;
    const bool rss_tick = h->rss_enabled && h->ego_first && rss_live(h);
    if (rss_tick) { // (allocations stay outside the capture)
        bool fresh = false;
        if ((rc = ensure_rss(h, &fresh)) || (rc = ensure_rssq(h))) return rc;
    }
    const bool same = h->tick_exec && h->tick_gen == h->generation && h->tick_w == width && h->tick_h == height &&
                      h->tick_rss == rss_tick && h->tick_nw == nw && h->tick_nh == nh && h->tick_nl == n_layers &&
                      std::equal(layers, layers + n_layers, h->tick_layers);
    if (!same) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (const int rcq = check_queue(h)) return rcq; // (a persistent launch that gave up: sticky)
        if (h->tick_exec) { HIP_TRY(h, hipGraphExecDestroy(h->tick_exec)); h->tick_exec = nullptr; }
        HIP_TRY(h, hipMemcpy(dl, layers, (size_t)n_layers * sizeof(int32_t), hipMemcpyHostToDevice));
        hipGraph_t graph = nullptr;
        HIP_TRY(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        size_t ev_next = 0;
        h->timing_now = false;
        h->n_launches = 0;
        h->launch_ev.clear();
        h->rss_fused = rss_tick && !unfused_rss(h); // (as launch_rollout_impl: the callback as a launch of its own where no fused variant exists)
        rc = h->wide ? launch_wide(h, 1, 0, 1, h->d_actions) : launch_main(h, 1, 0, 1, h->d_actions, nullptr, false, &ev_next);
        if (!rc && unfused_off_road(h)) // (launch_rollout_impl: the same launches behind the step)
            sg::ego_off_road_kernel<<<dim3((unsigned)((h->R + 63) / 64)), dim3(64), 0, h->stream>>>(h->p);
        if (!rc && unfused_rss(h))
            sg::rss_kernel<<<dim3((unsigned)h->R), dim3(512), 0, h->stream>>>(h->p, 0, h->d_rss_state, h->d_rss_code, h->d_rss_safe, h->d_rss_seen);
        h->rss_fused = false;
        hipError_t e = hipSuccess;
        if (!rc && h->wide) {
            // scenarios of more than 512 entities: the entity layers tile by tile (raster_kernel), empty surfaces (no road
            // networks at this width), the terminal conditions by the kernel of sg_terminal_flags
            if (any_surface && !h->has_road) e = hipMemsetAsync(d, 0, bytes, h->stream);
            for (int k = 0; k < n_layers && e == hipSuccess; ++k)
                if (layers[k] == 0) {
                    sg::raster_kernel<<<dim3((unsigned)h->R), dim3(512), 0, h->stream>>>(h->p, width, height, nw, nh, d + (size_t)k * plane, (int64_t)(n_layers * plane));
                    e = hipGetLastError();
                }
            if (any_surface && h->has_road && e == hipSuccess) {
                sg::raster_surface_kernel<<<dim3((unsigned)h->R), dim3(256), 0, h->stream>>>(h->p, h->road, width, height, nw, nh, n_layers, dl, d);
                e = hipGetLastError();
            }
            if (e == hipSuccess) {
                sg::terminal_flags_kernel<<<dim3((unsigned)h->R), dim3(64), 0, h->stream>>>(h->p, h->cfg.timestep, h->d_term_flags);
                e = hipGetLastError();
            }
        } else if (!rc) { // the whole observation (map layers + terminal flags) in one launch
            sg::observe_kernel<<<dim3((unsigned)h->R), dim3(h->EP > 256 ? 512 : 256), 0, h->stream>>>(h->p, h->road, h->has_road ? 1 : 0, width, height, nw,
                                                                                nh, n_layers, dl, d, h->d_term_flags);
            e = hipGetLastError();
        }
        hipError_t e2 = hipStreamEndCapture(h->stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || e2 != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            return fail(h, SG_ERR_HIP, "sg_tick: capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        }
        e = hipGraphInstantiate(&h->tick_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { h->tick_exec = nullptr; return fail(h, SG_ERR_HIP, "sg_tick: hipGraphInstantiate: %s", hipGetErrorString(e)); }
        h->tick_gen = h->generation;
        h->tick_rss = rss_tick;
        h->tick_w = width; h->tick_h = height; h->tick_nw = nw; h->tick_nh = nh; h->tick_nl = n_layers;
        std::copy(layers, layers + n_layers, h->tick_layers);
    }
    if (actions)
        HIP_TRY(h, hipMemcpyAsync(h->d_actions, actions, n_act * sizeof(double), actions_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    else
        HIP_TRY(h, hipMemsetAsync(h->d_actions, 0, n_act * sizeof(double), h->stream));
    HIP_TRY(h, hipGraphLaunch(h->tick_exec, h->stream));
    if (h->rss_enabled && !rss_tick && (rc = sg_rss_update(h, 0))) return rc;
    h->timed = false;
    if (d_obs) *d_obs = d;
    if (d_flags) *d_flags = h->d_term_flags;
    return SG_OK;
}

extern "C" int sg_set_external_poses(sg_handle *h, const double *poses)
{
    if (!h || !poses) return SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_set_external_poses: no scenarios uploaded");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t row = (size_t)h->E * 6 * sizeof(double);
    if (h->EP == h->E) {
        HIP_TRY(h, hipMemcpyAsync(h->d_ext, poses, (size_t)h->R * row, hipMemcpyHostToDevice, h->stream));
    } else { // padded entity stride on the device
        HIP_TRY(h, hipMemcpy2DAsync(h->d_ext, (size_t)h->EP * 6 * sizeof(double), poses, row, row, (size_t)h->R,
                                    hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream)); // the caller's buffer is free on return
    return SG_OK;
}

extern "C" int sg_rollout_async(sg_handle *h, int32_t max_steps, int32_t do_reset)
{
    if (!h) return SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_rollout: no scenarios uploaded");
    if (h->n_ext > 0 && max_steps > 0)
        return fail(h, SG_ERR_STATE, "sg_rollout: %d slots are driven by the caller's agents (SG_KIND_AGENT_EXTERNAL): "
                                     "use sg_set_external_poses + sg_step tick by tick", h->n_ext);
    if (max_steps < 0) return fail(h, SG_ERR_INVALID, "sg_rollout: max_steps < 0");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->rss_enabled && h->ego_first) {
        // the callback inside the rollout kernel (rollout_kernel_rss): after the reset and after every step of ONE launch
        bool fresh = false;
        int rc = ensure_rss(h, &fresh);
        if (!rc) rc = ensure_rssq(h);
        if (rc) return rc;
        h->rss_fused = true;
        rc = launch_rollout(h, max_steps, do_reset || fresh ? 1 : 0, 0, nullptr);
        h->rss_fused = false;
        return rc;
    }
    if (h->rss_enabled) { // (sg_rss_update reports why not: the ego is not entity 0) one step per launch
        int rc = SG_OK;
        if (do_reset && ((rc = launch_rollout(h, 0, 1, 0, nullptr)) || (rc = sg_rss_update(h, 1)))) return rc;
        for (int k = 0; k < max_steps; ++k)
            if ((rc = launch_rollout(h, 1, 0, 0, nullptr)) || (rc = sg_rss_update(h, 0))) return rc;
        return SG_OK;
    }
    if (do_reset && slicing_pays(h, max_steps)) {
        const int rc = launch_sliced(h, max_steps);
        if (rc != SG_SLICE_FALLBACK) return rc;
    }
    // external-action slots are fed (0, 0) here; drive them with sg_step(actions)
    return launch_rollout(h, max_steps, do_reset ? 1 : 0, 0, nullptr);
}

extern "C" int sg_rollout(sg_handle *h, int32_t max_steps)
{
    int rc = sg_rollout_async(h, max_steps, 1);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return check_queue(h);
}

extern "C" int sg_synchronize(sg_handle *h)
{
    if (!h) return SG_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return check_queue(h);
}

extern "C" void *sg_stream(sg_handle *h) { return h ? (void *)h->stream : nullptr; }

extern "C" int sg_state_view_get(sg_handle *h, sg_state_view *out)
{
    if (!h || !out) return SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_state_view_get: no scenarios uploaded");
    out->n_scenarios = h->R; out->n_entities = h->E; out->entity_stride = h->EP;
    out->n_blocks = (int32_t)(h->NE / 64);
    out->row_words = h->WV;
    out->block_rows = h->p.FROWS;
    out->blocks = h->p.dyn;
    out->scen = h->p.sdyn;
    return SG_OK;
}

extern "C" int sg_read_metrics(sg_handle *h, sg_metrics *out, sg_event *events, int32_t cap, int32_t *n_events)
{
    if (!h || !out) return SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_read_metrics: no scenarios uploaded");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (events && h->p.ev_cap > 0) { // CollisionMetric.record_collision for the Vehicle hazards recorded since the last read
        sg::classify_events_kernel<<<dim3((unsigned)h->R), dim3(64), 0, h->stream>>>(h->p, h->c_tol);
        HIP_TRY(h, hipGetLastError());
    }
    const int R = h->R;
    const Params &p = h->p;
    auto pinned = [&](void **buf, size_t *cap, size_t bytes) -> int { // grown on demand, kept on the handle
        if (*cap >= bytes) return SG_OK;
        if (*buf) HIP_TRY(h, hipHostFree(*buf));
        *buf = nullptr;
        *cap = 0;
        HIP_TRY(h, hipHostMalloc(buf, bytes, hipHostMallocDefault));
        *cap = bytes;
        return SG_OK;
    };
    int rc0 = pinned(&h->pin_sd, &h->pin_sd_cap, (size_t)R * sizeof(sg_scenario_state));
    if (rc0) return rc0;
    sg_scenario_state *sd = static_cast<sg_scenario_state *>(h->pin_sd);
    HIP_TRY(h, hipMemcpyAsync(sd, p.sdyn, (size_t)R * sizeof(sg_scenario_state), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if ((rc0 = check_queue(h))) return rc0;
    int64_t total = 0;
    bool overflow = false;
    for (int r = 0; r < R; ++r) {
        out[r].ego_avg_speed = sd[r].ego_avg_speed; out[r].ego_max_speed = sd[r].ego_max_speed;
        out[r].ego_distance_travelled = sd[r].ego_distance_travelled;
        out[r].final_t = sd[r].t; out[r].n_steps = sd[r].n_steps; out[r].done = sd[r].done;
        out[r].n_collisions = sd[r].n_events;
        out[r].reserved = 0;
        if (h->noise_mode == SG_NOISE_STREAM && h->has_ped && sd[r].noise_pos > h->noise_len)
            return fail(h, SG_ERR_CAPACITY, "sg_read_metrics: scenario %d needed %lld noise variates, the stream of sg_set_ped_noise holds %lld",
                        r, (long long)sd[r].noise_pos, (long long)h->noise_len);
        if (sd[r].n_events > p.ev_cap) overflow = true;
        total += std::min(sd[r].n_events, p.ev_cap);
    }
    if (n_events) *n_events = (int32_t)total;
    if (events && cap > 0 && p.ev_cap > 0 && total > 0) {
        int width = 0; // only the columns in use travel over PCIe
        for (int r = 0; r < R; ++r) width = std::max(width, std::min(sd[r].n_events, p.ev_cap));
        if ((rc0 = pinned(&h->pin_ev, &h->pin_ev_cap, (size_t)R * width * sizeof(sg_event)))) return rc0;
        sg_event *all = static_cast<sg_event *>(h->pin_ev);
        HIP_TRY(h, hipMemcpy2DAsync(all, (size_t)width * sizeof(sg_event), p.events, (size_t)p.ev_cap * sizeof(sg_event),
                                    (size_t)width * sizeof(sg_event), (size_t)R, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        int64_t k = 0;
        for (int r = 0; r < R; ++r)
            for (int i = 0; i < std::min(sd[r].n_events, p.ev_cap); ++i) {
                if (k >= cap) return fail(h, SG_ERR_CAPACITY, "sg_read_metrics: %lld events do not fit cap=%d", (long long)total, cap);
                events[k++] = all[(size_t)r * width + i];
            }
    }
    // more than event_capacity events in one scenario: the count (n_collisions) is exact, the table keeps the first ones
    (void)overflow;
    return SG_OK;
}

// CollisionPointMetric.get_state (metrics/collision.py:217-253) for the events sg_read_metrics lists, in its order
extern "C" int sg_read_collision_points(sg_handle *h, double *out, int32_t cap, int32_t *n_events)
{
    if (!h || !out) return h ? fail(h, SG_ERR_INVALID, "sg_read_collision_points: null argument") : SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_read_collision_points: no scenarios uploaded");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const int R = h->R;
    const Params &p = h->p;
    if (p.ev_cap > 0) {
        sg::classify_events_kernel<<<dim3((unsigned)R), dim3(64), 0, h->stream>>>(p, h->c_tol);
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (const int rcq = check_queue(h)) return rcq; // (a persistent launch that gave up: sticky)
    std::vector<sg_scenario_state> sd(R);
    HIP_TRY(h, hipMemcpy(sd.data(), p.sdyn, (size_t)R * sizeof(sg_scenario_state), hipMemcpyDeviceToHost));
    std::vector<double> all((size_t)R * std::max(p.ev_cap, 1) * 3);
    HIP_TRY(h, hipMemcpy(all.data(), p.ev_pose, all.size() * sizeof(double), hipMemcpyDeviceToHost));
    int64_t k = 0;
    for (int r = 0; r < R; ++r)
        for (int i = 0; i < std::min(sd[r].n_events, p.ev_cap); ++i, ++k) {
            if (k >= cap) return fail(h, SG_ERR_CAPACITY, "sg_read_collision_points: more events than cap=%d", cap);
            for (int c = 0; c < 3; ++c) out[k * 3 + c] = all[((size_t)r * p.ev_cap + i) * 3 + c];
        }
    if (n_events) *n_events = (int32_t)k;
    return SG_OK;
}

extern "C" int sg_read_record(sg_handle *h, int32_t n_rows, double *t_out, double *pose_out)
{
    if (!h || n_rows < 0) return SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_read_record: no scenarios uploaded");
    const Params &p = h->p;
    if (n_rows > p.rec_cap) return fail(h, SG_ERR_CAPACITY, "sg_read_record: n_rows=%d > record_capacity=%d", n_rows, p.rec_cap);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (const int rcq = check_queue(h)) return rcq; // (a persistent launch that gave up: sticky)
    const int R = h->R, E = h->E, EP = h->EP;
    if (t_out && n_rows) HIP_TRY(h, hipMemcpy(t_out, p.rec_t, (size_t)n_rows * R * 8, hipMemcpyDeviceToHost));
    if (pose_out && n_rows) {
        std::vector<double> raw((size_t)n_rows * 6 * R * EP);
        HIP_TRY(h, hipMemcpy(raw.data(), p.rec_pose, raw.size() * 8, hipMemcpyDeviceToHost));
        for (int s = 0; s < n_rows; ++s)
            for (int r = 0; r < R; ++r)
                for (int e = 0; e < E; ++e)
                    for (int c = 0; c < 6; ++c)
                        pose_out[(((size_t)s * R + r) * E + e) * 6 + c] = raw[((size_t)s * 6 + c) * R * EP + (size_t)r * EP + e];
    }
    return SG_OK;
}

extern "C" int sg_copy_to_host(sg_handle *h, const void *device_ptr, void *host_ptr, uint64_t bytes)
{
    if (!h || !device_ptr || !host_ptr) return SG_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(host_ptr, device_ptr, bytes, hipMemcpyDeviceToHost));
    return check_queue(h);
}

extern "C" int sg_last_kernel_ms(sg_handle *h, float *ms)
{
    if (!h || !ms) return SG_ERR_INVALID;
    if (!h->timed) return fail(h, SG_ERR_STATE, "sg_last_kernel_ms: the last call was not timed (nothing launched yet, or fewer than 16 steps)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    HIP_TRY(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return SG_OK;
}

// (start, end) of the hot-path launches of the last timed call, ms after the call's first event
static int launch_intervals(sg_handle *h, std::vector<std::pair<float, float>> &iv)
{
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    iv.clear();
    for (int i : h->launch_ev) {
        float a = 0.0f, d = 0.0f;
        HIP_TRY(h, hipEventElapsedTime(&a, h->ev0, h->ev_pool[i]));
        HIP_TRY(h, hipEventElapsedTime(&d, h->ev_pool[i], h->ev_pool[i + 1]));
        iv.emplace_back(a, a + d);
    }
    return SG_OK;
}

extern "C" int sg_last_launch_stats(sg_handle *h, int32_t *n_launches, float *kernel_ms_total)
{
    if (!h || !n_launches || !kernel_ms_total) return SG_ERR_INVALID;
    if (!h->timed) return fail(h, SG_ERR_STATE, "sg_last_launch_stats: the last call was not timed (nothing launched yet, or fewer than 16 steps)");
    std::vector<std::pair<float, float>> iv;
    int rc = launch_intervals(h, iv);
    if (rc) return rc;
    // the union of the launches' intervals: launches of the two pipelines overlap (launch_rollout), time counts once
    std::sort(iv.begin(), iv.end());
    float total = 0.0f, lo = 0.0f, hi = -1.0f;
    for (const auto &x : iv) {
        if (hi < lo || x.first > hi) {
            if (hi >= lo) total += hi - lo;
            lo = x.first;
            hi = x.second;
        } else {
            hi = std::max(hi, x.second);
        }
    }
    if (hi >= lo) total += hi - lo;
    *n_launches = h->n_launches;
    *kernel_ms_total = total;
    return SG_OK;
}

extern "C" const char *sg_last_kernel(sg_handle *h) { return h ? h->last_kernel : ""; }

extern "C" int sg_schedule_info(sg_handle *h, int32_t *info)
{
    if (!h || !info) return SG_ERR_INVALID;
    info[0] = h->last_schedule;
    info[1] = h->last_schedule == 2 ? h->last_chunks : 0;
    info[2] = h->last_schedule == 2 ? h->last_ring : 0;
    info[3] = h->last_schedule == 2 ? h->last_grid : 0;
    info[4] = h->p.n_ctl_pad / 64;
    info[5] = (int32_t)std::min<size_t>(0x7fffffff, h->NE / 64);
    info[6] = h->n_simd;
    info[7] = h->n_launches;
    return SG_OK;
}

extern "C" int sg_last_launch_gross_ms(sg_handle *h, float *kernel_ms_gross)
{
    if (!h || !kernel_ms_gross) return SG_ERR_INVALID;
    if (!h->timed) return fail(h, SG_ERR_STATE, "sg_last_launch_gross_ms: the last call was not timed (nothing launched yet, or fewer than 16 steps)");
    std::vector<std::pair<float, float>> iv;
    int rc = launch_intervals(h, iv);
    if (rc) return rc;
    float total = 0.0f;
    for (const auto &x : iv) total += x.second - x.first;
    *kernel_ms_gross = total;
    return SG_OK;
}

extern "C" int sg_debug_trig32(sg_handle *h, int64_t n, const double *heading, float *sin_out, float *cos_out)
{
    if (!h || n < 0 || !heading || !sin_out || !cos_out) return SG_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    double *d_h = nullptr;
    float *d_s = nullptr, *d_c = nullptr;
    const size_t m = (size_t)std::max<int64_t>(n, 1);
    HIP_TRY(h, hipMalloc((void **)&d_h, m * sizeof(double)));
    HIP_TRY(h, hipMalloc((void **)&d_s, m * sizeof(float)));
    HIP_TRY(h, hipMalloc((void **)&d_c, m * sizeof(float)));
    int rc = SG_OK;
    do {
        if (hipMemcpy(d_h, heading, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = SG_ERR_HIP; break; }
        if (n > 0) sg::trig32_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream>>>(d_h, d_s, d_c, n);
        if (hipStreamSynchronize(h->stream) != hipSuccess) { rc = SG_ERR_HIP; break; }
        if (hipMemcpy(sin_out, d_s, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { rc = SG_ERR_HIP; break; }
        if (hipMemcpy(cos_out, d_c, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { rc = SG_ERR_HIP; break; }
    } while (0);
    (void)hipFree(d_h); (void)hipFree(d_s); (void)hipFree(d_c);
    if (rc) return fail(h, rc, "sg_debug_trig32: HIP copy/launch failed");
    return SG_OK;
}

extern "C" int sg_host_alloc(int32_t device, uint64_t bytes, void **out)
{
    if (!out || bytes == 0) return SG_ERR_INVALID;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return SG_ERR_HIP;
    return hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault) == hipSuccess ? SG_OK : SG_ERR_HIP;
}

extern "C" int sg_host_free(void *p)
{
    if (!p) return SG_OK;
    return hipHostFree(p) == hipSuccess ? SG_OK : SG_ERR_HIP;
}

extern "C" int sg_set_slicing(sg_handle *h, int32_t mode)
{
    if (!h || mode < 0 || mode > 2) return h ? fail(h, SG_ERR_INVALID, "sg_set_slicing: mode 0, 1 or 2") : SG_ERR_INVALID;
    h->slice_mode = mode;
    return SG_OK;
}

extern "C" int sg_set_tuning(sg_handle *h, int32_t tab_min_steps, int32_t chunk_steps, int32_t overlap)
{
    if (!h) return SG_ERR_INVALID;
    if (tab_min_steps >= 0) h->tab_min = tab_min_steps;
    if (chunk_steps > 0) h->chunk_steps = chunk_steps;
    if (overlap >= 0) h->overlap = overlap != 0;
    ++h->generation;
    return SG_OK;
}

// the RSS record arrays of the current batch (freed by sg_upload); returns 1 in *fresh when they were just created
static int ensure_rss(sg_handle *h, bool *fresh)
{
    *fresh = false;
    if (h->d_rss_state && h->rss_NE != h->NE) { // (another padded entity count: sg_upload of a pedestrian batch narrower than 16)
        (void)hipFree(h->d_rss_state); (void)hipFree(h->d_rss_code); (void)hipFree(h->d_rss_safe); (void)hipFree(h->d_rss_seen);
        h->d_rss_state = nullptr; h->d_rss_code = nullptr; h->d_rss_safe = nullptr; h->d_rss_seen = nullptr;
    }
    if (!h->d_rss_state) {
        HIP_TRY(h, hipMalloc((void **)&h->d_rss_state, h->NE * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc((void **)&h->d_rss_code, h->NE * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc((void **)&h->d_rss_safe, h->NE * 2 * sizeof(double)));
        HIP_TRY(h, hipMalloc((void **)&h->d_rss_seen, (size_t)h->R * sizeof(int32_t)));
        h->rss_NE = h->NE;
        h->rss_stale = true;
    }
    if (h->rss_stale) { // first use after sg_create / sg_upload: the records of a new batch
        HIP_TRY(h, hipMemsetAsync(h->d_rss_state, 0, h->NE * sizeof(int32_t), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_rss_code, 0xff, h->NE * sizeof(int32_t), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_rss_safe, 0xff, h->NE * 2 * sizeof(double), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_rss_seen, 0xff, (size_t)h->R * sizeof(int32_t), h->stream));
        h->rss_stale = false;
        *fresh = true;
    }
    h->p.rss_state = h->d_rss_state; h->p.rss_code = h->d_rss_code; h->p.rss_safe = h->d_rss_safe; h->p.rss_seen = h->d_rss_seen;
    return SG_OK;
}

static int ensure_rssq(sg_handle *h)
{
    if (h->wide) return SG_OK; // (scenarios of more than 512 entities run the callback as a launch of its own: no line-test queue)
    if (h->d_rssq && h->rssq_NE != h->NE) {
        (void)hipFree(h->d_rssq); (void)hipFree(h->d_rssq_n);
        h->d_rssq = nullptr; h->d_rssq_n = nullptr;
    }
    if (!h->d_rssq) {
        h->rssq_NE = h->NE;
        const size_t nw = h->NE / 64, per_step = nw * 64 * sg::RSSQ_REC * sizeof(double);
        // (SG_RSSQ_MB; default: an eighth of the free device memory, at least 4 GiB -- every launch boundary costs the tail of a
        // launch, and 288 GB hold the queues of a whole 1000-step rollout of the largest batches)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        const int mb = env_int("SG_RSSQ_MB", 0);
        const size_t budget = mb > 0 ? (size_t)mb << 20 : std::max<size_t>((size_t)4096 << 20, free_b / 8);
        const int cap = std::max(1, env_int("SG_RSSQ_STEPS", 1024));
        int steps = (int)std::min<size_t>((size_t)cap, std::max<size_t>(2, budget / per_step) - 1);
        // the queue only sets how many steps one launch covers: when the device is short of memory, shorter launches
        //  instead of a failed sg_upload / sg_reset
        for (;; steps = std::max(1, steps / 2)) {
            const hipError_t e = hipMalloc((void **)&h->d_rssq, per_step * (size_t)(steps + 1));
            if (e == hipSuccess) break;
            (void)hipGetLastError();
            h->d_rssq = nullptr;
            if (steps == 1) return fail(h, SG_ERR_HIP, "sg_set_rss: no device memory for the line-test queue (%zu bytes per step)", per_step);
        }
        HIP_TRY(h, hipMalloc((void **)&h->d_rssq_n, nw * sizeof(int32_t)));
        HIP_TRY(h, hipMemsetAsync(h->d_rssq_n, 0, nw * sizeof(int32_t), h->stream));
        h->rssq_steps = steps;
    }
    h->p.rssq = h->d_rssq; h->p.rssq_n = h->d_rssq_n; h->p.rssq_cap = (h->rssq_steps + 1) * 64;
    return SG_OK;
}

extern "C" int sg_rss_update(sg_handle *h, int32_t reset)
{
    if (!h) return SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_rss_update: no scenarios uploaded");

    if (!h->ego_first) return fail(h, SG_ERR_STATE, "sg_rss_update: RSSDistances keeps its records for entities[1:], the ego has to be entity 0 of every scenario");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    bool fresh = false;
    int rc0 = ensure_rss(h, &fresh);
    if (rc0) return rc0;
    if (fresh) reset = 1;
    sg::rss_kernel<<<dim3((unsigned)h->R), dim3(h->EP > 256 ? 512 : 256), 0, h->stream>>>(h->p, reset ? 1 : 0, h->d_rss_state, h->d_rss_code, h->d_rss_safe, h->d_rss_seen);
    HIP_TRY(h, hipGetLastError());
    return SG_OK;
}

extern "C" int sg_set_rss(sg_handle *h, int32_t enabled)
{
    if (!h) return SG_ERR_INVALID;
    h->rss_enabled = enabled != 0;
    return SG_OK;
}

extern "C" int sg_rss_read(sg_handle *h, uint8_t *flags, int32_t *codes, double *safe)
{
    if (!h) return SG_ERR_INVALID;
    if (!h->uploaded || !rss_live(h)) return fail(h, SG_ERR_STATE, "sg_rss_read: sg_rss_update has not run on this batch");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (const int rcq = check_queue(h)) return rcq; // (a persistent launch that gave up: sticky)
    const int R = h->R, E = h->E, EP = h->EP;
    std::vector<int32_t> st(h->NE), cd(h->NE);
    std::vector<double> sf(h->NE * 2);
    HIP_TRY(h, hipMemcpy(st.data(), h->d_rss_state, h->NE * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (codes) HIP_TRY(h, hipMemcpy(cd.data(), h->d_rss_code, h->NE * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (safe) HIP_TRY(h, hipMemcpy(sf.data(), h->d_rss_safe, h->NE * 2 * sizeof(double), hipMemcpyDeviceToHost));
    for (int r = 0; r < R; ++r) {
        uint8_t f = 3;
        for (int e = 0; e < E; ++e) {
            const size_t i = (size_t)r * EP + e, o = (size_t)r * E + e;
            if ((st[i] & 0xff) == 2) f &= ~1u; // some entity's history holds "unsafe_longitudinal", rss.py:70-86
            if ((st[i] & 0xff) == 1) f &= ~2u;
            if (codes) codes[o] = cd[i];
            if (safe) { safe[o * 2] = sf[i * 2]; safe[o * 2 + 1] = sf[i * 2 + 1]; }
        }
        if (flags) flags[r] = f;
    }
    return SG_OK;
}

extern "C" int sg_set_collision_tolerance(sg_handle *h, double c_tol)
{
    if (!h || !(c_tol >= 0.0)) return h ? fail(h, SG_ERR_INVALID, "sg_set_collision_tolerance: c_tol must be >= 0") : SG_ERR_INVALID;
    h->c_tol = c_tol;
    return SG_OK;
}

extern "C" int sg_future_collision(sg_handle *h, double horizon, int32_t n_samples, uint8_t *out)
{
    if (!h || !out || n_samples < 1 || !(horizon >= 0.0)) return h ? fail(h, SG_ERR_INVALID, "sg_future_collision: bad argument") : SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_future_collision: no scenarios uploaded");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    unsigned char *d = nullptr;
    int rc = obs_scratch(h, (size_t)h->R, &d);
    if (rc) return rc;
    sg::future_kernel<<<dim3((unsigned)h->R), dim3(256), 0, h->stream>>>(h->p, horizon, n_samples, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, (size_t)h->R, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, SG_ERR_HIP, "sg_future_collision: %s", hipGetErrorString(e));
    return check_queue(h); // (a persistent launch that gave up: sticky)
}

// ---- road surfaces -------------------------------------------------------------------------------
// Index of one network: a uniform grid; per polygon the cells its edges touch (boxes grown by a margin far above the
// rounding of the device's cell lookup) become candidates of that polygon, the other cells of its bounding box are
// wholly inside or wholly outside -- decided with the exact test at the cell centre, once per run of untouched cells.
namespace {
struct RoadBuild {
    std::vector<sg::RoadNet> nets;
    std::vector<uint16_t> cells;
    std::vector<uint32_t> cell_off;
    std::vector<sg::RoadCand> cand;
    std::vector<int32_t> cand_edges;
    std::vector<double> edges;
    std::vector<int64_t> poly_edge_off;
    std::vector<uint32_t> poly_layers;
    std::vector<uint32_t> net_flags;   // bit 0: walkable surface has area, bit 1: impenetrable surface has area
    std::vector<int64_t> imp_off;      // per network: range of imp_edges
    std::vector<double> imp_edges;     // the ring edges of the impenetrable polygons, polygon by polygon
};

int build_road_network(const sg_road_networks *in, int n, RoadBuild &B)
{
    const int64_t q0 = in->poly_off[n], q1 = in->poly_off[n + 1];
    double lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
    uint32_t flags = 0;
    for (int64_t q = q0; q < q1; ++q) {
        for (int64_t r = in->ring_off[q]; r < in->ring_off[q + 1]; ++r) {
            const int64_t a = in->vert_off[r], b = in->vert_off[r + 1];
            for (int64_t i = a; i < b; ++i) {
                const int64_t j = i + 1 < b ? i + 1 : a;
                const double *v = in->verts + 2 * i, *w = in->verts + 2 * j;
                B.edges.insert(B.edges.end(), {v[0], v[1], w[0], w[1]});
                for (int c = 0; c < 2; ++c) { lo[c] = std::min(lo[c], v[c]); hi[c] = std::max(hi[c], v[c]); }
            }
        }
        B.poly_edge_off.push_back((int64_t)B.edges.size() / 4);
        B.poly_layers.push_back(in->layers[q]);
        {   // `surface.area > 0` (social_force.py:87, 97) and the edge list the nearest-point search walks
            const int64_t e0 = B.poly_edge_off[B.poly_edge_off.size() - 2], e1 = B.poly_edge_off.back();
            double a2 = 0.0;
            for (int64_t i = e0; i < e1; ++i) a2 += B.edges[4 * i] * B.edges[4 * i + 3] - B.edges[4 * i + 2] * B.edges[4 * i + 1];
            if (a2 != 0.0 && (in->layers[q] & SG_LAYER_WALKABLE)) flags |= 1u;
            if (a2 != 0.0 && (in->layers[q] & SG_LAYER_IMPENETRABLE)) flags |= 2u;
            if (in->layers[q] & SG_LAYER_IMPENETRABLE) B.imp_edges.insert(B.imp_edges.end(), B.edges.begin() + 4 * e0, B.edges.begin() + 4 * e1);
        }
    }
    B.net_flags.push_back(flags);
    B.imp_off.push_back((int64_t)B.imp_edges.size() / 4);
    sg::RoadNet N{};
    N.cell_base = (int64_t)B.cells.size();
    if (!(lo[0] <= hi[0])) { // no geometry: an empty 1 x 1 grid
        N.x0 = N.y0 = 0.0; N.inv_cell = 1.0; N.nx = N.ny = 1;
        B.nets.push_back(N);
        B.cells.push_back(0);
        B.cell_off.push_back((uint32_t)B.cand.size());
        return 0;
    }
    double c = 1.0; // cell side: 1 m unless the network is so large that this would take more than 2^21 cells
    while (((hi[0] - lo[0]) / c + 4) * ((hi[1] - lo[1]) / c + 4) > 2097152.0) c *= 2;
    const double eps = 1e-6; // >> rounding of (p - x0) * inv_cell for coordinates below 1e9 cells
    N.x0 = std::floor(lo[0] / c) * c - c;
    N.y0 = std::floor(lo[1] / c) * c - c;
    N.inv_cell = 1.0 / c;
    N.nx = (int32_t)std::ceil((hi[0] - N.x0) / c) + 2;
    N.ny = (int32_t)std::ceil((hi[1] - N.y0) / c) + 2;
    const size_t ncell = (size_t)N.nx * N.ny;
    B.cells.resize((size_t)N.cell_base + ncell, 0);
    uint16_t *cells = B.cells.data() + N.cell_base;
    struct Entry { uint32_t cell; sg::RoadCand cd; };
    std::vector<Entry> entries; // candidates of this network, sorted by cell below
    auto cix = [&](double x, double x0, int nmax) { return std::max(0, std::min(nmax - 1, (int)std::floor((x - x0) / c))); };
    std::vector<uint8_t> touched;
    std::vector<std::pair<uint32_t, int32_t>> hits; // (local cell, edge) of one polygon
    const int64_t gq0 = (int64_t)B.poly_layers.size() - (q1 - q0);
    for (int64_t q = q0; q < q1; ++q) {
        const int64_t gq = gq0 + (q - q0);
        const int64_t e0 = B.poly_edge_off[gq], e1 = B.poly_edge_off[gq + 1];
        const uint32_t L = B.poly_layers[gq] & 0xffu;
        if (e1 <= e0 || !L) continue;
        double plo[2] = {INFINITY, INFINITY}, phi[2] = {-INFINITY, -INFINITY};
        for (int64_t i = e0; i < e1; ++i)
            for (int c2 = 0; c2 < 2; ++c2) { plo[c2] = std::min(plo[c2], B.edges[4 * i + c2]); phi[c2] = std::max(phi[c2], B.edges[4 * i + c2]); }
        const int ix0 = cix(plo[0] - eps, N.x0, N.nx), ix1 = cix(phi[0] + eps, N.x0, N.nx);
        const int iy0 = cix(plo[1] - eps, N.y0, N.ny), iy1 = cix(phi[1] + eps, N.y0, N.ny);
        const int w = ix1 - ix0 + 1, hgt = iy1 - iy0 + 1;
        touched.assign((size_t)w * hgt, 0);
        hits.clear();
        for (int64_t i = e0; i < e1; ++i) {
            const double ax = B.edges[4 * i], ay = B.edges[4 * i + 1], bx = B.edges[4 * i + 2], by = B.edges[4 * i + 3];
            const int jx0 = cix(std::min(ax, bx) - eps, N.x0, N.nx), jx1 = cix(std::max(ax, bx) + eps, N.x0, N.nx);
            const int jy0 = cix(std::min(ay, by) - eps, N.y0, N.ny), jy1 = cix(std::max(ay, by) + eps, N.y0, N.ny);
            for (int iy = jy0; iy <= jy1; ++iy)
                for (int ix = jx0; ix <= jx1; ++ix) {
                    // the grown cell box and the segment overlap in x and in y (by the ranges above); they are disjoint
                    // iff the box lies strictly on one side of the segment's line
                    const double bx0 = N.x0 + ix * c - eps, bx1 = N.x0 + (ix + 1) * c + eps;
                    const double by0 = N.y0 + iy * c - eps, by1 = N.y0 + (iy + 1) * c + eps;
                    const double dx = bx - ax, dy = by - ay;
                    const double d0 = dx * (by0 - ay) - dy * (bx0 - ax), d1 = dx * (by0 - ay) - dy * (bx1 - ax);
                    const double d2 = dx * (by1 - ay) - dy * (bx0 - ax), d3 = dx * (by1 - ay) - dy * (bx1 - ax);
                    const double tol = 1e-9 * (std::fabs(dx) + std::fabs(dy)) * (c + std::fabs(bx0 - ax) + std::fabs(by0 - ay) + 1.0);
                    const double mn = std::min(std::min(d0, d1), std::min(d2, d3)), mx = std::max(std::max(d0, d1), std::max(d2, d3));
                    if (mn > tol || mx < -tol) continue;
                    touched[(size_t)(iy - iy0) * w + (ix - ix0)] = 1;
                    hits.emplace_back((uint32_t)((size_t)iy * N.nx + ix), (int32_t)i);
                }
        }
        std::sort(hits.begin(), hits.end());
        for (size_t a = 0; a < hits.size();) { // one candidate per touched cell: its edges + a reference point off the boundary
            size_t b = a;
            while (b < hits.size() && hits[b].first == hits[a].first) ++b;
            const uint32_t cell = hits[a].first;
            const int ix = (int)(cell % (uint32_t)N.nx), iy = (int)(cell / (uint32_t)N.nx);
            sg::RoadCand cd{};
            cd.poly = (int32_t)gq;
            cd.edge_off = (uint32_t)B.cand_edges.size();
            if (b - a > 65535) return -1;
            cd.n_edges = (uint16_t)(b - a);
            int loc = 2;
            for (int sel = 0; sel < RN_NREF && loc == 2; ++sel) {
                double rx, ry;
                sg::rn_ref_point(N, ix, iy, sel, rx, ry);
                loc = sg::rn_polygon_locate(B.edges.data(), e0, e1, rx, ry);
                cd.ref_sel = (uint8_t)sel;
            }
            if (loc == 2) return -2; // every reference point of the cell lies on this polygon's boundary
            cd.ref_inside = (uint8_t)(loc == 1);
            for (size_t k = a; k < b; ++k) B.cand_edges.push_back(hits[k].second);
            cells[cell] |= (uint16_t)(L << 8);
            entries.push_back({cell, cd});
            a = b;
        }
        for (int iy = iy0; iy <= iy1; ++iy) {
            bool known = false, inside = false;
            for (int ix = ix0; ix <= ix1; ++ix) {
                const uint32_t cell = (uint32_t)((size_t)iy * N.nx + ix);
                if (touched[(size_t)(iy - iy0) * w + (ix - ix0)]) {
                    known = false;
                } else {
                    if (!known) {
                        inside = sg::rn_polygon_locate(B.edges.data(), e0, e1, N.x0 + (ix + 0.5) * c, N.y0 + (iy + 0.5) * c) == 1;
                        known = true;
                    }
                    if (inside) cells[cell] |= (uint16_t)L;
                }
            }
        }
    }
    std::stable_sort(entries.begin(), entries.end(), [](const Entry &x, const Entry &y) { return x.cell < y.cell; });
    // CSR (global over all networks: cell_off has one entry per cell + a final one appended by the caller)
    size_t k = 0;
    for (size_t cell = 0; cell < ncell; ++cell) {
        B.cell_off.push_back((uint32_t)B.cand.size());
        while (k < entries.size() && entries[k].cell == cell) B.cand.push_back(entries[k++].cd);
    }
    B.nets.push_back(N);
    return 0;
}
} // namespace

extern "C" int sg_set_road_networks(sg_handle *h, const sg_road_networks *in)
{
    if (!h || !in) return h ? fail(h, SG_ERR_INVALID, "sg_set_road_networks: null argument") : SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_set_road_networks: no scenarios uploaded");
    if (in->n_networks < 0 || !in->net_of_scenario || (in->n_networks > 0 && (!in->poly_off || !in->ring_off || !in->vert_off || !in->layers)))
        return fail(h, SG_ERR_INVALID, "sg_set_road_networks: null array");
    for (int r = 0; r < h->R; ++r)
        if (in->net_of_scenario[r] < -1 || in->net_of_scenario[r] >= in->n_networks)
            return fail(h, SG_ERR_INVALID, "sg_set_road_networks: net_of_scenario[%d]=%d out of range", r, in->net_of_scenario[r]);
    const int64_t n_poly = in->n_networks ? in->poly_off[in->n_networks] : 0;
    for (int n = 0; n < in->n_networks; ++n)
        if (in->poly_off[n + 1] < in->poly_off[n] || in->poly_off[0] != 0) return fail(h, SG_ERR_INVALID, "sg_set_road_networks: poly_off not monotone");
    for (int64_t q = 0; q < n_poly; ++q) {
        if (in->ring_off[q + 1] < in->ring_off[q] || in->ring_off[0] != 0) return fail(h, SG_ERR_INVALID, "sg_set_road_networks: ring_off not monotone");
        for (int64_t r = in->ring_off[q]; r < in->ring_off[q + 1]; ++r)
            if (in->vert_off[r + 1] < in->vert_off[r] || in->vert_off[0] != 0) return fail(h, SG_ERR_INVALID, "sg_set_road_networks: vert_off not monotone");
    }
    const int64_t n_vert = n_poly ? in->vert_off[in->ring_off[n_poly]] : 0;
    if (n_vert > 0 && !in->verts) return fail(h, SG_ERR_INVALID, "sg_set_road_networks: null verts");
    for (int64_t i = 0; i < 2 * n_vert; ++i)
        if (!std::isfinite(in->verts[i])) return fail(h, SG_ERR_INVALID, "sg_set_road_networks: vertex %lld is not finite", (long long)(i / 2));
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_pool(h->road_allocs);
    h->has_road = false;
    h->p.road = nullptr;
    RoadBuild B;
    B.poly_edge_off.push_back(0);
    B.imp_off.push_back(0);
    for (int n = 0; n < in->n_networks; ++n)
        if (int brc = build_road_network(in, n, B))
            return fail(h, SG_ERR_INVALID, "sg_set_road_networks: network %d cannot be indexed (%s)", n,
                        brc == -1 ? "more than 65535 edges of one polygon in one cell" : "a cell whose reference points all lie on a polygon boundary");
    B.cell_off.push_back((uint32_t)B.cand.size());
    if (B.cand.empty()) B.cand.push_back(sg::RoadCand{});
    if (B.cand_edges.empty()) B.cand_edges.push_back(0);
    if (B.edges.empty()) B.edges.assign(4, 0.0);
    if (B.nets.empty()) { B.nets.push_back(sg::RoadNet{0.0, 0.0, 1.0, 1, 1, 0}); B.cells.push_back(0); B.cell_off.insert(B.cell_off.begin(), 0u); B.net_flags.push_back(0); B.imp_off.push_back(0); }
    if (B.imp_edges.empty()) B.imp_edges.assign(4, 0.0);
    std::vector<int32_t> nos(in->net_of_scenario, in->net_of_scenario + h->R);
    auto &A = h->road_allocs;
    sg::RoadIndex R{};
    int rc = 0;
    if ((rc = dev_upload(h, A, &R.nets, B.nets))) return rc;
    if ((rc = dev_upload(h, A, &R.net_of_scen, nos))) return rc;
    if ((rc = dev_upload(h, A, &R.cells, B.cells))) return rc;
    if ((rc = dev_upload(h, A, &R.cell_off, B.cell_off))) return rc;
    if ((rc = dev_upload(h, A, &R.cand, B.cand))) return rc;
    if ((rc = dev_upload(h, A, &R.cand_edges, B.cand_edges))) return rc;
    if ((rc = dev_upload(h, A, &R.edges, B.edges))) return rc;
    if ((rc = dev_upload(h, A, &R.poly_layers, B.poly_layers))) return rc;
    if ((rc = dev_upload(h, A, &R.net_flags, B.net_flags))) return rc;
    if ((rc = dev_upload(h, A, &R.imp_off, B.imp_off))) return rc;
    if ((rc = dev_upload(h, A, &R.imp_edges, B.imp_edges))) return rc;
    {   // the filter tables of ped_boundary_terms (sgym_road.hpp)
        const size_t ne = B.imp_edges.size() / 4;
        std::vector<double> aux(ne * 4, 0.0), big(B.imp_off.size() - 1, 0.0);
        for (size_t i = 0; i < ne; ++i) {
            const double *e = &B.imp_edges[i * 4];
            const double dx = e[2] - e[0], dy = e[3] - e[1];
            aux[i * 4] = dx;
            aux[i * 4 + 1] = dy;
            aux[i * 4 + 2] = 1.0 / (dx * dx + dy * dy); // (a point edge: inf -- the filter's clamp turns the NaN it makes into t = 0)
        }
        for (size_t n = 0; n + 1 < B.imp_off.size(); ++n)
            for (int64_t i = B.imp_off[n] * 4; i < B.imp_off[n + 1] * 4; ++i) big[n] = std::max(big[n], std::fabs(B.imp_edges[(size_t)i]));
        if (big.empty()) big.push_back(0.0);
        if ((rc = dev_upload(h, A, &R.imp_aux, aux))) return rc;
        if ((rc = dev_upload(h, A, &R.imp_m, big))) return rc;
    }
    R.n_nets = in->n_networks;
    std::vector<sg::RoadIndex> one(1, R);
    const sg::RoadIndex *dR = nullptr;
    if ((rc = dev_upload(h, A, &dR, one))) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream)); // host vectors go out of scope
    h->road = R;
    h->p.road = dR;
    h->has_road = true;
    ++h->generation;
    return SG_OK;
}

// the raster kernels of sg_raster_map / sg_raster_map_device on the handle's stream; *d_out = [R][n_layers][nh][nw]
static int raster_map_launch(sg_handle *h, const char *who, double width, double height, int32_t nw, int32_t nh, int32_t n_layers,
                             const int32_t *layers, unsigned char **d_out, size_t *bytes_out)
{
    if (!layers || n_layers < 1 || nw < 1 || nh < 1 || !(width >= 0.0) || !(height >= 0.0))
        return fail(h, SG_ERR_INVALID, "%s: bad argument", who);
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "%s: no scenarios uploaded", who);
    bool any_surface = false;
    for (int k = 0; k < n_layers; ++k) {
        const uint32_t L = (uint32_t)layers[k];
        if (layers[k] < 0 || L > 255u || (L & (L - 1))) return fail(h, SG_ERR_INVALID, "%s: layers[%d]=%d is not 0 or one SG_LAYER_* bit", who, k, layers[k]);
        any_surface = any_surface || L != 0;
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t plane = (size_t)nw * nh, bytes = (size_t)h->R * n_layers * plane, lay_off = (bytes + 15) & ~(size_t)15;
    unsigned char *d = nullptr;
    int rc = obs_scratch(h, lay_off + (size_t)n_layers * sizeof(int32_t), &d);
    if (rc) return rc;
    int32_t *dl = reinterpret_cast<int32_t *>(d + lay_off);
    HIP_TRY(h, hipMemcpyAsync(dl, layers, (size_t)n_layers * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    if (any_surface && !h->has_road) HIP_TRY(h, hipMemsetAsync(d, 0, bytes, h->stream)); // no networks: empty surfaces
    for (int k = 0; k < n_layers; ++k)
        if (layers[k] == 0) {
            sg::raster_kernel<<<dim3((unsigned)h->R), dim3(h->EP > 256 ? 512 : 256), 0, h->stream>>>(h->p, width, height, nw, nh, d + (size_t)k * plane,
                                                                               (int64_t)(n_layers * plane));
            HIP_TRY(h, hipGetLastError());
        }
    if (any_surface && h->has_road) {
        sg::raster_surface_kernel<<<dim3((unsigned)h->R), dim3(256), 0, h->stream>>>(h->p, h->road, width, height, nw, nh, n_layers, dl, d);
        HIP_TRY(h, hipGetLastError());
    }
    *d_out = d;
    *bytes_out = bytes;
    return SG_OK;
}

extern "C" int sg_raster_map(sg_handle *h, double width, double height, int32_t nw, int32_t nh, int32_t n_layers,
                             const int32_t *layers, uint8_t *out)
{
    if (!h || !out) return h ? fail(h, SG_ERR_INVALID, "sg_raster_map: bad argument") : SG_ERR_INVALID;
    unsigned char *d = nullptr;
    size_t bytes = 0;
    int rc = raster_map_launch(h, "sg_raster_map", width, height, nw, nh, n_layers, layers, &d, &bytes);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (const int rcq = check_queue(h)) return rcq; // (a persistent launch that gave up: sticky)
    return SG_OK;
}

extern "C" int sg_raster_map_device(sg_handle *h, double width, double height, int32_t nw, int32_t nh, int32_t n_layers,
                                    const int32_t *layers, const uint8_t **d_out)
{
    if (!h || !d_out) return h ? fail(h, SG_ERR_INVALID, "sg_raster_map_device: bad argument") : SG_ERR_INVALID;
    unsigned char *d = nullptr;
    size_t bytes = 0;
    int rc = raster_map_launch(h, "sg_raster_map_device", width, height, nw, nh, n_layers, layers, &d, &bytes);
    if (rc) return rc;
    *d_out = d;
    return SG_OK;
}

extern "C" int sg_raster_entities(sg_handle *h, double width, double height, int32_t nw, int32_t nh, uint8_t *out)
{
    if (!h || !out || nw < 1 || nh < 1 || !(width >= 0.0) || !(height >= 0.0))
        return h ? fail(h, SG_ERR_INVALID, "sg_raster_entities: bad argument") : SG_ERR_INVALID;
    if (!h->uploaded) return fail(h, SG_ERR_STATE, "sg_raster_entities: no scenarios uploaded");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t bytes = (size_t)h->R * nw * nh;
    unsigned char *d = nullptr;
    int rc = obs_scratch(h, bytes, &d);
    if (rc) return rc;
    sg::raster_kernel<<<dim3((unsigned)h->R), dim3(h->EP > 256 ? 512 : 256), 0, h->stream>>>(h->p, width, height, nw, nh, d, (int64_t)nw * nh);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, SG_ERR_HIP, "sg_raster_entities: %s", hipGetErrorString(e));
    return check_queue(h); // (a persistent launch that gave up: sticky)
}

// ---- several devices from one process ---------------------------------------------------------------
// Scenarios never interact (one State per gym, scenario_gym.py:178; run_scenarios loops over them, :24-27): the replica
// axis is cut into contiguous shards, one sg_handle per device, no exchange during the step loop.  (The multi-process
// form of the same sharding is scenario_gym_amd/distributed.py over torch.distributed.)
struct sg_group {
    std::vector<sg_handle *> hs;
    std::vector<int> first; // first[i] = first scenario of shard i, first[n] = total
    int E = 0;
    std::string err;
};

static int gfail(sg_group *g, int code, const std::string &msg)
{
    if (g) g->err = msg;
    return code;
}

extern "C" const char *sg_group_last_error(const sg_group *g) { return g ? g->err.c_str() : g_create_err.c_str(); }

extern "C" int sg_group_destroy(sg_group *g)
{
    if (!g) return SG_OK;
    for (sg_handle *h : g->hs) (void)sg_destroy(h);
    delete g;
    return SG_OK;
}

extern "C" int sg_group_create(const sg_config *cfg, int32_t n_dev, const int32_t *devs, sg_group **out)
{
    if (!cfg || !out || n_dev < 1 || !devs) return fail(nullptr, SG_ERR_INVALID, "sg_group_create: bad argument");
    if (cfg->n_scenarios < n_dev) return fail(nullptr, SG_ERR_INVALID, "sg_group_create: fewer scenarios (%d) than devices (%d)", cfg->n_scenarios, n_dev);
    sg_group *g = new sg_group();
    g->E = cfg->n_entities;
    for (int i = 0; i <= n_dev; ++i) g->first.push_back((int)((int64_t)cfg->n_scenarios * i / n_dev));
    for (int i = 0; i < n_dev; ++i) {
        sg_config c = *cfg;
        c.device = devs[i];
        c.n_scenarios = g->first[i + 1] - g->first[i];
        sg_handle *h = nullptr;
        int rc = sg_create(&c, &h);
        if (rc) { sg_group_destroy(g); return rc; } // message in sg_last_error(NULL)
        g->hs.push_back(h);
    }
    *out = g;
    return SG_OK;
}

extern "C" int32_t sg_group_size(const sg_group *g) { return g ? (int32_t)g->hs.size() : 0; }
extern "C" sg_handle *sg_group_handle(sg_group *g, int32_t i) { return (g && i >= 0 && i < (int32_t)g->hs.size()) ? g->hs[i] : nullptr; }

extern "C" int sg_group_upload(sg_group *g, const sg_scenarios *sc)
{
    if (!g || !sc || !sc->knot_off) return gfail(g, SG_ERR_INVALID, "sg_group_upload: null argument");
    const int E = g->E;
    for (size_t i = 0; i < g->hs.size(); ++i) {
        const size_t r0 = (size_t)g->first[i], r1 = (size_t)g->first[i + 1], n = (r1 - r0) * E;
        sg_scenarios s = *sc;
        s.kind = sc->kind ? sc->kind + r0 * E : nullptr;
        s.etype = sc->etype ? sc->etype + r0 * E : nullptr;
        s.bbox = sc->bbox ? sc->bbox + r0 * E * 4 : nullptr;
        s.ctrl = sc->ctrl ? sc->ctrl + r0 * E * SG_NCTRL : nullptr;
        s.ego = sc->ego ? sc->ego + r0 : nullptr;
        s.t0 = sc->t0 ? sc->t0 + r0 : nullptr;
        s.length = sc->length ? sc->length + r0 : nullptr;
        std::vector<int64_t> koff(n + 1), roff;
        const int64_t kb = sc->knot_off[r0 * E];
        for (size_t k = 0; k <= n; ++k) koff[k] = sc->knot_off[r0 * E + k] - kb;
        s.knot_off = koff.data();
        s.knots = sc->knots ? sc->knots + (size_t)kb * 7 : nullptr;
        if (sc->route_off) {
            roff.resize(n + 1);
            const int64_t rb = sc->route_off[r0 * E];
            for (size_t k = 0; k <= n; ++k) roff[k] = sc->route_off[r0 * E + k] - rb;
            s.route_off = roff.data();
            s.routes = sc->routes ? sc->routes + (size_t)rb * 2 : nullptr;
        }
        int rc = sg_upload(g->hs[i], &s);
        if (rc) return gfail(g, rc, std::string("sg_group_upload: shard ") + std::to_string(i) + ": " + sg_last_error(g->hs[i]));
    }
    return SG_OK;
}

// ScenarioGym.rollout on every shard: all devices are launched before any is waited for
extern "C" int sg_group_rollout(sg_group *g, int32_t max_steps)
{
    if (!g) return SG_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) {
        int rc = sg_rollout_async(g->hs[i], max_steps, 1);
        if (rc) return gfail(g, rc, std::string("sg_group_rollout: shard ") + std::to_string(i) + ": " + sg_last_error(g->hs[i]));
    }
    for (size_t i = 0; i < g->hs.size(); ++i) {
        int rc = sg_synchronize(g->hs[i]);
        if (rc) return gfail(g, rc, std::string("sg_group_rollout: shard ") + std::to_string(i) + ": " + sg_last_error(g->hs[i]));
    }
    return SG_OK;
}

// ScenarioGym.get_metrics of all shards, scenario indices of the whole batch
extern "C" int sg_group_read_metrics(sg_group *g, sg_metrics *out, sg_event *events, int32_t cap, int32_t *n_events)
{
    if (!g || !out) return gfail(g, SG_ERR_INVALID, "sg_group_read_metrics: null argument");
    int32_t total = 0;
    for (size_t i = 0; i < g->hs.size(); ++i) {
        int32_t n = 0;
        int rc = sg_read_metrics(g->hs[i], out + g->first[i], events ? events + total : nullptr, events ? cap - total : 0, &n);
        if (rc) return gfail(g, rc, std::string("sg_group_read_metrics: shard ") + std::to_string(i) + ": " + sg_last_error(g->hs[i]));
        if (events)
            for (int32_t k = 0; k < std::min(n, cap - total); ++k) events[total + k].scenario += g->first[i];
        total += events ? std::min(n, cap - total) : n;
    }
    if (n_events) *n_events = total;
    return SG_OK;
}
