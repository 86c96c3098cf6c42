// gms_host.hip -- the C-ABI of libgridmapslam.so (include/gridmapslam.h): handle lifetime, staging,
// stream ordering, and the launch sequences that make up each reference method.
// There is no CPU path in this file: every compute entry point launches gfx950 kernels.
#include <math.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <algorithm>
#include <vector>

#include "gms_internal.h"

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int gms_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(GMS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

#define REQUIRE(cond, msg)                                     \
    do {                                                       \
        if (!(cond)) return fail(GMS_ERR_INVALID, "%s", msg);  \
    } while (0)

// Java (int) of a double for the host-side constructor arithmetic
static int32_t j_d2i_host(double d) {
    if (d != d) return 0;
    if (d >= 2147483647.0) return INT32_MAX;
    if (d <= -2147483648.0) return INT32_MIN;
    return (int32_t)d;
}

extern "C" {

int gms_version(void) { return GMS_VERSION_MAJOR * 1000 + GMS_VERSION_MINOR; }
#ifndef GMS_SOURCE_HASH
#define GMS_SOURCE_HASH "unknown"
#endif
// sha256 prefix of the sources this binary was built from (gridmap_slam_robot_amd/build.py prints the same string)
static const char g_build_info[] = "GMS_SOURCE_HASH=" GMS_SOURCE_HASH;
const char *gms_build_info(void) { return g_build_info + 16; }
const char *gms_last_error(void) { return g_err; }

int gms_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- pure host helpers ---------------------------------------------------------------------------
double gms_log_odds(double p) { return log(p / ((double)1.0f - p)); }                    // Util.java:35-37
double gms_inv_log_odds(double l) { return (double)1.0f - (double)1.0f / (1.0 + exp(l)); }   // Util.java:46-48

int gms_generate_gaussian_kernel(double sigma, int32_t size, double *out) {             // Util.java:428-455
    REQUIRE(out && size >= 0, "gms_generate_gaussian_kernel: bad arguments");
    const double norm = 1.0 / (sqrt(2 * M_PI) * sigma);
    const double coeff = 2 * sigma * sigma;
    double total = 0;
    for (int x = -size; x <= size; x++) {
        const double g = norm * exp((double)(-x * x) / coeff);
        out[x + size] = g;
        total += g;
    }
    for (int i = 0; i < 2 * size + 1; i++) out[i] /= total;
    return GMS_OK;
}

int gms_grid_size(const gms_params *p, int32_t *W, int32_t *H) {                         // GridMap.java:85
    REQUIRE(p && W && H, "gms_grid_size: null argument");
    *W = j_d2i_host(ceil((double)(p->width_m / p->resolution)));
    *H = j_d2i_host(ceil((double)(p->height_m / p->resolution)));
    return GMS_OK;
}

int gms_params_default(gms_params *p, float width_m, float height_m, float resolution, float pos_x, float pos_y) {
    REQUIRE(p, "gms_params_default: null params");
    memset(p, 0, sizeof(*p));
    p->width_m = width_m; p->height_m = height_m; p->resolution = resolution;
    p->pos_x = pos_x; p->pos_y = pos_y;
    p->n_maps = 1;
    p->device = 0;
    p->l_free = gms_log_odds((double)0.30f);                  // SensorModel.java:23
    p->l_occ = gms_log_odds((double)0.9f);                    // SensorModel.java:24
    const double sigma = sqrt(0.05 / (double)resolution);     // GridMap.java:94
    const int32_t size = j_d2i_host(ceil(sigma * 3));         // GridMap.java:95
    if (size < 0 || 2 * size + 1 > GMS_MAX_TAPS)
        return fail(GMS_ERR_INVALID, "likelihood kernel of %d taps exceeds GMS_MAX_TAPS", 2 * size + 1);
    p->ktaps = 2 * size + 1;
    gms_generate_gaussian_kernel(sigma, size, p->kernel);
    p->extra_steps = 2;                                       // GridMap.java:210
    p->hit_tolerance = 2.0f;                                  // GridMap.java:223
    p->z_hit = 0.9;                                           // GridMap.java:259
    p->z_random = 1 - p->z_hit;
    p->max_range = 10.0f;                                     // SensorModel.java:20
    p->max_beams = 0;
    return GMS_OK;
}

// ---- profiling ------------------------------------------------------------------------------------
}  // extern "C"

static void prof_drain(gms_map *m) {
    for (ProfSlot &s : m->prof_pending) {
        hipEventSynchronize(s.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            m->prof_ms[s.k] += ms;
            m->prof_n[s.k] += 1;
        }
        m->prof_free.push_back(s);
    }
    m->prof_pending.clear();
}

void gms_prof_begin(gms_map *m, int32_t k) {
    if (m->prof_pending.size() >= 8192) prof_drain(m);
    ProfSlot s;
    if (!m->prof_free.empty()) {
        s = m->prof_free.back();
        m->prof_free.pop_back();
    } else {
        hipEventCreate(&s.a);
        hipEventCreate(&s.b);
    }
    s.k = k;
    hipEventRecord(s.a, m->stream);
    m->prof_pending.push_back(s);
}

void gms_prof_end(gms_map *m) { hipEventRecord(m->prof_pending.back().b, m->stream); }

extern "C" {

int gms_profile_enable(gms_map *m, int32_t on) {
    REQUIRE(m, "null map");
    if (!on) prof_drain(m);
    m->prof_on = on;          // bit k brackets kernel class k
    return GMS_OK;
}
int gms_profile_sample(gms_map *m, int32_t stride) {
    REQUIRE(m && stride >= 1, "gms_profile_sample: stride must be >= 1");
    m->prof_stride = stride;
    return GMS_OK;
}
int gms_profile_reset(gms_map *m) {
    REQUIRE(m, "null map");
    prof_drain(m);
    for (int k = 0; k < GMS_K_COUNT; k++) { m->prof_ms[k] = 0; m->prof_n[k] = 0; m->prof_seen[k] = 0; }
    return GMS_OK;
}
int gms_profile_get(gms_map *m, int32_t k, double *total_ms, int64_t *launches) {
    REQUIRE(m && k >= 0 && k < GMS_K_COUNT, "gms_profile_get: bad arguments");
    prof_drain(m);
    if (total_ms) *total_ms = m->prof_ms[k];
    if (launches) *launches = m->prof_n[k];
    return GMS_OK;
}

int gms_profile_calibrate(gms_map *m, int32_t reps, double *bracket_ms) {
    REQUIRE(m && bracket_ms && reps > 0 && reps <= 4096, "gms_profile_calibrate: bad arguments");
    HIPCHK(hipSetDevice(m->device));
    // event, empty kernel, event, empty kernel, ... enqueued back to back (the queue never drains, as in a scan loop)
    const int32_t skip = 8, n = reps + skip;
    std::vector<hipEvent_t> ev(n + 1);
    for (hipEvent_t &e : ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventRecord(ev[0], m->stream));
    for (int32_t i = 0; i < n; i++) {
        gms_launch_noop(m);
        HIPCHK(hipEventRecord(ev[i + 1], m->stream));
    }
    HIPCHK(hipEventSynchronize(ev[n]));
    double total = 0.0;
    for (int32_t i = skip; i < n; i++) {
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        total += ms;
    }
    for (hipEvent_t &e : ev) hipEventDestroy(e);
    *bracket_ms = total / reps;
    return GMS_OK;
}

// The same, split into its two parts.  bracket_ms: as gms_profile_calibrate.  kernel_ms: what ONE empty kernel takes when the
// launches are already queued and run back to back with no events between them (a spinning kernel holds the stream while the
// host enqueues them, so the host's launch rate does not enter): the floor a kernel-trace profiler shows for any launch.
// bracket_ms - kernel_ms is what the two event markers add to a bracketed launch.
int gms_profile_calibrate2(gms_map *m, int32_t reps, double *bracket_ms, double *kernel_ms) {
    REQUIRE(m && bracket_ms && kernel_ms && reps > 0 && reps <= 1024, "gms_profile_calibrate2: bad arguments");
    int rc = gms_profile_calibrate(m, reps, bracket_ms);
    if (rc) return rc;
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    gms_launch_spin(m, 30.0 + 6.0 * reps);             // microseconds: ample for the host to enqueue what follows
    gms_launch_noop(m);
    HIPCHK(hipEventRecord(a, m->stream));
    for (int32_t i = 0; i < reps; i++) gms_launch_noop(m);
    HIPCHK(hipEventRecord(b, m->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a); hipEventDestroy(b);
    *kernel_ms = (double)ms / reps;
    return GMS_OK;
}

// ---- pinned staging rings -----------------------------------------------------------------------------
static int ring_alloc(StageRing &r, size_t bytes) {
    for (int i = 0; i < GMS_STAGE_SLOTS; i++) {
        HIPCHK(hipHostMalloc(&r.slot[i], bytes));
        HIPCHK(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
    }
    return GMS_OK;
}
static void ring_free(StageRing &r) {
    for (int i = 0; i < GMS_STAGE_SLOTS; i++) {
        if (r.slot[i]) hipHostFree(r.slot[i]);
        if (r.ev[i]) hipEventDestroy(r.ev[i]);
        r.slot[i] = nullptr; r.ev[i] = nullptr; r.busy[i] = false;
    }
}
// next slot, free to be overwritten by the host (waits only for the copy that used it GMS_STAGE_SLOTS calls ago)
static int ring_acquire(StageRing &r, void **out) {
    const int i = r.next;
    if (r.busy[i]) { HIPCHK(hipEventSynchronize(r.ev[i])); r.busy[i] = false; }
    *out = r.slot[i];
    return GMS_OK;
}
// call after enqueueing the asynchronous copy out of the acquired slot
static int ring_commit(StageRing &r, hipStream_t stream) {
    const int i = r.next;
    HIPCHK(hipEventRecord(r.ev[i], stream));
    r.busy[i] = true;
    r.next = (i + 1) % GMS_STAGE_SLOTS;
    return GMS_OK;
}

// ---- GridMap --------------------------------------------------------------------------------------
static int map_free(gms_map *m) {
    if (!m) return GMS_OK;
    prof_drain(m);
    for (ProfSlot &s : m->prof_free) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    hipFree(m->d_log); hipFree(m->d_lik); hipFree(m->d_fac); hipFree(m->d_cnt); hipFree(m->d_cnt_pend); hipFree(m->d_bbox); hipFree(m->d_taps); hipFree(m->d_tile_state); hipFree(m->d_tile_stats);
    hipFree(m->d_beams); hipFree(m->d_poses); hipFree(m->d_scratch);
    hipFree(m->d_trace_cells); hipFree(m->d_trace_cls); hipFree(m->d_trace_cnt);
    if (m->h_beams) hipHostFree(m->h_beams);
    ring_free(m->beam_ring);
    if (m->pose_copy_ev_set) hipEventDestroy(m->pose_copy_ev);
    if (m->h_poses) hipHostFree(m->h_poses);
    if (m->own_stream) hipStreamDestroy(m->own_stream);

    delete m;
    return GMS_OK;
}

int gms_map_create(const gms_params *p, gms_map **out) {
    REQUIRE(p && out, "gms_map_create: null argument");
    *out = nullptr;
    REQUIRE(p->resolution > 0.0f && p->n_maps >= 1 && p->n_maps <= 1024, "gms_map_create: bad resolution / n_maps");
    REQUIRE(p->ktaps >= 1 && (p->ktaps & 1) && p->ktaps <= GMS_MAX_TAPS, "gms_map_create: ktaps must be odd and <= GMS_MAX_TAPS");
    // One scan holds at most GMS_MAX_BEAMS beams: the scoring kernel stages 128 beams per segment product and keeps at
    // most GMS_SCORE_MAXSEG of them, and the per-scan cell counts are 16-bit halves (a cell is visited at most
    // (1 + extra_steps) times per ray -- a zero-length ray emits its cell n = 1 + extra times, RayIterator.java:75 --
    // so (1 + extra_steps) * beams must stay below 65536).
    REQUIRE(p->max_beams >= 0 && p->max_beams <= GMS_MAX_BEAMS, "gms_map_create: max_beams exceeds GMS_MAX_BEAMS");
    REQUIRE(p->extra_steps >= 0 && (int64_t)(1 + p->extra_steps) * GMS_MAX_BEAMS <= 65535, "gms_map_create: extra_steps too large for the 16-bit per-scan cell counts");
    int32_t W, H;
    gms_grid_size(p, &W, &H);
    REQUIRE(W > 0 && H > 0 && (int64_t)W * H < (1ll << 31), "gms_map_create: grid size out of range");
    REQUIRE(W + 16 < (1 << 24) && H < (1 << 24) && (int64_t)(H + 1) * (W + 16) < (1ll << 32), "gms_map_create: grid size out of range");   // fac_index: 24-bit factors, 32-bit index
    int ndev = gms_device_count();
    if (ndev <= 0) return fail(GMS_ERR_NO_DEVICE, "no HIP device visible: libgridmapslam has no CPU path");
    if (p->device < 0 || p->device >= ndev) return fail(GMS_ERR_NO_DEVICE, "device %d of %d not available", p->device, ndev);
    HIPCHK(hipSetDevice(p->device));
    {   // the likelihood pass stages a tile with a halo of (ktaps - 1) / 2 cells in LDS: 11 taps 25 KiB, 65 taps 149 KiB
        int lds_max = 0;
        if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, p->device) != hipSuccess) lds_max = 64 * 1024;
        if (gms_likelihood_lds_bytes((p->ktaps - 1) / 2) + 4096 > (size_t)lds_max)
            return fail(GMS_ERR_INVALID, "gms_map_create: a blur kernel of %d taps needs %zu bytes of LDS per workgroup, this device offers %d",
                        p->ktaps, gms_likelihood_lds_bytes((p->ktaps - 1) / 2) + 4096, lds_max);
    }

    gms_map *m = new (std::nothrow) gms_map();
    if (!m) return fail(GMS_ERR_NOMEM, "out of host memory");
    m->prm = *p;
    m->n_maps = p->n_maps;
    m->device = p->device;
    m->max_beams = p->max_beams > 0 ? p->max_beams : 2048;
    {   // residency arithmetic of the persistent-workgroup launches (gms_likelihood_blocks_cap)
        int v = 0;
        m->n_cus = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, p->device) == hipSuccess && v > 0 ? v : 256;
        v = 0;
        m->lds_per_cu = hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, p->device) == hipSuccess && v > 0 ? v : 64 * 1024;
        v = 0;      // (a CU holds at least what one workgroup may ask for: some runtimes report the per-CU figure low)
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, p->device) == hipSuccess && v > m->lds_per_cu) m->lds_per_cu = v;
    }
    GridDev &g = m->gd;
    g.W = W; g.H = H; g.cells = (int64_t)W * H;
    g.posx = (double)p->pos_x; g.posy = (double)p->pos_y;
    g.res = (double)p->resolution; g.resf = p->resolution;
    g.rinv = 1.0 / g.res;
    g.l_free = p->l_free; g.l_occ = p->l_occ;
    g.extra = p->extra_steps;
    g.half_tol = p->hit_tolerance / 2;                               // SensorModel.java:35 float arithmetic
    g.z_hit = p->z_hit;
    g.c_rand = p->z_random * 1.0 / (double)p->max_range;             // GridMap.java:288
    g.inv_max = 1.0 / (double)p->max_range;                          // GridMap.java:286
    g.ktaps = p->ktaps; g.khalf = (p->ktaps - 1) / 2;                // Util.java:384
    m->lik_kh = (g.khalf == 3 || g.khalf == 5) ? g.khalf : 0;
    m->taps_plain = 1;
    for (int32_t i = 0; i < p->ktaps; i++) {
        const double a = p->kernel[i];
        if (std::signbit(a) || (a != 0.0 && !(a >= 0x1p-900 && a <= 0x1p900))) m->lik_kh = m->taps_plain = 0;
    }

    const size_t cells = (size_t)g.cells * m->n_maps;
    hipError_t e = hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete m; return fail(GMS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    m->stream = m->own_stream;
    bool ok = true;

    ok = ok && hipMalloc(&m->d_log, cells * sizeof(double)) == hipSuccess;
    ok = ok && hipMalloc(&m->d_lik, cells * sizeof(double)) == hipSuccess;
    g.fpitch = W + 16;                                               // factor table with a neutral border (fac_index)
    g.fneutral = (uint32_t)H * (uint32_t)g.fpitch + (uint32_t)W;
    m->fac_stride = (int64_t)(H + 1) * g.fpitch;
    ok = ok && hipMalloc(&m->d_fac, (size_t)m->fac_stride * m->n_maps * sizeof(double)) == hipSuccess;
    ok = ok && hipMalloc(&m->d_cnt, cells * sizeof(uint32_t)) == hipSuccess && hipMalloc(&m->d_cnt_pend, cells * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(&m->d_bbox, (size_t)m->n_maps * 8 * sizeof(int32_t)) == hipSuccess;
    ok = ok && hipMalloc(&m->d_taps, GMS_MAX_TAPS * sizeof(double)) == hipSuccess;
    ok = ok && hipMalloc(&m->d_tile_state, 2 * (size_t)((g.W + 63) / 64) * ((g.H + 31) / 32) * m->n_maps) == hipSuccess;
    ok = ok && hipMalloc(&m->d_tile_stats, 64 * 4 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(&m->d_beams, (size_t)m->n_maps * m->max_beams * sizeof(gms_beam)) == hipSuccess;
    ok = ok && hipMalloc(&m->d_poses, (size_t)m->n_maps * 3 * sizeof(float) + 16) == hipSuccess;   // (+16: copied in 16-byte units)
    ok = ok && hipMalloc(&m->d_scratch, 64 * sizeof(double)) == hipSuccess;
    ok = ok && hipHostMalloc(&m->h_beams, (size_t)m->n_maps * m->max_beams * sizeof(gms_beam)) == hipSuccess;
    ok = ok && ring_alloc(m->beam_ring, (size_t)m->n_maps * m->max_beams * sizeof(gms_beam)) == GMS_OK;
    ok = ok && hipHostMalloc(&m->h_poses, (size_t)m->n_maps * 3 * sizeof(float) + 64 * sizeof(double)) == hipSuccess;
    if (!ok) { map_free(m); return fail(GMS_ERR_NOMEM, "device allocation failed (%zu cells x %d maps)", (size_t)g.cells, m->n_maps); }
    hipMemcpyAsync(m->d_taps, p->kernel, p->ktaps * sizeof(double), hipMemcpyHostToDevice, m->stream);
    hipMemsetAsync(m->d_log, 0, cells * sizeof(double), m->stream);   // logOdds(0.5) == 0.0 (createMapData(null))
    hipMemsetAsync(m->d_lik, 0, cells * sizeof(double), m->stream);
    hipMemsetAsync(m->d_cnt, 0, cells * sizeof(uint32_t), m->stream);
    hipMemsetAsync(m->d_cnt_pend, 0, cells * sizeof(uint32_t), m->stream);
    hipMemsetAsync(m->d_bbox, 0, (size_t)m->n_maps * 8 * sizeof(int32_t), m->stream);
    hipMemsetAsync(m->d_tile_stats, 0, 64 * 4 * sizeof(uint32_t), m->stream);
    gms_launch_factors(m);        // likelihoodData == 0 everywhere (createMapData(null))
    HIPCHK(hipStreamSynchronize(m->stream));
    m->need_full_build = 1; m->fac_current = 0;
    m->pair_launches = 1;
    {   // the tiled batched ray cast: 8 KiB of slots + a 64 KiB tile + static LDS
        int lds_max = 0;
        if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, m->device) != hipSuccess) lds_max = 64 * 1024;
        m->raycast_tile = lds_max >= 80 * 1024;
        if (const char *v = getenv("GMS_RAYCAST_TILE")) m->raycast_tile = m->raycast_tile && atoi(v) != 0;
        m->raycast_tile_min = 4096;
        if (const char *v = getenv("GMS_RAYCAST_TILE_MIN")) m->raycast_tile_min = atoi(v);
        m->raycast_near = lds_max >= 40 * 1024;                       // 25 KiB tile + slots + static LDS
        if (const char *v = getenv("GMS_RAYCAST_NEAR")) m->raycast_near = !m->raycast_near ? 0 : (atoi(v) != 0 ? 2 : 0);   // 1: for every scan of 32 beams or more
    }
    m->prof_stride = 1;
    m->lik_lazy = 1;
    if (const char *v = getenv("GMS_LIK_LAZY")) m->lik_lazy = atoi(v) != 0;
    m->lik_skip = 1;
    if (const char *v = getenv("GMS_LIK_SKIP")) m->lik_skip = atoi(v) != 0;
    m->fac_current = 0;
    if (const char *v = getenv("GMS_PAIR_LAUNCHES")) m->pair_launches = atoi(v) != 0;
    if (const char *v = getenv("GMS_SLAM_TILE_CELLS")) m->slam_tile_cells = atoi(v);
    m->lik_split = 1;
    if (const char *v = getenv("GMS_LIK_SPLIT")) m->lik_split = atoi(v) != 0;
    if (const char *v = getenv("GMS_SLAM_THREADS")) m->slam_threads = atoi(v);
    *out = m;
    return GMS_OK;
}

int gms_map_destroy(gms_map *m) {
    if (!m) return GMS_OK;
    // a filter keeps a pointer to its map (stream, device, grids): destroy the filters first
    if (m->n_filters > 0) return fail(GMS_ERR_STATE, "gms_map_destroy: %d particle filter(s) still bound to this map", m->n_filters);
    hipSetDevice(m->device);
    hipStreamSynchronize(m->stream);
    return map_free(m);
}

int gms_map_get_size(const gms_map *m, int32_t *W, int32_t *H, int32_t *n_maps) {
    REQUIRE(m, "null map");
    if (W) *W = m->gd.W;
    if (H) *H = m->gd.H;
    if (n_maps) *n_maps = m->n_maps;
    return GMS_OK;
}

int gms_map_set_stream(gms_map *m, void *hip_stream) {
    REQUIRE(m, "null map");
    gms_flush_apply(m);
    HIPCHK(hipStreamSynchronize(m->stream));
    m->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : m->own_stream;
    return GMS_OK;
}

int gms_map_synchronize(gms_map *m) {
    REQUIRE(m, "null map");
    HIPCHK(hipStreamSynchronize(m->stream));
    return GMS_OK;
}

int gms_map_reset(gms_map *m) {                                        // GridMap.java:129-132
    REQUIRE(m, "null map");
    HIPCHK(hipSetDevice(m->device));
    gms_ensure_lik(m);                 // likelihoodData keeps the last field (reset touches logData only)
    gms_flush_apply(m);
    HIPCHK(hipMemsetAsync(m->d_log, 0, (size_t)m->gd.cells * m->n_maps * sizeof(double), m->stream));
    m->need_full_build = 1; m->fac_current = 0;
    return GMS_OK;
}

static int map_xfer(gms_map *m, void *dev, void *host, bool to_device) {
    const size_t bytes = (size_t)m->gd.cells * m->n_maps * sizeof(double);
    HIPCHK(hipSetDevice(m->device));
    if (to_device) HIPCHK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, m->stream));
    else HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return GMS_OK;
}

int gms_map_upload_log(gms_map *m, const double *log_data) {
    REQUIRE(m && log_data, "null argument");
    gms_ensure_lik(m);
    gms_flush_apply(m);
    m->need_full_build = 1; m->fac_current = 0;
    return map_xfer(m, m->d_log, const_cast<double *>(log_data), true);
}
int gms_map_download_log(gms_map *m, double *log_data) {
    REQUIRE(m && log_data, "null argument");
    gms_flush_apply(m);
    return map_xfer(m, m->d_log, log_data, false);
}
int gms_map_upload_likelihood(gms_map *m, const double *lik) {
    REQUIRE(m && lik, "null argument");
    m->need_full_build = 1; m->fac_current = 0;
    m->lik_stale = 0;                  // replaced wholesale
    int rc = map_xfer(m, m->d_lik, const_cast<double *>(lik), true);
    if (rc) return rc;
    gms_launch_factors(m);
    HIPCHK(hipGetLastError());
    return GMS_OK;
}
int gms_map_download_likelihood(gms_map *m, double *lik) {
    REQUIRE(m && lik, "null argument");
    gms_ensure_lik(m);
    return map_xfer(m, m->d_lik, lik, false);
}

// `later` waits (stream order, no host synchronise) for everything enqueued on `earlier` so far
static int stream_after(hipStream_t later, hipStream_t earlier, const char *what) {
    if (later == earlier) return GMS_OK;
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, earlier);
    if (e == hipSuccess) e = hipStreamWaitEvent(later, ev, 0);
    hipEventDestroy(ev);                                             // released once the recorded work has completed
    if (e != hipSuccess) return fail(GMS_ERR_HIP, "%s: stream hand-over: %s", what, hipGetErrorString(e));
    return GMS_OK;
}

int gms_map_copy(gms_map *dst, const gms_map *src) {                  // GridMap.java:106-124
    REQUIRE(dst && src, "null argument");
    REQUIRE(dst->gd.W == src->gd.W && dst->gd.H == src->gd.H && dst->n_maps == src->n_maps, "gms_map_copy: shape mismatch");
    REQUIRE(dst->device == src->device, "gms_map_copy: both handles must live on the same device (the copies and their stream hand-over are device-local)");
    const size_t bytes = (size_t)src->gd.cells * src->n_maps * sizeof(double);
    HIPCHK(hipSetDevice(dst->device));
    gms_ensure_lik(const_cast<gms_map *>(src));
    gms_flush_apply(const_cast<gms_map *>(src));
    gms_flush_apply(dst);
    dst->lik_stale = 0;
    // The copies run on dst's stream and read src's arrays: dst's stream waits for what src's stream holds so far (the flushed
    // apply pass among it), and src's stream waits for the copies before anything enqueued on it later may write those arrays
    // again (a ray cast's apply pass, reset, upload): both directions in stream order, no host synchronise.
    int rc = stream_after(dst->stream, src->stream, "gms_map_copy");
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(dst->d_log, src->d_log, bytes, hipMemcpyDeviceToDevice, dst->stream));
    HIPCHK(hipMemcpyAsync(dst->d_lik, src->d_lik, bytes, hipMemcpyDeviceToDevice, dst->stream));
    rc = stream_after(src->stream, dst->stream, "gms_map_copy");
    if (rc) return rc;
    gms_launch_factors(dst);
    dst->need_full_build = 1; dst->fac_current = 0;
    return GMS_OK;
}

int gms_map_combine(gms_map *dst, gms_map *src) {                        // GridMapApp.java:439-458
    REQUIRE(dst && src, "null argument");
    REQUIRE(dst->n_maps == 1 && dst->gd.W == src->gd.W && dst->gd.H == src->gd.H, "gms_map_combine: dst must be one map of the same size");
    REQUIRE(dst->device == src->device, "gms_map_combine: both handles must live on the same device");
    HIPCHK(hipSetDevice(dst->device));
    // src's deferred apply pass (a fused scan step or gms_map_update leaves the last scan's counts un-applied) is enqueued on
    // src's stream; the combine reads src's log-odds on dst's stream, so dst's stream must wait for it -- an event after the
    // flush, not a host synchronise before it (round 2 synchronised first and flushed afterwards: the combine could read
    // pre-apply log-odds).  And the other way round: whatever src's stream is given after this call (the next scan's apply
    // pass, a reset, an upload) writes the log-odds the combine reads, so src's stream waits for the combine (round 3 left
    // that direction open: a write-after-read race whenever the two handles run on different streams).
    gms_ensure_lik(dst);               // the destination's likelihoodData keeps its last field (its logData is about to be replaced)
    gms_flush_apply(src);
    gms_flush_apply(dst);
    int rc = stream_after(dst->stream, src->stream, "gms_map_combine");
    if (rc) return rc;
    gms_launch_combine(src, dst);
    rc = stream_after(src->stream, dst->stream, "gms_map_combine");
    if (rc) return rc;
    dst->need_full_build = 1; dst->fac_current = 0;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

// raw polar scan (host) -> de-skewed beams in the map's device staging buffer; returns its device address
int gms_map_deskew(gms_map *m, const double *angle, const double *distance, const uint8_t *hit, int32_t length,
                   double d_center, double d_theta, gms_beam *beams_out, const gms_beam **dev_beams_out) {
    REQUIRE(m && angle && distance && hit, "null argument");
    REQUIRE(length >= 0 && length <= m->max_beams, "measurement count exceeds gms_params.max_beams");
    HIPCHK(hipSetDevice(m->device));
    // The raw input goes into a pinned ring slot, [angle | distance | hit], and the de-skew kernel reads it in place over
    // PCIe (17 bytes per measurement): no stream synchronise at entry, no hipMemcpyAsync (45 us of host time on this stack).
    REQUIRE((size_t)length * 17 + 16 <= (size_t)m->n_maps * m->max_beams * sizeof(gms_beam), "scan too long for the staging buffer");
    void *slot = nullptr;
    int rc = ring_acquire(m->beam_ring, &slot);
    if (rc) return rc;
    double *h_a = static_cast<double *>(slot), *h_d = h_a + length;
    uint8_t *h_h = reinterpret_cast<uint8_t *>(h_d + length);
    memcpy(h_a, angle, (size_t)length * 8); memcpy(h_d, distance, (size_t)length * 8); memcpy(h_h, hit, (size_t)length);
    gms_launch_deskew(m, h_a, h_d, h_h, length, d_center, d_theta, m->d_beams);
    rc = ring_commit(m->beam_ring, m->stream);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    if (dev_beams_out) *dev_beams_out = m->d_beams;
    if (beams_out) {
        HIPCHK(hipMemcpyAsync(beams_out, m->d_beams, (size_t)length * sizeof(gms_beam), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
    }
    return GMS_OK;
}

int gms_map_get_raw_at(gms_map *m, int32_t mi, int32_t x, int32_t y, double *raw, double *prob) {
    REQUIRE(m, "null map");
    // Java would throw ArrayIndexOutOfBounds (GridMap.java:135)
    REQUIRE(mi >= 0 && mi < m->n_maps && x >= 0 && x < m->gd.W && y >= 0 && y < m->gd.H, "gms_map_get_raw_at: index out of bounds");
    HIPCHK(hipSetDevice(m->device));
    gms_flush_apply(m);
    gms_launch_get_raw(m, mi, x, y, m->d_scratch);
    double *h = reinterpret_cast<double *>(m->h_poses + (size_t)m->n_maps * 3);
    h = reinterpret_cast<double *>(((uintptr_t)h + 7) & ~(uintptr_t)7);
    HIPCHK(hipMemcpyAsync(h, m->d_scratch, 2 * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (raw) *raw = h[0];
    if (prob) *prob = h[1];
    return GMS_OK;
}

// Java (int) of a float (Float.intValue(): saturating, NaN -> 0)
static int32_t j_f2i_host(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}

int gms_map_get_at_point(gms_map *m, int32_t mi, float point_x, float point_y, double *raw, double *likelihood) {
    REQUIRE(m, "null map");
    REQUIRE(mi >= 0 && mi < m->n_maps, "gms_map_get_at_point: map index out of range");
    // tmp = point; tmp -= position; tmp /= resolution  (float Vec2 arithmetic, GridMap.java:143-145)
    const float tx = (point_x - m->prm.pos_x) / m->prm.resolution, ty = (point_y - m->prm.pos_y) / m->prm.resolution;
    const int32_t ix = j_f2i_host(tx), iy = j_f2i_host(ty);
    const int32_t idx = (int32_t)((uint32_t)ix + (uint32_t)iy * (uint32_t)m->gd.W);       // Java int arithmetic wraps
    if (idx < 0 || (int64_t)idx >= m->gd.cells)
        return fail(GMS_ERR_INVALID, "gms_map_get_at_point: index %d out of bounds (Java: ArrayIndexOutOfBoundsException)", idx);
    HIPCHK(hipSetDevice(m->device));
    gms_ensure_lik(m);
    gms_flush_apply(m);
    double *h = reinterpret_cast<double *>(m->h_poses + (size_t)m->n_maps * 3);
    h = reinterpret_cast<double *>(((uintptr_t)h + 7) & ~(uintptr_t)7);
    const size_t o = (size_t)mi * m->gd.cells + idx;
    HIPCHK(hipMemcpyAsync(h, m->d_log + o, sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(h + 1, m->d_lik + o, sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (raw) *raw = h[0];
    if (likelihood) *likelihood = h[1];
    return GMS_OK;
}

// beams [n_maps][B] (host) -> d_beams [n_maps][max_beams]
static int stage_beams(gms_map *m, const gms_beam *beams, int32_t B) {
    REQUIRE(beams, "null beams");
    REQUIRE(B >= 0 && B <= m->max_beams, "beam count exceeds gms_params.max_beams");
    HIPCHK(hipSetDevice(m->device));
    void *slot = nullptr;
    int rc = ring_acquire(m->beam_ring, &slot);
    if (rc) return rc;
    gms_beam *h = static_cast<gms_beam *>(slot);
    for (int32_t mi = 0; mi < m->n_maps; mi++)
        memcpy(h + (size_t)mi * m->max_beams, beams + (size_t)mi * B, (size_t)B * sizeof(gms_beam));
    // (the device-side staging buffer is safe to overwrite in stream order: earlier kernels that read it come first;
    // sizeof(gms_beam) is a multiple of 16, so the copy kernel's granularity fits)
    gms_launch_copy(m, m->d_beams, h, (m->n_maps == 1 ? (size_t)B : (size_t)m->n_maps * m->max_beams) * sizeof(gms_beam));
    return ring_commit(m->beam_ring, m->stream);
}

}  // extern "C"
int gms_stage_beams(gms_map *m, const gms_beam *beams, int32_t B) { return stage_beams(m, beams, B); }
extern "C" {

static int stage_poses(gms_map *m, const float *poses) {
    REQUIRE(poses, "null poses");
    // 12 bytes per map: passed through the beam ring's slot-sized buffers would be wasteful; the previous copy out
    // of h_poses is awaited instead (it is long done unless calls come back to back)
    if (m->pose_copy_ev_set) HIPCHK(hipEventSynchronize(m->pose_copy_ev));
    memcpy(m->h_poses, poses, (size_t)m->n_maps * 3 * sizeof(float));
    gms_launch_copy(m, m->d_poses, m->h_poses, (size_t)m->n_maps * 3 * sizeof(float));
    if (!m->pose_copy_ev_set) { HIPCHK(hipEventCreateWithFlags(&m->pose_copy_ev, hipEventDisableTiming)); m->pose_copy_ev_set = 1; }
    HIPCHK(hipEventRecord(m->pose_copy_ev, m->stream));
    return GMS_OK;
}

// device address of the filter's weighted pose (which = 0) or strongest particle's pose (1), map 0
static const float *stats_pose_ptr(const gms_pf *pf, int32_t which) {
    const char *base = reinterpret_cast<const char *>(pf->d_stats);
    return reinterpret_cast<const float *>(base + (which == 0 ? offsetof(PfStatsDev, wpose) : offsetof(PfStatsDev, spose)));
}

static int finish_likelihood(gms_map *m, int32_t dirty_only) {
    if (dirty_only && !m->bbox_dirty) return GMS_OK;     // nothing changed since the last build
    gms_launch_likelihood(m, dirty_only);
    if (m->bbox_dirty) {          // the box is consumed; the other half was cleared by k_apply
        m->bbox_cur = 1 - m->bbox_cur;
        m->bbox_dirty = 0;
    }
    m->need_full_build = 0;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_map_integrate(gms_map *m, const gms_beam *beams, int32_t B, const float *poses) {   // GridMap.java:173-191
    REQUIRE(m, "null map");
    gms_ensure_lik(m);                 // integrateObservation leaves likelihoodData as the last rebuild made it: have it made first
    int rc = stage_beams(m, beams, B);
    if (rc) return rc;
    rc = stage_poses(m, poses);
    if (rc) return rc;
    if (B > 0) {
        gms_launch_raycast(m, m->d_beams, B, m->max_beams, m->d_poses, 3);
        gms_launch_apply_counts(m);
    }
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_map_integrate_dev(gms_map *m, const gms_beam *dev_beams, int32_t B, const float *dev_poses) {
    REQUIRE(m && dev_beams && dev_poses, "null argument");
    REQUIRE(B >= 0 && B <= m->max_beams, "beam count exceeds gms_params.max_beams");
    HIPCHK(hipSetDevice(m->device));
    gms_ensure_lik(m);
    if (B > 0) {
        gms_launch_raycast(m, dev_beams, B, B, dev_poses, 3);
        gms_launch_apply_counts(m);
    }
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_map_integrate_at_dev(gms_map *m, const gms_beam *dev_beams, int32_t B, gms_pf *pf, int32_t which) {
    REQUIRE(m && dev_beams && pf && pf->map == m, "gms_map_integrate_at_dev: bad arguments");
    REQUIRE(which == 0 || which == 1, "which must be 0 (weighted pose) or 1 (strongest particle)");
    REQUIRE(B >= 0 && B <= m->max_beams, "beam count exceeds gms_params.max_beams");
    HIPCHK(hipSetDevice(m->device));
    gms_ensure_lik(m);
    if (B > 0) {
        gms_launch_raycast(m, dev_beams, B, B, stats_pose_ptr(pf, which), (int32_t)(sizeof(PfStatsDev) / sizeof(float)));
        gms_launch_apply_counts(m);
    }
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_map_integrate_at(gms_map *m, const gms_beam *beams, int32_t B, gms_pf *pf, int32_t which) {
    REQUIRE(m && pf && pf->map == m, "gms_map_integrate_at: filter does not belong to this map");
    REQUIRE(which == 0 || which == 1, "which must be 0 (weighted pose) or 1 (strongest particle)");
    gms_ensure_lik(m);
    int rc = stage_beams(m, beams, B);
    if (rc) return rc;
    if (B > 0) {
        gms_launch_raycast(m, m->d_beams, B, m->max_beams, stats_pose_ptr(pf, which), (int32_t)(sizeof(PfStatsDev) / sizeof(float)));
        gms_launch_apply_counts(m);
    }
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_map_apply_ray(gms_map *m, float sx, float sy, float ex, float ey, float measured, int32_t hit) {
    REQUIRE(m, "null map");
    HIPCHK(hipSetDevice(m->device));
    gms_ensure_lik(m);
    RayIn r;
    r.sx = sx; r.sy = sy; r.ex = ex; r.ey = ey; r.measured = measured; r.hit = hit != 0;
    gms_launch_apply_ray(m, r);
    gms_launch_apply_counts(m);
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

static int ensure_trace(gms_map *m, size_t cells, size_t counts) {
    // the three buffers have capacities of their own: a call with fewer cells but more counts than the last one
    // must still grow the count buffer
    if (cells <= m->trace_cap_cells && counts <= m->trace_cap_counts) return GMS_OK;
    HIPCHK(hipStreamSynchronize(m->stream));
    if (cells < m->trace_cap_cells) cells = m->trace_cap_cells;
    if (counts < m->trace_cap_counts) counts = m->trace_cap_counts;
    hipFree(m->d_trace_cells); hipFree(m->d_trace_cls); hipFree(m->d_trace_cnt);
    m->d_trace_cells = nullptr; m->d_trace_cls = nullptr; m->d_trace_cnt = nullptr;
    m->trace_cap_cells = 0; m->trace_cap_counts = 0;
    HIPCHK(hipMalloc(&m->d_trace_cells, cells * 2 * sizeof(int32_t)));
    HIPCHK(hipMalloc(&m->d_trace_cls, cells));
    HIPCHK(hipMalloc(&m->d_trace_cnt, counts * sizeof(int32_t)));
    m->trace_cap_cells = cells; m->trace_cap_counts = counts;
    return GMS_OK;
}

int gms_map_trace_ray(gms_map *m, float x0, float y0, float x1, float y1, int32_t extra, int32_t *cells_xy,
                      int32_t cap, int32_t *n) {
    REQUIRE(m && n && cap >= 0 && (cells_xy || cap == 0), "gms_map_trace_ray: bad arguments");
    HIPCHK(hipSetDevice(m->device));
    int rc = ensure_trace(m, (size_t)(cap > 0 ? cap : 1), 1);
    if (rc) return rc;
    gms_launch_trace_ray(m, x0, y0, x1, y1, extra, m->d_trace_cells, cap, m->d_trace_cnt);
    HIPCHK(hipMemcpyAsync(n, m->d_trace_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    if (cap > 0)
        HIPCHK(hipMemcpyAsync(cells_xy, m->d_trace_cells, (size_t)cap * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return GMS_OK;
}

int gms_map_trace_scan(gms_map *m, const gms_beam *beams, int32_t B, const float pose[3], int32_t *cells_xy,
                       uint8_t *classes, int32_t cap, int32_t *counts) {
    REQUIRE(m && pose && counts && cap > 0 && B > 0, "gms_map_trace_scan: bad arguments");
    // only map 0's slice of the staging buffers is used
    REQUIRE(B <= m->max_beams, "beam count exceeds gms_params.max_beams");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    memcpy(m->h_beams, beams, (size_t)B * sizeof(gms_beam));
    memcpy(m->h_poses, pose, 3 * sizeof(float));
    HIPCHK(hipMemcpyAsync(m->d_beams, m->h_beams, (size_t)B * sizeof(gms_beam), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_poses, m->h_poses, 3 * sizeof(float), hipMemcpyHostToDevice, m->stream));
    int rc = ensure_trace(m, (size_t)B * cap, (size_t)B);
    if (rc) return rc;
    gms_launch_trace_scan(m, m->d_beams, B, m->d_poses, cells_xy ? m->d_trace_cells : nullptr,
                          classes ? m->d_trace_cls : nullptr, cap, m->d_trace_cnt);
    HIPCHK(hipMemcpyAsync(counts, m->d_trace_cnt, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    if (cells_xy)
        HIPCHK(hipMemcpyAsync(cells_xy, m->d_trace_cells, (size_t)B * cap * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    if (classes)
        HIPCHK(hipMemcpyAsync(classes, m->d_trace_cls, (size_t)B * cap, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return GMS_OK;
}

int gms_map_build_likelihood(gms_map *m) {                              // GridMap.java:233-250
    REQUIRE(m, "null map");
    HIPCHK(hipSetDevice(m->device));
    return finish_likelihood(m, 0);
}

// integrateObservation + computeLikelihoodMap with the scan's apply pass deferred: [ray cast | the previous scan's apply pass]
// -> dirty-tile likelihood rebuild with the new counts added on the fly.  Two launches per scan instead of three; the map
// comes out bit-identical (tests/test_gpu_parity.py, test_gpu_closed_loop.py); anything else that touches the map first
// runs the pending pass (gms_flush_apply).  Single maps in the steady state (a field to rebuild incrementally exists).
static bool can_defer_update(const gms_map *m, int32_t B) {
    return m->pair_launches && m->n_maps == 1 && B > 0 && B <= 4096 && !m->need_full_build;
}
static int deferred_update(gms_map *m, const gms_beam *dev_beams, int32_t B, int32_t beam_stride, const float *dev_poses, int32_t pose_stride) {
    if (m->apply_pending) gms_launch_raycast_apply(m, dev_beams, B, beam_stride, dev_poses, pose_stride);
    else gms_launch_raycast(m, dev_beams, B, beam_stride, dev_poses, pose_stride);
    gms_launch_likelihood(m, 1, true);
    gms_defer_apply(m);
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_map_update(gms_map *m, const gms_beam *beams, int32_t B, const float *poses) {
    if (m && can_defer_update(m, B)) {
        int rc = stage_beams(m, beams, B);
        if (!rc) rc = stage_poses(m, poses);
        return rc ? rc : deferred_update(m, m->d_beams, B, m->max_beams, m->d_poses, 3);
    }
    int rc = gms_map_integrate(m, beams, B, poses);
    if (rc) return rc;
    return finish_likelihood(m, m->need_full_build ? 0 : 1);
}

int gms_map_update_at(gms_map *m, const gms_beam *beams, int32_t B, gms_pf *pf, int32_t which) {
    int rc = gms_map_integrate_at(m, beams, B, pf, which);
    if (rc) return rc;
    return finish_likelihood(m, m->need_full_build ? 0 : 1);
}

int gms_map_update_dev(gms_map *m, const gms_beam *dev_beams, int32_t B, const float *dev_poses) {
    REQUIRE(m && dev_beams && dev_poses, "null argument");
    REQUIRE(B >= 0 && B <= m->max_beams, "beam count exceeds gms_params.max_beams");
    if (can_defer_update(m, B)) {
        HIPCHK(hipSetDevice(m->device));
        return deferred_update(m, dev_beams, B, B, dev_poses, 3);
    }
    int rc = gms_map_integrate_dev(m, dev_beams, B, dev_poses);
    if (rc) return rc;
    return finish_likelihood(m, m->need_full_build ? 0 : 1);
}

int gms_map_update_at_dev(gms_map *m, const gms_beam *dev_beams, int32_t B, gms_pf *pf, int32_t which) {
    REQUIRE(m && dev_beams && pf && pf->map == m, "gms_map_update_at_dev: bad arguments");
    REQUIRE(which == 0 || which == 1, "which must be 0 (weighted pose) or 1 (strongest particle)");
    REQUIRE(B >= 0 && B <= m->max_beams, "beam count exceeds gms_params.max_beams");
    if (can_defer_update(m, B)) {
        HIPCHK(hipSetDevice(m->device));
        return deferred_update(m, dev_beams, B, B, stats_pose_ptr(pf, which), (int32_t)(sizeof(PfStatsDev) / sizeof(float)));
    }
    int rc = gms_map_integrate_at_dev(m, dev_beams, B, pf, which);
    if (rc) return rc;
    return finish_likelihood(m, m->need_full_build ? 0 : 1);
}

int gms_map_tile_stats(gms_map *m, int32_t enable, int64_t *out4) {
    REQUIRE(m, "null map");
    HIPCHK(hipSetDevice(m->device));
    if (out4) {
        uint32_t h[64 * 4];
        HIPCHK(hipMemcpyAsync(h, m->d_tile_stats, sizeof(h), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
        for (int k = 0; k < 4; k++) out4[k] = 0;
        for (int r = 0; r < 64; r++)
            for (int k = 0; k < 4; k++) out4[k] += h[4 * r + k];
    }
    HIPCHK(hipMemsetAsync(m->d_tile_stats, 0, 64 * 4 * sizeof(uint32_t), m->stream));
    m->gd.tile_stats = enable ? m->d_tile_stats : nullptr;        // (the grid descriptor travels by value with every launch)
    return GMS_OK;
}

// development: where instrumented builds (-DGMS_STAMPS) write their stage time stamps; GMS_ERR_STATE in a product build
int gms_debug_set_stamps(gms_map *m, void *dev_buffer) {
    REQUIRE(m, "null map");
    HIPCHK(hipSetDevice(m->device));
    return gms_set_stamp_buffer(m, dev_buffer) ? GMS_OK : fail(GMS_ERR_STATE, "this build carries no stage stamps (compile with -DGMS_STAMPS)");
}

int gms_debug_f32(gms_map *m, int32_t op, const float *in, float *out, int64_t n) {
    REQUIRE(m && in && out && n > 0 && op >= 0 && op <= 5 && (op != 3 || n % 64 == 0), "gms_debug_f32: bad arguments");
    HIPCHK(hipSetDevice(m->device));
    float *d_a = nullptr, *d_o = nullptr;
    HIPCHK(hipMalloc(&d_a, n * sizeof(float)));
    HIPCHK(hipMalloc(&d_o, n * sizeof(float)));
    HIPCHK(hipMemcpyAsync(d_a, in, n * sizeof(float), hipMemcpyHostToDevice, m->stream));
    gms_launch_debug_f32(m, op, d_a, d_o, n);
    HIPCHK(hipMemcpyAsync(out, d_o, n * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    hipFree(d_a); hipFree(d_o);
    return GMS_OK;
}

// ---- ParticleFilter ---------------------------------------------------------------------------------
static int64_t nblk_of(int64_t n) { return (n + GMS_BLOCK - 1) / GMS_BLOCK; }
static int64_t nchunks_of(int64_t n) { return (n + 63) / 64; }

static void pf_free_global(gms_pf *pf) {
    hipFree(pf->d_partials); hipFree(pf->d_p2); hipFree(pf->d_global_own); hipFree(pf->d_chunk_tot); hipFree(pf->d_cum);
    pf->d_partials = pf->d_p2 = nullptr; pf->d_global = pf->d_global_own = nullptr; pf->d_chunk_tot = pf->d_cum = nullptr;
}

static int pf_alloc_global(gms_pf *pf) {
    pf_free_global(pf);
    const size_t M = pf->n_maps;
    const size_t nblk = nblk_of(pf->n_global), nch = nchunks_of(pf->n_global);
    HIPCHK(hipMalloc(&pf->d_partials, M * nblk * GMS_PARTIAL_STRIDE * sizeof(double)));
    HIPCHK(hipMalloc(&pf->d_p2, M * nblk * 2 * sizeof(double)));
    HIPCHK(hipMalloc(&pf->d_global_own, M * pf->n_global * sizeof(PackedParticle)));
    pf->d_global = pf->d_global_own;
    HIPCHK(hipMalloc(&pf->d_chunk_tot, M * (nch + 1 + nch * 8) * sizeof(double)));     // chunk totals of all maps, then every chunk's eight octet boundaries (chunk_sub_of)
    HIPCHK(hipMalloc(&pf->d_cum, M * pf->n_global * sizeof(double)));
    return GMS_OK;
}

int gms_pf_destroy(gms_pf *pf) {
    if (!pf) return GMS_OK;
    hipSetDevice(pf->map->device);
    hipStreamSynchronize(pf->map->stream);
    pf->map->n_filters--;
    hipFree(pf->d_pose); hipFree(pf->d_pose2); hipFree(pf->d_part); hipFree(pf->d_cs2);
    hipFree(pf->d_w); hipFree(pf->d_w2); hipFree(pf->d_logw); hipFree(pf->d_cs); hipFree(pf->d_hitbeams);
    hipFree(pf->d_nhit); hipFree(pf->d_stats); hipFree(pf->d_r01); hipFree(pf->d_idx);
    hipFree(pf->d_ord); hipFree(pf->d_perm);
    pf_free_global(pf);
    if (pf->h_stats) hipHostFree(pf->h_stats);
    if (pf->h_stage) hipHostFree(pf->h_stage);
    ring_free(pf->pose_ring);
    ring_free(pf->r01_ring);
    delete pf;
    return GMS_OK;
}

int gms_pf_create(gms_map *m, int32_t n, gms_pf **out) {               // ParticleFilter.java:43, SLAM.java:65-77
    REQUIRE(m && out, "gms_pf_create: null argument");
    *out = nullptr;
    REQUIRE(n >= 1 && n <= GMS_MAX_PARTICLES, "gms_pf_create: particle count out of range (1 .. GMS_MAX_PARTICLES)");
    HIPCHK(hipSetDevice(m->device));
    gms_pf *pf = new (std::nothrow) gms_pf();
    if (!pf) return fail(GMS_ERR_NOMEM, "out of host memory");
    pf->map = m; pf->n = n; pf->offset = 0; pf->n_global = n; pf->n_maps = m->n_maps;
    m->n_filters++;
    const size_t T = (size_t)n * m->n_maps;
    bool ok = true;
    ok = ok && hipMalloc(&pf->d_pose, T * 12) == hipSuccess && hipMalloc(&pf->d_pose2, T * 12) == hipSuccess;
    ok = ok && hipMalloc(&pf->d_w, T * 8) == hipSuccess && hipMalloc(&pf->d_w2, T * 8) == hipSuccess;
    ok = ok && hipMalloc(&pf->d_logw, T * 8) == hipSuccess && hipMalloc(&pf->d_cs, T * 8) == hipSuccess;
    ok = ok && hipMalloc(&pf->d_cs2, T * 8) == hipSuccess;
    ok = ok && hipMalloc(&pf->d_hitbeams, (size_t)m->n_maps * m->max_beams * 16) == hipSuccess;
    ok = ok && hipMalloc(&pf->d_nhit, (size_t)m->n_maps * 4) == hipSuccess;
    ok = ok && hipMalloc(&pf->d_part, T * GMS_SCORE_MAXSEG * 8) == hipSuccess;
    ok = ok && hipMalloc(&pf->d_stats, (size_t)m->n_maps * 2 * sizeof(PfStatsDev)) == hipSuccess;
    ok = ok && hipMalloc(&pf->d_r01, (size_t)m->n_maps * 8 + 16) == hipSuccess;          // (+16: copied in 16-byte units)
    ok = ok && ring_alloc(pf->r01_ring, (size_t)m->n_maps * 8 + 16) == GMS_OK;
    ok = ok && hipMalloc(&pf->d_idx, T * 4) == hipSuccess;
    ok = ok && hipMalloc(&pf->d_ord, T * sizeof(float4)) == hipSuccess && hipMalloc(&pf->d_perm, T * 4) == hipSuccess;
    pf->order_mode = -1;
    if (const char *v = getenv("GMS_SCORE_ORDER")) pf->order_mode = atoi(v);
    pf->score_threads = 0;
    if (const char *v = getenv("GMS_SCORE_THREADS")) pf->score_threads = atoi(v);
    pf->score_spread = -1;
    if (const char *v = getenv("GMS_SCORE_SPREAD")) pf->score_spread = atoi(v) != 0;
    ok = ok && hipHostMalloc(&pf->h_stats, (size_t)m->n_maps * sizeof(PfStatsDev) + (size_t)m->n_maps * 8) == hipSuccess;
    ok = ok && hipHostMalloc(&pf->h_stage, T * 3 * sizeof(float)) == hipSuccess;
    ok = ok && ring_alloc(pf->pose_ring, T * 3 * sizeof(float)) == GMS_OK;
    if (!ok || pf_alloc_global(pf) != GMS_OK) { gms_pf_destroy(pf); return fail(GMS_ERR_NOMEM, "device allocation failed for %d particles", n); }
    hipMemsetAsync(pf->d_stats, 0, (size_t)m->n_maps * 2 * sizeof(PfStatsDev), m->stream);
    gms_launch_pf_init(pf);
    HIPCHK(hipGetLastError());
    *out = pf;
    return GMS_OK;
}

int gms_pf_set_shard(gms_pf *pf, int64_t offset, int64_t n_global) {
    REQUIRE(pf, "null filter");
    if (pf->slam_owned && !pf->d_epoch2)
        return fail(GMS_ERR_STATE, "this filter's particles own maps (gms_slam): resample through gms_slam_resample_maps[_if], update through gms_slam_update_per_particle");
    REQUIRE(offset >= 0 && offset % GMS_BLOCK == 0, "shard offset must be a multiple of GMS_BLOCK");
    REQUIRE(n_global >= offset + pf->n && n_global <= GMS_MAX_PARTICLES, "shard does not fit n_global (at most GMS_MAX_PARTICLES in all)");
    HIPCHK(hipSetDevice(pf->map->device));
    HIPCHK(hipStreamSynchronize(pf->map->stream));
    pf->offset = offset;
    pf->n_global = n_global;
    pf->have_global = 0;
    if (offset != 0 || n_global != pf->n) { pf->log_norm = 0; pf->reference_order = 0; }      // (stand-alone filters only)
    return pf_alloc_global(pf);
}

// host poses -> a pinned ring slot -> the filter (k_pose_trig reads the slot over PCIe: copy + trig in one launch; no
// stream synchronisation, no hipMemcpyAsync)
static int upload_poses(gms_pf *pf, const float *xytheta) {
    gms_map *m = pf->map;
    const size_t bytes = (size_t)pf->n * pf->n_maps * 3 * sizeof(float);
    void *slot = nullptr;
    int rc = ring_acquire(pf->pose_ring, &slot);
    if (rc) return rc;
    memcpy(slot, xytheta, bytes);
    gms_launch_pf_pose_trig(pf, static_cast<const float *>(slot));
    return ring_commit(pf->pose_ring, m->stream);
}

int gms_pf_set_poses(gms_pf *pf, const float *xytheta) {
    REQUIRE(pf && xytheta, "null argument");
    gms_map *m = pf->map;
    HIPCHK(hipSetDevice(m->device));
    int rc = upload_poses(pf, xytheta);
    if (rc) return rc;
    pf->have_global = 0;
    pf->stats_current = 0;
    return GMS_OK;
}

int gms_pf_get_poses(gms_pf *pf, float *xytheta) {
    REQUIRE(pf && xytheta, "null argument");
    gms_map *m = pf->map;
    HIPCHK(hipSetDevice(m->device));
    const size_t bytes = (size_t)pf->n * pf->n_maps * 3 * sizeof(float);
    HIPCHK(hipMemcpyAsync(pf->h_stage, pf->d_pose, bytes, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    memcpy(xytheta, pf->h_stage, bytes);
    return GMS_OK;
}

static int pf_copy_f64(gms_pf *pf, double *dev, double *host, bool to_device) {
    gms_map *m = pf->map;
    HIPCHK(hipSetDevice(m->device));
    const size_t bytes = (size_t)pf->n * pf->n_maps * sizeof(double);
    if (to_device) HIPCHK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, m->stream));
    else HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return GMS_OK;
}

int gms_pf_set_weights(gms_pf *pf, const double *w) {
    REQUIRE(pf && w, "null argument");
    pf->pending_nseg = 0;
    pf->have_global = 0;
    pf->stats_current = 0;
    pf->score_fresh = 0;                // d_w no longer belongs to d_logw: a log-normalising filter takes these weights as they are
    return pf_copy_f64(pf, pf->d_w, const_cast<double *>(w), true);
}
int gms_pf_get_weights(gms_pf *pf, double *w) {
    REQUIRE(pf && w, "null argument");
    HIPCHK(hipSetDevice(pf->map->device));
    gms_launch_pf_combine(pf);
    return pf_copy_f64(pf, pf->d_w, w, false);
}
int gms_pf_get_log_weights(gms_pf *pf, double *lw) {
    REQUIRE(pf && lw, "null argument");
    HIPCHK(hipSetDevice(pf->map->device));
    gms_launch_pf_combine(pf);
    return pf_copy_f64(pf, pf->d_logw, lw, false);
}

int gms_pf_score(gms_pf *pf, const gms_beam *beams, int32_t B) {       // GridMap.java:261-294 x N
    REQUIRE(pf, "null filter");
    gms_map *m = pf->map;
    int rc = stage_beams(m, beams, B);
    if (rc) return rc;
    gms_launch_pf_score(pf, m->d_beams, B, m->max_beams);
    pf->have_global = 0;
    pf->stats_current = 0;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

// poses := dev_xytheta (may be NULL: keep the current ones), then weights: one launch with the default scoring kernel.
// refine: the poses are replaced by findBestPose's argmax first (SLAM.java:96-97), which takes launches of its own.
static int set_poses_and_score_dev(gms_pf *pf, const float *dev_xytheta, const gms_beam *dev_beams, int32_t B, bool refine = false,
                                   const MotionModel *motion = nullptr) {
    REQUIRE(pf && dev_beams, "null argument");
    gms_map *m = pf->map;
    REQUIRE(B >= 0 && B <= m->max_beams, "beam count exceeds gms_params.max_beams");
    HIPCHK(hipSetDevice(m->device));
    if (refine) {
        if (dev_xytheta) gms_launch_pf_pose_trig(pf, dev_xytheta);                // SLAM.java:90
        if (motion) gms_launch_pf_motion(pf, motion->d_center, motion->d_theta, motion->seed, motion->sequence);
        gms_launch_pf_refine(pf, dev_beams, B, B);                                // :96-97
        dev_xytheta = nullptr;
        motion = nullptr;
    }
    gms_launch_pf_score(pf, dev_beams, B, B, dev_xytheta, motion);                // (a motion-model sample rides in the scoring launch)
    pf->have_global = 0;
    pf->stats_current = 0;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_pf_score_dev(gms_pf *pf, const gms_beam *dev_beams, int32_t B) {
    return set_poses_and_score_dev(pf, nullptr, dev_beams, B);
}

int gms_pf_set_poses_dev(gms_pf *pf, const float *dev_xytheta) {
    REQUIRE(pf && dev_xytheta, "null argument");
    HIPCHK(hipSetDevice(pf->map->device));
    gms_launch_pf_pose_trig(pf, dev_xytheta);          // copy + the per-particle trig, one launch
    pf->have_global = 0;
    pf->stats_current = 0;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

static void fill_stats(const gms_pf *pf, gms_pf_stats *stats) {
    for (int32_t mi = 0; mi < pf->n_maps; mi++) {
        const PfStatsDev &s = pf->h_stats[mi];
        stats[mi].weight_sum = s.weight_sum;
        stats[mi].neff = 1.0 / s.sq_sum;                                // SLAM.java:189
        stats[mi].strongest = s.strongest;
        stats[mi].n_zero = s.n_zero;
        stats[mi].max_log_weight = s.max_logw;
    }
}

static int pull_stats(gms_pf *pf) {
    gms_map *m = pf->map;
    HIPCHK(hipMemcpyAsync(pf->h_stats, pf->d_stats, (size_t)pf->n_maps * sizeof(PfStatsDev), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return GMS_OK;
}

int gms_pf_get_stats(gms_pf *pf, gms_pf_stats *stats) {
    REQUIRE(pf && stats, "null argument");
    HIPCHK(hipSetDevice(pf->map->device));
    if (pf->have_global) gms_launch_pf_fold_neff(pf);      // calculateNeff from the normalised population
    int rc = pull_stats(pf);
    if (rc) return rc;
    fill_stats(pf, stats);
    return GMS_OK;
}

int gms_pf_normalize(gms_pf *pf, gms_pf_stats *stats) {                 // SLAM.java:87-129
    REQUIRE(pf, "null filter");
    if (pf->offset != 0 || pf->n_global != pf->n)
        return fail(GMS_ERR_STATE, "sharded filter: use gms_pf_local_partials / apply_partials / import_global");
    HIPCHK(hipSetDevice(pf->map->device));
    if (pf->reference_order) {                         // the audit path: one lane, the reference's own loops
        gms_launch_pf_combine(pf);
        gms_launch_pf_normalize_seq(pf, pf->d_stats, true);
        pf->have_global = 0;
        pf->stats_current = 1;
        pf->score_fresh = 0;
        HIPCHK(hipGetLastError());
        if (stats) { int rc_ = pull_stats(pf); if (rc_) return rc_; fill_stats(pf, stats); }
        return GMS_OK;
    }
    pf->d_global = pf->d_global_own;
    gms_launch_pf_partials(pf, pf->d_partials);
    gms_launch_pf_apply_partials(pf, pf->d_partials, pf->d_global, true);
    pf->have_global = 1;
    pf->stats_current = 1;
    HIPCHK(hipGetLastError());
    if (stats) return gms_pf_get_stats(pf, stats);
    return GMS_OK;
}

int gms_pf_partials_len(const gms_pf *pf, int64_t *n_doubles) {
    REQUIRE(pf && n_doubles, "null argument");
    *n_doubles = (int64_t)pf->n_maps * nblk_of(pf->n_global) * GMS_PARTIAL_STRIDE;
    return GMS_OK;
}

int gms_pf_local_partials(gms_pf *pf, double *dev_partials) {
    REQUIRE(pf && dev_partials, "null argument");
    HIPCHK(hipSetDevice(pf->map->device));
    gms_launch_pf_partials(pf, dev_partials);
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_pf_apply_partials(gms_pf *pf, const double *dev_partials, void *dev_packed) {
    REQUIRE(pf && dev_partials && dev_packed, "null argument");
    HIPCHK(hipSetDevice(pf->map->device));
    gms_launch_pf_apply_partials(pf, dev_partials, reinterpret_cast<PackedParticle *>(dev_packed), false);
    pf->have_global = 0;
    pf->stats_current = 1;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_pf_stats_from_partials(gms_pf *pf, const double *dev_partials) {
    REQUIRE(pf && dev_partials, "null argument");
    HIPCHK(hipSetDevice(pf->map->device));
    gms_launch_pf_stats_only(pf, dev_partials, pf->d_stats + pf->n_maps);
    pf->stats_current = 2;        // current-particle statistics live in the second slot
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_pf_pack(gms_pf *pf, void *dev_packed) {
    REQUIRE(pf && dev_packed, "null argument");
    HIPCHK(hipSetDevice(pf->map->device));
    gms_launch_pf_combine(pf);
    gms_launch_pf_pack(pf, reinterpret_cast<PackedParticle *>(dev_packed));
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_pf_import_global(gms_pf *pf, const void *dev_packed_global) {
    REQUIRE(pf && dev_packed_global, "null argument");
    gms_map *m = pf->map;
    HIPCHK(hipSetDevice(m->device));
    // zero copy: the gathered buffer becomes the resampling source as it is (see the header for its lifetime)
    pf->d_global = const_cast<PackedParticle *>(reinterpret_cast<const PackedParticle *>(dev_packed_global));
    gms_launch_pf_after_gather(pf);
    pf->have_global = 1;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

// make d_global describe the current particles (stand-alone filters only)
static int ensure_global(gms_pf *pf) {
    if (pf->have_global) return GMS_OK;
    if (pf->offset != 0 || pf->n_global != pf->n)
        return fail(GMS_ERR_STATE, "sharded filter: all-gather the packed particles and call gms_pf_import_global first");
    pf->d_global = pf->d_global_own;
    gms_launch_pf_combine(pf);
    gms_launch_pf_pack(pf, pf->d_global);
    pf->have_global = 1;
    return GMS_OK;
}

int gms_pf_weighted_pose(gms_pf *pf, float *out) {                      // SLAM.java:165-178
    REQUIRE(pf && out, "null argument");
    gms_map *m = pf->map;
    HIPCHK(hipSetDevice(m->device));
    if (!pf->stats_current) {
        // the particles changed since the last normalise (resample, set_poses, ...): the reference
        // recomputes the pose from whatever the particles are now (J/app/GridMapApp.java:192)
        if (pf->offset != 0 || pf->n_global != pf->n)
            return fail(GMS_ERR_STATE, "sharded filter: gms_pf_local_partials -> all-reduce -> gms_pf_stats_from_partials first");
        if (pf->reference_order) {
            gms_launch_pf_combine(pf);
            gms_launch_pf_normalize_seq(pf, pf->d_stats + pf->n_maps, false);
        } else {
            gms_launch_pf_partials(pf, pf->d_partials);
            gms_launch_pf_stats_only(pf, pf->d_partials, pf->d_stats + pf->n_maps);
        }
        pf->stats_current = 2;
    }
    const PfStatsDev *src = pf->d_stats + (pf->stats_current == 2 ? pf->n_maps : 0);
    HIPCHK(hipStreamSynchronize(m->stream));
    PfStatsDev *tmp = pf->h_stats;      // staging; the cached normalise statistics are re-pulled on demand
    HIPCHK(hipMemcpyAsync(tmp, src, (size_t)pf->n_maps * sizeof(PfStatsDev), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    for (int32_t mi = 0; mi < pf->n_maps; mi++)
        for (int k = 0; k < 3; k++) out[3 * mi + k] = tmp[mi].wpose[k];
    return GMS_OK;
}

// Math.random() of SLAM.java:136, one draw per map, to where the resample kernel reads it.  Batched handles: the draws
// go into a pinned ring slot that the kernel reads in place (one 8-byte read per workgroup over PCIe, hidden beside the
// likelihood tiles; a copy kernel in front of it was a 16 us PCIe round trip on the stream's critical path).  The slot is
// released by commit_r01 once the resample launch is enqueued.
static int stage_r01(gms_pf *pf, const double *r01) {
    if (pf->n_maps == 1) {
        pf->r01_scalar = r01[0];                       // travels as a kernel argument: no copy, no synchronisation
        return GMS_OK;
    }
    void *slot = nullptr;
    int rc = ring_acquire(pf->r01_ring, &slot);
    if (rc) return rc;
    memcpy(slot, r01, (size_t)pf->n_maps * sizeof(double));
    pf->d_r01_src = static_cast<const double *>(slot);
    return GMS_OK;
}
static int commit_r01(gms_pf *pf) {
    if (pf->n_maps == 1) return GMS_OK;
    return ring_commit(pf->r01_ring, pf->map->stream);
}

static int do_resample(gms_pf *pf, const double *r01, double fraction, int32_t *indices, int32_t *n_ambiguous) {
    REQUIRE(pf && r01, "null argument");
    if (pf->slam_owned && !pf->d_epoch2)
        return fail(GMS_ERR_STATE, "this filter's particles own maps (gms_slam): resample through gms_slam_resample_maps[_if], update through gms_slam_update_per_particle");
    gms_map *m = pf->map;
    HIPCHK(hipSetDevice(m->device));
    int rc = GMS_OK;
    if (!pf->reference_order) rc = ensure_global(pf);
    else gms_launch_pf_combine(pf);
    if (rc) return rc;
    rc = stage_r01(pf, r01);
    if (rc) return rc;
    if (pf->reference_order) gms_launch_pf_resample_seq(pf, fraction);      // the audit path: the reference's own loop on one lane
    else gms_launch_pf_resample(pf, fraction);
    rc = commit_r01(pf);
    if (rc) return rc;
    std::swap(pf->d_pose, pf->d_pose2); std::swap(pf->d_cs, pf->d_cs2); std::swap(pf->d_w, pf->d_w2);
    pf->have_global = 0;
    pf->stats_current = 0;
    pf->score_fresh = 0;                // the copies' weights are not the log-weights' (d_logw is not permuted): never rescale them from those
    HIPCHK(hipGetLastError());
    if (indices) {
        HIPCHK(hipMemcpyAsync(indices, pf->d_idx, (size_t)pf->n * pf->n_maps * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
    }
    if (n_ambiguous) {
        rc = pull_stats(pf);
        if (rc) return rc;
        for (int32_t mi = 0; mi < pf->n_maps; mi++) n_ambiguous[mi] = pf->h_stats[mi].n_ambiguous;
    }
    return GMS_OK;
}

int gms_pf_resample(gms_pf *pf, const double *r01, int32_t *indices, int32_t *n_ambiguous) {   // SLAM.java:133-153
    return do_resample(pf, r01, -1.0, indices, n_ambiguous);
}

int gms_pf_resample_if(gms_pf *pf, const double *r01, double fraction) {   // GridMapApp.java:185-186
    REQUIRE(fraction >= 0.0, "fraction must be non-negative");
    return do_resample(pf, r01, fraction, nullptr, nullptr);
}

// the tail of a paired scan step: dirty-tile likelihood rebuild beside the conditional resample.  The scan's counts
// stay un-applied (the likelihood pass adds them on the fly): the apply pass runs beside the next step's weight
// reduction, or when anything else touches the map (gms_flush_apply).
static int paired_likelihood_resample(gms_pf *pf, const double *r01, double fraction) {
    gms_map *m = pf->map;
    if (fraction >= 0.0) {
        int rc = stage_r01(pf, r01);
        if (rc) return rc;
        gms_launch_lik_resample(pf, fraction);
        rc = commit_r01(pf);
        if (rc) return rc;
        std::swap(pf->d_pose, pf->d_pose2); std::swap(pf->d_cs, pf->d_cs2); std::swap(pf->d_w, pf->d_w2);
        pf->have_global = 0;
        pf->stats_current = 0;
        pf->score_fresh = 0;
        gms_defer_apply(m);
    } else {                                // no resample to pair with: the immediate protocol
        gms_launch_apply_counts(m);
        return finish_likelihood(m, 1);
    }
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

// SLAM.update(z, u) (SLAM.java:80-131) + the resampling rule of its caller (GridMapApp.java:185-186) as one
// call on device-resident inputs: poses := dev_xytheta (the motion-model samples), weights, bookkeeping,
// conditional resample, map update at the weighted pose, likelihood rebuild.  Nothing is read back.
static int slam_update_impl(gms_pf *pf, const float *dev_xytheta, const MotionModel *motion, const gms_beam *dev_beams, int32_t B,
                            const double *r01, double resample_fraction, int32_t integrate);

int gms_slam_update_dev(gms_pf *pf, const float *dev_xytheta, const gms_beam *dev_beams, int32_t B, const double *r01,
                        double resample_fraction, int32_t integrate) {
    return slam_update_impl(pf, dev_xytheta, nullptr, dev_beams, B, r01, resample_fraction, integrate);
}

// SLAM.update(z, u) with its motion-model sample inside (SLAM.java:80-131, :90 included) + the caller's resampling rule: the
// particles move by Odometry.apply (as gms_pf_sample_motion with the same seed and sequence: same Philox counters, same bits) on
// their way into the scoring launch.  Four launches, as gms_slam_update_dev.
int gms_slam_update_u_dev(gms_pf *pf, double d_center, double d_theta, uint64_t seed, uint64_t sequence, const gms_beam *dev_beams,
                          int32_t B, const double *r01, double resample_fraction, int32_t integrate) {
    MotionModel mo;
    mo.d_center = d_center; mo.d_theta = d_theta; mo.seed = seed; mo.sequence = sequence;
    return slam_update_impl(pf, nullptr, &mo, dev_beams, B, r01, resample_fraction, integrate);
}

static int slam_update_impl(gms_pf *pf, const float *dev_xytheta, const MotionModel *motion, const gms_beam *dev_beams, int32_t B,
                            const double *r01, double resample_fraction, int32_t integrate) {
    REQUIRE(pf && dev_beams && r01, "null argument");
    if (pf->slam_owned && !pf->d_epoch2)
        return fail(GMS_ERR_STATE, "this filter's particles own maps (gms_slam): resample through gms_slam_resample_maps[_if], update through gms_slam_update_per_particle");
    gms_map *m = pf->map;
    if (pf->offset != 0 || pf->n_global != pf->n)
        return fail(GMS_ERR_STATE, "sharded filter: the collectives belong to the caller (see distributed.py)");
    int rc = GMS_OK;
    rc = set_poses_and_score_dev(pf, dev_xytheta, dev_beams, B, pf->refine != 0, motion);   // SLAM.java:90, :96-97, :99
    if (!rc && integrate && !pf->reference_order && gms_can_pair_launches(pf, B)) {
        // The weight branch and the map branch are independent once the partials exist: they share launches
        // (gms_fused_kernels.hip).  (Two streams were measured: the event fork/join costs more than it hides.)
        pf->d_global = pf->d_global_own;
        gms_launch_partials_apply(pf, pf->d_partials, true);                              // :100-115
        gms_launch_norm_raycast(pf, pf->d_partials, pf->d_global, true, dev_beams, B);    // :120-124 | :93 | previous scan's :223
        pf->have_global = 1;
        pf->stats_current = 1;
        return paired_likelihood_resample(pf, r01, resample_fraction);           // :105 | GridMapApp.java:185-186
    }
    if (!rc && integrate && !pf->reference_order && pf->n_maps > 1 && B > 0 && !m->need_full_build && m->pair_launches) {
        // batched maps: the ray cast runs 16 rays per workgroup (1024 threads), so only the other two pairs apply:
        // [partials | previous apply] -> normalise -> ray cast -> [likelihood | resample]
        pf->d_global = pf->d_global_own;
        const bool ride = gms_raycast_tiled(m, B);     // the tiled ray cast takes the pending apply pass along
        gms_launch_partials_apply(pf, pf->d_partials, ride);
        gms_launch_pf_apply_partials(pf, pf->d_partials, pf->d_global, true);
        pf->have_global = 1;
        pf->stats_current = 1;
        gms_launch_raycast(m, dev_beams, B, B, stats_pose_ptr(pf, 0), (int32_t)(sizeof(PfStatsDev) / sizeof(float)), ride);
        return paired_likelihood_resample(pf, r01, resample_fraction);
    }
    if (!rc) rc = gms_pf_normalize(pf, nullptr);                                 // :100-124
    if (!rc && resample_fraction >= 0.0) rc = gms_pf_resample_if(pf, r01, resample_fraction);   // GridMapApp.java:185-186
    if (!rc && integrate) rc = gms_map_update_at_dev(m, dev_beams, B, pf, 0);    // SLAM.java:102-105, :93
    return rc;
}

// One recorded revolution, as GridMapApp.onHandleData treats it (J/app/GridMapApp.java:133-192): de-skew the raw measurements with
// the frame's odometry (:143-175), sample the motion model for every particle (SLAM.java:90), SLAM.update and the conditional
// resample (:87-131, GridMapApp.java:185-186).  What gms_map_deskew + gms_pf_sample_motion + gms_slam_update_dev do in three
// calls and six launches, as one call and five: the de-skew and the motion model are independent and share a launch.
int gms_slam_frame(gms_pf *pf, const double *angle, const double *distance, const uint8_t *hit, int32_t length, double d_center,
                   double d_theta, uint64_t seed, uint64_t sequence, const double *r01, double resample_fraction, int32_t integrate) {
    REQUIRE(pf && angle && distance && hit && r01, "null argument");
    gms_map *m = pf->map;
    if (pf->n_maps != 1) return fail(GMS_ERR_STATE, "gms_slam_frame: one map per handle (a frame is one robot's revolution)");
    if (pf->offset != 0 || pf->n_global != pf->n)
        return fail(GMS_ERR_STATE, "sharded filter: the collectives belong to the caller (see distributed.py)");
    REQUIRE(length > 0 && length <= m->max_beams, "measurement count exceeds gms_params.max_beams");
    REQUIRE((size_t)length * 17 + 16 <= (size_t)m->n_maps * m->max_beams * sizeof(gms_beam), "scan too long for the staging buffer");
    HIPCHK(hipSetDevice(m->device));
    void *slot = nullptr;
    int rc = ring_acquire(m->beam_ring, &slot);                        // the raw scan is read in place over PCIe (see gms_map_deskew)
    if (rc) return rc;
    double *h_a = static_cast<double *>(slot), *h_d = h_a + length;
    uint8_t *h_h = reinterpret_cast<uint8_t *>(h_d + length);
    memcpy(h_a, angle, (size_t)length * 8); memcpy(h_d, distance, (size_t)length * 8); memcpy(h_h, hit, (size_t)length);
    gms_launch_deskew_motion(pf, h_a, h_d, h_h, length, d_center, d_theta, seed, sequence);
    rc = ring_commit(m->beam_ring, m->stream);
    if (rc) return rc;
    pf->have_global = 0;
    pf->stats_current = 0;
    HIPCHK(hipGetLastError());
    return gms_slam_update_dev(pf, nullptr, m->d_beams, length, r01, resample_fraction, integrate);
}

// The same with host-resident inputs (what a JNI caller has): one staging copy of the scan, one of the poses.
int gms_slam_update(gms_pf *pf, const float *xytheta, const gms_beam *beams, int32_t B, const double *r01,
                    double resample_fraction, int32_t integrate, gms_pf_stats *stats) {
    REQUIRE(pf && beams && r01, "null argument");
    gms_map *m = pf->map;
    if (pf->offset != 0 || pf->n_global != pf->n)
        return fail(GMS_ERR_STATE, "sharded filter: the collectives belong to the caller (see distributed.py)");
    int rc = GMS_OK;
    const int32_t stride_ok = (m->n_maps == 1) || (B == m->max_beams);
    if (xytheta) rc = gms_pf_set_poses(pf, xytheta);
    if (!rc) rc = stage_beams(m, beams, B);
    const gms_beam *d = m->d_beams;
    if (!rc && !stride_ok) {
        // batched handles stage [n_maps][max_beams]; the *_dev entry points expect [n_maps][B]: go through the
        // staging-stride launchers instead
        if (pf->refine) gms_launch_pf_refine(pf, m->d_beams, B, m->max_beams);
        gms_launch_pf_score(pf, m->d_beams, B, m->max_beams);
        pf->have_global = 0; pf->stats_current = 0;
        rc = gms_pf_normalize(pf, nullptr);
        if (!rc && resample_fraction >= 0.0) rc = gms_pf_resample_if(pf, r01, resample_fraction);
        if (!rc && integrate) {
            gms_ensure_lik(m);
            gms_launch_raycast(m, m->d_beams, B, m->max_beams, stats_pose_ptr(pf, 0), (int32_t)(sizeof(PfStatsDev) / sizeof(float)));
            gms_launch_apply_counts(m);
            rc = finish_likelihood(m, m->need_full_build ? 0 : 1);
        }
    } else if (!rc) {
        rc = gms_slam_update_dev(pf, nullptr, d, B, r01, resample_fraction, integrate);
    }
    if (!rc && stats) rc = gms_pf_get_stats(pf, stats);
    return rc;
}

int gms_pf_sample_motion(gms_pf *pf, double d_center, double d_theta, uint64_t seed, uint64_t sequence) {   // Odometry.java:77-96
    REQUIRE(pf, "null filter");
    HIPCHK(hipSetDevice(pf->map->device));
    gms_launch_pf_motion(pf, d_center, d_theta, seed, sequence);
    pf->have_global = 0;
    pf->stats_current = 0;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_pf_last_resample_indices(gms_pf *pf, int32_t *indices) {
    REQUIRE(pf && indices, "null argument");
    gms_map *m = pf->map;
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipMemcpyAsync(indices, pf->d_idx, (size_t)pf->n * pf->n_maps * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return GMS_OK;
}

int gms_pf_count(const gms_pf *pf, int32_t *n, int32_t *n_maps, int64_t *n_global) {
    REQUIRE(pf, "null filter");
    if (n) *n = pf->n;
    if (n_maps) *n_maps = pf->n_maps;
    if (n_global) *n_global = pf->n_global;
    return GMS_OK;
}

int gms_pf_did_resample(gms_pf *pf, int32_t *flags) {
    REQUIRE(pf && flags, "null argument");
    HIPCHK(hipSetDevice(pf->map->device));
    int rc = pull_stats(pf);
    if (rc) return rc;
    for (int32_t mi = 0; mi < pf->n_maps; mi++) flags[mi] = pf->h_stats[mi].did_resample;
    return GMS_OK;
}

int gms_pf_set_log_normalize(gms_pf *pf, int32_t on) {
    REQUIRE(pf, "null filter");
    if (on && (pf->offset != 0 || pf->n_global != pf->n))
        return fail(GMS_ERR_STATE, "gms_pf_set_log_normalize: stand-alone filters only (a shard does not know the other shards' largest log-weight)");
    if (on && pf->reference_order) return fail(GMS_ERR_STATE, "gms_pf_set_log_normalize: the reference-order audit path is on (gms_pf_set_reference_order)");
    pf->log_norm = on ? 1 : 0;
    return GMS_OK;
}

int gms_pf_set_reference_order(gms_pf *pf, int32_t on) {
    REQUIRE(pf, "null filter");
    if (on && (pf->offset != 0 || pf->n_global != pf->n))
        return fail(GMS_ERR_STATE, "gms_pf_set_reference_order: stand-alone filters only (the audit path adds up on ONE lane of ONE device)");
    if (on && pf->log_norm) return fail(GMS_ERR_STATE, "gms_pf_set_reference_order: log-normalisation is on (not the reference's arithmetic): turn it off first");
    pf->reference_order = on ? 1 : 0;
    pf->have_global = 0;
    pf->stats_current = 0;
    return GMS_OK;
}

int gms_pf_set_refine(gms_pf *pf, int32_t on) {                          // SLAM.java:96-97
    REQUIRE(pf, "null filter");
    pf->refine = on != 0;
    return GMS_OK;
}

int gms_pf_last_step(gms_pf *pf, float *weighted_pose, float *strongest_pose, int32_t *did_resample, int32_t *n_ambiguous) {
    REQUIRE(pf, "null filter");
    HIPCHK(hipSetDevice(pf->map->device));
    int rc = pull_stats(pf);
    if (rc) return rc;
    for (int32_t mi = 0; mi < pf->n_maps; mi++) {
        const PfStatsDev &s = pf->h_stats[mi];
        for (int k = 0; k < 3; k++) {
            if (weighted_pose) weighted_pose[3 * mi + k] = s.wpose[k];
            if (strongest_pose) strongest_pose[3 * mi + k] = s.spose[k];
        }
        if (did_resample) did_resample[mi] = s.did_resample;
        if (n_ambiguous) n_ambiguous[mi] = s.n_ambiguous;
    }
    return GMS_OK;
}

int gms_pf_refine_poses(gms_pf *pf, const gms_beam *beams, int32_t B) {   // GridMap.java:319-346
    REQUIRE(pf, "null filter");
    gms_map *m = pf->map;
    int rc = stage_beams(m, beams, B);
    if (rc) return rc;
    gms_launch_pf_refine(pf, m->d_beams, B, m->max_beams);
    pf->have_global = 0;
    pf->stats_current = 0;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

// ---------------------------------------------------------------------------------------------
// Sharded filters over RCCL, one process per GPU.  The particle population is split by global index,
// the map is replicated; a scan step needs two exchanges (DESIGN.md "multi-GPU"):
//   1. all-reduce(SUM) of the block partials: every slot is non-zero on exactly one rank, so the sum is exact and
//      every rank folds the same vector in the same order as a stand-alone filter does;
//   2. all-gather of the packed normalised particles {w,x,y,theta}: the resampling source.  It runs on the
//      communicator's side stream while the map update runs on the handle's stream.
// RCCL is bound at run time (no link-time dependency: a single-GPU consumer never loads it).
// ---------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } rccl_unique_id;            // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES)
enum { RCCL_INT8 = 0, RCCL_FLOAT64 = 8, RCCL_SUM = 0 };           // ncclDataType_t / ncclRedOp_t values

static struct {
    void *dl;
    int (*GetUniqueId)(rccl_unique_id *);
    int (*CommInitRank)(void **, int, rccl_unique_id, int);
    int (*CommDestroy)(void *);
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t);
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t);
    const char *(*GetErrorString)(int);
} g_rccl;

#define RCCLCHK(expr)                                                                                              \
    do {                                                                                                           \
        int r_ = (expr);                                                                                           \
        if (r_ != 0) return fail(GMS_ERR_HIP, "%s: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); \
    } while (0)

int gms_comm_load(const char *librccl_path) {
    if (g_rccl.dl) return GMS_OK;
    void *dl = nullptr;
    if (librccl_path && *librccl_path) {
        dl = dlopen(librccl_path, RTLD_NOW | RTLD_GLOBAL);
    } else {
        // a copy that the process already holds (PyTorch bundles one) first: one RCCL per process
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *nm : names) if (!dl) dl = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char *nm : names) if (!dl) dl = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!dl) return fail(GMS_ERR_STATE, "RCCL not found: %s", dlerror());
    *(void **)&g_rccl.GetUniqueId = dlsym(dl, "ncclGetUniqueId");
    *(void **)&g_rccl.CommInitRank = dlsym(dl, "ncclCommInitRank");
    *(void **)&g_rccl.CommDestroy = dlsym(dl, "ncclCommDestroy");
    *(void **)&g_rccl.AllReduce = dlsym(dl, "ncclAllReduce");
    *(void **)&g_rccl.AllGather = dlsym(dl, "ncclAllGather");
    *(void **)&g_rccl.GroupStart = dlsym(dl, "ncclGroupStart");
    *(void **)&g_rccl.GroupEnd = dlsym(dl, "ncclGroupEnd");
    *(void **)&g_rccl.Send = dlsym(dl, "ncclSend");
    *(void **)&g_rccl.Recv = dlsym(dl, "ncclRecv");
    *(void **)&g_rccl.GetErrorString = dlsym(dl, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.AllGather ||
        !g_rccl.GroupStart || !g_rccl.GroupEnd)
        return fail(GMS_ERR_STATE, "RCCL library lacks the collective entry points");
    g_rccl.dl = dl;
    return GMS_OK;
}

int gms_comm_unique_id(void *id128) {
    REQUIRE(id128, "null argument");
    int rc = gms_comm_load(nullptr);
    if (rc) return rc;
    RCCLCHK(g_rccl.GetUniqueId(reinterpret_cast<rccl_unique_id *>(id128)));
    return GMS_OK;
}

int gms_comm_create(gms_comm **out, const void *id128, int32_t rank, int32_t world, int32_t device) {
    REQUIRE(out && id128, "null argument");
    REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank/world out of range");
    int rc = gms_comm_load(nullptr);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    gms_comm *c = new gms_comm();
    c->rank = rank; c->world = world; c->device = device;
    // The event fork/join that puts the all-gather beside the map update costs ~18 us on the stream (measured,
    // MI355X); an in-line gather costs its own latency.  Small groups gather little: in line up to 2 ranks.
    c->overlap = world > 2;
    if (const char *e = getenv("GMS_COMM_OVERLAP")) c->overlap = atoi(e) != 0;
    if (const char *e = getenv("GMS_EXCHANGE")) c->p2p = strcmp(e, "p2p") == 0;
    rccl_unique_id id;
    memcpy(&id, id128, sizeof(id));
    int r = g_rccl.CommInitRank(&c->nccl, world, id, rank);        // blocks until every rank has joined
    if (r != 0) { delete c; return fail(GMS_ERR_HIP, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"); }
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) { gms_comm_destroy(c); return fail(GMS_ERR_HIP, "communicator streams: %s", hipGetErrorString(e)); }
    *out = c;
    return GMS_OK;
}

int gms_comm_destroy(gms_comm *c) {
    if (!c) return GMS_OK;
    hipSetDevice(c->device);
    if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl);
    delete c;
    return GMS_OK;
}

int gms_comm_rank(const gms_comm *c, int32_t *rank, int32_t *world) {
    REQUIRE(c && rank && world, "null argument");
    *rank = c->rank; *world = c->world;
    return GMS_OK;
}

static int check_shard(const gms_pf *pf, const gms_comm *c) {
    if (pf->n_maps != 1) return fail(GMS_ERR_STATE, "sharded filters hold one map per handle");
    if (pf->n_global != (int64_t)pf->n * c->world || pf->offset != (int64_t)pf->n * c->rank)
        return fail(GMS_ERR_INVALID, "shard mismatch: n_local %d offset %lld n_global %lld on rank %d of %d (equal shards, rank order)",
                    pf->n, (long long)pf->offset, (long long)pf->n_global, c->rank, c->world);
    if (c->world > 1 && pf->n % GMS_BLOCK)
        return fail(GMS_ERR_INVALID, "shard size %d must be a multiple of GMS_BLOCK=%d", pf->n, GMS_BLOCK);
    if (pf->map->device != c->device) return fail(GMS_ERR_INVALID, "filter and communicator are on different devices");
    return GMS_OK;
}

// SLAM.java:100-124 across ranks: weight sum / strongest / normalise, and the START of the exchange of the
// normalised population.  Statistics and the weighted pose are complete on return (stream order).
int gms_pf_normalize_sharded_begin(gms_pf *pf, gms_comm *c) {
    REQUIRE(pf && c, "null argument");
    int rc = check_shard(pf, c);
    if (rc) return rc;
    gms_map *m = pf->map;
    HIPCHK(hipSetDevice(m->device));
    if (c->broken) return fail(GMS_ERR_STATE, "communicator is broken (an earlier exchange failed): destroy it");
    if (c->pending) return fail(GMS_ERR_STATE, "gms_pf_normalize_sharded_end has not been called for the previous exchange");
    const size_t np = (size_t)nblk_of(pf->n_global) * GMS_PARTIAL_STRIDE;
    gms_launch_pf_partials(pf, pf->d_partials);
    RCCLCHK(g_rccl.AllReduce(pf->d_partials, pf->d_partials, np, RCCL_FLOAT64, RCCL_SUM, c->nccl, m->stream));
    PackedParticle *own_slot = pf->d_global_own + pf->offset;
    gms_launch_pf_apply_partials(pf, pf->d_partials, own_slot, false);
    pf->have_global = 0;
    pf->stats_current = 1;
    hipStream_t s = m->stream;
    if (c->overlap) {                                   // the gather proceeds beside whatever the caller enqueues next
        HIPCHK(hipEventRecord(c->ev_fork, m->stream));
        HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        s = c->side;
    }
    RCCLCHK(g_rccl.AllGather(own_slot, pf->d_global_own, (size_t)pf->n * sizeof(PackedParticle), RCCL_INT8, c->nccl, s));
    if (c->overlap) HIPCHK(hipEventRecord(c->ev_join, c->side));
    c->pending = 1;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

// joins the exchange: the gathered population becomes the resampling source
int gms_pf_normalize_sharded_end(gms_pf *pf, gms_comm *c) {
    REQUIRE(pf && c, "null argument");
    if (!c->pending) return fail(GMS_ERR_STATE, "no exchange in flight");
    gms_map *m = pf->map;
    HIPCHK(hipSetDevice(m->device));
    if (c->overlap) HIPCHK(hipStreamWaitEvent(m->stream, c->ev_join, 0));
    c->pending = 0;
    pf->d_global = pf->d_global_own;
    gms_launch_pf_after_gather(pf);
    pf->have_global = 1;
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

// ---- sharded scan step with ONE exchange ------------------------------------------------------------------------
// The ranks exchange RAW weights: each rank's block partials and its raw pack {w, x, y, theta} are both all-gather
// payloads (the gathered partial vector is what the all-reduce used to assemble), so they travel in one grouped
// launch, and everything after it is local: fold, normalise own, cumulative sums of the normalised global weights
// (the owner's division, w / weightSum, repeated on the gathered copy: same operands, same bits), ray cast, resample.
static int sharded_shape_ok(const gms_pf *pf) {
    if (pf->n_maps != 1) return fail(GMS_ERR_STATE, "sharded filters hold one map per handle");
    if (pf->reference_order) return fail(GMS_ERR_STATE, "gms_pf_set_reference_order is on: stand-alone scan steps only");
    if (pf->log_norm)
        return fail(GMS_ERR_STATE, "gms_pf_set_log_normalize is on: the sharded scan steps exchange raw weights and do not rescale them; turn it off (stand-alone steps only)");
    if (pf->n_global != pf->n && pf->n % GMS_BLOCK)
        return fail(GMS_ERR_INVALID, "shard size %d must be a multiple of GMS_BLOCK=%d", pf->n, GMS_BLOCK);
    if (pf->n_global % pf->n || pf->offset % pf->n)
        return fail(GMS_ERR_INVALID, "equal shards in rank order are required (n_local %d, offset %lld, n_global %lld)", pf->n,
                    (long long)pf->offset, (long long)pf->n_global);
    return GMS_OK;
}

// poses := dev_xytheta (may be NULL), weights, this shard's block partials and raw pack in their slots of the
// gather buffers (gms_pf_gather_buffers)
static int sharded_begin(gms_pf *pf, const float *dev_xytheta, const gms_beam *dev_beams, int32_t B, bool apply_rides_later) {
    REQUIRE(pf && dev_beams, "null argument");
    int rc = sharded_shape_ok(pf);
    if (rc) return rc;
    rc = set_poses_and_score_dev(pf, dev_xytheta, dev_beams, B, pf->refine != 0);   // SLAM.java:90, :96-97, :99
    if (rc) return rc;
    gms_launch_partials_pack_apply(pf, apply_rides_later);                       // :100-115 | previous scan's GridMap.java:223 (unless it rides beside the ray cast)
    HIPCHK(hipGetLastError());
    return GMS_OK;
}

int gms_slam_update_sharded_begin_dev(gms_pf *pf, const float *dev_xytheta, const gms_beam *dev_beams, int32_t B) {
    return sharded_begin(pf, dev_xytheta, dev_beams, B, false);      // (the end half's `integrate` is not known yet)
}

int gms_pf_gather_buffers(gms_pf *pf, void **dev_packed_global, int64_t *packed_bytes_per_rank, double **dev_partials_global,
                          int64_t *partials_doubles_per_rank) {
    REQUIRE(pf, "null filter");
    if (dev_packed_global) *dev_packed_global = pf->d_global_own;
    if (packed_bytes_per_rank) *packed_bytes_per_rank = (int64_t)pf->n * (int64_t)sizeof(PackedParticle);
    if (dev_partials_global) *dev_partials_global = pf->d_partials;
    if (partials_doubles_per_rank) *partials_doubles_per_rank = nblk_of(pf->n) * GMS_PARTIAL_STRIDE;
    return GMS_OK;
}

// after both buffers have been all-gathered in place: statistics, normalisation, map update, conditional resample
int gms_slam_update_sharded_end_dev(gms_pf *pf, const gms_beam *dev_beams, int32_t B, const double *r01,
                                    double resample_fraction, int32_t integrate) {
    REQUIRE(pf && dev_beams && r01, "null argument");
    gms_map *m = pf->map;
    int rc = sharded_shape_ok(pf);
    if (rc) return rc;
    HIPCHK(hipSetDevice(m->device));
    const bool pair = integrate && gms_can_pair_launches(pf, B);
    gms_launch_raycast_norm_chunks(pf, dev_beams, B, pair);                      // :93 | :120-124 | level 0 of :140-149
    pf->have_global = 1;
    pf->stats_current = 1;
    HIPCHK(hipGetLastError());
    if (pair) return paired_likelihood_resample(pf, r01, resample_fraction);     // :105 | GridMapApp.java:185-186
    if (integrate) rc = gms_map_update_at_dev(m, dev_beams, B, pf, 0);
    if (!rc && resample_fraction >= 0.0) rc = gms_pf_resample_if(pf, r01, resample_fraction);
    return rc;
}

// One scan step of a sharded filter, the exchange included: what gms_slam_update_dev is for a stand-alone one.
int gms_slam_update_sharded_dev(gms_pf *pf, gms_comm *c, const float *dev_xytheta, const gms_beam *dev_beams, int32_t B,
                                const double *r01, double resample_fraction, int32_t integrate) {
    REQUIRE(pf && c && dev_beams && r01, "null argument");
    gms_map *m = pf->map;
    int rc = check_shard(pf, c);
    if (rc) return rc;
    if (c->broken) return fail(GMS_ERR_STATE, "communicator is broken (an earlier exchange failed): destroy it");
    if (c->pending) return fail(GMS_ERR_STATE, "gms_pf_normalize_sharded_end has not been called for the previous exchange");
    rc = sharded_begin(pf, dev_xytheta, dev_beams, B, integrate && gms_can_pair_launches(pf, B));
    if (rc) return rc;
    const size_t np = (size_t)nblk_of(pf->n) * GMS_PARTIAL_STRIDE;               // doubles per rank
    double *own_partials = pf->d_partials + (size_t)c->rank * np;
    PackedParticle *own_slot = pf->d_global_own + pf->offset;
    const size_t pbytes = (size_t)pf->n * sizeof(PackedParticle);
    int first = 0, end = 0;
    const char *what = "";
    {
    ProfScope ps(m, GMS_K_EXCHANGE);                                             // event-bracketed when asked for: the L of DESIGN.md section 7
    RCCLCHK(g_rccl.GroupStart());                                                // the whole exchange is one launch
    // Inside the bracket an error must not return: the thread's RCCL group depth would stay at 1 and every later
    // collective of this process (torch.distributed shares this librccl) would queue behind a group that never closes.
    // The first error is kept, GroupEnd always runs, and the communicator is marked broken.
#define RCCL_IN_GROUP(expr) do { if (!first) { first = (expr); if (first) what = #expr; } } while (0)
    if (c->p2p && g_rccl.Send && g_rccl.Recv) {
        // the same exchange as point-to-point transfers to and from every peer (direct xGMI links, no ring):
        // GMS_EXCHANGE=p2p, for comparison on a multi-GPU node
        for (int32_t peer = 0; peer < c->world; peer++) {
            if (peer == c->rank) continue;
            RCCL_IN_GROUP(g_rccl.Send(own_slot, pbytes, RCCL_INT8, peer, c->nccl, m->stream));
            RCCL_IN_GROUP(g_rccl.Recv(pf->d_global_own + (size_t)peer * pf->n, pbytes, RCCL_INT8, peer, c->nccl, m->stream));
            RCCL_IN_GROUP(g_rccl.Send(own_partials, np, RCCL_FLOAT64, peer, c->nccl, m->stream));
            RCCL_IN_GROUP(g_rccl.Recv(pf->d_partials + (size_t)peer * np, np, RCCL_FLOAT64, peer, c->nccl, m->stream));
        }
    } else {
        RCCL_IN_GROUP(g_rccl.AllGather(own_slot, pf->d_global_own, pbytes, RCCL_INT8, c->nccl, m->stream));
        RCCL_IN_GROUP(g_rccl.AllGather(own_partials, pf->d_partials, np, RCCL_FLOAT64, c->nccl, m->stream));
    }
#undef RCCL_IN_GROUP
    end = g_rccl.GroupEnd();
    }
    if (first || end) {
        c->broken = 1;
        // the begin half may have consumed a deferred apply pass; bring the map to a defined state for the fallback route
        gms_flush_apply(m);
        const int r_ = first ? first : end;
        return fail(GMS_ERR_HIP, "%s: %s (communicator marked broken)", first ? what : "ncclGroupEnd",
                    g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error");
    }
    return gms_slam_update_sharded_end_dev(pf, dev_beams, B, r01, resample_fraction, integrate);
}

int gms_slam_update_sharded(gms_pf *pf, gms_comm *c, const float *xytheta, const gms_beam *beams, int32_t B, const double *r01,
                            double resample_fraction, int32_t integrate, gms_pf_stats *stats) {
    REQUIRE(pf && c && beams && r01, "null argument");
    gms_map *m = pf->map;
    REQUIRE(m->n_maps == 1, "sharded filters hold one map per handle");
    int rc = GMS_OK;
    if (xytheta) rc = gms_pf_set_poses(pf, xytheta);
    if (!rc) rc = stage_beams(m, beams, B);
    if (!rc) rc = gms_slam_update_sharded_dev(pf, c, nullptr, m->d_beams, B, r01, resample_fraction, integrate);
    if (!rc && stats) rc = gms_pf_get_stats(pf, stats);
    return rc;
}

// ---- the reference-shape filter (particles with their maps) sharded, the exchanges inside the library -------------------------------
// The plan of a sharded resample() as a pure host function (also gridmap_slam_robot_amd/distributed.py: plan_map_exchange; a CPU test
// holds the two against each other).  all_sources [world][n_local]: the global source of every slot of every rank.  Out, for `rank`:
// send_counts [world] and send_lists [world][n_local] (its local particle indices whose records rank q needs: ascending, each once),
// recv_counts [world] (records arriving from rank q, in ascending order of their global index), src_local [n_local] (the slot's source
// in this rank's own previous generation, or -1), recv_pos [n_local] (for a remote source: its position in the received records
// concatenated in rank order).
int gms_slam_plan_exchange(const int32_t *all_sources, int32_t world, int32_t rank, int32_t n_local, int32_t *send_counts, int32_t *send_lists,
                           int32_t *recv_counts, int32_t *src_local, int32_t *recv_pos) {
    REQUIRE(all_sources && send_counts && send_lists && recv_counts && src_local && recv_pos, "null argument");
    REQUIRE(world >= 1 && rank >= 0 && rank < world && n_local >= 1, "rank / world / block size out of range");
    const int64_t lo = (int64_t)rank * n_local;
    const int32_t *mine = all_sources + (size_t)rank * n_local;
    for (int32_t m = 0; m < n_local; m++) {
        REQUIRE(mine[m] >= 0 && (int64_t)mine[m] < (int64_t)world * n_local, "a source index outside the population");
        src_local[m] = mine[m] / n_local == rank ? (int32_t)(mine[m] - lo) : -1;
        recv_pos[m] = -1;
    }
    std::vector<int32_t> uniq;
    int32_t base = 0;
    for (int32_t q = 0; q < world; q++) {
        send_counts[q] = 0; recv_counts[q] = 0;
        if (q == rank) continue;
        // what q's slots drew of MY particles: distinct values, ascending (the sources of a systematic draw are non-decreasing, but
        // nothing here depends on that)
        uniq.clear();
        const int32_t *theirs = all_sources + (size_t)q * n_local;
        for (int32_t m = 0; m < n_local; m++)
            if (theirs[m] / n_local == rank) uniq.push_back(theirs[m]);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        send_counts[q] = (int32_t)uniq.size();
        for (size_t k = 0; k < uniq.size(); k++) send_lists[(size_t)q * n_local + k] = (int32_t)(uniq[k] - lo);
        // what MY slots drew of q's particles
        uniq.clear();
        for (int32_t m = 0; m < n_local; m++)
            if (mine[m] / n_local == q) uniq.push_back(mine[m]);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        recv_counts[q] = (int32_t)uniq.size();
        for (int32_t m = 0; m < n_local; m++)
            if (mine[m] / n_local == q)
                recv_pos[m] = base + (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), mine[m]) - uniq.begin());
        base += (int32_t)uniq.size();
    }
    return GMS_OK;
}

// SLAM.update(z, u) of a sharded reference-shape filter in one call: the per-particle body for this rank's block, then weightSum /
// strongest / normalise / Neff over all ranks (RCCL all-reduce of the block partials + all-gather of the packed particles:
// gms_pf_normalize_sharded_begin / _end on the handle's filter).  Every rank: the same scan, odometry, seed and sequence.
int gms_slam_update_sharded_maps(gms_slam *s, gms_comm *c, const gms_beam *beams, int32_t B, int32_t sample_motion, double d_center, double d_theta,
                                 uint64_t seed, uint64_t sequence, gms_pf_stats *stats) {
    REQUIRE(s && c && beams, "null argument");
    int rc = gms_slam_update_local(s, beams, B, sample_motion, d_center, d_theta, seed, sequence);
    if (!rc) rc = gms_pf_normalize_sharded_begin(s->pf, c);
    if (!rc) rc = gms_pf_normalize_sharded_end(s->pf, c);
    if (!rc && stats) rc = gms_pf_get_stats(s->pf, stats);
    return rc;
}

// SLAM.resample() of a sharded reference-shape filter in one call (fraction >= 0: the caller's rule, GridMapApp.java:185-186): the draw
// for this rank's slots, an RCCL all-gather of the slots' sources, the plan, the records of the particles that crossed a rank boundary as
// ONE grouped launch of ncclSend / ncclRecv, and the copies.  Every rank: the same r01.  *did (may be NULL): it drew.
// (With more than one rank this has never executed: one GPU per box in the pool this was written on.  One rank: tests/test_gpu_slam_sharded.py.)
int gms_slam_resample_sharded_maps(gms_slam *s, gms_comm *c, double r01, double fraction, int32_t *did_out) {
    REQUIRE(s && c, "null argument");
    gms_pf *pf = s->pf;
    gms_map *m = s->map;
    int rc = check_shard(pf, c);
    if (rc) return rc;
    if (c->broken) return fail(GMS_ERR_STATE, "communicator is broken (an earlier exchange failed): destroy it");
    HIPCHK(hipSetDevice(m->device));
    const int32_t n = pf->n, world = c->world;
    std::vector<int32_t> src((size_t)n), all((size_t)n * world);
    int32_t did = 0;
    rc = gms_slam_shard_draw(s, r01, fraction, &did, src.data());
    if (rc) return rc;
    if (did_out) *did_out = did;
    if (!did) return GMS_OK;                           // (every rank decides alike: the same statistics)
    int64_t rec = 0;
    rc = gms_slam_record_doubles(s, &rec);
    if (rc) return rc;
    // the sources of every rank's slots
    int32_t *d_all = nullptr;
    double *d_send = nullptr, *d_recv = nullptr;
    auto cleanup = [&]() { hipFree(d_all); hipFree(d_send); hipFree(d_recv); };
    if (hipMalloc(&d_all, (size_t)n * world * sizeof(int32_t)) != hipSuccess) return fail(GMS_ERR_NOMEM, "gms_slam_resample_sharded_maps: device allocation failed");
    {
        const int r_ = g_rccl.AllGather(pf->d_idx, d_all, (size_t)n * sizeof(int32_t), RCCL_INT8, c->nccl, m->stream);
        if (r_ != 0) { c->broken = 1; cleanup(); return fail(GMS_ERR_HIP, "ncclAllGather (sources): %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); }
    }
    if (hipStreamSynchronize(m->stream) != hipSuccess ||
        hipMemcpy(all.data(), d_all, (size_t)n * world * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) { cleanup(); return fail(GMS_ERR_HIP, "gms_slam_resample_sharded_maps: reading the sources back failed"); }
    std::vector<int32_t> send_counts((size_t)world), send_lists((size_t)world * n), recv_counts((size_t)world), src_local((size_t)n), recv_pos((size_t)n);
    rc = gms_slam_plan_exchange(all.data(), world, c->rank, n, send_counts.data(), send_lists.data(), recv_counts.data(), src_local.data(), recv_pos.data());
    if (rc) { cleanup(); return rc; }
    int64_t n_send = 0, n_recv = 0;
    for (int32_t q = 0; q < world; q++) { n_send += send_counts[q]; n_recv += recv_counts[q]; }
    if (n_send && hipMalloc(&d_send, (size_t)n_send * rec * sizeof(double)) != hipSuccess) { cleanup(); return fail(GMS_ERR_NOMEM, "gms_slam_resample_sharded_maps: %lld records to send do not fit", (long long)n_send); }
    if (n_recv && hipMalloc(&d_recv, (size_t)n_recv * rec * sizeof(double)) != hipSuccess) { cleanup(); return fail(GMS_ERR_NOMEM, "gms_slam_resample_sharded_maps: %lld records to receive do not fit", (long long)n_recv); }
    int64_t off = 0;
    for (int32_t q = 0; q < world && !rc; q++) {
        if (!send_counts[q]) continue;
        rc = gms_slam_shard_export(s, send_lists.data() + (size_t)q * n, send_counts[q], d_send + (size_t)off * rec);
        off += send_counts[q];
    }
    if (rc) { cleanup(); return rc; }
    if (n_send || n_recv) {
        if (!g_rccl.Send || !g_rccl.Recv) { cleanup(); return fail(GMS_ERR_STATE, "this RCCL has no ncclSend / ncclRecv"); }
        int first = 0, end = 0;
        const char *what = "";
        {
            ProfScope ps(m, GMS_K_EXCHANGE);
            first = g_rccl.GroupStart();
            int64_t so = 0, ro = 0;
            for (int32_t q = 0; q < world; q++) {              // (inside the group an error must not return: see gms_slam_update_sharded_dev)
                if (send_counts[q]) {
                    if (!first) { first = g_rccl.Send(d_send + (size_t)so * rec, (size_t)send_counts[q] * rec, RCCL_FLOAT64, q, c->nccl, m->stream); if (first) what = "ncclSend"; }
                    so += send_counts[q];
                }
                if (recv_counts[q]) {
                    if (!first) { first = g_rccl.Recv(d_recv + (size_t)ro * rec, (size_t)recv_counts[q] * rec, RCCL_FLOAT64, q, c->nccl, m->stream); if (first) what = "ncclRecv"; }
                    ro += recv_counts[q];
                }
            }
            end = g_rccl.GroupEnd();
        }
        if (first || end) {
            c->broken = 1;
            hipStreamSynchronize(m->stream);
            cleanup();
            const int r_ = first ? first : end;
            return fail(GMS_ERR_HIP, "%s: %s (communicator marked broken; the maps of this generation are incomplete)", first ? what : "ncclGroupEnd",
                        g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error");
        }
    }
    rc = gms_slam_shard_gather(s, src_local.data(), recv_pos.data(), d_recv);
    hipStreamSynchronize(m->stream);                   // (the staging buffers are freed below)
    cleanup();
    return rc;
}

}  // extern "C"
