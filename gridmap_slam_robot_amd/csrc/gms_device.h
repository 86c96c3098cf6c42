// gms_device.h -- device-side arithmetic helpers with the reference's (Java) numeric semantics.
// Compiled with -ffp-contract=off: the JVM never fuses a multiply with an add.
// Citations: J/ = java/GridMapGL/src/main/java/com/fmsz/gridmapgl/ in the reference tree.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gms_internal.h"

#define GMS_WAVE 64

// A store that is written through to memory at once (agent scope) instead of staying dirty in the XCD's L2 until the kernel's end
// write-back: for data nobody in this launch reads again.
__device__ __forceinline__ void store_through(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_through(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Development builds only (-DGMS_STAMPS, tools/stamps.py): wall-clock stamps (100 MHz) of a kernel's stages, one row of
// GMS_STAMP_SLOTS per workgroup, written by thread 0.  Compiled out of the product library.
#define GMS_STAMP_SLOTS 16
#ifdef GMS_STAMPS
__device__ unsigned long long *g_gms_stamps;
#define GMS_STAMP_T(tid, row, slot) do { if (g_gms_stamps && threadIdx.x == (tid)) g_gms_stamps[(size_t)(row) * GMS_STAMP_SLOTS + (slot)] = wall_clock64(); } while (0)
#else
#define GMS_STAMP_T(tid, row, slot) do { } while (0)
#endif
#define GMS_STAMP(row, slot) GMS_STAMP_T(0, row, slot)
#define GMS_STAMP_ROW(kernel_id, wg) ((kernel_id) * 1024u + ((wg) < 1023u ? (wg) : 1023u))      // rows of the stamp buffer: [4 kernels][1024 workgroups]

// (int) of a double, JLS 5.1.3: truncate toward zero, saturate, NaN -> 0.
__device__ __forceinline__ int32_t j_d2i(double d) {
    if (d != d) return 0;
    if (d >= 2147483647.0) return INT32_MAX;
    if (d <= -2147483648.0) return INT32_MIN;
    return (int32_t)d;
}
// The same conversion in one instruction: v_cvt_i32_f64 truncates toward zero, saturates and maps
// NaN to 0 -- exactly JLS 5.1.3 (used in the scoring loop, where the three compares above cost more
// than the conversion).
__device__ __forceinline__ int32_t j_d2i_hw(double d) {
    int32_t r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(d));
    return r;
}

// (int)(d / res) without the division in the common case.  q' = d * RN(1/res) differs from the
// correctly rounded quotient Q = RN(d / res) by at most 1.5 * 2^-52 |Q|; unless q' lies within that
// distance of an integer, Q and q' truncate to the same int.  The guard used is absolute, 2^-19 >=
// 4 * 2^-52 * 2^31: it covers every |q| that can land inside a map (beyond |q| ~ 2^31 both quotients
// are outside the map, or saturate, alike) and trips for about 4e-6 of the calls.  NaN and
// infinities never trip it and convert like Q would.
//   j_cell_fast  branch-free: the int from q', and whether the guard tripped
//   j_cell_exact the reference's expression, (int)(d / res)
__device__ __forceinline__ int32_t j_cell_fast(double d, double rinv, bool &guard) {
    const double q = d * rinv;
    guard = guard | (fabs(q - rint(q)) <= 0x1p-19);
    return j_d2i_hw(q);
}
__device__ __forceinline__ int32_t j_cell_exact(double d, double res) { return j_d2i_hw(d / res); }

// Java int arithmetic wraps.
__device__ __forceinline__ int32_t j_iadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t j_isub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

// (float) Math.sqrt((double) s) for a float s == correctly rounded float sqrt (GridMap.java:217).
__device__ __forceinline__ float j_sqrtf(float s) { return __builtin_sqrtf(s); }

// Transform.fromRobotToWorld's trig: (double)(float) FastMath.cos((double) theta)
// (J/math/Transform.java:15-16 via J/math/MathUtil.java:30-40).
__device__ __forceinline__ void pose_trig(float theta, float &c, float &s) {
    double sd, cd;
    sincos((double)theta, &sd, &cd);                 // one argument reduction for both; same values as sin() and cos()
    c = (float)cd;                                   // (tests/test_gpu_parity.py::test_device_sqrt_and_trig_round_like_the_oracle)
    s = (float)sd;
}

// Transform.transformX / transformY (J/math/Transform.java:23,28): double, no FMA.
struct XformDev {
    double c, s, px, py;
};
__device__ __forceinline__ double xform_x(const XformDev &t, double x, double y) { return x * t.c - y * t.s + t.px; }
__device__ __forceinline__ double xform_y(const XformDev &t, double x, double y) { return x * t.s + y * t.c + t.py; }

// ---- RayIterator (J/slam/RayIterator.java:65-130) ---------------------------------------------
struct RayDev {
    int32_t x, y, x_inc, y_inc, n;
    float dx, dy, error;
};

__device__ __forceinline__ void ray_init(RayDev &r, float x0, float y0, float x1, float y1, int32_t extra) {
    r.dx = fabsf(x1 - x0);                                   // :68
    r.dy = fabsf(y1 - y0);                                   // :69
    const double fx0 = floor((double)x0), fy0 = floor((double)y0);
    r.x = j_d2i(fx0);                                        // :71
    r.y = j_d2i(fy0);                                        // :72
    r.n = j_iadd(1, extra);                                  // :75
    if (r.dx == 0.0f) {                                      // :78
        r.x_inc = 0;
        r.error = INFINITY;
    } else if (x1 > x0) {                                    // :81
        r.x_inc = 1;
        r.n = j_iadd(r.n, j_d2i(floor((double)x1) - (double)r.x));            // :83
        r.error = (float)((fx0 + 1.0 - (double)x0) * (double)r.dy);           // :84
    } else {
        r.x_inc = -1;
        r.n = j_iadd(r.n, j_isub(r.x, j_d2i(floor((double)x1))));             // :87
        r.error = (float)(((double)x0 - fx0) * (double)r.dy);                 // :88
    }
    if (r.dy == 0.0f) {                                      // :91
        r.y_inc = 0;
        r.error = r.error - INFINITY;                        // :93
    } else if (y1 > y0) {                                    // :94
        r.y_inc = 1;
        r.n = j_iadd(r.n, j_isub(j_d2i(floor((double)y1)), r.y));             // :96
        r.error = (float)((double)r.error - (fy0 + 1.0 - (double)y0) * (double)r.dx);   // :97
    } else {
        r.y_inc = -1;
        r.n = j_iadd(r.n, j_isub(r.y, j_d2i(floor((double)y1))));             // :100
        r.error = (float)((double)r.error - ((double)y0 - fy0) * (double)r.dx);         // :101
    }
}

__device__ __forceinline__ bool ray_has_next(const RayDev &r, int32_t W, int32_t H) {
    return r.n > 0 && !(r.x < 0 || r.x >= W || r.y < 0 || r.y >= H);        // :108
}

__device__ __forceinline__ void ray_step(RayDev &r) {                          // :117-126
    if (r.error > 0.0f) {
        r.y = j_iadd(r.y, r.y_inc);
        r.error = r.error - r.dx;
    } else {
        r.x = j_iadd(r.x, r.x_inc);
        r.error = r.error + r.dy;
    }
    r.n = j_isub(r.n, 1);
}

// probabilityOf's per-beam factor as a function of the likelihood value (GridMap.java:285-288)
__device__ __forceinline__ double lik_factor(const GridDev &g, double val) {
    return val == 0.5 ? g.inv_max : g.z_hit * val + g.c_rand;
}

// Where probabilityOf's look-up of cell (gx, gy) goes in the factor table, for ANY int gx, gy: the table has H + 1 rows
// of g.fpitch = W + 16 entries, column W of every row and the whole of row H hold the neutral factor 1.0, and a
// coordinate outside [0, W) x [0, H) (GridMap.java:276; negative values wrap to huge unsigned ones) is clamped onto that
// border.  Two v_min_u32 and one full-rate v_mad_u32_u24 replace two compares, a select and a quarter-rate 32-bit
// multiply: the scoring loop is as much bound by its ~40 vector instructions per look-up as by the look-up itself
// (DESIGN.md section 4, round 2).  likelihoodData itself (d_lik) stays W x H.
__device__ __forceinline__ uint32_t fac_index(const GridDev &g, int32_t gx, int32_t gy) {
    return __umul24(min((uint32_t)gy, (uint32_t)g.H), (uint32_t)g.fpitch) + min((uint32_t)gx, (uint32_t)g.W);
}

// SensorModel.inverseSensorModel (J/slam/SensorModel.java:31-41) -> class 0 free, 1 prior, 2 occupied
__device__ __forceinline__ int32_t sensor_class(float cur, float measured, int32_t hit, float half_tol) {
    if (!hit) return cur < measured ? 0 : 1;
    if (cur < measured - half_tol) return 0;
    if (cur > measured + half_tol) return 1;
    return 2;
}

// ---- the sensor class without a square root per cell ------------------------------------------------------------------------
// inverseSensorModel compares distance = (float) Math.sqrt(s), s = dX * dX + dY * dY (GridMap.java:215-217, a correctly rounded
// float square root of a float), with measured -+ hitTolerance / 2 (SensorModel.java:31-41).  A correctly rounded square root is
// monotonic, so for a threshold t the set {s : sqrt(s) >= t} is an upper set of the floats: d < t <=> s < sq_lower(t), its smallest
// element, and d > t <=> s > sq_upper(t), the largest s whose root is still <= t.  The two thresholds are found once per RAY (t * t,
// then a step or two along the floats, checked with the same square root), and every cell of the ray is classified by two compares:
// the same classes, bit for bit (tests/test_gpu_slam_particle_maps.py::test_squared_thresholds...), ~20 instructions per cell less.
__device__ __forceinline__ float f32_up(float c) { return __uint_as_float(__float_as_uint(c) + 1u); }       // c >= +0, finite
__device__ __forceinline__ float f32_down(float c) { return __uint_as_float(__float_as_uint(c) - 1u); }     // c > 0
// smallest float s >= 0 with (float)sqrt(s) >= t;  d < t <=> s < result  (NaN for a NaN t: never true, as in the reference)
__device__ __forceinline__ float sq_lower(float t) {
    if (t != t) return t;
    if (!(t > 0.0f)) return 0.0f;                      // d < t never holds for d >= 0
    float c = t * t;
    if (!(c < INFINITY)) return INFINITY;              // t = +Inf, or beyond sqrt(FLT_MAX): every finite s lies below
    for (int i = 0; i < 8 && c > 0.0f && j_sqrtf(f32_down(c)) >= t; i++) c = f32_down(c);
    for (int i = 0; i < 8 && j_sqrtf(c) < t; i++) c = f32_up(c);
    return c;
}
// largest float s with (float)sqrt(s) <= t;  d > t <=> s > result
__device__ __forceinline__ float sq_upper(float t) {
    if (t != t) return t;
    if (t < 0.0f) return -1.0f;                        // d > t always holds for d >= 0 (and never for a NaN s, as in the reference)
    if (t == INFINITY) return INFINITY;
    float c = t * t;
    if (!(c < INFINITY)) return 3.4028234663852886e38f;   // beyond sqrt(FLT_MAX): only s = +Inf has a larger root
    for (int i = 0; i < 8 && j_sqrtf(c) > t; i++) c = f32_down(c);           // (c > 0 here: sqrt(0) = 0 <= t)
    for (int i = 0; i < 8 && j_sqrtf(f32_up(c)) <= t; i++) c = f32_up(c);
    return c;
}
struct RayThr { float s_free, s_prior; };              // class 0 iff s < s_free; of a hit ray: class 1 iff s > s_prior, else 2; of a miss: else 1
__device__ __forceinline__ RayThr ray_thresholds(float measured, int32_t hit, float half_tol) {
    RayThr t;
    if (hit) { t.s_free = sq_lower(measured - half_tol); t.s_prior = sq_upper(measured + half_tol); }     // SensorModel.java:36-40
    else { t.s_free = sq_lower(measured); t.s_prior = 0.0f; }                                              // :33-34
    return t;
}
__device__ __forceinline__ int32_t sensor_class_sq(float s, const RayThr &t, int32_t hit) {
    return s < t.s_free ? 0 : (hit ? (s > t.s_prior ? 1 : 2) : 1);
}

// distance from the (un-shifted) ray start to the centre of cell (cx,cy), GridMap.java:215-217
__device__ __forceinline__ float cell_distance(float sx, float sy, int32_t cx, int32_t cy) {
    float dX = sx - ((float)cx + 0.5f);
    float dY = sy - ((float)cy + 0.5f);
    return j_sqrtf(dX * dX + dY * dY);
}

// GridMap.integrateObservation's per-beam locals (GridMap.java:175-188)
__device__ __forceinline__ RayIn make_ray(const GridDev &g, const gms_beam &m, const float *pose) {
    XformDev t;
    float c, s;
    pose_trig(pose[2], c, s);
    t.c = (double)c; t.s = (double)s; t.px = (double)pose[0]; t.py = (double)pose[1];
    RayIn r;
    r.sx = (float)((xform_x(t, 0.0, 0.0) - g.posx) / g.res);                  // :178
    r.sy = (float)((xform_y(t, 0.0, 0.0) - g.posy) / g.res);                  // :179
    r.ex = (float)((xform_x(t, m.local_x, m.local_y) - g.posx) / g.res);      // :185
    r.ey = (float)((xform_y(t, m.local_x, m.local_y) - g.posy) / g.res);      // :186
    r.measured = (float)m.distance / g.resf;                                  // :188
    r.hit = m.hit != 0;
    return r;
}

// MathUtil.angleConstrain (J/math/MathUtil.java:65-72).  The Java loops do not terminate for
// infinite or astronomically large angles; the device version gives up after 64 turns.
__device__ __forceinline__ double angle_constrain(double a) {
    const double PI = 3.141592653589793;
    for (int i = 0; i < 64 && a < PI; i++) a += PI * 2;
    for (int i = 0; i < 64 && a > PI; i++) a -= PI * 2;
    return a;
}

// ---- wave64 helpers ------------------------------------------------------------------------------
// The value lane (lane ^ O) holds, O one of 32, 16, 8, 4, 2, 1, without the LDS crossbar that __shfl_xor goes through
// (ds_bpermute_b32: ~65 clocks per dependent use; the nine butterflies of the block partials took 1.5 us of k_partials' 4.5):
// gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves / rows of 16, DPP row rotations and quad permutations do
// the rest.  FLIP: the other reading of the swap's result order / the rotation's direction (gms_debug_f32 op 3 checks both
// against __shfl_xor on the device; the product uses the one that matches).
template <int O, bool FLIP = false>
__device__ __forceinline__ uint32_t wave_xor_u32(uint32_t v) {
    static_assert(O == 32 || O == 16 || O == 8 || O == 4 || O == 2 || O == 1, "xor butterfly offsets of a 64-lane wavefront");
    const uint32_t lane = __lane_id();
    if constexpr (O == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // {lanes 32-63 of the first, lanes 0-31 of the second} exchanged
        return (((lane & 32u) != 0u) != FLIP) ? r[0] : r[1];
    } else if constexpr (O == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // odd rows of the first, even rows of the second
        return (((lane & 16u) != 0u) != FLIP) ? r[0] : r[1];
    } else if constexpr (O == 8) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);          // row_ror:8
    } else if constexpr (O == 4) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);   // row_ror:4
        const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x12c, 0xf, 0xf, false);   // row_ror:12
        return (((lane & 4u) != 0u) != FLIP) ? a : b;
    } else if constexpr (O == 2) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);           // quad_perm:[2,3,0,1]
    } else {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);           // quad_perm:[1,0,3,2]
    }
}
template <int O> __device__ __forceinline__ int32_t wave_xor(int32_t v) { return (int32_t)wave_xor_u32<O>((uint32_t)v); }
template <int O> __device__ __forceinline__ uint32_t wave_xor(uint32_t v) { return wave_xor_u32<O>(v); }
template <int O> __device__ __forceinline__ float wave_xor(float v) { return __uint_as_float(wave_xor_u32<O>(__float_as_uint(v))); }
template <int O> __device__ __forceinline__ double wave_xor(double v) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = wave_xor_u32<O>((uint32_t)u), hi = wave_xor_u32<O>((uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// the six steps of a descending butterfly: GMS_BUTTERFLY(STEP) expands STEP(32) ... STEP(1)
#define GMS_BUTTERFLY(STEP) STEP(32) STEP(16) STEP(8) STEP(4) STEP(2) STEP(1)

__device__ __forceinline__ double wave_sum_f64(double v) {   // fixed xor-butterfly shape
#define GMS_STEP_(O) v += wave_xor<O>(v);
    GMS_BUTTERFLY(GMS_STEP_)
#undef GMS_STEP_
    return v;
}
