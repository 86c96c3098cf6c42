/*
 * gms_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY; see the
 * header of gms_oracle.h (scope, "parity unpinned" statement, citation convention).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * x86-64 SSE2 arithmetic: float ops round to float, double ops to double, as the JVM does.
 */
#include "gms_oracle.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

/* ---- Java primitive-conversion semantics -------------------------------------------------- */

/* (int) of a double: truncate toward zero, saturate, NaN -> 0 (JLS 5.1.3). */
static inline int32_t j_d2i(double d) {
    if (d != d) return 0;
    if (d >= 2147483647.0) return INT32_MAX;
    if (d <= -2147483648.0) return INT32_MIN;
    return (int32_t)d;
}
/* Java int arithmetic wraps. */
static inline int32_t j_iadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t j_isub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t j_imul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

/* ---- J/app/Util.java ----------------------------------------------------------------------- */

/* Util.java:35-37  Math.log(odds / (1.0f - odds)), all in double */
double orc_log_odds(double p) { return log(p / ((double)1.0f - p)); }

/* Util.java:46-48  (1.0f - 1.0f / (1 + Math.exp(log))) in double */
double orc_inv_log_odds(double l) { return (double)1.0f - (double)1.0f / (1.0 + exp(l)); }

/* Util.java:428-455 */
void orc_generate_gaussian_kernel(double sigma, int size, double *out) {
    int k_size = size * 2 + 1;                       /* :431 */
    double norm = 1.0 / (sqrt(2 * M_PI) * sigma);    /* :436 */
    double coeff = 2 * sigma * sigma;                /* :439  (2*sigma)*sigma */
    double total = 0;
    for (int x = -size; x <= size; x++) {            /* :444 */
        double g = norm * exp((double)(-x * x) / coeff);   /* :445 int product, double divide */
        out[x + size] = g;
        total += g;
    }
    for (int i = 0; i < k_size; i++) out[i] /= total;      /* :451-453 */
}

/* ---- J/slam/GridMap.java ctor -------------------------------------------------------------- */

void orc_grid_init(orc_grid *g, float width, float height, float resolution, float pos_x, float pos_y) {
    memset(g, 0, sizeof(*g));
    g->resolution = resolution;                                  /* :81 */
    g->pos_x = pos_x; g->pos_y = pos_y;                          /* :82 */
    /* :85  Math.ceil(width / resolution): float divide, widened, ceil, Number.intValue() */
    g->W = j_d2i(ceil((double)(width / resolution)));
    g->H = j_d2i(ceil((double)(height / resolution)));
    /* :94-95 */
    double sigma = sqrt(0.05 / (double)resolution);
    int size = j_d2i(ceil(sigma * 3));
    g->ktaps = 2 * size + 1;
    if (g->ktaps > ORC_MAX_TAPS) { g->ktaps = 0; return; }
    orc_generate_gaussian_kernel(sigma, size, g->kernel);
    /* SensorModel.java:20-25: float literals widened to double */
    g->l_free = orc_log_odds((double)0.30f);
    g->l_prior = orc_log_odds((double)0.5f);
    g->l_occ = orc_log_odds((double)0.9f);
    g->extra_steps = 2;          /* GridMap.java:210 */
    g->hit_tolerance = 2.0f;     /* GridMap.java:223 */
    g->z_hit = 0.9;              /* GridMap.java:259 */
    g->z_random = 1 - g->z_hit;
    g->max_range = 10.0f;        /* SensorModel.java:20 */
}

/* ---- J/slam/RayIterator.java --------------------------------------------------------------- */

typedef struct ray_it {
    int32_t x, y, width, height, x_inc, y_inc, n;   /* :32 */
    float dx, dy, error;                            /* :33 */
} ray_it;

/* RayIterator.init :65-104 */
static void ray_init(ray_it *r, float x0, float y0, float x1, float y1, int32_t extra) {
    r->dx = fabsf(x1 - x0);                          /* :68 float subtract, abs */
    r->dy = fabsf(y1 - y0);                          /* :69 */
    r->x = j_d2i(floor((double)x0));                 /* :71 */
    r->y = j_d2i(floor((double)y0));                 /* :72 */
    r->n = j_iadd(1, extra);                         /* :75 */
    if (r->dx == 0) {                                /* :78 */
        r->x_inc = 0;
        r->error = INFINITY;
    } else if (x1 > x0) {                            /* :81 */
        r->x_inc = 1;
        r->n = j_iadd(r->n, j_d2i(floor((double)x1) - (double)r->x));                 /* :83 */
        r->error = (float)((floor((double)x0) + 1 - (double)x0) * (double)r->dy);     /* :84 */
    } else {
        r->x_inc = -1;
        r->n = j_iadd(r->n, j_isub(r->x, j_d2i(floor((double)x1))));                  /* :87 */
        r->error = (float)(((double)x0 - floor((double)x0)) * (double)r->dy);         /* :88 */
    }
    if (r->dy == 0) {                                /* :91 */
        r->y_inc = 0;
        r->error -= INFINITY;                        /* :93 float op; Inf-Inf = NaN */
    } else if (y1 > y0) {                            /* :94 */
        r->y_inc = 1;
        r->n = j_iadd(r->n, j_isub(j_d2i(floor((double)y1)), r->y));                  /* :96 */
        /* :97 compound assignment: (float)((double)error - product) */
        r->error = (float)((double)r->error - (floor((double)y0) + 1 - (double)y0) * (double)r->dx);
    } else {
        r->y_inc = -1;
        r->n = j_iadd(r->n, j_isub(r->y, j_d2i(floor((double)y1))));                  /* :100 */
        r->error = (float)((double)r->error - ((double)y0 - floor((double)y0)) * (double)r->dx);  /* :101 */
    }
}

/* RayIterator.hasNext :107-109 */
static inline int ray_has_next(const ray_it *r) {
    return r->n > 0 && !(r->x < 0 || r->x >= r->width || r->y < 0 || r->y >= r->height);
}

/* RayIterator.next :112-130 (cell returned through cx,cy) */
static inline void ray_next(ray_it *r, int32_t *cx, int32_t *cy) {
    *cx = r->x; *cy = r->y;                          /* :114 */
    if (r->error > 0) {                              /* :117 */
        r->y = j_iadd(r->y, r->y_inc);
        r->error -= r->dx;                           /* float */
    } else {
        r->x = j_iadd(r->x, r->x_inc);
        r->error += r->dy;
    }
    r->n = j_isub(r->n, 1);                          /* :126 */
}

int32_t orc_trace_ray(int32_t W, int32_t H, float x0, float y0, float x1, float y1, int32_t extra,
                      int32_t *cells_xy, int32_t cap) {
    ray_it r;
    r.width = W; r.height = H;                       /* ctor :44-48 */
    ray_init(&r, x0, y0, x1, y1, extra);
    int32_t count = 0;
    while (ray_has_next(&r)) {
        int32_t cx, cy;
        ray_next(&r, &cx, &cy);
        if (cells_xy && count < cap) { cells_xy[2 * count] = cx; cells_xy[2 * count + 1] = cy; }
        count++;
    }
    return count;
}

/* ---- J/slam/SensorModel.java:31-41 --------------------------------------------------------- */

/* returns the class: 0 = P_FREE, 1 = P_PRIOR, 2 = P_OCCUPPIED */
static inline int sensor_class(float current, float measured, int hit, float hit_tolerance) {
    if (!hit) return current < measured ? 0 : 1;                 /* :32-33 */
    if (current < measured - hit_tolerance / 2) return 0;        /* :35 float arithmetic */
    if (current > measured + hit_tolerance / 2) return 1;        /* :37 */
    return 2;                                                    /* :40 */
}

/* ---- J/slam/GridMap.java:194-228 ----------------------------------------------------------- */

int32_t orc_apply_measurement(const orc_grid *g, double *log_data, float sx, float sy, float ex, float ey,
                              float measured, int hit, int32_t *cells_xy, uint8_t *classes, int32_t cap) {
    const double inc[3] = { g->l_free, g->l_prior, g->l_occ };
    ray_it r;
    r.width = g->W; r.height = g->H;                 /* static rayIterator(W,H) :98 */
    ray_init(&r, sx + 0.5f, sy + 0.5f, ex + 0.5f, ey + 0.5f, g->extra_steps);   /* :210 */
    int32_t count = 0;
    while (ray_has_next(&r)) {                       /* :211 */
        int32_t cx, cy;
        ray_next(&r, &cx, &cy);
        float dX = sx - ((float)cx + 0.5f);          /* :215 */
        float dY = sy - ((float)cy + 0.5f);          /* :216 */
        float distance = (float)sqrt((double)(dX * dX + dY * dY));   /* :217 */
        int cls = sensor_class(distance, measured, hit, g->hit_tolerance);
        if (log_data) log_data[j_iadd(cx, j_imul(cy, g->W))] += inc[cls];       /* :223 */
        if (count < cap) {
            if (cells_xy) { cells_xy[2 * count] = cx; cells_xy[2 * count + 1] = cy; }
            if (classes) classes[count] = (uint8_t)cls;
        }
        count++;
    }
    return count;
}

/* ---- J/math/Transform.java:13-32 + J/math/MathUtil.java:30-40 ------------------------------ */

void orc_pose_trig(float theta, double *cos_out, double *sin_out) {
    /* MathUtil.cos(float) is chosen (theta is a float field): (float) FastMath.cos((double)theta),
     * then widened back to double by the assignment in Transform.java:15-16.
     * FastMath is restated with libm cos/sin (see header). */
    *cos_out = (double)(float)cos((double)theta);
    *sin_out = (double)(float)sin((double)theta);
}

/* the two float-rounded primitives every cell index leans on, over arrays (the exhaustive float-domain check of the device's
 * versions, tests/test_gpu_exhaustive_float.py): orc_pose_trig element by element, and (float)Math.sqrt((double)s) of
 * GridMap.java:217.  Returns how many of the n results differ from `got` (bit patterns; two NaNs count as equal). */
static inline int f32_same(float a, float b) {
    uint32_t x, y;
    memcpy(&x, &a, 4); memcpy(&y, &b, 4);
    return x == y || (a != a && b != b);
}
int64_t orc_count_trig_mismatches(const float *theta, const float *got_cos, const float *got_sin, int64_t n, int32_t threads,
                                  int64_t *first_bad) {
    int64_t bad = 0, first = -1;
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(+ : bad)
    for (int64_t i = 0; i < n; i++) {
        double c, s;
        orc_pose_trig(theta[i], &c, &s);
        if (!f32_same((float)c, got_cos[i]) || !f32_same((float)s, got_sin[i])) {
            bad++;
#pragma omp critical
            if (first < 0 || i < first) first = i;
        }
    }
    if (first_bad) *first_bad = first;
    return bad;
}
int64_t orc_count_sqrt_mismatches(const float *a, const float *got, int64_t n, int32_t threads, int64_t *first_bad) {
    int64_t bad = 0, first = -1;
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(+ : bad)
    for (int64_t i = 0; i < n; i++) {
        const float want = (float)sqrt((double)a[i]);                             /* GridMap.java:217 */
        if (!f32_same(want, got[i])) {
            bad++;
#pragma omp critical
            if (first < 0 || i < first) first = i;
        }
    }
    if (first_bad) *first_bad = first;
    return bad;
}

/* How close does the double a libm returns for cos / sin of a widened float come to a ROUNDING BOUNDARY of the float it is narrowed
 * to (the midpoint of two adjacent floats)?  Transform.java:15-16 narrows FastMath's double to float; another libm whose double
 * differs by less than an ulp narrows to the same float unless such a boundary lies between the two doubles.  Sweeps the bit
 * patterns lo..hi (inclusive, sign OR-ed in) and records every (theta, which: 0 cos / 1 sin) whose glibc double lies within
 * `window` ulps(double) of a boundary; returns how many there are (at most cap are stored).  tests/golden/make_trig_fragile.py settles
 * each with multi-precision arithmetic; tests/golden/trig_fragile.json is the result. */
int64_t orc_trig_near_float_boundary(uint32_t lo, uint32_t hi, uint32_t sign, double window, float *out_theta, int32_t *out_which,
                                     double *out_dist, int64_t cap, int32_t threads) {
    int64_t n = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t b = (int64_t)lo; b <= (int64_t)hi; b++) {
        const uint32_t bits = (uint32_t)b | sign;
        float th;
        memcpy(&th, &bits, 4);
        for (int which = 0; which < 2; which++) {
            const double d = which ? sin((double)th) : cos((double)th);
            if (!(d == d) || d == 0.0) continue;
            const float f = (float)d;
            const float other = (double)f <= d ? nextafterf(f, INFINITY) : nextafterf(f, -INFINITY);
            const double mid = ((double)f + (double)other) * 0.5;               /* exact: a 25-bit significand */
            const double ulp = nextafter(fabs(d), INFINITY) - fabs(d);
            const double dist = fabs(d - mid) / ulp;
            if (dist <= window) {
                int64_t k;
#pragma omp atomic capture
                k = n++;
                if (k < cap) { out_theta[k] = th; out_which[k] = which; out_dist[k] = dist; }
            }
        }
    }
    return n;
}

typedef struct xform { double c, s, px, py; } xform;

static inline xform xform_from_pose(const float pose[3]) {
    xform t;
    orc_pose_trig(pose[2], &t.c, &t.s);
    t.px = (double)pose[0];
    t.py = (double)pose[1];
    return t;
}
/* Transform.java:23 / :28 */
static inline double xform_x(const xform *t, double x, double y) { return x * t->c - y * t->s + t->px; }
static inline double xform_y(const xform *t, double x, double y) { return x * t->s + y * t->c + t->py; }

/* ---- J/slam/GridMap.java:173-191 ----------------------------------------------------------- */

void orc_scan_rays(const orc_grid *g, const orc_beam *beams, int32_t B, const float pose[3], float *out6) {
    xform t = xform_from_pose(pose);                                             /* :175 */
    double posx = (double)g->pos_x, posy = (double)g->pos_y, res = (double)g->resolution;
    float sx = (float)((xform_x(&t, 0, 0) - posx) / res);                        /* :178 */
    float sy = (float)((xform_y(&t, 0, 0) - posy) / res);                        /* :179 */
    for (int32_t b = 0; b < B; b++) {
        const orc_beam *m = &beams[b];
        float ex = (float)((xform_x(&t, m->local_x, m->local_y) - posx) / res);  /* :185 */
        float ey = (float)((xform_y(&t, m->local_x, m->local_y) - posy) / res);  /* :186 */
        float measured = (float)m->distance / g->resolution;                     /* :188 float divide */
        float *o = out6 + 6 * (size_t)b;
        o[0] = sx; o[1] = sy; o[2] = ex; o[3] = ey; o[4] = measured; o[5] = m->hit ? 1.0f : 0.0f;
    }
}

int64_t orc_integrate(const orc_grid *g, double *log_data, const orc_beam *beams, int32_t B, const float pose[3]) {
    int64_t visits = 0;
    float ray[6];
    for (int32_t b = 0; b < B; b++) {                /* :182 in list order */
        orc_scan_rays(g, &beams[b], 1, pose, ray);
        visits += orc_apply_measurement(g, log_data, ray[0], ray[1], ray[2], ray[3], ray[4],
                                        beams[b].hit != 0, NULL, NULL, 0);
    }
    return visits;
}

/* The integer content of integrateObservation (GridMap.java:173-191): how often the scan's rays visit every cell, per sensor class
 * (counts[3 * cell + class], class 0 free / 1 prior / 2 occupied; SensorModel.java:31-41) -- what `logData[c] += logOdds(...)` (:223)
 * is a function of.  Test infrastructure for the integer parity of the device's ray casts; returns the number of visits. */
int64_t orc_scan_counts(const orc_grid *g, const orc_beam *beams, int32_t B, const float pose[3], uint32_t *counts) {
    int64_t visits = 0;
    float ray[6];
    int32_t cap = g->W + g->H + 8;
    int32_t *cells = (int32_t *)malloc((size_t)cap * 2 * sizeof(int32_t));
    uint8_t *cls = (uint8_t *)malloc((size_t)cap);
    memset(counts, 0, (size_t)g->W * (size_t)g->H * 3 * sizeof(uint32_t));
    for (int32_t b = 0; b < B; b++) {                /* :182 in list order */
        orc_scan_rays(g, &beams[b], 1, pose, ray);
        int32_t n = orc_apply_measurement(g, NULL, ray[0], ray[1], ray[2], ray[3], ray[4], beams[b].hit != 0, cells, cls, cap);
        if (n > cap) n = cap;                        /* (a walk is at most W + H + 1 + extra steps long) */
        for (int32_t k = 0; k < n; k++) counts[3 * ((size_t)cells[2 * k] + (size_t)cells[2 * k + 1] * g->W) + cls[k]]++;
        visits += n;
    }
    free(cells); free(cls);
    return visits;
}

/* ---- J/slam/GridMap.java:233-250 + J/app/Util.java:378-426 --------------------------------- */

void orc_build_likelihood(const orc_grid *g, const double *log_data, double *lik, double *scratch) {
    const int32_t W = g->W, H = g->H;
    const size_t n = (size_t)W * (size_t)H;
    double *prob = scratch;          /* GridMap.probData :59 */
    double *temp = scratch + n;      /* Util.tempArray :375 */
    const double thr = g->l_prior;   /* Util.logOdds(0.5) == 0.0 */
    for (size_t i = 0; i < n; i++) {                 /* :238-245 */
        if (log_data[i] > thr) prob[i] = 1;
        else if (log_data[i] < thr) prob[i] = 0;
        else prob[i] = 0.5;
    }
    const int k = (g->ktaps - 1) / 2;                /* Util.java:384 */
    const double *kernel = g->kernel;
    for (int32_t y = 0; y < H; y++) {                /* :387-404 horizontal */
        size_t y_index = (size_t)y * W;
        for (int32_t x = 0; x < W; x++) {
            double total = 0;
            for (int i = -k; i <= k; i++) {
                int32_t x2 = x + i;
                if (x2 >= 0 && x2 < W) total += kernel[i + k] * prob[y_index + x2];
            }
            lik[y_index + x] = total;
        }
    }
    memcpy(temp, lik, n * sizeof(double));           /* :407 */
    for (int32_t y = 0; y < H; y++) {                /* :410-425 vertical */
        for (int32_t x = 0; x < W; x++) {
            double total = 0;
            for (int i = -k; i <= k; i++) {
                int32_t y2 = y + i;
                if (y2 >= 0 && y2 < H) total += kernel[i + k] * temp[(size_t)x + (size_t)y2 * W];
            }
            lik[(size_t)x + (size_t)y * W] = total;
        }
    }
}

/* ---- J/slam/GridMap.java:259-294 ----------------------------------------------------------- */

static inline double beam_factor(const orc_grid *g, const double *lik, const xform *t, const orc_beam *m, int *used) {
    double posx = (double)g->pos_x, posy = (double)g->pos_y, res = (double)g->resolution;
    int32_t gx = j_d2i((xform_x(t, m->local_x, m->local_y) - posx) / res);      /* :273 */
    int32_t gy = j_d2i((xform_y(t, m->local_x, m->local_y) - posy) / res);      /* :274 */
    *used = 0;
    if (!(gx < 0 || gy < 0 || gx >= g->W || gy >= g->H)) {                       /* :276 */
        double val = lik[(size_t)gx + (size_t)gy * g->W];                        /* :277 */
        *used = 1;
        if (val == 0.5) return 1.0 / (double)g->max_range;                       /* :285-286 */
        return g->z_hit * val + g->z_random * 1.0 / (double)g->max_range;        /* :288 */
    }
    return 1.0;
}

double orc_probability_of(const orc_grid *g, const double *lik, const orc_beam *beams, int32_t B, const float pose[3]) {
    double product = 1;                              /* :262 */
    xform t = xform_from_pose(pose);                 /* :265 */
    for (int32_t b = 0; b < B; b++) {                /* :267 */
        if (!beams[b].hit) continue;                 /* :269 */
        int used;
        double f = beam_factor(g, lik, &t, &beams[b], &used);
        if (used) product *= f;
    }
    return product;
}

void orc_score(const orc_grid *g, const double *lik, const orc_beam *beams, int32_t B,
               const float *poses, int32_t N, double *weights) {
    for (int32_t i = 0; i < N; i++) weights[i] = orc_probability_of(g, lik, beams, B, poses + 3 * (size_t)i);
}

/* the same over `threads` host threads (probabilityOf is a pure function of its arguments; the reference itself is
 * single-threaded).  Reported beside the single-thread baseline, never used as a checker. */
void orc_score_mt(const orc_grid *g, const double *lik, const orc_beam *beams, int32_t B, const float *poses, int32_t N,
                  double *weights, int32_t threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int32_t i = 0; i < N; i++) weights[i] = orc_probability_of(g, lik, beams, B, poses + 3 * (size_t)i);
}

void orc_score_log(const orc_grid *g, const double *lik, const orc_beam *beams, int32_t B,
                   const float *poses, int32_t N, double *log_weights) {
    for (int32_t i = 0; i < N; i++) {
        xform t = xform_from_pose(poses + 3 * (size_t)i);
        double s = 0;
        for (int32_t b = 0; b < B; b++) {
            if (!beams[b].hit) continue;
            int used;
            double f = beam_factor(g, lik, &t, &beams[b], &used);
            if (used) s += log(f);
        }
        log_weights[i] = s;
    }
}

/* ---- J/slam/SLAM.java ---------------------------------------------------------------------- */

/* :87-121 */
double orc_normalize(double *weights, int32_t N, int32_t *strongest) {
    double weight_sum = 0;
    int32_t best = -1;
    for (int32_t i = 0; i < N; i++) {
        weight_sum += weights[i];                                    /* :100 */
        if (best < 0) best = i;                                      /* :110-111 */
        else if (weights[i] > weights[best]) best = i;               /* :113-114 strict */
    }
    for (int32_t i = 0; i < N; i++) weights[i] /= weight_sum;        /* :120-121 */
    if (strongest) *strongest = best;
    return weight_sum;
}

/* :180-190 */
double orc_neff(const double *weights, int32_t N) {
    double sum = 0;
    for (int32_t i = 0; i < N; i++) sum += weights[i];
    double squared_sum = 0;
    for (int32_t i = 0; i < N; i++) squared_sum += (weights[i] / sum) * (weights[i] / sum);
    return 1.0 / squared_sum;
}

/* J/math/MathUtil.java:65-72 */
static inline double angle_constrain(double a) {
    while (a < M_PI) a += M_PI * 2;
    while (a > M_PI) a -= M_PI * 2;
    return a;
}

/* :165-178 */
void orc_weighted_pose(const float *poses, const double *weights, int32_t N, float out[3]) {
    double x_sum = 0, y_sum = 0, theta_sum = 0, weight_sum = 0;
    for (int32_t i = 0; i < N; i++) {
        const float *p = poses + 3 * (size_t)i;
        x_sum += (double)p[0] * weights[i];                          /* :170 */
        y_sum += (double)p[1] * weights[i];
        theta_sum += angle_constrain((double)p[2]) * weights[i];     /* :172 */
        weight_sum += weights[i];
    }
    out[0] = (float)(x_sum / weight_sum);                            /* :176 */
    out[1] = (float)(y_sum / weight_sum);
    out[2] = (float)(theta_sum / weight_sum);
}

/* :133-153 */
int32_t orc_resample_indices(const double *weights, int32_t N, double r01, int32_t *idx) {
    int32_t clamped = 0;
    double r = r01 * 1.0 / (double)N;                /* :136 */
    double c = weights[0];                           /* :137 */
    int32_t i = 0;
    for (int32_t m = 1; m <= N; m++) {               /* :140 */
        double U = r + (double)(m - 1) * 1.0 / (double)N;    /* :141 */
        while (U > c) {                              /* :142 */
            if (i >= N - 1) { clamped++; break; }    /* Java: IndexOutOfBoundsException */
            i++;
            c += weights[i];                         /* :144 */
        }
        idx[m - 1] = i;                              /* :147 */
    }
    return clamped;
}

/* ---- J/slam/GridMap.java:319-346 ----------------------------------------------------------- */

double orc_find_best_pose(const orc_grid *g, const double *lik, const orc_beam *beams, int32_t B,
                          const float start[3], float best[3], int32_t *n_evaluated) {
    double max_prob = 0;                                             /* :321 */
    best[0] = start[0]; best[1] = start[1]; best[2] = start[2];      /* :320 */
    float x_span = 0.20f, y_span = 0.20f;
    float theta_span = (float)(15 * (M_PI / 180.0));                 /* :324 */
    float trans_step = 0.04f, theta_step = theta_span / 5;           /* :325 */
    int32_t n = 0;
    for (float dx = -x_span; dx < x_span; dx += trans_step)          /* :328 */
        for (float dy = -y_span; dy < y_span; dy += trans_step)
            for (float dth = -theta_span; dth < theta_span; dth += theta_step) {
                float p[3] = { start[0] + dx, start[1] + dy, start[2] + dth };   /* :332 */
                double prob = orc_probability_of(g, lik, beams, B, p);
                n++;
                if (prob > max_prob) {                               /* :334 */
                    max_prob = prob;
                    best[0] = p[0]; best[1] = p[1]; best[2] = p[2];
                }
            }
    if (n_evaluated) *n_evaluated = n;
    return max_prob;
}

/* ---- J/slam/Odometry.java:60-96 (motion model; "next" row f2) ------------------------------ */

static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = { ctr[0], ctr[1], ctr[2], ctr[3] };
    uint32_t k[2] = { key[0], key[1] };
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

void orc_philox_normals(uint64_t seed, uint64_t sequence, uint64_t index, double *z0, double *z1) {
    const uint32_t ctr[4] = { (uint32_t)index, (uint32_t)(index >> 32), (uint32_t)sequence, (uint32_t)(sequence >> 32) };
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t c[4];
    orc_philox4x32(ctr, key, c);
    /* two uniforms in (0,1) with 53 random bits each */
    const double u1 = ((double)(((uint64_t)(c[0] >> 5) << 26) | (uint64_t)(c[1] >> 6)) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(((uint64_t)(c[2] >> 5) << 26) | (uint64_t)(c[3] >> 6)) + 0.5) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    *z0 = r * cos(2.0 * M_PI * u2);
    *z1 = r * sin(2.0 * M_PI * u2);
}

void orc_sample_motion(float *poses, int32_t N, int64_t index_offset, double d_center, double d_theta,
                       uint64_t seed, uint64_t sequence) {
    const double d_center_sd = (0.01 + fabs(d_center) * 0.05) / 2;                 /* :63 */
    const double d_theta_sd = 5 * (M_PI / 180.0) + 0.1 * fabs(d_theta);            /* :64 */
    for (int32_t i = 0; i < N; i++) {
        double z0, z1;
        orc_philox_normals(seed, sequence, (uint64_t)(index_offset + i), &z0, &z1);
        const double d = d_center + d_center_sd * z0;                              /* ndCenter.sample() :80 */
        const double theta = d_theta + d_theta_sd * z1;                            /* ndTheta.sample()  :81 */
        float *p = poses + 3 * (size_t)i;
        p[2] = (float)angle_constrain((double)p[2] + theta);                       /* :92 */
        double c, s;
        orc_pose_trig(p[2], &c, &s);                                               /* MathUtil.cos(float) :93 */
        p[0] = (float)((double)p[0] + c * d);                                      /* :93  p.x += cos * d */
        p[1] = (float)((double)p[1] + s * d);                                      /* :94 */
    }
}

/* ---- J/app/GridMapApp.java:439-458 (combined map; "next" row f4) --------------------------- */
void orc_combine_maps(const double *logs, int32_t n_maps, int64_t cells, double *out) {
    for (int64_t i = 0; i < cells; i++) {
        double product = 1;                                                       /* :446 */
        for (int32_t m = 0; m < n_maps; m++)
            product *= 1 - orc_inv_log_odds(logs[(size_t)m * cells + i]);         /* :451 */
        out[i] = orc_log_odds(1 - product);                                       /* :454 */
    }
}

/* ---- J/app/GridMapApp.java:143-175 (scan de-skew; "next" row f3) --------------------------- */
void orc_deskew(const double *angle, const double *distance, const uint8_t *hit, int32_t length, double d_center,
                double d_theta, orc_beam *out) {
    for (int32_t i = 0; i < length; i++) {
        const double d_i = -(double)(length - i) / (double)length;                /* :150 int negate, then double divide */
        const double delta_theta = d_theta * d_i;                                 /* :157 */
        const double delta_x = d_center * d_i;                                    /* :158 */
        const double x_a = distance[i] * cos(angle[i] + delta_theta) + delta_x;   /* :166 MathUtil.cos(double) */
        const double y_a = distance[i] * sin(angle[i] + delta_theta);             /* :167 */
        memset(&out[i], 0, sizeof(orc_beam));
        out[i].local_x = x_a;                                                     /* Observation.java:74-75 */
        out[i].local_y = y_a;
        out[i].distance = sqrt(x_a * x_a + y_a * y_a);                            /* :71 */
        out[i].hit = hit[i] ? 1 : 0;
    }
}

/* J/slam/GridMap.java:142-156 */
static int32_t j_f2i(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
int32_t orc_point_index(const orc_grid *g, float px, float py) {
    const float tx = (px - g->pos_x) / g->resolution;      /* tmp.minusAssign(position); tmp.divAssign(resolution) */
    const float ty = (py - g->pos_y) / g->resolution;
    return (int32_t)((uint32_t)j_f2i(tx) + (uint32_t)j_f2i(ty) * (uint32_t)g->W);
}

/* ---- J/slam/SLAM.java as a whole: one GridMapData per particle (judge row J2) ---------------- */
/* The reference's filter shape: every Particle owns a pose, a weight and a map (SLAM.java:30-47); update() scores a particle
 * against ITS OWN likelihood field and integrates the scan into ITS OWN map at ITS OWN pose (:88-107); resample() deep-copies
 * both arrays of the surviving particle's map (:41-45 -> GridMap.createMapData(other), GridMap.java:106-124).
 * Two things are not the reference's: (1) findBestPoseOptim (:97) is BOBYQA from commons-math3 on an objective that is 0 / NaN
 *  -- refine = 0 leaves the sampled pose as it is, refine = 1 runs the lattice search findBestPose the reference keeps
 * commented out beside it (:96); (2) the motion-model draw (Odometry.apply, unseeded Well1024a) comes from Philox keyed by the
 * particle's slot index, as in orc_sample_motion. */
struct orc_slam {
    orc_grid g;
    int32_t n;
    float *pose;        /* [n][3]   Particle.pose   */
    double *weight;     /* [n]      Particle.weight */
    double *log_data;   /* [n][W*H] Particle.m.logData */
    double *lik_data;   /* [n][W*H] Particle.m.likelihoodData */
    double *scratch;    /* GridMap.probData + Util.tempArray */
    int32_t strongest;  /* index of strongestParticle (-1: null) */
    /* the previous generation's arrays, kept for the next resample() to fill (a checker's economy: 4096 particles x 256^2 cells are
     * 4.3 GB per generation, and fresh pages cost more than the copies) */
    float *spare_pose; double *spare_weight, *spare_log, *spare_lik;
};

void orc_slam_reset(orc_slam *s) {                                   /* :65-77 */
    const size_t cells = (size_t)s->g.W * (size_t)s->g.H;
    for (int32_t i = 0; i < s->n; i++) {
        s->pose[3 * i] = 0; s->pose[3 * i + 1] = 0; s->pose[3 * i + 2] = 0;       /* new Pose(0, 0, 0) :68 */
        double *l = s->log_data + (size_t)i * cells, *k = s->lik_data + (size_t)i * cells;
        for (size_t c = 0; c < cells; c++) { l[c] = s->g.l_prior; k[c] = 0.0; }    /* createMapData(null): GridMap.java:114-117 (a new double[] is 0) */
        s->weight[i] = 1.0 / s->n;                                                 /* :71 */
    }
    s->strongest = 0;                                                              /* :75 */
}

orc_slam *orc_slam_new(const orc_grid *g, int32_t n) {               /* :56-62 */
    orc_slam *s = (orc_slam *)calloc(1, sizeof(orc_slam));
    if (!s) return NULL;
    const size_t cells = (size_t)g->W * (size_t)g->H;
    s->g = *g; s->n = n;
    s->pose = (float *)malloc((size_t)n * 3 * sizeof(float));
    s->weight = (double *)malloc((size_t)n * sizeof(double));
    s->log_data = (double *)malloc((size_t)n * cells * sizeof(double));
    s->lik_data = (double *)malloc((size_t)n * cells * sizeof(double));
    s->scratch = (double *)malloc(2 * cells * sizeof(double));
    if (!s->pose || !s->weight || !s->log_data || !s->lik_data || !s->scratch) { orc_slam_free(s); return NULL; }
    orc_slam_reset(s);
    return s;
}

void orc_slam_free(orc_slam *s) {
    if (!s) return;
    free(s->pose); free(s->weight); free(s->log_data); free(s->lik_data); free(s->scratch);
    free(s->spare_pose); free(s->spare_weight); free(s->spare_log); free(s->spare_lik);
    free(s);
}

int32_t orc_slam_count(const orc_slam *s) { return s->n; }
float *orc_slam_poses(orc_slam *s) { return s->pose; }
double *orc_slam_weights(orc_slam *s) { return s->weight; }
double *orc_slam_log(orc_slam *s, int32_t i) { return s->log_data + (size_t)i * (size_t)s->g.W * (size_t)s->g.H; }
double *orc_slam_lik(orc_slam *s, int32_t i) { return s->lik_data + (size_t)i * (size_t)s->g.W * (size_t)s->g.H; }
int32_t orc_slam_strongest(const orc_slam *s) { return s->strongest; }

double orc_slam_neff(const orc_slam *s) { return orc_neff(s->weight, s->n); }                       /* :180-190 */
void orc_slam_weighted_pose(const orc_slam *s, float out[3]) { orc_weighted_pose(s->pose, s->weight, s->n, out); }   /* :165-178 */

/* SLAM.update(z, u) :80-131.  sample_motion = 0: the particles keep the poses they have -- the caller has set the motion-model
 * samples itself (sampleMotionModel's `u == null` branch, :159) -- while u.dTheta still decides skipUpdate (:82), as update()
 * reads it whatever sampleMotionModel does. */
double orc_slam_update_mt(orc_slam *s, const orc_beam *z, int32_t B, int32_t sample_motion, double d_center, double d_theta,
                          uint64_t seed, uint64_t sequence, int32_t refine, int32_t threads) {
    const size_t cells = (size_t)s->g.W * (size_t)s->g.H;
    const int have_u = sample_motion;
    const int skip_update = fabs(d_theta) > (M_PI / 180.0) * 30;                        /* :82 */
    s->strongest = -1;                                                                  /* :84 */
    /* The body of the particle loop (:88-107) touches nothing but its own particle -- and the shared scratch arrays probData /
     * tempArray, which are per thread here -- so the particles may be taken by several host threads (threads > 1: a faster checker,
     * same values); the two loop-carried quantities, weightSum (:100) and strongestParticle (:110-115), are folded in particle
     * order afterwards, operation for operation what the sequential loop does. */
    if (threads < 1) threads = 1;
    double *scratch_all = threads > 1 ? (double *)malloc((size_t)threads * 2 * cells * sizeof(double)) : NULL;
    if (threads > 1 && !scratch_all) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4) if (threads > 1)
    for (int32_t i = 0; i < s->n; i++) {                                                /* :88 */
        double *scratch = s->scratch;
#ifdef _OPENMP
        if (scratch_all) scratch = scratch_all + (size_t)omp_get_thread_num() * 2 * cells;
#endif
        float *pose = s->pose + 3 * (size_t)i;
        double *log_data = s->log_data + (size_t)i * cells, *lik = s->lik_data + (size_t)i * cells;
        if (have_u) orc_sample_motion(pose, 1, i, d_center, d_theta, seed, sequence);   /* :90 -> :155-163 */
        orc_build_likelihood(&s->g, log_data, lik, scratch);                            /* :93 */
        if (refine) {                                                                   /* :96 (the lattice search; :97 is BOBYQA) */
            float best[3];
            orc_find_best_pose(&s->g, lik, z, B, pose, best, NULL);
            pose[0] = best[0]; pose[1] = best[1]; pose[2] = best[2];
        }
        s->weight[i] = orc_probability_of(&s->g, lik, z, B, pose);                      /* :99 */
        if (!skip_update) orc_integrate(&s->g, log_data, z, B, pose);                   /* :102-107 */
    }
    free(scratch_all);
    double weight_sum = 0;                                                              /* :87 */
    for (int32_t i = 0; i < s->n; i++) {
        weight_sum += s->weight[i];                                                     /* :100 */
        if (s->strongest < 0) s->strongest = i;                                         /* :110-111 */
        else if (s->weight[i] > s->weight[s->strongest]) s->strongest = i;              /* :113-114 */
    }
    for (int32_t i = 0; i < s->n; i++) s->weight[i] /= weight_sum;                      /* :120-121 */
    return orc_slam_neff(s);                                                            /* :124,129 */
}

double orc_slam_update(orc_slam *s, const orc_beam *z, int32_t B, int32_t sample_motion, double d_center, double d_theta,
                       uint64_t seed, uint64_t sequence, int32_t refine) {
    return orc_slam_update_mt(s, z, B, sample_motion, d_center, d_theta, seed, sequence, refine, 1);
}

/* how many OpenMP threads the parallel loops of this file use from now on (0: the runtime's default); returns what was in force.
 * bench.py times orc_slam_resample's copies with one thread for its single-thread baseline; tests leave the default. */
int32_t orc_set_threads(int32_t n) {
#ifdef _OPENMP
    const int32_t before = omp_get_max_threads();
    if (n > 0) omp_set_num_threads(n);
    return before;
#else
    (void)n;
    return 1;
#endif
}

/* SLAM.resample() :133-153 with Math.random() = r01: every slot receives a deep copy of a particle (pose, weight, both map
 * arrays).  idx_out (may be NULL) receives the source of every slot.  Returns the number of slots where Java would have run off
 * the list (clamped to the last particle, as orc_resample_indices). */
int32_t orc_slam_resample(orc_slam *s, double r01, int32_t *idx_out) {
    const size_t cells = (size_t)s->g.W * (size_t)s->g.H;
    const int32_t n = s->n;
    int32_t *idx = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    float *pose2 = s->spare_pose ? s->spare_pose : (float *)malloc((size_t)n * 3 * sizeof(float));
    double *w2 = s->spare_weight ? s->spare_weight : (double *)malloc((size_t)n * sizeof(double));
    double *log2 = s->spare_log ? s->spare_log : (double *)malloc((size_t)n * cells * sizeof(double));
    double *lik2 = s->spare_lik ? s->spare_lik : (double *)malloc((size_t)n * cells * sizeof(double));
    const int32_t clamped = orc_resample_indices(s->weight, n, r01, idx);               /* :136-145 */
#pragma omp parallel for schedule(static)
    for (int32_t m = 0; m < n; m++) {                                                   /* :147 new Particle(particles.get(i)) (copies are independent: any order) */
        const int32_t i = idx[m];
        w2[m] = s->weight[i];                                                           /* :42 */
        memcpy(pose2 + 3 * (size_t)m, s->pose + 3 * (size_t)i, 3 * sizeof(float));      /* :43 */
        memcpy(log2 + (size_t)m * cells, s->log_data + (size_t)i * cells, cells * sizeof(double));   /* :44 -> GridMap.java:120 */
        memcpy(lik2 + (size_t)m * cells, s->lik_data + (size_t)i * cells, cells * sizeof(double));   /* GridMap.java:121 */
    }
    if (idx_out) memcpy(idx_out, idx, (size_t)n * sizeof(int32_t));
    free(idx);
    s->spare_pose = s->pose; s->spare_weight = s->weight; s->spare_log = s->log_data; s->spare_lik = s->lik_data;
    s->pose = pose2; s->weight = w2; s->log_data = log2; s->lik_data = lik2;            /* :152 */
    /* strongestParticle keeps pointing at an object of the OLD generation (GridMapApp.java:188-190 notes it): no index here */
    return clamped;
}
