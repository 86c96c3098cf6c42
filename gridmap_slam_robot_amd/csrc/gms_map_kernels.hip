// gms_map_kernels.hip -- gfx950 kernels for the log-odds map update and the likelihood field.
//
//   k_raycast     GridMap.integrateObservation / applyMeasurement + RayIterator + SensorModel
//                 (J/slam/GridMap.java:173-228, J/slam/RayIterator.java:65-130,
//                  J/slam/SensorModel.java:31-41): one lane per ray walks the 4-connected DDA and
//                 accumulates per-cell (n_free, n_occ) counts with 32-bit atomics.
//   k_apply       log += n_free*l_free + n_occ*l_occ over the touched bounding box (the three
//                 possible increments of GridMap.java:223 are constants: J/app/Util.java:35-37).
//   k_likelihood  GridMap.computeLikelihoodMap (GridMap.java:233-250) + Util.doGaussianBlurdSeparable
//                 (J/app/Util.java:378-426): threshold, horizontal and vertical pass fused through
//                 an LDS tile; every sum runs in the reference's tap order, without FMA.
//
// HBM layout: log/lik [n_maps][H][W] doubles, row-major x + y*W as GridMapData's arrays;
// cnt [n_maps][H][W] uint32 = n_free | n_occ << 16, all zero between calls.
#include "gms_device.h"

// ---------------------------------------------------------------------------------------------
// bbox encoding: {max(W-1-x), max(H-1-y), max(x+1), max(y+1)}; all-zero == empty, so a memset
// resets it and every update is an atomicMax.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bbox_decode(const int32_t *bb, int32_t W, int32_t H, int32_t &x0, int32_t &y0,
                                            int32_t &x1, int32_t &y1) {
    x0 = W - 1 - bb[0]; y0 = H - 1 - bb[1]; x1 = bb[2]; y1 = bb[3];   // [x0,x1) x [y0,y1)
}

template <bool TRACE>
__global__ void __launch_bounds__(64)
k_raycast(GridDev g, const gms_beam *__restrict__ beams, int32_t B, int32_t beam_stride,
          const float *__restrict__ poses, const RayIn *__restrict__ single, uint32_t *__restrict__ cnt,
          int32_t *__restrict__ bbox, int32_t *__restrict__ t_cells, uint8_t *__restrict__ t_cls, int32_t cap,
          int32_t *__restrict__ t_counts) {
    const int32_t mi = blockIdx.y;
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = b < B;

    RayIn ray;
    if (single) {
        ray = *single;
    } else if (active) {
        ray = make_ray(g, beams[(size_t)mi * beam_stride + b], poses + 3 * mi);
    } else {
        ray.sx = ray.sy = ray.ex = ray.ey = ray.measured = 0.0f;
        ray.hit = 0;
    }

    int32_t bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;   // encoded bbox contributions
    int32_t count = 0;
    if (active) {
        RayDev r;
        ray_init(r, ray.sx + 0.5f, ray.sy + 0.5f, ray.ex + 0.5f, ray.ey + 0.5f, g.extra);   // GridMap.java:210
        uint32_t *mcnt = TRACE ? nullptr : cnt + (size_t)mi * g.cells;
        while (ray_has_next(r, g.W, g.H)) {                                                  // :211
            const int32_t cx = r.x, cy = r.y;
            ray_step(r);
            const float d = cell_distance(ray.sx, ray.sy, cx, cy);                           // :215-217
            const int32_t cls = sensor_class(d, ray.measured, ray.hit, g.half_tol);          // :223
            if (TRACE) {
                if (count < cap) {
                    const size_t o = (size_t)b * cap + count;
                    if (t_cells) { t_cells[2 * o] = cx; t_cells[2 * o + 1] = cy; }
                    if (t_cls) t_cls[o] = (uint8_t)cls;
                }
            } else if (cls != 1) {
                atomicAdd(&mcnt[(size_t)cy * g.W + cx], cls == 0 ? 1u : 0x10000u);
                bx0 = max(bx0, g.W - 1 - cx); by0 = max(by0, g.H - 1 - cy);
                bx1 = max(bx1, cx + 1);       by1 = max(by1, cy + 1);
            }
            count++;
        }
    }
    if (TRACE) {
        if (active && t_counts) t_counts[b] = count;
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            bx0 = max(bx0, __shfl_xor(bx0, o, GMS_WAVE)); by0 = max(by0, __shfl_xor(by0, o, GMS_WAVE));
            bx1 = max(bx1, __shfl_xor(bx1, o, GMS_WAVE)); by1 = max(by1, __shfl_xor(by1, o, GMS_WAVE));
        }
        if ((threadIdx.x & 63) == 0 && bx1 > 0) {
            int32_t *bb = bbox + 4 * mi;
            atomicMax(&bb[0], bx0); atomicMax(&bb[1], by0); atomicMax(&bb[2], bx1); atomicMax(&bb[3], by1);
        }
    }
}

// plain RayIterator walk (gms_map_trace_ray)
__global__ void k_trace_ray(int32_t W, int32_t H, float x0, float y0, float x1, float y1, int32_t extra,
                            int32_t *__restrict__ cells, int32_t cap, int32_t *__restrict__ count_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    RayDev r;
    ray_init(r, x0, y0, x1, y1, extra);
    int32_t count = 0;
    while (ray_has_next(r, W, H)) {
        if (count < cap) { cells[2 * count] = r.x; cells[2 * count + 1] = r.y; }
        ray_step(r);
        count++;
    }
    *count_out = count;
}

// log += n_free*l_free + n_occ*l_occ on the touched box; counts cleared.
#define APPLY_TW 256
#define APPLY_TH 4
__global__ void __launch_bounds__(256)
k_apply(GridDev g, double *__restrict__ logd, uint32_t *__restrict__ cnt, const int32_t *__restrict__ bbox) {
    const int32_t mi = blockIdx.z;
    int32_t x0, y0, x1, y1;
    bbox_decode(bbox + 4 * mi, g.W, g.H, x0, y0, x1, y1);
    const int32_t tx0 = blockIdx.x * APPLY_TW, ty0 = blockIdx.y * APPLY_TH;
    if (x1 <= 0 || tx0 >= x1 || tx0 + APPLY_TW <= x0 || ty0 >= y1 || ty0 + APPLY_TH <= y0) return;
    const int32_t y = ty0 + (threadIdx.x >> 6);
    if (y >= g.H) return;
    const int32_t xb = tx0 + (threadIdx.x & 63) * 4;
    const size_t row = (size_t)mi * g.cells + (size_t)y * g.W;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int32_t x = xb + i;
        if (x < g.W) {
            const uint32_t c = cnt[row + x];
            if (c) {
                const double nf = (double)(c & 0xffffu), no = (double)(c >> 16);
                logd[row + x] = logd[row + x] + (nf * g.l_free + no * g.l_occ);
                cnt[row + x] = 0u;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Likelihood field.  One workgroup = one LK_TW x LK_TH output tile.
//   phase 1: thresholded cells of the tile + k-halo -> LDS codes (0 -> 0.0, 1 -> 0.5, 2 -> 1.0,
//            255 -> outside the map, tap skipped: Util.java:396,418)
//   phase 2: horizontal sums for the tile's rows + k-halo rows -> LDS doubles
//   phase 3: vertical sums -> likelihoodData
// ---------------------------------------------------------------------------------------------
#define LK_TW 64
#define LK_TH 32

template <int KH>   // KH > 0: compile-time half width; KH == 0: runtime g.khalf
__global__ void __launch_bounds__(256)
k_likelihood(GridDev g, const double *__restrict__ logd, double *__restrict__ lik,
             const double *__restrict__ taps_g, const int32_t *__restrict__ bbox, int32_t dirty_only,
             int32_t tiles_x, int32_t tiles_y) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int32_t k = KH > 0 ? KH : g.khalf;
    const int32_t ntaps = 2 * k + 1;
    const int32_t RW = LK_TW + 2 * k;          // staged columns
    const int32_t RH = LK_TH + 2 * k;          // staged rows
    double *hs = reinterpret_cast<double *>(smem);                       // [RH][LK_TW]
    double *taps = hs + (size_t)RH * LK_TW;                              // [ntaps]
    unsigned char *codes = reinterpret_cast<unsigned char *>(taps + ntaps);   // [RH][RW]

    const int32_t mi = blockIdx.y;
    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each
    // XCD a contiguous band of tiles and let the halo re-reads hit its own L2.
    int32_t bid = blockIdx.x;
    const int32_t ntiles = tiles_x * tiles_y;
    if ((ntiles & 7) == 0) bid = (bid & 7) * (ntiles >> 3) + (bid >> 3);
    const int32_t tx0 = (bid % tiles_x) * LK_TW, ty0 = (bid / tiles_x) * LK_TH;

    if (dirty_only) {
        int32_t x0, y0, x1, y1;
        bbox_decode(bbox + 4 * mi, g.W, g.H, x0, y0, x1, y1);
        if (x1 <= 0) return;
        // outputs that can change: the touched box dilated by k
        if (tx0 >= x1 + k || tx0 + LK_TW <= x0 - k || ty0 >= y1 + k || ty0 + LK_TH <= y0 - k) return;
    }

    const double *mlog = logd + (size_t)mi * g.cells;
    for (int32_t i = threadIdx.x; i < ntaps; i += blockDim.x) taps[i] = taps_g[i];

    // phase 1
    for (int32_t idx = threadIdx.x; idx < RH * RW; idx += blockDim.x) {
        const int32_t r = idx / RW, c = idx - r * RW;
        const int32_t gy = ty0 - k + r, gx = tx0 - k + c;
        unsigned char code = 255;
        if (gx >= 0 && gx < g.W && gy >= 0 && gy < g.H) {
            const double v = mlog[(size_t)gy * g.W + gx];
            code = v > 0.0 ? 2 : (v < 0.0 ? 0 : 1);                       // GridMap.java:239-244
        }
        codes[idx] = code;
    }
    __syncthreads();

    // phase 2 (Util.java:387-404)
    for (int32_t idx = threadIdx.x; idx < RH * LK_TW; idx += blockDim.x) {
        const int32_t r = idx / LK_TW, c = idx - r * LK_TW;
        const int32_t gy = ty0 - k + r;
        double total = 0.0;
        if (gy >= 0 && gy < g.H) {
            const unsigned char *row = codes + r * RW + c;
#pragma unroll
            for (int32_t i = 0; i < (KH > 0 ? 2 * KH + 1 : ntaps); i++) {
                const unsigned char cd = row[i];
                if (cd != 255) total += taps[i] * ((double)cd * 0.5);
            }
        }
        hs[idx] = total;
    }
    __syncthreads();

    // phase 3 (Util.java:410-425)
    double *mlik = lik + (size_t)mi * g.cells;
    for (int32_t idx = threadIdx.x; idx < LK_TH * LK_TW; idx += blockDim.x) {
        const int32_t r = idx / LK_TW, c = idx - r * LK_TW;
        const int32_t gy = ty0 + r, gx = tx0 + c;
        if (gx < g.W && gy < g.H) {
            double total = 0.0;
#pragma unroll
            for (int32_t i = 0; i < (KH > 0 ? 2 * KH + 1 : ntaps); i++) {
                const int32_t y2 = gy - k + i;
                if (y2 >= 0 && y2 < g.H) total += taps[i] * hs[(r + i) * LK_TW + c];
            }
            mlik[(size_t)gy * g.W + gx] = total;
        }
    }
}

__global__ void k_fill(double *__restrict__ d, double v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = v;
}

// getRawAt / getProbAt (GridMap.java:134-140, Util.java:46-48)
__global__ void k_get_raw(GridDev g, const double *__restrict__ logd, int32_t mi, int32_t x, int32_t y, double *out2) {
    const double l = logd[(size_t)mi * g.cells + (size_t)y * g.W + x];
    out2[0] = l;
    out2[1] = (double)1.0f - (double)1.0f / (1.0 + exp(l));
}

// diagnostics: the float-rounded primitives the parity contract leans on
__global__ void k_debug_f32(int32_t op, const float *__restrict__ a, float *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float c, s;
        if (op == 0) out[i] = j_sqrtf(a[i]);
        else { pose_trig(a[i], c, s); out[i] = op == 1 ? c : s; }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void gms_launch_raycast(gms_map *m, const gms_beam *d_beams, int32_t B, int32_t beam_stride, const float *d_poses) {
    ProfScope ps(m, GMS_K_RAYCAST);
    dim3 grid((B + 63) / 64, m->n_maps);
    hipLaunchKernelGGL(k_raycast<false>, grid, dim3(64), 0, m->stream, m->gd, d_beams, B, beam_stride, d_poses,
                       (const RayIn *)nullptr, m->d_cnt, m->d_bbox, (int32_t *)nullptr, (uint8_t *)nullptr, 0,
                       (int32_t *)nullptr);
}

void gms_launch_trace_scan(gms_map *m, const gms_beam *d_beams, int32_t B, const float *d_pose, int32_t *d_cells,
                           uint8_t *d_cls, int32_t cap, int32_t *d_counts) {
    dim3 grid((B + 63) / 64, 1);
    hipLaunchKernelGGL(k_raycast<true>, grid, dim3(64), 0, m->stream, m->gd, d_beams, B, m->max_beams, d_pose,
                       (const RayIn *)nullptr, (uint32_t *)nullptr, (int32_t *)nullptr, d_cells, d_cls, cap, d_counts);
}

void gms_launch_trace_ray(gms_map *m, float x0, float y0, float x1, float y1, int32_t extra, int32_t *d_cells,
                          int32_t cap, int32_t *d_count) {
    hipLaunchKernelGGL(k_trace_ray, dim3(1), dim3(64), 0, m->stream, m->gd.W, m->gd.H, x0, y0, x1, y1, extra, d_cells,
                       cap, d_count);
}

__global__ void k_store_ray(RayIn *dst, RayIn r) { *dst = r; }

void gms_launch_apply_ray(gms_map *m, RayIn ray) {
    // the single ray travels through the beam staging buffer
    RayIn *d_ray = reinterpret_cast<RayIn *>(m->d_beams);
    hipLaunchKernelGGL(k_store_ray, dim3(1), dim3(1), 0, m->stream, d_ray, ray);
    ProfScope ps(m, GMS_K_RAYCAST);
    hipLaunchKernelGGL(k_raycast<false>, dim3(1, 1), dim3(64), 0, m->stream, m->gd, (const gms_beam *)nullptr, 1,
                       m->max_beams, (const float *)nullptr, (const RayIn *)d_ray, m->d_cnt, m->d_bbox,
                       (int32_t *)nullptr, (uint8_t *)nullptr, 0, (int32_t *)nullptr);
}

void gms_launch_apply_counts(gms_map *m) {
    ProfScope ps(m, GMS_K_APPLY);
    dim3 grid((m->gd.W + APPLY_TW - 1) / APPLY_TW, (m->gd.H + APPLY_TH - 1) / APPLY_TH, m->n_maps);
    hipLaunchKernelGGL(k_apply, grid, dim3(256), 0, m->stream, m->gd, m->d_log, m->d_cnt, m->d_bbox);
}

void gms_launch_likelihood(gms_map *m, int32_t dirty_only) {
    ProfScope ps(m, GMS_K_LIKELIHOOD);
    const int32_t k = m->gd.khalf;
    const int32_t tiles_x = (m->gd.W + LK_TW - 1) / LK_TW, tiles_y = (m->gd.H + LK_TH - 1) / LK_TH;
    const size_t RH = LK_TH + 2 * k, RW = LK_TW + 2 * k;
    const size_t smem = RH * LK_TW * sizeof(double) + (2 * k + 1) * sizeof(double) + RH * RW;
    dim3 grid(tiles_x * tiles_y, m->n_maps);
#define LK_LAUNCH(KH)                                                                                         \
    do {                                                                                                      \
        if (smem > 48 * 1024)                                                                                 \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_likelihood<KH>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                       \
        hipLaunchKernelGGL(k_likelihood<KH>, grid, dim3(256), smem, m->stream, m->gd, m->d_log, m->d_lik,     \
                           m->d_taps, m->d_bbox, dirty_only, tiles_x, tiles_y);                               \
    } while (0)
    if (k == 3) LK_LAUNCH(3);
    else if (k == 5) LK_LAUNCH(5);
    else LK_LAUNCH(0);
#undef LK_LAUNCH
}

void gms_launch_fill(gms_map *m, double *d, double v, int64_t n) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, m->stream, d, v, n);
}

void gms_launch_get_raw(gms_map *m, int32_t mi, int32_t x, int32_t y, double *d_out2) {
    hipLaunchKernelGGL(k_get_raw, dim3(1), dim3(1), 0, m->stream, m->gd, m->d_log, mi, x, y, d_out2);
}

void gms_launch_debug_f32(gms_map *m, int32_t op, const float *d_a, float *d_out, int64_t n) {
    hipLaunchKernelGGL(k_debug_f32, dim3(1024), dim3(256), 0, m->stream, op, d_a, d_out, n);
}
