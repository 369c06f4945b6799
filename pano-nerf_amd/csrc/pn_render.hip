// pn_render.hip — HBM-bound stages of the Pano-NeRF hot path (gfx950): ray generation, cone
// sampling, integrated positional encoding, per-ray alpha-composite scans (forward + adjoint),
// hierarchical PDF resampling, normal/albedo gathers, Lambertian shading, tone-mapped loss, Adam.
//
// Scan kernels give each ray to a group of G lanes of one wavefront (G = 16, 32 or 64, chosen from
// N so that a 10-sample light ray does not waste a 64-wide wave); every lane owns a run of
// consecutive samples, does its run serially in registers and the group combines the runs with
// __shfl_up / __shfl_xor — no LDS, no atomics, one pass over HBM.
#include "pn_common.h"
#include <math.h>

#define HALF_PI_F 1.5707964f /* fl32(0.5 * fl32(pi)), models/mip.py:428,437 */

// ------------------------------------------------------------------------------- helpers
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float softplus_d1(float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus_d2(float x) {
    if (x > 20.f) return 0.f;
    float s = 1.f / (1.f + expf(-x));
    return s * (1.f - s);
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// torch.linspace(a, b, steps)[i] (symmetric evaluation used by ATen)
__device__ __forceinline__ float linspace_at(float a, float b, int steps, int i) {
    float step = (b - a) / (float)(steps - 1);
    return (i < steps / 2) ? a + step * (float)i : b - step * (float)(steps - 1 - i);
}

struct Gauss {
    float mean[3], cov[3];
};
// conical frustum [t0,t1] -> diagonal Gaussian; models/mip.py:51-58 and 10-22
__device__ __forceinline__ Gauss cast_cone(float t0, float t1, const float o[3], const float d[3], float radius) {
    float mu = (t0 + t1) / 2.f, hw = (t1 - t0) / 2.f;
    float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
    float den = 3.f * mu2 + hw2;
    float t_mean = mu + (2.f * mu * hw2) / den;
    float t_var = hw2 / 3.f - (4.f / 15.f) * ((hw4 * (12.f * mu2 - hw2)) / (den * den));
    float r_var = radius * radius * (mu2 / 4.f + (5.f / 12.f) * hw2 - (4.f / 15.f) * hw4 / den);
    float dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + 1e-10f;
    Gauss g;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float d2 = d[c] * d[c];
        g.mean[c] = d[c] * t_mean + o[c];
        g.cov[c] = t_var * d2 + r_var * (1.f - d2 / dd);
    }
    return g;
}

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}
// exclusive prefix sum over the G lanes of a group
template <int G>
__device__ __forceinline__ float group_excl_scan(float v, int gl) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
        float n = __shfl_up(inc, o, G);
        if (gl >= o) inc += n;
    }
    return inc - v;
}

// --------------------------------------------------------------------------- ray generation
// One pixel of an equirectangular camera (datasets/pano_datasets.py:157-213): shared by the pool generator and by the
// batch sampler that regenerates rays from (camera, pixel), so that both give the same bits.
struct PanoCam {
    float r00, r01, r02, r10, r11, r12, r20, r21, r22, tx, ty, tz;
};
struct PanoRay {
    float d[3], nrm, radius, noise_var;
};
__device__ __forceinline__ PanoRay pano_ray(int H, int W, const PanoCam& c, int i, int j) {
    const float PI_F = 3.14159265358979323846f;
    auto cam_dir = [&](int ii, int jj, float out[3]) {
        float theta = -((float)jj + 0.5f) / (float)W * 2.f * PI_F;
        float phi = ((float)ii + 0.5f) / (float)H * PI_F;
        float sp = sinf(phi);
        float x = sp * sinf(theta), y = cosf(phi), z = sp * cosf(theta);
        out[0] = x * c.r00 + y * c.r01 + z * c.r02;  // camera_dirs @ c2w[:3,:3].T
        out[1] = x * c.r10 + y * c.r11 + z * c.r12;
        out[2] = x * c.r20 + y * c.r21 + z * c.r22;
    };
    PanoRay r;
    cam_dir(i, j, r.d);
    r.nrm = sqrtf(r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2]);
    // constant pixel radius: |dir(H/2, jj) - dir(H/2, jj+1)| * 2 / sqrt(12); column W-1 repeats column W-3
    int jj = (j < W - 1) ? j : W - 3;
    if (jj < 0) jj = 0;
    float a[3], b[3];
    cam_dir(H / 2, jj, a);
    cam_dir(H / 2, jj + 1 < W ? jj + 1 : jj, b);
    float dx = sqrtf((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
    float phi = ((float)i + 0.5f) / (float)H * PI_F;
    r.radius = (float)((double)dx * 2.0 / sqrt(12.0));
    r.noise_var = sinf(phi) * PI_F / (float)W;
    return r;
}
__device__ __forceinline__ void store_pano_ray(int64_t o, const PanoRay& r, const PanoCam& c, float near_, float far_,
                                               float* origins, float* directions, float* viewdirs, float* radii,
                                               float* lossmult, float* near_out, float* far_out, float* noise_var) {
    origins[o * 3 + 0] = c.tx;
    origins[o * 3 + 1] = c.ty;
    origins[o * 3 + 2] = c.tz;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        directions[o * 3 + k] = r.d[k];
        viewdirs[o * 3 + k] = r.d[k] / r.nrm;
    }
    radii[o] = r.radius;
    lossmult[o] = 1.f;
    near_out[o] = near_;
    far_out[o] = far_;
    noise_var[o] = r.noise_var;
}

__global__ void k_raygen_pano(int H, int W, PanoCam c, float near_, float far_,
                              float* origins, float* directions, float* viewdirs, float* radii, float* lossmult,
                              float* near_out, float* far_out, float* noise_var) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const PanoRay r = pano_ray(H, W, c, idx / W, idx % W);
    store_pano_ray(idx, r, c, near_, far_, origins, directions, viewdirs, radii, lossmult, near_out, far_out, noise_var);
}

// Training-batch sampler that REGENERATES the rays: batch ray b is pixel idx[b] % (H W) of camera idx[b] / (H W); only the
// 12-byte target colour is read from a stored pool (no 56-byte-per-ray pool in HBM, no pool reads).
__global__ void k_sample_pano_rays(int64_t B, int n_cam, int H, int W, const int64_t* idx, const float* c2ws, float near_,
                                   float far_, const float* rgb_pool, float* origins, float* directions, float* viewdirs,
                                   float* radii, float* lossmult, float* near_out, float* far_out, float* noise_var,
                                   float* rgb_out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t hw = (int64_t)H * W;
    int64_t r = idx[b];
    r = (r >= 0 && r < hw * n_cam) ? r : 0;
    const int cam = (int)(r / hw), pix = (int)(r % hw);
    const float* m = c2ws + 16 * cam;
    const PanoCam c{m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10], m[3], m[7], m[11]};
    const PanoRay ray = pano_ray(H, W, c, pix / W, pix % W);
    store_pano_ray(b, ray, c, near_, far_, origins, directions, viewdirs, radii, lossmult, near_out, far_out, noise_var);
    if (rgb_pool) {
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb_out[b * 3 + k] = rgb_pool[r * 3 + k];
    }
}

__device__ __forceinline__ uint16_t f64_to_half_bits(double x) {
    _Float16 h = (_Float16)x;  // round-to-nearest-even, single rounding from fp64
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

__global__ void k_lit_rays(int D, double radius, double near_, double far_, uint16_t* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    const double PI_D = 3.14159265358979323846;
    double ga = PI_D * (3.0 - sqrt(5.0));
    double y = 1.0 - ((double)i / (double)(D - 1)) * 2.0;
    double r = sqrt(1.0 - y * y);
    double th = ga * (double)i;
    double d[3] = {cos(th) * r, y, sin(th) * r};
    double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    uint16_t* o_ = out;
    uint16_t* d_ = out + 3 * D;
    uint16_t* v_ = out + 6 * D;
    uint16_t* s_ = out + 9 * D;
    for (int c = 0; c < 3; ++c) {
        o_[i * 3 + c] = f64_to_half_bits(0.0);
        d_[i * 3 + c] = f64_to_half_bits(d[c]);
        v_[i * 3 + c] = f64_to_half_bits(d[c] / n);
    }
    s_[0 * D + i] = f64_to_half_bits(radius);
    s_[1 * D + i] = f64_to_half_bits(4.0 * PI_D / (double)D);
    s_[2 * D + i] = f64_to_half_bits(near_);
    s_[3 * D + i] = f64_to_half_bits(far_);
    s_[4 * D + i] = f64_to_half_bits(0.0);
}

// --------------------------------------------------------------------------- coarse sampling
// the s-th of S sample positions between near and far: linear in depth, or - disparity - linear in inverse depth
// (models/mip.py:134-138: 1 / (1 / near * (1 - x) + 1 / far * x), in that order of operations)
__device__ __forceinline__ float base_t(float nr, float fr, int S, int s, bool disparity) {
    const float x = linspace_at(0.f, 1.f, S, s);
    if (disparity) return 1.f / (1.f / nr * (1.f - x) + 1.f / fr * x);
    return nr + (fr - nr) * x;
}
__device__ __forceinline__ float coarse_t(float nr, float fr, int S, int s, const float* rnd_row, bool disparity = false) {
    float t = base_t(nr, fr, S, s, disparity);
    if (rnd_row) {
        float tm1 = (s > 0) ? base_t(nr, fr, S, s - 1, disparity) : t;
        float tp1 = (s < S - 1) ? base_t(nr, fr, S, s + 1, disparity) : t;
        float lower = (s > 0) ? 0.5f * (t + tm1) : t;
        float upper = (s < S - 1) ? 0.5f * (tp1 + t) : t;
        t = lower + (upper - lower) * rnd_row[s];
    }
    return t;
}

__global__ void k_sample_coarse(int64_t B, int N, int disparity, const float* origins, const float* directions, const float* radii,
                                const float* near_, const float* far_, const float* t_rand, float* t_out, float* mean,
                                float* cov) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N) return;
    int64_t b = idx / N;
    int n = (int)(idx % N);
    const int S = N + 1;
    float nr = near_[b], fr = far_[b];
    const float* rr = t_rand ? t_rand + b * S : nullptr;
    float t0 = coarse_t(nr, fr, S, n, rr, disparity != 0), t1 = coarse_t(nr, fr, S, n + 1, rr, disparity != 0);
    t_out[b * S + n] = t0;
    if (n == N - 1) t_out[b * S + N] = t1;
    float o[3] = {origins[b * 3], origins[b * 3 + 1], origins[b * 3 + 2]};
    float d[3] = {directions[b * 3], directions[b * 3 + 1], directions[b * 3 + 2]};
    Gauss g = cast_cone(t0, t1, o, d, radii[b]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mean[idx * 3 + c] = g.mean[c];
        cov[idx * 3 + c] = g.cov[c];
    }
}

__global__ void k_sample_env(int64_t B, int D, int Ne, const float* origins, const float* directions,
                             const float* distance, const float* env_dirs, const float* env_radii,
                             const float* env_near, const float* env_far, const float* env_rand, float* t_out,
                             float* mean, float* cov) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D * Ne) return;
    int k = (int)(idx % Ne);
    int64_t ray = idx / Ne;
    int j = (int)(ray % D);
    int64_t b = ray / D;
    const int S = Ne + 1;
    float t0 = coarse_t(env_near[j], env_far[j], S, k, env_rand), t1 = coarse_t(env_near[j], env_far[j], S, k + 1, env_rand);
    t_out[ray * S + k] = t0;
    if (k == Ne - 1) t_out[ray * S + Ne] = t1;
    float dist = distance[b];
    float o[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = origins[b * 3 + c] + directions[b * 3 + c] * dist;
        d[c] = env_dirs[j * 3 + c];
    }
    Gauss g = cast_cone(t0, t1, o, d, env_radii[j]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mean[idx * 3 + c] = g.mean[c];
        cov[idx * 3 + c] = g.cov[c];
    }
}

// ------------------------------------------------------------------- hierarchical resampling
// One wavefront per ray; cdf / bins / new t live in LDS (3 * 516 floats per wave).
#define RS_MAXS (PN_MAX_SAMPLES + 4)
__global__ __launch_bounds__(256) void k_resample(int64_t B, int N, const float* t_in, const float* weights,
                                                   float padding, const float* u_rand, const float* origins,
                                                   const float* directions, const float* radii, float* t_out,
                                                   float* mean, float* cov) {
    __shared__ float sh[4][3][RS_MAXS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * 4 + wv;
    if (b >= B) return;  // whole wave exits together; no block-level barrier is used below
    float* cdf = sh[wv][0];
    float* bins = sh[wv][1];
    float* tnew = sh[wv][2];
    const int S = N + 1;
    const float* w = weights + b * N;
    // blur-pool + padding; each lane owns a run of consecutive bins so the CDF scan is run-local + wave scan
    const int run = (N + 63) / 64;
    const int n0 = lane * run;
    float wl[(PN_MAX_SAMPLES + 63) / 64];
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < (PN_MAX_SAMPLES + 63) / 64; ++i) {
        int n = n0 + i;
        float v = 0.f;
        if (i < run && n < N) {
            float wm1 = w[n > 0 ? n - 1 : 0], wc = w[n], wp1 = w[n < N - 1 ? n + 1 : N - 1];
            v = 0.5f * (fmaxf(wm1, wc) + fmaxf(wc, wp1)) + padding;
        }
        wl[i] = v;
        lsum += v;
    }
    float wsum = group_sum<64>(lsum);
    float pad = fmaxf(0.f, 1e-5f - wsum);
    float padn = pad / (float)N;
    wsum += pad;
    lsum = 0.f;
#pragma unroll
    for (int i = 0; i < (PN_MAX_SAMPLES + 63) / 64; ++i) {
        int n = n0 + i;
        float p = 0.f;
        if (i < run && n < N) p = (wl[i] + padn) / wsum;
        wl[i] = p;
        lsum += p;
    }
    float pre = group_excl_scan<64>(lsum, lane);
#pragma unroll
    for (int i = 0; i < (PN_MAX_SAMPLES + 63) / 64; ++i) {
        int n = n0 + i;
        if (i < run && n < N) {
            pre += wl[i];                              // inclusive sum of pdf[0..n]
            if (n < N - 1) cdf[n + 1] = fminf(1.f, pre);  // cdf[k] = min(1, sum_{m<k} pdf_m), k = 1..N-1
        }
    }
    if (lane == 0) {
        cdf[0] = 0.f;
        cdf[N] = 1.f;
    }
    for (int s = lane; s < S; s += 64) bins[s] = t_in[b * S + s];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const float EPS = 1.1920929e-07f;
    for (int s = lane; s < S; s += 64) {
        float u;
        if (u_rand) {
            u = (float)s * (float)(1.0 / (double)S) + u_rand[b * S + s];
            u = fminf(u, 1.f - EPS);
        } else {
            u = linspace_at(0.f, 1.f - EPS, S, s);
        }
        // searchsorted(cdf, u, right=True): first index with cdf[idx] > u, in [0, S]
        int lo = 0, hi = S;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (cdf[mid] > u) hi = mid; else lo = mid + 1;
        }
        int below = lo - 1 < 0 ? 0 : lo - 1;
        int above = lo > S - 1 ? S - 1 : lo;
        float c0 = cdf[below], c1 = cdf[above], b0 = bins[below], b1 = bins[above];
        float den = c1 - c0;
        if (den < 1e-5f) den = 1.f;
        float tt = (u - c0) / den;
        float v = b0 + tt * (b1 - b0);
        tnew[s] = v;
        t_out[b * S + s] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    float o[3] = {origins[b * 3], origins[b * 3 + 1], origins[b * 3 + 2]};
    float d[3] = {directions[b * 3], directions[b * 3 + 1], directions[b * 3 + 2]};
    float rad = radii[b];
    for (int n = lane; n < N; n += 64) {
        Gauss g = cast_cone(tnew[n], tnew[n + 1], o, d, rad);
        int64_t row = b * N + n;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mean[row * 3 + c] = g.mean[c];
            cov[row * 3 + c] = g.cov[c];
        }
    }
}

// ------------------------------------------------------------------------------- encodings
__global__ void k_ipe_encode(int64_t M, const float* mean, const float* cov, float* enc) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * 48) return;
    int64_t row = idx / 48;
    int f = (int)(idx % 48), l = f / 3, c = f % 3;
    float sc = (float)(1 << l);
    float y = mean[row * 3 + c] * sc;
    float v = cov[row * 3 + c] * (sc * sc);
    float e = expf(-0.5f * v);
    enc[row * PN_ENC_DIM + f] = e * sinf(y);
    enc[row * PN_ENC_DIM + 48 + f] = e * sinf(y + HALF_PI_F);
}

// d enc -> d mean: d/dmean_c [e * sin(2^l mean_c (+pi/2))] = 2^l * e * cos(.)
__global__ void k_ipe_backward(int64_t M, const float* mean, const float* cov, const float* d_enc, float* d_mean) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * 3) return;
    int64_t row = idx / 3;
    int c = (int)(idx % 3);
    float m = mean[idx], cv = cov[idx];
    float acc = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) {
        float sc = (float)(1 << l);
        float y = m * sc, v = cv * (sc * sc);
        float e = expf(-0.5f * v) * sc;
        acc += d_enc[row * PN_ENC_DIM + l * 3 + c] * e * cosf(y);
        acc += d_enc[row * PN_ENC_DIM + 48 + l * 3 + c] * e * cosf(y + HALF_PI_F);
    }
    d_mean[idx] = acc;
}

// tangent of the encoding along v (forward-mode): edot[f] = d enc[f]/d mean_c * v_c
__global__ void k_ipe_tangent(int64_t M, const float* mean, const float* cov, const float* v_in, float* edot) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * 48) return;
    int64_t row = idx / 48;
    int f = (int)(idx % 48), l = f / 3, c = f % 3;
    float sc = (float)(1 << l);
    float y = mean[row * 3 + c] * sc;
    float v = cov[row * 3 + c] * (sc * sc);
    float e = expf(-0.5f * v) * sc * v_in[row * 3 + c];
    edot[row * PN_ENC_DIM + f] = e * cosf(y);
    edot[row * PN_ENC_DIM + 48 + f] = e * cosf(y + HALF_PI_F);
}

__global__ void k_pos_enc_view(int64_t R, const float* viewdirs, float* viewenc) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * PN_VIEW_DIM) return;
    int64_t r = idx / PN_VIEW_DIM;
    int i = (int)(idx % PN_VIEW_DIM);
    float out;
    if (i < 3) {
        out = viewdirs[r * 3 + i];
    } else {
        int f = (i - 3) % 12, half = (i - 3) / 12, l = f / 3, c = f % 3;
        float xb = viewdirs[r * 3 + c] * (float)(1 << l);
        out = sinf(half ? xb + HALF_PI_F : xb);
    }
    viewenc[idx] = out;
}

// ------------------------------------------------------------------ alpha compositing (scan)
#define CP_MAXRUN 8  // samples per lane: N <= 64 * 8
struct CompArgs {
    int64_t R;
    int N, nc;
    float density_bias, rgb_padding;
    int white_bkgd;
    const float *raw_rgb, *raw_density, *t, *dirs;
    int64_t dir_mod;
};

template <int G>
__global__ __launch_bounds__(256) void k_composite_fwd(CompArgs a, float* comp_rgb, float* distance, float* acc_out,
                                                        float* weights) {
    const int gl = threadIdx.x % G;
    const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool live = ray < a.R;
    const int64_t r = live ? ray : a.R - 1;  // dead groups shadow the last ray and skip the stores
    const int N = a.N, S = N + 1;
    const int run = (N + G - 1) / G;
    const int n0 = gl * run;
    const float* dd = a.dirs + (r % a.dir_mod) * 3;
    const float dn = sqrtf(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
    const float* tr = a.t + r * S;
    float x[CP_MAXRUN];
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < CP_MAXRUN; ++i) {
        int n = n0 + i;
        float xv = 0.f;
        if (i < run && n < N) {
            float sigma = softplus_f(a.raw_density[(r * N + n) * a.nc] + a.density_bias);
            xv = sigma * ((tr[n + 1] - tr[n]) * dn);
        }
        x[i] = xv;
        lsum += xv;
    }
    float pre = group_excl_scan<G>(lsum, gl);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, ac = 0.f, wt = 0.f;
#pragma unroll
    for (int i = 0; i < CP_MAXRUN; ++i) {
        int n = n0 + i;
        if (i < run && n < N) {
            float w = (1.f - expf(-x[i])) * expf(-pre);
            pre += x[i];
            const float* rr = a.raw_rgb + (r * N + n) * 3;
            float s = 1.f + 2.f * a.rgb_padding;
            c0 += w * (softplus_f(rr[0]) * s - a.rgb_padding);
            c1 += w * (softplus_f(rr[1]) * s - a.rgb_padding);
            c2 += w * (softplus_f(rr[2]) * s - a.rgb_padding);
            ac += w;
            wt += w * (0.5f * (tr[n] + tr[n + 1]));
            if (live) weights[r * N + n] = w;
        }
    }
    c0 = group_sum<G>(c0);
    c1 = group_sum<G>(c1);
    c2 = group_sum<G>(c2);
    ac = group_sum<G>(ac);
    wt = group_sum<G>(wt);
    if (live && gl == 0) {
        float dist = wt / ac;
        if (isnan(dist)) dist = 0.f;
        else if (isinf(dist)) dist = dist > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
        dist = fminf(fmaxf(dist, tr[0]), tr[N]);
        if (a.white_bkgd) {
            c0 += 1.f - ac;
            c1 += 1.f - ac;
            c2 += 1.f - ac;
        }
        comp_rgb[r * 3 + 0] = c0;
        comp_rgb[r * 3 + 1] = c1;
        comp_rgb[r * 3 + 2] = c2;
        distance[r] = dist;
        acc_out[r] = ac;
    }
}

template <int G>
__global__ __launch_bounds__(256) void k_composite_bwd(CompArgs a, const float* d_comp, const float* d_dist,
                                                        const float* d_w_ext, float* d_raw_rgb, float* d_raw_density) {
    const int gl = threadIdx.x % G;
    const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool live = ray < a.R;
    const int64_t r = live ? ray : a.R - 1;
    const int N = a.N, S = N + 1;
    const int run = (N + G - 1) / G;
    const int n0 = gl * run;
    const float* dd = a.dirs + (r % a.dir_mod) * 3;
    const float dn = sqrtf(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
    const float* tr = a.t + r * S;
    const float pscale = 1.f + 2.f * a.rgb_padding;
    float x[CP_MAXRUN], w[CP_MAXRUN], tb[CP_MAXRUN];  // tb = T_n * exp(-x_n) = T_{n+1}
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < CP_MAXRUN; ++i) {
        int n = n0 + i;
        float xv = 0.f;
        if (i < run && n < N) {
            float sigma = softplus_f(a.raw_density[(r * N + n) * a.nc] + a.density_bias);
            xv = sigma * ((tr[n + 1] - tr[n]) * dn);
        }
        x[i] = xv;
        lsum += xv;
    }
    float pre = group_excl_scan<G>(lsum, gl);
    float ac = 0.f, wt = 0.f;
#pragma unroll
    for (int i = 0; i < CP_MAXRUN; ++i) {
        int n = n0 + i;
        w[i] = 0.f;
        tb[i] = 0.f;
        if (i < run && n < N) {
            float T = expf(-pre), ex = expf(-x[i]);
            w[i] = (1.f - ex) * T;
            tb[i] = T * ex;
            pre += x[i];
            ac += w[i];
            wt += w[i] * (0.5f * (tr[n] + tr[n + 1]));
        }
    }
    ac = group_sum<G>(ac);
    wt = group_sum<G>(wt);
    const float g0 = d_comp[r * 3], g1 = d_comp[r * 3 + 1], g2 = d_comp[r * 3 + 2];
    float d_acc = a.white_bkgd ? -(g0 + g1 + g2) : 0.f;
    float d_wt = 0.f;
    if (d_dist) {
        float q = wt / ac;
        bool pass = isfinite(q) && q >= tr[0] && q <= tr[N];
        if (pass) {
            float gd = d_dist[r];
            d_wt = gd / ac;
            d_acc += -gd * q / ac;
        }
    }
    // total dL/dw_n, and the running sum of dw*w needed by the transmittance adjoint
    float dw[CP_MAXRUN];
    float lsum2 = 0.f;
#pragma unroll
    for (int i = 0; i < CP_MAXRUN; ++i) {
        int n = n0 + i;
        dw[i] = 0.f;
        if (i < run && n < N) {
            const float* rr = a.raw_rgb + (r * N + n) * 3;
            float rgb0 = softplus_f(rr[0]) * pscale - a.rgb_padding;
            float rgb1 = softplus_f(rr[1]) * pscale - a.rgb_padding;
            float rgb2 = softplus_f(rr[2]) * pscale - a.rgb_padding;
            float v = g0 * rgb0 + g1 * rgb1 + g2 * rgb2 + d_acc + d_wt * (0.5f * (tr[n] + tr[n + 1]));
            if (d_w_ext) v += d_w_ext[r * N + n];
            dw[i] = v;
            lsum2 += v * w[i];
            if (live) {
                float* o = d_raw_rgb + (r * N + n) * 3;
                o[0] = w[i] * g0 * pscale * softplus_d1(rr[0]);
                o[1] = w[i] * g1 * pscale * softplus_d1(rr[1]);
                o[2] = w[i] * g2 * pscale * softplus_d1(rr[2]);
            }
        }
    }
    float total = group_sum<G>(lsum2);
    float incl = group_excl_scan<G>(lsum2, gl);  // sum over lanes before this one
#pragma unroll
    for (int i = 0; i < CP_MAXRUN; ++i) {
        int n = n0 + i;
        if (i < run && n < N) {
            incl += dw[i] * w[i];  // inclusive through sample n
            float dx = dw[i] * tb[i] - (total - incl);
            float z = a.raw_density[(r * N + n) * a.nc] + a.density_bias;
            if (live) d_raw_density[(r * N + n) * a.nc] = dx * ((tr[n + 1] - tr[n]) * dn) * softplus_d1(z);
        }
    }
}

// ------------------------------------------------------------- normals / albedo gather (a12)
struct GatherArgs {
    int64_t B;
    int N, nc;
    const float *grad_mean, *weights, *raw_density, *directions;
};
__device__ __forceinline__ void unit_neg(const float* g, float n[3], float& len) {
    float x = -g[0], y = -g[1], z = -g[2];
    len = sqrtf(x * x + y * y + z * z);
    float d = fmaxf(len, 1e-12f);
    n[0] = x / d;
    n[1] = y / d;
    n[2] = z / d;
}

template <int G>
__global__ __launch_bounds__(256) void k_gather_fwd(GatherArgs a, float* normal, float* ort_ray, float* albedo) {
    const int gl = threadIdx.x % G;
    const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool live = ray < a.B;
    const int64_t b = live ? ray : a.B - 1;
    const int N = a.N;
    const float d0 = a.directions[b * 3], d1 = a.directions[b * 3 + 1], d2 = a.directions[b * 3 + 2];
    float ws = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, so = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int n = gl; n < N; n += G) {
        int64_t row = b * N + n;
        float w = a.weights[row];
        float nn[3], len;
        unit_neg(a.grad_mean + row * 3, nn, len);
        float dot = fmaxf(nn[0] * d0 + nn[1] * d1 + nn[2] * d2, 0.f);
        ws += w;
        s0 += w * nn[0];
        s1 += w * nn[1];
        s2 += w * nn[2];
        so += w * dot * dot;
        if (a.nc >= 5) {
            const float* rd = a.raw_density + row * a.nc;
            a0 += w * (sigmoid_f(rd[1]) * 0.77f + 0.03f);
            a1 += w * (sigmoid_f(rd[2]) * 0.77f + 0.03f);
            a2 += w * (sigmoid_f(rd[3]) * 0.77f + 0.03f);
        }
    }
    ws = group_sum<G>(ws);
    s0 = group_sum<G>(s0);
    s1 = group_sum<G>(s1);
    s2 = group_sum<G>(s2);
    so = group_sum<G>(so);
    a0 = group_sum<G>(a0);
    a1 = group_sum<G>(a1);
    a2 = group_sum<G>(a2);
    if (live && gl == 0) {
        s0 /= ws;
        s1 /= ws;
        s2 /= ws;
        float len = fmaxf(sqrtf(s0 * s0 + s1 * s1 + s2 * s2), 1e-12f);
        normal[b * 3] = s0 / len;
        normal[b * 3 + 1] = s1 / len;
        normal[b * 3 + 2] = s2 / len;
        if (ort_ray) ort_ray[b] = so / ws;
        if (albedo && a.nc >= 5) {
            albedo[b * 3] = a0 / ws;
            albedo[b * 3 + 1] = a1 / ws;
            albedo[b * 3 + 2] = a2 / ws;
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void k_gather_bwd(GatherArgs a, const float* d_normal, const float* d_ort_ray,
                                                     const float* d_albedo, float* d_weights, float* v_gradmean,
                                                     float* d_raw_density) {
    const int gl = threadIdx.x % G;
    const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool live = ray < a.B;
    const int64_t b = live ? ray : a.B - 1;
    const int N = a.N;
    const float d0 = a.directions[b * 3], d1 = a.directions[b * 3 + 1], d2 = a.directions[b * 3 + 2];
    // pass 1: W = sum w, Nr = sum w^ n
    float ws = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int n = gl; n < N; n += G) {
        int64_t row = b * N + n;
        float w = a.weights[row];
        float nn[3], len;
        unit_neg(a.grad_mean + row * 3, nn, len);
        ws += w;
        s0 += w * nn[0];
        s1 += w * nn[1];
        s2 += w * nn[2];
    }
    ws = group_sum<G>(ws);
    s0 = group_sum<G>(s0) / ws;
    s1 = group_sum<G>(s1) / ws;
    s2 = group_sum<G>(s2) / ws;
    float nlen = sqrtf(s0 * s0 + s1 * s1 + s2 * s2);
    float gn0 = d_normal[b * 3], gn1 = d_normal[b * 3 + 1], gn2 = d_normal[b * 3 + 2];
    float q0, q1, q2;  // dL/dNr
    if (nlen > 1e-12f) {
        float u0 = s0 / nlen, u1 = s1 / nlen, u2 = s2 / nlen;
        float dp = u0 * gn0 + u1 * gn1 + u2 * gn2;
        q0 = (gn0 - u0 * dp) / nlen;
        q1 = (gn1 - u1 * dp) / nlen;
        q2 = (gn2 - u2 * dp) / nlen;
    } else {
        q0 = gn0 / 1e-12f;
        q1 = gn1 / 1e-12f;
        q2 = gn2 / 1e-12f;
    }
    const float go = d_ort_ray ? d_ort_ray[b] : 0.f;
    float ga0 = 0.f, ga1 = 0.f, ga2 = 0.f;
    if (d_albedo && a.nc >= 5) {
        ga0 = d_albedo[b * 3];
        ga1 = d_albedo[b * 3 + 1];
        ga2 = d_albedo[b * 3 + 2];
    }
    // pass 2: sum_m dwhat_m * what_m
    float cross = 0.f;
    for (int n = gl; n < N; n += G) {
        int64_t row = b * N + n;
        float wh = a.weights[row] / ws;
        float nn[3], len;
        unit_neg(a.grad_mean + row * 3, nn, len);
        float dot = fmaxf(nn[0] * d0 + nn[1] * d1 + nn[2] * d2, 0.f);
        float dwh = q0 * nn[0] + q1 * nn[1] + q2 * nn[2] + go * dot * dot;
        if (a.nc >= 5) {
            const float* rd = a.raw_density + row * a.nc;
            dwh += ga0 * (sigmoid_f(rd[1]) * 0.77f + 0.03f) + ga1 * (sigmoid_f(rd[2]) * 0.77f + 0.03f) +
                   ga2 * (sigmoid_f(rd[3]) * 0.77f + 0.03f);
        }
        cross += dwh * wh;
    }
    cross = group_sum<G>(cross);
    if (!live) return;
    for (int n = gl; n < N; n += G) {
        int64_t row = b * N + n;
        float wh = a.weights[row] / ws;
        float nn[3], len;
        unit_neg(a.grad_mean + row * 3, nn, len);
        float dotr = nn[0] * d0 + nn[1] * d1 + nn[2] * d2;
        float dot = fmaxf(dotr, 0.f);
        float dwh = q0 * nn[0] + q1 * nn[1] + q2 * nn[2] + go * dot * dot;
        if (a.nc >= 5) {
            const float* rd = a.raw_density + row * a.nc;
            float sg1 = sigmoid_f(rd[1]), sg2 = sigmoid_f(rd[2]), sg3 = sigmoid_f(rd[3]);
            dwh += ga0 * (sg1 * 0.77f + 0.03f) + ga1 * (sg2 * 0.77f + 0.03f) + ga2 * (sg3 * 0.77f + 0.03f);
            float* o = d_raw_density + row * a.nc;
            o[1] = wh * ga0 * 0.77f * sg1 * (1.f - sg1);
            o[2] = wh * ga1 * 0.77f * sg2 * (1.f - sg2);
            o[3] = wh * ga2 * 0.77f * sg3 * (1.f - sg3);
            o[4] = 0.f;
        }
        d_weights[row] = (dwh - cross) / ws;
        // dL/dn_s, then through n = u/|u|, u = -g
        float k = go * wh * 2.f * dot;
        float e0 = wh * q0 + k * d0, e1 = wh * q1 + k * d1, e2 = wh * q2 + k * d2;
        float dl = fmaxf(len, 1e-12f);
        float v0, v1, v2;
        if (len > 1e-12f) {
            float dp = nn[0] * e0 + nn[1] * e1 + nn[2] * e2;
            v0 = (e0 - nn[0] * dp) / dl;
            v1 = (e1 - nn[1] * dp) / dl;
            v2 = (e2 - nn[2] * dp) / dl;
        } else {
            v0 = e0 / dl;
            v1 = e1 / dl;
            v2 = e2 / dl;
        }
        v_gradmean[row * 3] = -v0;
        v_gradmean[row * 3 + 1] = -v1;
        v_gradmean[row * 3 + 2] = -v2;
    }
}

// ----------------------------------------------------------------- Lambertian surface (a14)
__global__ void k_surface_fwd(int64_t B, int D, const float* env_rgb, const float* albedo, const float* normal,
                              const float* env_dirs, const float* solid_angle, float* diffuse, float* shading) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float n0 = normal[b * 3], n1 = normal[b * 3 + 1], n2 = normal[b * 3 + 2];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < D; ++j) {
        float nol = fmaxf(n0 * env_dirs[j * 3] + n1 * env_dirs[j * 3 + 1] + n2 * env_dirs[j * 3 + 2], 0.f);
        float k = nol * solid_angle[j];
        const float* e = env_rgb + (b * D + j) * 3;
        s0 += e[0] * k;
        s1 += e[1] * k;
        s2 += e[2] * k;
    }
    const float INV_PI = (float)(1.0 / 3.14159265358979323846);
    shading[b * 3] = s0;
    shading[b * 3 + 1] = s1;
    shading[b * 3 + 2] = s2;
    diffuse[b * 3] = albedo[b * 3] * INV_PI * s0;
    diffuse[b * 3 + 1] = albedo[b * 3 + 1] * INV_PI * s1;
    diffuse[b * 3 + 2] = albedo[b * 3 + 2] * INV_PI * s2;
}

__global__ void k_surface_bwd(int64_t B, int D, const float* env_rgb, const float* albedo, const float* normal,
                              const float* env_dirs, const float* solid_angle, const float* d_diffuse,
                              const float* d_shading, float* d_env_rgb, float* d_albedo, float* d_normal) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float INV_PI = (float)(1.0 / 3.14159265358979323846);
    float n0 = normal[b * 3], n1 = normal[b * 3 + 1], n2 = normal[b * 3 + 2];
    float s[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < D; ++j) {
        float nol = fmaxf(n0 * env_dirs[j * 3] + n1 * env_dirs[j * 3 + 1] + n2 * env_dirs[j * 3 + 2], 0.f);
        float k = nol * solid_angle[j];
        const float* e = env_rgb + (b * D + j) * 3;
        s[0] += e[0] * k;
        s[1] += e[1] * k;
        s[2] += e[2] * k;
    }
    float gs[3], gn[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float gd = d_diffuse ? d_diffuse[b * 3 + c] : 0.f;
        d_albedo[b * 3 + c] = gd * INV_PI * s[c];
        gs[c] = gd * albedo[b * 3 + c] * INV_PI + (d_shading ? d_shading[b * 3 + c] : 0.f);
    }
    for (int j = 0; j < D; ++j) {
        float l0 = env_dirs[j * 3], l1 = env_dirs[j * 3 + 1], l2 = env_dirs[j * 3 + 2];
        float raw = n0 * l0 + n1 * l1 + n2 * l2;
        float nol = fmaxf(raw, 0.f);
        float om = solid_angle[j];
        const float* e = env_rgb + (b * D + j) * 3;
        float* o = d_env_rgb + (b * D + j) * 3;
        o[0] = gs[0] * nol * om;
        o[1] = gs[1] * nol * om;
        o[2] = gs[2] * nol * om;
        if (raw > 0.f) {
            float dn = (gs[0] * e[0] + gs[1] * e[1] + gs[2] * e[2]) * om;
            gn[0] += dn * l0;
            gn[1] += dn * l1;
            gn[2] += dn * l2;
        }
    }
    d_normal[b * 3] = gn[0];
    d_normal[b * 3 + 1] = gn[1];
    d_normal[b * 3 + 2] = gn[2];
}

// d x_surf -> d distance  (x_surf = o + d * distance; every light sample of ray b shares x_surf)
__global__ __launch_bounds__(256) void k_env_origin_bwd(int64_t B, int rows_per_ray, const float* d_mean,
                                                         const float* directions, float* d_distance) {
    const int lane = threadIdx.x & 63;
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= B) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < rows_per_ray; i += 64) {
        const float* g = d_mean + (b * rows_per_ray + i) * 3;
        s0 += g[0];
        s1 += g[1];
        s2 += g[2];
    }
    s0 = group_sum<64>(s0);
    s1 = group_sum<64>(s1);
    s2 = group_sum<64>(s2);
    if (lane == 0)
        d_distance[b] += s0 * directions[b * 3] + s1 * directions[b * 3 + 1] + s2 * directions[b * 3 + 2];
}

// ---------------------------------------------------------------------- tone-mapped loss (a15)
__device__ __forceinline__ float aces(float c) { return (c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f); }
__device__ __forceinline__ float aces_d(float c) {
    float num = c * (2.51f * c + 0.03f), den = c * (2.43f * c + 0.59f) + 0.14f;
    return ((5.02f * c + 0.03f) * den - num * (4.86f * c + 0.59f)) / (den * den);
}
__device__ __forceinline__ float ldr(float c) { return powf(fminf(fmaxf(aces(c), 0.f), 1.f), 1.f / 2.2f); }
__device__ __forceinline__ float ldr_gt(float c) {
    float a = fminf(fmaxf(aces(c), 0.f), 1.f);
    a = (float)(uint8_t)(a * 255.f) / 255.f;  // .to(torch.uint8) truncates
    return powf(a, 1.f / 2.2f);
}
__device__ __forceinline__ float ldr_d(float c) {  // d ldr / d c
    float a = aces(c);
    if (!(a >= 0.f && a <= 1.f)) return 0.f;
    return (1.f / 2.2f) * powf(a, 1.f / 2.2f - 1.f) * aces_d(c);
}

struct LossArgs {
    int64_t B;
    const float *gt, *mask, *coarse, *fine, *surface, *albedo;
};
// partial[block][8]: 0 coarse num, 1 fine num, 2 surface num, 3 chrom sum, 4 mask sum
__global__ __launch_bounds__(256) void k_loss_partial(LossArgs a, float* partial) {
    __shared__ float red[4][8];
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (b < a.B) {
        float m = a.mask[b];
        float g[3], gl2 = 0.f;
        for (int c = 0; c < 3; ++c) {
            g[c] = ldr_gt(a.gt[b * 3 + c]);
            gl2 += g[c] * g[c];
            float e = ldr(a.coarse[b * 3 + c]) - g[c];
            v[0] += m * e * e;
            e = ldr(a.fine[b * 3 + c]) - g[c];
            v[1] += m * e * e;
            if (a.surface) {
                e = ldr(a.surface[b * 3 + c]) - g[c];
                v[2] += m * e * e;
            }
        }
        if (a.albedo) {
            float al2 = 0.f;
            for (int c = 0; c < 3; ++c) al2 += a.albedo[b * 3 + c] * a.albedo[b * 3 + c];
            float gn = fmaxf(sqrtf(gl2), 1e-12f), an = fmaxf(sqrtf(al2), 1e-12f);
            for (int c = 0; c < 3; ++c) {
                float e = g[c] / gn - a.albedo[b * 3 + c] / an;
                v[3] += e * e;
            }
        }
        v[4] = m;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = 0; k < 5; ++k) {
        float s = group_sum<64>(v[k]);
        if (lane == 0) red[wv][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 5) partial[blockIdx.x * 8 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void k_loss_final(int nblocks, int64_t B, const float* partial, float* terms, float cw, float sw,
                                                     float chw, int has_surface, int has_albedo) {
    __shared__ float red[4][8];
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nblocks; i += 256)
        for (int k = 0; k < 5; ++k) v[k] += partial[i * 8 + k];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = 0; k < 5; ++k) {
        float s = group_sum<64>(v[k]);
        if (lane == 0) red[wv][k] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[5];
        for (int k = 0; k < 5; ++k) t[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        terms[0] = t[0] / t[4];
        terms[1] = t[1] / t[4];
        terms[2] = t[2] / t[4];
        terms[3] = t[3] / (float)(B * 3);
        terms[4] = t[4];
        // the weighted total, in the order the caller used to form it with four small torch kernels (same fp32 operations)
        float total = cw * terms[0] + terms[1];
        if (has_surface) total = total + sw * terms[2];
        if (has_albedo) total = total + chw * terms[3];
        terms[5] = total;
        terms[6] = terms[7] = 0.f;
    }
}

__global__ void k_loss_grad(LossArgs a, const float* terms, float cw, float sw, float chw, float* d_coarse,
                            float* d_fine, float* d_surface, float* d_albedo) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    float msum = terms[4];
    float m = a.mask[b];
    float g[3], gl2 = 0.f;
    for (int c = 0; c < 3; ++c) {
        g[c] = ldr_gt(a.gt[b * 3 + c]);
        gl2 += g[c] * g[c];
        float x = a.coarse[b * 3 + c];
        d_coarse[b * 3 + c] = cw * 2.f * m * (ldr(x) - g[c]) / msum * ldr_d(x);
        x = a.fine[b * 3 + c];
        d_fine[b * 3 + c] = 2.f * m * (ldr(x) - g[c]) / msum * ldr_d(x);
        if (a.surface && d_surface) {
            x = a.surface[b * 3 + c];
            d_surface[b * 3 + c] = sw * 2.f * m * (ldr(x) - g[c]) / msum * ldr_d(x);
        }
    }
    if (a.albedo && d_albedo) {
        float al[3], al2 = 0.f;
        for (int c = 0; c < 3; ++c) {
            al[c] = a.albedo[b * 3 + c];
            al2 += al[c] * al[c];
        }
        float gn = fmaxf(sqrtf(gl2), 1e-12f), alen = sqrtf(al2), an = fmaxf(alen, 1e-12f);
        float e[3], dp = 0.f;
        float k = chw * 2.f / (float)(a.B * 3);
        for (int c = 0; c < 3; ++c) {
            e[c] = -k * (g[c] / gn - al[c] / an);  // dL/d(unit albedo)
            dp += e[c] * al[c] / an;
        }
        for (int c = 0; c < 3; ++c)
            d_albedo[b * 3 + c] = (alen > 1e-12f) ? (e[c] - (al[c] / an) * dp) / an : e[c] / an;
    }
}

// ------------------------------------------------------------------------------------ Adam
__global__ void k_adam(int64_t n, float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps,
                       float bc1, float bc2_sqrt, float gscale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * gscale;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

// graph-replay friendly form: the step counter and the learning rate live in device memory
__global__ void k_adam_tick(int* step) { *step += 1; }
__global__ void k_adam_dev(int64_t n, float* p, const float* g, float* m, float* v, const float* lr_dev, float b1,
                           float b2, float eps, const int* step_dev, float gscale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = (float)(*step_dev);
    const float bc1 = 1.f - powf(b1, t), bc2_sqrt = sqrtf(1.f - powf(b2, t));
    float gi = g[i] * gscale;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (*lr_dev / bc1) * (mi / denom);
}

// ============================================================================ C entry points
static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }
#define ST(s) ((hipStream_t)(s))

int pn_launch_ipe_backward(int64_t M, const float* mean, const float* cov, const float* d_enc, float* d_mean,
                           hipStream_t s) {
    hipLaunchKernelGGL(k_ipe_backward, dim3(nblk(M * 3, 256)), dim3(256), 0, s, M, mean, cov, d_enc, d_mean);
    PN_CHECK_LAUNCH();
    return PN_OK;
}
int pn_launch_ipe_tangent(int64_t M, const float* mean, const float* cov, const float* v, float* edot, hipStream_t s) {
    hipLaunchKernelGGL(k_ipe_tangent, dim3(nblk(M * 48, 256)), dim3(256), 0, s, M, mean, cov, v, edot);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

// A training batch out of the HBM-resident ray pool: out_f[b] = pool_f[idx[b]] for the eight Rays fields
// (3+3+3+1+1+1+1+1 floats) and the target colour (3) in ONE launch (datasets/pano_datasets.py:271-275 does the
// same gather per __getitem__ on the host).  One thread per output float; an index outside the pool reads ray 0.
struct GatherTable {
    const float* src[9];
    float* dst[9];
};
__global__ void k_gather_rays(int64_t B, int64_t pool_rays, const int64_t* idx, GatherTable t) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * 17) return;
    const int64_t b = e / 17;
    const int c = (int)(e % 17);
    // column c of the packed 17-float record -> (field, component)
    int f, k, w;
    if (c < 9) {
        f = c / 3;
        k = c % 3;
        w = 3;
    } else if (c < 14) {
        f = 3 + (c - 9);
        k = 0;
        w = 1;
    } else {
        f = 8;
        k = c - 14;
        w = 3;
    }
    if (!t.src[f]) return;
    int64_t r = idx[b];
    r = (r >= 0 && r < pool_rays) ? r : 0;
    t.dst[f][b * w + k] = t.src[f][r * w + k];
}

extern "C" {

int pn_raygen_pano(int H, int W, const float* c, float near_, float far_, float* origins, float* directions,
                   float* viewdirs, float* radii, float* lossmult, float* near_out, float* far_out, float* noise_var,
                   void* stream) {
    if (H <= 0 || W < 3) return PN_ERR_BAD_SHAPE;
    if (!c || !origins || !directions || !viewdirs || !radii || !lossmult || !near_out || !far_out || !noise_var)
        return PN_ERR_NULL;
    const PanoCam cam{c[0], c[1], c[2], c[4], c[5], c[6], c[8], c[9], c[10], c[3], c[7], c[11]};
    hipLaunchKernelGGL(k_raygen_pano, dim3(nblk((int64_t)H * W, 256)), dim3(256), 0, ST(stream), H, W, cam, near_, far_,
                       origins, directions, viewdirs, radii, lossmult, near_out, far_out, noise_var);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_sample_pano_rays(int64_t B, int n_cam, int H, int W, const int64_t* idx, const float* c2ws, float near_, float far_,
                        const float* rgb_pool, float* origins, float* directions, float* viewdirs, float* radii,
                        float* lossmult, float* near_out, float* far_out, float* noise_var, float* rgb_out, void* stream) {
    if (B <= 0 || n_cam <= 0 || H <= 0 || W < 3) return PN_ERR_BAD_SHAPE;
    if (!idx || !c2ws || !origins || !directions || !viewdirs || !radii || !lossmult || !near_out || !far_out || !noise_var)
        return PN_ERR_NULL;
    if ((rgb_pool == nullptr) != (rgb_out == nullptr)) return PN_ERR_NULL;  // target colours: both or neither
    hipLaunchKernelGGL(k_sample_pano_rays, dim3(nblk(B, 128)), dim3(128), 0, ST(stream), B, n_cam, H, W, idx, c2ws, near_,
                       far_, rgb_pool, origins, directions, viewdirs, radii, lossmult, near_out, far_out, noise_var, rgb_out);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_gather_rays(int64_t B, int64_t pool_rays, const int64_t* idx, const float* const* pool_host, float* const* out_host,
                   void* stream) {
    if (B <= 0 || pool_rays <= 0) return PN_ERR_BAD_SHAPE;
    if (!idx || !pool_host || !out_host) return PN_ERR_NULL;
    GatherTable t;
    for (int f = 0; f < 9; ++f) {
        t.src[f] = pool_host[f];
        t.dst[f] = out_host[f];
        if (f < 8 && (!t.src[f] || !t.dst[f])) return PN_ERR_NULL;       // the eight Rays fields are mandatory
        if (f == 8 && ((t.src[f] == nullptr) != (t.dst[f] == nullptr))) return PN_ERR_NULL;  // rgb: both or neither
    }
    hipLaunchKernelGGL(k_gather_rays, dim3(nblk(B * 17, 256)), dim3(256), 0, ST(stream), B, pool_rays, idx, t);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_lit_rays(int D, double radius, double near_, double far_, uint16_t* out_half, void* stream) {
    if (D < 2) return PN_ERR_BAD_SHAPE;
    if (!out_half) return PN_ERR_NULL;
    hipLaunchKernelGGL(k_lit_rays, dim3(nblk(D, 64)), dim3(64), 0, ST(stream), D, radius, near_, far_, out_half);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_sample_coarse(int64_t B, int N, int disparity, const float* origins, const float* directions, const float* radii,
                     const float* near_, const float* far_, const float* t_rand, float* t_out, float* mean, float* cov,
                     void* stream) {
    if (B <= 0 || N <= 0) return PN_ERR_BAD_SHAPE;
    if (N > PN_MAX_SAMPLES) return PN_ERR_UNSUPPORTED;
    if (!origins || !directions || !radii || !near_ || !far_ || !t_out || !mean || !cov) return PN_ERR_NULL;
    hipLaunchKernelGGL(k_sample_coarse, dim3(nblk(B * N, 256)), dim3(256), 0, ST(stream), B, N, disparity, origins, directions,
                       radii, near_, far_, t_rand, t_out, mean, cov);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_resample(int64_t B, int N, const float* t_in, const float* weights, float padding, const float* u_rand,
                const float* origins, const float* directions, const float* radii, float* t_out, float* mean,
                float* cov, void* stream) {
    if (B <= 0 || N <= 1) return PN_ERR_BAD_SHAPE;
    if (N > PN_MAX_SAMPLES) return PN_ERR_UNSUPPORTED;
    if (!t_in || !weights || !origins || !directions || !radii || !t_out || !mean || !cov) return PN_ERR_NULL;
    hipLaunchKernelGGL(k_resample, dim3(nblk(B, 4)), dim3(256), 0, ST(stream), B, N, t_in, weights, padding, u_rand,
                       origins, directions, radii, t_out, mean, cov);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_sample_env(int64_t B, int D, int Ne, const float* origins, const float* directions, const float* distance,
                  const float* env_dirs, const float* env_radii, const float* env_near, const float* env_far,
                  const float* env_rand, float* t_out, float* mean, float* cov, void* stream) {
    if (B <= 0 || D <= 0 || Ne <= 0) return PN_ERR_BAD_SHAPE;
    if (!origins || !directions || !distance || !env_dirs || !env_radii || !env_near || !env_far || !t_out || !mean ||
        !cov)
        return PN_ERR_NULL;
    hipLaunchKernelGGL(k_sample_env, dim3(nblk(B * D * Ne, 256)), dim3(256), 0, ST(stream), B, D, Ne, origins,
                       directions, distance, env_dirs, env_radii, env_near, env_far, env_rand, t_out, mean, cov);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_ipe_encode(int64_t M, const float* mean, const float* cov, float* enc, void* stream) {
    if (M <= 0) return PN_ERR_BAD_SHAPE;
    if (!mean || !cov || !enc) return PN_ERR_NULL;
    hipLaunchKernelGGL(k_ipe_encode, dim3(nblk(M * 48, 256)), dim3(256), 0, ST(stream), M, mean, cov, enc);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_pos_enc_view(int64_t R, const float* viewdirs, float* viewenc, void* stream) {
    if (R <= 0) return PN_ERR_BAD_SHAPE;
    if (!viewdirs || !viewenc) return PN_ERR_NULL;
    hipLaunchKernelGGL(k_pos_enc_view, dim3(nblk(R * PN_VIEW_DIM, 256)), dim3(256), 0, ST(stream), R, viewdirs,
                       viewenc);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

static int group_for(int N) { return N <= 16 ? 16 : (N <= 32 ? 32 : 64); }

int pn_composite_forward(int64_t R, int N, int nc, float density_bias, float rgb_padding, int white_bkgd,
                         const float* raw_rgb, const float* raw_density, const float* t, const float* dirs,
                         int64_t dir_mod, float* comp_rgb, float* distance, float* acc, float* weights, void* stream) {
    if (R <= 0 || N <= 0 || dir_mod <= 0) return PN_ERR_BAD_SHAPE;
    if (N > PN_MAX_SAMPLES || (nc != 1 && nc != 5)) return PN_ERR_UNSUPPORTED;
    if (!raw_rgb || !raw_density || !t || !dirs || !comp_rgb || !distance || !acc || !weights) return PN_ERR_NULL;
    CompArgs a{R, N, nc, density_bias, rgb_padding, white_bkgd, raw_rgb, raw_density, t, dirs, dir_mod};
    int G = group_for(N);
    dim3 grid(nblk(R * G, 256)), blk(256);
    if (G == 16) hipLaunchKernelGGL(k_composite_fwd<16>, grid, blk, 0, ST(stream), a, comp_rgb, distance, acc, weights);
    else if (G == 32) hipLaunchKernelGGL(k_composite_fwd<32>, grid, blk, 0, ST(stream), a, comp_rgb, distance, acc, weights);
    else hipLaunchKernelGGL(k_composite_fwd<64>, grid, blk, 0, ST(stream), a, comp_rgb, distance, acc, weights);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_composite_backward(int64_t R, int N, int nc, float density_bias, float rgb_padding, int white_bkgd,
                          const float* raw_rgb, const float* raw_density, const float* t, const float* dirs,
                          int64_t dir_mod, const float* d_comp_rgb, const float* d_distance, const float* d_weights,
                          float* d_raw_rgb, float* d_raw_density, void* stream) {
    if (R <= 0 || N <= 0 || dir_mod <= 0) return PN_ERR_BAD_SHAPE;
    if (N > PN_MAX_SAMPLES || (nc != 1 && nc != 5)) return PN_ERR_UNSUPPORTED;
    if (!raw_rgb || !raw_density || !t || !dirs || !d_comp_rgb || !d_raw_rgb || !d_raw_density) return PN_ERR_NULL;
    CompArgs a{R, N, nc, density_bias, rgb_padding, white_bkgd, raw_rgb, raw_density, t, dirs, dir_mod};
    int G = group_for(N);
    dim3 grid(nblk(R * G, 256)), blk(256);
    if (G == 16) hipLaunchKernelGGL(k_composite_bwd<16>, grid, blk, 0, ST(stream), a, d_comp_rgb, d_distance, d_weights, d_raw_rgb, d_raw_density);
    else if (G == 32) hipLaunchKernelGGL(k_composite_bwd<32>, grid, blk, 0, ST(stream), a, d_comp_rgb, d_distance, d_weights, d_raw_rgb, d_raw_density);
    else hipLaunchKernelGGL(k_composite_bwd<64>, grid, blk, 0, ST(stream), a, d_comp_rgb, d_distance, d_weights, d_raw_rgb, d_raw_density);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_surf_gather_forward(int64_t B, int N, int nc, const float* grad_mean, const float* weights,
                           const float* raw_density, const float* directions, float* normal, float* ort_ray,
                           float* albedo, void* stream) {
    if (B <= 0 || N <= 0) return PN_ERR_BAD_SHAPE;
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    if (!grad_mean || !weights || !raw_density || !directions || !normal) return PN_ERR_NULL;
    GatherArgs a{B, N, nc, grad_mean, weights, raw_density, directions};
    int G = group_for(N);
    dim3 grid(nblk(B * G, 256)), blk(256);
    if (G == 16) hipLaunchKernelGGL(k_gather_fwd<16>, grid, blk, 0, ST(stream), a, normal, ort_ray, albedo);
    else if (G == 32) hipLaunchKernelGGL(k_gather_fwd<32>, grid, blk, 0, ST(stream), a, normal, ort_ray, albedo);
    else hipLaunchKernelGGL(k_gather_fwd<64>, grid, blk, 0, ST(stream), a, normal, ort_ray, albedo);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_surf_gather_backward(int64_t B, int N, int nc, const float* grad_mean, const float* weights,
                            const float* raw_density, const float* directions, const float* d_normal,
                            const float* d_ort_ray, const float* d_albedo, float* d_weights, float* v_gradmean,
                            float* d_raw_density, void* stream) {
    if (B <= 0 || N <= 0) return PN_ERR_BAD_SHAPE;
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    if (!grad_mean || !weights || !raw_density || !directions || !d_normal || !d_weights || !v_gradmean) return PN_ERR_NULL;
    if (nc >= 5 && !d_raw_density) return PN_ERR_NULL;
    GatherArgs a{B, N, nc, grad_mean, weights, raw_density, directions};
    int G = group_for(N);
    dim3 grid(nblk(B * G, 256)), blk(256);
    if (G == 16) hipLaunchKernelGGL(k_gather_bwd<16>, grid, blk, 0, ST(stream), a, d_normal, d_ort_ray, d_albedo, d_weights, v_gradmean, d_raw_density);
    else if (G == 32) hipLaunchKernelGGL(k_gather_bwd<32>, grid, blk, 0, ST(stream), a, d_normal, d_ort_ray, d_albedo, d_weights, v_gradmean, d_raw_density);
    else hipLaunchKernelGGL(k_gather_bwd<64>, grid, blk, 0, ST(stream), a, d_normal, d_ort_ray, d_albedo, d_weights, v_gradmean, d_raw_density);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_surface_forward(int64_t B, int D, const float* env_rgb, const float* albedo, const float* normal,
                       const float* env_dirs, const float* solid_angle, float* diffuse, float* shading, void* stream) {
    if (B <= 0 || D <= 0) return PN_ERR_BAD_SHAPE;
    if (!env_rgb || !albedo || !normal || !env_dirs || !solid_angle || !diffuse || !shading) return PN_ERR_NULL;
    hipLaunchKernelGGL(k_surface_fwd, dim3(nblk(B, 256)), dim3(256), 0, ST(stream), B, D, env_rgb, albedo, normal,
                       env_dirs, solid_angle, diffuse, shading);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_surface_backward(int64_t B, int D, const float* env_rgb, const float* albedo, const float* normal,
                        const float* env_dirs, const float* solid_angle, const float* d_diffuse, const float* d_shading,
                        float* d_env_rgb, float* d_albedo, float* d_normal, void* stream) {
    if (B <= 0 || D <= 0) return PN_ERR_BAD_SHAPE;
    if (!env_rgb || !albedo || !normal || !env_dirs || !solid_angle || !d_env_rgb || !d_albedo || !d_normal)
        return PN_ERR_NULL;
    hipLaunchKernelGGL(k_surface_bwd, dim3(nblk(B, 256)), dim3(256), 0, ST(stream), B, D, env_rgb, albedo, normal,
                       env_dirs, solid_angle, d_diffuse, d_shading, d_env_rgb, d_albedo, d_normal);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_env_origin_backward(int64_t B, int rows_per_ray, const float* d_mean, const float* directions,
                           float* d_distance, void* stream) {
    if (B <= 0 || rows_per_ray <= 0) return PN_ERR_BAD_SHAPE;
    if (!d_mean || !directions || !d_distance) return PN_ERR_NULL;
    hipLaunchKernelGGL(k_env_origin_bwd, dim3(nblk(B * 64, 256)), dim3(256), 0, ST(stream), B, rows_per_ray, d_mean,
                       directions, d_distance);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_tonemap_loss(int64_t B, const float* rgb_gt_hdr, const float* lossmult, const float* rgb_coarse,
                    const float* rgb_fine, const float* rgb_surface, const float* albedo, float coarse_w,
                    float surface_w, float chrom_w, float* loss_terms, float* d_coarse, float* d_fine,
                    float* d_surface, float* d_albedo, float* work, void* stream) {
    if (B <= 0) return PN_ERR_BAD_SHAPE;
    if (!rgb_gt_hdr || !lossmult || !rgb_coarse || !rgb_fine || !loss_terms || !work) return PN_ERR_NULL;
    LossArgs a{B, rgb_gt_hdr, lossmult, rgb_coarse, rgb_fine, rgb_surface, albedo};
    int nb = (int)nblk(B, 256);
    hipLaunchKernelGGL(k_loss_partial, dim3(nb), dim3(256), 0, ST(stream), a, work);
    PN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(256), 0, ST(stream), nb, B, work, loss_terms, coarse_w, surface_w, chrom_w,
                       rgb_surface != nullptr, albedo != nullptr);
    PN_CHECK_LAUNCH();
    if (d_coarse && d_fine) {
        hipLaunchKernelGGL(k_loss_grad, dim3(nb), dim3(256), 0, ST(stream), a, loss_terms, coarse_w, surface_w, chrom_w,
                           d_coarse, d_fine, d_surface, d_albedo);
        PN_CHECK_LAUNCH();
    }
    return PN_OK;
}

int pn_adam_step(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                 float beta2, float eps, int step, float grad_scale, void* stream) {
    if (n <= 0 || step <= 0) return PN_ERR_BAD_SHAPE;
    if (!params || !grads || !exp_avg || !exp_avg_sq) return PN_ERR_NULL;
    double bc1 = 1.0 - pow((double)beta1, (double)step);
    double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(k_adam, dim3(nblk(n, 256)), dim3(256), 0, ST(stream), n, params, grads, exp_avg, exp_avg_sq, lr,
                       beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_adam_step_dev(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                     const float* lr_dev, float beta1, float beta2, float eps, int* step_dev, float grad_scale,
                     void* stream) {
    if (n <= 0) return PN_ERR_BAD_SHAPE;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev) return PN_ERR_NULL;
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, ST(stream), step_dev);
    PN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_adam_dev, dim3(nblk(n, 256)), dim3(256), 0, ST(stream), n, params, grads, exp_avg, exp_avg_sq,
                       lr_dev, beta1, beta2, eps, step_dev, grad_scale);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

}  // extern "C"
