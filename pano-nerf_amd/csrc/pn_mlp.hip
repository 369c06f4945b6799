// pn_mlp.hip — the radiance MLP (8x256 trunk with a skip concat into layer 5, 5/1-channel density
// head, 256->256 bottleneck, 283->128 view layer, 128->3 colour head) as chains of the exact-fp32
// MFMA GEMMs of pn_gemm.hip, plus the narrow heads on the VALU.
//
//   pn_mlp_forward   : enc -> h0..h7 -> {raw_density, bottleneck -> view hidden -> raw_rgb}
//   pn_density_grad  : ONE reverse sweep r_l = relu'(h_l) * (r_{l+1} W_{l+1}) seeded with
//                      softplus'(z) * density_layer.weight[0]  (the reference differentiates all 8 outputs
//                      with vmap(jacrev) and keeps one row)
//   pn_mlp_backward  : data + weight gradients; with v = dL/d(grad_mean) it also runs the forward-mode
//                      tangent sweep hdot_l = relu'(h_l) * (hdot_{l-1} W_l^T) whose outer products with
//                      the saved r_l give the second-order weight gradients (ReLU'' = 0 a.e.).
//
// Layer activations are written to HBM once (they are needed by the weight gradients and as ReLU
// gates); at fp32-MFMA rate every layer GEMM is compute-bound (64 FLOP/B vs a 25 FLOP/B machine
// balance), see DESIGN.md.
#include "pn_common.h"
#include <math.h>
#include <mutex>
#include <vector>

#define ST(s) ((hipStream_t)(s))
// PN_DEBUG_SYNC=1: synchronise the device after every step of the MLP entry points and name the step on stderr, so
// an asynchronous fault is attributed to the launch that caused it (debugging aid; never set in production)
static bool debug_sync() {
    static const bool on = getenv("PN_DEBUG_SYNC") != nullptr;
    return on;
}
#define RUN(x)                                                                 \
    do {                                                                       \
        int rc_ = (x);                                                         \
        if (rc_ != PN_OK) return rc_;                                          \
        if (debug_sync()) {                                                    \
            fprintf(stderr, "[pn] %s:%d %s\n", __func__, __LINE__, #x);        \
            if (hipDeviceSynchronize() != hipSuccess) return PN_ERR_HIP;       \
        }                                                                      \
    } while (0)
static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

// -------------------------------------------------------------------------------- layouts
PnLayout pn_layout(int nc) {
    PnLayout L;
    L.nc = nc;
    int64_t o = 0;
    for (int l = 0; l < 8; ++l) {
        int k = (l == 0) ? PN_ENC_DIM : (l == 5 ? PN_WIDTH + PN_ENC_DIM : PN_WIDTH);
        L.w[l] = o;
        o += (int64_t)PN_WIDTH * k;
        L.b[l] = o;
        o += PN_WIDTH;
    }
    L.we = o; o += PN_WIDTH * PN_WIDTH;
    L.be = o; o += PN_WIDTH;
    L.wv = o; o += PN_WIDTH_COND * (PN_WIDTH + PN_VIEW_DIM);
    L.bv = o; o += PN_WIDTH_COND;
    L.wd = o; o += (int64_t)nc * PN_WIDTH;
    L.wc = o; o += 3 * PN_WIDTH_COND;
    L.bd = o; o += nc;
    L.bc = o; o += 3;
    L.total = o;
    return L;
}

PnPack pn_pack_layout() {
    PnPack P;
    int64_t o = 0;
    for (int l = 0; l < 8; ++l) {
        P.wt[l] = o;
        o += (int64_t)PN_WIDTH * ((l == 0) ? PN_ENC_DIM : PN_WIDTH);
    }
    P.w5e_t = o; o += PN_ENC_DIM * PN_WIDTH;
    P.we_t = o; o += PN_WIDTH * PN_WIDTH;
    P.wvm = o; o += PN_WIDTH_COND * PN_WIDTH;
    P.wvm_t = o; o += PN_WIDTH * PN_WIDTH_COND;
    P.wvv = o; o += PN_WIDTH_COND * 32;
    P.total = o;
    return P;
}

// dst[r][c] = (c < cols) ? src[r*ld + c0 + c] : 0   (dst leading dim = dcols)
__global__ void k_copy_cols(const float* src, int ld, int c0, int rows, int cols, int dcols, float* dst) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * dcols) return;
    int r = idx / dcols, c = idx % dcols;
    dst[idx] = (c < cols) ? src[(int64_t)r * ld + c0 + c] : 0.f;
}

// all weight transposes of one pn_pack_weights call in ONE launch: a workgroup finds its job in a small table
struct TrJob {
    const float* src;
    float* dst;
    int ld, c0, rows, cols, tile0, tiles_x;
};
#define TR_MAXJOBS 12
struct TrBatch {
    TrJob j[TR_MAXJOBS];
    int n;
};
// per job: dst[c][r] = src[r*ld + c0 + c]  for r < rows, c < cols   (dst leading dim = rows)
__global__ void k_transpose_batch(TrBatch tb) {
    __shared__ float tile[32][33];
    int k = 0;
    for (int i = 1; i < tb.n; ++i)
        if ((int)blockIdx.x >= tb.j[i].tile0) k = i;
    const TrJob jb = tb.j[k];
    const int t = blockIdx.x - jb.tile0;
    const int bx = (t % jb.tiles_x) * 32, by = (t / jb.tiles_x) * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        int r = by + i, c = bx + threadIdx.x;
        tile[i][threadIdx.x] = (r < jb.rows && c < jb.cols) ? jb.src[(int64_t)r * jb.ld + jb.c0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        int c = bx + i, r = by + threadIdx.x;
        if (r < jb.rows && c < jb.cols) jb.dst[(int64_t)c * jb.rows + r] = tile[threadIdx.x][i];
    }
}
struct TrQueue {
    TrBatch b;
    int tiles;
    TrQueue() : tiles(0) { b.n = 0; }
    int add(const float* src, int ld, int c0, int rows, int cols, float* dst) {
        if (b.n >= TR_MAXJOBS) return PN_ERR_BAD_SHAPE;
        const int tx = (cols + 31) / 32, ty = (rows + 31) / 32;
        b.j[b.n++] = TrJob{src, dst, ld, c0, rows, cols, tiles, tx};
        tiles += tx * ty;
        return PN_OK;
    }
    int launch(hipStream_t s) {
        if (!tiles) return PN_OK;
        hipLaunchKernelGGL(k_transpose_batch, dim3(tiles), dim3(32, 8), 0, s, b);
        PN_CHECK_LAUNCH();
        return PN_OK;
    }
};

// -------------------------------------------------------------------------- narrow heads (VALU, HBM-bound)
// Rows are [K] floats, K = 128 or 256.  A wave covers FOUR rows at a time: 16 lanes per row, each lane owning
// J = K/64 float4 (columns (16j + q)*4 .. +3, q = lane & 15), so every load/store instruction moves four full
// 256-B row segments, a row reduction is four DPP rotations inside the 16-lane row, and the NC x K weights are
// read as LDS broadcasts.  (One wave per row needed 6 cross-lane steps per output and ran at 1.7 TB/s.)
__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row; every lane gets it
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}
__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) {
    return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
}
#define HEAD_U 2  // row quads in flight per wave iteration

// out[row,c] = b[c] + sum_k x[row,k] * W[c,k]
template <int K, int NC>
__global__ __launch_bounds__(256) void k_head_fwd(int64_t M, const float* x, int ldx, const float* W, const float* b,
                                                   float* out, int ldo) {
    constexpr int J = K / 64;
    __shared__ __attribute__((aligned(16))) float Ws[NC * K];
    for (int i = threadIdx.x; i < NC * K; i += 256) Ws[i] = W[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, q = lane & 15;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float bq = (b && q < NC) ? b[q] : 0.f;
    for (int64_t row0 = wave * (4 * HEAD_U); row0 < M; row0 += nwaves * (4 * HEAD_U)) {
        f32x4 xv[HEAD_U][J];
#pragma unroll
        for (int u = 0; u < HEAD_U; ++u) {
            int64_t row = row0 + 4 * u + g;
            row = row < M ? row : M - 1;
#pragma unroll
            for (int j = 0; j < J; ++j) xv[u][j] = *reinterpret_cast<const f32x4*>(x + row * ldx + (16 * j + q) * 4);
        }
        float s[HEAD_U][NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int u = 0; u < HEAD_U; ++u) s[u][c] = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(Ws + c * K + (16 * j + q) * 4);
#pragma unroll
                for (int u = 0; u < HEAD_U; ++u) s[u][c] += dot4(xv[u][j], w4);
            }
        }
#pragma unroll
        for (int u = 0; u < HEAD_U; ++u) {
            float mine = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float t = row16_sum(s[u][c]);
                mine = (q == c) ? t : mine;
            }
            const int64_t row = row0 + 4 * u + g;
            if (q < NC && row < M) out[row * ldo + q] = mine + bq;
        }
    }
}

// out[row,k] = sum_c d[row,c] * W[c,k]  (optionally gated by gate[row,k] > 0)
template <int K, int NC>
__global__ __launch_bounds__(256) void k_head_bwd_data(int64_t M, const float* d, int ldd, const float* W, float* out,
                                                        int ldo, const float* gate, int ldg) {
    constexpr int J = K / 64;
    __shared__ __attribute__((aligned(16))) float Ws[NC * K];
    for (int i = threadIdx.x; i < NC * K; i += 256) Ws[i] = W[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, q = lane & 15;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t row0 = wave * (4 * HEAD_U); row0 < M; row0 += nwaves * (4 * HEAD_U)) {
        float dv[HEAD_U][NC];
        f32x4 gt[HEAD_U][J];
#pragma unroll
        for (int u = 0; u < HEAD_U; ++u) {
            int64_t row = row0 + 4 * u + g;
            row = row < M ? row : M - 1;
#pragma unroll
            for (int c = 0; c < NC; ++c) dv[u][c] = d[row * ldd + c];
            if (gate) {
#pragma unroll
                for (int j = 0; j < J; ++j) gt[u][j] = *reinterpret_cast<const f32x4*>(gate + row * ldg + (16 * j + q) * 4);
            }
        }
        f32x4 o[HEAD_U][J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
#pragma unroll
            for (int u = 0; u < HEAD_U; ++u) o[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(Ws + c * K + (16 * j + q) * 4);
#pragma unroll
                for (int u = 0; u < HEAD_U; ++u) o[u][j] += dv[u][c] * w4;
            }
        }
#pragma unroll
        for (int u = 0; u < HEAD_U; ++u) {
            const int64_t row = row0 + 4 * u + g;
            if (row >= M) continue;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                f32x4 v = o[u][j];
                if (gate) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gt[u][j][e] > 0.f ? v[e] : 0.f;
                }
                *reinterpret_cast<f32x4*>(out + row * ldo + (16 * j + q) * 4) = v;
            }
        }
    }
}

// partial[blk][c][k] = sum_{rows of blk} coef(row) * d[row,c] * x[row,k]; partial bias in [blk][NC*K + c]
// coef(row) = coef ? coef[row] : 1.  16 row slots per block (4 waves x 4 row groups).
template <int K, int NC>
__global__ __launch_bounds__(256) void k_head_bwd_weight(int64_t M, int rows_per_block, const float* d, int ldd,
                                                          const float* coef, const float* x, int ldx, float* partial) {
    constexpr int J = K / 64;
    __shared__ __attribute__((aligned(16))) float red[4][NC * K + NC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, q = lane & 15;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    f32x4 acc[NC][J];
    float bacc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        bacc[c] = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) acc[c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t rb = r0 + wv * 4 + g; rb < r1; rb += 16 * HEAD_U) {
        f32x4 xv[HEAD_U][J];
        float dv[HEAD_U][NC];
#pragma unroll
        for (int u = 0; u < HEAD_U; ++u) {
            const int64_t row = rb + 16 * u;
            const bool in = row < r1;
            const int64_t rc = in ? row : r1 - 1;
            const float cf = in ? (coef ? coef[rc] : 1.f) : 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) xv[u][j] = *reinterpret_cast<const f32x4*>(x + rc * ldx + (16 * j + q) * 4);
#pragma unroll
            for (int c = 0; c < NC; ++c) dv[u][c] = d[rc * ldd + c] * cf;
        }
#pragma unroll
        for (int u = 0; u < HEAD_U; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                bacc[c] += dv[u][c];
#pragma unroll
                for (int j = 0; j < J; ++j) acc[c][j] += dv[u][c] * xv[u][j];
            }
    }
    // the four row groups of a wave, then the four waves
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            f32x4 v = acc[c][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[e];
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                v[e] = t;
            }
            if (g == 0) *reinterpret_cast<f32x4*>(&red[wv][c * K + (16 * j + q) * 4]) = v;
        }
        float t = bacc[c];
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (lane == 0) red[wv][NC * K + c] = t;
    }
    __syncthreads();
    float* out = partial + (int64_t)blockIdx.x * (NC * K + NC);
    for (int i = threadIdx.x; i < NC * K + NC; i += 256) out[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// S[ray][col] = sum over the rows_per_ray consecutive rows of a ray of X[row][col], K = 128 columns: feeds the
// view-layer bias gradient (sum over rays) and its view-encoding weight gradient (the view encoding is constant
// along a ray, so dWv[:, 256:] = sum_rays S[ray]^T viewenc[ray] needs no per-sample expansion).  One wave per ray.
template <int K>
__global__ __launch_bounds__(256) void k_ray_colsum(int64_t R, int rows_per_ray, const float* X, int ldx, float* S) {
    constexpr int J = K / 64;
    const int lane = threadIdx.x & 63, g = lane >> 4, q = lane & 15;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t ray = wave; ray < R; ray += nwaves) {
        const float* base = X + ray * rows_per_ray * (int64_t)ldx;
        f32x4 acc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int r = g;
        for (; r + 12 < rows_per_ray; r += 16) {
            f32x4 t[4][J];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < J; ++j)
                    t[u][j] = *reinterpret_cast<const f32x4*>(base + (int64_t)(r + 4 * u) * ldx + (16 * j + q) * 4);
#pragma unroll
            for (int j = 0; j < J; ++j) acc[j] += (t[0][j] + t[1][j]) + (t[2][j] + t[3][j]);
        }
        for (; r < rows_per_ray; r += 4) {
#pragma unroll
            for (int j = 0; j < J; ++j) acc[j] += *reinterpret_cast<const f32x4*>(base + (int64_t)r * ldx + (16 * j + q) * 4);
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            f32x4 v = acc[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[e];
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                v[e] = t;
            }
            if (g == 0) *reinterpret_cast<f32x4*>(S + ray * K + (16 * j + q) * 4) = v;
        }
    }
}

// VE[v][i] = viewenc[v][i] padded to 32 columns (16-B rows for the TN GEMM dWv[:, 256:] = S2^T VE, one row per RAY)
__global__ void k_pad_viewenc(int64_t V, const float* viewenc, float* VE) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= V * 32) return;
    const int64_t v = idx >> 5;
    const int i = (int)(idx & 31);
    VE[idx] = (i < PN_VIEW_DIM) ? viewenc[v * PN_VIEW_DIM + i] : 0.f;
}

struct BiasOffsets {
    int64_t off[9];
};
__global__ void k_bias_scatter(const float* tmp, BiasOffsets bo, float* grads) {
    grads[bo.off[blockIdx.x] + threadIdx.x] += tmp[blockIdx.x * PN_WIDTH + threadIdx.x];
}

// ------------------------------------------------------------------------------ small fused bits
// per-view-row bias of the view layer: vb[r][j] = bv[j] + sum_i viewenc[r][i] * Wv[j][256 + i]
__global__ void k_view_bias(int64_t R, const float* viewenc, const float* Wv, const float* bv, float* vb) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * PN_WIDTH_COND) return;
    int64_t r = idx / PN_WIDTH_COND;
    int j = (int)(idx % PN_WIDTH_COND);
    const float* w = Wv + (int64_t)j * (PN_WIDTH + PN_VIEW_DIM) + PN_WIDTH;
    float s = 0.f;
    for (int i = 0; i < PN_VIEW_DIM; ++i) s += viewenc[r * PN_VIEW_DIM + i] * w[i];
    vb[idx] = s + bv[j];
}

__device__ __forceinline__ float sp_d1(float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float sp_d2(float x) {
    if (x > 20.f) return 0.f;
    float s = 1.f / (1.f + expf(-x));
    return s * (1.f - s);
}

// seed of the density-gradient sweep: r7[row][j] = softplus'(z_row) * Wd[0][j] * [h7[row][j] > 0]
__global__ void k_dgrad_seed(int64_t M, int nc, float bias, const float* raw_density, const float* Wd,
                             const uint32_t* mask7, float* r7) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per float4 of the output
    if (idx >= M * (PN_WIDTH / 4)) return;
    int64_t row = idx >> 6;
    int j = (int)(idx & 63) * 4;
    float s = sp_d1(raw_density[row * nc] + bias);
    uint32_t wbits = mask7[row * PN_MASK_WORDS + (j >> 5)];
    const int bi = (j >> 2) & 7;
    f32x4 wv = *reinterpret_cast<const f32x4*>(Wd + j), o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = ((wbits >> (c * 8 + bi)) & 1u) ? s * wv[c] : 0.f;
    *reinterpret_cast<f32x4*>(r7 + row * PN_WIDTH + j) = o;
}

// second-order seed: dden[row][0] += softplus''(z) * sdot[row];   coef[row] = softplus'(z)
__global__ void k_second_order_seed(int64_t M, int nc, float bias, const float* raw_density, const float* sdot,
                                    const float* dden_in, float* dden_out, float* coef) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * nc) return;
    int64_t row = idx / nc;
    int c = (int)(idx % nc);
    float z = raw_density[row * nc] + bias;
    float v = dden_in[idx];
    if (c == 0) {
        v += sp_d2(z) * sdot[row];
        coef[row] = sp_d1(z);
    }
    dden_out[idx] = v;
}

// -------------------------------------------------------------------------------- GEMM sugar
static PnGemmNt nt(int64_t M, int N, const float* A, int lda, const float* B, int ldb, int K, float* C, int ldc) {
    PnGemmNt g{};
    g.seg[0] = PnSeg{A, B, lda, ldb, K};
    g.nseg = 1;
    g.C = C;
    g.ldc = ldc;
    g.M = M;
    g.N = N;
    return g;
}
static void seg2(PnGemmNt& g, const float* A, int lda, const float* B, int ldb, int K) {
    g.seg[1] = PnSeg{A, B, lda, ldb, K};
    g.nseg = 2;
}

static unsigned head_grid(int64_t M) {  // 32 rows per block iteration, grid-stride, at most 8 blocks per CU
    const int64_t nb = (M + 4 * 4 * HEAD_U - 1) / (4 * 4 * HEAD_U);
    return (unsigned)(nb < 2048 ? (nb < 1 ? 1 : nb) : 2048);
}
template <int VEC, int NC>
static int head_fwd(int64_t M, const float* x, int ldx, const float* W, const float* b, float* out, int ldo,
                    hipStream_t s) {
    if ((ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return PN_ERR_BAD_SHAPE;
    hipLaunchKernelGGL((k_head_fwd<64 * VEC, NC>), dim3(head_grid(M)), dim3(256), 0, s, M, x, ldx, W, b, out, ldo);
    PN_CHECK_LAUNCH();
    return PN_OK;
}
template <int VEC, int NC>
static int head_bwd_data(int64_t M, const float* d, int ldd, const float* W, float* out, int ldo, const float* gate,
                         int ldg, hipStream_t s) {
    if ((ldo & 3) || (reinterpret_cast<uintptr_t>(out) & 15)) return PN_ERR_BAD_SHAPE;
    if (gate && ((ldg & 3) || (reinterpret_cast<uintptr_t>(gate) & 15))) return PN_ERR_BAD_SHAPE;
    hipLaunchKernelGGL((k_head_bwd_data<64 * VEC, NC>), dim3(head_grid(M)), dim3(256), 0, s, M, d, ldd, W, out, ldo, gate,
                       ldg);
    PN_CHECK_LAUNCH();
    return PN_OK;
}
#define HEAD_ROWS 128
// dW[NC][K] += sum coef * d^T x ; db[NC] += sum coef * d  (db may be null)
template <int VEC, int NC>
static int head_bwd_weight(int64_t M, const float* d, int ldd, const float* coef, const float* x, int ldx, float* dW,
                           float* db, float* partial, hipStream_t s) {
    constexpr int K = 64 * VEC;
    int nb = (int)nblk(M, HEAD_ROWS);
    if ((ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return PN_ERR_BAD_SHAPE;
    hipLaunchKernelGGL((k_head_bwd_weight<64 * VEC, NC>), dim3(nb), dim3(256), 0, s, M, HEAD_ROWS, d, ldd, coef, x, ldx,
                       partial);
    PN_CHECK_LAUNCH();
    float* scratch = partial + (int64_t)nb * (NC * K + NC);
    RUN(pn_launch_reduce_rows(partial, nb, NC * K + NC, NC, K, K, dW, K, 1, scratch, s));
    if (db) RUN(pn_launch_reduce_rows(partial + NC * K, nb, NC * K + NC, 1, NC, NC, db, NC, 1, scratch, s));
    return PN_OK;
}
static int wgrad(int64_t M, const float* X, int ldx, int N1, const float* Y, int ldy, int N2, float* dW, int ldw,
                 float* work, int64_t work_avail, hipStream_t s) {
    PnSegTn sg{X, Y, ldx, ldy, M};
    return pn_launch_gemm_tn(&sg, 1, N1, N2, dW, ldw, 1, work, work_avail, s);
}


extern "C" {

int64_t pn_param_layout(int nc, int64_t* off) {
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    PnLayout L = pn_layout(nc);
    if (off) {
        for (int l = 0; l < 8; ++l) {
            off[2 * l] = L.w[l];
            off[2 * l + 1] = L.b[l];
        }
        off[16] = L.we; off[17] = L.be; off[18] = L.wv; off[19] = L.bv;
        off[20] = L.wd; off[21] = L.wc; off[22] = L.bd; off[23] = L.bc;
    }
    return L.total;
}

int64_t pn_wpack_floats(int nc) {
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    return pn_pack_layout().total;
}

int pn_pack_weights(const float* params, int nc, float* wpack, void* stream) {
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    if (!params || !wpack) return PN_ERR_NULL;
    PnLayout L = pn_layout(nc);
    PnPack P = pn_pack_layout();
    hipStream_t s = ST(stream);
    TrQueue tq;
    for (int l = 0; l < 8; ++l) {
        int k = (l == 0) ? PN_ENC_DIM : (l == 5 ? PN_WIDTH + PN_ENC_DIM : PN_WIDTH);
        int kk = (l == 0) ? PN_ENC_DIM : PN_WIDTH;
        RUN(tq.add(params + L.w[l], k, 0, PN_WIDTH, kk, wpack + P.wt[l]));  // [kk][256]
    }
    RUN(tq.add(params + L.w[5], PN_WIDTH + PN_ENC_DIM, PN_WIDTH, PN_WIDTH, PN_ENC_DIM, wpack + P.w5e_t));
    RUN(tq.add(params + L.we, PN_WIDTH, 0, PN_WIDTH, PN_WIDTH, wpack + P.we_t));
    const int ldv = PN_WIDTH + PN_VIEW_DIM;
    RUN(tq.add(params + L.wv, ldv, 0, PN_WIDTH_COND, PN_WIDTH, wpack + P.wvm_t));  // [256][128]
    RUN(tq.launch(s));
    hipLaunchKernelGGL(k_copy_cols, dim3(nblk(PN_WIDTH_COND * PN_WIDTH, 256)), dim3(256), 0, s, params + L.wv, ldv, 0,
                       PN_WIDTH_COND, PN_WIDTH, PN_WIDTH, wpack + P.wvm);
    PN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_copy_cols, dim3(nblk(PN_WIDTH_COND * 32, 256)), dim3(256), 0, s, params + L.wv, ldv, PN_WIDTH,
                       PN_WIDTH_COND, PN_VIEW_DIM, 32, wpack + P.wvv);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

int pn_mlp_forward(int64_t M, int rows_per_ray, int64_t view_rows, int nc, const float* params, const float* wpack,
                   const float* mean, const float* cov, const float* viewdirs, float* enc, float* viewenc,
                   float* viewbias, float* acts, uint32_t* masks, float* raw_rgb, float* raw_density, void* stream) {
    if (M <= 0 || rows_per_ray <= 0 || view_rows <= 0) return PN_ERR_BAD_SHAPE;
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    if (!params || !wpack || !mean || !cov || !viewdirs || !enc || !viewenc || !viewbias || !acts || !masks ||
        !raw_rgb || !raw_density)
        return PN_ERR_NULL;
    hipStream_t s = ST(stream);
    PnLayout L = pn_layout(nc);
    PnPack P = pn_pack_layout();
    const int64_t Mp = pn_pad(M);
    auto act = [&](int i) { return acts + (int64_t)i * Mp * PN_WIDTH; };

    RUN(pn_ipe_encode(M, mean, cov, enc, stream));
    RUN(pn_pos_enc_view(view_rows, viewdirs, viewenc, stream));
    hipLaunchKernelGGL(k_view_bias, dim3(nblk(view_rows * PN_WIDTH_COND, 256)), dim3(256), 0, s, view_rows, viewenc,
                       params + L.wv, params + L.bv, viewbias);
    PN_CHECK_LAUNCH();

    // trunk
    for (int l = 0; l < 8; ++l) {
        PnGemmNt g;
        if (l == 0) {
            g = nt(M, PN_WIDTH, enc, PN_ENC_DIM, params + L.w[0], PN_ENC_DIM, PN_ENC_DIM, act(0), PN_WIDTH);
        } else if (l == 5) {
            const int ld5 = PN_WIDTH + PN_ENC_DIM;
            g = nt(M, PN_WIDTH, act(4), PN_WIDTH, params + L.w[5], ld5, PN_WIDTH, act(5), PN_WIDTH);
            seg2(g, enc, PN_ENC_DIM, params + L.w[5] + PN_WIDTH, ld5, PN_ENC_DIM);
        } else {
            g = nt(M, PN_WIDTH, act(l - 1), PN_WIDTH, params + L.w[l], PN_WIDTH, PN_WIDTH, act(l), PN_WIDTH);
        }
        g.bias = params + L.b[l];
        g.mask_out = masks + (int64_t)l * Mp * PN_MASK_WORDS;
        g.flags = PN_EPI_BIAS | PN_EPI_RELU | PN_EPI_MASKOUT;
        RUN(pn_launch_gemm_nt(g, s));
    }
    // density head
    if (nc == 5) RUN((head_fwd<4, 5>(M, act(7), PN_WIDTH, params + L.wd, params + L.bd, raw_density, 5, s)));
    else RUN((head_fwd<4, 1>(M, act(7), PN_WIDTH, params + L.wd, params + L.bd, raw_density, 1, s)));
    // bottleneck (no activation)
    {
        PnGemmNt g = nt(M, PN_WIDTH, act(7), PN_WIDTH, params + L.we, PN_WIDTH, PN_WIDTH, act(8), PN_WIDTH);
        g.bias = params + L.be;
        g.flags = PN_EPI_BIAS;
        RUN(pn_launch_gemm_nt(g, s));
    }
    // view layer: relu(bott * Wv[:, :256]^T + (bv + viewenc * Wv[:, 256:]^T)[ray])
    {
        PnGemmNt g = nt(M, PN_WIDTH_COND, act(8), PN_WIDTH, wpack + P.wvm, PN_WIDTH, PN_WIDTH, act(9), PN_WIDTH);
        g.rowbias = viewbias;
        g.ldrb = PN_WIDTH_COND;
        g.rows_per_ray = rows_per_ray;
        g.rb_mod = view_rows;
        g.mask_out = masks + (int64_t)8 * Mp * PN_MASK_WORDS;
        g.flags = PN_EPI_ROWBIAS | PN_EPI_RELU | PN_EPI_MASKOUT;
        RUN(pn_launch_gemm_nt(g, s));
    }
    RUN((head_fwd<2, 3>(M, act(9), PN_WIDTH, params + L.wc, params + L.bc, raw_rgb, 3, s)));
    return PN_OK;
}

int pn_density_grad(int64_t M, int nc, float density_bias, const float* params, const float* wpack, const float* mean,
                    const float* cov, const float* acts, const uint32_t* masks, const float* raw_density, float* rsweep,
                    float* scratch, float* grad_mean, void* stream) {
    if (M <= 0) return PN_ERR_BAD_SHAPE;
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    if (!params || !wpack || !mean || !cov || !acts || !masks || !raw_density || !rsweep || !scratch || !grad_mean)
        return PN_ERR_NULL;
    hipStream_t s = ST(stream);
    PnLayout L = pn_layout(nc);
    PnPack P = pn_pack_layout();
    const int64_t Mp = pn_pad(M);
    auto act = [&](int i) { return acts + (int64_t)i * Mp * PN_WIDTH; };
    auto rs = [&](int i) { return rsweep + (int64_t)i * Mp * PN_WIDTH; };
    hipLaunchKernelGGL(k_dgrad_seed, dim3(nblk(M * (PN_WIDTH / 4), 256)), dim3(256), 0, s, M, nc, density_bias,
                       raw_density, params + L.wd, masks + (int64_t)7 * Mp * PN_MASK_WORDS, rs(7));
    PN_CHECK_LAUNCH();
    for (int l = 7; l >= 1; --l) {  // r_{l-1} = [h_{l-1} > 0] * (r_l * W_l[:, :256])
        PnGemmNt g = nt(M, PN_WIDTH, rs(l), PN_WIDTH, wpack + P.wt[l], PN_WIDTH, PN_WIDTH, rs(l - 1), PN_WIDTH);
        g.gate_bits = masks + (int64_t)(l - 1) * Mp * PN_MASK_WORDS;
        g.flags = PN_EPI_GATEBITS;
        RUN(pn_launch_gemm_nt(g, s));
    }
    {  // d sigma / d enc = r_0 * W_0 + r_5 * W_5[:, 256:]
        PnGemmNt g = nt(M, PN_ENC_DIM, rs(0), PN_WIDTH, wpack + P.wt[0], PN_WIDTH, PN_WIDTH, scratch, PN_ENC_DIM);
        seg2(g, rs(5), PN_WIDTH, wpack + P.w5e_t, PN_WIDTH, PN_WIDTH);
        RUN(pn_launch_gemm_nt(g, s));
    }
    RUN(pn_launch_ipe_backward(M, mean, cov, scratch, grad_mean, s));
    return PN_OK;
}

// events for the fork/join between the data-gradient chain (main stream) and the weight-gradient work (side
// stream); a small process-wide pool, re-recorded every call (graph-capturable fork/join pattern)
static std::vector<hipEvent_t> g_events;
static std::mutex g_events_mu;
struct EventRing {
    size_t next = 0;
    hipEvent_t get() {
        std::lock_guard<std::mutex> lock(g_events_mu);
        if (next == g_events.size()) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            g_events.push_back(e);
        }
        return g_events[next++];
    }
};

// layout of the first part of the pn_mlp_backward workspace: the buffers a LATER call reads when the weight
// gradients of this evaluation are deferred (batched with other evaluations of the same step)
struct BwdKeep {
    float *dbuf, *tbuf, *dbott, *edot;
    int64_t Mp;
    float* delta(int l) const { return dbuf + (int64_t)l * Mp * PN_WIDTH; }  // l = 0..7; 8 = head addend
    float* tang(int l) const { return tbuf + (int64_t)l * Mp * PN_WIDTH; }
};
static BwdKeep carve_keep(float* work, int64_t Mp, float** rest) {
    BwdKeep k;
    k.Mp = Mp;
    float* w = work;
    k.dbuf = w; w += 9 * Mp * PN_WIDTH;
    k.tbuf = w; w += 8 * Mp * PN_WIDTH;
    k.dbott = w; w += Mp * PN_WIDTH;
    k.edot = w; w += Mp * PN_ENC_DIM;
    if (rest) *rest = w;
    return k;
}

// split-K slabs shared by every weight-gradient GEMM of one call: the largest of the shapes launched
static int64_t slab_floats(int64_t rows) {
    const int shapes[4][2] = {{PN_WIDTH, PN_WIDTH}, {PN_WIDTH, PN_ENC_DIM}, {PN_WIDTH_COND, PN_WIDTH}, {PN_WIDTH_COND, 32}};
    int64_t n = 0;
    for (auto& sh : shapes) {
        const int64_t f = pn_tn_work_floats(rows, sh[0], sh[1]);
        if (f > n) n = f;
    }
    return n;
}

// per-ray sums of the view-layer gradient: S [R][128], folded S2 [view_rows][128], reduction scratch
static int64_t view_sum_floats(int64_t M, int rows_per_ray, int64_t view_rows) {
    const int64_t R = M / rows_per_ray;
    int64_t n = R * PN_WIDTH_COND + view_rows * PN_WIDTH_COND + 64 * PN_WIDTH_COND + view_rows * 32 + PN_WIDTH_COND * 32;
    if (R > view_rows && view_rows <= 1024) n += 64 * view_rows * PN_WIDTH_COND;
    return (n + 3) & ~(int64_t)3;
}

int64_t pn_mlp_backward_work_floats(int64_t M, int rows_per_ray, int64_t view_rows, int64_t M_batched) {
    if (M <= 0 || rows_per_ray <= 0 || view_rows <= 0 || M % rows_per_ray) return PN_ERR_BAD_SHAPE;
    const int64_t Mp = pn_pad(M);
    if (M_batched < 2 * Mp) M_batched = 2 * Mp;
    int64_t n = 0;
    n += 9 * Mp * PN_WIDTH;       // delta_0..delta_7 (kept: the weight-gradient stream reads them) + head addend
    n += 8 * Mp * PN_WIDTH;       // tangent hdot_0..hdot_7
    n += Mp * PN_WIDTH;           // d_bott
    n += Mp * PN_ENC_DIM;         // edot            (up to here: carve_keep)
    n += Mp * PN_WIDTH_COND;      // d view hidden
    n += Mp * PN_ENC_DIM;         // d_enc
    n += view_sum_floats(M, rows_per_ray, view_rows);
    n += Mp * 8 + Mp * 2;         // dden copy, sdot, coef
    n += slab_floats(M_batched + 4 * PN_ROW_PAD);  // slabs
    n += 9 * (Mp / 64) * PN_WIDTH + 64 * 9 * PN_WIDTH + 9 * PN_WIDTH;  // epilogue column sums + reduce scratch
    int64_t nb = (M + HEAD_ROWS - 1) / HEAD_ROWS;
    int64_t hp = nb * (5 * PN_WIDTH + 5) + 64 * (5 * PN_WIDTH + 5);
    n += hp + 64;
    return n;
}

int pn_mlp_backward(int64_t M, int rows_per_ray, int64_t view_rows, int nc, float density_bias, const float* params,
                    const float* wpack, const float* mean, const float* cov, const float* enc, const float* viewenc,
                    const float* acts, const uint32_t* masks, const float* raw_density, const float* d_raw_rgb,
                    const float* d_raw_density, const float* rsweep, const float* v_gradmean, float* d_mean,
                    float* grads, float* work, int64_t M_batched, int defer_wgrad, int n_deferred,
                    const int64_t* dM_host, const float* const* denc_host, const float* const* dacts_host,
                    const float* const* drsweep_host, float* const* dwork_host, const int* dtangent_host, void* stream,
                    void* side_stream) {
    if (M <= 0 || rows_per_ray <= 0 || view_rows <= 0) return PN_ERR_BAD_SHAPE;
    if (M % rows_per_ray || (M / rows_per_ray) % view_rows) return PN_ERR_BAD_SHAPE;  // whole rays, cycling view rows
    if (n_deferred < 0 || n_deferred > 2 || (defer_wgrad && n_deferred)) return PN_ERR_BAD_SHAPE;
    if (n_deferred && (!dM_host || !denc_host || !dacts_host || !drsweep_host || !dwork_host || !dtangent_host))
        return PN_ERR_NULL;
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    if (!params || !wpack || !mean || !cov || !enc || !viewenc || !acts || !masks || !raw_density || !d_raw_rgb ||
        !d_raw_density || !grads || !work)
        return PN_ERR_NULL;
    if (v_gradmean && !rsweep) return PN_ERR_NULL;
    hipStream_t s = ST(stream);
    // ws: the stream every "accumulate into grads" kernel runs on (in order, so they never race each other).
    // With a side stream the weight-gradient GEMMs fill the MFMA slots the data-gradient chain leaves idle.
    hipStream_t ws = side_stream ? ST(side_stream) : s;
    const bool forked = ws != s;
    EventRing ring;
    auto hand_off = [&]() -> int {  // everything enqueued on `s` so far is visible to later work on `ws`
        if (!forked) return PN_OK;
        hipEvent_t e = ring.get();
        if (!e || hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(ws, e, 0) != hipSuccess) return PN_ERR_HIP;
        return PN_OK;
    };
    PnLayout L = pn_layout(nc);
    PnPack P = pn_pack_layout();
    const int64_t Mp = pn_pad(M);
    auto act = [&](int i) { return acts + (int64_t)i * Mp * PN_WIDTH; };
    auto rs = [&](int i) { return rsweep + (int64_t)i * Mp * PN_WIDTH; };
    auto mask = [&](int i) { return masks + (int64_t)i * Mp * PN_MASK_WORDS; };
    // carve the workspace (every piece is a multiple of 4 floats -> 16-B aligned if `work` is)
    float* w = nullptr;
    const BwdKeep keep = carve_keep(work, Mp, &w);
    auto delta = [&](int l) { return keep.delta(l); };
    auto tang = [&](int l) { return keep.tang(l); };
    float* dbott = keep.dbott;
    float* edot = keep.edot;
    float* dvh = w; w += Mp * PN_WIDTH_COND;
    float* denc = w; w += Mp * PN_ENC_DIM;
    const int64_t R = M / rows_per_ray;  // rays of this evaluation (view encoding is constant along a ray)
    float* S = w;
    float* S2buf = S + R * PN_WIDTH_COND;
    float* bias_scratch = S2buf + view_rows * PN_WIDTH_COND;
    float* VE = bias_scratch + 64 * PN_WIDTH_COND;      // [view_rows][32]
    float* vw_tmp = VE + view_rows * 32;                // [128][32]
    float* fold_scratch = (R > view_rows && view_rows <= 1024) ? vw_tmp + PN_WIDTH_COND * 32 : nullptr;
    w += view_sum_floats(M, rows_per_ray, view_rows);
    float* dden = w; w += Mp * 8;
    float* sdot = w; w += Mp;
    float* coef = w; w += Mp;
    if (M_batched < 2 * Mp) M_batched = 2 * Mp;
    const int64_t slab_avail = slab_floats(M_batched + 4 * PN_ROW_PAD);
    float* slab = w; w += slab_avail;
    float* csum = w; w += 9 * (Mp / 64) * PN_WIDTH;   // [9 biases][row partials][256] from the GEMM epilogues
    float* csum_scratch = w; w += 64 * 9 * PN_WIDTH + 9 * PN_WIDTH;
    float* partial = w;
    const int64_t csum_rows = 2 * ((M + 127) / 128);
    auto csum_slot = [&](int slot) { return csum + (int64_t)slot * csum_rows * PN_WIDTH; };
    const int ld5 = PN_WIDTH + PN_ENC_DIM;
    const int ldv = PN_WIDTH + PN_VIEW_DIM;
    // Weight gradients of the trunk and the extra layer: either deferred (operands stay in `work` for a later call)
    // or computed here over the rows of THIS evaluation plus the deferred ones — one TN GEMM per layer over up to
    // four row segments (own, env, level-1 first-order, level-1 tangent) instead of one per evaluation.
    BwdKeep dk[2];
    int64_t dMp[2] = {0, 0};
    for (int e = 0; e < n_deferred; ++e) {
        if (dM_host[e] <= 0 || !denc_host[e] || !dacts_host[e] || !dwork_host[e]) return PN_ERR_NULL;
        if (dtangent_host[e] && !drsweep_host[e]) return PN_ERR_NULL;
        dMp[e] = pn_pad(dM_host[e]);
        dk[e] = carve_keep(dwork_host[e], dMp[e], nullptr);
    }
    // layer l in 0..7: dW_l (+ skip columns for l == 5); l == 8: extra layer
    auto trunk_wgrad = [&](int l, bool own_first_order, bool own_tangent) -> int {
        if (defer_wgrad) return PN_OK;
        const int nparts = (l == 5) ? 2 : 1;
        for (int part = 0; part < nparts; ++part) {
            PnSegTn segs[4];
            int ns = 0;
            bool overflow = false;
            const bool skip = part == 1;                 // layer 5, columns 256..351 (against enc / edot)
            const bool from_enc = skip || l == 0;
            const int ldy = from_enc ? PN_ENC_DIM : PN_WIDTH;
            auto add = [&](const float* X, const float* Y, int64_t rows) {
                if (ns < 4) segs[ns++] = PnSegTn{X, Y, PN_WIDTH, ldy, rows};
                else overflow = true;
            };
            auto acts_of = [&](const float* a, int64_t mp, int i) { return a + (int64_t)i * mp * PN_WIDTH; };
            if (l == 8) {
                if (own_first_order) add(dbott, act(7), M);
                for (int e = 0; e < n_deferred; ++e) add(dk[e].dbott, acts_of(dacts_host[e], dMp[e], 7), dM_host[e]);
            } else {
                if (own_first_order) add(delta(l), from_enc ? enc : act(l - 1), M);
                if (own_tangent) add(rs(l), from_enc ? edot : tang(l - 1), M);
                for (int e = 0; e < n_deferred; ++e) {
                    add(dk[e].delta(l), from_enc ? denc_host[e] : acts_of(dacts_host[e], dMp[e], l - 1), dM_host[e]);
                    if (dtangent_host[e])
                        add(drsweep_host[e] + (int64_t)l * dMp[e] * PN_WIDTH, from_enc ? dk[e].edot : dk[e].tang(l - 1),
                            dM_host[e]);
                }
            }
            if (overflow) return PN_ERR_UNSUPPORTED;  // more than four row segments
            if (ns == 0) continue;
            float* dst = (l == 8) ? grads + L.we : grads + L.w[l] + (skip ? PN_WIDTH : 0);
            const int n2 = from_enc ? PN_ENC_DIM : PN_WIDTH;
            const int ldw = (l == 8) ? PN_WIDTH : (l == 0 ? PN_ENC_DIM : (l == 5 ? ld5 : PN_WIDTH));
            RUN(pn_launch_gemm_tn(segs, ns, PN_WIDTH, n2, dst, ldw, 1, slab, slab_avail, ws));
        }
        return PN_OK;
    };
    const bool inline_tangent = v_gradmean && !defer_wgrad && n_deferred == 0;  // stand-alone call: as before

    RUN(hand_off());  // inputs produced earlier on the main stream
    // colour-head weight gradient only needs inputs: start the side stream with it
    RUN((head_bwd_weight<2, 3>(M, d_raw_rgb, 3, nullptr, act(9), PN_WIDTH, grads + L.wc, grads + L.bc, partial, ws)));

    const float* dden_use = d_raw_density;
    // ---------------- second-order path: tangent sweep (main) + its weight gradients (ws) -----------
    if (v_gradmean) {
        RUN(pn_launch_ipe_tangent(M, mean, cov, v_gradmean, edot, s));
        RUN(hand_off());
        const float* prev = edot;
        int prev_ld = PN_ENC_DIM;
        for (int l = 0; l < 8; ++l) {
            // dW_l += r_l^T * hdot_{l-1}: stand-alone calls do it here, otherwise it rides in the batched trunk GEMMs
            if (inline_tangent) RUN(trunk_wgrad(l, false, true));
            // hdot_l = gate_l * (hdot_{l-1} * W_l^T)
            float* cur = tang(l);
            PnGemmNt g;
            if (l == 0) {
                g = nt(M, PN_WIDTH, edot, PN_ENC_DIM, params + L.w[0], PN_ENC_DIM, PN_ENC_DIM, cur, PN_WIDTH);
            } else if (l == 5) {
                g = nt(M, PN_WIDTH, prev, PN_WIDTH, params + L.w[5], ld5, PN_WIDTH, cur, PN_WIDTH);
                seg2(g, edot, PN_ENC_DIM, params + L.w[5] + PN_WIDTH, ld5, PN_ENC_DIM);
            } else {
                g = nt(M, PN_WIDTH, prev, PN_WIDTH, params + L.w[l], PN_WIDTH, PN_WIDTH, cur, PN_WIDTH);
            }
            g.gate_bits = mask(l);
            g.flags = PN_EPI_GATEBITS;
            RUN(pn_launch_gemm_nt(g, s));
            RUN(hand_off());
            prev = cur;
            prev_ld = PN_WIDTH;
        }
        // sigma_dot_raw = Wd[0] . hdot_7 ;  dWd[0] += sum softplus'(z) * hdot_7
        RUN((head_fwd<4, 1>(M, prev, PN_WIDTH, params + L.wd, nullptr, sdot, 1, s)));
        hipLaunchKernelGGL(k_second_order_seed, dim3(nblk(M * nc, 256)), dim3(256), 0, s, M, nc, density_bias,
                           raw_density, sdot, d_raw_density, dden, coef);
        PN_CHECK_LAUNCH();
        RUN(hand_off());
        dden_use = dden;
        RUN((head_bwd_weight<4, 1>(M, coef, 1, nullptr, prev, PN_WIDTH, grads + L.wd, nullptr, partial, ws)));
    }

    // ---------------- colour head + view layer ------------------------------------------------
    RUN((head_bwd_data<2, 3>(M, d_raw_rgb, 3, params + L.wc, dvh, PN_WIDTH_COND, act(9), PN_WIDTH, s)));
    RUN(hand_off());
    RUN(wgrad(M, dvh, PN_WIDTH_COND, PN_WIDTH_COND, act(8), PN_WIDTH, PN_WIDTH, grads + L.wv, ldv, slab, slab_avail, ws));
    {  // bias and view-encoding columns of the view layer from per-ray sums of dvh (no per-sample expansion)
        unsigned gr = (unsigned)((R + 3) / 4);
        if (gr > 2048) gr = 2048;
        hipLaunchKernelGGL((k_ray_colsum<PN_WIDTH_COND>), dim3(gr), dim3(256), 0, ws, R, rows_per_ray, dvh, PN_WIDTH_COND, S);
        PN_CHECK_LAUNCH();
        const float* S2 = S;
        if (R > view_rows) {  // rays cycle through the view rows (env light: ray b*D + d looks along direction d)
            RUN(pn_launch_reduce_rows(S, R / view_rows, view_rows * PN_WIDTH_COND, (int)view_rows, PN_WIDTH_COND,
                                      PN_WIDTH_COND, S2buf, PN_WIDTH_COND, 0, fold_scratch, ws));
            S2 = S2buf;
        }
        hipLaunchKernelGGL(k_pad_viewenc, dim3(nblk(view_rows * 32, 256)), dim3(256), 0, ws, view_rows, viewenc, VE);
        PN_CHECK_LAUNCH();
        {  // [128][32] product over the rays; only the first 27 columns exist in the parameter: land it in scratch, then add
            PnSegTn sg{S2, VE, PN_WIDTH_COND, 32, view_rows};
            RUN(pn_launch_gemm_tn(&sg, 1, PN_WIDTH_COND, 32, vw_tmp, 32, 0, slab, slab_avail, ws));
            RUN(pn_launch_reduce_rows(vw_tmp, 1, 0, PN_WIDTH_COND, PN_VIEW_DIM, 32, grads + L.wv + PN_WIDTH, ldv, 1, nullptr, ws));
        }
        RUN(pn_launch_reduce_rows(S2, view_rows, PN_WIDTH_COND, 1, PN_WIDTH_COND, PN_WIDTH_COND, grads + L.bv,
                                  PN_WIDTH_COND, 1, bias_scratch, ws));
    }
    {  // d bottleneck = dvh * Wv[:, :256]
        PnGemmNt g = nt(M, PN_WIDTH, dvh, PN_WIDTH_COND, wpack + P.wvm_t, PN_WIDTH_COND, PN_WIDTH_COND, dbott, PN_WIDTH);
        g.colsum = csum_slot(0);
        g.flags = PN_EPI_COLSUM;
        RUN(pn_launch_gemm_nt(g, s));
    }
    // ---------------- density head ----------------------------------------------------------
    if (nc == 5) RUN((head_bwd_data<4, 5>(M, dden_use, 5, params + L.wd, delta(8), PN_WIDTH, nullptr, 0, s)));
    else RUN((head_bwd_data<4, 1>(M, dden_use, 1, params + L.wd, delta(8), PN_WIDTH, nullptr, 0, s)));
    RUN(hand_off());
    RUN(trunk_wgrad(8, true, false));
    if (nc == 5) RUN((head_bwd_weight<4, 5>(M, dden_use, 5, nullptr, act(7), PN_WIDTH, grads + L.wd, grads + L.bd, partial, ws)));
    else RUN((head_bwd_weight<4, 1>(M, dden_use, 1, nullptr, act(7), PN_WIDTH, grads + L.wd, grads + L.bd, partial, ws)));
    {  // delta_7 = gate_7 * (d_bott * We + d_raw_density * Wd)
        PnGemmNt g = nt(M, PN_WIDTH, dbott, PN_WIDTH, wpack + P.we_t, PN_WIDTH, PN_WIDTH, delta(7), PN_WIDTH);
        g.addc = delta(8);
        g.ldadd = PN_WIDTH;
        g.gate_bits = mask(7);
        g.colsum = csum_slot(1);
        g.flags = PN_EPI_ADDC | PN_EPI_GATEBITS | PN_EPI_COLSUM;
        RUN(pn_launch_gemm_nt(g, s));
    }
    // ---------------- trunk ---------------------------------------------------------------------
    for (int l = 7; l >= 0; --l) {
        RUN(hand_off());  // delta_l is complete on the main stream
        RUN(trunk_wgrad(l, true, v_gradmean && !inline_tangent));
        if (l > 0) {
            PnGemmNt g = nt(M, PN_WIDTH, delta(l), PN_WIDTH, wpack + P.wt[l], PN_WIDTH, PN_WIDTH, delta(l - 1), PN_WIDTH);
            g.gate_bits = mask(l - 1);
            g.colsum = csum_slot(1 + (7 - (l - 1)));
            g.flags = PN_EPI_GATEBITS | PN_EPI_COLSUM;
            RUN(pn_launch_gemm_nt(g, s));
        }
    }
    {  // bias gradients: one reduction over the nine column-sum slots (slot 0: extra, slot 1 + (7 - l): layer l)
        RUN(hand_off());
        float* tmp = csum_scratch + 64 * 9 * PN_WIDTH;
        RUN(pn_launch_reduce_rows(csum, csum_rows, PN_WIDTH, 9, PN_WIDTH, (int)(csum_rows * PN_WIDTH), tmp, PN_WIDTH, 0,
                                  csum_scratch, ws));
        BiasOffsets bo;
        bo.off[0] = L.be;
        for (int l = 0; l < 8; ++l) bo.off[1 + (7 - l)] = L.b[l];
        hipLaunchKernelGGL(k_bias_scatter, dim3(9), dim3(256), 0, ws, tmp, bo, grads);
        PN_CHECK_LAUNCH();
    }
    if (d_mean) {  // d enc = delta_0 * W_0 + delta_5 * W_5[:, 256:]  ->  d mean
        PnGemmNt g = nt(M, PN_ENC_DIM, delta(0), PN_WIDTH, wpack + P.wt[0], PN_WIDTH, PN_WIDTH, denc, PN_ENC_DIM);
        seg2(g, delta(5), PN_WIDTH, wpack + P.w5e_t, PN_WIDTH, PN_WIDTH);
        RUN(pn_launch_gemm_nt(g, s));
        RUN(pn_launch_ipe_backward(M, mean, cov, denc, d_mean, s));
    }
    if (forked) {  // join: later work on the main stream sees the finished gradients
        hipEvent_t e = ring.get();
        if (!e || hipEventRecord(e, ws) != hipSuccess || hipStreamWaitEvent(s, e, 0) != hipSuccess) return PN_ERR_HIP;
    }
    return PN_OK;
}

int pn_gemm_nt(int64_t M, int N, int K, const float* A, int lda, const float* Bt, int ldb, float* C, int ldc,
               const float* bias, const float* gate, int ldg, int flags, void* stream) {
    PnGemmNt g = nt(M, N, A, lda, Bt, ldb, K, C, ldc);
    g.bias = bias;
    g.gate = gate;
    g.ldg = ldg;
    #ifdef PN_ABLATE
    g.flags = flags & (PN_EPI_BIAS | PN_EPI_RELU | PN_EPI_GATE | 0x300);  // 0x100/0x200: ablation (tools/experiments/bench_gemm.py)
#else
    g.flags = flags & (PN_EPI_BIAS | PN_EPI_RELU | PN_EPI_GATE);
#endif
    return pn_launch_gemm_nt(g, ST(stream));
}

int64_t pn_gemm_tn_work_floats(int64_t M, int N1, int N2) { return pn_tn_work_floats(M, N1, N2); }

int pn_gemm_tn(int64_t M, int N1, int N2, const float* X, int ldx, const float* Y, int ldy, float* C, int ldc,
               int accumulate, float* work, void* stream) {
    PnSegTn sg{X, Y, ldx, ldy, M};
    return pn_launch_gemm_tn(&sg, 1, N1, N2, C, ldc, accumulate, work, -1, ST(stream));
}

const char* pn_strerror(int code) {
    switch (code) {
        case PN_OK: return "ok";
        case PN_ERR_BAD_SHAPE: return "bad shape / alignment";
        case PN_ERR_UNSUPPORTED: return "unsupported configuration";
        case PN_ERR_NULL: return "required pointer is null";
        case PN_ERR_HIP: return "HIP launch failed";
        default: return "unknown error";
    }
}
int pn_abi_version(void) { return 2; }
int64_t pn_pad_rows(int64_t m) { return pn_pad(m); }

}  // extern "C"
