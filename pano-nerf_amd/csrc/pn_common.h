// pn_common.h — shared declarations for the gfx950 kernels of libpanonerf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/panonerf_hip.h"

#define PN_WAVE 64

#define PN_CHECK_LAUNCH()                                  \
    do {                                                   \
        if (hipGetLastError() != hipSuccess) return PN_ERR_HIP; \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- GEMM (pn_gemm.hip) -------------------------------------------------------------
struct PnSeg {
    const float* A;   // [M, K] row-major, leading dim lda
    const float* B;   // [N, K] row-major, leading dim ldb   (C = A * B^T)
    int lda, ldb, K;  // K % 4 == 0, lda % 4 == 0, ldb % 4 == 0, 16-byte aligned bases
};
enum {
    PN_EPI_BIAS = 1, PN_EPI_RELU = 2, PN_EPI_GATE = 4, PN_EPI_ROWBIAS = 8, PN_EPI_ADDC = 16,
    PN_EPI_GATEBITS = 32,  // gate by the bit mask written by an earlier PN_EPI_MASKOUT launch
    PN_EPI_MASKOUT = 64,   // write (value > 0) bits, one u32 per (row, 32 columns)
    PN_EPI_COLSUM = 128    // write per-wave column sums of the stored tile (bias gradients)
};
// bit (c*8 + i) of mask word [row][col/32] <-> column 32*(col/32) + 4*i + c   (i = 0..7, c = 0..3)
// PN_MASK_WORDS (8 u32 per row) / PN_MASK_SLOTS (h0..h7, view hidden) come from panonerf_hip.h
struct PnGemmNt {
    PnSeg seg[2];
    int nseg;
    float* C;
    int ldc;
    int64_t M;
    int N;
    const float* bias;     // [N]
    const float* rowbias;  // [rays, ldrb]; ray = (row / rows_per_ray) % rb_mod
    int ldrb, rows_per_ray;
    int64_t rb_mod;
    const float* addc;  // [M, N] addend, leading dim ldadd
    int ldadd;
    const float* gate;  // [M, N] gate source (> 0 passes), leading dim ldg
    int ldg;
    const uint32_t* gate_bits;  // [M, PN_MASK_WORDS]
    uint32_t* mask_out;         // [M, PN_MASK_WORDS]
    float* colsum;              // [2 * ceil(M/128), N] per-wave-row column sums (PN_EPI_COLSUM)
    int flags;
};
// dst[r*ldd + c] (+)= sum_b src[b*stride + r*src_ld + c]  for r < rows, c < cols; scratch >= 64*rows*cols floats
int pn_launch_reduce_rows(const float* src, int64_t nb, int64_t stride, int rows, int cols, int src_ld, float* dst,
                          int ldd, int accumulate, float* scratch, hipStream_t s);
int pn_launch_gemm_nt(const PnGemmNt& g, hipStream_t s);

struct PnSegTn {
    const float* X;  // [M, N1] leading dim ldx
    const float* Y;  // [M, N2] leading dim ldy
    int ldx, ldy;
    int64_t M;
};
// C[N1, N2] (ldc) (+)= sum_seg X^T Y ; work >= pn_tn_work_floats(...)
int64_t pn_tn_work_floats(int64_t Mtotal, int N1, int N2);
// work_avail: floats available at `work` (< 0: unknown, trust the caller); an undersized slab is refused on the host
int pn_launch_gemm_tn(const PnSegTn* segs, int nseg, int N1, int N2, float* C, int ldc, int accumulate, float* work,
                      int64_t work_avail, hipStream_t s);

// launch timing (pn_prof_enable / pn_prof_read): bracket a launch with HIP events on its stream.  Classes: 0 k_gemm_nt,
// 1 k_gemm_tn, 2 k_chain_fwd, 3 k_chain_dgrad, 4 k_chain_tangent, 5 k_chain_bwd, 6..10 k_chain_wgrad (one per tile
// configuration: 256x256, 256x96, 128x288, 32x256, 32x128)
struct PnProfScope {
    void* impl;
    PnProfScope(int cls, double flops, hipStream_t s);
    ~PnProfScope();
};

// ---- encodings (pn_render.hip) ---------------------------------------------------------
int pn_launch_ipe_backward(int64_t M, const float* mean, const float* cov, const float* d_enc, float* d_mean,
                           hipStream_t s);
int pn_launch_ipe_tangent(int64_t M, const float* mean, const float* cov, const float* v, float* edot, hipStream_t s);

// ---- parameter block layout (pn_api.hip) -------------------------------------------------
struct PnLayout {
    int nc;
    int64_t w[8], b[8];  // trunk
    int64_t we, be, wv, bv, wd, wc, bd, bc;
    int64_t total;
};
PnLayout pn_layout(int nc);
// packed weights workspace offsets (floats)
struct PnPack {
    int64_t wt[8];   // W_l^T for l = 0..7 ; l = 0: [96][256]; l = 5: hidden part [256][256]
    int64_t w5e_t;   // W5[:,256:352]^T  [96][256]
    int64_t we_t;    // extra^T [256][256]
    int64_t wvm;     // view W[:, :256]   [128][256]
    int64_t wvm_t;   // its transpose     [256][128]
    int64_t wvv;     // view W[:, 256:283] padded [128][32]
    int64_t total;
};
PnPack pn_pack_layout();

static inline int64_t pn_pad(int64_t m) { return (m + PN_ROW_PAD - 1) / PN_ROW_PAD * PN_ROW_PAD; }
