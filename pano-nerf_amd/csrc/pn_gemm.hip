// pn_gemm.hip — exact-fp32 MFMA GEMMs for the radiance MLP (gfx950).
//
//   NT : C[M,N] = epi( sum_seg A_seg[M,K] * B_seg[N,K]^T )      forward layers, data-gradient,
//        density-gradient sweep and tangent sweep (all with the weights in [N][K] form:
//        the transposes are pre-packed once per step by pn_pack_weights)
//   TN : C[N1,N2] (+)= sum_rows X[r,N1] * Y[r,N2]               weight gradients, split over rows
//
// Both use v_mfma_f32_32x32x2_f32 (bit-exact fp32 fma chains, 64 FLOP/clk/SIMD = 157 TF chip peak).
// Block tile 128x128, 4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).
// K is consumed in chunks of 32 staged through LDS; the next chunk's global loads are issued
// before the current chunk's MFMAs (register prefetch).
//
// Operand mapping of v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31].  The k index inside an instruction is only a label, so for the NT
// form each lane reads FOUR consecutive k of its row with one ds_read_b128 (lanes 0-31: k = 8j..8j+3,
// lanes 32-63: k = 8j+4..8j+7) and feeds them to four consecutive MFMAs.
#include "pn_common.h"
#include <stdlib.h>
#include <mutex>
#include <vector>

// ---------------------------------------------------------------------------- launch timing
// Optional (off by default): bracket every GEMM launch with HIP events on the launch stream so that
// bench.py can report the dominant kernel's average duration and algorithmic FLOP rate live.
struct ProfRec {
    int cls;
    hipEvent_t e0, e1;
    double flops;
};
static bool g_prof_on = false;
// Ablation switches of the tools/ micro-benchmarks (skip the stores / loads of a GEMM, force a kernel variant) exist only
// in -DPN_ABLATE builds (PN_EXTRA=-DPN_ABLATE csrc/build.sh): pn_prof_enable(on | dbg << 8).  The shipped library has none.
#ifdef PN_ABLATE
static int g_dbg = 0;
#define PN_DBG g_dbg
#define PN_ABL(x) (x)
#else
#define PN_DBG 0
#define PN_ABL(x) 0
#endif
// (autograd runs the backward of different devices on different host threads: the bookkeeping below is locked)
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_free_events;
#define PN_PROF_CLASSES 14
static double g_prof_ms[PN_PROF_CLASSES] = {0}, g_prof_flops[PN_PROF_CLASSES] = {0};
static int64_t g_prof_n[PN_PROF_CLASSES] = {0};

static hipEvent_t prof_event() {
    if (!g_free_events.empty()) {
        hipEvent_t e = g_free_events.back();
        g_free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
struct ProfScope {
    ProfRec r;
    hipStream_t s;
    bool live;
    ProfScope(int cls, double flops, hipStream_t s_) : s(s_), live(false) {
        if (!g_prof_on) return;
        std::lock_guard<std::mutex> lock(g_prof_mu);
        if (g_prof.size() > (1u << 20)) return;
        r.cls = cls;
        r.flops = flops;
        r.e0 = prof_event();
        r.e1 = prof_event();
        if (!r.e0 || !r.e1) return;
        live = hipEventRecord(r.e0, s) == hipSuccess;
    }
    ~ProfScope() {
        if (!live) return;
        std::lock_guard<std::mutex> lock(g_prof_mu);
        if (hipEventRecord(r.e1, s) == hipSuccess) g_prof.push_back(r);
    }
};
// the same bracket for the kernels of other translation units (pn_chain.hip)
PnProfScope::PnProfScope(int cls, double flops, hipStream_t s) : impl(new ProfScope(cls, flops, s)) {}
PnProfScope::~PnProfScope() { delete static_cast<ProfScope*>(impl); }
static void prof_drain() {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            g_prof_ms[r.cls] += ms;
            g_prof_flops[r.cls] += r.flops;
            g_prof_n[r.cls] += 1;
        }
        g_free_events.push_back(r.e0);
        g_free_events.push_back(r.e1);
    }
    g_prof.clear();
}
extern "C" int pn_prof_enable(int on) {
    prof_drain();
    g_prof_on = (on & 1) != 0;
#ifdef PN_ABLATE
    g_dbg = on >> 8;
#endif
    for (int i = 0; i < PN_PROF_CLASSES; ++i) {
        g_prof_ms[i] = 0;
        g_prof_flops[i] = 0;
        g_prof_n[i] = 0;
    }
    return PN_OK;
}
extern "C" int pn_prof_read(int cls, double* total_ms, int64_t* launches, double* flops) {
    if (cls < 0 || cls >= PN_PROF_CLASSES) return PN_ERR_BAD_SHAPE;
    prof_drain();
    if (total_ms) *total_ms = g_prof_ms[cls];
    if (launches) *launches = g_prof_n[cls];
    if (flops) *flops = g_prof_flops[cls];
    return PN_OK;
}

#define BM 128
#define BN 128
#define BK 32
#define LDT (BK + 4)   // NT tiles: [128][36] floats; +4 keeps 16-B alignment and is conflict-free for b128
#define LDX (BM + 4)   // TN tiles: [32][132] floats
#define EPL 68         // epilogue transposition rows: 64 + 4 floats
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // blocks b and b+8 share an XCD: give each XCD a contiguous range of tiles (bijective form).
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// ------------------------------------------------------------------------------------- NT
struct NtRegs {
    f32x4 a[4], b[4];
};

// Branch-free staging loads: addresses are clamped into the operand (always legal to read) and the
// out-of-range lanes are zeroed by a select, so no exec-mask branches and no scalar (kernarg) loads sit
// inside the K loop (an s_waitcnt lgkmcnt(0) for an s_load would also drain the LDS reads).
__device__ __forceinline__ void nt_load(const float* A, int lda, const float* B, int ldb, int K, int64_t M, int N,
                                        int k0, int64_t m0, int n0, int tid, NtRegs& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int idx = tid + 256 * i;
        int row = idx >> 3, c4 = idx & 7;
        int k = k0 + c4 * 4;
        const bool kin = k < K;
        const int kc = kin ? k : K - 4;
        int64_t gr = m0 + row;
        gr = gr < M ? gr : M - 1;
        int gn = n0 + row;
        const bool nin = gn < N;
        gn = nin ? gn : N - 1;
        f32x4 va = *reinterpret_cast<const f32x4*>(A + gr * lda + kc);
        f32x4 vb = *reinterpret_cast<const f32x4*>(B + (int64_t)gn * ldb + kc);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        r.a[i] = kin ? va : z;
        r.b[i] = (kin && nin) ? vb : z;
    }
}

__device__ __forceinline__ void nt_store(float* As, float* Bs, int tid, const NtRegs& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int idx = tid + 256 * i;
        int row = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<f32x4*>(As + row * LDT + c4 * 4) = r.a[i];
        *reinterpret_cast<f32x4*>(Bs + row * LDT + c4 * 4) = r.b[i];
    }
}

// One 32-row slab of the shared epilogue: lanes (lane >> 4) pick one of 4 rows per step, (lane & 15) a float4 of the
// wave's 64 columns; eight steps read the transposed values from the wave's LDS slab, apply bias / per-ray bias /
// addend / ReLU / gate / mask bits / mask emission, store, and add to the running column sums.
// The epilogue's vector instructions only get the issue slots the other workgroups' MFMA streams leave (~25 ns each,
// profiles/r01_nt_phase_trace_K256.txt), so their COUNT is what matters: F >= 0 fixes the flag set at compile time
// (the eight sets the MLP uses are instantiated; F < 0 keeps the run-time flags), INTERIOR drops every bounds test
// for waves whose 64 x 64 block lies inside the matrix, and the row pointers advance by a constant per step.
template <int F, bool INTERIOR>
__device__ __forceinline__ void nt_epi_slab(const PnGemmNt& g, const float* Ls, int rtflags, int64_t row0, int col4, int gcol,
                                            bool col_ok, const f32x4& bias4, f32x4& csum, int lane) {
    const int flags = F < 0 ? rtflags : F;
    const int r0 = lane >> 4;
    int64_t row = row0 + r0;
    float* cp = g.C + row * g.ldc + gcol;
    const int64_t cstep = 4 * (int64_t)g.ldc;
    const float* ap = (flags & PN_EPI_ADDC) ? g.addc + row * g.ldadd + gcol : nullptr;
    const int64_t astep = 4 * (int64_t)g.ldadd;
    const uint32_t* bp = (flags & PN_EPI_GATEBITS) ? g.gate_bits + row * PN_MASK_WORDS + (gcol >> 5) : nullptr;
    uint32_t* mp = (flags & PN_EPI_MASKOUT) ? g.mask_out + row * PN_MASK_WORDS + (gcol >> 5) : nullptr;
    const int bi = (gcol >> 2) & 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool ok = INTERIOR ? true : (col_ok && row < g.M);
        f32x4 v = *reinterpret_cast<const f32x4*>(Ls + (r0 + 4 * i) * EPL + col4);
        if (flags & PN_EPI_BIAS) v += bias4;
        if (ok) {
            if (flags & PN_EPI_ROWBIAS) {
                int64_t ray = row / g.rows_per_ray;
                if (g.rb_mod > 0) ray %= g.rb_mod;
                v += *reinterpret_cast<const f32x4*>(g.rowbias + ray * g.ldrb + gcol);
            }
            if (flags & PN_EPI_ADDC) v += *reinterpret_cast<const f32x4*>(ap);
        }
        if (flags & PN_EPI_RELU) {
            v[0] = fmaxf(v[0], 0.f);
            v[1] = fmaxf(v[1], 0.f);
            v[2] = fmaxf(v[2], 0.f);
            v[3] = fmaxf(v[3], 0.f);
        }
        if ((flags & PN_EPI_GATE) && ok) {
            f32x4 gt = *reinterpret_cast<const f32x4*>(g.gate + row * g.ldg + gcol);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = gt[c] > 0.f ? v[c] : 0.f;
        }
        if ((flags & PN_EPI_GATEBITS) && ok) {
            const uint32_t w = *bp;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = ((w >> (c * 8 + bi)) & 1u) ? v[c] : 0.f;
        }
        if (flags & PN_EPI_MASKOUT) {  // all 64 lanes take part in the ballots
            uint32_t word = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned long long b = __ballot(v[c] > 0.f);
                word |= (uint32_t)((b >> ((lane >> 3) * 8)) & 0xffull) << (c * 8);
            }
            if (ok && (lane & 7) == 0) *mp = word;
        }
        if (ok) {
            if (!PN_ABL(flags & 0x100) || v[0] == 12345.678f) *reinterpret_cast<f32x4*>(cp) = v;
            if (flags & PN_EPI_COLSUM) csum += v;
        }
        row += 4;
        cp += cstep;
        if (flags & PN_EPI_ADDC) ap += astep;
        if (flags & PN_EPI_GATEBITS) bp += 4 * PN_MASK_WORDS;
        if (flags & PN_EPI_MASKOUT) mp += 4 * PN_MASK_WORDS;
    }
}
// column sums of one wave's 64 rows -> colsum[(m0 / 64 + wm)][gcol .. gcol + 3]
__device__ __forceinline__ void nt_epi_colsum(const PnGemmNt& g, f32x4 csum, int64_t m0, int wm, int gcol, bool col_ok,
                                              int lane) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float x = csum[c];
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        csum[c] = x;
    }
    if (lane < 16 && col_ok) {
        const int64_t prow = (m0 / 64) + wm;
        *reinterpret_cast<f32x4*>(g.colsum + prow * g.N + gcol) = csum;
    }
}

// Shared epilogue of the NT kernels (see the comment inside).  `smem` is the (now idle) staging LDS, at least
// 4 * 32 * EPL floats.
template <int NT, int OFF>
__device__ __forceinline__ void nt_epilogue_t(const PnGemmNt& g, f32x16 (&acc)[2][NT], float* smem, int64_t m0, int n0,
                                              int lane, int wid, int wm, int wn) {
    // ---- epilogue.  The accumulators hold one column per lane (row = (r&3) + 8*(r>>2) + 4*(lane>>5)), so a
    // direct store is 64 dword stores per lane, each touching two 128-B row segments.  Instead every wave
    // transposes its 32x64 half-tile through the (now free) staging LDS and streams full 256-B row segments
    // as float4; bias / per-ray bias / addend / ReLU / gate / mask bits / column sums are applied there.
    const int flags = g.flags;
    float* Ls = smem + wid * (32 * EPL);
    const int col4 = (lane & 15) * 4;  // column of this lane's float4 inside the wave's 64 columns
    const int gcol = n0 + wn * 64 + col4;
    const bool col_ok = gcol < g.N;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if ((flags & PN_EPI_BIAS) && col_ok) bias4 = *reinterpret_cast<const f32x4*>(g.bias + gcol);
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();  // every wave is done reading As / Bs
    // wave-uniform: this wave's 64 x 64 block lies inside the matrix -> no bounds tests in the slab steps
    const bool interior = (m0 + wm * 64 + 64 <= g.M) && (n0 + wn * 64 + 64 <= g.N);
    const int known = PN_EPI_BIAS | PN_EPI_ROWBIAS | PN_EPI_ADDC | PN_EPI_RELU | PN_EPI_GATE | PN_EPI_GATEBITS | PN_EPI_MASKOUT |
                      PN_EPI_COLSUM | PN_ABL(0x100);
    const int fsel = flags & known;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ls[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPL + tn * 32 + (lane & 31)] = acc[tm][OFF + tn][r];
        // The slab is private to this wave and a wave's LDS instructions execute in order: no workgroup barrier is
        // needed between the transposed write and the row reads (a __syncthreads() here also carries a fence that
        // waits for the previous slab's global stores).
        __builtin_amdgcn_wave_barrier();
        const int64_t row0 = m0 + wm * 64 + tm * 32;
#define PN_SLAB(FSET)                                                                                             \
    case (FSET):                                                                                                  \
        if (interior) nt_epi_slab<(FSET), true>(g, Ls, flags, row0, col4, gcol, col_ok, bias4, csum, lane);        \
        else nt_epi_slab<(FSET), false>(g, Ls, flags, row0, col4, gcol, col_ok, bias4, csum, lane);                \
        break;
        switch (fsel) {  // the flag sets pn_mlp.hip launches with; anything else takes the run-time path
            PN_SLAB(PN_EPI_BIAS | PN_EPI_RELU | PN_EPI_MASKOUT)
            PN_SLAB(PN_EPI_ROWBIAS | PN_EPI_RELU | PN_EPI_MASKOUT)
            PN_SLAB(PN_EPI_BIAS)
            PN_SLAB(PN_EPI_GATEBITS)
            PN_SLAB(PN_EPI_GATEBITS | PN_EPI_COLSUM)
            PN_SLAB(PN_EPI_ADDC | PN_EPI_GATEBITS | PN_EPI_COLSUM)
            PN_SLAB(PN_EPI_COLSUM)
            PN_SLAB(0)
            default:
                if (interior) nt_epi_slab<-1, true>(g, Ls, flags, row0, col4, gcol, col_ok, bias4, csum, lane);
                else nt_epi_slab<-1, false>(g, Ls, flags, row0, col4, gcol, col_ok, bias4, csum, lane);
        }
#undef PN_SLAB
        __builtin_amdgcn_wave_barrier();
    }
    if (flags & PN_EPI_COLSUM) nt_epi_colsum(g, csum, m0, wm, gcol, col_ok, lane);
}

__device__ __forceinline__ void nt_epilogue(const PnGemmNt& g, f32x16 (&acc)[2][2], float* smem, int64_t m0, int n0,
                                            int lane, int wid, int wm, int wn) {
    nt_epilogue_t<2, 0>(g, acc, smem, m0, n0, lane, wid, wm, wn);
}

#ifndef PN_NT_OCC
#define PN_NT_OCC 3
#endif
__global__ __launch_bounds__(256, PN_NT_OCC) void k_gemm_nt(PnGemmNt g, int tiles_n, int ntiles) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDT];
    float* As = smem;
    float* Bs = smem + BM * LDT;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    // chunk schedule over the (up to two) K segments
    const int nc0 = (g.seg[0].K + BK - 1) / BK;
    const int nc1 = (g.nseg > 1) ? (g.seg[1].K + BK - 1) / BK : 0;
    const int nchunks = nc0 + nc1;
    // segment descriptors as scalars (selected with s_cselect in the loop, never re-read from kernarg memory)
    const float* const A0 = g.seg[0].A;
    const float* const B0 = g.seg[0].B;
    const int lda0 = g.seg[0].lda, ldb0 = g.seg[0].ldb, K0 = g.seg[0].K;
    const float* const A1 = g.nseg > 1 ? g.seg[1].A : A0;
    const float* const B1 = g.nseg > 1 ? g.seg[1].B : B0;
    const int lda1 = g.nseg > 1 ? g.seg[1].lda : lda0, ldb1 = g.nseg > 1 ? g.seg[1].ldb : ldb0;
    const int K1 = g.nseg > 1 ? g.seg[1].K : K0;
    const int64_t Mrows = g.M;
    const int Ncols = g.N;
    const int arow = (wm * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int brow = (wn * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;

    const int t = xcd_remap(blockIdx.x, ntiles);
    const int64_t m0 = (int64_t)(t / tiles_n) * BM;
    const int n0 = (t % tiles_n) * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto issue_load = [&](int cidx, NtRegs& r) {
        if (cidx < nchunks && !PN_ABL(g.flags & 0x200)) {
            const bool s1 = cidx >= nc0;
            nt_load(s1 ? A1 : A0, s1 ? lda1 : lda0, s1 ? B1 : B0, s1 ? ldb1 : ldb0, s1 ? K1 : K0, Mrows, Ncols,
                    (s1 ? cidx - nc0 : cidx) * BK, m0, n0, tid, r);
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int j = 0; j < BK / 8; ++j) {
            f32x4 a0 = *reinterpret_cast<const f32x4*>(As + arow + j * 8);
            f32x4 a1 = *reinterpret_cast<const f32x4*>(As + arow + 32 * LDT + j * 8);
            f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + brow + j * 8);
            f32x4 b1 = *reinterpret_cast<const f32x4*>(Bs + brow + 32 * LDT + j * 8);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], b0[kk], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], b1[kk], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b0[kk], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b1[kk], acc[1][1], 0, 0, 0);
            }
        }
    };
    NtRegs regs;
    nt_load(A0, lda0, B0, ldb0, K0, Mrows, Ncols, 0, m0, n0, tid, regs);
    nt_store(As, Bs, tid, regs);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        issue_load(c + 1, regs);
        compute();
        __syncthreads();
        if (c + 1 < nchunks) {
            nt_store(As, Bs, tid, regs);
            __syncthreads();
        }
    }

    nt_epilogue(g, acc, smem, m0, n0, lane, wid, wm, wn);
}

#ifdef PN_TRACE_NT  // debug build only (PN_EXTRA=-DPN_TRACE_NT): per-workgroup phase stamps of k_gemm_nt_dma
__device__ unsigned long long g_nt_trace[16384 * 8];
extern "C" int pn_trace_read(unsigned long long* out, int nblocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nt_trace), sizeof(unsigned long long) * 8 * nblocks) == hipSuccess ? 0 : -4;
}
#define NT_STAMP(i)                                                                              \
    do {                                                                                         \
        if (threadIdx.x == 0 && blockIdx.x < 16384) g_nt_trace[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define NT_CLOCK(i)                                                                              \
    do {                                                                                         \
        if (threadIdx.x == 0 && blockIdx.x < 16384) g_nt_trace[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define NT_STAMP(i)
#define NT_CLOCK(i)
#endif

// ---- NT with LDS-DMA staging ------------------------------------------------------------------------
// K-chunks of 16 floats (64-B rows).  global_load_lds_dwordx4 fills LDS lane-linearly (64 lanes x 16 B = 16 rows
// per wave-instruction), so the bank-conflict swizzle goes on the per-lane SOURCE address: LDS slot q' of row r
// holds the 16-B k-piece q = q' ^ ((r >> 2) & 3), and the fragment reads apply the same XOR.  Two LDS buffers,
// the next chunk's DMA in flight under the current chunk's 32 MFMAs, ONE barrier per chunk, no staging VGPRs.
#define DK 16
__global__ __launch_bounds__(256, 3) void k_gemm_nt_dma(PnGemmNt g, int tiles_n, int ntiles) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 32 * EPL];  // 34 816 B >= 2 buffers x (A + B) x 128 x 16 floats
    NT_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int nc0 = g.seg[0].K / DK;
    const int nc1 = (g.nseg > 1) ? g.seg[1].K / DK : 0;
    const int nchunks = nc0 + nc1;
    const float* const A0 = g.seg[0].A;
    const float* const B0 = g.seg[0].B;
    const int lda0 = g.seg[0].lda, ldb0 = g.seg[0].ldb;
    const float* const A1 = g.nseg > 1 ? g.seg[1].A : A0;
    const float* const B1 = g.nseg > 1 ? g.seg[1].B : B0;
    const int lda1 = g.nseg > 1 ? g.seg[1].lda : lda0, ldb1 = g.nseg > 1 ? g.seg[1].ldb : ldb0;
    const int t = xcd_remap(blockIdx.x, ntiles);
    const int64_t m0 = (int64_t)(t / tiles_n) * BM;
    const int n0 = (t % tiles_n) * BN;

    // staging: wave w moves rows 32w .. 32w+31 of A and of B (two 16-row pieces each)
    const int srow = (lane >> 2);                     // row inside a 16-row piece
    int64_t arow0 = m0 + wid * 32 + srow, arow1 = arow0 + 16;
    arow0 = arow0 < g.M ? arow0 : g.M - 1;
    arow1 = arow1 < g.M ? arow1 : g.M - 1;
    int brow0 = n0 + wid * 32 + srow, brow1 = brow0 + 16;
    brow0 = brow0 < g.N ? brow0 : g.N - 1;
    brow1 = brow1 < g.N ? brow1 : g.N - 1;
    // slot q' = lane & 3 of row r holds source piece q = q' ^ ((r >> 2) & 3); r = (wid*32 + piece*16 + srow)
    const int q0 = (lane & 3) ^ ((srow >> 2) & 3);    // same for both pieces: (16 >> 2) & 3 == 0
    auto stage = [&](int c, int buf) {
        const bool s1 = c >= nc0;
        const float* A = s1 ? A1 : A0;
        const float* B = s1 ? B1 : B0;
        const int lda = s1 ? lda1 : lda0, ldb = s1 ? ldb1 : ldb0;
        const int k0 = (s1 ? c - nc0 : c) * DK + q0 * 4;
        float* as = smem + buf * (2 * BM * DK);
        float* bs = as + BM * DK;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(A + arow0 * lda + k0), (lds_ptr_t)(as + (wid * 32) * DK), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(A + arow1 * lda + k0), (lds_ptr_t)(as + (wid * 32 + 16) * DK), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(B + (int64_t)brow0 * ldb + k0), (lds_ptr_t)(bs + (wid * 32) * DK), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(B + (int64_t)brow1 * ldb + k0), (lds_ptr_t)(bs + (wid * 32 + 16) * DK), 16, 0, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment reads: lane (r = lane & 31, h = lane >> 5) takes k-piece q = 2j + h of rows r and r + 32
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> 2) & 3;  // ((r + 32) >> 2) & 3 is the same
    const int a_base = (wm * 64 + fr) * DK, b_base = (wn * 64 + fr) * DK;

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    NT_STAMP(1);
    NT_CLOCK(5);
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) stage(c + 1, buf ^ 1);
        const float* As = smem + buf * (2 * BM * DK);
        const float* Bs = As + BM * DK;
#pragma unroll
        for (int j = 0; j < DK / 8; ++j) {
            const int qo = (((2 * j + fh) ^ sw) & 3) * 4;
            f32x4 a0 = *reinterpret_cast<const f32x4*>(As + a_base + qo);
            f32x4 a1 = *reinterpret_cast<const f32x4*>(As + a_base + 32 * DK + qo);
            f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + b_base + qo);
            f32x4 b1 = *reinterpret_cast<const f32x4*>(Bs + b_base + 32 * DK + qo);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], b0[kk], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], b1[kk], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b0[kk], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b1[kk], acc[1][1], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the next chunk has landed
        __syncthreads();
    }
    NT_STAMP(2);
    NT_CLOCK(6);
    nt_epilogue(g, acc, smem, m0, n0, lane, wid, wm, wn);
    NT_STAMP(3);
#ifdef PN_TRACE_NT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NT_STAMP(4);
#endif
}

int pn_launch_gemm_nt(const PnGemmNt& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.nseg < 1 || g.nseg > 2) return PN_ERR_BAD_SHAPE;
    for (int i = 0; i < g.nseg; ++i) {
        const PnSeg& sg = g.seg[i];
        if (sg.K <= 0 || (sg.K & 3) || (sg.lda & 3) || (sg.ldb & 3)) return PN_ERR_BAD_SHAPE;
        if (!sg.A || !sg.B) return PN_ERR_NULL;
        if ((reinterpret_cast<uintptr_t>(sg.A) & 15) || (reinterpret_cast<uintptr_t>(sg.B) & 15)) return PN_ERR_BAD_SHAPE;
    }
    if (!g.C) return PN_ERR_NULL;
    if ((g.flags & PN_EPI_BIAS) && !g.bias) return PN_ERR_NULL;
    if ((g.flags & PN_EPI_GATE) && !g.gate) return PN_ERR_NULL;
    if ((g.flags & PN_EPI_ROWBIAS) && (!g.rowbias || g.rows_per_ray <= 0)) return PN_ERR_NULL;
    if ((g.flags & PN_EPI_ADDC) && !g.addc) return PN_ERR_NULL;
    if ((g.flags & PN_EPI_GATEBITS) && !g.gate_bits) return PN_ERR_NULL;
    if ((g.flags & PN_EPI_MASKOUT) && !g.mask_out) return PN_ERR_NULL;
    if ((g.flags & PN_EPI_COLSUM) && !g.colsum) return PN_ERR_NULL;
    if ((g.N & 3) || (g.ldc & 3)) return PN_ERR_BAD_SHAPE;
    if ((g.flags & PN_EPI_GATE) && (g.ldg & 3)) return PN_ERR_BAD_SHAPE;
    int64_t tiles_m = (g.M + BM - 1) / BM;
    int tiles_n = (g.N + BN - 1) / BN;
    int64_t nwg = tiles_m * tiles_n;
    if (nwg > 0x7fffffff) return PN_ERR_BAD_SHAPE;
    double ksum = 0;
    for (int i = 0; i < g.nseg; ++i) ksum += g.seg[i].K;
    ProfScope prof(0, 2.0 * (double)g.M * g.N * ksum, s);
    bool dma = !(PN_DBG & 64) && !PN_ABL(g.flags & 0x200);
    for (int i = 0; i < g.nseg; ++i) dma = dma && (g.seg[i].K % DK == 0);
    if (dma) hipLaunchKernelGGL(k_gemm_nt_dma, dim3((unsigned)nwg), dim3(256), 0, s, g, tiles_n, (int)nwg);
    else hipLaunchKernelGGL(k_gemm_nt, dim3((unsigned)nwg), dim3(256), 0, s, g, tiles_n, (int)nwg);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

// pure-MFMA loop (no memory): what the chip sustains on v_mfma_f32_32x32x2_f32 at its loaded clock
__global__ __launch_bounds__(256) void k_mfma_probe(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = (float)threadIdx.x * 1e-3f, b = (float)blockIdx.x * 1e-4f + 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[0] = s;
}
extern "C" int pn_mfma_probe(float* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(k_mfma_probe, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

// ------------------------------------------------------------------------------------- TN
#define TN_MAXSEG 4
struct PnTnArgs {
    PnSegTn seg[TN_MAXSEG];
    int nseg;
    int N1, N2;
    int64_t rows_per_split;  // multiple of BK
    int64_t cum[TN_MAXSEG + 1];  // first 32-row chunk of every segment; cum[nseg..] = chunks_total
    int64_t chunks_total;
    float* slab;             // [nsplit][N1][N2]
};
// A workgroup walks consecutive chunks, so the segment a chunk belongs to changes at most 3 times: keep the current
// segment's operands in scalars and re-read the kernel-argument table only when a boundary is crossed (a table
// lookup per chunk costs scalar loads whose s_waitcnt lgkmcnt(0) also drains the LDS reads, or ~30 live SGPRs).
struct TnCursor {
    const float* X;
    const float* Y;
    int ldx, ldy, seg;
    int64_t M, c0, c_end;
};
__device__ __forceinline__ void tn_cursor_load(const PnTnArgs& g, TnCursor& t, int seg) {
    t.seg = seg;
    t.X = g.seg[seg].X;
    t.Y = g.seg[seg].Y;
    t.ldx = g.seg[seg].ldx;
    t.ldy = g.seg[seg].ldy;
    t.M = g.seg[seg].M;
    t.c0 = g.cum[seg];
    t.c_end = g.cum[seg + 1];
}
__device__ __forceinline__ void tn_cursor_init(const PnTnArgs& g, TnCursor& t, int64_t c) {
    int seg = 0;
    if (c >= g.cum[1]) seg = 1;
    if (c >= g.cum[2]) seg = 2;
    if (c >= g.cum[3]) seg = 3;
    tn_cursor_load(g, t, seg);
}
// advance to the segment holding chunk c (c never decreases); returns the first row of the chunk in its segment
__device__ __forceinline__ int64_t tn_cursor_seek(const PnTnArgs& g, TnCursor& t, int64_t c) {
    while (c >= t.c_end && t.seg < TN_MAXSEG - 1) tn_cursor_load(g, t, t.seg + 1);
    return (c - t.c0) * BK;
}

struct TnRegs {
    f32x4 x[4], y[4];
};

__device__ __forceinline__ void tn_load(const float* X, int ldx, const float* Y, int ldy, int64_t Mseg, int N1, int N2,
                                        int64_t r0, int i0, int j0, int tid, TnRegs& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int idx = tid + 256 * i;
        int row = idx >> 5, c4 = idx & 31;
        int64_t gr = r0 + row;
        const bool rin = gr < Mseg;
        gr = rin ? gr : Mseg - 1;
        int cx = i0 + c4 * 4, cy = j0 + c4 * 4;
        const bool xin = cx < N1, yin = cy < N2;
        cx = xin ? cx : N1 - 4;
        cy = yin ? cy : N2 - 4;
        f32x4 vx = *reinterpret_cast<const f32x4*>(X + gr * ldx + cx);
        f32x4 vy = *reinterpret_cast<const f32x4*>(Y + gr * ldy + cy);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        r.x[i] = (rin && xin) ? vx : z;
        r.y[i] = (rin && yin) ? vy : z;
    }
}

__device__ __forceinline__ void tn_store(float* Xs, float* Ys, int tid, const TnRegs& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int idx = tid + 256 * i;
        int row = idx >> 5, c4 = idx & 31;
        *reinterpret_cast<f32x4*>(Xs + row * LDX + c4 * 4) = r.x[i];
        *reinterpret_cast<f32x4*>(Ys + row * LDX + c4 * 4) = r.y[i];
    }
}

__global__ __launch_bounds__(256) void k_gemm_tn(PnTnArgs g, int tiles2, int ntiles, int nsplit) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDX];
    float* Xs = smem;
    float* Ys = smem + BK * LDX;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // blocks b and b+8 share an XCD (own L2): put the tiles of one row-split on one XCD, adjacent in
    // dispatch order, so the X / Y panels they share are fetched from HBM once.
    const int L = blockIdx.x, xcd = L & 7, jj = L >> 3;
    const int tile = jj % ntiles;
    const int64_t split = (int64_t)(jj / ntiles) * 8 + xcd;
    if (split >= nsplit) return;
    const int i0 = (tile / tiles2) * BM, j0 = (tile % tiles2) * BN;
    const int64_t c_begin = split * (g.rows_per_split / BK);
    int64_t c_end = c_begin + g.rows_per_split / BK;
    if (c_end > g.chunks_total) c_end = g.chunks_total;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int N1 = g.N1, N2 = g.N2;
    TnCursor cur;
    tn_cursor_init(g, cur, c_begin);
    auto load_chunk = [&](int64_t c, TnRegs& regs) {
        const int64_t r0 = tn_cursor_seek(g, cur, c);
        tn_load(cur.X, cur.ldx, cur.Y, cur.ldy, cur.M, N1, N2, r0, i0, j0, tid, regs);
    };
    if (c_begin < c_end) {
        TnRegs regs;
        load_chunk(c_begin, regs);
        tn_store(Xs, Ys, tid, regs);
        __syncthreads();
        const int xo = (lane >> 5) * LDX + wm * 64 + (lane & 31);
        const int yo = (lane >> 5) * LDX + wn * 64 + (lane & 31);
        for (int64_t c = c_begin; c < c_end; ++c) {
            if (c + 1 < c_end) load_chunk(c + 1, regs);
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                float a0 = Xs[xo + kk * 2 * LDX], a1 = Xs[xo + kk * 2 * LDX + 32];
                float b0 = Ys[yo + kk * 2 * LDX], b1 = Ys[yo + kk * 2 * LDX + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __syncthreads();
            if (c + 1 < c_end) {
                tn_store(Xs, Ys, tid, regs);
                __syncthreads();
            }
        }
    }
    float* out = g.slab + split * (int64_t)g.N1 * g.N2;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = j0 + wn * 64 + tn * 32 + (lane & 31);
            if (col >= g.N2) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.N1) out[(int64_t)row * g.N2 + col] = acc[tm][tn][r];
            }
        }
}

// ---- TN, 128 x 256 output tile (weight gradients of the 256 x 256 layers) ---------------------------------
// 4 waves, each a 128 x 64 slab = 4 x 2 MFMA tiles (128 accumulator registers): per k-step ONE ds_read_b128 of X
// (columns 4i..4i+3 -> tile tm covers the stride-4 set {4i + tm}) and ONE ds_read_b64 of Y feed 8 MFMAs, half the
// LDS instructions per MFMA of the 128 x 128 kernel.  16-row sub-chunks by LDS-DMA, two buffers (48 KB).
#define TW 16
__global__ __launch_bounds__(256, 3) void k_gemm_tn_wide(PnTnArgs g, int tiles2, int ntiles, int nsplit) {
    __shared__ __attribute__((aligned(16))) float smem[2 * TW * (128 + 256)];  // 48 KB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int L = blockIdx.x, xcd = L & 7, jj = L >> 3;
    const int tile = jj % ntiles;
    const int64_t split = (int64_t)(jj / ntiles) * 8 + xcd;
    if (split >= nsplit) return;
    const int i0 = (tile / tiles2) * 128, j0 = (tile % tiles2) * 256;
    const int64_t c_begin = split * (g.rows_per_split / BK);
    int64_t c_end = c_begin + g.rows_per_split / BK;
    if (c_end > g.chunks_total) c_end = g.chunks_total;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // X sub-chunk [16][128]: 8 pieces of 2 rows; Y sub-chunk [16][256]: 16 pieces of 1 row.  Wave w: X pieces 2w, 2w+1
    // and Y pieces 4w .. 4w+3.  stage() returns the number of valid rows of the sub-chunk it issued.
    TnCursor cur;
    tn_cursor_init(g, cur, c_begin);
    auto stage = [&](int64_t sc, int buf) -> int {
        const int64_t r0 = tn_cursor_seek(g, cur, sc >> 1) + (sc & 1) * TW;
        const float* X = cur.X;
        const float* Y = cur.Y;
        const int ldx = cur.ldx, ldy = cur.ldy;
        const int64_t Mseg = cur.M;
        float* xs = smem + buf * (TW * 384);
        float* ys = xs + TW * 128;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int piece = wid * 2 + p;
            int64_t gr = r0 + piece * 2 + (lane >> 5);
            gr = gr < Mseg ? gr : Mseg - 1;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(X + gr * ldx + i0 + (lane & 31) * 4), (lds_ptr_t)(xs + piece * 256), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = wid * 4 + p;  // = row
            int64_t gr = r0 + piece;
            gr = gr < Mseg ? gr : Mseg - 1;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Y + gr * ldy + j0 + lane * 4), (lds_ptr_t)(ys + piece * 256), 16, 0, 0);
        }
        const int64_t v = Mseg - r0;
        return v >= TW ? TW : (v < 0 ? 0 : (int)v);
    };

    const int64_t s_begin = 2 * c_begin, s_end = 2 * c_end;
    if (s_begin < s_end) {
        int vr_next = stage(s_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const int xo = (lane >> 5) * 128 + 4 * (lane & 31);
        const int yo = (lane >> 5) * 256 + wid * 64 + 2 * (lane & 31);
        for (int64_t sc = s_begin; sc < s_end; ++sc) {
            const int buf = (int)((sc - s_begin) & 1);
            float* Xs = smem + buf * (TW * 384);
            float* Ys = Xs + TW * 128;
            const int vr = vr_next;
            if (vr < TW) {  // ragged tail of a segment: zero the rows past the end (uniform branch)
                for (int e = tid; e < (TW - vr) * 128; e += 256) Xs[vr * 128 + e] = 0.f;
                for (int e = tid; e < (TW - vr) * 256; e += 256) Ys[vr * 256 + e] = 0.f;
                __syncthreads();
            }
            if (sc + 1 < s_end) vr_next = stage(sc + 1, buf ^ 1);
#pragma unroll
            for (int kk = 0; kk < TW / 2; ++kk) {
                f32x4 a = *reinterpret_cast<const f32x4*>(Xs + xo + kk * 2 * 128);
                f32x2 b = *reinterpret_cast<const f32x2*>(Ys + yo + kk * 2 * 256);
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
                    acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[0], acc[tm][0], 0, 0, 0);
                    acc[tm][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[1], acc[tm][1], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    float* out = g.slab + split * (int64_t)g.N1 * g.N2;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = j0 + wid * 64 + 2 * (lane & 31) + tn;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) + tm;
                out[(int64_t)row * g.N2 + col] = acc[tm][tn][r];
            }
        }
}

// Deterministic sum over a leading "partials" dimension: out[e] = sum_b src[b*stride + e'] for the
// elements e of a [rows, cols] block.  One block = 64 elements x 4 partial lanes; grid.y splits the partials.
__global__ __launch_bounds__(256) void k_reduce_rows(const float* src, int64_t nb, int64_t stride, int rows, int cols,
                                                      int src_ld, float* dst, int64_t dst_stride, int ldd,
                                                      int accumulate) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx;
    const int64_t per = (nb + gridDim.y - 1) / gridDim.y;
    const int64_t b0 = (int64_t)blockIdx.y * per;
    int64_t b1 = b0 + per;
    if (b1 > nb) b1 = nb;
    float acc = 0.f;
    int r = 0, c = 0;
    if (e < rows * cols) {
        r = e / cols;
        c = e % cols;
        const float* p = src + (int64_t)r * src_ld + c;
        int64_t b = b0 + ty;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
        for (; b + 28 < b1; b += 32) {  // eight independent loads in flight per thread
            a0 += p[b * stride];
            a1 += p[(b + 4) * stride];
            a2 += p[(b + 8) * stride];
            a3 += p[(b + 12) * stride];
            a4 += p[(b + 16) * stride];
            a5 += p[(b + 20) * stride];
            a6 += p[(b + 24) * stride];
            a7 += p[(b + 28) * stride];
        }
        for (; b < b1; b += 4) a0 += p[b * stride];
        acc = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && e < rows * cols) {
        float v = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
        float* d = dst + blockIdx.y * dst_stride + (int64_t)r * ldd + c;
        *d = accumulate ? (*d + v) : v;
    }
}

// Split-K slabs [nb][n] (contiguous, n % 4 == 0): first stage of their reduction, in place.  Block (x, y) sums the
// slabs of its range [y*per, (y+1)*per) over the 256 floats x owns and writes the result over the FIRST slab of its
// range (only this block touches that region), float4 per lane, 4 partial lanes per element, fixed order.
__global__ __launch_bounds__(256) void k_reduce_slabs4(float* slab, int64_t nb, int64_t stride4, int n4, int64_t per) {
    __shared__ f32x4 red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx;
    const int64_t b0 = (int64_t)blockIdx.y * per;
    int64_t b1 = b0 + per;
    if (b1 > nb) b1 = nb;
    if (b0 >= nb) return;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a0 = z, a1 = z, a2 = z, a3 = z;
    f32x4* p = reinterpret_cast<f32x4*>(slab) + (e < n4 ? e : 0);
    int64_t b = b0 + ty;
    for (; b + 12 < b1; b += 16) {
        a0 += p[b * stride4];
        a1 += p[(b + 4) * stride4];
        a2 += p[(b + 8) * stride4];
        a3 += p[(b + 12) * stride4];
    }
    for (; b < b1; b += 4) a0 += p[b * stride4];
    red[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && e < n4) p[b0 * stride4] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

int pn_launch_reduce_rows(const float* src, int64_t nb, int64_t stride, int rows, int cols, int src_ld, float* dst,
                          int ldd, int accumulate, float* scratch, hipStream_t s) {
    const int n = rows * cols;
    const unsigned gx = (unsigned)((n + 63) / 64);
    if (nb > 256 && scratch) {  // two stages: 64 partial sums, then the final (accumulating) one
        hipLaunchKernelGGL(k_reduce_rows, dim3(gx, 64), dim3(256), 0, s, src, nb, stride, rows, cols, src_ld, scratch,
                           (int64_t)n, cols, 0);
        PN_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_reduce_rows, dim3(gx, 1), dim3(256), 0, s, scratch, (int64_t)64, (int64_t)n, rows, cols,
                           cols, dst, (int64_t)0, ldd, accumulate);
        PN_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(k_reduce_rows, dim3(gx, 1), dim3(256), 0, s, src, nb, stride, rows, cols, src_ld, dst,
                           (int64_t)0, ldd, accumulate);
        PN_CHECK_LAUNCH();
    }
    return PN_OK;
}

static int tn_splits(int64_t Mtotal, int N1, int N2) {
    int tiles = ((N1 + BM - 1) / BM) * ((N2 + BN - 1) / BN);
    int64_t chunks = (Mtotal + BK - 1) / BK + 1;
    int64_t want = (1024 + tiles - 1) / tiles;   // ~4 workgroups per CU
    if ((N1 % 128 == 0) && (N2 % 256 == 0)) {    // wide-tile kernel: 3 workgroups per CU
        tiles = (N1 / 128) * (N2 / 256);
        want = 768 / tiles;
    }
    int64_t per = (chunks + want - 1) / want;    // chunks per split
    if (per < 4) per = 4;
    return (int)((chunks + per - 1) / per);
}

// Upper bound of tn_splits over every row count <= Mtotal (tn_splits itself is not monotone in Mtotal: crossing a
// multiple of `want` chunks raises the chunks per split and LOWERS the split count), with up to TN_MAXSEG segments.
int64_t pn_tn_work_floats(int64_t Mtotal, int N1, int N2) {
    int tiles = ((N1 + BM - 1) / BM) * ((N2 + BN - 1) / BN);
    int64_t want = (1024 + tiles - 1) / tiles;
    if ((N1 % 128 == 0) && (N2 % 256 == 0)) want = 768 / ((N1 / 128) * (N2 / 256));
    if (want < 1) want = 1;
    const int64_t chunks = (Mtotal + BK - 1) / BK + 1;
    int64_t bound = (chunks + 3) / 4;
    if (bound > want) bound = want;
    if (bound < 1) bound = 1;
    return bound * N1 * N2;
}

int pn_launch_gemm_tn(const PnSegTn* segs, int nseg, int N1, int N2, float* C, int ldc, int accumulate, float* work,
                      int64_t work_avail, hipStream_t s) {
    if (nseg < 1 || nseg > TN_MAXSEG || N1 <= 0 || N2 <= 0) return PN_ERR_BAD_SHAPE;
    if (!C || !work) return PN_ERR_NULL;
    PnTnArgs g;
    g.nseg = nseg;
    int64_t Mtotal = 0, chunks = 0;
    for (int i = 0; i < TN_MAXSEG; ++i) {
        g.seg[i] = segs[i < nseg ? i : 0];
        g.cum[i] = chunks;
        if (i >= nseg) continue;
        if (segs[i].M <= 0 || (segs[i].ldx & 3) || (segs[i].ldy & 3) || !segs[i].X || !segs[i].Y) return PN_ERR_BAD_SHAPE;
        if ((reinterpret_cast<uintptr_t>(segs[i].X) & 15) || (reinterpret_cast<uintptr_t>(segs[i].Y) & 15)) return PN_ERR_BAD_SHAPE;
        Mtotal += segs[i].M;
        chunks += (segs[i].M + BK - 1) / BK;
    }
    g.cum[TN_MAXSEG] = chunks;
    for (int i = nseg; i < TN_MAXSEG; ++i) g.cum[i] = chunks;
    if ((N1 & 3) || (N2 & 3)) return PN_ERR_BAD_SHAPE;
    g.N1 = N1;
    g.N2 = N2;
    g.chunks_total = chunks;
    const bool wide = (N1 % 128 == 0) && (N2 % 256 == 0) && !(PN_DBG & 128);
    int nsplit = tn_splits(Mtotal, N1, N2);
    if (work_avail >= 0 && (int64_t)nsplit * N1 * N2 > work_avail) return PN_ERR_BAD_SHAPE;  // slab too small
    int64_t per = (g.chunks_total + nsplit - 1) / nsplit;
    g.rows_per_split = per * BK;
    g.slab = work;
    int tiles1 = (N1 + BM - 1) / BM, tiles2 = (N2 + BN - 1) / BN;
    {
        ProfScope prof(1, 2.0 * (double)Mtotal * N1 * N2, s);
        const int ntiles = tiles1 * tiles2;
        const int groups = (nsplit + 7) / 8;
        if (wide) {
            const int t2 = N2 / 256, nt2 = (N1 / 128) * t2;
            hipLaunchKernelGGL(k_gemm_tn_wide, dim3(groups * nt2 * 8), dim3(256), 0, s, g, t2, nt2, nsplit);
        } else hipLaunchKernelGGL(k_gemm_tn, dim3(groups * ntiles * 8), dim3(256), 0, s, g, tiles2, ntiles, nsplit);
    }
    PN_CHECK_LAUNCH();
    const int64_t n = (int64_t)N1 * N2;
    if (nsplit >= 32) {  // two stages: Y in-place partial sums (float4 streams), then the final accumulate into C
        const int n4 = (int)(n / 4);
        const int bx = (n4 + 63) / 64;
        int64_t Y = (1024 + bx - 1) / bx;          // ~4096 waves in flight
        if (Y > nsplit / 8) Y = nsplit / 8;
        if (Y < 1) Y = 1;
        const int64_t per = (nsplit + Y - 1) / Y;
        Y = (nsplit + per - 1) / per;
        hipLaunchKernelGGL(k_reduce_slabs4, dim3(bx, (unsigned)Y), dim3(256), 0, s, work, (int64_t)nsplit, n / 4, n4, per);
        PN_CHECK_LAUNCH();
        return pn_launch_reduce_rows(work, Y, per * n, N1, N2, N2, C, ldc, accumulate, nullptr, s);
    }
    return pn_launch_reduce_rows(work, nsplit, n, N1, N2, N2, C, ldc, accumulate, nullptr, s);
}
