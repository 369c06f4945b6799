// pn_chain.hip — the radiance MLP as FUSED on-chip chains on the 16-bit (bf16 / fp16) matrix cores (gfx950).
//
// Orientation.  Every layer is computed transposed: H_out^T [features x samples] = W [features x K] * H_in^T
// [K x samples] on the bf16 MFMA.  An accumulator tile then holds one SAMPLE per lane column and FEATURES in its
// registers, which is exactly the B-operand shape of the next layer's MFMA (k runs over the registers): a wave owns
// TILE samples and carries their activations through the whole chain IN REGISTERS.  No activation ever goes through LDS,
// nothing is re-read from HBM between layers; activations are written out once (the weight-gradient GEMMs need them) in
// a sample-minor tiled layout ("T layout": [sample block of TILE][feature][TILE samples]).
//
// Two MFMA shapes share this code (PN_CHAIN_TILE):
//   TILE = 16: v_mfma_f32_16x16x32_bf16, 16 samples per wave, 96 + 64 registers of activations + accumulators, so TWO
//              waves fit a SIMD (8 waves per workgroup): one wave's epilogue (ReLU, gate bits, stores, 3-term split),
//              encoding and DMA issue run under the other's MFMAs.  Default.
//   TILE = 32: v_mfma_f32_32x32x16_bf16, 32 samples per wave, 192 + 128 registers, one wave per SIMD (4 per workgroup).
// Both are instances of one index scheme.  With NG = 64 / TILE lane groups, features come in QUAD BLOCKS of QB = 4 NG:
// lane (sample c, group g) holds features QB*qb + 4g + i (i = 0..3) of quad block qb.  An accumulator tile is ACCQ =
// TILE / QB consecutive quad blocks (registers 4q + i), a k-step of the B operand is two consecutive quad blocks
// (elements j = 4 (qb & 1) + i), so "accumulators -> next B operand" is a re-grouping of registers and the weights (A
// operand) are packed with the matching k order.
//
// Weights are the A operand.  pn_chain_pack lays them out in fragment order (one 1-KB wave fragment per (k-step, feature
// tile, plane)), cut into uniform chunks that a ring of LDS slots receives by LDS-DMA (buffer_load ... lds), several
// chunks ahead, across layer and tile boundaries; all waves of a workgroup consume the same chunk for their own samples.
//
// Arithmetic.  NP = 3: every fp32 operand is split exactly into three bf16 terms (x = h + m + l) and a product is
// accumulated in fp32 from its six partial products of weight >= 2^-16 (the error is that of an fp32 fma chain);
// NP = 2 (the default mode): every fp32 operand times a power of two as an fp16 pair, x 2^e = h + l (|error| < 2^-24 |x|),
// three partial products; the powers of two - one per weight matrix, one per SAMPLE in the chains, one per tensor in the
// weight-gradient GEMMs - keep fp16 in range and are undone exactly (see chain_gemm, split_into, ex_of, k_chain_wgrad);
// NP = 1: plain bf16 operands, fp32 accumulate (BASELINE configs[1]).
//
// Reference lines replaced: MLP.forward models/pano_mip_nerf.py:95-114 (PureMLP models/mip_nerf.py:81-102),
// integrated_pos_enc models/mip.py:394-428, pos_enc 431-441, and the autograd / functorch passes through them
// (vmap(jacrev) models/pano_mip_nerf.py:299-303).
#include "pn_common.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include <vector>

// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant expression
// in the body (immediate operands of inline asm)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#ifndef PN_CHAIN_TILE
#define PN_CHAIN_TILE 16
#endif
#ifndef PN_ABL_CHAIN  // timing ablations of the GEMM step (wrong results; never in the shipped build): bit 0 no refill DMA,
#define PN_ABL_CHAIN 0  // bit 1 no fragment reads, bit 2 no MFMAs, bit 3 no ring barrier, bit 4 no T-layout stores
#endif
#define HALF_PI_F 1.5707963705062866f
constexpr int TILE = PN_CHAIN_TILE;  // features per accumulator tile = samples per wave
static_assert(TILE == 16 || TILE == 32, "PN_CHAIN_TILE is 16 or 32");
constexpr int NG = 64 / TILE;        // lane groups
constexpr int QB = 4 * NG;           // features per quad block
constexpr int KSTEP = 2 * QB;        // features per k-step of the MFMA
constexpr int ACCQ = TILE / QB;      // quad blocks per accumulator tile
constexpr int ACCR = 4 * ACCQ;       // accumulator registers per tile
#ifndef PN_CHAIN_WAVES
#define PN_CHAIN_WAVES (PN_CHAIN_TILE == 16 ? 8 : 4)
#endif
// Waves per workgroup.  TILE 32: 4 (one per SIMD, up to 512 registers).  TILE 16: waves of <= 256 registers, two per SIMD,
// either as ONE workgroup of 8 waves with a 5-slot ring (125 KB; the default since round 3) or as TWO INDEPENDENT workgroups
// of 4 waves with a 3-slot ring each (-DPN_CHAIN_WAVES=4; the round-2 default).  Measured back to back on the same box, three
// boxes (profiles/r03_experiments.txt): forward 2.38 / 2.30, 2.35 / 2.31, 2.37 / 2.38 ms (two workgroups / one), reverse sweep
// 1.77 / 1.72, 1.77 / 1.70, 1.86 / 1.76, tangent sweep 1.71 / 1.66, 1.71 / 1.67, 1.85 / 1.79, backward 2.10 / 2.05, 2.16 /
// 2.09, 2.28 / 2.24: one workgroup is 0-6 % faster and streams every weight chunk into a CU's LDS once instead of twice.
// (In one workgroup the chunk barrier keeps the two waves of a SIMD in the same phase; in two they drift apart - neither
// matters: a kernel's time is close to the SUM of its MFMA time and of its VALU / store time either way (timing ablations,
// PN_ABL_CHAIN) - a 16x16x32 MFMA holds the SIMD's vector issue for 8 of its 16 cycles, so one wave's epilogue and its
// partner's products mostly take turns.  Also without effect in the 8-wave form: s_setprio 1 for waves 4-7, a sixth ring
// slot, 48-KB chunks (half as many barriers).)
constexpr int CH_WAVES = PN_CHAIN_WAVES;
constexpr int CH_THREADS = 64 * CH_WAVES;
constexpr int CH_SAMPLES = CH_WAVES * TILE;                    // samples per workgroup tile
constexpr int CH_WG_PER_CU = (TILE == 16 && CH_WAVES == 4) ? 2 : 1;
constexpr int CH_MIN_WAVES = CH_WAVES * CH_WG_PER_CU / 4;      // waves per SIMD (__launch_bounds__)
typedef float accv __attribute__((ext_vector_type(ACCR)));
// shapes of the MLP in k-steps / feature tiles
constexpr int KS_H = 256 / KSTEP, NT_H = 256 / TILE;      // a 256-wide hidden vector
constexpr int KS_ENC = 96 / KSTEP, NT_ENC = 96 / TILE;    // the 96-feature integrated encoding
constexpr int KS_PAD = 32 / KSTEP;                        // a 32-feature padded block (view encoding)
constexpr int KS_C = 128 / KSTEP, NT_C = 128 / TILE;      // the 128-wide view hidden vector

template <int NP>
struct Cfg {
    // fragments (1 KB each) per chunk; with two workgroups per CU a ring of three slots must stay under 80 KB
#ifdef PN_CHAIN_CF  // (experiments: chunk size in fragments)
    static constexpr int CF = PN_CHAIN_CF;
#else
    static constexpr int CF = NP >= 2 ? 24 : (CH_WG_PER_CU == 2 ? 16 : 32);
#endif
    static constexpr int PER = CF / NP;                // GEMM steps (one A fragment set each) per chunk
    static constexpr int SLOT = (CF + 1) * 1024;       // + 1 KB of aux floats (bias) per chunk
#ifdef PN_CHAIN_NSLOT
    static constexpr int NSLOT = PN_CHAIN_NSLOT;
#else
    static constexpr int NSLOT = CH_WG_PER_CU == 2 ? 3 : (NP >= 2 ? 5 : 4);  // ring slots
#endif
    static constexpr int D = NSLOT - 1;                // chunks in flight ahead of the one being consumed
    static constexpr int SHARE = CF / CH_WAVES;        // DMA instructions EVERY wave issues per chunk (waves 0-3 one more)
    static constexpr int LDS_BYTES = SLOT * NSLOT;
};

// Element type of the T tensors: fp32, except with plain bf16 operands (NP = 1), where the value that is stored is the
// very bf16 the next GEMM and the weight-gradient GEMM consume - storing it in 2 bytes loses nothing and halves the HBM
// traffic of a mode that is HBM-bound.
template <int NP>
struct TEl {
    typedef float type;
};
template <>
struct TEl<1> {
    typedef __bf16 type;
};
#define TP(p) reinterpret_cast<TE*>(p)  // a T-tensor pointer of an argument struct as its element type (TE: per kernel)
template <int NP>
struct PlaneOf {
    typedef bf16x8 type;
};
template <>
struct PlaneOf<2> {
    typedef f16x8 type;
};
template <int NP>
struct BFrag {
    typename PlaneOf<NP>::type p[NP];
};

// feature held by lane group g in position i of quad block qb
__host__ __device__ constexpr int feat(int qb, int g, int i) { return QB * qb + 4 * g + i; }
// k index (input feature) of element j of lane group g in k-step ks
__host__ __device__ constexpr int kmap(int ks, int g, int j) { return feat(2 * ks + (j >> 2), g, j & 3); }

// ----------------------------------------------------------------------------------------------- chain schedules
// One entry per GEMM of a chain, in execution order: KS k-steps, NT feature tiles; its steps (k-step major, tile minor)
// are cut into chunks of PER steps, the last chunk padded.  Layer 5 ([h4 | enc], K = 352) and the d enc GEMM ([x0 | x5],
// K = 512) are two accumulating GEMMs each (F_L5 + F_L5E, B_DENC0 + B_DENC1): the second operand is re-split from its
// stored T tensor when it is needed instead of being held in registers across the layers in between.
enum {  // forward-direction GEMMs
    F_L0 = 0, F_L1, F_L2, F_L3, F_L4, F_L5, F_L5E, F_L6, F_L7, F_DEN, F_EXTRA, F_VIEW, F_COLOR, F_COUNT
};
enum {  // backward-direction GEMMs
    B_COLOR = 0, B_VIEW, B_EXTRA, B_L7, B_L6, B_L5, B_L4, B_L3, B_L2, B_L1, B_DENC0, B_DENC1, B_COUNT
};
__host__ __device__ constexpr int fwd_ks(int i) {
    return (i == F_L0 || i == F_L5E) ? KS_ENC : i == F_VIEW ? KS_H + KS_PAD : i == F_COLOR ? KS_C : KS_H;
}
__host__ __device__ constexpr int fwd_nt(int i) { return (i == F_DEN || i == F_COLOR) ? 1 : i == F_VIEW ? NT_C : NT_H; }
__host__ __device__ constexpr int bwd_ks(int i) {
    return i == B_COLOR ? 1 : i == B_VIEW ? KS_C : i == B_EXTRA ? KS_H + 1 : KS_H;
}
__host__ __device__ constexpr int bwd_nt(int i) { return i == B_COLOR ? NT_C : (i == B_DENC0 || i == B_DENC1) ? NT_ENC : NT_H; }
template <int NP>
__host__ __device__ constexpr int chunks_of(int KS, int NT) { return (KS * NT + Cfg<NP>::PER - 1) / Cfg<NP>::PER; }
template <int NP>
__host__ __device__ constexpr int fwd_chunk0(int i) {
    int c = 0;
    for (int k = 0; k < i; ++k) c += chunks_of<NP>(fwd_ks(k), fwd_nt(k));
    return c;
}
template <int NP>
__host__ __device__ constexpr int bwd_chunk0(int i) {
    int c = 0;
    for (int k = 0; k < i; ++k) c += chunks_of<NP>(bwd_ks(k), bwd_nt(k));
    return c;
}

// ------------------------------------------------------------------------------------------------------- packing
struct PackSeg {
    int64_t off;     // float offset of the source matrix in the parameter block
    int ld;          // its leading dimension
    int k0, kvalid;  // this segment covers k in [k0, k0 + kvalid)
    int transposed;  // 0: A[i][k] = W[i*ld + col0 + (k-k0)] ; 1: A[i][k] = W[(k-k0)*ld + col0 + i]
    int col0;
};
struct PackLayer {
    int chunk0, KS, NT;
    int rows_valid;
    int nseg;
    PackSeg seg[2];
    int64_t aux_off;  // floats copied to the aux KB of the layer's LAST chunk (bias, added when the sum is complete); < 0: zeros
    int aux_n;
};
#define PACK_MAXL 32  // both directions in one table (F_COUNT + B_COUNT = 25 GEMMs)
struct PackTable {
    PackLayer L[PACK_MAXL];
    int n, nchunks;
};

// NP = 2 only: one power-of-two scale per packed GEMM, 2^wexp, that takes the largest |weight| of its segments into
// [2^14, 2^15) (fp16 holds 65504).  Exponents are capped from above only - +30 for a weight matrix, +80 for a sample's
// vector or a tensor (zeros, or magnitudes below 2^-66 = 1.4e-20, which then lose precision gradually): a bias enters a
// sum as bias * 2^(wexp + bex) and must stay inside fp32's range; large values are never capped, they scale down.
constexpr int EXP_TOP = 15;    // frexp exponent of the scaled maximum
constexpr int EXP_CAP = 80;    // samples / tensors of the forward chain (its sums start from a bias)
constexpr int EXP_CAP_W = 30;  // weight matrices
constexpr int EXP_CAP_Z = 120; // samples / tensors of the backward-direction chains and of the weight gradients: their sums
                               // start at zero, so only the operands' own range matters (deltas of 1e-30 keep full precision)
__device__ __forceinline__ int scale_exp(float amax, int cap = EXP_CAP) {
    const int e = EXP_TOP - __builtin_amdgcn_frexp_expf(amax);
    return e > cap ? cap : e;
}
// Small tables that a training step accumulates into (the per-evaluation tensor maxima) are cleared by a KERNEL (k_view_table's first
// workgroup), not by hipMemsetAsync - a WORKAROUND: with 256-byte memset nodes a replayed HIP graph gave NaN or stale-scale steps in 4
// of 14 runs of the 512-ray bench (never with eager launches), although the captured graph is a single linear chain in which every
// memset node has the kernel before it as predecessor and the accumulating kernel as successor (hipGraphGetEdges on the pre-fix
// tree: profiles/r03_graph_memset_nodes.txt): not a missing dependency, but how replayed memset nodes execute on this runtime
// (ROCm 7.2).  The weight maxima need no clear at all any more: every slice of k_chain_wexp writes its own word.
constexpr int WEXP_SLICES = 16;  // workgroups per GEMM of the table (one alone took 67 us on the transposed 256 x 352 matrix)
// every slice WRITES its own maximum (wmax[GEMM * WEXP_SLICES + slice]; k_chain_pack takes the largest): no table to clear, no atomics
__global__ void k_chain_wexp(PackTable tab, const float* params, uint32_t* wmax) {
    const PackLayer& L = tab.L[blockIdx.x];
    float m = 0.f;
    for (int s = 0; s < L.nseg; ++s) {
        const PackSeg& sg = L.seg[s];
        const int n = L.rows_valid * sg.kvalid;
        for (int idx = threadIdx.x + blockDim.x * blockIdx.y; idx < n; idx += blockDim.x * WEXP_SLICES) {
            const int i = idx / sg.kvalid, kr = idx - i * sg.kvalid;
            const float x = sg.transposed ? params[sg.off + (int64_t)kr * sg.ld + sg.col0 + i]
                                          : params[sg.off + (int64_t)i * sg.ld + sg.col0 + kr];
            m = fmaxf(m, fabsf(x));
        }
    }
    __shared__ float red[256];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) wmax[blockIdx.x * WEXP_SLICES + blockIdx.y] = __float_as_uint(red[0]);
}

// wexp: [0, 32) the exponents the chain kernels read (written here), then WEXP_SLICES maxima per GEMM from k_chain_wexp
template <int NP>
__global__ void k_chain_pack(PackTable tab, const float* params, unsigned char* out, int* wexp) {
    constexpr int CF = Cfg<NP>::CF, SLOT = Cfg<NP>::SLOT, PER = Cfg<NP>::PER;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const int64_t fi = gid >> 6;  // fragment slot index over all chunks, CF + 1 per chunk (the last is the aux KB)
    const int chunk = (int)(fi / (CF + 1)), f = (int)(fi % (CF + 1));
    if (chunk >= tab.nchunks) return;
    int li = 0;
    for (int i = 1; i < tab.n; ++i)
        if (chunk >= tab.L[i].chunk0) li = i;
    const PackLayer& L = tab.L[li];
    unsigned char* dst = out + (int64_t)chunk * SLOT + f * 1024 + lane * 16;
    if (f == CF) {  // aux: 256 floats, 4 per lane
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (chunk == L.chunk0 + (L.KS * L.NT + PER - 1) / PER - 1 && L.aux_off >= 0)  // the GEMM's LAST chunk
            for (int q = 0; q < 4; ++q)
                if (lane * 4 + q < L.aux_n) v[q] = params[L.aux_off + lane * 4 + q];
        memcpy(dst, v, 16);
        return;
    }
    const int p = f % NP;
    const int step = (chunk - L.chunk0) * PER + f / NP;  // k-step major, tile minor
    int wex = 0;
    if constexpr (NP == 2) {
        const uint32_t* wm = reinterpret_cast<const uint32_t*>(wexp) + 32 + li * WEXP_SLICES;
        uint32_t top = 0u;  // (non-negative floats order as unsigned integers)
#pragma unroll
        for (int sl = 0; sl < WEXP_SLICES; ++sl) top = wm[sl] > top ? wm[sl] : top;
        wex = scale_exp(__uint_as_float(top), EXP_CAP_W);
        if (chunk == L.chunk0 && f == 0 && lane == 0) wexp[li] = wex;
    }
    unsigned short o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (step < L.KS * L.NT) {
        const int ks = step / L.NT, tt = step % L.NT;
        const int i = TILE * tt + (lane % TILE), g = lane / TILE;
        for (int j = 0; j < 8; ++j) {
            const int k = kmap(ks, g, j);
            float x = 0.f;
            if (i < L.rows_valid)
                for (int s = 0; s < L.nseg; ++s) {
                    const PackSeg& sg = L.seg[s];
                    const int kr = k - sg.k0;
                    if (kr >= 0 && kr < sg.kvalid)
                        x = sg.transposed ? params[sg.off + (int64_t)kr * sg.ld + sg.col0 + i]
                                          : params[sg.off + (int64_t)i * sg.ld + sg.col0 + kr];
                }
            unsigned short bits;
            if constexpr (NP == 2) {  // fp16 pair of the scaled weight
                const float t = ldexpf(x, wex);
                const _Float16 hh = (_Float16)t;
                const _Float16 r = p == 0 ? hh : (_Float16)(t - (float)hh);
                __builtin_memcpy(&bits, &r, 2);
            } else {
                const __bf16 hb = (__bf16)x;
                __bf16 r = hb;
                if (p >= 1) {
                    const float r1 = x - (float)hb;
                    const __bf16 mb = (__bf16)r1;
                    r = mb;
                    if (p == 2) r = (__bf16)(r1 - (float)mb);
                }
                __builtin_memcpy(&bits, &r, 2);
            }
            o[j] = bits;
        }
    }
    memcpy(dst, o, 16);
}

#if defined(PN_TRACE_CHAIN) || defined(PN_TRACE_WG)  // debug builds only (-DPN_TRACE_CHAIN / -DPN_TRACE_WG): shader-clock stamps
__device__ unsigned long long g_chain_trace[64];
extern "C" int pn_chain_trace_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_trace), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -4;
}
#endif
#ifdef PN_TRACE_CHAIN
#define TR(i)                                                                                          \
    do {                                                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0 && st == (int64_t)gridDim.x) g_chain_trace[i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define TRX(i)                                                                                 \
    do {                                                                                       \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_chain_trace[i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define TR(i)
#define TRX(i)
#endif

// ------------------------------------------------------------------------------------------------ the weight ring
// All waves issue their share of every chunk's DMA and all consume every chunk.  acquire(): wait for my share of the
// oldest chunk in flight, barrier (everyone's share has landed; everyone is done with the chunk consumed before), open
// the refill round of the slot that chunk vacated; the GEMM steps of the chunk then issue the round's DMA instructions
// one at a time behind their MFMAs.
template <int NP>
struct Ring {
    static constexpr int CF = Cfg<NP>::CF, SLOT = Cfg<NP>::SLOT;
    static constexpr int PIECES = CF / CH_WAVES + 1;  // per wave and chunk: CF / waves fragments, + (waves 0-3) a quarter of the aux KB
    __amdgpu_buffer_rsrc_t rsrc;  // the packed (sub-)chain: the DMA source is descriptor + SGPR offset + lane * 16
    unsigned char* lds;
    uint32_t lds_addr;            // LDS byte address of `lds`
    int nchunk;                   // chunks per pass over the chain
    int pf, pslot, cslot;
    int wid, lane;
    int cur_soff;                 // refill target of the current round
    unsigned char* cur_lds;
    __device__ __forceinline__ void begin_round() {
        cur_soff = pf * SLOT;
        cur_lds = lds + pslot * SLOT;
        pf = (pf + 1 == nchunk) ? 0 : pf + 1;
        pslot = (pslot + 1 == Cfg<NP>::NSLOT) ? 0 : pslot + 1;
    }
    // An LDS-DMA instruction costs the issuing wave 60-185 cycles whatever it carries (MI355X_MICROARCH.md, cycle
    // constants).  The aux KB (the bias of a GEMM, in its LAST chunk's slot) travels with EVERY round (AUX is always true):
    // leaving it out of the rounds that cannot reach a chunk with a bias saved 14 % of the DMA instructions and no time, and as
    // a run-time test it cost far more than the instruction (a branch splits the step sequence into blocks with full waits at
    // their joins).  Waves 0-3 therefore issue CF / CH_WAVES + 1 instructions per round, waves 4-7 CF / CH_WAVES.
    template <int I, bool AUX>
    __device__ __forceinline__ void piece() {
        if constexpr (I < CF / CH_WAVES) {
            const int f = wid + CH_WAVES * I;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(cur_lds + f * 1024), 16, lane * 16, cur_soff + f * 1024, 0, 0);
        } else if constexpr (I == CF / CH_WAVES && AUX) {
            if (wid < 4) {  // wave-uniform
                const int o = CF * 1024 + wid * 256;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(cur_lds + o), 4, lane * 4, cur_soff + o, 0, 0);
            }
        }
    }
    template <int I0, int STRIDE, bool AUX = true>
    __device__ __forceinline__ void pieces_from() {  // pieces I0, I0 + STRIDE, ... of the current round
        if constexpr (I0 < PIECES) {
            piece<I0, AUX>();
            pieces_from<I0 + STRIDE, STRIDE, AUX>();
        }
    }
    __device__ __forceinline__ void start(const unsigned char* s, unsigned char* l, int n, int w, int ln, int first) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)s, 0, n * SLOT, 0x00020000);
        lds = l; nchunk = n; wid = w; lane = ln;
        lds_addr = (uint32_t)(uintptr_t)(lds_ptr_t)l;
        pf = first; pslot = 0; cslot = 0;
#pragma unroll
        for (int i = 0; i < Cfg<NP>::D; ++i) {
            begin_round();
            pieces_from<0, 1, true>();
        }
    }
#ifdef PN_TRACE_CHAIN
    unsigned long long t_lds = 0, t_dma = 0, t_bar = 0, n_acq = 0;
#endif
    __device__ __forceinline__ uint32_t acquire() {
        // my share of the oldest chunk in flight has landed when at most the D - 1 younger rounds are outstanding.  Counted with
        // SHARE = CF / CH_WAVES per round although waves 0-3 issue one more (the aux quarter): for them the wait is one round's
        // aux piece stricter than necessary (it over-waits, never under-waits: vmcnt retires in issue order)
        constexpr int N = Cfg<NP>::SHARE * (Cfg<NP>::D - 1);
#ifdef PN_TRACE_CHAIN
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t_lds += t1 - t0; t_dma += t2 - t1; t_bar += t3 - t2; ++n_acq;
#else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
        if constexpr (!(PN_ABL_CHAIN & 8)) __builtin_amdgcn_s_barrier();
#endif
        begin_round();
        const uint32_t p = lds_addr + cslot * SLOT;
        cslot = (cslot + 1 == Cfg<NP>::NSLOT) ? 0 : cslot + 1;
        return p;
    }
    __device__ __forceinline__ void drain() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef PN_TRACE_CHAIN
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            g_chain_trace[56] = t_lds; g_chain_trace[57] = t_dma; g_chain_trace[58] = t_bar; g_chain_trace[59] = n_acq;
        }
#endif
    }
};

// Eight waves in one workgroup: the second-dispatched half (waves 4-7) loses every issue arbitration against its SIMD
// partner at equal priority (MI355X_MICROARCH.md, two waves per SIMD, item 4): one static priority step for that half.
__device__ __forceinline__ void chain_prio(int wid) {
#ifdef PN_CHAIN_PRIO
    if (wid >= CH_WAVES / 2) __builtin_amdgcn_s_setprio(PN_CHAIN_PRIO);
#endif
}

// ------------------------------------------------------------------------------------------------------ the GEMM
__device__ __forceinline__ accv mfma1(const bf16x8& a, const bf16x8& b, accv v) {
#if PN_CHAIN_TILE == 32
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, v, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, v, 0, 0, 0);
#endif
}
__device__ __forceinline__ accv mfma1(const f16x8& a, const f16x8& b, accv v) {
#if PN_CHAIN_TILE == 32
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, v, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, v, 0, 0, 0);
#endif
}
template <int NP>
__device__ __forceinline__ accv mfma_split(const BFrag<NP>& a, const BFrag<NP>& b, accv v) {
    if constexpr (NP == 3) {  // small terms first
        v = mfma1(a.p[2], b.p[0], v);
        v = mfma1(a.p[0], b.p[2], v);
        v = mfma1(a.p[1], b.p[1], v);
        v = mfma1(a.p[1], b.p[0], v);
        v = mfma1(a.p[0], b.p[1], v);
        v = mfma1(a.p[0], b.p[0], v);
    } else if constexpr (NP == 2) {
        v = mfma1(a.p[1], b.p[0], v);
        v = mfma1(a.p[0], b.p[1], v);
        v = mfma1(a.p[0], b.p[0], v);
    } else {
        v = mfma1(a.p[0], b.p[0], v);
    }
    return v;
}

// A-fragment reads are inline asm with hand-counted waits: hipcc's own counting turned every other step's wait into
// lgkmcnt(0), which also waits for the reads just issued for the NEXT step (an LDS latency exposed per two steps).
template <int NP, int IDX>
__device__ __forceinline__ void read_a(BFrag<NP>& a, uint32_t addr) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a.p[p]) : "v"(addr), "n"((IDX * NP + p) * 1024) : "memory");
}
// all but the newest `LEFT` LDS reads of this wave have returned; ties the fragment registers to the wait
template <int NP, int LEFT>
__device__ __forceinline__ void wait_a(BFrag<NP>& a) {
    if constexpr (NP == 3)
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a.p[0]), "+v"(a.p[1]), "+v"(a.p[2]) : "n"(LEFT) : "memory");
    else if constexpr (NP == 2)
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a.p[0]), "+v"(a.p[1]) : "n"(LEFT) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a.p[0]) : "n"(LEFT) : "memory");
}
// Steps per unit.  U = 2 runs two feature tiles of the same k-step with their products interleaved (dependent products
// two issue slots apart).  Measured: no gain (NP = 2 forward 2.34 -> 2.52 ms with the extra fragment registers): a chain of
// dependent 16x16x32 products already issues at full rate, so every GEMM runs single steps.
template <int NP, int NT>
struct Unit {
    static constexpr int U = 1;
};
// Fragment look-ahead, in GEMM steps (a multiple of the unit): an LDS read takes ~110-130 cycles, a step's MFMAs
// 16 NP (NP + 1) / 2.  (Also measured without effect on the NP = 2 step time, 1 vs 3 vs 4 steps: a wave's GEMM phase is
// paced by its LDS-DMA issue, see Ring.)
template <int NP>
struct Look {
    static constexpr int N = NP == 3 ? 1 : 3;
};
// steps of chunk c of a GEMM of S steps
template <int NP>
__host__ __device__ constexpr int chunk_steps(int S, int c) {
    return (c == (S - 1) / Cfg<NP>::PER && S % Cfg<NP>::PER) ? S % Cfg<NP>::PER : Cfg<NP>::PER;
}
// fragments of step PP (if the GEMM has one): the chunk is acquired when the look-ahead reaches its first step (every
// read of the chunk before it has been issued by then)
template <int NP, int S, int PP>
__device__ __forceinline__ void read_step(Ring<NP>& R, uint32_t& sa, int lane, BFrag<NP>& f) {
    if constexpr (PP < S) {
        if constexpr (PP % Cfg<NP>::PER == 0) sa = R.acquire() + lane * 16;
        if constexpr (PN_ABL_CHAIN & 2) f = BFrag<NP>{};
        else read_a<NP, PP % Cfg<NP>::PER>(f, sa);
    }
}
// The share of the refill round (opened by the acquire of PP's chunk) that look-ahead position PP stands for.  The round
// of chunk c refills the chunk D after it: with BIASNEXT (a forward chain whose biases are read) the rounds of a GEMM's
// last D chunks carry the aux KB - they are the ones that can reach the first chunk of a later GEMM.
template <int NP, int S, int PP, bool BIASNEXT>
__device__ __forceinline__ void piece_step(Ring<NP>& R) {
    if constexpr (PP < S && !(PN_ABL_CHAIN & 1)) {
        constexpr int PER = Cfg<NP>::PER, c = PP / PER, n = (S + PER - 1) / PER;
        // (every round carries the aux KB: which rounds reach a chunk with a bias is static, but leaving the others out
        // bought nothing measurable - 14 % fewer DMA instructions, same time)
        R.template pieces_from<PP % PER, chunk_steps<NP>(S, c), true>();
    }
}
template <int NP>
__device__ __forceinline__ void tie(BFrag<NP>& a) {  // orders a use of `a` after the wait asm before it
    if constexpr (NP == 3) asm volatile("" : "+v"(a.p[0]), "+v"(a.p[1]), "+v"(a.p[2]));
    else if constexpr (NP == 2) asm volatile("" : "+v"(a.p[0]), "+v"(a.p[1]));
    else asm volatile("" : "+v"(a.p[0]));
}
__device__ __forceinline__ void mfma_pair(const BFrag<2>& a0, const BFrag<2>& a1, const BFrag<2>& b, accv& v0, accv& v1) {
    v0 = mfma1(a0.p[1], b.p[0], v0);
    v1 = mfma1(a1.p[1], b.p[0], v1);
    v0 = mfma1(a0.p[0], b.p[1], v0);
    v1 = mfma1(a1.p[0], b.p[1], v1);
    v0 = mfma1(a0.p[0], b.p[0], v0);
    v1 = mfma1(a1.p[0], b.p[0], v1);
}
// unit S0 .. S0 + U - 1 of a GEMM of KS x NT steps (k-step major): q[0..U) hold its fragments, q[U..] those of the steps
// after it; the fragments of steps S0 + LA .. are read before the unit's MFMAs, across chunk boundaries too; the unit then
// issues the shares of the refill round that those look-ahead positions stand for.
template <int NP, int KS, int NT, int S0, bool BIASNEXT, bool ZERO = false>
struct GemmStep {
    static constexpr int LA = Look<NP>::N, U = Unit<NP, NT>::U;
    static_assert(LA >= U && LA % U == 0 && (KS * NT) % U == 0, "look-ahead in whole units");
    static __device__ __forceinline__ void run(Ring<NP>& R, const BFrag<NP> (&b)[KS], accv (&acc)[NT], uint32_t& sa,
                                               BFrag<NP> (&q)[LA], int lane) {
        constexpr int S = KS * NT;
        constexpr int hi = (S0 + LA + U) < S ? (S0 + LA + U) : S;
        constexpr int young = hi > S0 + U ? hi - (S0 + U) : 0;  // fragment sets younger than the unit's in flight
        BFrag<NP> nw[U];
        read_step<NP, S, S0 + LA>(R, sa, lane, nw[0]);
        if constexpr (U == 2) read_step<NP, S, S0 + LA + 1>(R, sa, lane, nw[1]);
        wait_a<NP, NP * young>(q[0]);
        constexpr int ks = S0 / NT, t = S0 % NT;
        if constexpr (U == 2) {
            tie<NP>(q[1]);
            mfma_pair(q[0], q[1], b[ks], acc[t], acc[t + 1]);
        } else if constexpr (!(PN_ABL_CHAIN & 4)) {
            if constexpr (ZERO && ks == 0) {  // first k-step of a sum that starts at zero: C is the inline constant, not 64 v_mov
                accv z;
#pragma unroll
                for (int e = 0; e < ACCR; ++e) z[e] = 0.f;
                acc[t] = mfma_split<NP>(q[0], b[ks], z);
            } else {
                acc[t] = mfma_split<NP>(q[0], b[ks], acc[t]);
            }
        }
#ifdef PN_ABL_FILL  // timing experiment: PN_ABL_FILL independent VALU instructions behind every step's MFMAs
        {
            float f0 = __int_as_float(lane), f1 = f0;
#pragma unroll
            for (int k = 0; k < PN_ABL_FILL; ++k) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(f0) : "v"(f1));
            asm volatile("" ::"v"(f0));
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        piece_step<NP, S, S0 + LA, BIASNEXT>(R);
        if constexpr (U == 2) piece_step<NP, S, S0 + LA + 1, BIASNEXT>(R);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S0 + U < S) {
#pragma unroll
            for (int i = 0; i + U < LA; ++i) q[i] = q[i + U];
#pragma unroll
            for (int u = 0; u < U; ++u) q[LA - U + u] = nw[u];
            GemmStep<NP, KS, NT, S0 + U, BIASNEXT, ZERO>::run(R, b, acc, sa, q, lane);
        }
    }
};
// the first LA steps' fragments of a GEMM (chunk 0, just acquired) and their shares of its refill round
template <int NP, int S, int I, bool BIASNEXT>
__device__ __forceinline__ void gemm_prologue(Ring<NP>& R, uint32_t sa, BFrag<NP> (&q)[Look<NP>::N]) {
    if constexpr (I < Look<NP>::N && I < S) {
        read_a<NP, I>(q[I], sa);
        piece_step<NP, S, I, BIASNEXT>(R);
        gemm_prologue<NP, S, I + 1, BIASNEXT>(R, sa, q);
    }
}
// acc[t] (+)= W-chunks * b[0..KS).  ZERO: the sum starts at zero (the first k-step's products take C = 0); BIAS: the LAST
// chunk's aux KB holds the layer's bias, added when the sum is complete (position i of quad block qb <- bias[feat(qb, g, i)]),
// like the reference's addmm (models/pano_mip_nerf.py:95-114).
//
// NP = 2 works in scaled units: the weights of the GEMM carry 2^wexp and the B operand of this lane's sample 2^bex, so
// the MFMAs accumulate 2^sc times the true sums (sc = wexp + bex, per lane: an accumulator register belongs to ONE
// sample).  A running sum enters multiplied by 2^sc and the result leaves divided by it: exact (powers of two); with a bias
// the division and the addition are ONE fma per element (the bias used to enter the sum scaled, 64 v_ldexp per layer, and
// the result left through 64 more).
template <int NP, int KS, int NT, bool BIAS, bool ZERO, bool BIASNEXT = false>
__device__ __forceinline__ void chain_gemm(Ring<NP>& R, const BFrag<NP> (&b)[KS], accv (&acc)[NT], int lane, int sc = 0) {
    constexpr int CF = Cfg<NP>::CF;
    static_assert(Look<NP>::N <= Cfg<NP>::PER, "the look-ahead stays within one chunk boundary");
    const uint32_t s0 = R.acquire();
    if constexpr (!ZERO && NP == 2) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < ACCR; ++e) {
                const float x = acc[t][e];
                acc[t][e] = ldexpf(x, sc);
            }
    }
    uint32_t sa = s0 + lane * 16;
    BFrag<NP> q[Look<NP>::N];
    gemm_prologue<NP, KS * NT, 0, BIASNEXT>(R, sa, q);
    GemmStep<NP, KS, NT, 0, BIASNEXT, ZERO>::run(R, b, acc, sa, q, lane);
    if constexpr (BIAS) {
        // (sa: the GEMM's last chunk; its slot is refilled only behind the barrier of the NEXT acquire)
        const float* aux = reinterpret_cast<const float*>(R.lds + ((sa - lane * 16) - R.lds_addr) + CF * 1024) + 4 * (lane / TILE);
        const float inv = NP == 2 ? ldexpf(1.0f, -sc) : 1.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int qd = 0; qd < ACCQ; ++qd) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(aux + QB * (ACCQ * t + qd));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = acc[t][4 * qd + i];
                    acc[t][4 * qd + i] = NP == 2 ? __builtin_fmaf(x, inv, v[i]) : x + v[i];
                }
            }
    } else if constexpr (NP == 2) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < ACCR; ++e) {
                const float x = acc[t][e];
                acc[t][e] = ldexpf(x, -sc);
            }
    }
}

// --------------------------------------------------------------------------------------------- register plumbing
// NP = 2: x * 2^ex = h + l in fp16 (11 + 11 significant bits and the sign of l: the error is below 2^-24 |x| while
// l stays normal, and below 2^-39 of the column's largest element otherwise); NP = 3: x = h + m + l in bf16, exactly.
// The fp16 pairs of two elements, packed: h = fl16(x s), l = fl16(x s - h) for a power of two s, by mixed-precision FMAs
// that round once into their half of the destination - two instructions per element where ldexp, convert, convert
// back, subtract, convert take 3.5 (bit-identical to that sequence up to the sign of a zero; the chain kernels and the
// weight-gradient GEMMs are bound by their vector + matrix instruction time, and the split is a third of the former).
__device__ __forceinline__ void split2_pair(float x0, float x1, float s, uint32_t& h, uint32_t& l) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
}
// N pairs at once, one of the four steps for all pairs before the next: every instruction of a pair reads the register the
// previous one wrote (half-register writes), which costs a wait state (an s_nop 0, as expensive to issue as the FMA) when the
// two are adjacent - written pair by pair the split ran 3.5 issue slots per element instead of 2.
template <int N>
__device__ __forceinline__ void split2_pairs(const float (&x)[2 * N], float s, uint32_t (&h)[N], uint32_t (&l)[N]) {
#pragma unroll
    for (int q = 0; q < N; ++q) asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[q]) : "v"(x[2 * q]), "v"(s));
#pragma unroll
    for (int q = 0; q < N; ++q) asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[q]) : "v"(x[2 * q + 1]), "v"(s));
#pragma unroll
    for (int q = 0; q < N; ++q)
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l[q]) : "v"(x[2 * q]), "v"(s), "v"(h[q]));
#pragma unroll
    for (int q = 0; q < N; ++q)
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l[q]) : "v"(x[2 * q + 1]), "v"(s), "v"(h[q]));
}
__device__ __forceinline__ float pow2f(int e) { return __int_as_float((127 + e) << 23); }  // -126 <= e <= 127
template <int NP>
__device__ __forceinline__ void split_into(const float (&x)[8], BFrag<NP>& f, int ex = 0) {
    if constexpr (NP == 2) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const float s = pow2f(ex);  // (ex = min(15 - frexp exponent, cap <= 120) lies in [-113, 120])
        uint32_t hq[4], lq[4];
        split2_pairs<4>(x, s, hq, lq);
        u32x4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h[q] = hq[q];
            l[q] = lq[q];
        }
        f.p[0] = __builtin_bit_cast(f16x8, h);
        f.p[1] = __builtin_bit_cast(f16x8, l);
    }
    if constexpr (NP != 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 h = (__bf16)x[j];
            f.p[0][j] = h;
            if constexpr (NP == 3) {
                const float r1 = x[j] - (float)h;
                const __bf16 m = (__bf16)r1;
                f.p[1][j] = m;
                f.p[2][j] = (__bf16)(r1 - (float)m);
            }
        }
    }
}
// The exponent of a B operand (NP = 2) and, for the weight-gradient GEMMs that read the same values back from their T
// tensor, the largest |x| over the wave's samples (float bits; 0 when not NP = 2).
struct Ex {
    int ex;
    uint32_t top;
};
// largest |x| of this lane's SAMPLE from the largest |x| this lane holds of it (the lane groups of a column are reduced)
// (v_permlane32_swap / v_permlane16_swap: both halves of the exchange in one vector instruction, no LDS round trip - a
// ds_bpermute per step stood here, ~120 cycles of latency each in the middle of every layer's epilogue.  With both operands
// the same register, lanes l and l ^ 32 (l ^ 16) end up holding the pair (x[l & ~32], x[l | 32]).)
__device__ __forceinline__ void swap32(uint32_t x, uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void swap16(uint32_t x, uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ float col_max(float m) {  // m >= 0: the bit patterns order as unsigned integers
    uint32_t a, b;
    swap32(__float_as_uint(m), a, b);
    uint32_t u = a > b ? a : b;
    if constexpr (NG == 4) {
        swap16(u, a, b);
        u = a > b ? a : b;
    }
    return __uint_as_float(u);
}
// sum over the lane groups of a sample column (every lane group ends with it)
__device__ __forceinline__ float col_sum(float v) {
    uint32_t a, b;
    swap32(__float_as_uint(v), a, b);
    v = __uint_as_float(a) + __uint_as_float(b);
    if constexpr (NG == 4) {
        swap16(__float_as_uint(v), a, b);
        v = __uint_as_float(a) + __uint_as_float(b);
    }
    return v;
}
// maximum over the 16 lanes of a DPP row (every lane ends with it); bit patterns of non-negative floats order as integers
__device__ __forceinline__ uint32_t row_umax(uint32_t v) {
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v = v > o ? v : o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v = v > o ? v : o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    v = v > o ? v : o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror
    v = v > o ? v : o;
    return v;
}
// column maximum -> the operand's exponent (the column's maximum goes to [2^14, 2^15)) + the wave's maximum
__device__ __forceinline__ Ex ex_of(float lane_max, int cap = EXP_CAP) {
    const float cm = col_max(lane_max);
    Ex e;
    e.ex = scale_exp(cm, cap);
    uint32_t v = row_umax(__float_as_uint(cm));
    if constexpr (NG == 2) {  // 32 sample columns: two rows of 16
        const uint32_t o = (uint32_t)__shfl_xor((int)v, 16, 64);
        v = v > o ? v : o;
    }
    e.top = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    return e;
}
// Running maxima of a kernel's T tensors over all tiles of a wave (wave-uniform values; static indices only), added to
// the evaluation's table at the end: the weight-gradient GEMMs (NP = 2) scale each tensor by ONE power of two.
template <int N>
struct RunMax {
    uint32_t v[N];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = 0u;
    }
    __device__ __forceinline__ void upd(int idx, uint32_t x) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = (idx == k && x > v[k]) ? x : v[k];
    }
    // End of the kernel (the ring has drained: its LDS is free).  The workgroup's waves are reduced through LDS and one
    // lane per slot adds the result - only if it exceeds what the table already holds: every wave adding its own ten
    // maxima cost ~100 us per launch in atomics queueing on ten addresses (a 512-ray step ran 20 % slower than with the
    // six-product kernels).
    __device__ __forceinline__ void flush(uint32_t* slots, int lane, int wid, unsigned char* lds) const {
        if (!slots) return;  // (uniform)
        uint32_t* w = reinterpret_cast<uint32_t*>(lds);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < N; ++k) w[wid * N + k] = v[k];
        }
        __syncthreads();
        if (wid == 0 && lane < N) {
            uint32_t m = w[lane];
#pragma unroll
            for (int i = 1; i < CH_WAVES; ++i) m = m > w[i * N + lane] ? m : w[i * N + lane];
            if (m > __atomic_load_n(slots + lane, __ATOMIC_RELAXED)) atomicMax(slots + lane, m);
        }
    }
};
// slots of an evaluation's table of maxima (uint32 float bits, AM_COUNT per evaluation)
enum {
    AM_ENC = 0, AM_ACT0 = 1 /* h0..h7, [bottleneck | view encoding], view hidden */, AM_DELTA0 = 11 /* delta_0..7 */,
    AM_D8B = 19, AM_D8D = 20, AM_DHV = 21, AM_DRGB = 22, AM_RS0 = 23 /* r_0..7 */, AM_TANG0 = 31 /* hdot_0..7 */,
    AM_EDOT = 39, AM_COEF = 40, AM_COUNT = 64
};
template <int NT>
__device__ __forceinline__ float lane_amax(const accv (&acc)[NT]) {
    float m[4] = {0.f, 0.f, 0.f, 0.f};  // four running maxima: one chain of NT * ACCR dependent v_max3 otherwise
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < ACCR; ++e) {
            const float x = acc[t][e];
            m[e & 3] = fmaxf(m[e & 3], fabsf(x));
        }
    return fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]));
}
// value at position i of quad block qb of a vector held as accumulator tiles
#define AQ(acc, qb, i) (acc)[(qb) / ACCQ][4 * ((qb) % ACCQ) + (i)]
// accumulator tiles (NT tiles = NT * ACCQ / 2 k-steps of quad-block pairs) -> the first k-steps of a B operand
template <int NP, int NT, int KB>
__device__ __forceinline__ void split_acc(const accv (&acc)[NT], BFrag<NP> (&b)[KB], int ex) {
#pragma unroll
    for (int s = 0; s < NT * ACCQ / 2; ++s) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = AQ(acc, 2 * s + (j >> 2), j & 3);
        split_into<NP>(x, b[s], ex);
    }
}
// Returns the operand's exponent (NP = 2; `also`: largest |x| of further elements the caller appends to the operand).
template <int NP, int NT, int KB>
__device__ __forceinline__ Ex acc_to_b(const accv (&acc)[NT], BFrag<NP> (&b)[KB], float also = 0.f, int cap = EXP_CAP) {
    constexpr int KS = NT * ACCQ / 2;
    static_assert((NT * ACCQ) % 2 == 0 && KS <= KB, "quad blocks");
    Ex e{0, 0u};
    if constexpr (NP == 2) e = ex_of(fmaxf(lane_amax<NT>(acc), also), cap);
    split_acc<NP, NT, KB>(acc, b, e.ex);
    return e;
}
// T-layout store: base points at [block][0][0] + (lane % TILE) + 4 * (lane / TILE) * TILE floats
// (Measured: these one-dword-per-feature stores cost 15-21 % of a chain kernel's time, PN_ABL_CHAIN bit 4 - and a
// [feature / 4][sample][4] layout with one 16-byte store per quad block, 1 KB contiguous per wave instruction, gained only
// 0-4 % while the weight-gradient GEMM, which then needs transposing fragment reads, lost 19 %: the cost is the written
// bytes, not the instruction count.  Nor their burstiness: handing a vector's 64 stores to the NEXT layer's GEMM, one per
// step behind its MFMAs, left k_chain_dgrad<2> at 1.77 ms (1.79).  Non-temporal stores: forward 2.30 ms (2.36), dgrad 1.83
// (1.78), tangent 1.70 (1.72), backward unchanged - not adopted.)
#ifndef PN_T_NT
#define PN_T_NT 1
#endif
template <int NT, typename TE>
__device__ __forceinline__ void store_t(TE* base, const accv (&acc)[NT]) {
    if constexpr (PN_ABL_CHAIN & 16) return;  // (timing ablation: no T stores)
#pragma unroll
    for (int qb = 0; qb < NT * ACCQ; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = AQ(acc, qb, i);
#if PN_T_NT  // non-temporal like the Q24 stores (see store_q24): written once, read much later by one weight-gradient GEMM
            __builtin_nontemporal_store((TE)x, base + (QB * qb + i) * TILE);
#else
            base[(QB * qb + i) * TILE] = (TE)x;
#endif
        }
}

// ---- Q24: a T tensor in THREE bytes per element (round 3) --------------------------------------------------------------
// The tensors that only the weight-gradient GEMMs read back - h0..h6, hdot_0..6, delta_l and r_l for l = 1-4, 6, 7: 26 of the
// 32 KB a second-order evaluation writes per sample in its 256-wide vectors - are stored as fp32 ROUNDED TO 16 SIGNIFICANT BITS,
// the four features of a quad block of a sample in 12 bytes: [block][F / 4][16 samples][12 B].  A quad block of a lane is ONE
// 12-byte store (768 contiguous bytes per wave instruction) where four dword stores stood, and a quarter of the bytes of the
// step's HBM traffic in these tensors is gone on both sides - the chains' exposed store time and the weight-gradient GEMMs' read
// time follow the BYTES (profiles/r03_experiments.txt sections 9, 11, 13).  A 2^-17 rounding error per operand leaves a sum
// over samples ~1e-6 of its tensor's largest element off, the level of an fp32 GEMM's own summation noise (3e-6 on the same
// data: tests/test_gpu_chain.py::test_large_weight_gradient_sums_against_fp64_and_an_fp32_gemm); what the chains re-read
// themselves (encodings, r_5, delta_5) and the narrow tensors stay fp32.  fp16-pair builds with 16-sample tiles only.
#ifndef PN_NO_Q24  // (-DPN_NO_Q24=1: every T tensor fp32, for A/B measurements)
#define PN_NO_Q24 0
#endif
template <int NP>
constexpr bool kQ24 = (NP == 2 && TILE == 16 && !PN_NO_Q24);
__host__ __device__ constexpr bool q24_act(int slot) { return slot <= 6; }                  // h_l / hdot_l
__host__ __device__ constexpr bool q24_delta(int slot) { return slot != 0 && slot != 5; }   // delta_l / r_l
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
// Non-finite values: Inf stays Inf (0x7F800000 + 0x80 keeps its three upper bytes) and the NaN the hardware generates (0x7FC00000, and
// every quiet NaN whose payload does not reach bits 8 - 22 all set) stays a NaN, so a diverged activation or delta still gives a
// non-finite weight gradient (tests/test_gpu_chain.py::test_non_finite_values_survive_q24); the largest finite values (|x| >=
// 0x7F7FFF80) round up to Inf, as round-to-nearest does; only a NaN with payload bits 7 - 22 all ones (0x7FFFFF80 ...: never
// produced by arithmetic here) would wrap to zero - an exponent test per element (two more vector instructions on 26 K elements a
// sample) is not spent on it.  pano_nerf_amd.tlayout.q24_encode mirrors this bit for bit.
__device__ __forceinline__ u32x3 pack_q24(float x0, float x1, float x2, float x3) {
    // round to nearest on the dropped byte (ties away from zero: one tie in 256, a bias of 2^-26), then bytes 1-3 of each element
    const uint32_t e0 = __float_as_uint(x0) + 0x80u, e1 = __float_as_uint(x1) + 0x80u;
    const uint32_t e2 = __float_as_uint(x2) + 0x80u, e3 = __float_as_uint(x3) + 0x80u;
    u32x3 v;
    v[0] = __builtin_amdgcn_perm(e1, e0, 0x05030201u);
    v[1] = __builtin_amdgcn_perm(e2, e1, 0x06050302u);
    v[2] = __builtin_amdgcn_perm(e3, e2, 0x07060503u);
    return v;
}
// byte offset of this lane's part of a Q24 block: quad block qb of the lane is at + qb * 768
__device__ __forceinline__ int q24_lane(int c, int g) { return g * 192 + c * 12; }
#ifndef PN_Q24_NT
#define PN_Q24_NT 1
#endif
template <int NT>
__device__ __forceinline__ void store_q24(unsigned char* base, const accv (&acc)[NT]) {
    if constexpr (PN_ABL_CHAIN & 16) return;
#pragma unroll
    for (int qb = 0; qb < NT * ACCQ; ++qb)
        // non-temporal: written once, read once by a weight-gradient GEMM much later (same box, training step: 22.58 -> 22.30 ms; forward
        // 1771 -> 1744 us, backward 1714 -> 1682, and the 256 x 256 weight-gradient tile that reads them 794 -> 756: the stores no longer
        // displace the packed weights and the other operand from L2.  -DPN_Q24_NT=0: plain stores)
#if PN_Q24_NT
        __builtin_nontemporal_store(pack_q24(AQ(acc, qb, 0), AQ(acc, qb, 1), AQ(acc, qb, 2), AQ(acc, qb, 3)), reinterpret_cast<u32x3*>(base + qb * 768));
#else
        *reinterpret_cast<u32x3*>(base + qb * 768) = pack_q24(AQ(acc, qb, 0), AQ(acc, qb, 1), AQ(acc, qb, 2), AQ(acc, qb, 3));
#endif
}

// ReLU gate bits of a lane, MW words (4 with TILE 32, 2 with TILE 16): element j of k-step ks of the lane's B operand is half
// j & 1 of dword d = 4 ks + (j >> 1) of its packed planes, and its gate is bit (d & 15) + 16 (j & 1) of word d >> 4.  Seen
// from an accumulator, position i of quad block qb is element 4 (qb & 1) + i of k-step qb >> 1.
constexpr int MW = 4 / (NG / 2);  // gate words per lane for a 256-wide vector
__host__ __device__ constexpr int gate_word(int ks, int j) { return (4 * ks + (j >> 1)) >> 4; }
__host__ __device__ constexpr int gate_bit(int ks, int j) { return ((4 * ks + (j >> 1)) & 15) + 16 * (j & 1); }
struct Gate {
    uint32_t w[MW];
};
template <int NT>
__device__ __forceinline__ void relu(accv (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < ACCR; ++e) {
            const float x = acc[t][e];
            acc[t][e] = fmaxf(x, 0.f);
        }
}
// The gate words of a ReLU output from the LEADING plane of its B operand: an element is alive iff its leading 16-bit term
// is not zero (the term of a non-negative value is a non-negative half: min(bits, 1) as an unsigned 16-bit integer is the
// gate) - one packed minimum and one shift-or per TWO elements, where a compare, a select and an or per element stood.
// (A positive value whose leading term rounds to zero lies 2^-40 below its sample's largest activation with the fp16 pair,
// below 2^-133 with bf16: it counts as dead, in the forward chain - whose next operand holds a zero there - and in every
// backward pass alike.)
template <int NP, int KS, int KB>
__device__ __forceinline__ void gate_words(const BFrag<NP> (&b)[KB], Gate& g) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < MW; ++i) g.w[i] = 0u;
    uint32_t one = 0x00010001u;
    asm volatile("" : "+v"(one));  // (in a register: the packed minimum takes it as an operand)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const u32x4 h = __builtin_bit_cast(u32x4, b[ks].p[0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t hq = h[q];
            uint32_t m;  // (hipcc expands a vector min of two unsigned shorts into compares and selects: five instructions)
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(hq), "v"(one));
            asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(g.w[(4 * ks + q) >> 4]) : "v"(m), "n"((4 * ks + q) & 15));
        }
    }
}
template <int NT>
__device__ __forceinline__ void gate_bits(accv (&acc)[NT], const Gate& g) {
#pragma unroll
    for (int qb = 0; qb < NT * ACCQ; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ks = qb >> 1, j = 4 * (qb & 1) + i;
            const int m = __builtin_amdgcn_sbfe((int)g.w[gate_word(ks, j)], gate_bit(ks, j), 1);  // 0 or -1
            const float x = AQ(acc, qb, i);  // (a bit_cast applied directly to the vector element reads element 0)
            AQ(acc, qb, i) = __int_as_float(__float_as_int(x) & m);
        }
}
// gate words of slot `slot` of this lane's sample: [slot][Mp][NG][MW] u32
__device__ __forceinline__ Gate load_gate(const uint32_t* masks, int64_t slot, int64_t Mp, int64_t row, int g) {
    Gate r;
    const uint32_t* p = masks + ((slot * Mp + row) * 8 + MW * g);
    if constexpr (MW == 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    } else {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        r.w[0] = v.x; r.w[1] = v.y;
    }
    return r;
}
__device__ __forceinline__ void store_gate(uint32_t* masks, int64_t slot, int64_t Mp, int64_t row, int g, const Gate& r) {
    uint32_t* p = masks + ((slot * Mp + row) * 8 + MW * g);
    if constexpr (MW == 4) *reinterpret_cast<uint4*>(p) = make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]);
    else *reinterpret_cast<uint2*>(p) = make_uint2(r.w[0], r.w[1]);
}
// The gate words of a tile's layers are loaded once, in the order the chain uses them, and consumed from the front:
// static indices only (a run-time index puts the array in scratch memory).
template <int N>
__device__ __forceinline__ Gate pop_front(Gate (&q)[N]) {
    const Gate m = q[0];
#pragma unroll
    for (int i = 0; i + 1 < N; ++i) q[i] = q[i + 1];
    return m;
}


// ---------------------------------------------------------------------------------- shared pieces of the chains
__device__ __forceinline__ float ch_sp_d1(float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float ch_sp_d2(float x) {
    if (x > 20.f) return 0.f;
    const float s = 1.f / (1.f + expf(-x));
    return s * (1.f - s);
}
__device__ __forceinline__ float sel3(const float (&v)[3], int ch) { return ch == 0 ? v[0] : (ch == 1 ? v[1] : v[2]); }

// sin / cos of an fp32 argument as large as 2^20 (the encoding's 2^15 * mean), absolute error < 2e-7, in ~25 VALU
// instructions and without the divergent Payne-Hanek path of the library sinf (the encoding was 12 % + 9 % of a wave's
// forward tile): y / 2pi as an exact double-float product (fma error term + the low part of 1 / 2pi), the integer part
// removed exactly, the quarter turn folded, then Taylor kernels on [-pi/4, pi/4] (tools/check_fast_sincos.py validates the
// scheme in numpy: max error 1.9e-7 over 2^-5 <= |y| < 2^19).  `y` is the ALREADY ROUNDED fp32 argument: the reference's
// cosine is sin(fl32(y + pi/2)) (models/mip.py:428,437), which the callers reproduce by passing that rounded sum.
__device__ __forceinline__ float red_turns(float y, int& quadrant) {
    const float c_hi = 0.15915494f;    // fl32(1 / 2pi)
    const float c_lo = 6.4206382e-9f;  // 1 / 2pi - c_hi
    const float p = y * c_hi;
    float e = __builtin_fmaf(y, c_hi, -p);
    e = __builtin_fmaf(y, c_lo, e);
    const float f = (p - __builtin_rintf(p)) + e;  // fraction of a turn in [-0.5, 0.5] (+ a hair)
    const float q4 = f * 4.0f;
    const float qn = __builtin_rintf(q4);          // nearest quarter turn
    quadrant = (int)qn;
    return (q4 - qn) * 1.5707963267948966f;        // remainder in [-pi/4, pi/4]
}
__device__ __forceinline__ float ksin(float x) {
    const float z = x * x;
    float p = __builtin_fmaf(z, 2.7557319e-6f, -1.9841270e-4f);
    p = __builtin_fmaf(z, p, 8.3333333e-3f);
    p = __builtin_fmaf(z, p, -1.6666667e-1f);
    return __builtin_fmaf(x * z, p, x);
}
__device__ __forceinline__ float kcos(float x) {
    const float z = x * x;
    float p = __builtin_fmaf(z, -2.7557319e-7f, 2.4801587e-5f);
    p = __builtin_fmaf(z, p, -1.3888889e-3f);
    p = __builtin_fmaf(z, p, 4.1666667e-2f);
    p = __builtin_fmaf(z, p, -0.5f);
    return __builtin_fmaf(z, p, 1.0f);
}
__device__ __forceinline__ float fast_sin(float y) {
    int q;
    const float r = red_turns(y, q);
    const float sv = ksin(r), cv = kcos(r);
    const float v = (q & 1) ? cv : sv;
    return (q & 2) ? -v : v;
}
__device__ __forceinline__ float fast_cos(float y) {
    int q;
    const float r = red_turns(y, q);
    const float sv = ksin(r), cv = kcos(r);
    const float v = (q & 1) ? sv : cv;
    return ((q + 1) & 2) ? -v : v;
}

// The lane group g = lane / TILE has NG values and hipcc knows it: feature arithmetic written in terms of g is turned into
// one specialised code path per lane group, executed one after the other under exec masks (the encoding ran 3-4x longer
// than its instruction count).  An opaque copy keeps it as plain per-lane arithmetic.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
// level l = f / 3 and channel ch = f % 3 of encoding feature f (< 96) without an integer division; 2^l
__device__ __forceinline__ void level_of(int f, int& l, int& ch) {
    l = (f * 43691) >> 17;
    ch = f - 3 * l;
}
__device__ __forceinline__ float pow2i(int l) { return __int_as_float((127 + l) << 23); }

// per-lane coordinates of a wave's tile
struct Tile {
    int64_t blk, row, rc;  // sample block, sample row, row clamped into [0, M)
    int c, g, lo;          // sample in the block, lane group, lane part of every T-layout address
    bool live;
};
__device__ __forceinline__ Tile tile_of(int64_t st, int wid, int lane, int64_t M) {
    Tile t;
    t.c = lane % TILE;
    t.g = lane / TILE;
    t.blk = st * CH_WAVES + wid;
    t.row = t.blk * TILE + t.c;
    t.live = t.row < M;
    t.rc = t.live ? t.row : M - 1;
    t.lo = t.c + 4 * t.g * TILE;
    return t;
}
// T store of an F-feature vector of this wave's tile into the tensor at `slot`: fp32 T layout, or Q24 (`q`: wave-uniform)
// (The lane's part of the address goes through `opaque`: as a plain expression hipcc hoists "tensor base + lane offset" for every slot
// and both formats out of the tile loop as 64-bit pointers - with the run-time format flag of ABI 2 four of them no longer fitted the
// 256 registers, were SPILLED, and every epilogue reloaded one from scratch in front of its stores: a vector-memory load, i.e. an
// s_waitcnt vmcnt(0) that also waits for the weight ring's DMA pieces in flight and for the previous layer's stores.)
template <int NP, int NT>
__device__ __forceinline__ void store_vec(typename TEl<NP>::type* slot, const Tile& T, int F, bool q, const accv (&acc)[NT]) {
    if constexpr (kQ24<NP>) {
        if (q) {
            store_q24<NT>(reinterpret_cast<unsigned char*>(slot) + T.blk * (int64_t)(F * TILE * 3) + opaque(q24_lane(T.c, T.g)), acc);
            return;
        }
    }
    store_t<NT>(slot + T.blk * (int64_t)(F * TILE) + opaque(T.lo), acc);
}

// integrated positional encoding (MODE 0), or its tangent along v (MODE 1), of this lane's 96 / NG features -> T-layout
// store + B operand.  Features f and f + 48 (sine / "cosine" of the same argument) sit in the same lane.
template <int NP, int MODE>
__device__ __forceinline__ Ex encode(const float (&mu)[3], const float (&cv)[3], const float (&vv)[3], int g,
                                      typename TEl<NP>::type* et, BFrag<NP> (&benc)[KS_ENC]) {
    typedef typename TEl<NP>::type TE;
    constexpr int NQ = 96 / QB;  // quad blocks of the encoding; the first half are the sines
    float x[NQ][4];
    const int g4 = 4 * opaque(g);
#pragma unroll
    for (int qb = 0; qb < NQ / 2; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int l, ch;
            level_of(QB * qb + g4 + i, l, ch);
            const float sc = pow2i(l);
            const float y = sel3(mu, ch) * sc;
            const float e0 = __expf(-0.5f * (sel3(cv, ch) * (sc * sc)));
            if constexpr (MODE == 0) {
                x[qb][i] = e0 * fast_sin(y);
                x[qb + NQ / 2][i] = e0 * fast_sin(y + HALF_PI_F);
            } else {
                const float e = e0 * sc * sel3(vv, ch);
                x[qb][i] = e * fast_cos(y);
                x[qb + NQ / 2][i] = e * fast_cos(y + HALF_PI_F);
            }
        }
#ifdef PN_TRACE_CHAIN
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(x[qb][i]));
    if (MODE == 0) TRX(31);
#endif
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i) et[(QB * qb + i) * TILE] = (TE)x[qb][i];
    if (MODE == 0) TRX(32);
    Ex e{0, 0u};
    if constexpr (NP == 2) {
        if constexpr (MODE == 0) {
            e.ex = EXP_TOP - 1;  // |feature| <= 1
            e.top = 0x3f800000u;
        } else {
            float m = 0.f;
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb)
#pragma unroll
                for (int i = 0; i < 4; ++i) m = fmaxf(m, fabsf(x[qb][i]));
            e = ex_of(m, EXP_CAP_Z);
        }
    }
#pragma unroll
    for (int s = 0; s < KS_ENC; ++s) {
        float y8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y8[j] = x[2 * s + (j >> 2)][j & 3];
        split_into<NP>(y8, benc[s], e.ex);
    }
    return e;
}
// B operand k-steps <- a stored T-layout block of this wave (fp32): the encoding for the skip columns, r5 / delta5
template <int NP, int KS>
__device__ __forceinline__ int reload_b(const typename TEl<NP>::type* src, BFrag<NP> (&b)[KS], int cap = EXP_CAP) {
    if constexpr (NP == 2) {  // all values first: the exponent comes from their maximum
        float x[KS][8];
        float m = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[ks][j] = (float)src[(QB * (2 * ks + (j >> 2)) + (j & 3)) * TILE];
                m = fmaxf(m, fabsf(x[ks][j]));
            }
        const int ex = scale_exp(col_max(m), cap);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) split_into<NP>(x[ks], b[ks], ex);
        return ex;
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (float)src[(QB * (2 * ks + (j >> 2)) + (j & 3)) * TILE];  // + 4 g rows via the lane offset
            split_into<NP>(x, b[ks]);
        }
        return 0;
    }
}
// d enc (accumulator tiles over 96 features) -> d mean of this lane's sample; every lane group ends with the sum
__device__ __forceinline__ void ipe_backward_tiles(const accv (&acc)[NT_ENC], const float (&mu)[3], const float (&cv)[3], int g,
                                                   float (&dm)[3]) {
    dm[0] = dm[1] = dm[2] = 0.f;
    const int g4 = 4 * opaque(g);
#pragma unroll
    for (int qb = 0; qb < 96 / QB; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool cosine = qb >= 96 / QB / 2;  // features 48..95 (QB divides 48)
            int l, ch;
            level_of(QB * (cosine ? qb - 96 / QB / 2 : qb) + g4 + i, l, ch);
            const float sc = pow2i(l);
            const float y = sel3(mu, ch) * sc;
            const float ex = __expf(-0.5f * (sel3(cv, ch) * (sc * sc))) * sc;
            const float d = AQ(acc, qb, i) * ex * fast_cos(cosine ? y + HALF_PI_F : y);
            dm[0] += ch == 0 ? d : 0.f;
            dm[1] += ch == 1 ? d : 0.f;
            dm[2] += ch == 2 ? d : 0.f;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dm[i] = col_sum(dm[i]);
    }
}
// gate, T-layout store and next B operand of a 256-wide hidden vector (backward-direction sweeps and the tangent sweep)
template <int NP>
__device__ __forceinline__ Ex finish_gated(accv (&acc)[NT_H], const Gate& m, typename TEl<NP>::type* slot, const Tile& T, bool q24,
                                           BFrag<NP> (&bh)[KS_H]) {
    gate_bits<NT_H>(acc, m);
    if (slot) store_vec<NP, NT_H>(slot, T, 256, q24, acc);  // (wave-uniform)
    return acc_to_b<NP, NT_H, KS_H>(acc, bh, 0.f, EXP_CAP_Z);  // (backward-direction and tangent sweeps only)
}

// pos_enc of the view directions (models/mip.py:431-441: [x | sin(x 2^l) | sin(x 2^l + pi/2)], l < 4) per view row, padded to
// 32 features: feature v < 3 is the direction, 3 <= v < 27 the sines, the rest zero.
// The first workgroup also clears the evaluation's table of tensor maxima (`amax`, `amax_n` <= 256 words; null: none): one launch
// in front of the forward chain instead of two.
__global__ void k_view_table(int64_t view_rows, const float* viewdirs, float* tab, uint32_t* amax, int amax_n) {
    if (amax && blockIdx.x == 0 && (int)threadIdx.x < amax_n) amax[threadIdx.x] = 0u;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= view_rows * 32) return;
    const int64_t r = idx >> 5;
    const int v = (int)(idx & 31);
    const float vd0 = viewdirs[r * 3], vd1 = viewdirs[r * 3 + 1], vd2 = viewdirs[r * 3 + 2];
    auto vsel = [&](int ch) { return ch == 0 ? vd0 : (ch == 1 ? vd1 : vd2); };
    const int i = v - 3, half = i >= 12 ? 1 : 0;
    int l, ch;
    level_of(i < 0 ? 0 : i - 12 * half, l, ch);
    const float xb = vsel(ch) * pow2i(l & 3);
    const float sv = fast_sin(half ? xb + HALF_PI_F : xb);
    tab[idx] = v < 3 ? vsel(v) : (v < PN_VIEW_DIM ? sv : 0.f);
}

// ------------------------------------------------------------------------------------------------- forward chain
struct FwdArgs {
    int64_t M, nst;        // sample rows, workgroup tiles of CH_SAMPLES
    int rows_per_ray, nc;
    int64_t view_rows;
    const unsigned char* pack;
    const int* wexp;       // NP = 2: exponent of every forward-direction GEMM's weights
    uint32_t* amax;        // NP = 2: this evaluation's table of maxima (AM_COUNT slots) or null
    const float* mean;     // [M,3]
    const float* cov;      // [M,3]
    const float* viewdirs; // [view_rows,3]
    float* view_tab;       // [view_rows,32]: the view encoding of every view row (27 features + zeros), built by k_view_table
    float* enc_t;          // T [96]
    float* acts_t;         // T: h0..h7 [256] x 8, then bottleneck + view encoding [288], then view hidden [128]; null: not kept
    uint32_t* masks;       // [9][Mp][8]
    float* raw_rgb;        // [M,3]
    float* raw_den;        // [M,nc]
    int q24;               // NP = 2, 16-sample tiles: h0..h6 in Q24 (see pack_q24); 0: every T tensor fp32
};
__host__ __device__ constexpr int64_t act_off(int slot, int64_t Mp) {  // float offset of activation slot in acts_t
    return slot <= 8 ? (int64_t)slot * Mp * 256 : 8 * Mp * 256 + Mp * 288;
}
__host__ __device__ constexpr int64_t acts_floats(int64_t Mp) { return 8 * Mp * 256 + Mp * 288 + Mp * 128; }

template <int NP>
__global__ __launch_bounds__(CH_THREADS, CH_MIN_WAVES) void k_chain_fwd(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef typename TEl<NP>::type TE;  // element type of the T tensors
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t Mp = a.nst * CH_SAMPLES;
    chain_prio(wid);
    Ring<NP> R;
    R.start(a.pack, lds, fwd_chunk0<NP>(F_COUNT), wid, lane, 0);
    RunMax<10> RM;  // activation slots 0..9
    RM.clear();
#ifdef PN_TRACE_CHAIN  // shader clock (s_memtime) against the 100 MHz constant clock (s_memrealtime) over the whole kernel
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_chain_trace[60] = __builtin_amdgcn_s_memtime();
        g_chain_trace[61] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // NP = 2: the weight exponents of this direction, one per lane, fetched once (a load per layer inside the tile loop
    // brought a vmcnt(0) in front of every GEMM; the other waves covered it, the timings did not move)
    int wtab = 0;
    if constexpr (NP == 2) wtab = a.wexp[lane & 15];
    for (int64_t st = blockIdx.x; st < a.nst; st += gridDim.x) {
        const Tile T = tile_of(st, wid, lane, a.M);
        TR(0);
        TE* et = TP(a.enc_t) + T.blk * (96 * TILE) + T.lo;
        BFrag<NP> bh[KS_H];
        accv acc[NT_H];
        Gate mw;
        int bex = 0;  // NP = 2: exponent of the B operand in flight
        auto wx = [&](int i) {
            if constexpr (NP == 2) return __builtin_amdgcn_readlane(wtab, i);
            else return 0;
        };
        {
            // ---- integrated positional encoding -> B operand of layer 0
            float mu[3], cv[3];
            const float zero3[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                mu[i] = a.mean[T.rc * 3 + i];
                cv[i] = a.cov[T.rc * 3 + i];
            }
            BFrag<NP> benc[KS_ENC];
#ifdef PN_TRACE_CHAIN
            asm volatile("" ::"v"(mu[0]), "v"(mu[1]), "v"(mu[2]), "v"(cv[0]), "v"(cv[1]), "v"(cv[2]));
            TRX(30);
#endif
            bex = encode<NP, 0>(mu, cv, zero3, T.g, et, benc).ex;
            TR(1);
            chain_gemm<NP, KS_ENC, NT_H, true, true, true>(R, benc, acc, lane, wx(F_L0) + bex);
            TR(2);
        }
        auto finish_hidden = [&](int slot) {  // ReLU, T store, next B operand, gate bits
            relu<NT_H>(acc);
            if (TP(a.acts_t)) store_vec<NP, NT_H>(TP(a.acts_t) + act_off(slot, Mp), T, 256, a.q24 && q24_act(slot), acc);  // (uniform)
            const Ex e = acc_to_b<NP, NT_H, KS_H>(acc, bh);
            gate_words<NP, KS_H, KS_H>(bh, mw);
            store_gate(a.masks, slot, Mp, T.blk * TILE + T.c, T.g, mw);
            bex = e.ex;
            RM.upd(slot, e.top);
        };
        finish_hidden(0);
        TR(3);
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) {
            chain_gemm<NP, KS_H, NT_H, true, true, true>(R, bh, acc, lane, wx(F_L0 + l) + bex);
            TR(2 + 2 * l);
            finish_hidden(l);
            TR(3 + 2 * l);
        }
        {  // ---- layer 5: [h4 | enc] as two accumulating GEMMs (F_L5, F_L5E)
            chain_gemm<NP, KS_H, NT_H, false, true, true>(R, bh, acc, lane, wx(F_L5) + bex);  // (the bias comes with F_L5E)
            BFrag<NP> benc[KS_ENC];
            const int eex = reload_b<NP, KS_ENC>(et, benc);
            chain_gemm<NP, KS_ENC, NT_H, true, false, true>(R, benc, acc, lane, wx(F_L5E) + eex);
            TR(12);
            finish_hidden(5);
            TR(13);
        }
#pragma unroll 1
        for (int l = 6; l <= 7; ++l) {
            chain_gemm<NP, KS_H, NT_H, true, true, true>(R, bh, acc, lane, wx(F_L6 + l - 6) + bex);
            TR(2 + 2 * l);
            finish_hidden(l);
            TR(3 + 2 * l);
        }
        {  // ---- density head (one tile; channel ch is feature ch: quad block 0 of lane group ch / 4)
            accv ad[1];
            chain_gemm<NP, KS_H, 1, true, true, true>(R, bh, ad, lane, wx(F_DEN) + bex);
            if (T.live) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = 4 * T.g + i;
                    if (ch < a.nc) a.raw_den[T.row * a.nc + ch] = ad[0][i];
                }
            }
        }
        TR(18);
        // ---- bottleneck (no activation), then the view layer over [bottleneck | view encoding]
        BFrag<NP> bv[KS_H + KS_PAD];
        int vex;
        {
            chain_gemm<NP, KS_H, NT_H, true, true, true>(R, bh, acc, lane, wx(F_EXTRA) + bex);
            TR(19);
            TE* bt = a.acts_t ? TP(a.acts_t) + act_off(8, Mp) + T.blk * (288 * TILE) + T.lo : nullptr;
            if (bt) store_t<NT_H>(bt, acc);
            const Ex e = acc_to_b<NP, NT_H, KS_H + KS_PAD>(acc, bv, 1.0f);  // the view encoding appended below is <= 1
            vex = e.ex;
            RM.upd(8, e.top);
            // the view encoding is a function of the view row (128 samples of a ray share it): read from the per-row table
            // (it used to be evaluated per sample here - 8 sines per lane, a third of this block's time)
            const float* vt = a.view_tab + ((T.rc / a.rows_per_ray) % a.view_rows) * 32 + 4 * T.g;
#pragma unroll
            for (int q = 0; q < KS_PAD; ++q) {
                float x[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(vt + QB * (2 * q + h));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        x[4 * h + i] = v[i];
                        if (bt) bt[(256 + QB * (2 * q + h) + i) * TILE] = (TE)v[i];
                    }
                }
                split_into<NP>(x, bv[KS_H + q], vex);
            }
        }
        TR(20);
        BFrag<NP> bc[KS_C];
        int cex;
        {
            accv av[NT_C];
            chain_gemm<NP, KS_H + KS_PAD, NT_C, true, true, true>(R, bv, av, lane, wx(F_VIEW) + vex);
            TR(21);
            Gate w4;
            relu<NT_C>(av);
            if (TP(a.acts_t)) store_t<NT_C>(TP(a.acts_t) + act_off(9, Mp) + T.blk * (128 * TILE) + T.lo, av);
            const Ex e = acc_to_b<NP, NT_C, KS_C>(av, bc);
            gate_words<NP, KS_C, KS_C>(bc, w4);
            store_gate(a.masks, 8, Mp, T.blk * TILE + T.c, T.g, w4);
            cex = e.ex;
            RM.upd(9, e.top);
        }
        TR(22);
        {
            accv ac[1];
            chain_gemm<NP, KS_C, 1, true, true, true>(R, bc, ac, lane, wx(F_COLOR) + cex);
            TR(23);
            if (T.live && T.g == 0) {
#pragma unroll
                for (int e = 0; e < 3; ++e) a.raw_rgb[T.row * 3 + e] = ac[0][e];
            }
        }
        TR(24);
    }
    R.drain();
#ifdef PN_TRACE_CHAIN
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_chain_trace[62] = __builtin_amdgcn_s_memtime();
        g_chain_trace[63] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if constexpr (NP == 2) {
        RM.flush(a.amax ? a.amax + AM_ACT0 : nullptr, lane, wid, lds);
        if (a.amax && blockIdx.x == 0 && tid == 0) a.amax[AM_ENC] = 0x3f800000u;  // |encoding| <= 1
    }
}

// ------------------------------------------------------------------------- density-gradient reverse sweep (level 1)
// r_7 = softplus'(z) * Wd[0] * gate_7 ; r_{l-1} = gate_{l-1} * (W_l[:, :256]^T r_l) ; d sigma / d enc = W_0^T r_0 +
// W_5[:, 256:]^T r_5 ; grad_mean = IPE^T (d sigma / d enc).  Replaces vmap(jacrev(compute_graph))[1]
// (models/pano_mip_nerf.py:299-303): one reverse sweep instead of the 8-output Jacobian.
struct SweepArgs {
    int64_t M, nst;
    int nc;
    float density_bias;
    const unsigned char* pack;   // first chunk of the sub-chain this kernel walks
    int nchunk;
    const int* wexp;             // NP = 2: weight exponents of the direction this kernel walks (indexed F_* / B_*)
    uint32_t* amax;              // NP = 2: this evaluation's table of maxima or null
    const uint32_t* masks;       // [9][Mp][8]
    const float* raw_den;        // [M,nc]
    const float* mean;
    const float* cov;
    const float* wd0;            // density_layer.weight[0] (256 floats, in the parameter block)
    const float* v;              // [M,3] tangent direction (tangent sweep)
    float* vec_t;                // T [8][256]: r_0..r_7 (reverse sweep) or hdot_0..hdot_7 (tangent sweep)
    int keep_all;                // reverse sweep: 0 = inference, vec_t is ONE slot and only r_5 is stored (for the reload)
    float* edot_t;               // T [96] (tangent sweep)
    float* out3;                 // [M,3] grad_mean (reverse sweep)
    float* sdot;                 // [M] (tangent sweep)
    int q24;                     // r_l / hdot_l in Q24 where q24_delta / q24_act say so; 0: fp32
};

template <int NP>
__global__ __launch_bounds__(CH_THREADS, CH_MIN_WAVES) void k_chain_dgrad(SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef typename TEl<NP>::type TE;  // element type of the T tensors
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t Mp = a.nst * CH_SAMPLES;
    chain_prio(wid);
    Ring<NP> R;
    R.start(a.pack, lds, a.nchunk, wid, lane, 0);
    RunMax<8> RM;  // r_0..r_7
    RM.clear();
    // NP = 2: the weight exponents of this direction, one per lane, fetched once (a load per layer inside the tile loop
    // brought a vmcnt(0) in front of every GEMM; the other waves covered it, the timings did not move)
    int wtab = 0;
    if constexpr (NP == 2) wtab = a.wexp[lane & 15];
    for (int64_t st = blockIdx.x; st < a.nst; st += gridDim.x) {
        const Tile T = tile_of(st, wid, lane, a.M);
        Gate mk[8];  // gates of h7, h6, ..., h0
#pragma unroll
        for (int l = 0; l < 8; ++l) mk[l] = load_gate(a.masks, 7 - l, Mp, T.blk * TILE + T.c, T.g);
        float mu[3], cv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            mu[i] = a.mean[T.rc * 3 + i];
            cv[i] = a.cov[T.rc * 3 + i];
        }
        const float sgm = T.live ? ch_sp_d1(a.raw_den[T.rc * a.nc] + a.density_bias) : 0.f;
        BFrag<NP> bh[KS_H];
        int bex = 0;
        auto wx = [&](int i) {
            if constexpr (NP == 2) return __builtin_amdgcn_readlane(wtab, i);
            else return 0;
        };
        {  // seed r_7
            const bool Q7 = kQ24<NP> && a.q24 && q24_delta(7);  // r_7 in three bytes per element: a quad block is one store (uniform)
            TE* rt = (a.keep_all && !Q7) ? TP(a.vec_t) + (int64_t)7 * Mp * 256 + T.blk * (256 * TILE) + T.lo : nullptr;
            unsigned char* rq = (a.keep_all && Q7) ? reinterpret_cast<unsigned char*>(TP(a.vec_t) + (int64_t)7 * Mp * 256) +
                                                         T.blk * (int64_t)(256 * TILE * 3) + q24_lane(T.c, T.g)
                                                   : nullptr;
            const Gate m7 = pop_front(mk);
            if constexpr (NP == 2) {  // |r_7| <= softplus'(z) * max |Wd[0]| over this lane's features
                float wm = 0.f;
#pragma unroll
                for (int qb = 0; qb < NT_H * ACCQ; ++qb) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(a.wd0 + QB * qb + 4 * T.g);
                    wm = fmaxf(fmaxf(wm, fmaxf(fabsf(wv[0]), fabsf(wv[1]))), fmaxf(fabsf(wv[2]), fabsf(wv[3])));
                }
                const Ex e = ex_of(sgm * wm, EXP_CAP_Z);
                bex = e.ex;
                RM.upd(7, e.top);
            }
#pragma unroll
            for (int ks = 0; ks < KS_H; ++ks) {
                float x[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(a.wd0 + QB * (2 * ks + h) + 4 * T.g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = 4 * h + i;
                        const uint32_t bit = (m7.w[gate_word(ks, j)] >> gate_bit(ks, j)) & 1u;
                        x[j] = bit ? sgm * wv[i] : 0.f;
                        if (rt) rt[(QB * (2 * ks + h) + i) * TILE] = (TE)x[j];
                    }
                    if constexpr (kQ24<NP>) {
                        if (rq) {
#if PN_Q24_NT
                            __builtin_nontemporal_store(pack_q24(x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]), reinterpret_cast<u32x3*>(rq + (2 * ks + h) * 768));
#else
                            *reinterpret_cast<u32x3*>(rq + (2 * ks + h) * 768) = pack_q24(x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]);
#endif
                        }
                    }
                }
                split_into<NP>(x, bh[ks], bex);
            }
        }
        accv acc[NT_H];
#pragma unroll 1
        for (int l = 7; l >= 1; --l) {
            chain_gemm<NP, KS_H, NT_H, false, true>(R, bh, acc, lane, wx(B_L7 + 7 - l) + bex);
            TE* dst = a.keep_all ? TP(a.vec_t) + (int64_t)(l - 1) * Mp * 256 : (l - 1 == 5 ? TP(a.vec_t) : nullptr);
            const Ex e = finish_gated<NP>(acc, pop_front(mk), dst, T, a.q24 && q24_delta(l - 1), bh);  // (r_5, re-read below, is never Q24)
            bex = e.ex;
            RM.upd(l - 1, e.top);
        }
        {  // d sigma / d enc over [r_0 | r_5] (two accumulating GEMMs: B_DENC0, B_DENC1), then the encoding's adjoint
            accv a3[NT_ENC];
            chain_gemm<NP, KS_H, NT_ENC, false, true>(R, bh, a3, lane, wx(B_DENC0) + bex);
            bex = reload_b<NP, KS_H>(TP(a.vec_t) + (a.keep_all ? (int64_t)5 * Mp * 256 : 0) + T.blk * (256 * TILE) + T.lo, bh, EXP_CAP_Z);
            chain_gemm<NP, KS_H, NT_ENC, false, false>(R, bh, a3, lane, wx(B_DENC1) + bex);
            float dm[3];
            ipe_backward_tiles(a3, mu, cv, T.g, dm);
            if (T.live && T.g == 0) {
                a.out3[T.row * 3] = dm[0];
                a.out3[T.row * 3 + 1] = dm[1];
                a.out3[T.row * 3 + 2] = dm[2];
            }
        }
    }
    R.drain();
    if constexpr (NP == 2) RM.flush(a.amax ? a.amax + AM_RS0 : nullptr, lane, wid, lds);
}

// ------------------------------------------------------------------------------------- tangent sweep (level 1)
// hdot_l = gate_l * (W_l hdot_{l-1}) from edot = d enc / d mean . v ; sdot = Wd[0] . hdot_7.  With the saved r_l this gives
// the second-order weight gradients dW_l += r_l^T hdot_{l-1} (the double backward of models/pano_mip_nerf.py:299-313).
template <int NP>
__global__ __launch_bounds__(CH_THREADS, CH_MIN_WAVES) void k_chain_tangent(SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef typename TEl<NP>::type TE;  // element type of the T tensors
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t Mp = a.nst * CH_SAMPLES;
    chain_prio(wid);
    Ring<NP> R;
    R.start(a.pack, lds, a.nchunk, wid, lane, 0);
    RunMax<9> RM;  // hdot_0..hdot_7, edot
    RM.clear();
    Ex e;
    // NP = 2: the weight exponents of this direction, one per lane, fetched once (a load per layer inside the tile loop
    // brought a vmcnt(0) in front of every GEMM; the other waves covered it, the timings did not move)
    int wtab = 0;
    if constexpr (NP == 2) wtab = a.wexp[lane & 15];
    for (int64_t st = blockIdx.x; st < a.nst; st += gridDim.x) {
        const Tile T = tile_of(st, wid, lane, a.M);
        Gate mk[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) mk[l] = load_gate(a.masks, l, Mp, T.blk * TILE + T.c, T.g);
        TE* et = TP(a.edot_t) + T.blk * (96 * TILE) + T.lo;
        BFrag<NP> bh[KS_H];
        accv acc[NT_H];
        int bex = 0;
        auto wx = [&](int i) {
            if constexpr (NP == 2) return __builtin_amdgcn_readlane(wtab, i);
            else return 0;
        };
        {
            float mu[3], cv[3], vv[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                mu[i] = a.mean[T.rc * 3 + i];
                cv[i] = a.cov[T.rc * 3 + i];
                vv[i] = T.live ? a.v[T.rc * 3 + i] : 0.f;
            }
            BFrag<NP> benc[KS_ENC];
            e = encode<NP, 1>(mu, cv, vv, T.g, et, benc);
            bex = e.ex;
            RM.upd(8, e.top);
            chain_gemm<NP, KS_ENC, NT_H, false, true>(R, benc, acc, lane, wx(F_L0) + bex);
        }
        e = finish_gated<NP>(acc, pop_front(mk), TP(a.vec_t), T, a.q24 && q24_act(0), bh);
        bex = e.ex;
        RM.upd(0, e.top);
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) {
            chain_gemm<NP, KS_H, NT_H, false, true>(R, bh, acc, lane, wx(F_L0 + l) + bex);
            e = finish_gated<NP>(acc, pop_front(mk), TP(a.vec_t) + (int64_t)l * Mp * 256, T, a.q24 && q24_act(l), bh);
            bex = e.ex;
            RM.upd(l, e.top);
        }
        {
            chain_gemm<NP, KS_H, NT_H, false, true>(R, bh, acc, lane, wx(F_L5) + bex);
            BFrag<NP> benc[KS_ENC];
            const int eex = reload_b<NP, KS_ENC>(et, benc, EXP_CAP_Z);
            chain_gemm<NP, KS_ENC, NT_H, false, false>(R, benc, acc, lane, wx(F_L5E) + eex);
            e = finish_gated<NP>(acc, pop_front(mk), TP(a.vec_t) + (int64_t)5 * Mp * 256, T, a.q24 && q24_act(5), bh);
            bex = e.ex;
            RM.upd(5, e.top);
        }
        chain_gemm<NP, KS_H, NT_H, false, true>(R, bh, acc, lane, wx(F_L6) + bex);
        e = finish_gated<NP>(acc, pop_front(mk), TP(a.vec_t) + (int64_t)6 * Mp * 256, T, a.q24 && q24_act(6), bh);
        bex = e.ex;
        RM.upd(6, e.top);
        chain_gemm<NP, KS_H, NT_H, false, true>(R, bh, acc, lane, wx(F_L7) + bex);
        {
            gate_bits<NT_H>(acc, pop_front(mk));
            store_t<NT_H>(TP(a.vec_t) + (int64_t)7 * Mp * 256 + T.blk * (256 * TILE) + T.lo, acc);
            if constexpr (NP == 2) RM.upd(7, ex_of(lane_amax<NT_H>(acc), EXP_CAP_Z).top);
            float sd = 0.f;
#pragma unroll
            for (int qb = 0; qb < NT_H * ACCQ; ++qb) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(a.wd0 + QB * qb + 4 * T.g);
#pragma unroll
                for (int i = 0; i < 4; ++i) sd += AQ(acc, qb, i) * wv[i];
            }
            sd = col_sum(sd);
            if (T.live && T.g == 0) a.sdot[T.row] = sd;
        }
    }
    R.drain();
    if constexpr (NP == 2) RM.flush(a.amax ? a.amax + AM_TANG0 : nullptr, lane, wid, lds);
}

// ------------------------------------------------------------------------------------------------- backward chain
// d raw_rgb, d raw_density -> delta of every layer (T layout, for the weight-gradient GEMMs) -> optionally d mean.
struct BwdArgs {
    int64_t M, nst;
    int nc;
    float density_bias;
    const unsigned char* pack;  // backward chain
    int nchunk;                 // chunks walked per tile (with or without the d enc GEMM)
    const int* wexp;            // NP = 2: weight exponents of the backward-direction GEMMs (indexed B_*)
    uint32_t* amax;             // NP = 2: this evaluation's table of maxima or null
    const uint32_t* masks;
    const float* raw_den;       // [M,nc]
    const float* d_rgb;         // [M,3]
    const float* d_den;         // [M,nc]
    const float* sdot;          // [M] or null: second-order addend softplus''(z) * sdot on channel 0
    const float* mean;
    const float* cov;
    float* drgb_t;              // T [32]
    float* dhv_t;               // T [128]
    float* d8_t;                // T [288]: d bottleneck | d raw_density (padded)
    float* delta_t;             // T [8][256]
    float* coef_t;              // T [32] or null: row 0 = softplus'(z) (second-order dWd[0] term)
    float* d_mean;              // [M,3] or null
    int q24;                    // delta_l in Q24 where q24_delta says so; 0: fp32
};

template <int NP>
__global__ __launch_bounds__(CH_THREADS, CH_MIN_WAVES) void k_chain_bwd(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef typename TEl<NP>::type TE;  // element type of the T tensors
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t Mp = a.nst * CH_SAMPLES;
    chain_prio(wid);
    Ring<NP> R;
    R.start(a.pack, lds, a.nchunk, wid, lane, 0);
    RunMax<12> RM;  // delta_0..7, d bottleneck, d raw_density, d view hidden, d rgb
    RM.clear();
    Ex e;
    // NP = 2: the weight exponents of this direction, one per lane, fetched once (a load per layer inside the tile loop
    // brought a vmcnt(0) in front of every GEMM; the other waves covered it, the timings did not move)
    int wtab = 0;
    if constexpr (NP == 2) wtab = a.wexp[lane & 15];
    for (int64_t st = blockIdx.x; st < a.nst; st += gridDim.x) {
        const Tile T = tile_of(st, wid, lane, a.M);
        Gate mk[9];  // gates of the view hidden, h7, h6, ..., h0
#pragma unroll
        for (int l = 0; l < 9; ++l) mk[l] = load_gate(a.masks, 8 - l, Mp, T.blk * TILE + T.c, T.g);
        // ---- colour head: d hv = gate * (Wc^T d rgb); the k-step holds KSTEP features, 3 real (lane group 0)
        BFrag<NP> b1[1];
        int bex = 0;
        auto wx = [&](int i) {
            if constexpr (NP == 2) return __builtin_amdgcn_readlane(wtab, i);
            else return 0;
        };
        {
            float x[8];
            TE* dt = TP(a.drgb_t) + T.blk * (32 * TILE) + T.lo;
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f = QB * (j >> 2) + 4 * T.g + (j & 3);
                x[j] = (T.live && f < 3) ? a.d_rgb[T.rc * 3 + (f < 3 ? f : 0)] : 0.f;
                dt[(QB * (j >> 2) + (j & 3)) * TILE] = (TE)x[j];
                m = fmaxf(m, fabsf(x[j]));
            }
            if constexpr (NP == 2) {
                e = ex_of(m, EXP_CAP_Z);
                bex = e.ex;
                RM.upd(11, e.top);
            }
            split_into<NP>(x, b1[0], bex);
        }
        BFrag<NP> bc[KS_C];
        {
            accv av[NT_C];
            chain_gemm<NP, 1, NT_C, false, true>(R, b1, av, lane, wx(B_COLOR) + bex);
            gate_bits<NT_C>(av, pop_front(mk));
            store_t<NT_C>(TP(a.dhv_t) + T.blk * (128 * TILE) + T.lo, av);
            e = acc_to_b<NP, NT_C, KS_C>(av, bc, 0.f, EXP_CAP_Z);
            bex = e.ex;
            RM.upd(10, e.top);
        }
        // ---- view layer: d bottleneck = Wv[:, :256]^T d hv ; then [d bottleneck | d raw_density] through [We ; Wd]^T
        BFrag<NP> be[KS_H + 1];
        accv acc[NT_H];
        {
            chain_gemm<NP, KS_C, NT_H, false, true>(R, bc, acc, lane, wx(B_VIEW) + bex);
            TE* bt = TP(a.d8_t) + T.blk * (288 * TILE) + T.lo;
            store_t<NT_H>(bt, acc);
            const float z = a.raw_den[T.rc * a.nc] + a.density_bias;
            const float add0 = (a.sdot && T.live) ? ch_sp_d2(z) * a.sdot[T.rc] : 0.f;
            float x[8];
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = QB * (j >> 2) + 4 * T.g + (j & 3);
                float v = 0.f;
                if (T.live && ch < a.nc) v = a.d_den[T.rc * a.nc + ch] + (ch == 0 ? add0 : 0.f);
                x[j] = v;
                bt[(256 + QB * (j >> 2) + (j & 3)) * TILE] = (TE)v;
                m = fmaxf(m, fabsf(v));
            }
            if constexpr (NP == 2) {  // one exponent for the whole operand, separate maxima for the two T tensors
                e = ex_of(lane_amax<NT_H>(acc), EXP_CAP_Z);
                const Ex ed = ex_of(m, EXP_CAP_Z);
                RM.upd(8, e.top);
                RM.upd(9, ed.top);
                bex = e.ex < ed.ex ? e.ex : ed.ex;
            }
            split_acc<NP, NT_H, KS_H + 1>(acc, be, bex);
            split_into<NP>(x, be[KS_H], bex);
            if (TP(a.coef_t)) {
                TE* ct = TP(a.coef_t) + T.blk * (32 * TILE) + T.lo;
                const float cf = T.live ? ch_sp_d1(z) : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = QB * (j >> 2) + 4 * T.g + (j & 3);
                    ct[(QB * (j >> 2) + (j & 3)) * TILE] = (TE)(f == 0 ? cf : 0.f);
                }
            }
        }
        BFrag<NP> bh[KS_H];
        chain_gemm<NP, KS_H + 1, NT_H, false, true>(R, be, acc, lane, wx(B_EXTRA) + bex);
        e = finish_gated<NP>(acc, pop_front(mk), TP(a.delta_t) + (int64_t)7 * Mp * 256, T, a.q24 && q24_delta(7), bh);
        bex = e.ex;
        RM.upd(7, e.top);
#pragma unroll 1
        for (int l = 7; l >= 1; --l) {
            chain_gemm<NP, KS_H, NT_H, false, true>(R, bh, acc, lane, wx(B_L7 + 7 - l) + bex);
            e = finish_gated<NP>(acc, pop_front(mk), TP(a.delta_t) + (int64_t)(l - 1) * Mp * 256, T, a.q24 && q24_delta(l - 1), bh);
            bex = e.ex;
            RM.upd(l - 1, e.top);
        }
        if (a.d_mean) {  // uniform: d enc over [delta_0 | delta_5], then the encoding's adjoint
            float mu[3], cv[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                mu[i] = a.mean[T.rc * 3 + i];
                cv[i] = a.cov[T.rc * 3 + i];
            }
            accv a3[NT_ENC];
            chain_gemm<NP, KS_H, NT_ENC, false, true>(R, bh, a3, lane, wx(B_DENC0) + bex);
            bex = reload_b<NP, KS_H>(TP(a.delta_t) + (int64_t)5 * Mp * 256 + T.blk * (256 * TILE) + T.lo, bh, EXP_CAP_Z);
            chain_gemm<NP, KS_H, NT_ENC, false, false>(R, bh, a3, lane, wx(B_DENC1) + bex);
            float dm[3];
            ipe_backward_tiles(a3, mu, cv, T.g, dm);
            if (T.live && T.g == 0) {
                a.d_mean[T.row * 3] = dm[0];
                a.d_mean[T.row * 3 + 1] = dm[1];
                a.d_mean[T.row * 3 + 2] = dm[2];
            }
        }
    }
    R.drain();
    if constexpr (NP == 2) {
        RM.flush(a.amax ? a.amax + AM_DELTA0 : nullptr, lane, wid, lds);
        if (a.amax && blockIdx.x == 0 && tid == 0) a.amax[AM_COEF] = 0x3f800000u;  // softplus' <= 1
    }
}

// ------------------------------------------------------------------------------------- weight gradients (TN GEMM)
// dW[N1 x N2] (+)= sum over samples X[s][i] * Y[s][j] with X, Y in the T layout (sample-minor), on the bf16 matrix
// cores: A = X^T (k = sample), B = Y.  A workgroup owns the whole [TMW x TNW] result for a contiguous range of
// 16-sample half blocks (split over samples; per-workgroup slabs are reduced afterwards).  Staging: fp32 from HBM to
// registers (full 64-B half rows, coalesced), split ONCE per element into NP bf16 planes, 8-B LDS writes into
// [plane][feature][16 samples] images (the two 16-B pieces of a row swapped on features with bit 3 set: conflict-free
// for the writes and for the ds_read_b128 fragment reads); two LDS buffers, one barrier per half block.  Row sums of X
// (bias gradients) are accumulated from the fp32 values on the way in.
// (Tried: one 256 x 352 tile for layer 5 over [h4 | enc] and one 288 x 256 tile for [extra ; density] over h7, so that
// delta_5 and h7 are read once - 12 accumulator tiles per wave at two waves per SIMD spill 180-280 bytes per lane and the
// weight gradients of an evaluation took 7.2 ms instead of 5.0: the re-reads, 3 GB per step, stay.
// Also tried on the 256 x 256 tile (4.85 ms per evaluation as it stands): staging half block h + 1 piecewise between the
// matrix products of half block h instead of in its own phase in front of the barrier (5.10 ms: the issue port is shared,
// the loads go out later); refilling each piece's registers as soon as it is converted (6.3 ms: the waits degrade to
// vmcnt(0)); four register sets in flight instead of three (4.88 ms: depth is not the limit).  In fact hipcc drains the
// prefetched sets in front of every staging phase (vmcnt(3), (2), (1), (0): it cannot count the younger loads behind the
// conditional `load`); a condition-free steady-state loop with a conditional tail gets vmcnt(13)..(10) - and runs 4.67 ms
// against 4.57 (two sets; three spill): the loop is bound by its conversion + product instruction time, not by latency.
// Round 3: the second half of the waves ONE barrier interval behind the first (same code, three LDS buffers, so that on every
// SIMD one wave stages while its partner multiplies): 6.0-6.2 ms against 4.5 (5.3 with s_setprio around the products) - with
// one multiplying wave per SIMD nothing covers the fragment-read latency at the head of every interval; a barrier that
// waits for LDS traffic only (not vmcnt) in the loop as it stands: no change (profiles/r03_experiments.txt).)
struct WSeg {
    const float* X;  // feature 0 of the X sub-range in block 0
    const float* Y;
    int64_t nhalf;   // 16-sample half blocks
    int FX, FY;      // features per block of the tensors X / Y live in
    int bias;        // rows of this segment count towards the row sums of X (the second-order rows do not)
    const uint32_t* ax;  // NP = 2: largest |x| of the whole X / Y tensor (float bits, written by the chain kernels): the
    const uint32_t* ay;  // segment's operands are scaled by ONE power of two each (the sum runs over all samples)
};
struct WgArgs {
    WSeg seg[4];
    int nseg;
    int64_t half_total, per;
    float* slab;
    int64_t slab_stride;
    int bias;
};
// ONE launch runs up to WG_MAXJ jobs of the same tile configuration and operand format (grid.y = job): the six 256 x 256 trunk
// layers whose operands are both Q24, or layer 0 and the skip columns of layer 5.  Every job gets 1 / n of the CUs and n times the
// sample range per workgroup: the same parallelism with 1 / n of the launches, of the slab bytes (a workgroup writes its 257 KB of
// partial sums once per n times as many half blocks) and of the slab reductions - a job costs ~40 us whatever its size (prologue,
// slab write, reduction launch), 18 % of the 512-ray step in fifteen separate launches.
constexpr int WG_MAXJ = 8;
struct WgMulti {
    WgArgs job[WG_MAXJ];
};

template <int NP>
__device__ __forceinline__ f32x16 mfma_split32(const BFrag<NP>& a, const BFrag<NP>& b, f32x16 v) {
    if constexpr (NP == 3) {  // small terms first
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b.p[0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[2], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[1], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[1], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], v, 0, 0, 0);
    } else if constexpr (NP == 2) {
        v = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.p[1], b.p[0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.p[0], b.p[1], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.p[0], b.p[0], v, 0, 0, 0);
    } else {
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], v, 0, 0, 0);
    }
    return v;
}
// the T tensors are read once per GEMM: non-temporal loads (4.70 -> 4.57 ms for the GEMMs of one evaluation)
#define WG_LD(p) __builtin_nontemporal_load(p)
// The three 16-byte loads of a Q24 unit (48 bytes, stride 48 across the lanes): each instruction touches a third of every 128-byte
// line of the wave's 3 KB.  As NON-temporal loads each of the three fetched its lines from L2 on its own - TCP_TCC_READ_REQ 7.63e7 per
// launch of the 256 x 256 tile against 2.57e7 TCC_EA0_RDREQ, two L2 hits per miss; as plain loads the second and third hit the lines
// the first brought into the vector L1: 2.52e7 requests, no L2 hits (profiles/r04_wgrad_unit_loads.txt).  The kernel itself is no
// faster (761 us under the counters either way: it is not bound by L2 requests), the training step 0.65 % (same box, twice).
#ifndef PN_WG_UNIT_NT
#define PN_WG_UNIT_NT 0
#endif
#if PN_WG_UNIT_NT
#define WG_LDU(p) __builtin_nontemporal_load(p)
#else
#define WG_LDU(p) (*(p))
#endif
// Y24 / X24: the operand tensor is stored in Q24 (see pack_q24): its work item is a UNIT of 48 contiguous bytes - four features of
// four samples - unpacked to fp32 (one byte permute per element) and transposed in registers into four (feature, 4 samples) pieces.
// With both operands in Q24 the X and Y units form one list over the threads (one unit per thread on the 256 x 256 tile).  The LDS
// image then holds feature f in row (f & ~3) | ((f + (f >> 2)) & 3): a unit's four writes go to rows 4 qb + f for a fixed f across
// the wave, which in the plain image are 128 bytes apart - the same eight banks sixteen times.
template <int NP, int TM, int TN, int WM, int WN, bool X24 = false, bool Y24 = false>
__global__ __launch_bounds__(64 * WM * WN) void k_chain_wgrad(WgMulti multi) {
    const WgArgs& a = multi.job[blockIdx.y];
    static_assert(!(X24 || Y24) || (NP == 2 && TILE == 16), "Q24 tensors exist in fp16-pair builds with 16-sample tiles");
    static_assert(!X24 || Y24, "combinations in use: (0,0), (0,1), (1,1)");
    constexpr bool ROT = X24 || Y24;  // row permutation of the LDS images
    constexpr int NTH = 64 * WM * WN, TMW = 32 * TM * WM, TNW = 32 * TN * WN;
    constexpr int PX = TMW * 16, PY = TNW * 16;     // bf16 elements per plane
    constexpr int BUF = NP * (PX + PY);             // per buffer
    typedef typename TEl<NP>::type TE;              // element type of the T tensors (bf16 for NP = 1: staging is a copy)
    constexpr int PPR = 16 * (int)sizeof(TE) / 16;  // 16-B pieces per row of 16 samples: 4 (fp32) or 2 (bf16)
    constexpr int SPP = 16 / PPR;                   // samples per piece
    constexpr int CX = X24 ? 0 : TMW * PPR, CY = Y24 ? 0 : TNW * PPR;   // 16-B pieces per half block (fp32 / bf16 operands)
    constexpr int LX = X24 ? 1 : (CX + NTH - 1) / NTH, LY = Y24 ? 1 : (CY + NTH - 1) / NTH;  // (1: a dummy register)
    constexpr int UX = X24 ? TMW : 0, UY = Y24 ? TNW : 0;               // Q24 units per half block
    constexpr int LU = (UX + UY + NTH - 1) / NTH;                       // units per thread (X units first, then Y units)
    static_assert(UX % 64 == 0, "a wave's units are all X or all Y");
    // Q24 units: a wave LOADS its 64 units (3 KB contiguous) as three fully coalesced 1-KB instructions - lane i takes the 16-byte
    // pieces i, i + 64, i + 128 of the chunk - and turns pieces into units (unit i = pieces 3 i .. 3 i + 2) through a wave-private
    // 3-KB scratch in LDS when it stages them (PN_WG_UNIT_COAL; both access patterns are conflict-free: contiguous 16-byte writes,
    // 16-byte reads at a stride of 48 bytes).  Loading a unit as three 16-byte pieces of its own - lanes 48 bytes apart, every
    // instruction touching a third of each of 24 lines - streamed at 2.8 TB/s with nothing else in the kernel (timing ablations,
    // profiles/r04_wgrad_timing_ablations.txt: 570 us for 1.61 GB, the same with or without staging and barriers, 824 us for the
    // whole kernel), against 5.8 - 6.0 TB/s for the tiles whose loads are contiguous across the lanes.
#ifndef PN_WG_UNIT_COAL
#define PN_WG_UNIT_COAL 0
#endif
    constexpr bool COAL = PN_WG_UNIT_COAL && (X24 || Y24);
    constexpr int SCR = COAL ? (NTH / 64) * 1536 : 0;  // unsigned shorts: 3 KB per wave
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BUF + SCR];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int64_t h0 = (int64_t)blockIdx.x * a.per;
    if (h0 >= a.half_total) return;  // (uniform: a job of a multi-job launch with fewer workgroups than grid.x; no slab of its own)
    int64_t h1 = h0 + a.per;
    if (h1 > a.half_total) h1 = a.half_total;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bsum[LX];
#pragma unroll
    for (int i = 0; i < LX; ++i) bsum[i] = 0.f;
    float bsum4[LU > 0 ? LU : 1][4] = {};  // X24: row sums of a unit's four features

    // NSET register sets: the loads of half block h + NSET are issued when h's set has been staged, so they have NSET
    // half blocks to land (one workgroup per CU: nothing else hides the HBM latency).  Three where the registers allow:
    // with three products per fp32 product the kernel is HBM-bound and two sets keep only 32-64 KB per CU in flight.
#ifndef PN_WG_Q24_NSET
#define PN_WG_Q24_NSET 4
#endif
#ifndef PN_WG_Q24_ALT  // 1: the two waves of a SIMD take staging and products in opposite orders (see the loop)
#define PN_WG_Q24_ALT 1
#endif
#ifndef PN_WG_Q24_STEADY  // 1: whole trips without a condition (see the loop); measured slower in the training step, off
#define PN_WG_Q24_STEADY 0
#endif
    // (both operands in Q24: a thread holds ONE 48-byte unit per set - four sets are the 48 registers of three fp32 sets)
    constexpr int NSET = (X24 && LU == 1) ? PN_WG_Q24_NSET : ((NP <= 2 && (X24 ? 0 : LX) + (Y24 ? 0 : LY) + 3 * LU <= 4) ? 3 : 2);
    f32x4 xr[NSET][LX], yr[NSET][LY];
    f32x4 ur[NSET][LU > 0 ? LU : 1][3];  // a unit: 48 bytes
    // the 256 x 256 tile on Q24 operands (one or both) and on bf16 tensors: measured same box, us per launch in the training step,
    // common -> alternating order: both in Q24 821 -> 783, Y in Q24 892 -> 832, bf16 tensors 426 -> 358; NOT on fp32 tensors with
    // fp16 pairs (589 -> 644) or the six-product split (1230 -> 1258)
    constexpr bool ALT = PN_WG_Q24_ALT && (WM * WN == 8 && TM == 2 && TN == 4) && (X24 || Y24 || NP == 1);
    // (Measured and dropped in the alternating loop: the two halves of the workgroup swapping the X units - whose row sums cost a
    // wave ~500 cycles more per half block - with the parity of the half block: 828 us per launch with or without.)
    auto unit_of = [&](int i) { return tid + NTH * i; };
    float bw[NSET] = {};  // bias weight of the half block held in each set
    // NP = 2: ONE unit for the whole job, 2^unit = the scale of every product in the accumulators: the smallest sx + sy over
    // the job's segments (the segment with the LARGEST products).  A segment whose own exponents add up to more is scaled
    // down by the difference (its products are that many binades below the dominant segment's; what falls below fp16's
    // range there is below 2^-39 of the dominant products) - no accumulator is ever re-based, so sums cannot overflow
    // however far the segments' magnitudes are apart, and the result does not depend on where a workgroup's range starts.
    int sx[NSET] = {}, sy[NSET] = {}, unit = 0;
    auto seg_exps = [&](const WSeg& S, int& ex, int& ey) {
        ex = scale_exp(__uint_as_float(*S.ax), EXP_CAP_Z);
        ey = scale_exp(__uint_as_float(*S.ay), EXP_CAP_Z);
        int over = ex + ey - unit;                      // >= 0
        const int rx = over < ex + 126 ? over : ex + 126;  // pow2f takes -126 .. 127
        ex -= rx;
        ey -= over - rx;
        if (ey < -126) ey = -126;                       // (everything of the segment has long been flushed to zero)
    };
    if constexpr (NP == 2) {
        unit = 1 << 20;
        for (int i = 0; i < a.nseg; ++i) {
            const int e = scale_exp(__uint_as_float(*a.seg[i].ax), EXP_CAP_Z) + scale_exp(__uint_as_float(*a.seg[i].ay), EXP_CAP_Z);
            unit = e < unit ? e : unit;
        }
    }
    // Segment cursor of the loads (half blocks are loaded in increasing order): the segment's pointers, widths and
    // exponents are fetched when the range crosses into it, not per half block - two dependent scalar loads in front of
    // every half block's global loads otherwise.
    // (Do NOT force the cursor into scalar registers with v_readfirstlane: hipcc then waits s_waitcnt vmcnt(0) - for every operand
    // load in flight - in front of every staging instead of the counted vmcnt(9) / (11) of the alternating loop; as written the
    // segment's fields come by scalar loads and only the segment switch, which reads the two tensor maxima, drains.)
    int csg = 0, cFX = a.seg[0].FX, cFY = a.seg[0].FY, csx = 0, csy = 0;
    int64_t cbase = 0, cend = a.seg[0].nhalf;
    const TE* cX = reinterpret_cast<const TE*>(a.seg[0].X);
    const TE* cY = reinterpret_cast<const TE*>(a.seg[0].Y);
    float cbw = a.seg[0].bias ? 1.f : 0.f;
    if constexpr (NP == 2) seg_exps(a.seg[0], csx, csy);
    auto load = [&](int64_t h, int set) __attribute__((always_inline)) {
        while (csg + 1 < a.nseg && h >= cend) {  // (uniform; the last segment takes what is left)
            ++csg;
            const WSeg& S = a.seg[csg];
            cbase = cend;
            cend += S.nhalf;
            cX = reinterpret_cast<const TE*>(S.X);
            cY = reinterpret_cast<const TE*>(S.Y);
            cFX = S.FX;
            cFY = S.FY;
            cbw = S.bias ? 1.f : 0.f;
            if constexpr (NP == 2) seg_exps(S, csx, csy);
        }
        const int64_t hb = h - cbase;
        bw[set] = cbw;
        if constexpr (NP == 2) {
            sx[set] = csx;
            sy[set] = csy;
        }
        const int64_t blk = hb / (TILE / 16);  // a T-layout sample block holds TILE / 16 half blocks
        const int half = (int)(hb % (TILE / 16));
        const TE* xb = cX + blk * ((int64_t)cFX * TILE) + half * 16;
        const TE* yb = cY + blk * ((int64_t)cFY * TILE) + half * 16;
        if constexpr (!X24) {
#pragma unroll
            for (int i = 0; i < LX; ++i) {
                const int idx = tid + NTH * i;
                if (CX % NTH == 0 || idx < CX) xr[set][i] = WG_LD(reinterpret_cast<const f32x4*>(xb + (idx / PPR) * TILE + (idx % PPR) * SPP));
            }
        }
        if constexpr (!Y24) {
#pragma unroll
            for (int i = 0; i < LY; ++i) {
                const int idx = tid + NTH * i;
                if (CY % NTH == 0 || idx < CY) yr[set][i] = WG_LD(reinterpret_cast<const f32x4*>(yb + (idx / PPR) * TILE + (idx % PPR) * SPP));
            }
        }
        if constexpr (X24 || Y24) {  // (TILE 16: a block is one half block; a Q24 block of F features is F * 48 bytes, a unit 48)
            const unsigned char* xq = reinterpret_cast<const unsigned char*>(cX) + blk * ((int64_t)cFX * 48);
            const unsigned char* yq = reinterpret_cast<const unsigned char*>(cY) + blk * ((int64_t)cFY * 48);
#pragma unroll
            for (int i = 0; i < LU; ++i) {
                const int u = unit_of(i);
                if ((UX + UY) % NTH == 0 || u < UX + UY) {
                    if constexpr (COAL) {  // the wave's 64 units as one contiguous 3-KB chunk: pieces lane, lane + 64, lane + 128
                        const int u0 = u - lane;
                        const f32x4* p = reinterpret_cast<const f32x4*>(u0 < UX ? xq + u0 * 48 : yq + (u0 - UX) * 48) + lane;  // (wave-uniform base)
                        ur[set][i][0] = WG_LD(p);
                        ur[set][i][1] = WG_LD(p + 64);
                        ur[set][i][2] = WG_LD(p + 128);
                    } else {
                        const f32x4* p = reinterpret_cast<const f32x4*>(u < UX ? xq + u * 48 : yq + (u - UX) * 48);  // (wave-uniform)
                        ur[set][i][0] = WG_LDU(p);
                        ur[set][i][1] = WG_LDU(p + 1);
                        ur[set][i][2] = WG_LDU(p + 2);
                    }
                }
            }
        }
    };
    auto put = [&](unsigned short* plane0, int pstride, int idx, const f32x4& v, int ex) {
        if constexpr (NP == 1) {  // the piece already holds 8 bf16 samples of one feature: copy
            const int f = idx >> 1, q = idx & 1;
            *reinterpret_cast<f32x4*>(plane0 + f * 16 + ((q ^ ((f >> 3) & 1)) << 3)) = v;
            return;
        }
        const int f0 = idx >> 2, q = idx & 3;
        const int f = ROT ? ((f0 & ~3) | ((f0 + (f0 >> 2)) & 3)) : f0;
        const int o = f * 16 + (((q >> 1) ^ ((f >> 3) & 1)) << 3) + (q & 1) * 4;
        if constexpr (NP == 2) {
            const float s = pow2f(ex);
            const float x4[4] = {v[0], v[1], v[2], v[3]};
            uint32_t h2[2], l2[2];
            split2_pairs<2>(x4, s, h2, l2);
            *reinterpret_cast<uint2*>(plane0 + o) = make_uint2(h2[0], h2[1]);
            *reinterpret_cast<uint2*>(plane0 + pstride + o) = make_uint2(l2[0], l2[1]);
        } else {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 hv, mv, lv;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const __bf16 hb = (__bf16)v[c];
                hv[c] = hb;
                if constexpr (NP == 3) {
                    const float r1 = v[c] - (float)hb;
                    const __bf16 mb = (__bf16)r1;
                    mv[c] = mb;
                    lv[c] = (__bf16)(r1 - (float)mb);
                }
            }
            *reinterpret_cast<bf16x4*>(plane0 + o) = hv;
            if constexpr (NP == 3) {
                *reinterpret_cast<bf16x4*>(plane0 + pstride + o) = mv;
                *reinterpret_cast<bf16x4*>(plane0 + 2 * pstride + o) = lv;
            }
        }
    };
    // a Q24 unit -> four (feature, 4 samples) pieces of the image at `plane0`; `bs`: the unit's row sums (X with bias) or null
    auto put_unit = [&](unsigned short* plane0, int pstride, int u, const f32x4 (&r)[3], int ex, float* bs, float w) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        uint32_t d[12];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const u32x4 t = __builtin_bit_cast(u32x4, r[k]);
#pragma unroll
            for (int c = 0; c < 4; ++c) d[4 * k + c] = t[c];
        }
        float e[4][4];  // [sample][feature]
#pragma unroll
        for (int sm = 0; sm < 4; ++sm) {
            e[sm][0] = __uint_as_float(d[3 * sm] << 8);
            e[sm][1] = __uint_as_float(__builtin_amdgcn_perm(d[3 * sm + 1], d[3 * sm], 0x0504030cu));
            e[sm][2] = __uint_as_float(__builtin_amdgcn_perm(d[3 * sm + 2], d[3 * sm + 1], 0x0403020cu));
            e[sm][3] = __uint_as_float(d[3 * sm + 2] & 0xffffff00u);
        }
        const int qb = u >> 2, sg = u & 3;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const f32x4 v{e[0][f], e[1][f], e[2][f], e[3][f]};
            put(plane0, pstride, (4 * qb + f) * 4 + sg, v, ex);
            if (bs) bs[f] += w * ((v[0] + v[1]) + (v[2] + v[3]));
        }
    };
#ifndef PN_ABL_WG  // timing ablations of the weight-gradient tile (wrong results; never in the shipped build): bit 0 no matrix
#define PN_ABL_WG 0  // products, bit 1 no staging (the loaded registers are only consumed), bit 2 no barriers
#endif
    auto stage = [&](int buf, int set) __attribute__((always_inline)) {
        unsigned short* xs = smem + buf * BUF;
        unsigned short* ys = xs + NP * PX;
        if constexpr (PN_ABL_WG & 2) {  // keep the loads alive, convert nothing
            if constexpr (X24 || Y24) {
#pragma unroll
                for (int i = 0; i < LU; ++i) asm volatile("" ::"v"(ur[set][i][0]), "v"(ur[set][i][1]), "v"(ur[set][i][2]));
            }
            if constexpr (!X24) {
#pragma unroll
                for (int i = 0; i < LX; ++i) asm volatile("" ::"v"(xr[set][i]));
            }
            if constexpr (!Y24) {
#pragma unroll
                for (int i = 0; i < LY; ++i) asm volatile("" ::"v"(yr[set][i]));
            }
            return;
        }
        if constexpr (X24 || Y24) {
#pragma unroll
            for (int i = 0; i < LU; ++i) {
                const int u = unit_of(i);
                if ((UX + UY) % NTH == 0 || u < UX + UY) {
                    if constexpr (COAL) {  // pieces -> this lane's unit, through the wave's scratch (LDS operations of a wave are in order)
                        f32x4* scr = reinterpret_cast<f32x4*>(smem + 2 * BUF) + wid * 192;
                        scr[lane] = ur[set][i][0];
                        scr[lane + 64] = ur[set][i][1];
                        scr[lane + 128] = ur[set][i][2];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        f32x4 un[3];
                        un[0] = scr[3 * lane];
                        un[1] = scr[3 * lane + 1];
                        un[2] = scr[3 * lane + 2];
                        __builtin_amdgcn_wave_barrier();  // (the next unit's pieces overwrite the scratch only behind these reads)
                        if (u < UX) put_unit(xs, PX, u, un, sx[set], bsum4[i], bw[set]);  // (wave-uniform)
                        else put_unit(ys, PY, u - UX, un, sy[set], nullptr, 0.f);
                    } else {
                        if (u < UX) put_unit(xs, PX, u, ur[set][i], sx[set], bsum4[i], bw[set]);  // (wave-uniform)
                        else put_unit(ys, PY, u - UX, ur[set][i], sy[set], nullptr, 0.f);
                    }
                }
            }
        }
        if constexpr (!X24)
#pragma unroll
        for (int i = 0; i < LX; ++i) {
            const int idx = tid + NTH * i;
            if (CX % NTH == 0 || idx < CX) {
                put(xs, PX, idx, xr[set][i], sx[set]);
                if constexpr (NP == 1) {
                    const bf16x8 hv = __builtin_bit_cast(bf16x8, xr[set][i]);
                    float t = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) t += (float)hv[c];
                    bsum[i] += bw[set] * t;
                } else {
                    bsum[i] += bw[set] * ((xr[set][i][0] + xr[set][i][1]) + (xr[set][i][2] + xr[set][i][3]));
                }
            }
        }
        if constexpr (!Y24)
#pragma unroll
        for (int i = 0; i < LY; ++i) {
            const int idx = tid + NTH * i;
            if (CY % NTH == 0 || idx < CY) put(ys, PY, idx, yr[set][i], sy[set]);
        }
    };
    const int fr = lane & 31, fh = lane >> 5;
    auto frag = [&](const unsigned short* plane, int feature) {
        if constexpr (ROT) feature = (feature & ~3) | ((feature + (feature >> 2)) & 3);
        return *reinterpret_cast<const typename PlaneOf<NP>::type*>(plane + feature * 16 + ((fh ^ ((feature >> 3) & 1)) << 3));
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
        if constexpr (PN_ABL_WG & 1) return;
        const unsigned short* xs = smem + buf * BUF;
        const unsigned short* ys = xs + NP * PX;
        // The Y fragments of tile column j + 1 are read BEFORE the products of column j, by inline asm with hand-counted waits.  As one
        // fragment set re-used per column (round 3) every column's products waited for an LDS round trip with the matrix pipe idle -
        // four bubbles of 150 - 200 cycles in a wave's 768 cycles of products per half block, in the phase in which its SIMD partner
        // is staging and cannot fill them: the 256 x 256 tile ran 3300 cycles per half block against 1536 of matrix pipe, with its
        // loads long landed (timing ablations: the loads alone stream at 5.7 TB/s, the kernel moved 3.9:
        // profiles/r04_wgrad_timing_ablations.txt).  Written as plain C++ with two fragment sets hipcc still allocates ONE and
        // issues each read behind the last product that uses the old value, one instruction in front of its wait.
        // NOT on the tile with both operands in Q24: its four register sets of units leave no room for a second fragment set
        // (12 registers spilled: 950 us per launch against 739; with three sets 744 - no gain either way: that tile's critical
        // path is the staging of its X units, section 14 of profiles/r03_experiments.txt).  Measured on the others, same box, us
        // per launch in the training step: fp32 tensors 596 -> 545, Y in Q24 788 -> 752 (profiles/r04_wgrad_fragment_prefetch.txt).
#ifndef PN_WG_X24_PREFETCH  // (experiment switch: the prefetching form on the Q24 tile too; needs PN_WG_Q24_NSET=3 to fit)
#define PN_WG_X24_PREFETCH 0
#endif
        if constexpr (X24 && !PN_WG_X24_PREFETCH) {
            BFrag<NP> af[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) af[i].p[p] = frag(xs + p * PX, 32 * (wm * TM + i) + fr);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                BFrag<NP> bf;
#pragma unroll
                for (int p = 0; p < NP; ++p) bf.p[p] = frag(ys + p * PY, 32 * (wn * TN + j) + fr);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = mfma_split32<NP>(af[i], bf, acc[i][j]);
            }
            return;
        }
        typedef typename PlaneOf<NP>::type Frag;
        auto lds_of = [](const unsigned short* q) { return (uint32_t)(uintptr_t)(lds_ptr_t)q; };
        auto fidx = [&](int feature) {  // element index of this lane's fragment of `feature` in a plane (see frag)
            if constexpr (ROT) feature = (feature & ~3) | ((feature + (feature >> 2)) & 3);
            return feature * 16 + ((fh ^ ((feature >> 3) & 1)) << 3);
        };
        // 32 more features are 512 more elements whatever the lane (the row permutation and the 16-byte swap act on fr alone)
        const uint32_t xa = lds_of(xs) + 2 * fidx(32 * (wm * TM) + fr), ya = lds_of(ys) + 2 * fidx(32 * (wn * TN) + fr);
        // (offsets as immediates: one address register per operand instead of one per read)
        auto rd = [](Frag& dst, uint32_t addr, auto off) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(off)::value) : "memory");
        };
        static_assert(2 * NP * PX + 1024 * TM < 65536 && 2 * NP * PY + 1024 * TN < 65536, "16-bit offset field of ds_read_b128");
        BFrag<NP> af[TM];
        BFrag<NP> bf[2];
        auto read_b = [&](auto jc, BFrag<NP>& f) {  // Y fragments of tile column j
            constexpr int J = decltype(jc)::value;
            static_for<NP>([&](auto pc) {
                constexpr int P = decltype(pc)::value;
                rd(f.p[P], ya, std::integral_constant<int, 1024 * J + 2 * P * PY>{});
            });
        };
        // all but the newest `left` reads have returned; ties the registers the products read to the wait
        auto settle = [&](auto left, BFrag<NP>& f) {
            constexpr int L = decltype(left)::value;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (p == 0) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f.p[p]) : "n"(L) : "memory");
                else asm volatile("" : "+v"(f.p[p]));
            }
        };
        read_b(std::integral_constant<int, 0>{}, bf[0]);
        static_for<TM>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            static_for<NP>([&](auto pc) {
                constexpr int P = decltype(pc)::value;
                rd(af[I].p[P], xa, std::integral_constant<int, 1024 * I + 2 * P * PX>{});
            });
        });
        if constexpr (TN > 1) {
            read_b(std::integral_constant<int, 1>{}, bf[1]);
            settle(std::integral_constant<int, NP>{}, bf[0]);
        } else {
            settle(std::integral_constant<int, 0>{}, bf[0]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(af[i].p[p]));
        static_for<TN>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][J] = mfma_split32<NP>(af[i], bf[J & 1], acc[i][J]);
            if constexpr (J + 2 < TN) {  // column J + 2 into the set column J's products have just been issued from
                __builtin_amdgcn_sched_barrier(0);
                read_b(std::integral_constant<int, J + 2>{}, bf[J & 1]);
            }
            if constexpr (J + 1 < TN) {
                if constexpr (J + 2 < TN) settle(std::integral_constant<int, NP>{}, bf[(J + 1) & 1]);
                else settle(std::integral_constant<int, 0>{}, bf[(J + 1) & 1]);
            }
        });
    };
#ifdef PN_TRACE_WG  // debug build only: phase times of workgroup 0 (every wave), summed over its half blocks
    unsigned long long tw[4] = {0, 0, 0, 0};
#define WGT(i, expr)                                                   \
    do {                                                               \
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime();   \
        expr;                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
        tw[i] += __builtin_amdgcn_s_memtime() - t0_;                   \
    } while (0)
#else
#define WGT(i, expr) expr
#endif
    // half block h + K of a trip: register set K % NSET, LDS buffer K % 2 (static indices)
    auto one = [&](auto kc, int64_t h) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        if (h + K < h1) {  // (uniform)
            WGT(0, stage(K % 2, K % NSET));
            WGT(1, __syncthreads());
            WGT(2, if (h + K + NSET < h1) load(h + K + NSET, K % NSET));
            WGT(3, compute(K % 2));
        }
    };
    if constexpr (ALT) {
        // ALTERNATING ORDER.  In the common order (below) both waves of a SIMD stage, then both multiply; the phase trace
        // (tools/experiments/trace_wgrad.py) showed the second-dispatched wave of every SIMD losing the arbitration for the matrix pipe,
        // finishing its products last and only then starting to stage: the pipe idle for half of every half block.  Here wave
        // type t (0: waves 0 .. NW/2 - 1, 1: their SIMD partners) runs   products(K) ; stage(K + 1 + t) ; refill   per half block
        // K - ONE instruction sequence - with its barrier behind the products (t = 1) or behind the staging (t = 0): after a
        // barrier one wave of a SIMD multiplies while its partner converts, then they swap.  The partner is one half block
        // ahead in its staging (it staged half block 1 in the prologue), its LDS buffer index the only run-time difference; the
        // register set of a staging is (K + 1) % NSET for both, the loads are unconditional (past the end: the last half block
        // again, never staged) so that hipcc counts them.  Hazards: a wave stages into buffer b only behind a barrier that
        // everyone passed after its products from b; products from b start behind a barrier everyone passed after staging b.
        const int t = wid >= (WM * WN) / 2 ? 1 : 0;  // (wave-uniform)
        const int64_t n = h1 - h0;
        auto clampd = [&](int64_t hb) { return h0 + (hb < n ? hb : n - 1); };
        if (n > 0) {
            load(h0, 0);
            stage(0, 0);
            load(clampd(1), 0);          // (type 0 does not need it: issued by every wave so that hipcc's count of the loads in
            if (t && 1 < n) stage(1, 0);  //  flight is the same on every path into the loop)
#pragma unroll
            for (int sidx = 1; sidx < NSET; ++sidx) load(clampd(sidx + t), sidx);
            load(clampd(NSET + t), 0);
            __syncthreads();
            constexpr int TRIPA = NSET == 3 ? 6 : (NSET == 4 ? 4 : 2);
            auto alt = [&](auto kc, auto guard, int64_t k0) __attribute__((always_inline)) {
                constexpr int K = decltype(kc)::value;
                if (!decltype(guard)::value || k0 + K < n) {  // (uniform)
                    WGT(3, compute(K % 2));
                    if (t && !(PN_ABL_WG & 4)) WGT(1, __syncthreads());
                    const int64_t hb = k0 + K + 1 + t;
                    if (hb < n) WGT(0, stage((K + 1 + t) & 1, (K + 1) % NSET));
                    WGT(2, load(clampd(hb + NSET), (K + 1) % NSET));
                    if (!t && !(PN_ABL_WG & 4)) WGT(1, __syncthreads());
                }
            };
            int64_t k0 = 0;
            for (; k0 + TRIPA <= n; k0 += TRIPA) {  // whole trips: no condition around a half block (hipcc counts the loads)
                alt(std::integral_constant<int, 0>{}, std::false_type{}, k0);
                alt(std::integral_constant<int, 1>{}, std::false_type{}, k0);
                if constexpr (TRIPA >= 4) {
                    alt(std::integral_constant<int, 2>{}, std::false_type{}, k0);
                    alt(std::integral_constant<int, 3>{}, std::false_type{}, k0);
                }
                if constexpr (TRIPA == 6) {
                    alt(std::integral_constant<int, 4>{}, std::false_type{}, k0);
                    alt(std::integral_constant<int, 5>{}, std::false_type{}, k0);
                }
            }
            if (k0 < n) {  // the last, partial trip
                alt(std::integral_constant<int, 0>{}, std::true_type{}, k0);
                alt(std::integral_constant<int, 1>{}, std::true_type{}, k0);
                if constexpr (TRIPA >= 4) {
                    alt(std::integral_constant<int, 2>{}, std::true_type{}, k0);
                    alt(std::integral_constant<int, 3>{}, std::true_type{}, k0);
                }
                if constexpr (TRIPA == 6) {
                    alt(std::integral_constant<int, 4>{}, std::true_type{}, k0);
                    alt(std::integral_constant<int, 5>{}, std::true_type{}, k0);
                }
            }
        }
    } else
    if (h0 < h1) {
#pragma unroll
        for (int k = 0; k < NSET; ++k) {
            if constexpr (X24 && PN_WG_Q24_STEADY) load(h0 + k < h1 ? h0 + k : h1 - 1, k);  // (unconditional: see the steady-state loop below)
            else if (h0 + k < h1) load(h0 + k, k);
        }
        constexpr int TRIP = NSET == 3 ? 6 : (NSET == 4 ? 4 : 2);  // a multiple of the sets and of the two LDS buffers
        int64_t hs = h0;
        if constexpr (X24 && PN_WG_Q24_STEADY) {
            // Whole trips without a condition: every half block refills its set (past the end: the last half block again), so that
            // hipcc can COUNT the loads in flight - behind a conditional `load` it drains the prefetched sets in front of every
            // staging phase (vmcnt(2), (1), (0) with twelve loads in flight) where this form gets vmcnt(9) / (6).  Measured on one
            // box in the training step (multi-segment jobs of 1.9 M rows): 827 - 831 us per launch against 807 us for the
            // conditional form with four sets (861 with three) - the waits are not what the half block's time is made of.
            auto steady = [&](auto kc, int64_t h) __attribute__((always_inline)) {
                constexpr int K = decltype(kc)::value;
                stage(K % 2, K % NSET);
                __syncthreads();
                const int64_t hn = h + K + NSET;
                load(hn < h1 ? hn : h1 - 1, K % NSET);
                compute(K % 2);
            };
            for (; hs + TRIP <= h1; hs += TRIP) {
                steady(std::integral_constant<int, 0>{}, hs);
                steady(std::integral_constant<int, 1>{}, hs);
                if constexpr (TRIP >= 4) {
                    steady(std::integral_constant<int, 2>{}, hs);
                    steady(std::integral_constant<int, 3>{}, hs);
                }
                if constexpr (TRIP == 6) {
                    steady(std::integral_constant<int, 4>{}, hs);
                    steady(std::integral_constant<int, 5>{}, hs);
                }
            }
        }
        for (int64_t h = hs; h < h1; h += TRIP) {
            one(std::integral_constant<int, 0>{}, h);
            one(std::integral_constant<int, 1>{}, h);
            if constexpr (TRIP == 4) {
                one(std::integral_constant<int, 2>{}, h);
                one(std::integral_constant<int, 3>{}, h);
            }
            if constexpr (TRIP == 6) {
                one(std::integral_constant<int, 2>{}, h);
                one(std::integral_constant<int, 3>{}, h);
                one(std::integral_constant<int, 4>{}, h);
                one(std::integral_constant<int, 5>{}, h);
            }
        }
    }
#ifdef PN_TRACE_WG
    if (blockIdx.x == 0 && lane == 0 && (X24 && Y24)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g_chain_trace[8 * i + wid] = tw[i];
        if (wid == 0) g_chain_trace[32] = (unsigned long long)(h1 - h0);
    }
#endif
    float* out = a.slab + (int64_t)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * (wm * TM + i) + (e & 3) + 8 * (e >> 2) + 4 * fh;
                const float x = acc[i][j][e];
                out[(int64_t)row * TNW + 32 * (wn * TN + j) + fr] = NP == 2 ? ldexpf(x, -unit) : x;
            }
    if constexpr (X24) {
        if (a.bias) {
#pragma unroll
            for (int i = 0; i < LU; ++i) {
                const int u = tid + NTH * i;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    float v = bsum4[i][f];
                    v += __shfl_xor(v, 1, 64);
                    v += __shfl_xor(v, 2, 64);
                    if ((u & 3) == 0 && u < UX) out[(int64_t)TMW * TNW + 4 * (u >> 2) + f] = v;
                }
            }
        }
    } else if (a.bias) {
#pragma unroll
        for (int i = 0; i < LX; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 1, 64);
            if constexpr (PPR == 4) v += __shfl_xor(v, 2, 64);
            const int idx = tid + NTH * i;
            if (idx % PPR == 0 && (CX % NTH == 0 || idx < CX)) out[(int64_t)TMW * TNW + idx / PPR] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------- host side
static void fill_layer(PackLayer& L, int chunk0, int KS, int NT, int rows_valid) {
    L.chunk0 = chunk0; L.KS = KS; L.NT = NT; L.rows_valid = rows_valid;
    L.nseg = 0; L.aux_off = -1; L.aux_n = 0;
}
static void add_seg(PackLayer& L, int64_t off, int ld, int k0, int kvalid, int transposed, int col0) {
    L.seg[L.nseg++] = PackSeg{off, ld, k0, kvalid, transposed, col0};
}

template <int NP>
static PackTable fwd_table(int nc) {
    const PnLayout P = pn_layout(nc);
    PackTable T;
    T.n = F_COUNT;
    const int ld5 = PN_WIDTH + PN_ENC_DIM;
    for (int i = 0; i < F_COUNT; ++i) {
        PackLayer& L = T.L[i];
        const int KS = fwd_ks(i), NT = fwd_nt(i);
        fill_layer(L, fwd_chunk0<NP>(i), KS, NT, TILE * NT);
        if (i == F_L5) {  // hidden columns of layer 5; the skip columns follow as F_L5E, which completes the sum (+ bias)
            add_seg(L, P.w[5], ld5, 0, PN_WIDTH, 0, 0);
        } else if (i == F_L5E) {
            add_seg(L, P.w[5], ld5, 0, PN_ENC_DIM, 0, PN_WIDTH);
            L.aux_off = P.b[5]; L.aux_n = PN_WIDTH;
        } else if (i <= F_L7) {
            const int l = i < F_L5E ? i : i - 1;
            const int k = (l == 0) ? PN_ENC_DIM : PN_WIDTH;
            add_seg(L, P.w[l], k, 0, k, 0, 0);
            L.aux_off = P.b[l]; L.aux_n = PN_WIDTH;
        } else if (i == F_DEN) {
            L.rows_valid = nc;
            add_seg(L, P.wd, PN_WIDTH, 0, PN_WIDTH, 0, 0);
            L.aux_off = P.bd; L.aux_n = nc;
        } else if (i == F_EXTRA) {
            add_seg(L, P.we, PN_WIDTH, 0, PN_WIDTH, 0, 0);
            L.aux_off = P.be; L.aux_n = PN_WIDTH;
        } else if (i == F_VIEW) {
            L.rows_valid = PN_WIDTH_COND;
            add_seg(L, P.wv, PN_WIDTH + PN_VIEW_DIM, 0, PN_WIDTH + PN_VIEW_DIM, 0, 0);
            L.aux_off = P.bv; L.aux_n = PN_WIDTH_COND;
        } else {  // F_COLOR
            L.rows_valid = 3;
            add_seg(L, P.wc, PN_WIDTH_COND, 0, PN_WIDTH_COND, 0, 0);
            L.aux_off = P.bc; L.aux_n = 3;
        }
    }
    T.nchunks = fwd_chunk0<NP>(F_COUNT);
    return T;
}

template <int NP>
static PackTable bwd_table(int nc) {
    const PnLayout P = pn_layout(nc);
    PackTable T;
    T.n = B_COUNT;
    const int ldv = PN_WIDTH + PN_VIEW_DIM, ld5 = PN_WIDTH + PN_ENC_DIM;
    for (int i = 0; i < B_COUNT; ++i) {
        PackLayer& L = T.L[i];
        const int KS = bwd_ks(i), NT = bwd_nt(i);
        fill_layer(L, bwd_chunk0<NP>(i), KS, NT, TILE * NT);
        if (i == B_COLOR) {  // A[i = hv feature][k = rgb channel] = Wc[k][i]
            L.rows_valid = PN_WIDTH_COND;
            add_seg(L, P.wc, PN_WIDTH_COND, 0, 3, 1, 0);
        } else if (i == B_VIEW) {  // A[i = bottleneck feature][k = hv feature] = Wv[k][i]
            add_seg(L, P.wv, ldv, 0, PN_WIDTH_COND, 1, 0);
        } else if (i == B_EXTRA) {  // k < 256: We[k][i] ; 256 <= k < 256 + nc: Wd[k - 256][i]
            add_seg(L, P.we, PN_WIDTH, 0, PN_WIDTH, 1, 0);
            add_seg(L, P.wd, PN_WIDTH, PN_WIDTH, nc, 1, 0);
        } else if (i >= B_L7 && i <= B_L1) {  // A[i = input feature][k = output feature] = W_l[k][i]
            const int l = 7 - (i - B_L7);
            add_seg(L, P.w[l], l == 5 ? ld5 : PN_WIDTH, 0, PN_WIDTH, 1, 0);
        } else if (i == B_DENC0) {  // A[i = enc feature][k] = W0[k][i]
            L.rows_valid = PN_ENC_DIM;
            add_seg(L, P.w[0], PN_ENC_DIM, 0, PN_WIDTH, 1, 0);
        } else {  // B_DENC1: W5[k][256 + i]
            L.rows_valid = PN_ENC_DIM;
            add_seg(L, P.w[5], ld5, 0, PN_WIDTH, 1, PN_WIDTH);
        }
    }
    T.nchunks = bwd_chunk0<NP>(B_COUNT);
    return T;
}

// the packed blob: [forward chain | backward chain | int wexp[F_COUNT + B_COUNT] (NP = 2; 256 bytes)]
template <int NP>
static constexpr int64_t chain_bytes() { return (int64_t)(fwd_chunk0<NP>(F_COUNT) + bwd_chunk0<NP>(B_COUNT)) * Cfg<NP>::SLOT; }
constexpr int64_t WEXP_BYTES = 4 * (32 + 32 * WEXP_SLICES);
static_assert(F_COUNT + B_COUNT <= 32, "wexp table: 32 exponents + 32 x WEXP_SLICES maxima");
static_assert(F_COUNT <= 16 && B_COUNT <= 16, "the chain kernels keep a direction's exponents in lanes 0..15");
// Both directions as ONE table: the backward chain's chunks follow the forward chain's in the blob, its GEMMs follow in the
// exponent table - one maxima launch (NP = 2) and one pack launch per training step (five launches before: a clear and two
// each, 50 us of the 512-ray step).
template <int NP>
static int pack_both(int nc, const float* params, unsigned char* out, hipStream_t s) {
    int* wexp = reinterpret_cast<int*>(out + chain_bytes<NP>());
    PackTable T = fwd_table<NP>(nc);
    const PackTable B = bwd_table<NP>(nc);
    static_assert(F_COUNT + B_COUNT <= PACK_MAXL, "one table for both directions");
    for (int i = 0; i < B.n; ++i) {
        T.L[T.n + i] = B.L[i];
        T.L[T.n + i].chunk0 += T.nchunks;
    }
    T.n += B.n;
    T.nchunks += B.nchunks;
    if (NP == 2) {
        hipLaunchKernelGGL(k_chain_wexp, dim3(T.n, WEXP_SLICES), dim3(256), 0, s, T, params, reinterpret_cast<uint32_t*>(wexp) + 32);
        PN_CHECK_LAUNCH();
    }
    const int64_t threads = (int64_t)T.nchunks * (Cfg<NP>::CF + 1) * 64;
    hipLaunchKernelGGL(k_chain_pack<NP>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, T, params, out, wexp);
    PN_CHECK_LAUNCH();
    return PN_OK;
}

// Per-device launch state: the chains need Cfg<NP>::LDS_BYTES of dynamic LDS (the ring: 5 slots x 25 KB = 125 KB with fp16 pairs
// or the bf16 split in the default 8-wave form, 4 x 33 KB with plain bf16), an attribute that is set per (kernel, device), and the
// grid is sized by the device's CU count - a process may drive several devices (one host thread per device).
static_assert(Cfg<1>::LDS_BYTES <= 160 * 1024 && Cfg<2>::LDS_BYTES <= 160 * 1024 && Cfg<3>::LDS_BYTES <= 160 * 1024,
              "the weight ring must fit a CU's 160 KB of LDS");
static_assert(PN_ROW_PAD % CH_SAMPLES == 0, "sample buffers are padded to whole workgroup tiles");
constexpr int PN_MAX_DEVICES = 64;
static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PN_MAX_DEVICES) dev = 0;
    return dev;
}
struct AttrDone {
    bool done[PN_MAX_DEVICES];
};
template <typename K, typename A>
static int launch_chain(K kernel, int lds_bytes, AttrDone& attr, int64_t nst, int max_wgs, const A& a, hipStream_t s, int cls, double flops);

static int g_chain_cus[PN_MAX_DEVICES] = {};
static int chain_cus() {
    const int dev = current_device();
    if (!g_chain_cus[dev]) {
        hipDeviceProp_t pr;
        g_chain_cus[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256;
    }
    return g_chain_cus[dev];
}
// max_wgs > 0: the launch may occupy at most that many workgroups (= CUs: a chain workgroup fills one), so that a kernel of the
// other family - the weight-gradient GEMMs, on another stream - finds the remaining CUs free (see pn_chain_wgrad)
static int chain_grid(int64_t nst, int max_wgs) {
    int64_t wgs = (int64_t)chain_cus() * CH_WG_PER_CU;
    if (max_wgs > 0 && max_wgs < wgs) wgs = max_wgs;
#ifdef PN_TRACE_CHAIN  // diagnostic: PN_TRACE_ONE_WG=1 leaves every SIMD with ONE wave (is a GEMM phase slowed by its neighbour?)
    if (getenv("PN_TRACE_ONE_WG")) wgs = chain_cus();
#endif
    return (int)(nst < wgs ? nst : wgs);
}

template <typename K, typename A>
static int launch_chain(K kernel, int lds_bytes, AttrDone& attr, int64_t nst, int max_wgs, const A& a, hipStream_t s, int cls, double flops) {
    PnProfScope prof(cls, flops, s);
    const int dev = current_device();
    if (!attr.done[dev]) {  // (two threads racing here both set the same value)
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return PN_ERR_HIP;
        attr.done[dev] = true;
    }
    hipLaunchKernelGGL(kernel, dim3(chain_grid(nst, max_wgs)), dim3(CH_THREADS), lds_bytes, s, a);
    PN_CHECK_LAUNCH();
    return PN_OK;
}
#define LAUNCH_CHAIN(KERNEL, planes, nst, wgs, args, s, cls, flops)                                                   \
    do {                                                                                                             \
        static AttrDone done3 = {}, done2 = {}, done1 = {};                                                          \
        if ((planes) == 3) return launch_chain(KERNEL<3>, Cfg<3>::LDS_BYTES, done3, nst, wgs, args, s, cls, flops);  \
        if ((planes) == 2) return launch_chain(KERNEL<2>, Cfg<2>::LDS_BYTES, done2, nst, wgs, args, s, cls, flops);  \
        if ((planes) == 1) return launch_chain(KERNEL<1>, Cfg<1>::LDS_BYTES, done1, nst, wgs, args, s, cls, flops);  \
        return PN_ERR_UNSUPPORTED;                                                                                   \
    } while (0)
// where the pieces of a packed blob are, per arithmetic mode
struct ChainGeom {
    int64_t slot, bytes;                      // bytes per chunk; bytes of both chains (the wexp table follows)
    int f_all, f_den, b_all, b_l7, b_denc0;   // chunk counts / first chunks
};
template <int NP>
static ChainGeom geom_of() {
    return ChainGeom{Cfg<NP>::SLOT, chain_bytes<NP>(), fwd_chunk0<NP>(F_COUNT), fwd_chunk0<NP>(F_DEN),
                     bwd_chunk0<NP>(B_COUNT), bwd_chunk0<NP>(B_L7), bwd_chunk0<NP>(B_DENC0)};
}
static bool chain_geom(int planes, ChainGeom& g) {
    if (planes == 3) g = geom_of<3>();
    else if (planes == 2) g = geom_of<2>();
    else if (planes == 1) g = geom_of<1>();
    else return false;
    return true;
}
// algorithmic MACs x 2 per sample row: forward / data-gradient chain, and the trunk-only sweeps
static double flops_mlp(int nc) { return 2.0 * ((nc == 5 ? 611328.0 : 610304.0)); }
static const double kFlopsSweep = 1016320.0;


// ---- weight gradients of one training step -------------------------------------------------------------------
struct WgJob {
    WSeg seg[4];
    int nseg;
    int cfg;            // 0: 256x256, 1: 256x96, 2: 128x288, 3: 32x256, 4: 32x128
    int rows, cols;     // valid part of the result
    float* dst; int ldd;
    float* dbias;       // or null
    int fmt;            // cfg 0, fp16 pairs: 0 both operands fp32, 1 Y in Q24, 3 X and Y in Q24 (see pack_q24)
};
static const int kCfgM[5] = {256, 256, 128, 32, 32};
static const int kCfgN[5] = {256, 96, 288, 256, 128};

// Slab reduction of the jobs of one launch (grid.y = job): dst[r][c] += sum over the job's nb slabs of their [rows x cols] part
// (leading dimension src_ld), and dbias[i] += sum of the slabs' row sums (at slab offset bias_off), in a fixed order (64 elements per
// workgroup, four partial sums each, eight loads in flight): 31 reduction launches per training step were 16 % of its launches at
// the 512-ray share.
struct RedJob {
    const float* slabs;
    int64_t nb, stride, bias_off;
    int rows, cols, src_ld, ldd;
    float* dst;
    float* dbias;
};
constexpr int RED_MAXJ = WG_MAXJ;
struct RedMulti {
    RedJob job[RED_MAXJ];
};
__global__ __launch_bounds__(256) void k_reduce_job(RedMulti multi) {
    const RedJob& J = multi.job[blockIdx.y];
    const float* slabs = J.slabs;
    const int64_t nb = J.nb, stride = J.stride;
    const int rows = J.rows, cols = J.cols;
    float* dbias = J.dbias;
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx, nw = rows * cols, n = nw + (dbias ? rows : 0);
    if (blockIdx.x * 64 >= n) return;  // (uniform: a smaller job of the launch)
    float acc = 0.f;
    float* d = nullptr;
    if (e < n) {
        const float* p;
        if (e < nw) {
            const int r = e / cols, c = e - r * cols;
            p = slabs + (int64_t)r * J.src_ld + c;
            d = J.dst + (int64_t)r * J.ldd + c;
        } else {
            p = slabs + J.bias_off + (e - nw);
            d = dbias + (e - nw);
        }
        int64_t b = ty;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
        for (; b + 28 < nb; b += 32) {
            a0 += p[b * stride];
            a1 += p[(b + 4) * stride];
            a2 += p[(b + 8) * stride];
            a3 += p[(b + 12) * stride];
            a4 += p[(b + 16) * stride];
            a5 += p[(b + 20) * stride];
            a6 += p[(b + 24) * stride];
            a7 += p[(b + 28) * stride];
        }
        for (; b < nb; b += 4) a0 += p[b * stride];
        acc = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && e < n) *d += (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// jobs [j0, j0 + nj) of `jobs`: same tile configuration and operand format, disjoint destinations -> one GEMM launch + one reduction
static int run_reduce(const std::vector<RedJob>& red, size_t i0, size_t i1, hipStream_t s);
// (One reduction launch per group, the workspace reused by the next group.  Keeping every group's slabs until ONE reduction launch
// at the end of the step - thirteen launches fewer, 0.9 GB of workspace - measured no gain, eager or replayed:
// profiles/r04_launch_merge_ab.txt.)
template <int NP>
static int run_wgrad_group(const WgJob* jobs, int nj, float* work, int64_t work_floats, int max_wgs, hipStream_t s) {
    int64_t used = 0;
    std::vector<RedJob> red;
    if (nj < 1 || nj > WG_MAXJ) return PN_ERR_BAD_SHAPE;
    const WgJob& j0 = jobs[0];
    const int TMW = kCfgM[j0.cfg], TNW = kCfgN[j0.cfg];
    const int64_t stride = (int64_t)TMW * TNW + TMW;
    // workgroups per CU the configuration's registers and LDS allow (the narrow tiles: 146 / 100 registers per lane,
    // 37 / 20 KB): more sample ranges in flight, their staging and product phases interleave
    static const int kPerCu[5] = {1, 1, 1, 3, 4};
    int cus = chain_cus();
    if (max_wgs > 0 && max_wgs < cus) cus = max_wgs;  // the CUs this launch may occupy (the rest run a chain kernel of another stream)
    const int64_t slots = (int64_t)cus * kPerCu[j0.cfg] / nj;  // workgroups per job
    WgMulti m{};
    int64_t nsplit_max = 0, slab0 = 0;
    float* const base = work + used;
    double flops = 0;
    for (int q = 0; q < nj; ++q) {
        const WgJob& j = jobs[q];
        if (j.cfg != j0.cfg || j.fmt != j0.fmt) return PN_ERR_BAD_SHAPE;
        WgArgs& a = m.job[q];
        int64_t total = 0;
        for (int i = 0; i < j.nseg; ++i) {
            a.seg[i] = j.seg[i];
            total += j.seg[i].nhalf;
        }
        a.nseg = j.nseg;
        a.half_total = total;
        int64_t nsplit = slots < 1 ? 1 : slots;
        if (nsplit > (total + 3) / 4) nsplit = (total + 3) / 4;  // at least four half blocks per workgroup
        if (nsplit < 1) nsplit = 1;
        a.per = (total + nsplit - 1) / nsplit;
        nsplit = (total + a.per - 1) / a.per;
        a.slab = base + slab0 * stride;
        a.slab_stride = stride;
        a.bias = j.dbias != nullptr;
        red.push_back(RedJob{a.slab, nsplit, stride, (int64_t)TMW * TNW, j.rows, j.cols, TNW, j.ldd, j.dst, j.dbias});
        slab0 += nsplit;
        nsplit_max = nsplit > nsplit_max ? nsplit : nsplit_max;
        double rows = 0;
        for (int i = 0; i < j.nseg; ++i) rows += 16.0 * (double)j.seg[i].nhalf;
        flops += 2.0 * rows * j.rows * j.cols;
    }
    if (used + slab0 * stride > work_floats) return PN_ERR_BAD_SHAPE;
    used += slab0 * stride;
    // (a job with fewer workgroups than grid.x: its surplus workgroups find h0 >= half_total and write a zero slab that the
    // reduction does not read - nsplit is the job's own)
    const dim3 grid((unsigned)nsplit_max, (unsigned)nj);
    const WgJob& j = j0;
    const WgMulti& a = m;
    {
    // (class per kernel instantiation: 6 + cfg; the Q24 forms of the 256 x 256 tile are 11 (Y in Q24) and 12 (X and Y in Q24))
    PnProfScope prof((j.cfg == 0 && j.fmt) ? (j.fmt == 3 ? 12 : 11) : 6 + j.cfg, flops, s);  // the GEMM kernel alone
    switch (j.cfg) {
        case 0:
            if constexpr (kQ24<NP>) {
                if (j.fmt == 3) { hipLaunchKernelGGL((k_chain_wgrad<NP, 2, 4, 4, 2, true, true>), grid, dim3(512), 0, s, a); break; }
                if (j.fmt == 1) { hipLaunchKernelGGL((k_chain_wgrad<NP, 2, 4, 4, 2, false, true>), grid, dim3(512), 0, s, a); break; }
            }
            if (j.fmt != 0) return PN_ERR_UNSUPPORTED;
            hipLaunchKernelGGL((k_chain_wgrad<NP, 2, 4, 4, 2>), grid, dim3(512), 0, s, a);
            break;
        case 1: hipLaunchKernelGGL((k_chain_wgrad<NP, 1, 3, 8, 1>), grid, dim3(512), 0, s, a); break;
        // (128 x 288 by twelve waves of 1 x 3 tiles: as four waves of 1 x 9 it held 392 registers per lane - one wave per
        // SIMD, two register sets in flight - and ran 3.8 TB/s)
        case 2: hipLaunchKernelGGL((k_chain_wgrad<NP, 1, 3, 4, 3>), grid, dim3(768), 0, s, a); break;
        case 3: hipLaunchKernelGGL((k_chain_wgrad<NP, 1, 2, 1, 4>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((k_chain_wgrad<NP, 1, 1, 1, 4>), grid, dim3(256), 0, s, a); break;
    }
    }
    PN_CHECK_LAUNCH();
    return run_reduce(red, 0, red.size(), s);
}
// the slab reductions of jobs [i0, i1) in one launch
static int run_reduce(const std::vector<RedJob>& red, size_t i0, size_t i1, hipStream_t s) {
    while (i0 < i1) {
        const size_t n = i1 - i0 < (size_t)RED_MAXJ ? i1 - i0 : (size_t)RED_MAXJ;
        RedMulti r{};
        int nmax = 0;
        for (size_t q = 0; q < n; ++q) {
            r.job[q] = red[i0 + q];
            const int e = r.job[q].rows * r.job[q].cols + (r.job[q].dbias ? r.job[q].rows : 0);
            nmax = e > nmax ? e : nmax;
        }
        hipLaunchKernelGGL(k_reduce_job, dim3((unsigned)((nmax + 63) / 64), (unsigned)n), dim3(256), 0, s, r);
        PN_CHECK_LAUNCH();
        i0 += n;
    }
    return PN_OK;
}

// T-tensor format of a call: 0 = every tensor fp32 (bf16 with planes = 1), 1 = Q24 where pn_chain_q24_slots says so (fp16 pairs with
// 16-sample tiles only)
static bool tfmt_ok(int planes, int t_format) { return t_format == 0 || (t_format == 1 && planes == 2 && kQ24<2>); }

extern "C" {

/* samples per T-layout block (= samples per wave of the chain kernels): 16 or 32 */
int pn_chain_tile(void) { return TILE; }

int pn_chain_q24_slots(int planes, int t_format, int tensor) {
    if (planes != 2 || t_format != 1 || !kQ24<2> || tensor < 0 || tensor > 3) return 0;
    int m = 0;
    for (int sl = 0; sl < 8; ++sl)
        if (tensor < 2 ? q24_act(sl) : q24_delta(sl)) m |= 1 << sl;
    return m;
}

// bytes of the packed chains for `planes` (3: exact split, 1: plain bf16): [forward chain | backward chain]
int64_t pn_chain_pack_bytes(int planes) {
    ChainGeom g;
    if (!chain_geom(planes, g)) return PN_ERR_UNSUPPORTED;
    return g.bytes + WEXP_BYTES;
}

int pn_chain_pack(const float* params, int nc, int planes, void* pack, void* stream) {
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    if (!params || !pack) return PN_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    unsigned char* out = (unsigned char*)pack;
    if (planes == 3) return pack_both<3>(nc, params, out, s);
    if (planes == 2) return pack_both<2>(nc, params, out, s);
    if (planes == 1) return pack_both<1>(nc, params, out, s);
    return PN_ERR_UNSUPPORTED;
}

int64_t pn_chain_acts_floats(int64_t M) { return acts_floats(pn_pad(M)); }

int pn_chain_amax_slots(void) { return AM_COUNT; }

int pn_chain_forward(int64_t M, int rows_per_ray, int64_t view_rows, int nc, int planes, const void* pack,
                     const float* mean, const float* cov, const float* viewdirs, float* view_tab, float* enc_t, float* acts_t,
                     uint32_t* masks, float* raw_rgb, float* raw_den, uint32_t* amax, int t_format, int max_wgs, void* stream) {
    if (M <= 0 || rows_per_ray <= 0 || view_rows <= 0 || max_wgs < 0) return PN_ERR_BAD_SHAPE;
    if (!tfmt_ok(planes, t_format)) return PN_ERR_UNSUPPORTED;
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    if (!pack || !mean || !cov || !viewdirs || !view_tab || !enc_t || !masks || !raw_rgb || !raw_den) return PN_ERR_NULL;  // acts_t may be null
    FwdArgs a{};
    a.M = M;
    a.nst = pn_pad(M) / CH_SAMPLES;
    a.rows_per_ray = rows_per_ray;
    a.nc = nc;
    a.view_rows = view_rows;
    ChainGeom g;
    if (!chain_geom(planes, g)) return PN_ERR_UNSUPPORTED;
    a.pack = (const unsigned char*)pack;
    a.wexp = reinterpret_cast<const int*>(a.pack + g.bytes);
    a.mean = mean; a.cov = cov; a.viewdirs = viewdirs; a.view_tab = view_tab;
    a.enc_t = enc_t; a.acts_t = acts_t; a.masks = masks; a.raw_rgb = raw_rgb; a.raw_den = raw_den;
    a.amax = planes == 2 ? amax : nullptr;
    a.q24 = t_format;
    static_assert(AM_COUNT <= 256, "k_view_table's first workgroup clears the table of maxima");
    hipLaunchKernelGGL(k_view_table, dim3((unsigned)((view_rows * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, view_rows,
                       viewdirs, view_tab, a.amax, AM_COUNT);
    PN_CHECK_LAUNCH();
    LAUNCH_CHAIN(k_chain_fwd, planes, a.nst, max_wgs, a, (hipStream_t)stream, 2, (double)M * flops_mlp(nc));
}

/* d sigma / d mean by one reverse sweep (see k_chain_dgrad).  rs_t: T32 [8][Mp*256] (r_0..r_7, kept for the second-order
 * weight gradients); grad_mean [M,3] = + d sigma / d mean. */
int pn_chain_density_grad(int64_t M, int nc, int planes, float density_bias, const float* params, const void* pack,
                          const float* mean, const float* cov, const uint32_t* masks, const float* raw_den, float* rs_t,
                          int keep_all, float* grad_mean, uint32_t* amax, int t_format, int max_wgs, void* stream) {
    if (M <= 0 || max_wgs < 0) return PN_ERR_BAD_SHAPE;
    if (!tfmt_ok(planes, t_format)) return PN_ERR_UNSUPPORTED;
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    ChainGeom g;
    if (!chain_geom(planes, g)) return PN_ERR_UNSUPPORTED;
    if (!params || !pack || !mean || !cov || !masks || !raw_den || !rs_t || !grad_mean) return PN_ERR_NULL;
    SweepArgs a{};
    a.M = M;
    a.nst = pn_pad(M) / CH_SAMPLES;
    a.nc = nc;
    a.density_bias = density_bias;
    a.pack = (const unsigned char*)pack + (int64_t)(g.f_all + g.b_l7) * g.slot;
    a.nchunk = g.b_all - g.b_l7;
    a.wexp = reinterpret_cast<const int*>((const unsigned char*)pack + g.bytes) + F_COUNT;
    a.masks = masks; a.raw_den = raw_den; a.mean = mean; a.cov = cov;
    a.wd0 = params + pn_layout(nc).wd;
    a.vec_t = rs_t; a.out3 = grad_mean; a.keep_all = keep_all != 0;
    a.amax = planes == 2 ? amax : nullptr;
    a.q24 = t_format;
    LAUNCH_CHAIN(k_chain_dgrad, planes, a.nst, max_wgs, a, (hipStream_t)stream, 3, (double)M * kFlopsSweep);
}

/* forward-mode tangent sweep along v (see k_chain_tangent): edot_t T32 [Mp*96], tang_t T32 [8][Mp*256], sdot [M]. */
int pn_chain_tangent(int64_t M, int nc, int planes, const float* params, const void* pack, const float* mean,
                     const float* cov, const uint32_t* masks, const float* v, float* edot_t, float* tang_t, float* sdot,
                     uint32_t* amax, int t_format, int max_wgs, void* stream) {
    if (M <= 0 || max_wgs < 0) return PN_ERR_BAD_SHAPE;
    if (!tfmt_ok(planes, t_format)) return PN_ERR_UNSUPPORTED;
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    ChainGeom g;
    if (!chain_geom(planes, g)) return PN_ERR_UNSUPPORTED;
    if (!params || !pack || !mean || !cov || !masks || !v || !edot_t || !tang_t || !sdot) return PN_ERR_NULL;
    SweepArgs a{};
    a.M = M;
    a.nst = pn_pad(M) / CH_SAMPLES;
    a.nc = nc;
    a.pack = (const unsigned char*)pack;
    a.nchunk = g.f_den;
    a.wexp = reinterpret_cast<const int*>((const unsigned char*)pack + g.bytes);
    a.masks = masks; a.mean = mean; a.cov = cov; a.v = v;
    a.wd0 = params + pn_layout(nc).wd;
    a.vec_t = tang_t; a.edot_t = edot_t; a.sdot = sdot;
    a.amax = planes == 2 ? amax : nullptr;
    a.q24 = t_format;
    LAUNCH_CHAIN(k_chain_tangent, planes, a.nst, max_wgs, a, (hipStream_t)stream, 4, (double)M * kFlopsSweep);
}

/* backward chain (see k_chain_bwd).  sdot / coef_t: second-order path (both or neither); d_mean nullable. */
int pn_chain_backward(int64_t M, int nc, int planes, float density_bias, const void* pack, const uint32_t* masks,
                      const float* raw_den, const float* d_raw_rgb, const float* d_raw_den, const float* sdot,
                      const float* mean, const float* cov, float* drgb_t, float* dhv_t, float* d8_t, float* delta_t,
                      float* coef_t, float* d_mean, uint32_t* amax, int t_format, int max_wgs, void* stream) {
    if (M <= 0 || max_wgs < 0) return PN_ERR_BAD_SHAPE;
    if (!tfmt_ok(planes, t_format)) return PN_ERR_UNSUPPORTED;
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    ChainGeom g;
    if (!chain_geom(planes, g)) return PN_ERR_UNSUPPORTED;
    if (!pack || !masks || !raw_den || !d_raw_rgb || !d_raw_den || !drgb_t || !dhv_t || !d8_t || !delta_t) return PN_ERR_NULL;
    if (d_mean && (!mean || !cov)) return PN_ERR_NULL;
    if ((sdot == nullptr) != (coef_t == nullptr)) return PN_ERR_NULL;
    BwdArgs a{};
    a.M = M;
    a.nst = pn_pad(M) / CH_SAMPLES;
    a.nc = nc;
    a.density_bias = density_bias;
    a.pack = (const unsigned char*)pack + (int64_t)g.f_all * g.slot;
    a.nchunk = d_mean ? g.b_all : g.b_denc0;
    a.wexp = reinterpret_cast<const int*>((const unsigned char*)pack + g.bytes) + F_COUNT;
    a.masks = masks; a.raw_den = raw_den; a.d_rgb = d_raw_rgb; a.d_den = d_raw_den; a.sdot = sdot;
    a.mean = mean; a.cov = cov;
    a.drgb_t = drgb_t; a.dhv_t = dhv_t; a.d8_t = d8_t; a.delta_t = delta_t; a.coef_t = coef_t; a.d_mean = d_mean;
    a.amax = planes == 2 ? amax : nullptr;
    a.q24 = t_format;
    LAUNCH_CHAIN(k_chain_bwd, planes, a.nst, max_wgs, a, (hipStream_t)stream, 5, (double)M * (flops_mlp(nc) - (d_mean ? 0.0 : 2.0 * 2 * 96 * 256)));
}


int64_t pn_chain_wgrad_work_floats(void) {
    // the partial sums of one launch (a group of jobs of one tile configuration): (workgroups) x (tile + row sums) floats, the
    // largest being one 256 x 256 slab per CU; sized for up to 320 CUs
    const int64_t stride = 256 * 288 + 256;
    return (256 + 64) * stride + 1024;
}

/* Weight and bias gradients of ONE training step from the T32 tensors the chain kernels left behind: one split-bf16
 * (or plain bf16) TN GEMM per layer over the sample blocks of all `n` evaluations (and, for an evaluation with
 * rs_t / tang_t, its second-order rows r_l^T hdot_{l-1}); accumulates (+=) into the flat gradient block. */
int pn_chain_wgrad(int n, const PnChainEval* ev, int nc, int planes, float* grads, float* work, int64_t work_floats,
                   int which, int t_format, int max_wgs, void* stream) {
    if (n < 1 || n > 3 || which < 1 || which > 3 || max_wgs < 0) return PN_ERR_BAD_SHAPE;
    if (!tfmt_ok(planes, t_format)) return PN_ERR_UNSUPPORTED;
    const bool first = which & 1, second = which & 2;
    if (nc != 1 && nc != 5) return PN_ERR_UNSUPPORTED;
    if (planes != 1 && planes != 2 && planes != 3) return PN_ERR_UNSUPPORTED;
    if (!ev || !grads || !work) return PN_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    const PnLayout L = pn_layout(nc);
    int n2 = 0;
    for (int e = 0; e < n; ++e) {
        if (ev[e].M <= 0 || !ev[e].enc_t || !ev[e].acts_t) return PN_ERR_NULL;
        if (first && (!ev[e].drgb_t || !ev[e].dhv_t || !ev[e].d8_t || !ev[e].delta_t)) return PN_ERR_NULL;  // (written by the backward chain)
        if (ev[e].rs_t) {
            if (!ev[e].edot_t || !ev[e].tang_t || (first && !ev[e].coef_t)) return PN_ERR_NULL;
            if (second) ++n2;
        }
        if (planes == 2 && !ev[e].amax) return PN_ERR_NULL;
    }
    if ((first ? n : 0) + n2 > 4) return PN_ERR_UNSUPPORTED;
    if (!first && !n2) return PN_OK;  // second-order rows only, and no evaluation has any
    // planes = 2: a weight gradient sums over ALL samples, so each operand tensor gets ONE power of two, from the maxima
    // the chain kernels left in the evaluation's table
    // the step's jobs are collected first and launched in groups of the same tile configuration and operand format (see WgMulti);
    // `after`: a job that adds into the destination of an earlier one (the softplus' row of the density head) goes in a later launch
    std::vector<WgJob> jobs;
    std::vector<int> after;
    auto run = [&](const WgJob& j, int later = 0) {
        if (j.nseg) {
            jobs.push_back(j);
            after.push_back(later);
        }
        return (int)PN_OK;
    };
    auto am = [&](int e, int slot) -> const uint32_t* { return ev[e].amax ? ev[e].amax + slot : nullptr; };
    auto mp = [&](int e) { return pn_pad(ev[e].M); };
    // the T tensors hold 2-byte elements with planes = 1 (see TEl): offsets are in ELEMENTS of the mode's type
    const int64_t esz = planes == 1 ? 2 : 4;
    auto at = [&](const float* p, int64_t elems) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + elems * esz); };
    auto act = [&](int e, int slot) { return at(ev[e].acts_t, act_off(slot, mp(e))); };
    int rc;
    // trunk layers (layer 5: hidden columns here, skip columns below)
    for (int l = 0; l < 8; ++l) {
        WgJob j{};
        for (int e = 0; e < n; ++e) {
            const int64_t Mp = mp(e);
            if (first)
                j.seg[j.nseg++] = WSeg{at(ev[e].delta_t, (int64_t)l * Mp * 256), l == 0 ? ev[e].enc_t : act(e, l - 1), Mp / 16, 256,
                                       l == 0 ? 96 : 256, 1, am(e, AM_DELTA0 + l), am(e, l == 0 ? AM_ENC : AM_ACT0 + l - 1)};
            if (second && ev[e].rs_t)
                j.seg[j.nseg++] = WSeg{at(ev[e].rs_t, (int64_t)l * Mp * 256),
                                       l == 0 ? ev[e].edot_t : at(ev[e].tang_t, (int64_t)(l - 1) * Mp * 256), Mp / 16, 256,
                                       l == 0 ? 96 : 256, 0, am(e, AM_RS0 + l), am(e, l == 0 ? AM_EDOT : AM_TANG0 + l - 1)};
        }
        j.cfg = l == 0 ? 1 : 0;
        // fp16 pairs: delta_l / r_l (l = 1-4, 6, 7) and h_{l-1} / hdot_{l-1} (l >= 1) come in Q24
        j.fmt = (t_format == 1 && l >= 1) ? (q24_delta(l) ? 3 : 1) : 0;
        j.rows = 256;
        j.cols = l == 0 ? 96 : 256;
        j.dst = grads + L.w[l];
        j.ldd = l == 0 ? 96 : (l == 5 ? 352 : 256);
        j.dbias = first ? grads + L.b[l] : nullptr;
        if ((rc = run(j)) != PN_OK) return rc;
        if (l == 5) {
            WgJob k{};
            for (int e = 0; e < n; ++e) {
                const int64_t Mp = mp(e);
                if (first)
                    k.seg[k.nseg++] = WSeg{at(ev[e].delta_t, (int64_t)5 * Mp * 256), ev[e].enc_t, Mp / 16, 256, 96, 0,
                                           am(e, AM_DELTA0 + 5), am(e, AM_ENC)};
                if (second && ev[e].rs_t)
                    k.seg[k.nseg++] = WSeg{at(ev[e].rs_t, (int64_t)5 * Mp * 256), ev[e].edot_t, Mp / 16, 256, 96, 0,
                                           am(e, AM_RS0 + 5), am(e, AM_EDOT)};
            }
            k.cfg = 1; k.rows = 256; k.cols = 96;
            k.dst = grads + L.w[5] + 256; k.ldd = 352; k.dbias = nullptr;
            if ((rc = run(k)) != PN_OK) return rc;
        }
    }
    if (first) {  // extra layer: d bottleneck^T h7
        WgJob j{};
        for (int e = 0; e < n; ++e) j.seg[j.nseg++] = WSeg{ev[e].d8_t, act(e, 7), mp(e) / 16, 288, 256, 1, am(e, AM_D8B), am(e, AM_ACT0 + 7)};
        j.cfg = 0; j.rows = 256; j.cols = 256; j.dst = grads + L.we; j.ldd = 256; j.dbias = grads + L.be;
        if ((rc = run(j)) != PN_OK) return rc;
    }
    {  // density head: d raw_density^T h7 (+ softplus' rows against hdot_7 into row 0)
        WgJob j{};
        if (first)
            for (int e = 0; e < n; ++e) j.seg[j.nseg++] = WSeg{at(ev[e].d8_t, 256 * TILE), act(e, 7), mp(e) / 16, 288, 256, 1, am(e, AM_D8D), am(e, AM_ACT0 + 7)};
        j.cfg = 3; j.rows = nc; j.cols = 256; j.dst = grads + L.wd; j.ldd = 256; j.dbias = grads + L.bd;
        if ((rc = run(j)) != PN_OK) return rc;
        WgJob k{};  // (with the first-order products: coef_t is written by the evaluation's backward chain)
        for (int e = 0; e < n; ++e)
            if (first && ev[e].rs_t)
                k.seg[k.nseg++] = WSeg{ev[e].coef_t, at(ev[e].tang_t, (int64_t)7 * mp(e) * 256), mp(e) / 16, 32, 256, 0,
                                       am(e, AM_COEF), am(e, AM_TANG0 + 7)};
        if (k.nseg) {
            k.cfg = 3; k.rows = 1; k.cols = 256; k.dst = grads + L.wd; k.ldd = 256; k.dbias = nullptr;
            if ((rc = run(k, 1)) != PN_OK) return rc;
        }
    }
    if (first) {  // view layer: d hv^T [bottleneck | view encoding]
        WgJob j{};
        for (int e = 0; e < n; ++e) j.seg[j.nseg++] = WSeg{ev[e].dhv_t, act(e, 8), mp(e) / 16, 128, 288, 1, am(e, AM_DHV), am(e, AM_ACT0 + 8)};
        j.cfg = 2; j.rows = 128; j.cols = PN_WIDTH + PN_VIEW_DIM; j.dst = grads + L.wv; j.ldd = PN_WIDTH + PN_VIEW_DIM;
        j.dbias = grads + L.bv;
        if ((rc = run(j)) != PN_OK) return rc;
    }
    if (first) {  // colour head: d rgb^T hv
        WgJob j{};
        for (int e = 0; e < n; ++e) j.seg[j.nseg++] = WSeg{ev[e].drgb_t, act(e, 9), mp(e) / 16, 32, 128, 1, am(e, AM_DRGB), am(e, AM_ACT0 + 9)};
        j.cfg = 4; j.rows = 3; j.cols = 128; j.dst = grads + L.wc; j.ldd = 128; j.dbias = grads + L.bc;
        if ((rc = run(j)) != PN_OK) return rc;
    }
    // Jobs share a launch only as far as a workgroup's range stays within WG_RANGE half blocks (8192 samples: what one job alone
    // gives a workgroup at the 4096-ray batch).  Sharing pays where the jobs are short - the 512-ray share of an 8-GPU run: 3.36 ->
    // 3.22 - 3.27 ms per step, nothing at 4096 rays (profiles/r04_multijob_ab.txt) - and a range n times as long is an n times
    // longer fp32 accumulation chain per accumulator: on the cancelling-sum stress test (1.97 M rows, fp32 tensors, seven jobs in
    // one launch) the error grew from 1.9e-6 to 7.4e-6 of the tensor's largest element (an fp32 GEMM: 1.5e-5).
    constexpr int64_t WG_RANGE = 512;
    std::vector<char> done(jobs.size(), 0);
    for (int pass = 0; pass < 2; ++pass)
    for (size_t i = 0; i < jobs.size(); ++i) {
        if (done[i] || after[i] != pass) continue;
        WgJob grp[WG_MAXJ];
        int ng = 0;
        int64_t total_i = 0;
        for (int q = 0; q < jobs[i].nseg; ++q) total_i += jobs[i].seg[q].nhalf;
        int cus_i = chain_cus();
        if (max_wgs > 0 && max_wgs < cus_i) cus_i = max_wgs;
        int64_t gmax = WG_RANGE * cus_i / (total_i > 0 ? total_i : 1);
        gmax = gmax < 1 ? 1 : (gmax > WG_MAXJ ? WG_MAXJ : gmax);
        for (size_t k = i; k < jobs.size() && ng < gmax; ++k)
            if (!done[k] && jobs[k].cfg == jobs[i].cfg && jobs[k].fmt == jobs[i].fmt && after[k] == after[i]) {
                grp[ng++] = jobs[k];
                done[k] = 1;
            }
        rc = planes == 3 ? run_wgrad_group<3>(grp, ng, work, work_floats, max_wgs, s)
                         : (planes == 2 ? run_wgrad_group<2>(grp, ng, work, work_floats, max_wgs, s)
                                        : run_wgrad_group<1>(grp, ng, work, work_floats, max_wgs, s));
        if (rc != PN_OK) return rc;
    }
    return PN_OK;
}

}  // extern "C"
