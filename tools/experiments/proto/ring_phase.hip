// MICROBENCHMARK (not part of the library): the GEMM-phase loop of gemm_phase.hip WITH a weight ring, piece by piece - which part of the
// ring costs the fused chains their ~30 points of matrix-pipe utilisation?
//   MODE bit 0: an s_barrier per 24-KB chunk (8 waves)
//   MODE bit 1: the chunk stream refilled by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per wave instruction, 3 per wave and chunk), 5 slots,
//               counted vmcnt before a chunk is consumed (implies the barrier)
//   MODE bit 2: the DMA instructions of a round issued one per GEMM step behind its MFMAs (as the chains do) instead of as a burst behind the barrier
//   MODE bit 3: only waves 0-3 issue the DMA (6 instructions each)
// v_mfma_f32_16x16x32_f16, 8 waves (two per SIMD), 12 steps per chunk, 2 fragment reads + 3 MFMAs per step.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int SLOT = 24 * 1024, NSLOT = 5, D = NSLOT - 1;

template <int MODE>
__global__ __launch_bounds__(512) void k_ring_phase(const unsigned char* w, int nchunk, float* out, int rounds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool DMA = MODE & 2, BAR = (MODE & 1) || DMA, SPREAD = MODE & 4, HALF = MODE & 8;
    constexpr int PPW = HALF ? 6 : 3;  // pieces per issuing wave and chunk
    for (int i = tid; i < NSLOT * SLOT / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(lds)[i] = 0x3c003c00u ^ ((uint32_t)i * 2654435761u & 0x03ff03ffu);
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, nchunk * SLOT, 0x00020000);
    f16x8 bh, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bh[j] = (_Float16)(1.0f + 0.01f * (float)((lane + j) & 7));
        bl[j] = (_Float16)(0.001f * (float)((lane * 3 + j) & 7));
    }
    f32x4 acc[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    int pf = 0, pslot = 0, cslot = 0;
    auto piece = [&](int i, int soff, unsigned char* dst) {  // piece i of this wave's share of a round
        const int f = HALF ? wid + 4 * i : wid + 8 * i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + f * 1024), 16, lane * 16, soff + f * 1024, 0, 0);
    };
    const bool issuer = !HALF || wid < 4;
    if constexpr (DMA) {
        for (int d = 0; d < D; ++d) {
            if (issuer)
                for (int i = 0; i < PPW; ++i) piece(i, pf * SLOT, lds + pslot * SLOT);
            pf = pf + 1 == nchunk ? 0 : pf + 1;
            pslot = pslot + 1 == NSLOT ? 0 : pslot + 1;
        }
    }
    for (int r = 0; r < rounds; ++r) {
        int soff = 0;
        unsigned char* dst = nullptr;
        if constexpr (DMA) {
            if (issuer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (D - 1)) : "memory");
        }
        if constexpr (BAR) __builtin_amdgcn_s_barrier();
        if constexpr (DMA) {
            soff = pf * SLOT;
            dst = lds + pslot * SLOT;
            pf = pf + 1 == nchunk ? 0 : pf + 1;
            pslot = pslot + 1 == NSLOT ? 0 : pslot + 1;
            if constexpr (!SPREAD) {
                if (issuer)
#pragma unroll
                    for (int i = 0; i < PPW; ++i) piece(i, soff, dst);
            }
        }
        const unsigned char* base = lds + cslot * SLOT + lane * 16;
        cslot = cslot + 1 == NSLOT ? 0 : cslot + 1;
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            f16x8 ah, al;
            ah = *reinterpret_cast<const f16x8*>(base + (2 * s) * 1024);
            al = *reinterpret_cast<const f16x8*>(base + (2 * s + 1) * 1024);
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[s], 0, 0, 0);
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[s], 0, 0, 0);
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[s], 0, 0, 0);
            if constexpr (DMA && SPREAD) {
                constexpr int every = 12 / PPW;
                if (issuer && s % every == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    piece(s / every, soff, dst);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float rs = 0.f;
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) rs += acc[t][e];
    if (rs == 12345.678f) out[tid] = rs;
}

extern "C" int ring_phase_run(int mode, const unsigned char* w, int nchunk, float* out, int blocks, int rounds, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define CASE(M) case M: hipLaunchKernelGGL((k_ring_phase<M>), dim3(blocks), dim3(512), NSLOT * SLOT, s, w, nchunk, out, rounds); break;
    switch (mode) {
        CASE(0) CASE(1) CASE(2) CASE(6) CASE(10) CASE(14)
        default: return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
