// MICROBENCHMARK (not part of the library): the ceiling of a chain kernel's GEMM phase as the fused chains organise it - every wave
// reads its A fragments (weights) from LDS with ds_read_b128 and multiplies them with B operands it holds in registers, three MFMAs per
// fp32 product (fp16 pairs) - WITHOUT the refill DMA, the chunk barriers and the epilogues: how much of the ~60 % matrix-pipe
// utilisation of the real GEMM phases is the shape itself?
//   SHAPE 0: v_mfma_f32_16x16x32_f16, a step = 2 fragment reads (h, l planes, 1 KB each per wave) + 3 MFMAs of 16 cycles
//   SHAPE 1: v_mfma_f32_32x32x16_f16, a step = 2 fragment reads + 3 MFMAs of 32 cycles (half the LDS bytes per FLOP)
// The chunk (24 KB = 12 steps) sits in LDS and is walked `iters` times; accumulators rotate over NT tiles as in chain_gemm.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/experiments/proto/gemm_phase.hip -o tools/experiments/proto/libgemm_phase.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NT>
__global__ __launch_bounds__(512) void k_gemm_phase(const uint32_t* seed, float* out, int iters, int waves_active) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 24 * 1024 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(lds)[i] = 0x3c003c00u ^ (seed[i & 255] & 0x03ff03ffu);
    __syncthreads();
    if (wid >= waves_active) return;
    f16x8 bh, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bh[j] = (_Float16)(1.0f + 0.01f * (float)((lane + j) & 7));
        bl[j] = (_Float16)(0.001f * (float)((lane * 3 + j) & 7));
    }
    typedef typename std::conditional<SHAPE == 0, f32x4, f32x16>::type acc_t;
    acc_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < (SHAPE == 0 ? 4 : 16); ++e) acc[t][e] = 0.f;
    uint32_t off = lane * 16;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(off));  // (opaque: the fragment reads are not loop-invariant to the compiler)
        const unsigned char* base = lds + off;
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(base + (2 * s) * 1024);
            const f16x8 al = *reinterpret_cast<const f16x8*>(base + (2 * s + 1) * 1024);
            const int t = s % NT;
            if constexpr (SHAPE == 0) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t], 0, 0, 0);
            } else {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < (SHAPE == 0 ? 4 : 16); ++e) r += acc[t][e];
    if (r == 12345.678f) out[tid] = r;  // (keeps the sums alive)
}

// the same with the NEXT pass's 24 fragment reads issued before this pass's MFMAs (two register sets: the reads of a pass have a whole
// pass of matrix work to land) - what a hand-scheduled look-ahead can reach
template <int SHAPE>
__global__ __launch_bounds__(512) void k_gemm_phase_pipe(const uint32_t* seed, float* out, int iters, int waves_active) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 24 * 1024 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(lds)[i] = 0x3c003c00u ^ (seed[i & 255] & 0x03ff03ffu);
    __syncthreads();
    if (wid >= waves_active) return;
    f16x8 bh, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bh[j] = (_Float16)(1.0f + 0.01f * (float)((lane + j) & 7));
        bl[j] = (_Float16)(0.001f * (float)((lane * 3 + j) & 7));
    }
    typedef typename std::conditional<SHAPE == 0, f32x4, f32x16>::type acc_t;
    constexpr int NT = SHAPE == 0 ? 12 : 4;
    acc_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < (SHAPE == 0 ? 4 : 16); ++e) acc[t][e] = 0.f;
    uint32_t off = lane * 16;
    f16x8 fa[24], fb[24];
    auto rd = [&](f16x8 (&f)[24]) {
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int k = 0; k < 24; ++k) f[k] = *reinterpret_cast<const f16x8*>(lds + off + k * 1024);
    };
    auto mm = [&](const f16x8 (&f)[24]) {
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            const int t = s % NT;
            if constexpr (SHAPE == 0) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[2 * s + 1], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[2 * s], bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[2 * s], bh, acc[t], 0, 0, 0);
            } else {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * s + 1], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * s], bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * s], bh, acc[t], 0, 0, 0);
            }
        }
    };
    rd(fa);
    for (int it = 0; it < iters; it += 2) {
        rd(fb);
        mm(fa);
        rd(fa);
        mm(fb);
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < (SHAPE == 0 ? 4 : 16); ++e) r += acc[t][e];
    if (r == 12345.678f) out[tid] = r;
}

// the 16x16x32 loop with HAND-PLACED fragment reads LA steps ahead (inline asm, counted lgkmcnt - the chains' read_a / wait_a idiom):
// what look-ahead alone buys over hipcc's read-then-use order
template <int LA>
__global__ __launch_bounds__(512) void k_gemm_phase_la(const uint32_t* seed, float* out, int iters, int waves_active) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 24 * 1024 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(lds)[i] = 0x3c003c00u ^ (seed[i & 255] & 0x03ff03ffu);
    __syncthreads();
    if (wid >= waves_active) return;
    f16x8 bh, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bh[j] = (_Float16)(1.0f + 0.01f * (float)((lane + j) & 7));
        bl[j] = (_Float16)(0.001f * (float)((lane * 3 + j) & 7));
    }
    f32x4 acc[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + lane * 16;
    f16x8 qh[LA + 1], ql[LA + 1];
    auto rd = [&](int slot, int s) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qh[slot]) : "v"(addr), "n"(0) : "memory");
    };
    (void)rd;
#define RD(slot, s)                                                                                                      \
    do {                                                                                                                 \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qh[slot]) : "v"(addr), "n"((2 * (s)) * 1024) : "memory");      \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ql[slot]) : "v"(addr), "n"((2 * (s) + 1) * 1024) : "memory");  \
    } while (0)
    // prologue: steps 0 .. LA-1 of the first pass
#pragma unroll
    for (int s = 0; s < LA; ++s) RD(s, s);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            constexpr int dummy = 0;
            (void)dummy;
            // read step s + LA (wrapping into the next pass: same addresses) into slot (s + LA) % (LA + 1)
            switch ((s + LA) % 12) {
#define C(n) case n: RD((s + LA) % (LA + 1), n); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11)
#undef C
            }
            // all but the newest LA sets have landed
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(qh[s % (LA + 1)]), "+v"(ql[s % (LA + 1)]) : "n"(2 * LA) : "memory");
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ql[s % (LA + 1)], bh, acc[s], 0, 0, 0);
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[s % (LA + 1)], bl, acc[s], 0, 0, 0);
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[s % (LA + 1)], bh, acc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) r += acc[t][e];
    if (r == 12345.678f) out[tid] = r;
}

extern "C" int gemm_phase_run(int shape, int waves, const uint32_t* seed, float* out, int blocks, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int threads = 64 * (waves > 4 ? 8 : 4);
    if (shape == 0) hipLaunchKernelGGL((k_gemm_phase<0, 12>), dim3(blocks), dim3(threads), 24 * 1024, s, seed, out, iters, waves);
    else if (shape == 1) hipLaunchKernelGGL((k_gemm_phase<1, 4>), dim3(blocks), dim3(threads), 24 * 1024, s, seed, out, iters, waves);
    else if (shape == 2) hipLaunchKernelGGL((k_gemm_phase_pipe<0>), dim3(blocks), dim3(threads), 24 * 1024, s, seed, out, iters, waves);
    else if (shape == 3) hipLaunchKernelGGL((k_gemm_phase_pipe<1>), dim3(blocks), dim3(threads), 24 * 1024, s, seed, out, iters, waves);
    else if (shape == 4) hipLaunchKernelGGL((k_gemm_phase_la<1>), dim3(blocks), dim3(threads), 24 * 1024, s, seed, out, iters, waves);
    else if (shape == 5) hipLaunchKernelGGL((k_gemm_phase_la<3>), dim3(blocks), dim3(threads), 24 * 1024, s, seed, out, iters, waves);
    else hipLaunchKernelGGL((k_gemm_phase_la<5>), dim3(blocks), dim3(threads), 24 * 1024, s, seed, out, iters, waves);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
