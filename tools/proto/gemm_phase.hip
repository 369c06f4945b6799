// MICROBENCHMARK (not part of the library): the ceiling of a chain kernel's GEMM phase as the fused chains organise it - every wave
// reads its A fragments (weights) from LDS with ds_read_b128 and multiplies them with B operands it holds in registers, three MFMAs per
// fp32 product (fp16 pairs) - WITHOUT the refill DMA, the chunk barriers and the epilogues: how much of the ~60 % matrix-pipe
// utilisation of the real GEMM phases is the shape itself?
//   SHAPE 0: v_mfma_f32_16x16x32_f16, a step = 2 fragment reads (h, l planes, 1 KB each per wave) + 3 MFMAs of 16 cycles
//   SHAPE 1: v_mfma_f32_32x32x16_f16, a step = 2 fragment reads + 3 MFMAs of 32 cycles (half the LDS bytes per FLOP)
// The chunk (24 KB = 12 steps) sits in LDS and is walked `iters` times; accumulators rotate over NT tiles as in chain_gemm.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/proto/gemm_phase.hip -o tools/proto/libgemm_phase.so
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

extern "C" int gemm_phase_run(int shape, int waves, const uint32_t* seed, float* out, int blocks, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int threads = 64 * (waves > 4 ? 8 : 4);
    if (shape == 0) hipLaunchKernelGGL((k_gemm_phase<0, 12>), dim3(blocks), dim3(threads), 24 * 1024, s, seed, out, iters, waves);
    else hipLaunchKernelGGL((k_gemm_phase<1, 4>), dim3(blocks), dim3(threads), 24 * 1024, s, seed, out, iters, waves);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
