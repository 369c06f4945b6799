// PROTOTYPE / MICROBENCHMARK (not part of the library): one hidden layer of the radiance MLP in WEIGHT-STATIONARY form - the
// question behind the next round's design: how close to the matrix cores' rate does a 256 x 256 layer run
// when a CU keeps the layer's fp16-pair weights in its REGISTERS (256 KB = half a CU's register file) and only the activations
// stream - no weight ring in LDS, no LDS-DMA, no per-chunk barriers, one wave per SIMD?
//   one workgroup per CU, 4 waves; wave w owns output features 64 w .. 64 w + 63 for every sample (64 fragments of 16 x 32
//   weights x 2 planes = 256 registers); per 16-sample tile: fp32 T-layout input [256][16] from global memory (L2-resident in the
//   pipeline this stands for), per-sample maximum, fp16 pair split ONCE per element into an LDS image of the B operand,
//   96 MFMAs (16x16x32 f16, three per fp32 product) per wave, bias + ReLU, gate words, fp32 T-layout output.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/experiments/proto/ws_layer.hip -o tools/experiments/proto/libws_layer.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2(float x0, float x1, float s, uint32_t& h, uint32_t& l) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
}
__device__ __forceinline__ float pow2f(int e) { return __int_as_float((127 + e) << 23); }

struct WsArgs {
    const float* w;     // [256 out][256 in] fp32, already scaled so that fp16 holds it (the library packs with a per-GEMM exponent)
    const float* bias;  // [256]
    const float* x;     // T layout [tiles][256][16]
    float* y;           // T layout [tiles][256][16]
    uint32_t* gates;    // [tiles * 16][8]
    int64_t tiles;
    int64_t x_tiles;    // the input tile of tile t is t % x_tiles (x_tiles small: an L2-resident input, as in the pipeline)
    int mode;           // bit 0: no MFMA; bit 1: no stores; bit 2: no loads (timing ablations)
};

template <int ABL, int ORI>
__global__ __launch_bounds__(256, 1) void k_ws_layer(WsArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][8 * 2 * 64 * 16];  // [buf][ks][plane][lane] 16 B
    __shared__ uint32_t colmax[2][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    // ---- the layer's weights into registers: fragment (ft, ks): W[64 w + 16 ft + c][32 ks + 8 g + j]
    f16x8 wh[4][8], wl[4][8];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const float* p = a.w + (64 * w + 16 * ft + c) * 256 + 32 * ks + 8 * g;
            uint32_t h[4], l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) split2(p[2 * q], p[2 * q + 1], 1.0f, h[q], l[q]);
            wh[ft][ks] = __builtin_bit_cast(f16x8, u32x4{h[0], h[1], h[2], h[3]});
            wl[ft][ks] = __builtin_bit_cast(f16x8, u32x4{l[0], l[1], l[2], l[3]});
        }
    float bias[4][4];  // ORI 0: feature 4 g + i of tile ft; ORI 1: feature c of tile ft (in [ft][0])
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias[ft][i] = a.bias[64 * w + 16 * ft + (ORI ? c : 4 * g + i)];
    if (tid < 32) colmax[tid >> 4][tid & 15] = 0;

    // staging items of this thread: item id = tid + 256 r (r = 0, 1): sample id & 15, feature group id >> 4 (8 features)
    float xin[2][2][8];  // [set][item][feature]
    // Loads by inline asm with HAND-COUNTED waits: hipcc's own counting across the loop's back edge put vmcnt(18) .. (4) in front of
    // the staging - 3 to 12 of the 16 loads issued a moment before had to land, a memory latency exposed per tile.  Every tile
    // issues its 16 loads (past the end: the last tile again) and NST stores, so the distance is static.
    constexpr int NST = ORI ? 5 : 17;
    auto load = [&](int64_t t, int set) {
        if (t >= a.tiles) t = a.tiles - 1;
        const float* xb = a.x + (t % a.x_tiles) * 4096;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int id = tid + 256 * r, cc = id & 15, grp = id >> 4;
            const float* p = xb + (8 * grp) * 16 + cc;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (ABL & 4) xin[set][r][j] = 1.0f;
                else asm volatile("global_load_dword %0, %1, off offset:%2 nt" : "=&v"(xin[set][r][j]) : "v"(p), "n"(j * 64) : "memory");
            }
        }
    };
    auto landed = [&](int set) {  // the loads of `set` were followed by NST stores, 16 loads, NST stores
        if (ABL & 4) return;
        constexpr int N = (ABL & 2) ? 16 : 16 + 2 * NST;
        asm volatile("s_waitcnt vmcnt(%16)"
                     : "+v"(xin[set][0][0]), "+v"(xin[set][0][1]), "+v"(xin[set][0][2]), "+v"(xin[set][0][3]), "+v"(xin[set][0][4]),
                       "+v"(xin[set][0][5]), "+v"(xin[set][0][6]), "+v"(xin[set][0][7]), "+v"(xin[set][1][0]), "+v"(xin[set][1][1]),
                       "+v"(xin[set][1][2]), "+v"(xin[set][1][3]), "+v"(xin[set][1][4]), "+v"(xin[set][1][5]), "+v"(xin[set][1][6]),
                       "+v"(xin[set][1][7])
                     : "n"(N)
                     : "memory");
    };
    auto stage = [&](int buf, int set) {  // per-sample maximum -> exponent -> pairs -> LDS image of the B operand
        landed(set);
        float m[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) v = fmaxf(v, fabsf(xin[set][r][j]));
            m[r] = v;
        }
        atomicMax(&colmax[buf][tid & 15], __float_as_uint(m[0]));         // (the two items of a thread belong to the same sample)
        atomicMax(&colmax[buf][tid & 15], __float_as_uint(m[1]));
        __syncthreads();
        const float top = __uint_as_float(colmax[buf][tid & 15]);
        int e;
        frexpf(top, &e);
        const int ex = top > 0.f ? 14 - e : 0;
        const float s = pow2f(ex < -100 ? -100 : (ex > 100 ? 100 : ex));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int id = tid + 256 * r, cc = id & 15, grp = id >> 4, ks = grp >> 2, gg = grp & 3;
            uint32_t h[4], l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) split2(xin[set][r][2 * q], xin[set][r][2 * q + 1], s, h[q], l[q]);
            *reinterpret_cast<u32x4*>(&lds[buf][((ks * 2 + 0) * 64 + gg * 16 + cc) * 16]) = u32x4{h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4*>(&lds[buf][((ks * 2 + 1) * 64 + gg * 16 + cc) * 16]) = u32x4{l[0], l[1], l[2], l[3]};
        }
        return ex;
    };
    auto ex_of_sample = [&](int buf, int smp) {
        const float top = __uint_as_float(colmax[buf][smp]);
        int e;
        frexpf(top, &e);
        const int ex = top > 0.f ? 14 - e : 0;
        return ex < -100 ? -100 : (ex > 100 ? 100 : ex);
    };

    const int64_t t0 = blockIdx.x, dt = gridDim.x;
    if (t0 >= a.tiles) return;
    load(t0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ex = stage(0, 0);
    load(t0 + dt, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the static distance holds from the first tile of the loop on)
    // one tile: B operand from the LDS image `buf`, products, THEN the loads of the tile after next (register set `set`, free
    // since the image of tile t was built), THEN the epilogue's stores (vmcnt retires in issue order: loads issued behind the
    // stores would make the next staging wait for the stores' write latency), then the image of the next tile
    auto tile = [&](int64_t t, int buf, int set) __attribute__((always_inline)) {
        __syncthreads();  // the image of tile t is complete (and colmax[buf ^ 1] may be cleared)
        if (tid < 16) colmax[buf ^ 1][tid] = 0;
        f16x8 bh[8], bl[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            bh[ks] = *reinterpret_cast<const f16x8*>(&lds[buf][((ks * 2 + 0) * 64 + lane) * 16]);
            bl[ks] = *reinterpret_cast<const f16x8*>(&lds[buf][((ks * 2 + 1) * 64 + lane) * 16]);
        }
        f32x4 acc[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) acc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(ABL & 1)) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) {
                    if constexpr (ORI == 0) {
                        acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ft][ks], bh[ks], acc[ft], 0, 0, 0);
                        acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ft][ks], bl[ks], acc[ft], 0, 0, 0);
                        acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ft][ks], bh[ks], acc[ft], 0, 0, 0);
                    } else {  // rows = samples, columns = features: the same registers, operands swapped
                        acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ks], wl[ft][ks], acc[ft], 0, 0, 0);
                        acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[ks], wh[ft][ks], acc[ft], 0, 0, 0);
                        acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ks], wh[ft][ks], acc[ft], 0, 0, 0);
                    }
                }
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) asm volatile("" ::"v"(bh[ks]), "v"(bl[ks]));
        }
        load(t + 2 * dt, set);
        // ---- epilogue: unscale + bias (one fma), ReLU, gate word, T store
        if constexpr (ORI == 1) {
            // register i of a lane: sample 4 g + i, feature c of tile ft: the four samples are 16 contiguous bytes of the T layout
            // and the wave's store instruction covers features 16 ft .. 16 ft + 15 whole: 1 KB contiguous
            float inv4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) inv4[i] = pow2f(-ex_of_sample(buf, 4 * g + i));
            uint64_t m[4][4];
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = __builtin_fmaf(acc[ft][i], inv4[i], bias[ft][0]);
                    m[ft][i] = __ballot(x > 0.f);
                    v[i] = fmaxf(x, 0.f);
                }
                if (!(ABL & 2)) *reinterpret_cast<f32x4*>(a.y + t * 4096 + (64 * w + 16 * ft + c) * 16 + 4 * g) = v;
                else asm volatile("" ::"v"(v));
            }
            // gate half words of sample 4 g + i: bits 16 g .. 16 g + 15 of the ballots of element i; lane (g, c = i) writes them
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t w0 = (uint32_t)((m[0][i] >> (16 * g)) & 0xffffu) | ((uint32_t)((m[1][i] >> (16 * g)) & 0xffffu) << 16);
                const uint32_t w1 = (uint32_t)((m[2][i] >> (16 * g)) & 0xffffu) | ((uint32_t)((m[3][i] >> (16 * g)) & 0xffffu) << 16);
                lo = c == i ? w0 : lo;
                hi = c == i ? w1 : hi;
            }
            if (c < 4) {
                if (!(ABL & 2)) *reinterpret_cast<uint2*>(a.gates + (t * 16 + 4 * g + c) * 8 + 2 * w) = make_uint2(lo, hi);
                else asm volatile("" ::"v"(lo), "v"(hi));
            }
        }
        const float inv = pow2f(-ex);
        float* yb = a.y + t * 4096 + c;
        uint32_t gate = 0;
        if constexpr (ORI == 0)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = __builtin_fmaf(acc[ft][i], inv, bias[ft][i]);
                gate |= (v > 0.f ? 1u : 0u) << (4 * ft + i);
                v = fmaxf(v, 0.f);
                if (!(ABL & 2)) yb[(64 * w + 16 * ft + 4 * g + i) * 16] = v;
                else asm volatile("" ::"v"(v));
            }
        if (ORI == 1) {
        } else if (!(ABL & 2)) {
            // 16 bits per lane: lanes g = 0..3 of a sample and waves 0..3 make its 256 bits; this lane's half word
            reinterpret_cast<unsigned short*>(a.gates)[((t * 16 + c) * 8 + 2 * w) * 2 + g] = (unsigned short)gate;
        } else {
            asm volatile("" ::"v"(gate));
        }
        if (t + dt < a.tiles) ex = stage(buf ^ 1, set ^ 1);
    };
    for (int64_t t = t0; t < a.tiles; t += 2 * dt) {
        tile(t, 0, 0);
        if (t + dt < a.tiles) tile(t + dt, 1, 1);
    }
}

extern "C" int ws_layer_run(const float* w, const float* bias, const float* x, float* y, uint32_t* gates, int64_t tiles, int64_t x_tiles, int blocks,
                            int mode, void* stream) {
    WsArgs a{w, bias, x, y, gates, tiles, x_tiles, mode};
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_ws_layer<0, 0>), dim3(blocks), dim3(256), 0, s, a); break;
        case 1: hipLaunchKernelGGL((k_ws_layer<1, 0>), dim3(blocks), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_ws_layer<2, 0>), dim3(blocks), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_ws_layer<4, 0>), dim3(blocks), dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL((k_ws_layer<6, 0>), dim3(blocks), dim3(256), 0, s, a); break;
        case 8 + 0: hipLaunchKernelGGL((k_ws_layer<0, 1>), dim3(blocks), dim3(256), 0, s, a); break;
        case 8 + 1: hipLaunchKernelGGL((k_ws_layer<1, 1>), dim3(blocks), dim3(256), 0, s, a); break;
        case 8 + 2: hipLaunchKernelGGL((k_ws_layer<2, 1>), dim3(blocks), dim3(256), 0, s, a); break;
        case 8 + 4: hipLaunchKernelGGL((k_ws_layer<4, 1>), dim3(blocks), dim3(256), 0, s, a); break;
        case 8 + 6: hipLaunchKernelGGL((k_ws_layer<6, 1>), dim3(blocks), dim3(256), 0, s, a); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
