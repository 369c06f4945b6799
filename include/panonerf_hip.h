/*
 * panonerf_hip.h — C ABI of libpanonerf_hip.so (gfx950 / MI355X).
 *
 * The reference (Lu-Zhan/Pano-NeRF) has no FFI: its hot path is ATen op chains
 * inside Python functions.  Each entry point below replaces one such chain; the
 * reference file:line it stands in for is cited per function.  The Python module
 * pano_nerf_amd binds these through ctypes (see INTEGRATION.md for the stub a
 * maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32, row-major, contiguous, unless the
 *     parameter name ends in _host; the caller owns all device memory and the library
 *     allocates none;
 *   - process model: one host thread per device (how the reference runs under DDP).  No entry
 *     point keeps results or configuration between calls: every operand, workspace and packed
 *     weight block is passed in.  The only process-wide host state is bookkeeping that never
 *     reaches a result: the pool of hipEvent_t used to fork / join the optional side stream and
 *     by the opt-in launch timing (pn_prof_*), the cached CU count of the device, and one flag
 *     per kernel recording that its dynamic-LDS attribute has been set;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and the
 *     call returns without synchronising (graph-capturable);
 *   - return value: 0 on success, negative PN_ERR_* otherwise (the Python shim
 *     raises RuntimeError(pn_strerror(code)));
 *   - B rays, N samples per ray, S = N + 1 fence posts, M = B*N sample rows.
 *     Sample-row buffers written by the GEMM kernels must be allocated with
 *     pn_pad_rows(M) rows.
 */
#ifndef PANONERF_HIP_H
#define PANONERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN_OK 0
#define PN_ERR_BAD_SHAPE (-1)   /* a size is <= 0 or violates a documented bound   */
#define PN_ERR_UNSUPPORTED (-2) /* e.g. N > PN_MAX_SAMPLES, density channels not 1/5 */
#define PN_ERR_NULL (-3)        /* a required pointer is null                       */
#define PN_ERR_HIP (-4)         /* a HIP launch failed (hipGetLastError != success) */

#define PN_MAX_SAMPLES 512 /* fence posts per ray handled by one wave: S <= 513 */
#define PN_ENC_DIM 96      /* 2 * 3 * (max_deg_point - min_deg_point), degrees 0..15 */
#define PN_VIEW_DIM 27     /* 3 + 2 * 3 * deg_view, deg_view = 4 */
#define PN_WIDTH 256
#define PN_WIDTH_COND 128
#define PN_ROW_PAD 128

const char* pn_strerror(int code);
int pn_abi_version(void);
/* rows a [M, *] sample buffer must be allocated with (M rounded up to PN_ROW_PAD) */
int64_t pn_pad_rows(int64_t m);

/* ---- flat parameter block --------------------------------------------------------
 * The 24 tensors of MLP / PureMLP (models/pano_mip_nerf.py:35-76, models/mip_nerf.py:19-60)
 * live in ONE fp32 block so that gradient all-reduce and Adam are one pass each.
 * Order: layers.0..7 {weight,bias}, extra_layer {w,b}, view_layers.0.0 {w,b},
 * density_layer.weight, color_layer.weight, density_layer.bias, color_layer.bias.
 * pn_param_layout fills offsets[24] (in floats, same order as above) and returns the
 * total float count (613768 for nc = 5, 612740 for nc = 1), or a negative error. */
int64_t pn_param_layout(int num_density_channels, int64_t* offsets_host);

/* workspace (floats) pn_pack_weights needs: transposed / split copies of the weights */
int64_t pn_wpack_floats(int num_density_channels);
/* (re)build the packed weights from the flat parameter block; call after every
 * optimizer step.  Replaces nothing upstream (ATen reads nn.Linear.weight directly). */
int pn_pack_weights(const float* params, int num_density_channels, float* wpack, void* stream);

/* ---- ray generation ----------------------------------------------------------------
 * PanoDataset._generate_rays, datasets/pano_datasets.py:152-216 (== sample_dir_by_pano,
 * utils/sampling.py:5-20).  One camera; outputs are [H*W, C] with C = 3,3,3,1,1,1,1,1. */
int pn_raygen_pano(int H, int W, const float* c2w_host /*[16] row-major 4x4*/, float near_, float far_,
                   float* origins, float* directions, float* viewdirs, float* radii, float* lossmult,
                   float* near_out, float* far_out, float* noise_var, void* stream);
/* PanoDataset.generate_lit_rays, datasets/pano_datasets.py:218-263 (== sample_dir_by_unifrom,
 * utils/sampling.py:23-38): fp64 math, stored as IEEE half.  out_half: 14*D uint16 laid out as
 * origins[D,3] directions[D,3] viewdirs[D,3] radii[D] lossmult[D] near[D] far[D] noise_var[D]. */
int pn_lit_rays(int D, double radius, double near_, double far_, uint16_t* out_half, void* stream);

/* Training-batch gather out of a device-resident ray pool — the host-side __getitem__ gather of
 * datasets/pano_datasets.py:271-275 (one DataLoader worker per 56-byte ray upstream).
 *   idx [B] int64 (device): pool row of every batch ray (out-of-range rows read row 0);
 *   pool_host / out_host: HOST arrays of 9 device pointers in the order origins, directions, viewdirs [.,3],
 *   radii, lossmult, near, far, noise_var [.,1], rgb [.,3]; entry 8 (rgb) may be null in both. */
int pn_gather_rays(int64_t B, int64_t pool_rays, const int64_t* idx, const float* const* pool_host,
                   float* const* out_host, void* stream);

/* Training-batch sampler that REGENERATES the rays instead of reading a stored pool (replaces
 * PanoDataset.__getitem__, datasets/pano_datasets.py:271-275, over the pool of _generate_rays, :152-216): batch ray b is
 * pixel idx[b] % (H W) of camera idx[b] / (H W), computed with the arithmetic of pn_raygen_pano (bit-identical to a
 * gather out of its output).  c2ws: DEVICE [n_cam,16] row-major 4x4; rgb_pool [n_cam H W,3] / rgb_out [B,3]: target
 * colours, both or neither; out-of-range idx reads ray 0. */
int pn_sample_pano_rays(int64_t B, int n_cam, int H, int W, const int64_t* idx, const float* c2ws, float near_, float far_,
                        const float* rgb_pool, float* origins, float* directions, float* viewdirs, float* radii,
                        float* lossmult, float* near_out, float* far_out, float* noise_var, float* rgb_out, void* stream);

/* ---- sampling ----------------------------------------------------------------------
 * sample_along_rays (models/mip.py:113-151; disparity != 0: positions linear in inverse depth, :134-136) + cast_rays (67-89) +
 * conical_frustum_to_gaussian (36-64, stable) + lift_gaussian (8-22, diagonal).
 * t_rand: [B,S] uniforms or null (deterministic). */
int pn_sample_coarse(int64_t B, int N, int disparity, const float* origins, const float* directions, const float* radii,
                     const float* near_, const float* far_, const float* t_rand, float* t_out, float* mean,
                     float* cov, void* stream);
/* resample_along_rays (models/mip.py:304-352, stop_grad branch) + sorted_piecewise_constant_pdf
 * (240-301) + cast_rays.  u_rand: [B,S] uniforms in [0, 1/S - eps) or null. */
int pn_resample(int64_t B, int N, const float* t_in, const float* weights, float padding, const float* u_rand,
                const float* origins, const float* directions, const float* radii, float* t_out, float* mean,
                float* cov, void* stream);
/* sample_each_points (models/mip.py:154-194) for x_surf = origins + directions * distance
 * (models/pano_mip_nerf.py:324-334).  Light ray r = b*D + j.  env_rand: [Ne+1] uniforms or null.
 * env_dirs/env_radii/env_near/env_far: [D,*] fp32 (fp16-rounded values up-cast by the caller). */
int pn_sample_env(int64_t B, int D, int Ne, const float* origins, const float* directions, const float* distance,
                  const float* env_dirs, const float* env_radii, const float* env_near, const float* env_far,
                  const float* env_rand, float* t_out /*[B*D,Ne+1]*/, float* mean /*[B*D*Ne,3]*/,
                  float* cov /*[B*D*Ne,3]*/, void* stream);

/* ---- encodings ---------------------------------------------------------------------
 * integrated_pos_enc (models/mip.py:394-428, diagonal) -> enc [Mpad,96] */
int pn_ipe_encode(int64_t M, const float* mean, const float* cov, float* enc, void* stream);
/* pos_enc (models/mip.py:431-441, min_deg 0, max_deg 4, identity prepended) -> [R,27] */
int pn_pos_enc_view(int64_t R, const float* viewdirs, float* viewenc, void* stream);

/* ---- MLP ---------------------------------------------------------------------------
 * MLP.forward / PureMLP.forward (models/pano_mip_nerf.py:95-114, models/mip_nerf.py:81-102).
 * Rows are samples; sample row i belongs to view row (i / rows_per_ray) % view_mod
 * (view_mod = number of rows of viewdirs).  acts: [PN_ACT_SLOTS][Mpad,256] saved activations
 * (h0..h7, bottleneck, view hidden [.,128 used]) — required (backward and the density
 * gradient re-read them).  enc: [Mpad,96] (written).  viewenc: [view_rows,27] (written),
 * viewbias: [view_rows,128] scratch. */
#define PN_ACT_SLOTS 10
/* masks: [PN_MASK_SLOTS][Mpad][PN_MASK_WORDS] u32 — ReLU gates of h0..h7 and the view hidden as bit masks
 * (written here, read by pn_density_grad / pn_mlp_backward: 32 B/row instead of re-reading 1 KB/row). */
#define PN_MASK_SLOTS 9
#define PN_MASK_WORDS 8
int pn_mlp_forward(int64_t M, int rows_per_ray, int64_t view_rows, int num_density_channels, const float* params,
                   const float* wpack, const float* mean, const float* cov, const float* viewdirs, float* enc,
                   float* viewenc, float* viewbias, float* acts, uint32_t* masks, float* raw_rgb /*[M,3]*/,
                   float* raw_density /*[M,nc]*/, void* stream);

/* d sigma / d mean per sample: what vmap(jacrev(compute_graph))[1] keeps
 * (models/pano_mip_nerf.py:299-303, models/mip_nerf.py:261-265), computed as ONE reverse
 * sweep seeded with softplus'(raw_density0 + bias) * density_layer.weight[0].
 * rsweep: [8][Mpad,256] saved sweep vectors (needed by pn_mlp_backward's second-order term);
 * scratch: [Mpad,96].  Output grad_mean [M,3] = + d sigma / d mean (caller negates). */
int pn_density_grad(int64_t M, int num_density_channels, float density_bias, const float* params,
                    const float* wpack, const float* mean, const float* cov, const float* acts,
                    const uint32_t* masks, const float* raw_density, float* rsweep, float* scratch,
                    float* grad_mean, void* stream);

/* Backward of pn_mlp_forward (+ optionally of pn_density_grad).  Accumulates (+=) into
 * `grads` (flat block, layout of pn_param_layout).
 *   d_raw_rgb [M,3], d_raw_density [M,nc]: upstream gradients;
 *   v_gradmean [M,3] or null: upstream gradient w.r.t. pn_density_grad's output (second-order
 *     path: needs rsweep from pn_density_grad; d_raw_density[:,0] receives the
 *     softplus'' term internally);
 *   d_mean [M,3] or null: if non-null receives d loss / d mean (first-order, env-light path).
 * M is a whole number of rays (M % rows_per_ray == 0) and the rays cycle through the view rows
 * ((M / rows_per_ray) % view_rows == 0): ray r uses viewenc[r % view_rows], as in pn_mlp_forward.
 * work: scratch of pn_mlp_backward_work_floats(M, rows_per_ray, view_rows, M_batched) floats.
 * Batching the weight gradients of one training step: the evaluations of a step (env light, level 1, level 0) share
 * the weights, so their trunk / extra-layer weight gradients are ONE TN GEMM per layer over all their rows.  Calls with
 * defer_wgrad = 1 skip those GEMMs and leave their operands in `work` (keep it alive); the last call passes
 * n_deferred (<= 2) and, per deferred evaluation, its M, enc, acts, rsweep (or null), work and whether it ran the
 * tangent sweep (host arrays), and reduces everything.  M_batched = total rows of those GEMMs (own + deferred, tangent
 * rows counted) sizes the slab part of `work` (0 = stand-alone).
 * side_stream (nullable): a second hipStream_t; when given, the weight-gradient GEMMs / reductions run there,
 * forked from and joined back to `stream` with events inside the call, so they overlap the data-gradient chain. */
int64_t pn_mlp_backward_work_floats(int64_t M, int rows_per_ray, int64_t view_rows, int64_t M_batched);
int pn_mlp_backward(int64_t M, int rows_per_ray, int64_t view_rows, int num_density_channels, float density_bias,
                    const float* params, const float* wpack, const float* mean, const float* cov,
                    const float* enc, const float* viewenc, const float* acts, const uint32_t* masks,
                    const float* raw_density, const float* d_raw_rgb, const float* d_raw_density,
                    const float* rsweep, const float* v_gradmean, float* d_mean, float* grads, float* work,
                    int64_t M_batched, int defer_wgrad, int n_deferred, const int64_t* dM_host,
                    const float* const* denc_host, const float* const* dacts_host,
                    const float* const* drsweep_host, float* const* dwork_host, const int* dtangent_host,
                    void* stream, void* side_stream);


/* ---- fused on-chip MLP chains (pn_chain.hip) ---------------------------------------------------------------------
 * The same MLP (models/pano_mip_nerf.py:95-114, models/mip_nerf.py:81-102), encodings (models/mip.py:394-441) and
 * their reverse / forward-mode passes as ONE kernel per pass: every layer is computed transposed on
 * v_mfma_f32_32x32x16_bf16, a wave carries the activations of its 32 samples from layer to layer in registers, the
 * weights stream through an LDS ring by LDS-DMA.  planes = 3: exact 3-term bf16 split, six partial products (fp32
 * accuracy); planes = 2: fp16 pairs (x 2^e = h + l, |error| < 2^-24 |x|), three partial products, one power-of-two scale
 * per weight matrix and per sample (chains) or per tensor (weight gradients), fp32 accumulate; planes = 1: plain bf16
 * operands, fp32 accumulate.  Sample-row tensors these kernels exchange are in the "T layout":
 * elem[Mp/tile][F][tile] (sample-minor, tile = pn_chain_tile(), Mp = pn_pad_rows(M)), elem = float for planes 3 and 2,
 * bf16 for planes = 1 (the stored value is the bf16 the next GEMM consumes: the float* parameters below then point at
 * 2-byte elements, and a buffer sized in floats is twice as large as needed); gate words are uint32 [9][Mp][8]
 * (per row: tile-dependent lane-group order, see pn_chain.hip; producers and consumers are all in this library). */
/* samples per block of the sample-minor tensors (= samples per wave of the chain kernels): 16 (v_mfma_f32_16x16x32_bf16,
 * two waves per SIMD; the build default) or 32 (v_mfma_f32_32x32x16_bf16, -DPN_CHAIN_TILE=32).  Below, "T layout" means
 * elem[Mp / tile][F][tile]. */
int pn_chain_tile(void);
/* "Q24": with planes = 2 and 16-sample tiles, the 256-wide tensors that only pn_chain_wgrad reads back are stored in THREE bytes per
 * element - fp32 rounded to 16 significant bits (round to nearest on the dropped byte), the four features of a quad block of a
 * sample in 12 bytes: elem[Mp / 16][F / 4][16][12 B], byte b of a feature's three = bits 8 (b + 1) .. 8 (b + 1) + 7 of the rounded
 * fp32 - a quarter of the step's HBM traffic in these tensors; each keeps its slot's address (slot * Mp * 256 floats) and uses the
 * first three quarters of it.  Bit s of the result: slot s of the tensor is Q24; tensor 0: activations h_s (acts_t), 1: tangents
 * hdot_s (tang_t), 2: deltas (delta_t), 3: reverse-sweep vectors r_s (rs_t).  0 for every other mode: all fp32 (or bf16). */
int pn_chain_q24_slots(int planes, int t_format, int tensor);
/* Two arguments every chain entry point below takes (ABI 2):
 *   t_format  0: every T tensor fp32 (bf16 with planes = 1); 1: Q24 where pn_chain_q24_slots says so (planes = 2 only).  The calls of
 *             one evaluation - forward, sweeps, backward, weight gradients - must agree on it: it is the layout of what they exchange.
 *   max_wgs   0: the launch sizes its persistent grid to every CU of the device.  > 0: it occupies at most that many workgroups (a chain
 *             workgroup or a 256-wide weight-gradient workgroup fills a CU), so that a kernel of the OTHER family, launched on another
 *             stream with the complementary budget, finds the remaining CUs free: the chains are bound by matrix-instruction issue and
 *             their stores (2.7 TB/s of HBM traffic), the weight gradients by their operand reads - side by side on disjoint CUs each
 *             sees less HBM contention than alone on the whole chip.  Results do not depend on it for the
 *             chains; for pn_chain_wgrad it moves the split points of the sum over samples (deterministic for a given value). */
int64_t pn_chain_pack_bytes(int planes);
int pn_chain_pack(const float* params, int num_density_channels, int planes, void* pack, void* stream);
/* floats of acts_t: h0..h7 [256] x 8, bottleneck | view encoding [288], view hidden [128].  acts_t may be NULL in
 * pn_chain_forward (inference: nothing re-reads the activations; enc_t and the gate words are still written). */
int64_t pn_chain_acts_floats(int64_t M);
/* amax (planes = 2, training; NULL otherwise): ONE evaluation's table of pn_chain_amax_slots() uint32, the largest |x| of
 * every T tensor the weight gradients will read (float bits).  pn_chain_forward clears it, the four chain kernels of the
 * evaluation add their tensors' maxima, pn_chain_wgrad derives one power-of-two scale per tensor from it. */
int pn_chain_amax_slots(void);
/* view_tab: scratch of view_rows * 32 floats - the view encoding (pos_enc, models/mip.py:431-441) depends on the view row
 * only, so it is evaluated once per row (a small kernel in front of the chain) and read by the row's samples. */
int pn_chain_forward(int64_t M, int rows_per_ray, int64_t view_rows, int num_density_channels, int planes,
                     const void* pack, const float* mean, const float* cov, const float* viewdirs, float* view_tab,
                     float* enc_t, float* acts_t, uint32_t* masks, float* raw_rgb /*[M,3]*/, float* raw_density /*[M,nc]*/,
                     uint32_t* amax, int t_format, int max_wgs, void* stream);
/* vmap(jacrev(compute_graph))[1] (models/pano_mip_nerf.py:299-303) as one reverse sweep.  keep_all != 0: rs_t is
 * T [8][Mp*256] and receives r_0..r_7 (the second-order weight gradients need them); keep_all = 0 (inference): rs_t is ONE
 * slot T [Mp*256], used only for the kernel's own reload of r_5. */
int pn_chain_density_grad(int64_t M, int num_density_channels, int planes, float density_bias, const float* params,
                          const void* pack, const float* mean, const float* cov, const uint32_t* masks,
                          const float* raw_density, float* rs_t, int keep_all, float* grad_mean /*[M,3]*/,
                          uint32_t* amax, int t_format, int max_wgs, void* stream);
/* forward-mode tangent sweep along v_gradmean (the double backward of the normals block) */
int pn_chain_tangent(int64_t M, int num_density_channels, int planes, const float* params, const void* pack,
                     const float* mean, const float* cov, const uint32_t* masks, const float* v_gradmean,
                     float* edot_t /*T [Mp*96]*/, float* tang_t /*T [8][Mp*256]*/, float* sdot /*[M]*/, uint32_t* amax,
                     int t_format, int max_wgs, void* stream);
/* data-gradient chain.  drgb_t T [Mp*32], d8_t T [Mp*288], coef_t T [Mp*32]: with pn_chain_tile() = 32 the caller
 * zero-fills them once (the kernel writes 16 of their 32 padded features); with the default 16 the kernel writes them
 * whole.  sdot / coef_t: second-order path (both or neither); d_mean nullable. */
int pn_chain_backward(int64_t M, int num_density_channels, int planes, float density_bias, const void* pack,
                      const uint32_t* masks, const float* raw_density, const float* d_raw_rgb,
                      const float* d_raw_density, const float* sdot, const float* mean, const float* cov,
                      float* drgb_t, float* dhv_t /*T [Mp*128]*/, float* d8_t, float* delta_t /*T [8][Mp*256]*/,
                      float* coef_t, float* d_mean /*[M,3]*/, uint32_t* amax, int t_format, int max_wgs, void* stream);
/* one evaluation's tensors for the weight gradients (host struct of device pointers) */
typedef struct PnChainEval {
    int64_t M;
    const float* enc_t;
    const float* acts_t;
    const float* drgb_t;
    const float* dhv_t;
    const float* d8_t;
    const float* delta_t;
    const float* rs_t;   /* second-order rows: all four or none */
    const float* edot_t;
    const float* tang_t;
    const float* coef_t;
    const uint32_t* amax; /* planes = 2: the evaluation's table of maxima; NULL otherwise */
} PnChainEval;
int64_t pn_chain_wgrad_work_floats(void);
/* which: bit 1 = the second-order TRUNK rows (r_l^T hdot_{l-1}, r_0^T edot) of the evaluations that carry rs_t - their operands exist as
 * soon as the tangent sweep has run, before the evaluation's backward chain, so a caller can reduce them under that chain on another
 * stream; bit 0 = everything else (delta_l^T h_{l-1}, heads incl. the softplus' row against hdot_7, view / colour layer, bias gradients).
 * grads is accumulated into (+=) by the job reductions, in launch order: concurrent calls on the same `grads` must share a stream. */
int pn_chain_wgrad(int n_evals, const PnChainEval* evals_host, int num_density_channels, int planes, float* grads,
                   float* work, int64_t work_floats, int which, int t_format, int max_wgs, void* stream);

/* ---- volumetric rendering ---------------------------------------------------------
 * compute_graph activations (models/pano_mip_nerf.py:273-278) + volumetric_rendering
 * (models/mip.py:444-483).  R rays of N samples; dirs [R or dir_mod, 3] (ray r uses
 * dirs[r % dir_mod]).  Outputs comp_rgb [R,3], distance [R], acc [R], weights [R,N]. */
int pn_composite_forward(int64_t R, int N, int num_density_channels, float density_bias, float rgb_padding,
                         int white_bkgd, const float* raw_rgb, const float* raw_density, const float* t,
                         const float* dirs, int64_t dir_mod, float* comp_rgb, float* distance, float* acc,
                         float* weights, void* stream);
/* adjoint: d_comp_rgb [R,3], d_distance [R] (nullable), d_weights [R,N] (nullable) ->
 * d_raw_rgb [R*N,3] (=), d_raw_density [R*N,nc] channel 0 (=; other channels untouched). */
int pn_composite_backward(int64_t R, int N, int num_density_channels, float density_bias, float rgb_padding,
                          int white_bkgd, const float* raw_rgb, const float* raw_density, const float* t,
                          const float* dirs, int64_t dir_mod, const float* d_comp_rgb, const float* d_distance,
                          const float* d_weights, float* d_raw_rgb, float* d_raw_density, void* stream);

/* ---- normals / albedo gather (models/pano_mip_nerf.py:296-317, mip_nerf.py:258-275) --
 * grad_mean [B*N,3] = d sigma/d mean; weights [B,N]; raw_density [B*N,nc].
 * normals_s = normalize(-grad_mean); normal = normalize(sum w^ n_s); ort_ray[b] = sum w^ relu(n_s.d)^2
 * (caller takes the mean over B); albedo = sum w^ (sigmoid(raw[1:4])*0.77+0.03) (nc = 5 only). */
int pn_surf_gather_forward(int64_t B, int N, int num_density_channels, const float* grad_mean,
                           const float* weights, const float* raw_density, const float* directions,
                           float* normal /*[B,3]*/, float* ort_ray /*[B] or null*/, float* albedo /*[B,3] or null*/,
                           void* stream);
/* adjoint: d_normal [B,3], d_ort_ray [B] (nullable), d_albedo [B,3] (nullable) ->
 * d_weights [B,N] (=), v_gradmean [B*N,3] (=), d_raw_density channels 1..3 (=, nc = 5). */
int pn_surf_gather_backward(int64_t B, int N, int num_density_channels, const float* grad_mean,
                            const float* weights, const float* raw_density, const float* directions,
                            const float* d_normal, const float* d_ort_ray, const float* d_albedo,
                            float* d_weights, float* v_gradmean, float* d_raw_density, void* stream);

/* ---- Lambertian surface rendering (utils/surface_rendering.py:104-126, 129-165) -------
 * env_rgb [B,D,3], albedo/normal [B,3], env_dirs [D,3], solid_angle [D].
 * -> surface_rgb (= diffuse) [B,3], shading [B,3]. */
int pn_surface_forward(int64_t B, int D, const float* env_rgb, const float* albedo, const float* normal,
                       const float* env_dirs, const float* solid_angle, float* diffuse, float* shading,
                       void* stream);
int pn_surface_backward(int64_t B, int D, const float* env_rgb, const float* albedo, const float* normal,
                        const float* env_dirs, const float* solid_angle, const float* d_diffuse,
                        const float* d_shading, float* d_env_rgb, float* d_albedo, float* d_normal, void* stream);
/* d x_surf -> d distance: d_distance[b] (+=) sum_{rows of ray b} d_mean[row] . directions[b]
 * (models/pano_mip_nerf.py:324; rows_per_ray = D*Ne). */
int pn_env_origin_backward(int64_t B, int rows_per_ray, const float* d_mean, const float* directions,
                           float* d_distance, void* stream);

/* ---- tone-mapped loss (utils/surface_rendering.py:319-344, systems/panonerf_system.py:17,44-67,
 * systems/mipnerf_system.py:24,37-45).  Per-ray partials are reduced in-kernel; `loss_terms`
 * receives [mse_coarse, mse_fine, mse_surface, chrom, mask_sum, total]; d_* receive the gradients of
 *   total = cw*mse_coarse + mse_fine + sw*mse_surface + chw*chrom     (ort term added by caller).
 * rgb_surface / albedo may be null (terms skipped).  work: >= 8 + 8*ceil(B/256) floats. */
int pn_tonemap_loss(int64_t B, const float* rgb_gt_hdr, const float* lossmult, const float* rgb_coarse,
                    const float* rgb_fine, const float* rgb_surface, const float* albedo, float coarse_w,
                    float surface_w, float chrom_w, float* loss_terms /*[8]*/, float* d_coarse, float* d_fine,
                    float* d_surface, float* d_albedo, float* work, void* stream);

/* ---- optimizer: Adam over the flat block (torch.optim.Adam defaults,
 * systems/base_system.py:82) ; grads are scaled by grad_scale first (1/world_size after a
 * sum all-reduce). */
int pn_adam_step(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float lr,
                 float beta1, float beta2, float eps, int step, float grad_scale, void* stream);

/* same update with the step counter (incremented first) and the learning rate read from device memory, so the
 * call can be captured once in a HIP graph and replayed every step */
int pn_adam_step_dev(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                     const float* lr_dev, float beta1, float beta2, float eps, int* step_dev, float grad_scale,
                     void* stream);

/* ---- building blocks exposed for tests / profiling ------------------------------------
 * C[M,N] = epi(A[M,K] * Bt[N,K]^T) on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * flags: 1 = +bias[N], 2 = relu, 4 = gate by (gate[row,col] > 0). lda/ldb/ldc/ldg in floats. */
int pn_gemm_nt(int64_t M, int N, int K, const float* A, int lda, const float* Bt, int ldb, float* C, int ldc,
               const float* bias, const float* gate, int ldg, int flags, void* stream);
/* C[N1,N2] = X[M,N1]^T * Y[M,N2] (split over rows, deterministic two-pass reduction);
 * accumulate != 0 adds into C.  work: pn_gemm_tn_work_floats(M, N1, N2) floats. */
int64_t pn_gemm_tn_work_floats(int64_t M, int N1, int N2);
int pn_gemm_tn(int64_t M, int N1, int N2, const float* X, int ldx, const float* Y, int ldy, float* C, int ldc,
               int accumulate, float* work, void* stream);

/* ---- launch timing (bench.py roofline leg; off by default) ------------------------------------
 * pn_prof_enable(on): bit 0 switches the timing on or off; while on, every GEMM / chain launch is bracketed by HIP
 * events on its own stream (the other bits are ignored: the ablation switches of the tools/ micro-benchmarks exist only
 * in -DPN_ABLATE builds).  pn_prof_read waits for the recorded events and returns, for kernel class cls (0 k_gemm_nt,
 * 1 k_gemm_tn, 2 k_chain_fwd, 3 k_chain_dgrad, 4 k_chain_tangent, 5 k_chain_bwd, 6..10 k_chain_wgrad by tile configuration
 * 256x256 / 256x96 / 128x288 / 32x256 / 32x128 on fp32 tensors; 11 / 12 the 256x256 tile with Y / with X and Y in Q24), the summed duration in ms, the launch count and the summed algorithmic FLOPs (2*M*N*K, unpadded). */
/* diagnostic: `blocks` workgroups x 4 waves each issue 4*iters back-to-back fp32 MFMAs (no memory traffic) */
int pn_mfma_probe(float* out, int blocks, int iters, void* stream);
int pn_prof_enable(int on);
int pn_prof_read(int cls, double* total_ms, int64_t* launches, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* PANONERF_HIP_H */
