/*
 * pvd_hip.h -- C ABI of libpvd_hip.so, the MI355X (gfx950) native volume-rendering
 * inner loop for PVD's distillation trainer.
 *
 * This is the drop-in boundary.  Each entry point replaces one
 * function of the reference's three pybind11 modules; the reference interface it
 * replaces is cited as <file>:<line> under /root/reference.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes, no torch types; every pointer is a DEVICE
 *     pointer on the current HIP device unless stated otherwise;
 *   - the caller allocates everything; the library never allocates, frees or
 *     synchronises (reference: "Python allocates everything",
 *     raymarching/raymarching.py:240-250, gridencoder/grid.py:55-64);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  The
 *     reference launches on the legacy default stream with no device guard
 *     (raymarching.cu:156); here the caller passes torch's current stream;
 *   - returns PVD_OK (0) or a negative pvd_status; no exceptions cross the ABI.
 *     PVD_ERR_UNSUPPORTED corresponds to the reference's
 *     `throw std::runtime_error{"GridEncoding: C must be 1, 2, 4, or 8."}`
 *     (gridencoder.cu:355,370);
 *   - re-entrant: no entry point keeps state between calls (the reference constructs
 *     its RNG per call, raymarching.cu:488,816 -- so do we, on the device).  The ONE
 *     exception is the pair of measurement knobs pvd_grid_set_variant /
 *     pvd_grid_set_fwd_kernel: they set a process-wide choice among kernels whose
 *     results are identical (bit-identical forward), are read once per launch, are meant
 *     to be set before any concurrent use and never need to be called at all.
 *
 * Buffers that the reference's Python zero-fills before the call (xyzs, dirs,
 * deltas, grad_sigmas, grad_rgbs, grad_embeddings, grad_inputs) must arrive
 * zero-filled here too; `rays`, `nears`, `fars`, `outputs` may be uninitialised.
 */
#ifndef PVD_HIP_H
#define PVD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *pvd_stream_t; /* hipStream_t */

typedef enum {
    PVD_OK = 0,
    PVD_ERR_INVALID = -1,     /* null pointer / bad size */
    PVD_ERR_UNSUPPORTED = -2, /* D, C, L, degree or dtype outside what the reference dispatches */
    PVD_ERR_LAUNCH = -3       /* hipGetLastError() != hipSuccess after a launch */
} pvd_status;

/* table / activation element type of the grid encoder (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * gridencoder.cu:438; double is never used on this path and is not provided) */
typedef enum { PVD_F32 = 0, PVD_F16 = 1 } pvd_dtype;

int pvd_abi_version(void);
const char *pvd_status_string(int status);
/* last hipError_t name seen by this thread after a failed launch ("" if none) */
const char *pvd_last_hip_error(void);

/* ------------------------------------------------------------------------
 * _raymarching  (raymarching/src/raymarching.h:7-19, bindings.cpp:5-20)
 * ---------------------------------------------------------------------- */

/* near_far_from_aabb -- raymarching.cu:150-158 (kernel :93-147).
 * rays_o, rays_d [N,3] f32; aabb [6] f32; nears, fars [N] f32. */
int pvd_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                           uint32_t N, float min_near, float *nears, float *fars, pvd_stream_t stream);

/* get_rays -- torch code in the reference: distill_mutual/utils.py:324-404.  pose [4,4] f32 row-major cam2world
 * (device), intrinsics by value, inds [N] i64 flat pixel ids (k = row*W + col) or NULL for k = n.
 * rays_o, rays_d [N,3] f32.  d = normalise(((i+.5-cx)/fx, (j+.5-cy)/fy, 1)) rotated by pose[:3,:3]. */
int pvd_get_rays(const float *pose, float fx, float fy, float cx, float cy, const int64_t *inds, uint32_t W, uint32_t N,
                 float *rays_o, float *rays_d, pvd_stream_t stream);

/* One training batch in one launch: what Trainer.train_one_epoch + get_rays + run_cuda's near_far_from_aabb do with
 * ~10 torch ops (utils.py:354, 987-995; renderer.py:339): N random pixels of pose poses[state[0]] (PCG32 keyed by seed,
 * batch counter and ray), their rays, a random background colour per ray (bg may be NULL), near/far against aabb.
 * state: DEVICE int64[3] = {pose index, batch counter, 0}; advanced by the kernel (pose index cycles through P poses). */
int pvd_make_ray_batch(const float *poses, uint32_t P, int64_t *state, uint64_t seed, float fx, float fy, float cx,
                       float cy, uint32_t H, uint32_t W, uint32_t N, const float *aabb, float min_near, int64_t *inds,
                       float *rays_o, float *rays_d, float *bg, float *nears, float *fars, pvd_stream_t stream);

/* polar_from_ray -- raymarching.cu:203-211 (kernel :164-200).  coords [N,2]. */
int pvd_polar_from_ray(const float *rays_o, const float *rays_d, float radius,
                       uint32_t N, float *coords, pvd_stream_t stream);

/* morton3D / morton3D_invert -- raymarching.cu:231-234, 259-262.  coords [N,3] i32, indices [N] i32. */
int pvd_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, pvd_stream_t stream);
int pvd_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, pvd_stream_t stream);

/* packbits -- raymarching.cu:294-302.  grid [N*8] f32 -> bitfield [N] u8, bit i = grid[8n+i] > thresh. */
int pvd_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, pvd_stream_t stream);

/* march_rays_train -- raymarching.cu:485-494 (kernel :313-483).
 * grid: density bitfield [C*H^3/8] u8.  xyzs/dirs [M,3], deltas [M,2] f32 (zero-filled by caller),
 * rays [N,3] i32 (id, offset, count), counter [2] i32 (points, rays; accumulated like the
 * reference's atomicAdd, so the caller zeroes it).
 * Slot allocation is a deterministic exclusive prefix sum in ray order (row n of `rays`
 * is ray n): one of the orders the reference's atomics (:408-409) can produce.
 * A ray is written only if offset + count < M (strict, :419). */
int pvd_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid,
                         float bound, float dt_gamma, uint32_t max_steps,
                         uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                         const float *nears, const float *fars,
                         float *xyzs, float *dirs, float *deltas,
                         int32_t *rays, int32_t *counter, uint32_t perturb, pvd_stream_t stream);

/* Same, with a scratch buffer: pvd_march_workspace_bytes(N) bytes of device memory (256 B per ray).  The count pass
 * then records, per ray, the emit masks of the lattice chunks that produced samples, and the write pass rebuilds the
 * samples from them and folds the scan in (2 launches, no second walk of the occupancy grid).  Results are identical
 * to pvd_march_rays_train; workspace == NULL (or too small, or dt_gamma != 0, or N > 16384) takes that path.
 * flags & PVD_MARCH_FRESH: xyzs / dirs / deltas and counter arrive UNINITIALISED (the reference's wrapper zero-fills
 *   them first, raymarching.py:240-242, and run_cuda clears the counter, renderer.py:374): the counter is treated as
 *   {0, 0} and every output element no ray owns is written as zero by the march itself -- same results, no fill launches. */
#define PVD_MARCH_FRESH 1u
size_t pvd_march_workspace_bytes(uint32_t N);
int pvd_march_rays_train_ws(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                            uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears,
                            const float *fars, float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                            uint32_t perturb, void *workspace, size_t workspace_bytes, uint32_t flags,
                            const int32_t *budget_dev, pvd_stream_t stream);
/* budget_dev (here and in pvd_composite_rays_train_bg_*): NULL, or a DEVICE int32 holding the LOGICAL sample budget: rays
 * are dropped against min(M, *budget_dev) (the reference's `point_index + num_steps >= M`, raymarching.cu:419, with the
 * running mean of the sample count as M, renderer.py:773-775), while M rows stay allocated and initialised.  A captured
 * training step can then follow update_extra_state's new mean_count without being re-captured. */

/* composite_rays_train_forward -- raymarching.cu:585-593 (kernel :504-582). */
int pvd_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                     const int32_t *rays, uint32_t M, uint32_t N,
                                     float *weights_sum, float *depth, float *image, pvd_stream_t stream);

/* composite_rays_train_backward -- raymarching.cu:689-697 (kernel :606-686).
 * grad_sigmas [M], grad_rgbs [M,3] zero-filled by caller. */
int pvd_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                      const float *sigmas, const float *rgbs, const float *deltas,
                                      const int32_t *rays, const float *weights_sum, const float *image,
                                      uint32_t M, uint32_t N, float *grad_sigmas, float *grad_rgbs,
                                      pvd_stream_t stream);

/* march_rays -- raymarching.cu:814-822 (kernel :704-811).  xyzs/dirs [n_alive*n_step(+pad),3],
 * deltas [..,2] zero-filled by caller.  perturb doubles as the RNG seed (:816). */
int pvd_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                   const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                   uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars,
                   float *xyzs, float *dirs, float *deltas, uint32_t perturb, pvd_stream_t stream);

/* composite_rays -- raymarching.cu:912-918 (kernel :825-909).  In place on rays_t, weights_sum, depth, image. */
int pvd_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, float *rays_t,
                       const float *sigmas, const float *rgbs, const float *deltas,
                       float *weights_sum, float *depth, float *image, pvd_stream_t stream);

/* compact_rays -- raymarching.cu:942-948 (kernel :921-939).  alive_counter [1] i32 (caller zeroes).
 * One atomic per workgroup (wave ballot + scan inside it): survivors keep their relative order inside a
 * 256-ray workgroup; the order of workgroups is whatever the atomic gives, as in the reference. */
int pvd_compact_rays(uint32_t n_alive, int32_t *rays_alive, const int32_t *rays_alive_old,
                     float *rays_t, const float *rays_t_old, int32_t *alive_counter, pvd_stream_t stream);

/* Inference rounds with the round state on the device (the reference reads the alive count back every round,
 * distill_mutual/renderer.py:488): state = int32[8] = {cnt[2], n_alive, n_step, rows, steps_done, rounds, pad}.  Round i:
 *   (i > 0) pvd_infer_compact(parity = i & 1): survivors (rays_t >= 0) of the previous round's lists -> the new lists,
 *           counted in cnt[parity];
 *   pvd_infer_round_begin(parity): n_alive = cnt[parity] (0 once steps_done >= max_steps), n_step = max(min(N / n_alive, 8), 1)
 *           (renderer.py:493), rows = n_alive * n_step, steps_done += n_step, cnt[parity ^ 1] = 0;
 *   pvd_infer_march: pvd_march_rays for the first n_alive rays, rows beyond a ray's last sample zero-filled;
 *   (model forward on `rows` rows: the rows_dev argument of the forward entry points)
 *   pvd_infer_composite: pvd_composite_rays with sigmas * sigma_scale.
 * n_upper: a host-side upper bound of n_alive that sizes the (persistent) launches.  Initial state: cnt[0] = N, rest 0. */
int pvd_infer_round_begin(int32_t *state, uint32_t parity, uint32_t N, uint32_t max_steps, pvd_stream_t stream);
int pvd_infer_compact(int32_t *state, uint32_t parity, uint32_t n_upper, int32_t *rays_alive, const int32_t *rays_alive_old,
                      float *rays_t, const float *rays_t_old, pvd_stream_t stream);
int pvd_infer_march(const int32_t *state, uint32_t n_upper, const int32_t *rays_alive, const float *rays_t,
                    const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                    uint32_t H, const uint8_t *grid, const float *fars, float *xyzs, float *dirs, float *deltas,
                    uint32_t perturb, pvd_stream_t stream);
int pvd_infer_composite(const int32_t *state, uint32_t n_upper, const int32_t *rays_alive, float *rays_t,
                        const float *sigmas, const float *rgbs, const float *deltas, float sigma_scale, float *weights_sum,
                        float *depth, float *image, pvd_stream_t stream);

/* ------------------------------------------------------------------------
 * _gridencoder  (gridencoder/src/gridencoder.h:12-13, bindings.cpp:5-8)
 * ---------------------------------------------------------------------- */

/* grid_encode_forward -- gridencoder.cu:419-442 (kernel :75-224).
 * inputs [B,D] f32 in [0,1]; embeddings [offsets[L],C] dtype; offsets [L+1] i32;
 * outputs [L,B,C] dtype; dy_dx [B,L,D,C] dtype when calc_grad_inputs (else ignored).
 * S = log2(per_level_scale), H = base resolution; gridtype 0 = hash, 1 = tiled. */
int pvd_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets,
                            void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                            float S, uint32_t H, int calc_grad_inputs, void *dy_dx,
                            uint32_t gridtype, int align_corners, int dtype, pvd_stream_t stream);

/* Tuning knob for A/B measurements (tools/bench_grid.py): workgroup schedule of the grid kernels, 1 = XCD-aware
 * (big levels dealt out one per XCD, default), 0 = plain level-major.  Returns the previous value.  Results are identical. */
int pvd_grid_set_variant(int variant);

/* A/B knob of the f16 / D = 3 / C = 2 kernels (results are bit-identical forward, within summation order backward):
 * lanes_per_sample 0 = one thread per (sample, level), 2 (default) / 4 = the corners of a sample spread over 2 / 4 adjacent
 * lanes (corners sharing a cache line are fetched by one load instruction); persistent_blocks > 0 (default 4096) = that
 * many workgroups loop over the (level, point block) work items; | 1 << 30 = XCD-affine item order (rejected, see
 * gridencoder.hip); | 1 << 29 = backward through the thread-per-sample run-merging kernel instead of the two-lane one.
 * Returns the previous setting (lanes | blocks << 4) or PVD_ERR_INVALID. */
int pvd_grid_set_fwd_kernel(int lanes_per_sample, int persistent_blocks);

/* grid_encode_backward -- gridencoder.cu:444-474 (kernels :227-343).
 * grad [L,B,C] dtype; grad_embeddings like embeddings (zero-filled); grad_inputs [B,D] dtype
 * when calc_grad_inputs.  `embeddings` is unused by the arithmetic (as in the reference) and may be null. */
/* pvd_grid_encode_forward (no dy_dx) on positions given in [-bound, bound]: every coordinate is mapped with
 * x01 = (x + in_add) / in_div while it is read -- GridEncoder.forward's (inputs + bound) / (2 * bound), grid.py:211, as the
 * same two IEEE operations inside the kernel instead of two elementwise launches. */
int pvd_grid_encode_forward_affine(const float *inputs, float in_add, float in_div, const void *embeddings,
                                   const int32_t *offsets, void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                   float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype,
                                   pvd_stream_t stream);
/* (ABI 6) ... and its backward: pvd_grid_encode_backward (no grad_inputs) reading the positions through the same mapping, so that a
 * training step needs no normalised copy of them at all.  dtype PVD_F16, D = 3, C = 2 (the PVD table's scatter) only: anything else
 * is PVD_ERR_UNSUPPORTED. */
/* (ABI 6) a head's packed f16 weight image riding on a lookup's forward launch (see pvd_vm_forward_pack_rider below) */
typedef struct pvd_head_pack_rider {
    int kind;                                 /* 1: VM head (pvd_vm_forward_pack_rider); 0: hash head (pvd_grid_encode_forward_affine_pack) */
    const float *Wa1, *Wa2, *Wc1, *Wc2, *Wc3; /* DEVICE fp32 masters, as pvd_head_pack_weights takes them (Wa2: hash head only, else NULL) */
    void *image;                              /* DEVICE f16 [pvd_head_image_halfs(kind)] */
} pvd_head_pack_rider;
/* (ABI 6) pvd_grid_encode_forward_affine + the HASH head's packed weight image (pvd_head_pack_weights, kind 0) written by extra workgroups
 * at the end of the same grid (the f16 / D 3 / C 2 lookup; any other variant packs in a launch of its own behind the lookup): the
 * teacher-training / hash-student step then has no pack launch between the lookup and the head's forward.  pack->kind must be 0. */
int pvd_grid_encode_forward_affine_pack(const float *inputs, float in_add, float in_div, const void *embeddings,
                                        const int32_t *offsets, void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                        float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype,
                                        const pvd_head_pack_rider *pack, pvd_stream_t stream);
int pvd_grid_encode_backward_affine(const void *grad, const float *inputs, float in_add, float in_div, const void *embeddings,
                                    const int32_t *offsets, void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                    float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype, pvd_stream_t stream);

int pvd_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings,
                             const int32_t *offsets, void *grad_embeddings,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             int calc_grad_inputs, const void *dy_dx, void *grad_inputs,
                             uint32_t gridtype, int align_corners, int dtype, pvd_stream_t stream);

/* ------------------------------------------------------------------------
 * _shencoder  (shencoder/src/shencoder.h:9-12, bindings.cpp:5-8)
 * ---------------------------------------------------------------------- */

/* sh_encode_forward -- shencoder.cu:402-419 (kernel :27-356).  inputs [B,3] f32, outputs [B,C*C] f32,
 * dy_dx [B,3,C*C] f32 when calc_grad_inputs.  C = degree in [1,8]. */
int pvd_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t C,
                          int calc_grad_inputs, float *dy_dx, pvd_stream_t stream);

/* sh_encode_backward -- shencoder.cu:421-440 (kernel :359-383).  grad_inputs [B,3] += sum grad*dy_dx. */
int pvd_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t C,
                           const float *dy_dx, float *grad_inputs, pvd_stream_t stream);

/* ------------------------------------------------------------------------
 * VM (TensoRF plane x line) feature lookup -- in the reference this is torch code, not a native
 * module: NeRFNetwork.get_sigma_feat / get_color_feat (distill_mutual/network.py:216-309), i.e.
 * 12 F.grid_sample(align_corners=True) calls + products; tables from init_one_vm (:193-214).
 *
 * xyz [M,3] f32 device, un-normalised; aabb_host[6] HOST floats (aabb_train; x_n = 2(x-lo)/(hi-lo)-1,
 * network.py:345-350).  tables_host[12]: HOST array of DEVICE pointers
 *   {sigma_mat[0..2], sigma_vec[0..2], color_mat[0..2], color_vec[0..2]},
 * each factor in CHANNELS-LAST order: plane i is [H_i][W_i][R], line i is [L_i][R] with
 * R = 16 (sigma) / 48 (colour), W_i = res[m0_i], H_i = res[m1_i], L_i = res[vec_id_i],
 * mat_ids = {{0,1},{0,2},{1,2}}, vec_ids = {2,1,0}; res_host[3] HOST uint32.
 * sigma_feat [M] f32; color_prod [M,144] prod_dtype (f32, or f16 when the caller runs under AMP:
 * the products feed basis_mat, an autocast-to-half Linear).
 * texel_stride_host: NULL (densely packed tables), or HOST uint32[4] = elements between consecutive texels of
 *   {sigma planes, sigma lines, colour planes, colour lines} (>= R).  With stride 64 and colour base = sigma base + 16 the
 *   sigma and colour factors of a plane interleave in one [H][W][64] buffer: a tap is then ONE aligned 256-byte access
 *   (two cache lines) instead of 64 B + 192 B (three).
 * ---------------------------------------------------------------------- */
int pvd_vm_forward(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host,
                   const uint32_t *res_host, float *sigma_feat, void *color_prod, int prod_dtype, const int32_t *rows_dev,
                   const uint32_t *texel_stride_host, pvd_stream_t stream);

/* (ABI 6) The VM head's packed weight image riding on the lookup's launch: pvd_vm_forward_pack_rider = pvd_vm_forward + what
 * pvd_head_pack_weights(kind 1, Wa1 = basis_mat, Wc1..3 = color_net, image) writes, in extra workgroups at the end of the same grid.
 * In a training step the head's forward follows the lookup and both read the weights the update has just written: with the pack on
 * the lookup's launch the step's dependent chain has neither a pack launch nor a cross-stream wait for one.  The image
 * (pvd_head_image_halfs(1) halfs) is complete when the launch is; it must not be in use by another stream meanwhile.  M == 0 is
 * PVD_ERR_INVALID (no launch to ride on: call pvd_head_pack_weights). */
int pvd_vm_forward_pack_rider(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host,
                              const uint32_t *res_host, float *sigma_feat, void *color_prod, int prod_dtype, const int32_t *rows_dev,
                              const uint32_t *texel_stride_host, const pvd_head_pack_rider *pack, pvd_stream_t stream);

/* grad_tables_host[12]: HOST array of DEVICE pointers laid out like tables_host, f32, accumulated into
 * with atomics (zero-filled by the caller, or a gradient buffer to accumulate into). */
int pvd_vm_backward(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host,
                    const uint32_t *res_host, const float *grad_sigma_feat, const void *grad_color_prod, int prod_dtype,
                    void *const *grad_tables_host, const uint32_t *texel_stride_host, pvd_stream_t stream);

/* The VM head's weight-gradient reduction riding on the VM table scatter's launch (both depend only on the head's backward):
 * pvd_head_backward_defer = pvd_head_backward (kind = PVD_HEAD_VM only) WITHOUT the launch that sums the per-workgroup
 * accumulator tiles into gW*; it fills *rider_out instead, and pvd_vm_backward_rider = pvd_vm_backward + that reduction in
 * extra workgroups of the scatter's launch.  The caller owes exactly one pvd_vm_backward_rider per deferral, on the same
 * stream, before anything reads gW* (the gradients are incomplete until then); workspace must stay alive until it ran. */
typedef struct pvd_head_dw_rider {
    const float *partials;  /* DEVICE: the head backward's workspace */
    uint32_t nblocks;
    float *gWa1, *gWc1, *gWc2, *gWc3;
    /* (ABI 4; pvd_head_backward_defer sets it to NULL, the caller may fill it in) found_inf != NULL (DEVICE float scalar, never cleared
     * here): pvd_vm_backward_rider's launch stores 1 into it when an incoming gradient value it reads (grad_sigma_feat,
     * grad_color_prod) or a sum it adds into gW* is inf / nan -- GradScaler's inf check of everything this launch completes (the
     * twelve table gradients are sums of those values times finite interpolation weights and table entries; the four weight
     * gradients), folded into the launch, so that the step's chain has no separate check between the scatter and the update.
     * LIMIT of the folded check (it looks at the launch's INPUTS and the weight-gradient sums, not at the table gradients it writes): it
     * is GradScaler's check exactly when finite inputs imply finite table gradients -- finite table entries, and no fp32 overflow of a
     * product or of the accumulated sums (under AMP the inputs are loss-scaled f16-range values times weights in [0, 1] and table
     * entries: orders of magnitude below 3e38).  A caller that cannot promise finite tables runs pvd_check_finite / pvd_segments_op(3) on
     * the gradients instead (the harness: PVD_INF_CHECK_RIDE=0; under ray-DP the exchange's gather looks at the written values). */
    float *found_inf;
} pvd_head_dw_rider;
int pvd_vm_backward_rider(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host,
                          const uint32_t *res_host, const float *grad_sigma_feat, const void *grad_color_prod, int prod_dtype,
                          void *const *grad_tables_host, const uint32_t *texel_stride_host, const pvd_head_dw_rider *rider,
                          pvd_stream_t stream);
int pvd_head_backward_defer(int kind, const void *x0, const float *sigma_raw, const float *dirs, uint32_t M, const float *Wa1,
                            const float *Wa2, const float *Wc1, const float *Wc2, const float *Wc3, const void *image,
                            float clip_sigma_min, float clip_feat_min, float clip_max, const float *g_sigma, const float *g_rgb,
                            const float *g_rgb2, const float *g_feat16, float *g_sigma_raw, void *g_x0, float *gWa1, float *gWa2,
                            float *gWc1, float *gWc2, float *gWc3, float *workspace, pvd_head_dw_rider *rider_out,
                            pvd_stream_t stream);

/* NeRF positional encoding -- torch code in the reference: FreqEncoder.forward, tools/encoding.py:6-49.
 * x [M,D] f32 -> out [M, row_stride] (out_dtype PVD_F32 / PVD_F16): columns [x (if include_input), sin(f_0 x), cos(f_0 x),
 * sin(f_1 x), ...], each block D wide, f = freq_bands_host[n_freqs] (<= 16; the reference's 2^linspace(0, max_freq_log2, N));
 * columns from D (1 + 2 n_freqs) up to row_stride are written as zero (padding to a GEMM-friendly row length). */
int pvd_freq_encode(const float *x, uint32_t M, uint32_t D, const float *freq_bands_host, uint32_t n_freqs, int include_input,
                    void *out, int out_dtype, uint32_t row_stride, pvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused sigma / colour head (MFMA).  Torch code in the reference: NeRFNetwork.forward,
 * distill_mutual/network.py:335-437 (sigma_net :103-118, color_net :135-152, basis_mat :88-90, trunc_exp,
 * SH degree 4) evaluated under autocast(fp16).  kind 0 = hash model, 1 = VM model.
 *   x0: kind 0: grid-encoder output [14][M][2] f16 (the level-major layout pvd_grid_encode_forward writes);
 *       kind 1: plane x line products [M][144] f16 (pvd_vm_forward).
 *   sigma_raw [M] f32 (kind 1 only), dirs [M][3] f32.
 *   Wa1/Wa2: kind 0: sigma_net.0 [64][28], sigma_net.1 [16][64]; kind 1: basis_mat [15][144], NULL.
 *   Wc1 [64][31], Wc2 [64][64], Wc3 [3][64]: color_net.  All weights fp32 masters, row-major [out][in];
 *   they are rounded to f16 inside the kernel exactly like autocast's cast.
 *   clip_*: args.sigma_clip_min / sigma_clip_max (network.py:353-362, 418-420).
 * Outputs: sigma [M] = exp(clamped log-sigma), rgb [M][3], feat16 [M][16] = feature_sigma_color (all f32).
 * ---------------------------------------------------------------------- */
int pvd_head_forward(int kind, const void *x0, const float *sigma_raw, const float *dirs, uint32_t M,
                     const float *Wa1, const float *Wa2, const float *Wc1, const float *Wc2, const float *Wc3,
                     const void *image, float clip_sigma_min, float clip_feat_min, float clip_max,
                     float *sigma, float *rgb, float *feat16, const int32_t *rows_dev, pvd_stream_t stream);

/* The frozen hash model in one launch: pvd_grid_encode_forward_affine (f16 table, D = 3, C = 2, L = 14) + pvd_head_forward
 * (kind = PVD_HEAD_HASH) without the [14][M][2] intermediate -- the teacher of a distillation run, inference, occupancy-grid
 * density queries (network.py:413-437 under no_grad).  xyz [M,3] in [-bound, bound] mapped with (x + in_add) / in_div;
 * embeddings_f16 [offsets[14], 2] f16; the other arguments as in the two calls it replaces.  Outputs are bit-identical to
 * theirs.  rows_dev (here, in pvd_head_forward and in pvd_vm_forward): NULL, or a DEVICE int32 -- only the first
 * min(M, *rows_dev) rows are computed (M is then just the upper bound that sizes the launch): the inference rounds keep
 * their row count on the device (pvd_infer_*). */
int pvd_hash_head_forward_fused(const float *xyz, float in_add, float in_div, const void *embeddings_f16,
                                const int32_t *offsets, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                const float *dirs, uint32_t M, const float *Wa1, const float *Wa2, const float *Wc1,
                                const float *Wc2, const float *Wc3, const void *image, float clip_sigma_min,
                                float clip_max, float *sigma, float *rgb, float *feat16, const int32_t *rows_dev,
                                pvd_stream_t stream);

/* The same launch, leaving its own extent behind: span (DEVICE uint64 [2], initialised {~0, 0} by the caller; NULL = plain
 * pvd_hash_head_forward_fused) receives span[0] = min over the workgroups of their start and span[1] = max of their end in ticks of
 * the device's constant 100 MHz counter (s_memrealtime) -- (span[1] - span[0]) * 10 ns is how long the launch lasted WHERE IT
 * RAN: recorded into a hipGraph next to other kernels, where no host event can bracket it (bench.py's `roofline.in_step`).  Only
 * the G = 7 / 14 kernels of the default build write it (PVD_FUSED_VARIANT=0 leaves it untouched). */
int pvd_hash_head_forward_fused_span(const float *xyz, float in_add, float in_div, const void *embeddings_f16,
                                     const int32_t *offsets, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                     const float *dirs, uint32_t M, const float *Wa1, const float *Wa2, const float *Wc1,
                                     const float *Wc2, const float *Wc3, const void *image, float clip_sigma_min,
                                     float clip_max, float *sigma, float *rgb, float *feat16, const int32_t *rows_dev,
                                     uint64_t *span, pvd_stream_t stream);

/* A whole inference render of a frozen hash model in ONE persistent launch (+ a small compaction launch): what run_cuda's eval
 * branch does in rounds -- march_rays -> model -> composite_rays -> compact_rays until every ray has ended
 * (distill_mutual/renderer.py:450-543, raymarching.cu:704-948) -- with the rays' state in registers, the samples of a round in
 * LDS and the alive queue in device memory; a thread-per-ray launch first walks every ray to its first
 * occupied cell.  rays_o / rays_d [N,3], nears / fars [N] (pvd_near_far_from_aabb),
 * bitfield / bound / dt_gamma / max_steps / C / H as for pvd_march_rays (perturb = 0); sigma_scale = density_scale
 * (renderer.py:528); the model as for pvd_hash_head_forward_fused (H0 = the encoder's base resolution).  workspace: 2 N + 12 int32 ([2 N + 2 .. 2 N + 6): local rounds, rows shaded, walk-only rounds, workgroups used).
 * weights_sum [N], depth [N], image [N,3]: ZERO-FILLED by the caller, written for every ray that enters the box -- the values the
 * round loop leaves there (background compositing and depth normalisation stay with the caller, renderer.py:545-548).  A ray's
 * result does not depend on how rays are grouped into rounds, so the image is the round loop's; the reference stops ALL rays
 * once the rounds' steps add up to max_steps, here a ray stops after its own max_steps steps. */
int pvd_infer_image_hash(const float *rays_o, const float *rays_d, const float *nears, const float *fars, uint32_t N,
                         const uint8_t *bitfield, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                         float sigma_scale, float in_add, float in_div, const void *embeddings_f16, const int32_t *offsets, float S,
                         uint32_t H0, uint32_t gridtype, int align_corners, const float *Wa1, const float *Wa2, const float *Wc1,
                         const float *Wc2, const float *Wc3, const void *image, float clip_sigma_min, float clip_max,
                         int32_t *workspace, float *weights_sum, float *depth, float *image_out, pvd_stream_t stream);

/* The same persistent render for a frozen VM (TensoRF plane x line) model -- run_cuda's eval branch (renderer.py:450-543) over
 * NeRFNetwork.forward's vm branch (network.py:344-381): rays / bitfield / workspace / outputs as for pvd_infer_image_hash; the tables
 * (aabb_host, tables_host[12], res_host[3], texel_stride_host) as for pvd_vm_forward; Wb = basis_mat [15][144], Wc1..3 = color_net;
 * image: NULL or the PVD_HEAD_VM weight image of pvd_head_pack_weights; clip_sigma_min bounds the sigma feature (-100 when
 * enable_edit_plenoxel, network.py:355-360), clip_feat_min the colour features.  Per ray the arithmetic of pvd_vm_forward (f16
 * products) + pvd_head_forward(PVD_HEAD_VM) + pvd_composite_rays: the image is the round loop's, bit for bit. */
int pvd_infer_image_vm(const float *rays_o, const float *rays_d, const float *nears, const float *fars, uint32_t N,
                       const uint8_t *bitfield, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                       float sigma_scale, const float *aabb_host, const void *const *tables_host, const uint32_t *res_host,
                       const uint32_t *texel_stride_host, const float *Wb, const float *Wc1, const float *Wc2, const float *Wc3,
                       const void *image, float clip_sigma_min, float clip_feat_min, float clip_max, int32_t *workspace,
                       float *weights_sum, float *depth, float *image_out, pvd_stream_t stream);

/* ... and for a frozen Plenoxel ("tensors") model -- run_cuda's eval branch over NeRFNetwork.forward's tensors branch
 * (network.py:383-409): rays / bitfield / workspace / outputs as for pvd_infer_image_hash (cascade, grid_size: the occupancy grid's);
 * volume / dims_host / C / degree / clip_min / clip_max / aabb_host as for pvd_plenoxel_forward.  Per row the values of
 * pvd_plenoxel_forward, per ray pvd_composite_rays' sums: the image is the round loop's, bit for bit. */
int pvd_infer_image_plenoxel(const float *rays_o, const float *rays_d, const float *nears, const float *fars, uint32_t N,
                             const uint8_t *bitfield, float bound, float dt_gamma, uint32_t max_steps, uint32_t cascade,
                             uint32_t grid_size, float sigma_scale, const float *aabb_host, const float *volume,
                             const uint32_t *dims_host, uint32_t C, uint32_t degree, float clip_min, float clip_max,
                             int32_t *workspace, float *weights_sum, float *depth, float *image_out, pvd_stream_t stream);

/* The frozen `mlp` model (NeRF trunk + sigma / colour head; NeRFNetwork.forward with model_type "mlp", network.py:154-182 and
 * :413-437, under no_grad + fp16 autocast) in one launch.
 *   pts_f16 [M][64] f16: positional encoding padded to 64 columns (pvd_freq_encode(out_dtype = PVD_F16, row_stride = 64));
 *   trunk: Linear(63,256) ReLU, n_before x [Linear(256,256) ReLU], skip layer Linear(63+256, 256) ReLU on [pts | x],
 *          n_after x [Linear(256,256) ReLU], Linear(256,28); all with bias -- the reference's nerf_mlp with
 *          nerf_layer_num = n_before + n_after + 3, skip = n_before, nerf_layer_wide = 256, PE = 10;
 *   wstream_f16: the weights as the kernel streams them through LDS -- layer after layer, each in chunks of 64 output rows
 *          (the last layer: one chunk of 32 rows, rows 28..31 zero): rows x (K + 8) halfs row-major (K = 64 / 256 / 320, input
 *          columns zero-padded 63 -> 64, then permuted inside every group of 32: logical column 32 p + 16 s + 4 h + j is stored
 *          at 32 p + 8 h + 4 s + j; 8 halfs of padding per row), then `rows` bias halfs (fusedhead.mlp_weight_stream builds it);
 *   the head arguments and outputs as in pvd_head_forward(kind = PVD_HEAD_HASH). */
int pvd_mlp_head_forward_fused(const void *pts_f16, uint32_t M, const void *wstream_f16, uint32_t n_before, uint32_t n_after,
                               const float *dirs, const float *Wa1, const float *Wa2, const float *Wc1, const float *Wc2,
                               const float *Wc3, const void *image, float clip_sigma_min, float clip_max, float *sigma,
                               float *rgb, float *feat16, pvd_stream_t stream);

/* Optional weight image.  Every workgroup of the head kernels stages all weights in LDS; converting and
 * (for the backward) transposing the fp32 masters there is the kernels' fixed cost.  pvd_head_pack_weights does it
 * once into `image` (pvd_head_image_halfs(kind) f16 elements, device memory), and pvd_head_forward /
 * pvd_head_backward given image != NULL copy it with 16-byte loads instead.  The image must be re-packed whenever
 * the fp32 weights change (once per optimizer step for a student, once ever for a frozen teacher). */
int pvd_head_image_halfs(int kind);
int pvd_head_pack_weights(int kind, const float *Wa1, const float *Wa2, const float *Wc1, const float *Wc2,
                          const float *Wc3, void *image, pvd_stream_t stream);

/* Backward of the fused head (training).  Recomputes the forward from (x0, sigma_raw, dirs), then writes the
 * gradient of the head's input and ACCUMULATES (+=) the f32 weight gradients.
 *   kind = PVD_HEAD_VM  : g_sigma_raw [M] f32, g_x0 = d/d products [M][144] f16 (the layout pvd_vm_backward
 *                         reads); gWa1 = basis_mat [15][144]; Wa2/gWa2 unused (NULL).
 *   kind = PVD_HEAD_HASH: g_x0 = d/d encoder output, level-major [14][M][2] f16 (what pvd_grid_encode_backward
 *                         reads as `grad`); gWa1 = sigma_net.0 [64][28], gWa2 = sigma_net.1 [16][64];
 *                         sigma_raw / g_sigma_raw unused (NULL).
 *   gWc1 [64][31], gWc2 [64][64], gWc3 [3][64]: color_net.
 * g_sigma [M], g_rgb [M][3], g_feat16 [M][16]: incoming gradients (f32) of pvd_head_forward's three outputs; g_feat16 may be NULL
 *   (ABI 6: no gradient reaches feature_sigma_color -- a model trained on pixels alone -- read as zeros).
 * g_rgb2 [M][3] or NULL: a second gradient of the rgb output (it feeds both the compositing and the colour term of the
 *   distillation objective, utils.py:1158-1176), added while loading instead of by a separate elementwise launch.
 * workspace: pvd_head_backward_workspace_floats(kind, M) floats of scratch (per-wave dW partials).
 * Replaces autograd through network.py:395-447 (hash) / 353-393 (vm). */
int pvd_head_backward_workspace_floats(int kind, uint32_t M);
int pvd_head_backward(int kind, const void *x0, const float *sigma_raw, const float *dirs, uint32_t M,
                      const float *Wa1, const float *Wa2, const float *Wc1, const float *Wc2, const float *Wc3,
                      const void *image, float clip_sigma_min, float clip_feat_min, float clip_max,
                      const float *g_sigma, const float *g_rgb, const float *g_rgb2, const float *g_feat16,
                      float *g_sigma_raw, void *g_x0, float *gWa1, float *gWa2, float *gWc1, float *gWc2, float *gWc3,
                      float *workspace, pvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Occupancy-grid maintenance on the device: the torch code of NeRFRenderer.update_extra_state
 * (distill_mutual/renderer.py:647-775) without its host round trips.  Per cascade c (grid slice [H^3], Morton order):
 *   pvd_occ_sample : which cells to query and where.  full != 0: every cell (slot i = Morton index i; the first 16
 *                    updates).  Otherwise n_uniform uniformly random cells followed by n_occupied cells drawn with
 *                    replacement from the cells with density > 0 (indices -1 if there are none).  Positions = cell
 *                    centre +- half a cell of jitter in the cascade's box [-bound_c, bound_c], bound_c = min(2^c, bound).
 *                    occ_list [H^3] int32 and occ_count [1] are scratch; PCG32 keyed by (seed, slot).
 *   (the caller evaluates the density at xyz)
 *   pvd_occ_update : tmp = -1; tmp[indices] = sigmas * sigma_scale; grid = max(grid * decay, tmp) where both >= 0.
 *                    tmp [H^3] is scratch.
 * After all cascades:
 *   pvd_occ_finish : mean_thresh[0] = mean(clamp(grid, 0)), mean_thresh[1] = min(mean, density_thresh) (DEVICE floats),
 *                    bitfield = packbits(grid > mean_thresh[1]) over all n_cells = C * H^3 cells; scratch: 1024 floats.
 * ---------------------------------------------------------------------- */
int pvd_occ_sample(const float *density_grid, uint32_t H, uint32_t n_uniform, uint32_t n_occupied, int full, float bound_c,
                   uint64_t seed, int32_t *occ_list, uint32_t *occ_count, int32_t *indices, float *xyz,
                   pvd_stream_t stream);
int pvd_occ_update(float *density_grid, float *tmp, const int32_t *indices, const float *sigmas, uint32_t n, uint32_t H,
                   float sigma_scale, float decay, pvd_stream_t stream);
/* Replay of a run of the reference (its random draws are torch's; pvd_occ_sample draws its own):
 *   pvd_occ_sample_replay  : pvd_occ_sample's positions from supplied draws -- cells [n_uniform][3] int32 = randint(0, H, (n, 3));
 *                            occ_list = the occupied cells' Morton indices ASCENDING (nonzero()), picks [n_occupied] int32 =
 *                            randint(0, #occupied, n) into it; jitter [n][3] f32 in [0, 1) = rand(n, 3), row k for the k-th point
 *                            the reference queries (full sweep, H <= 128: the meshgrid point (x H + y) H + z; renderer.py:700-741).
 *   pvd_occ_update_ordered : pvd_occ_update with duplicate cells resolved as a sequential assignment resolves them (the last
 *                            position wins; plain pvd_occ_update: whichever write lands last); owner [H^3] int32 is scratch. */
int pvd_occ_sample_replay(uint32_t H, uint32_t n_uniform, uint32_t n_occupied, int full, float bound_c, const int32_t *cells,
                          const int32_t *occ_list, const int32_t *picks, const float *jitter, int32_t *indices, float *xyz,
                          pvd_stream_t stream);
int pvd_occ_update_ordered(float *density_grid, float *tmp, int32_t *owner, const int32_t *indices, const float *sigmas, uint32_t n,
                           uint32_t H, float sigma_scale, float decay, pvd_stream_t stream);
int pvd_occ_finish(const float *density_grid, uint32_t n_cells, float density_thresh, float *mean_thresh, float *scratch,
                   uint8_t *bitfield, pvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Plenoxel ("tensors") model: dense-volume lookup + SH colour head, forward and backward.
 * Replaces compute_plenoxel_fea (3-D F.grid_sample, trilinear, align_corners=True, zero padding;
 * distill_mutual/network.py:311-322) and the head around it (network.py:383-409):
 *   h = sample(volume, 2*(x-lo)/(hi-lo)-1);  sigma_l = clamp(h[0], clip_min, clip_max);  sigma = exp(sigma_l);
 *   rgb[c] = sigmoid(sum_k h[1 + c*deg^2 + k] * SH_k(d)).
 * volume: the [1,C,D,H,W] parameter stored CHANNELS-LAST ([D][H][W][C] f32), C = 3*degree^2 + 1, degree 1..3.
 * dims_host = {D, H, W}; aabb_host = {lo[3], hi[3]} (host memory).
 * forward: feat [M][C] raw features (optional, NULL to skip); if dirs != NULL also h0_raw [M], sigma_l [M],
 *   sigma [M], rgb [M][3].  dirs == NULL: feature lookup only.
 * backward: ACCUMULATES (+=) into grad_volume (same layout).  g_feat [M][C], g_sigma [M], g_sigma_l [M],
 *   g_rgb [M][3] are the gradients of the forward's outputs; any of them may be NULL (= zero).  With dirs the
 *   saved h0_raw and rgb of the forward are required.
 * ---------------------------------------------------------------------- */
int pvd_plenoxel_forward(const float *xyz, const float *dirs, uint32_t M, const float *aabb_host, const float *volume,
                         const uint32_t *dims_host, uint32_t C, uint32_t degree, float clip_min, float clip_max,
                         float *feat, float *h0_raw, float *sigma_l, float *sigma, float *rgb, pvd_stream_t stream);
int pvd_plenoxel_backward(const float *xyz, const float *dirs, uint32_t M, const float *aabb_host, const uint32_t *dims_host,
                          uint32_t C, uint32_t degree, float clip_min, float clip_max, const float *h0_raw, const float *rgb,
                          const float *g_feat, const float *g_sigma, const float *g_sigma_l, const float *g_rgb,
                          float *grad_volume, pvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused pieces of run_cuda / train_step that are torch code in the reference.
 * ---------------------------------------------------------------------- */

/* composite_rays_train + the epilogue of run_cuda (distill_mutual/renderer.py:442-446):
 *   image += (1 - weights_sum) * bg ; depth = clamp(depth - near, 0) / (far - near + depth_eps).
 * bg [N,3] f32 per-ray background or NULL (then bg_scalar).  Outputs are the blended image / normalised depth. */
int pvd_composite_rays_train_bg_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays,
                                        uint32_t M, uint32_t N, const float *bg, float bg_scalar, const float *nears,
                                        const float *fars, float depth_eps, float *weights_sum, float *depth, float *image,
                                        const int32_t *budget_dev, pvd_stream_t stream);
/* grad_image is w.r.t. the BLENDED image; `image` is the blended image the forward returned; grad_weights_sum may be NULL.
 * flags & PVD_MARCH_FRESH: grad_sigmas / grad_rgbs arrive uninitialised (the reference zero-fills them,
 *   raymarching.py:339-340) and `rays` is a table written by pvd_march_rays_train (offsets = exclusive prefix sum of the
 *   counts in row order, from 0): every slot no ray owns is written as zero by the kernel. */
int pvd_composite_rays_train_bg_backward(const float *grad_weights_sum, const float *grad_image, const float *sigmas,
                                         const float *rgbs, const float *deltas, const int32_t *rays, const float *weights_sum,
                                         const float *image, uint32_t M, uint32_t N, const float *bg, float bg_scalar,
                                         float *grad_sigmas, float *grad_rgbs, uint32_t flags, const int32_t *budget_dev,
                                         pvd_stream_t stream);

/* The student's compositing of a stage-3 DISTILLATION step with the objective riding on its two launches (no counterpart in the
 * reference's modules: Trainer.train_step's four normL2 terms, distill_mutual/utils.py:1109-1176, are torch code there).
 * forward  = pvd_composite_rays_train_bg_forward + pvd_distill_sumsq: besides weights_sum / depth / image the launch leaves
 *            the per-workgroup partial sums of the four squared norms in S4 + 4 (pvd_composite_objective_blocks(N, rows) float4
 *            entries; S4 holds 4 + 4 * that many floats) for pvd_distill_loss_final(..., reduce = that block count, ...);
 * backward = pvd_distill_sumsq_backward + pvd_composite_rays_train_bg_backward: the image gradient coef4[0] * upstream *
 *            (image - img_tea) is formed per ray inside the compositing backward, g_fea / g_col are written by extra
 *            workgroups of the same launch.
 *            rates4 != NULL (DEVICE [4]): the backward launch also FINISHES the objective -- every workgroup reduces the
 *            forward launch's partial sums (at S4 + 4) for itself, coef4 becomes an output, workgroup 0 publishes S4[0..3],
 *            loss, norms4 (+ the parameter-only partial sums `extra`) -- so no pvd_distill_loss_final runs between the
 *            passes; the forward launch then applies the feature rate's decay (rates4_decay[1] *= fea_decay, one thread).
 * Ray data parallelism (the four sums of squares are sums over the RANKS' rows): both launches with flags &
 *            PVD_OBJECTIVE_FIXED_PARTS -- the number of partial sums then depends on N alone
 *            (pvd_composite_objective_blocks_fixed(N); workgroups beyond the feature rows write zeros), so every rank leaves
 *            a buffer of the same size and the host all-reduces (SUM) the PARTIALS S4[4 ..) between the two launches: one
 *            collective and nothing else between them; the backward launch reduces the summed partials like its own.
 * img_tea [N,3] by ray index, fea_* [rows,16] (column 0 = sigma_l), col_* [rows,3], all f32; rows > 0. */
#define PVD_OBJECTIVE_FIXED_PARTS 2u
uint32_t pvd_composite_objective_blocks(uint32_t N, uint32_t rows);
uint32_t pvd_composite_objective_blocks_fixed(uint32_t N);
int pvd_composite_objective_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, uint32_t M,
                                    uint32_t N, const float *bg, float bg_scalar, const float *nears, const float *fars,
                                    float depth_eps, float *weights_sum, float *depth, float *image, const int32_t *budget_dev,
                                    const float *img_tea, const float *fea_stu, const float *fea_tea, const float *col_stu,
                                    const float *col_tea, uint32_t rows, float *S4, float *rates4_decay, float fea_decay,
                                    uint32_t flags, pvd_stream_t stream);
int pvd_composite_objective_backward(const float *grad_weights_sum, const float *sigmas, const float *rgbs, const float *deltas,
                                     const int32_t *rays, const float *weights_sum, const float *image, uint32_t M, uint32_t N,
                                     const float *bg, float bg_scalar, float *grad_sigmas, float *grad_rgbs, uint32_t flags,
                                     const int32_t *budget_dev, const float *img_tea, const float *fea_stu, const float *fea_tea,
                                     const float *col_stu, const float *col_tea, uint32_t rows, float *coef4,
                                     const float *upstream, float *g_fea, float *g_col, const float *rates4, const float *extra,
                                     uint32_t n_extra, float *S4, float *loss, float *norms4, pvd_stream_t stream);

/* (ABI 6) The teacher's objective -- torch code in the reference: MSELoss(reduction='none'), .mean(-1), .mean(),
 * just_train_tea/utils.py:573-581 -- and its gradient in one launch: loss[0] = sum (pred - target)^2 / n, dloss_dpred[i] =
 * 2 (pred[i] - target[i]) / n (the caller multiplies by the upstream gradient, e.g. the loss scale).  n = rays x 3. */
int pvd_mse_forward(const float *pred, const float *target, uint32_t n, float *loss, float *dloss_dpred, pvd_stream_t stream);

/* Stage-3 distillation objective with loss_type = normL2 (distill_mutual/utils.py:941-952, 1109-1189):
 *   S4 = { |I_tea - I_stu|^2, |F_stu - F_tea|^2, |F_stu[:,0] - F_tea[:,0]|^2, |c_stu - c_tea|^2 }  (sums over all rows)
 *   loss = sum_i rates4[i] * sqrt(S4[i]) + sum(extra);  coef4[i] = rates4[i] / sqrt(S4[i])  (0 if S4[i] == 0)
 * img [n_img] f32 (= N*3), fea [M,fea_width] f32 (16-byte aligned) with column 0 = the log-density feature, col [M,3] f32.
 * fea_width must be 16 (1 + geo_feat_dim of the reference's default models, network.py:30: the kernels read rows as four
 * float4) or 1 (a model without a feature vector, e.g. the Plenoxel student: fea = sigma_l alone, S4[1] = 0, no feature
 * term); anything else returns PVD_ERR_INVALID.  rates4 / upstream are DEVICE scalars.
 * S4 must hold 4 + 4*1024 floats: the four sums, followed by scratch for per-workgroup partials.
 * pvd_distill_sumsq: reduce != 0 finishes S4[0..3] itself (ray data parallelism: the host all-reduces them before
 *   pvd_distill_loss_final(reduce = 0)); reduce == 0 leaves the partials for pvd_distill_loss_final(reduce = 1, same
 *   n_img / M), saving a launch.
 * pvd_distill_loss_final: reduce 0 = S4[0..3] are final, 1 = reduce pvd_distill_sumsq's partials, >= 2 = reduce that many
 *   float4 partials at S4 + 4 as they are (pvd_composite_objective_forward); fea_decay multiplies rates4[1] in place before it is used (the per-step 0.995 decay of the
 *   feature rate, utils.py:1044; 1.0 = leave it); extra [n_extra] are partial sums of a parameter-only term added
 *   to the loss value (pvd_l1_ranges partials), or NULL.
 * pvd_distill_loss_backward = pvd_distill_loss_final + pvd_distill_sumsq_backward in ONE launch (every workgroup finishes
 *   the sums for itself; workgroup 0 publishes loss / coef4 / norms4 / S4[0..3]) for callers that do not need the loss
 *   value before the backward pass.  It only READS rates4: the decay of the feature rate is then applied by
 *   pvd_distill_sumsq (rates4_decay != NULL: rates4_decay[1] *= fea_decay; NULL: leave the rates alone). */
int pvd_distill_sumsq(const float *img_stu, const float *img_tea, uint32_t n_img, const float *fea_stu,
                      const float *fea_tea, uint32_t M, uint32_t fea_width, const float *col_stu, const float *col_tea,
                      float *S4, int reduce, float *rates4_decay, float fea_decay, pvd_stream_t stream);
int pvd_distill_loss_final(float *S4, uint32_t n_img, uint32_t M, int reduce, float *rates4, float fea_decay,
                           const float *extra, uint32_t n_extra, float *loss, float *coef4, float *norms4,
                           pvd_stream_t stream);
int pvd_distill_loss_backward(const float *img_stu, const float *img_tea, uint32_t n_img, const float *fea_stu,
                              const float *fea_tea, uint32_t M, uint32_t fea_width, const float *col_stu, const float *col_tea,
                              float *S4, int reduce, const float *rates4, const float *extra, uint32_t n_extra,
                              const float *upstream, float *loss, float *coef4, float *norms4, float *g_img, float *g_fea,
                              float *g_col, pvd_stream_t stream);
int pvd_distill_sumsq_backward(const float *img_stu, const float *img_tea, uint32_t n_img, const float *fea_stu,
                               const float *fea_tea, uint32_t M, uint32_t fea_width, const float *col_stu, const float *col_tea,
                               const float *coef4, const float *upstream, float *g_img, float *g_fea, float *g_col,
                               pvd_stream_t stream);

/* AdamW over flat fp32 buffers (the optimiser is torch.optim.AdamW in the reference, main_distill_mutual.py:334-339;
 * arithmetic of torch's fused kernel, ADAMW mode).  p, g, m, v: [n] device, n and every segment end multiples of 4.
 * segment_ends_host[n_segments]: HOST array; segment k = [end[k-1], end[k]) uses the DEVICE learning rate lr[k].
 * step: DEVICE float step count (incremented unless found_inf); grad_scale / found_inf: DEVICE scalars of a
 * GradScaler or NULL. */
int pvd_adamw_step(float *p, const float *g, float *m, float *v, uint64_t n, const uint64_t *segment_ends_host,
                   uint32_t n_segments, const float *lr, double beta1, double beta2, double eps, double weight_decay,
                   float *step, const float *grad_scale, const float *found_inf, pvd_stream_t stream);

/* pvd_adamw_step with two pieces of the training loop folded in (extras_host may be NULL = plain step):
 *  - the learning-rate schedule, evaluated on the device into lr[] before the update (so a captured HIP graph
 *    sees it and no host-side scheduler ops run per step):
 *      sched_kind 1: CosineAnnealingLR closed form, lr = eta_min + (base - eta_min)(1 + cos(pi t / T))/2
 *                    (main_distill_mutual.py:346-348; sched_T = T_max, sched_param = eta_min)
 *      sched_kind 2: LambdaLR(factor ** min(t / T, 1)) (main_just_train_tea.py:293-296; sched_param = factor)
 *    t = sched_step[0] (DEVICE scalar, advanced by one per call, also on skipped steps, like scheduler.step()).
 *  - an L1 regulariser on parameter ranges (NeRFNetwork.density_loss, network.py:549-557): inside range r the
 *    unscaled gradient gets l1_coef[r] * sign(p), i.e. the gradient of l1_coef[r] * sum|p|.
 *  - the GradScaler's scale update (see amp_* below). */
typedef struct pvd_adamw_extras {
    int32_t sched_kind;
    float sched_T, sched_param;
    const float *base_lr;  /* DEVICE [n_segments] */
    float *sched_step;     /* DEVICE scalar */
    uint32_t n_l1;         /* <= 16 ranges, begin/end multiples of 4 elements */
    const uint64_t *l1_begin_host, *l1_end_host;
    const float *l1_coef_host;
    /* GradScaler.update() folded in (amp_scale != NULL): after the update, scale / growth tracker are advanced the way
     * torch's amp_update_scale does (x backoff on inf; x growth after amp_interval clean steps) and found_inf is cleared. */
    float *amp_scale;             /* DEVICE scalar (== grad_scale) */
    int32_t *amp_growth_tracker;  /* DEVICE scalar */
    double amp_growth, amp_backoff;
    int32_t amp_interval;
    /* a second gradient in half precision for elements [g16_begin, g16_end) (multiples of 4; g16 8-byte aligned), added to g
     * while it is read: the hash table's f16 scatter-add result, which the reference widens and adds in a separate pass
     * (grid.py:105-136).  NULL = none. */
    const void *g16;
    uint64_t g16_begin, g16_end;
    /* l1_next != NULL (DEVICE, >= 4096 floats): one partial sum per workgroup of l1_next_scale * sum_r l1_coef[r] * |p| over
     * the UPDATED parameters, i.e. the value of the L1 term at the next step's forward (entries beyond the launch's
     * workgroups are left alone: zero them once).  Not written when the step is skipped. */
    float *l1_next;
    float l1_next_scale;
    /* cold_bits != NULL (DEVICE, ceil(n / 128) words): bit i set = parameters [4i, 4i + 4) are "cold" -- the caller
     * guarantees their gradient and both moments are zero and stay zero (table rows the occupancy grid lets no sample
     * reach; outside every L1 and g16 range).  Their update is the weight decay alone, bit-identical to what the full
     * expression gives for g = m = v = 0, and g / m / v are neither read nor written for them (8 instead of 28 B/parameter). */
    const uint32_t *cold_bits;
    /* lazy_log != NULL (with cold_bits): the weight decay of the cold groups is DEFERRED -- the update skips them entirely
     * and logs the learning rates of every applied step in lazy_log[lazy_count++][n_segments] (DEVICE, lazy_capacity steps;
     * lazy_count DEVICE scalar); pvd_adamw_lazy_flush replays the logged decays in order, in the same arithmetic, leaving
     * the bits the per-step decay would have left.  Nothing may read a cold parameter before the flush. */
    float *lazy_log;
    uint32_t *lazy_count;
    uint32_t lazy_capacity;
    /* warm_groups != NULL (only with lazy_log): DEVICE uint32 [n_warm_groups], ascending -- exactly the groups whose cold bit is
     * clear.  The update walks this list instead of every group (the cold ones have nothing to do). */
    const uint32_t *warm_groups;
    uint32_t n_warm_groups;
    /* warm_zero_grad_from (0 = none; only with warm_groups): entries [warm_zero_grad_from, n_warm_groups) of the list are groups whose
     * GRADIENT is structurally zero (warm through an L1 range or through moments that are still decaying, outside everything a
     * backward pass can write): their g is neither read nor zeroed -- the update uses 0, the very value the buffer holds.  The list
     * is then two ascending runs: first the groups that may hold a gradient, then these. */
    uint32_t warm_zero_grad_from;
    /* Two-part update.  The groups of a step fall into (B) those whose gradient the step's backward may have written (table
     * rows a sample can reach, the MLP heads) and (A) those whose gradient is structurally zero (L1-only rows of the sigma
     * planes, rows whose moments are still decaying): nothing reads an A parameter before the next step's objective adds up
     * the L1 term, so part A can run LATER, next to latency-bound kernels, instead of behind the table scatter on the step's
     * critical chain.  The B launch (warm_groups = B, with tail) records the scalars the step used in
     * snapshot = {found_inf, step count before the step, grad scale, 0, lr[0..n_segments)} (DEVICE, 4 + n_segments floats);
     * the A launch (warm_groups = A) passes that record as `replay`: it uses the recorded scalars instead of the live ones
     * (no schedule evaluation, no tail kernel, nothing logged), leaving bit for bit what a single launch would have. */
    float *snapshot;
    const float *replay;
    /* Fewer launches around the update (both off = the three-launch form: zero_grad ... update, tail):
     *  zero_grad_after != 0: every gradient group the update visits is ZEROED once it has been read (also on a skipped step), so
     *    the next step needs no zero_grad launch -- the caller guarantees that whatever may be non-zero is visited (dense walk,
     *    or a warm list that contains the touched set).  g is written although the signature says const.
     *  arrivals != NULL (DEVICE uint32 [65 * 32] = 65 counters on 128-byte lines of their own, zero before the first call; left
     *    zero): the tail's work -- step count, schedule tick,
     *    published rates, lazy log, snapshot, GradScaler.update(), clearing found_inf -- is done by the LAST WORKGROUP TO ARRIVE
     *    inside the update kernel (every workgroup reads the step's scalars first thing and announces itself with a
     *    relaxed atomic when it is done: the last one knows nobody will read them again), and no tail kernel is launched. */
    uint32_t zero_grad_after;
    uint32_t *arrivals;
    /* Ray data parallelism (new work: the reference has no multi-GPU path, tools/details.md:24).  Only with warm_groups, not
     * with replay / warm_zero_grad_from / g16:
     *  compact_grad != NULL (DEVICE f32 [4 * n_warm_groups]): the gradient of list entry j is compact_grad[4 j .. 4 j + 4) -- the
     *    compact exchange buffer after its collective (pvd_segments_gather_zero_check filled it in list order) -- instead of
     *    g[4 warm_groups[j] ..): no pass that puts the summed rows back into g.  A SHARDED update passes its slice of the list
     *    (warm_groups + j0, n_warm_groups = j1 - j0) with the reduce-scattered chunk that belongs to it;
     *  compact_param_out != NULL (DEVICE f32 [4 * n_warm_groups]): the updated parameters of entry j are ALSO written there (on
     *    a skipped step: the unchanged ones) -- the chunk a sharded update all-gathers;
     *  tail_clear (DEVICE f32), tail_clear_stride, tail_clear_n: the tail also zeroes tail_clear[k * tail_clear_stride], k <
     *    tail_clear_n -- the flag words of the exchange buffer's chunks, so that the next step's gather starts from zeros. */
    const float *compact_grad;
    float *compact_param_out;
    float *tail_clear;
    uint32_t tail_clear_stride, tail_clear_n;
} pvd_adamw_extras;
int pvd_adamw_step_ex(float *p, const float *g, float *m, float *v, uint64_t n, const uint64_t *segment_ends_host,
                      uint32_t n_segments, float *lr, double beta1, double beta2, double eps, double weight_decay,
                      float *step, const float *grad_scale, const float *found_inf, const pvd_adamw_extras *extras_host,
                      pvd_stream_t stream);

/* Replay the deferred decay of the cold groups (see pvd_adamw_extras.lazy_log) and empty the log.  status (DEVICE int32, may
 * be NULL) receives the number of replayed steps, or -1 if the log had overflowed (the decays beyond lazy_capacity are lost:
 * flush before that). */
int pvd_adamw_lazy_flush(float *p, uint64_t n, const uint64_t *segment_ends_host, uint32_t n_segments,
                         const uint32_t *cold_bits, const float *lazy_log, uint32_t *lazy_count, double weight_decay,
                         int32_t *status, pvd_stream_t stream);

/* found_inf[0] = 1 if any of g[0..n) is inf/nan (never cleared): the read-only inf check GradScaler.step needs
 * for an optimizer that unscales inside its own kernel (n multiple of 4). */
int pvd_check_finite(const float *g, uint64_t n, float *found_inf, pvd_stream_t stream);
int pvd_check_finite_f16(const void *g, uint64_t n, float *found_inf, pvd_stream_t stream); /* n multiple of 8 */
/* Both in one launch, for a flat fp32 gradient one of whose ranges lives in a half-precision buffer instead (pvd_adamw_extras.g16: the
 * fp32 elements [skip_begin, skip_end) are written by nobody and not read here): g[0, n) outside the range, then g16[0, n16).
 * n, skip_begin, skip_end multiples of 4; n16 a multiple of 8. */
int pvd_check_finite_mixed(const float *g, uint64_t n, uint64_t skip_begin, uint64_t skip_end, const void *g16, uint64_t n16,
                           float *found_inf, pvd_stream_t stream);

/* Segment-table operations over a flat f32 buffer.  segs = n_segs x {start, dst, len} (uint32, in elements): `start`
 * indexes `flat`, `dst` indexes the compact buffer `buf`.  Only the table rows the occupancy grid lets a sample touch can
 * ever hold a gradient, so the per-step zero_grad (utils.py:1012 `optimizer.zero_grad()`), GradScaler's inf check
 * (utils.py:1016 `scaler.step`) and the ray-DP gradient exchange only need those rows.
 *   op 0: flat[start+i] = 0            op 1: buf[dst+i] = flat[start+i]   (gather)
 *   op 2: flat[start+i] = buf[dst+i]   op 3: found_inf[0] = 1 if any flat[start+i] is inf/nan (never cleared)
 *   op 4: op 2 and op 3 in one pass (the exchanged gradient is checked while it is put back: ray-DP's step has one launch fewer)
 * Segments whose start, dst and len are multiples of 4 move as float4. */
int pvd_segments_op(int op, float *flat, float *buf, const uint32_t *segs, uint32_t n_segs, float *found_inf,
                    pvd_stream_t stream);
/* Ray-DP, what goes on the wire, in one pass: buf[dst+i] = flat[start+i] (gather), flat[start+i] = 0 (the NEXT step's zero_grad),
 * and a workgroup that moves an inf / nan stores 1 into slots[k * slot_stride] for every k < n_slots (<= 64) -- one flag word per
 * chunk of the exchange buffer.  The collective that follows sums the flag words with the data, so after it every rank holds the
 * step's GLOBAL found_inf in its chunk (GradScaler's inf check + all-reduce of the verdict, with no launch of their own); the
 * update's tail zeroes the words again (pvd_adamw_extras.tail_clear).  The slots must be zero on entry. */
int pvd_segments_gather_zero_check(float *flat, float *buf, const uint32_t *segs, uint32_t n_segs, float *slots,
                                   uint32_t slot_stride, uint32_t n_slots, pvd_stream_t stream);

/* out[0] = sum_r coef[r] * sum_{i in [begin[r], end[r])} |p[i]|  (value of the L1 regulariser; scratch: 1024 floats).
 * out == NULL: only the 1024 partial sums are left in scratch (for pvd_distill_loss_final's `extra`). */
int pvd_l1_ranges(const float *p, const uint64_t *begin_host, const uint64_t *end_host, const float *coef_host,
                  uint32_t n_ranges, float *scratch, float *out, pvd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PVD_HIP_H */
