/* bags_raster.h -- C ABI of libbags_raster.so, the MI355X (gfx950) pose-differentiable Gaussian rasterizer.
 *
 * Drop-in boundary.  The reference reaches its rasterizer through the Python package
 * `diff_gaussian_rasterization` (gaussian_renderer/__init__.py:14), whose native half is a pybind11 torch
 * extension living in the (empty) `3dgs-pose` submodule (.gitmodules:4-6, README.md:126).  That extension's
 * entry points -- rasterize forward, rasterize backward -- are what this header replaces, as plain C:
 *
 *   reference call site                                              replaced by
 *   ---------------------------------------------------------------  ----------------------------------
 *   GaussianRasterizer.forward(...)  gaussian_renderer/__init__.py:110-121   bags_forward_prepare + bags_forward_finish
 *   autograd backward of that call   train.py:331 (loss.backward)            bags_backward
 *   GaussianRasterizationSettings    gaussian_renderer/__init__.py:50-65     BagsSettings (POD)
 *   state kept between fwd and bwd   (fork: geom/binning/image byte tensors)  caller-owned buffers sized by bags_*_size
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; fp32, contiguous, row-major;
 *   - the library allocates nothing that outlives a call, keeps no per-call state, never frees caller memory;
 *   - all work is enqueued on `stream` (a hipStream_t; NULL = default stream) of the current device;
 *     bags_forward_prepare performs ONE stream synchronisation (to hand the instance count to the host);
 *   - return value: 0 = ok, negative = error; bags_last_error() gives a thread-local message;
 *   - matrices follow the reference's row-vector convention: p_view = [x y z 1] * viewmatrix
 *     (utils/graphics_utils.py:26-33, scene/cameras.py:107-108), 16 floats row-major.
 */
#ifndef BAGS_RASTER_H
#define BAGS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BAGS_ABI_VERSION 10
#define BAGS_TILE 16

enum { BAGS_OK = 0, BAGS_ERR_ARG = -1, BAGS_ERR_HIP = -2, BAGS_ERR_SIZE = -3, BAGS_ERR_DEVICE = -4 };
enum { BAGS_DEPTH_Z = 0, BAGS_DEPTH_DISTANCE = 1 };   /* README.md:126: sort key is view z, or distance for cubemaps */
/* Which tiles a Gaussian is binned into.  BAGS_TILES_AABB is the stock rule: every tile overlapping the square
 * [p - r, p + r + 15] with r = ceil(3 sqrt(lambda_max)).  BAGS_TILES_OPACITY intersects that rectangle with the
 * axis-aligned bounds of the ellipse alpha >= 1/255 (half extents sqrt(2 ln(255 o) cov_xx), sqrt(.. cov_yy), with a
 * 2 % + 0.1 px safety margin): every (tile, Gaussian) pair it drops has alpha < 1/255 on all 256 pixels, i.e. is skipped
 * by the compositing loop anyway, so image, radii and every gradient are unchanged while the sorted instance list
 * shrinks.  Inside a rectangle of at most 8 x 8 tiles it further drops every tile the ellipse itself does not reach
 * (minimum of d^T Q d over the square of the tile's pixel centres > 1.02 (2 ln(255 o) + 0.022)): the same guarantee.
 * Together 40 % fewer instances on BASELINE config 3 (3.45 M -> 2.07 M). */
enum { BAGS_TILES_AABB = 0, BAGS_TILES_OPACITY = 1 };
/* How the per-tile depth-ordered instance lists are built.  AUTO: tile-binned (csrc/binning.hip: (block, tile) count matrix
 * in LDS, per-tile sort of (depth, id) pairs in LDS; five launches, no global sort) whenever the image has at most 32768
 * tiles and P <= 16.7 M, else RADIX.  RADIX: depth-sort the Gaussians, emit, stable radix sort by tile (csrc/sort.hip).
 * Both produce the same lists bit for bit. */
enum { BAGS_BINNING_AUTO = 0, BAGS_BINNING_RADIX = 1 };
/* Gradient of the EWA Jacobian for a Gaussian whose view-space point lies outside 1.3 x the field of view, where the forward
 * uses the clamped t.x = +-1.3 tanfovx t.z (same for y).  Forward values do not depend on this switch.
 * STOCK (default): the rule of upstream diff-gaussian-rasterization's computeCov2DCUDA backward, which the reference's fork
 * (README.md:126) inherits: dL/dt.x is zeroed (x_grad_mul) and dL/dt.z takes 2 h_x t.x / t.z^3 dL/dJ02 with the clamped t.x
 * held CONSTANT.  EXACT: differentiates the clamped expression itself (t.x moves with t.z), i.e. half of that one term. */
enum { BAGS_CLAMP_GRAD_STOCK = 0, BAGS_CLAMP_GRAD_EXACT = 1 };
/* Backward of conic = cov2D^-1.  STOCK (default, ABI 8): upstream computeCov2DCUDA divides by det^2 + 1e-7 where the derivative of the
 * 2x2 inverse has det^2 ("denom2inv"); the reference's fork (README.md:126) inherits it.  EXACT: det^2 (what rounds 1-4 shipped).
 * det >= 0.09 (the 0.3 px dilation), so the two differ by at most 1.2e-5 relative on one Gaussian's dL/dcov2D (measured 2e-7 on the
 * gradient tensors of a scene of small splats, tests/test_oracle_cpu.py).  Forward values do not depend on this switch. */
enum { BAGS_CONIC_GRAD_STOCK = 0, BAGS_CONIC_GRAD_EXACT = 1 };
enum { BAGS_BWD_ALL = 0, BAGS_BWD_BLEND = 1, BAGS_BWD_PREPROCESS = 2 };   /* BagsBackwardArgs.phase (ABI 9) */

/* GaussianRasterizationSettings (gaussian_renderer/__init__.py:50-65) */
typedef struct BagsSettings {
    int32_t image_height, image_width;
    float tanfovx, tanfovy;          /* from the camera's STATIC fov (gaussian_renderer/__init__.py:47-48) */
    float scale_modifier;
    int32_t sh_degree;               /* active degree 0..3 */
    int32_t sh_coeffs;               /* M: coefficients stored per Gaussian in `shs` ((max_sh_degree+1)^2) */
    int32_t depth_key;               /* BAGS_DEPTH_Z | BAGS_DEPTH_DISTANCE */
    int32_t debug;                   /* !=0: synchronise + check after every kernel */
    int32_t debug_iter;              /* carried for error messages only */
    int32_t tile_bounds;             /* BAGS_TILES_AABB | BAGS_TILES_OPACITY */
    int32_t binning;                 /* BAGS_BINNING_AUTO | BAGS_BINNING_RADIX */
    int32_t clamp_grad;              /* BAGS_CLAMP_GRAD_STOCK | BAGS_CLAMP_GRAD_EXACT (backward only) */
    int32_t conic_grad;              /* BAGS_CONIC_GRAD_STOCK | BAGS_CONIC_GRAD_EXACT (backward only; was reserved0 = 0 before ABI 8) */
    const float* bg;                 /* (3)   */
    const float* viewmatrix;         /* (4,4) world->view, transposed (W2C^T) */
    const float* projmatrix;         /* (4,4) viewmatrix * intrinsic */
    const float* intrinsic;          /* (4,4) projection^T; [0][0],[1][1] give the focal lengths, row 2 the shift direction */
    const float* campos;             /* (3)   */
} BagsSettings;

/* the keyword tensors of GaussianRasterizer.forward (gaussian_renderer/__init__.py:110-121) */
typedef struct BagsInputs {
    int32_t P;
    const float* means3D;            /* (P,3) */
    const float* means2D;            /* (P,3) additive NDC offset, normally zeros; may be NULL */
    const float* shift_factors;      /* (3) entrance-pupil polynomial; may be NULL (= zeros) */
    const float* shs;                /* (P,M,3) or NULL */
    const float* colors_precomp;     /* (P,3)   or NULL  (exactly one of shs / colors_precomp) */
    const float* opacities;          /* (P,1) */
    const float* scales;             /* (P,3) or NULL */
    const float* rotations;          /* (P,4) (w,x,y,z) or NULL */
    const float* cov3D_precomp;      /* (P,6) xx,xy,xz,yy,yz,zz or NULL (exactly one of scales+rotations / cov3D) */
    /* (ABI 7) the reference's two feature parameters as they are stored, without GaussianModel.get_features' torch.cat
     * (scene/gaussian_model.py:131-134: 96 MB read + 96 MB written per call at 500 k Gaussians, and as much again for the split of
     * the gradient): when non-NULL, `shs` is features_dc (P,1,3) and this is features_rest (P,M-1,3), M = settings.sh_coeffs >= 2 */
    const float* shs_rest;
} BagsInputs;

/* caller-owned state that survives from forward to backward */
typedef struct BagsState {
    void* geom;    size_t geom_bytes;     /* >= bags_geom_size(P)            */
    void* binning; size_t binning_bytes;  /* >= bags_binning_size(I, W, H)   */
    void* image;   size_t image_bytes;    /* >= bags_image_size(W, H)        */
} BagsState;

typedef struct BagsForwardOut {
    float*   color;      /* (3,H,W) */
    int32_t* radii;      /* (P)     */
    float*   depth;      /* (1,H,W) expected view-space z            */
    float*   weights;    /* (1,H,W) accumulated alpha = 1 - T_final  */
    float*   mean2D;     /* (P,2)   pixel centres (0 for culled)     */
} BagsForwardOut;

typedef struct BagsBackwardArgs {
    const float* grad_color;         /* (3,H,W) dL/dimage */
    int64_t num_rendered;            /* I returned by bags_forward_prepare */
    void* workspace; size_t workspace_bytes;   /* >= bags_backward_workspace_size(P, I) */
    /* outputs; any may be NULL when not needed.  All are fully overwritten unless `accumulate` (below) says otherwise. */
    float* grad_means3D;             /* (P,3)   */
    float* grad_means2D;             /* (P,3)   NDC units, z = 0 */
    float* grad_means2D_densify;     /* (P,3)   sum over pixels of |per-pixel NDC gradient|, z = 0 */
    float* grad_shs;                 /* (P,M,3) */
    float* grad_colors_precomp;      /* (P,3)   */
    float* grad_opacities;           /* (P,1)   */
    float* grad_scales;              /* (P,3)   */
    float* grad_rotations;           /* (P,4)   */
    float* grad_cov3D_precomp;       /* (P,6)   */
    float* grad_viewmatrix;          /* (4,4)   */
    float* grad_projmatrix;          /* (4,4)   */
    float* grad_intrinsic;           /* (4,4)   */
    float* grad_campos;              /* (3)     */
    float* grad_shift_factors;       /* (3)     */
    /* capacity the forward's binning buffer was carved for when that was a speculative finish (state.binning then holds
     * bags_binning_size(binning_capacity, W, H) bytes and num_rendered is the TRUE count, which sizes the workspace);
     * 0 = the buffer was sized for num_rendered itself (bags_forward_finish) */
    int64_t binning_capacity;
    /* != 0: the seven Gaussian-parameter gradients (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp) are
     * ADDED to what their buffers hold instead of overwriting them -- the views of one optimisation step accumulate in place, as
     * autograd would do with one add pass per view and tensor; every other output is overwritten as before (ABI 6) */
    int32_t accumulate;
    /* Dense-scene mode of the backward.  blend_bwd writes one 48-byte record per instance and preprocess_bwd sums each Gaussian's
     * records; an instance behind its tile's deepest contributor holds a ZERO record.  Above this many instances per tile (scene
     * average) a byte per record says whether blend_bwd wrote it: no zero record is written or read (at 1800 instances per tile
     * 85 % of them are).  0 = the library's default (480; profiles/r05/ab_dense.txt), < 0 = never, > 0 = that threshold.  Results
     * do not depend on it (ABI 8; the field was reserved1 = 0 before, and rounds 3-4 read an environment variable here). */
    int32_t dense_per_tile;
    float* grad_shs_rest;            /* (P,M-1,3), with inputs.shs_rest: grad_shs is then the (P,1,3) gradient of features_dc (ABI 7) */
    /* (ABI 9) Which half of the backward this call enqueues.  BAGS_BWD_ALL (0): everything, as before.  BAGS_BWD_BLEND: the per-tile
     * half only (blend_bwd: dL/dimage -> one partial-gradient record per instance, in the workspace).  BAGS_BWD_PREPROCESS: the
     * per-Gaussian half only (preprocess_bwd + pose_reduce: records -> every output above); same workspace, same arguments, after a
     * BLEND call.  The split exists for callers that differentiate several views of one step on several streams with `accumulate`:
     * the read-modify-write of the shared gradient buffers must run in view order, so the second half of view k waits for the
     * second half of view k-1 (an event between two calls) while its first half -- 85 % of the backward's time -- does not. */
    int32_t phase;
    int32_t reserved2;               /* 0 */
    /* (ABI 10) Factored SH gradient, for steps that differentiate SEVERAL views into the same Gaussians.  dL/dshs of one view is an
     * outer product per Gaussian -- basis(direction from that view's camera) x dL/dcolour, 16 x 3 floats made of 3 -- and writing
     * (and, with `accumulate`, first reading) that 192-byte row is two thirds of the per-Gaussian half's traffic.  With grad_dldc
     * non-NULL (and grad_shs / grad_shs_rest NULL) the backward writes only the (P,3) dL/dcolour (after the colour clamp; zero for a
     * culled Gaussian) here; bags_sh_gradient_from_views then forms the gradient rows of ALL the step's views in one pass, reading
     * 12 bytes per Gaussian and view and writing each row once.  Same products, added in view order: bit-identical to V backwards
     * with `accumulate`.  SH colour path only (inputs.shs given). */
    float* grad_dldc;
} BagsBackwardArgs;

/* integer artefacts for bit-exact parity checks (all device pointers, any may be NULL) */
typedef struct BagsDebugViews {
    uint32_t* tiles_touched;         /* (P) */
    uint32_t* rect;                  /* (P,4) minx,miny,maxx,maxy */
    uint32_t* depth_bits;            /* (P) float bits of the sort depth (0xFFFFFFFF if culled) */
    uint32_t* point_list;            /* (I) Gaussian id per sorted instance */
    uint64_t* keys_sorted;           /* (I) (tile<<32)|depth_bits per sorted instance */
    uint32_t* ranges;                /* (T,2) */
    uint32_t* n_contrib;             /* (H,W) */
    float*    final_T;               /* (H,W) */
} BagsDebugViews;

int         bags_abi_version(void);
const char* bags_last_error(void);
/* "src=<12 hex digits> commit=<git short hash|nogit>[+dirty]": a hash of the kernel sources this library was compiled from and
 * the last commit that touched them (ABI 8).  Measurement bookkeeping only: profiles/rNN/traffic.json records the string of the
 * build its counters were collected on, and bench.py prints roofline.traffic only when it matches the library it is timing. */
const char* bags_build_info(void);

size_t bags_geom_size(int32_t P);
size_t bags_binning_size(int64_t num_rendered, int32_t width, int32_t height);
size_t bags_image_size(int32_t width, int32_t height);
size_t bags_backward_workspace_size(int32_t P, int64_t num_rendered);

/* Phase 1: per-Gaussian projection/covariance/colour, depth ordering of the Gaussians, instance offsets.
 * Writes radii and mean2D.  Synchronises `stream` once and returns the instance count in *host_num_rendered. */
int bags_forward_prepare(const BagsSettings*, const BagsInputs*, const BagsState*, const BagsForwardOut*,
                         int64_t* host_num_rendered, void* stream);
/* Phase 2: instance emission, per-tile stable sort, tile ranges, front-to-back blend.  state.binning must hold
 * bags_binning_size(num_rendered, W, H) bytes. */
int bags_forward_finish(const BagsSettings*, const BagsInputs*, const BagsState*, const BagsForwardOut*,
                        int64_t num_rendered, void* stream);
/* Forward without a host round trip in the middle (speculative instance capacity).
 *   1. bags_forward_prepare_async: phase 1 as above, without the synchronisation.  *host_num_rendered is a caller-owned
 *      PINNED host word (hipHostMalloc / torch pinned memory: device-mapped) that will receive the instance count; set it to a
 *      sentinel (e.g. 0xFFFFFFFF) before the call.
 *   2. bags_forward_finish_speculative: phase 2 enqueued immediately, sized for a caller-guessed upper bound `capacity`
 *      (e.g. 1.2 x the previous frame's count; state.binning holds bags_binning_size(capacity, W, H) bytes); the
 *      kernels read the true count on the device and render every tile empty if it exceeds `capacity`.
 *      WHEN THE COUNT ARRIVES (ABI 5): on the tile-binned path no launch of phase 1 computes it any more -- the first launch
 *      of THIS call does (tile ranges, count and block bases are formed inside the emission launch) and stores it into the
 *      pinned word at system scope, a few tens of microseconds into phase 2.  On the radix path, for P == 0, or when the
 *      word is not device-mapped, step 1 delivers it by an enqueued copy as before.  Either way: the word holds the count
 *      no later than the end of step 2; a caller that needs the count BEFORE it enqueues phase 2 uses bags_forward_prepare.
 *   3. The caller polls the word (or waits for an event recorded behind step 2) -- right away, while the GPU is busy with
 *      phase 2, or as late as the entry of bags_backward -- and compares: if the count exceeds `capacity` the outputs are
 *      invalid (every tile was rendered empty) and phase 2 must be redone with bags_forward_finish on a buffer of
 *      bags_binning_size(count, W, H) bytes (phase 1 results in state.geom stay valid; that exact re-run does not touch the
 *      pinned word again, so the word may be reused for another call as soon as it has been read).  When it fits,
 *      bags_backward takes the true count as num_rendered and `capacity` as binning_capacity. */
int bags_forward_prepare_async(const BagsSettings*, const BagsInputs*, const BagsState*, const BagsForwardOut*,
                               uint32_t* host_num_rendered, void* stream);
int bags_forward_finish_speculative(const BagsSettings*, const BagsInputs*, const BagsState*, const BagsForwardOut*,
                                    int64_t capacity, void* stream);
/* Backward of the whole op from dL/dimage. */
int bags_backward(const BagsSettings*, const BagsInputs*, const BagsState*, const BagsBackwardArgs*, void* stream);

/* (ABI 10) The SH-gradient rows of up to BAGS_MAX_SH_VIEWS views of one step from their factored form (BagsBackwardArgs.grad_dldc):
 *   grad_shs[g][t][c] (+)= sum over the views v, in order, of basis_t(normalize(means3D[g] - campos_v)) * dldc_v[g][c]
 * for the active degree's coefficients (the others are written as zeros / left as they are when accumulating).  M = coefficients
 * per Gaussian in grad_shs ((P,M,3)), or -- grad_shs_rest non-NULL -- grad_shs is (P,1,3) and grad_shs_rest (P,M-1,3) (the pair
 * of BagsInputs.shs_rest).  accumulate != 0: the views are added, in order, to what the buffers hold; else the first view overwrites.
 * More than BAGS_MAX_SH_VIEWS views: call again with accumulate = 1 for the next batch (the order of the additions is preserved). */
#define BAGS_MAX_SH_VIEWS 16
typedef struct BagsShViews {
    int32_t n_views;
    int32_t reserved;
    const float* campos[BAGS_MAX_SH_VIEWS];   /* (3) each: the campos tensor the view's op call was given */
    const float* dldc[BAGS_MAX_SH_VIEWS];     /* (P,3) each: that view's BagsBackwardArgs.grad_dldc */
} BagsShViews;
int bags_sh_gradient_from_views(int32_t P, int32_t M, int32_t sh_degree, const float* means3D, const BagsShViews* views,
                                float* grad_shs, float* grad_shs_rest, int32_t accumulate, void* stream);

/* Copies integer artefacts out of the state buffers (parity tests / debugging). */
int bags_debug_views(const BagsSettings*, const BagsInputs*, const BagsState*, int64_t num_rendered,
                     const BagsDebugViews*, void* stream);

/* Opt-in per-stage device timing on the caller's stream: bench.py's roofline leg.  mode 0 = off (default; the hot path records
 * nothing), 1 = only the dominant kernel (blend_bwd), by a start / stop hipEvent pair ATTACHED TO THE KERNEL'S DISPATCH
 * (hipExtLaunchKernelGGL: no packet between the step's launches; round 5 bracketed it with two hipEventRecord calls, 10-25 us of
 * bubbles per step inside the region being timed), 2 = every stage, bracketed by recorded events (each pair costs a few microseconds
 * of stream bubble: for a separate, untimed pass).  bags_profile_stride(n): in mode 1 only every n-th launch carries events (the mean
 * launch time of a steady loop needs no more; n = 1 by default).  bags_profile_read synchronises on the events, returns the number of
 * stages, fills up to `max_stages` entries (stage name, summed milliseconds, number of timed intervals) and clears the accumulators. */
#define BAGS_PROFILE_MAX_STAGES 16
int bags_profile_enable(int mode);
int bags_profile_stride(int n);
int bags_profile_read(int max_stages, const char** names, double* total_ms, int64_t* calls);

/* Fused photometric loss terms.  Replaces the reference's l1_loss + ssim pair
 * (utils/loss_utils.py:18-19 and :48-76: five dense 11x11 depthwise conv2d calls and their autograd backward; call
 * site train.py:311-313, combined at train.py:325 as (1-lambda) L1 + lambda (1-SSIM)).
 *   image, gt     device, fp32, contiguous (C,H,W)
 *   workspace     caller-owned, bags_loss_workspace_size(C,H,W) bytes, kept from forward to backward
 *   out_terms     device float[2]: { mean |image-gt| , mean SSIM }
 *   grad_terms    device float[2]: upstream dL/d{L1 mean, SSIM mean} (read on the device: no host round trip)
 *   grad_image    device (C,H,W): dL/dimage
 * Stream-ordered on `stream`; sums are taken in a fixed order (bitwise reproducible). */
size_t bags_loss_workspace_size(int32_t C, int32_t H, int32_t W);
int bags_loss_forward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, void* workspace,
                      size_t workspace_bytes, float* out_terms, void* stream);
int bags_loss_backward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, const void* workspace,
                       size_t workspace_bytes, const float* grad_terms, float* grad_image, void* stream);

/* The same pair with train.py:325's combination inside (ABI 7): loss = (1 - lambda_dssim) * L1 + lambda_dssim * (1 - SSIM), formed
 * in fp32 with the roundings of the reference's PyTorch expression.  The scalar arithmetic around the two terms and its autograd
 * backward were ten one-element PyTorch launches per iteration (48 us of a 1.05 ms iteration at 1080p).
 *   out_loss_terms  device float[3]: { loss, mean |image-gt|, mean SSIM }
 *   grad_loss       device float[1]: upstream dL/dloss (read on the device)
 * Same workspace as above (bags_loss_workspace_size), kept from forward to backward. */
int bags_photometric_loss_forward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, void* workspace,
                                  size_t workspace_bytes, float lambda_dssim, float* out_loss_terms, void* stream);
int bags_photometric_loss_backward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, const void* workspace,
                                   size_t workspace_bytes, float lambda_dssim, const float* grad_loss, float* grad_image, void* stream);

/* The pose -> matrix chain of scene/cameras.py:356-381 in one launch each way: replaces
 * get_world_view_transform / get_full_proj_transform / get_intrinsic / get_camera_center (~40 PyTorch kernels and six 4x4
 * inversions per render() call, gaussian_renderer/__init__.py:57,58,61) and their autograd backward.  All pointers are
 * device pointers; optional ones may be NULL. */
typedef struct BagsCamera {
    const float* init_quaternion;    /* (4) w,x,y,z of the world-to-camera rotation (scene/cameras.py:98) */
    const float* delta_quaternion;   /* (4) learnable, added before normalisation (scene/cameras.py:360) */
    const float* init_translation;   /* (3) */
    const float* delta_translation;  /* (3) learnable */
    const float* fovx;               /* scalar, learnable (scene/cameras.py:109-110) */
    const float* fovy;
    const float* global_rotation;    /* (3,3) row-major or NULL: R <- G R (scene/cameras.py:361) */
    const float* global_translation_scale;   /* scalar or NULL: t <- s t (scene/cameras.py:367-370) */
    float znear, zfar;
} BagsCamera;
/* viewmatrix, projmatrix, intrinsic: (4,4); campos: (3) */
int bags_camera_forward(const BagsCamera* cam, float* viewmatrix, float* projmatrix, float* intrinsic, float* campos, void* stream);
/* upstream gradients may be NULL (= zero); gradient outputs may be NULL (= not wanted) */
int bags_camera_backward(const BagsCamera* cam, const float* g_viewmatrix, const float* g_projmatrix, const float* g_intrinsic,
                         const float* g_campos, float* g_delta_quaternion, float* g_delta_translation, float* g_fovx,
                         float* g_fovy, float* g_global_rotation, float* g_global_translation_scale, void* stream);

/* Image-space distortion resampling: the apply2gt == False branch of apply_distortion
 * (utils/util_distortion.py:271-311, call site train.py:255-263) in one pass each way:
 *   flow  = interpolate(control flow (h,w,2) -> (flow_H, flow_W), bilinear, align_corners=False)
 *   out   = center_crop(grid_sample(image (C,H,W), flow, bilinear, zeros, align_corners=True), crop_H, crop_W)
 *   mask  = !(out[0] == 0 && out[1] == 0)                                    (may be NULL)
 *   flow_out (crop_H, crop_W, 2): the upsampled flow at the cropped pixels     (may be NULL)
 * Pass h == flow_H, w == flow_W to resample with an already dense flow (the cached flow_apply2_gt_or_img path).
 * Backward: grad_image (C,H,W) is gathered per 16x16 source tile into a 64-bit fixed-point accumulator in LDS (every
 * element is written: no memset needed); grad_ctrl (h,w,2) is gathered per control node in a fixed order.  No global float
 * atomics (integer ones only on per-tile list counters), so both gradients are bitwise reproducible.  Needs a caller-owned
 * workspace of bags_resample_workspace_size(H, W, crop_H, crop_W) bytes (per-tile lists and the dense dL/dflow).  Either
 * gradient may be NULL.  C <= 24 per call (2 KB of LDS per channel).  A non-finite grad_out element makes the source tiles
 * its output tile is listed on NaN; magnitudes below 2^-100 are treated as zero. */
size_t bags_resample_workspace_size(int32_t H, int32_t W, int32_t crop_H, int32_t crop_W);
int bags_resample_forward(const float* image, int32_t C, int32_t H, int32_t W, const float* ctrl_flow, int32_t h, int32_t w,
                          int32_t flow_H, int32_t flow_W, int32_t crop_H, int32_t crop_W, float* out, float* mask,
                          float* flow_out, void* stream);
int bags_resample_backward(const float* image, int32_t C, int32_t H, int32_t W, const float* ctrl_flow, int32_t h, int32_t w,
                           int32_t flow_H, int32_t flow_W, int32_t crop_H, int32_t crop_W, const float* grad_out,
                           void* workspace, size_t workspace_bytes, float* grad_image, float* grad_ctrl, void* stream);

/* The parameter activations that feed the op, one launch each way: GaussianModel.get_features
 * (cat of features_dc and features_rest), get_opacity (sigmoid), get_scaling (exp), get_rotation (normalize) --
 * scene/gaussian_model.py:118-141.  K = SH coefficients per Gaussian (1 + rest).  Outputs / gradients may be NULL; with
 * shs (g_shs) NULL the feature pointers may be NULL too and the launch is one thread per Gaussian (a container that keeps
 * features_dc and features_rest as views of one (P,K,3) tensor needs no concatenation: bags_raster.gaussians.GaussianBag). */
typedef struct BagsRawGaussians {
    int32_t P, K;
    const float* features_dc;        /* (P,1,3)   */
    const float* features_rest;      /* (P,K-1,3) */
    const float* opacity;            /* (P,1) pre-sigmoid */
    const float* scaling;            /* (P,3) log-scale   */
    const float* rotation;           /* (P,4) unnormalised quaternion */
} BagsRawGaussians;
int bags_activations_forward(const BagsRawGaussians* raw, float* shs, float* opacity, float* scales, float* rotations, void* stream);
int bags_activations_backward(const BagsRawGaussians* raw, const float* g_shs, const float* g_opacity, const float* g_scales,
                              const float* g_rotations, float* g_features_dc, float* g_features_rest, float* g_opacity_raw,
                              float* g_scaling, float* g_rotation, void* stream);

/* distCUDA2 of the reference's second native dependency (simple_knn._C, imported at scene/gaussian_model.py:20, called at
 * scene/gaussian_model.py:177 to initialise the scales): out[i] = mean of the squared distances from point i to its three
 * nearest neighbours (self excluded by index; coincident points count with distance 0; with fewer than four points the
 * missing neighbours contribute FLT_MAX).  points: device (P,3) fp32; out: device (P); workspace: caller-owned,
 * bags_knn_workspace_size(P) bytes.  Exact (uniform grid + shell walk), no host round trip, stream-ordered. */
size_t bags_knn_workspace_size(int32_t P);
int bags_knn_mean_dist2(const float* points, int32_t P, void* workspace, size_t workspace_bytes, float* out, void* stream);

/* compute_relocation of the fork's MCMC path (utils/reloc_utils.py:11-13): its only caller is commented out in
 * the reference (scene/gaussian_model.py:23,494-504); exported so the symbol exists, returns BAGS_ERR_ARG. */
int bags_compute_relocation(const float* opacity_old, const float* scale_old, const int32_t* N, const float* binoms,
                            int32_t n_max, int32_t P, float* opacity_new, float* scale_new, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BAGS_RASTER_H */
