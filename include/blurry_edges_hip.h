/*
 * blurry_edges_hip.h -- C ABI of libblurry_edges_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary of the Blurry-Edges hot path: local_stage CNN -> blurred-wedge renderer with ridge
 * colour solve -> DfD depth solve (+ overlap tiling).  The reference has no FFI of its own (100 % Python on
 * ATen ops); each entry point below replaces the ATen op sequence of the reference function it cites and is
 * what a binding for that function would call (ctypes stub: INTEGRATION.md; the build's own Python host
 * side in blurry-edges_amd/{models,utils} binds exactly these symbols).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in _host; tensors are dense, fp32,
 *     row-major in the layout written next to the parameter; nothing is allocated or freed by the library
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all calls are asynchronous on
 *     it and graph-capturable (no allocation, no synchronisation inside)
 *   - return value: BE_OK (0) or a negative BE_E* code; be_last_error() gives the message of the last
 *     failure on the calling thread.  Invalid shapes are refused on the host before anything is launched.
 */
#ifndef BLURRY_EDGES_HIP_H
#define BLURRY_EDGES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BE_OK            0
#define BE_EINVAL       -1   /* bad argument (null pointer, size, alignment, unsupported shape) */
#define BE_EWORKSPACE   -2   /* workspace too small                                          */
#define BE_ELAUNCH      -3   /* HIP reported an error at launch                              */

#define BE_R             21  /* patch side, utils/args.py:11                                 */
#define BE_NPIX          441
#define BE_LOCAL_OUT     10  /* LocalStage output_dim, models/local_stage.py:31              */

int         be_version(void);
const char* be_last_error(void);

/* ---------------------------------------------------------------------------------------------------
 * DepthEtas  (utils/depth_etas.py:3-37)
 * ------------------------------------------------------------------------------------------------- */

/* Constants of DepthEtas.__init__ (utils/depth_etas.py:4-21), computed by the caller exactly as the
 * reference does (python float64 -> rounded to fp32 on use; intercept / sin / cos are the fp32 tensor
 * values) so branch decisions agree bit for bit. */
typedef struct be_depth_consts {
    float s;            /* cam_params['s']                                   */
    float numerator;    /* 2 s^2 (rho_2 - rho_1)                        :13 */
    float den_const;    /* -s (rho_1-rho_2)(rho_1 s + rho_2 s - 2)      :14 */
    float k;            /* denominator_factor_root                      :15 */
    float k2;           /* denominator_factor                           :16 */
    float intercept;    /*                                              :18 */
    float sin_w, cos_w; /* sin/cos(theta_wng = pi/4)                    :21 */
    float sin_m, cos_m; /* sin/cos(theta_mid = 3pi/4)                   :20 */
} be_depth_consts;

/* params2etas (utils/postprocessing_loss.py:88-89): eta = 10^(2 erf(p) - 2), elementwise, n elements. */
int be_params2etas_f32(const float* p, float* eta, int64_t n, void* stream);

/* DepthEtas.etas2depth (utils/depth_etas.py:23-34), elementwise over n (eta1[i], eta2[i]) pairs.
 * branch (optional, may be NULL): int32 id 0..3 of the locus segment taken. */
int be_etas2depth_f32(const be_depth_consts* consts_host, const float* eta1, const float* eta2,
                      float* depth, int32_t* branch, int64_t n, void* stream);

/* DepthEtas.depth2sigma (utils/depth_etas.py:36-37), elementwise. */
int be_depth2sigma_f32(const be_depth_consts* consts_host, const float* depth, float rho_prime,
                       float* eta, int64_t n, void* stream);

/* Config-2 composition: params10 [2P,10] image-major (rows 0..P-1 aperture 1, rows
 * P..2P-1 aperture 2, the order of blurry_edges_test.py:121); depth [P,2]:
 *   depth[i][k] = etas2depth(params2etas(params10[i][8+k]), params2etas(params10[P+i][8+k])). */
int be_local_depth_f32(const be_depth_consts* consts_host, const float* params10, float* depth,
                       int64_t n_pairs, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Blurred-wedge renderer, colours-only pass ("pass A")
 *   replaces PostProcess.forward(colors_only=True) -> get_patches -> get_colors
 *   (blurry_edges_test.py:19-34,81-92) and LocalLoss.get_patches (local_training.py:32-45):
 *   params2dists :43-86, params2etas :88-89, dists2indicators :91-95, A^T A + lambda I, A^T y,
 *   3x3 solve (utils/postprocessing_loss.py:104-112), composite.
 * ------------------------------------------------------------------------------------------------- */
typedef struct be_render_opts {
    float lambda_ridge;    /* (alpha_lambda R^2)^2, utils/postprocessing_loss.py:14                 */
    float w;               /* args.w (1.0), :71-76                                                   */
    float delta_sq;        /* normalized_gaussian delta^2 = fp32(0.07**2), :97-98                    */
    int   wrap_angles;     /* 1: params[4:8] <- remainder(., 2pi) first (blurry_edges_test.py:124)   */
    float lin[BE_R];       /* torch.linspace(-1,1,21) as fp32, :15-16                                */
} be_render_opts;

/* params10 [N,10], patches [N,3,21,21]  ->  colors [N,3(rgb),3(wedge)].
 * Optional outputs (NULL to skip): recon [N,3,21,21] composited patch, boundary [N,21,21],
 * dists [N,2,21,21], wedges [N,3,21,21], gram [N,3,3] (A^T A + lambda I), aty [N,3(wedge),3(rgb)].
 * The 3x3 system is solved in fp64 by cofactors (numerically better than the reference's fp32
 * Cayley-Hamilton inverse; parity is judged against the fp64 oracle). */
int be_render_colors_f32(const be_render_opts* opts_host, const float* params10, const float* patches,
                         float* colors, float* recon, float* boundary, float* dists, float* wedges,
                         float* gram, float* aty, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Full render pass ("pass B") + overlap aggregation
 *   replaces PostProcess.get_patches(colors_only=False) (blurry_edges_test.py:36-79) and the six nn.Fold
 *   aggregations local2global_{color,bndry,depth} (utils/postprocessing_loss.py:151-173; the big-image
 *   stitching of blurry_edges_test_big.py:166-189 is the same fold over a 284x284 record grid).
 * ------------------------------------------------------------------------------------------------- */

/* Strided view of the pixel data of a grid of patch positions, so the kernels gather on read from whatever the
 * caller has: an image pair [2,3,H,W] (s_aperture=3HW, s_chan=HW, s_row=W, s_col=1, s_pi=stride*W, s_pj=stride),
 * the reference's unfolded tensor [2,3,21,21,Hp,Wp] (s_chan=441*P, s_row=21*P, s_col=P, s_pi=Wp, s_pj=1) or
 * flat patches [2,P,3,21,21] (s_aperture=1323*P, s_chan=441, s_row=21, s_col=1, s_pi=1323*Wp, s_pj=1323).
 * Strides are in floats; patch p sits at grid position (p / wp, p % wp). */
typedef struct be_patch_view {
    const float* base;
    int64_t s_aperture, s_chan, s_row, s_col, s_pi, s_pj;
    int wp;
} be_patch_view;

/* be_render_colors_f32 reading its pixels through a view (no [N,3,21,21] copy of the unfolded pair):
 * patch n is aperture n / patches_per_image at grid position n % patches_per_image, i.e. the order of
 * img_patches.flatten(0,1) in blurry_edges_test.py:120-123. */
int be_render_colors_view_f32(const be_render_opts* opts_host, const float* params10, const be_patch_view* view_host,
                              int64_t patches_per_image, float* colors, float* recon, float* boundary, float* dists,
                              float* wedges, float* gram, float* aty, int64_t n, void* stream);

#define BE_RECORD_FLOATS 32   /* per-patch state handed from be_render_full_f32 to be_fold_records_f32 (128 B):
                                 geometry 14, sqrt2*eta for aperture 1 / aperture 2 / refocus 2+2+2, colours 9
                                 ([rgb][wedge]), wedge depths 2, mask-presence flags 1 */

/* params12 [P,12] = 8 shared geometry + eta coefficients (w1,img1),(w2,img1),(w1,img2),(w2,img2), de-normalised
 * (blurry_edges_test.py:135-138); records [P,32] out.  Optional per-patch tensors in the reference's pixel order
 * (NULL to skip): patches [P,2,3,21,21], shpd [P,3,21,21], refoc [P,3,21,21], boundary [P,21,21],
 * depth_map [P,21,21], depth_mask int32 [P,21,21].  densify_w: 1 = the '--densify w' mask rule (:47-50).
 * opts->wrap_angles is honoured (0 for de-normalised parameters, which are already in [0,2pi)). */
int be_render_full_f32(const be_render_opts* opts_host, const be_depth_consts* consts_host, float rho_prime,
                       int densify_w, const float* params12, const be_patch_view* view_host, float* records,
                       float* patches, float* shpd, float* refoc, float* boundary, float* depth_map,
                       int32_t* depth_mask, int64_t n, void* stream);

/* Owner-computes fold of a hp x wp record grid onto an H x W image (stride = patch stride, 2).  Outputs (NULL to
 * skip): image [2,3,H,W], shpd [3,H,W], refoc [3,H,W], bndry [H,W] (all / overlap count), depth [H,W] (mean
 * over patches whose mask covers the pixel) and conf [H,W] (that count / overlap count). */
int be_fold_records_f32(const be_render_opts* opts_host, const float* records, int hp, int wp, int H, int W,
                        int stride, int densify_w, float* image, float* shpd, float* refoc, float* bndry,
                        float* depth, float* conf, void* stream);
/* The same for B images in one launch (records [B][hp*wp][32], every map with a leading batch dimension): a 147 x 147 image is
 * 100 workgroups, a batch fills the chip (GlobalLoss folds the current image and boundary map of every image of a batch). */
int be_fold_records_batch_f32(const be_render_opts* o, const float* records, int B, int hp, int wp, int H, int W, int stride,
                              int densify_w, float* image, float* shpd, float* refoc, float* bndry, float* depth, float* conf,
                              void* stream);

/* nn.Unfold(21, stride) in the order blurry_edges_test.py:120-121 consumes it:
 * img [B,C,H,W] -> out [B, Hp*Wp, C, 21, 21], patch (i,j) = rows stride*i.., cols stride*j.., index i*Wp+j. */
int be_unfold_patches_f32(const float* img, float* out, int B, int C, int H, int W, int stride, void* stream);

/* Feature normalisation between the two stages (blurry_edges_test.py:123-132):
 * params10 [2,P,10] (raw CNN output, image-major) + colors [2,P,9] -> pm [P,38]. */
int be_local_features_f32(const float* params10, const float* colors, float* pm, int64_t P, void* stream);
/* De-normalisation of the GlobalStage output (blurry_edges_test.py:134-138): y [P,12] -> est [P,12]. */
int be_global_denorm_f32(const float* y, float* est, int64_t P, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * The base-class methods of PostProcessBase one by one (for callers that subclass the reference's
 * PostProcessLocalBase / PostProcessGlobalBase and chain them themselves), flat layout:
 * ------------------------------------------------------------------------------------------------- */
/* params2dists (utils/postprocessing_loss.py:43-86): params8 [N,8] -> dists [N,2,21,21]. */
int be_params2dists_f32(const be_render_opts* opts_host, const float* params8, float* dists, int64_t n, void* stream);
/* dists2indicators (:91-95): dists [N,2,21,21], etas [N,2] -> wedges [N,3,21,21]. */
int be_dists2indicators_f32(const float* dists, const float* etas, float* wedges, int64_t n, void* stream);
/* inverse_3by3 (:104-112): n row-major 3x3 matrices (cofactor inverse, fp64 inside). */
int be_inverse3x3_f32(const float* a, float* out, int64_t n, void* stream);
/* get_image_derivative (:114-117): per-plane Sobel magnitude, valid padding: [planes,H,W] -> [planes,H-2,W-2]. */
int be_image_derivative_f32(const float* img, float* out, int64_t planes, int H, int W, void* stream);
/* nn.Fold of a strided patch tensor (local2global_*, :151-173), owner computes.  Element (b,c,r,col,i,j) of the
 * source sits at b*s_b + c*s_c + r*s_r + col*s_col + i*s_pi + j*s_pj (floats).  mode 0: overlap sum, 1: sum /
 * overlap count, 2: number of covering patches whose entry is > 0 (src_int, if not NULL, is read instead of src). */
int be_fold_patches_f32(const float* src, const int32_t* src_int, float* out, int B, int C, int hp, int wp, int H, int W,
                        int stride, int64_t s_b, int64_t s_c, int64_t s_r, int64_t s_col, int64_t s_pi, int64_t s_pj,
                        int mode, void* stream);

/* Adjoints of the methods above and of DepthEtas, for `loss.backward()` through a caller's own LocalLoss / GlobalLoss subclass
 * (local_training.py:32-52,106; global_training.py:62-157,212).  g* = cotangent (dL/d.) with the shape of the tensor it
 * belongs to; derivatives are evaluated in fp64 from the fp32 operands; per-patch sums have a fixed order. */
/* params2dists: gdists [N,2,21,21] -> gparams8 [N,8] (d/d x0,y0,x1,y1,theta1,phi1,theta2,phi2; remainder has slope 1). */
int be_params2dists_bwd_f32(const be_render_opts* opts_host, const float* params8, const float* gdists, float* gparams8, int64_t n,
                            void* stream);
/* dists2indicators: gwedges [N,3,21,21] -> gdists [N,2,21,21], getas [N,2]. */
int be_dists2indicators_bwd_f32(const float* dists, const float* etas, const float* gwedges, float* gdists, float* getas, int64_t n,
                                void* stream);
/* inverse_3by3: inv = the forward's output, gout [n,3,3] -> ga = -inv^T gout inv^T. */
int be_inverse3x3_bwd_f32(const float* inv, const float* gout, float* ga, int64_t n, void* stream);
/* get_image_derivative: img [planes,H,W], gout [planes,H-2,W-2] -> gimg [planes,H,W]. */
int be_image_derivative_bwd_f32(const float* img, const float* gout, float* gimg, int64_t planes, int H, int W, void* stream);
/* params2etas: gp = geta * d eta / d p. */
int be_params2etas_bwd_f32(const float* p, const float* geta, float* gp, int64_t n, void* stream);
/* normalized_gaussian (utils/postprocessing_loss.py:97-98): y = exp(-x^2 / delta_sq), and its adjoint. */
int be_normalized_gaussian_f32(const float* x, float* y, float delta_sq, int64_t n, void* stream);
int be_normalized_gaussian_bwd_f32(const float* x, const float* gy, float* gx, float delta_sq, int64_t n, void* stream);
/* DepthEtas.etas2depth / depth2sigma (utils/depth_etas.py:23-37): the branch taken is the forward's. */
int be_etas2depth_bwd_f32(const be_depth_consts* consts_host, const float* eta1, const float* eta2, const float* gdepth,
                          float* geta1, float* geta2, int64_t n, void* stream);
int be_depth2sigma_bwd_f32(const be_depth_consts* consts_host, const float* depth, float rho_prime, const float* geta, float* gdepth,
                           int64_t n, void* stream);
/* be_fold_patches_f32, modes 0 (sum) and 1 (mean): gout [B,C,H,W] -> gsrc in the source's strided layout. */
int be_fold_patches_bwd_f32(const float* gout, float* gsrc, int B, int C, int hp, int wp, int H, int W, int stride, int64_t s_b,
                            int64_t s_c, int64_t s_r, int64_t s_col, int64_t s_pi, int64_t s_pj, int mode, void* stream);
/* est[:, col0:col1] <- remainder(est[:, col0:col1], 2 pi) IN PLACE, est [n,ld] (LocalLoss.get_patches, local_training.py:33). */
int be_wrap_angles_inplace_f32(float* est, int64_t n, int ld, int col0, int col1, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * LocalLoss forward + backward (training)
 *   replaces LocalLoss.get_patches + LocalLoss.forward (local_training.py:32-52) and the autograd graph under
 *   them.  est [B,10] raw CNN output (angles are wrapped inside, :33), img_fit / gt [B,21,21,3] channels-last
 *   (the dataset layout; local_training.py:105 passes the clean image as both), bdist [B,21,21],
 *   deri [B,19,19,3].  partial [B,3] = per-patch sums of the three terms:
 *     loss = sum(partial[:,0])/(B*441) + beta_bndry*sum(partial[:,1])/(B*441) + beta_smooth*sum(partial[:,2])/(B*361)
 *   grad_est [B,10] = d loss / d est (NULL to skip); patches [B,3,21,21], boundary [B,21,21] optional.
 * ------------------------------------------------------------------------------------------------- */
int be_local_loss_f32(const be_render_opts* opts_host, const float* est, const float* img_fit, const float* gt,
                      const float* bdist, const float* deri, float beta_bndry, float beta_smooth, float* partial,
                      float* grad_est, float* patches, float* boundary, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * GlobalLoss forward + backward (global-stage training)
 *   replaces GlobalLoss.get_patches + get_loss (global_training.py:69-139) and the autograd graph under them.
 *   est [B,P,12] raw GlobalStage output; img_fit / img_gt [B,2,H,W,3] channels-last (dataset layout);
 *   G [B,2,3,H,W], Gderi [B,2,3,H-2,W-2], Gbndry [B,H,W]: the CURRENT folded image, its Sobel magnitude and the
 *   folded boundary map (the reference detaches them, :94,:100,:106; produce them with be_render_full_f32 +
 *   be_fold_records_f32 + be_image_derivative_f32); bdist [B,H,W]; deri [B,2,H-2,W-2,3]; bdepth [B,H,W];
 *   gamma6 (host) = gamma_{color, color_cons, bndry_cons, smthns, smthns_cons, bndry_loc}.
 *   partial [B*P,8] = per-patch sums of the six mean terms, the depth numerator and the depth-mask count;
 *   grad [B*P,12] = gradient of  sum_k gamma_k * mean_k  w.r.t. est;  grad_depth [B*P,4] = gradient of the depth
 *   NUMERATOR w.r.t. est[:,8:12] (the caller adds gamma_depth / total_mask_count * grad_depth).
 * ------------------------------------------------------------------------------------------------- */
int be_global_loss_f32(const be_render_opts* opts_host, const be_depth_consts* consts_host, const float* est,
                       const float* img_fit, const float* img_gt, const float* G, const float* Gderi,
                       const float* Gbndry, const float* bdist, const float* deri, const float* bdepth,
                       const float* gamma6_host, float* partial, float* grad, float* grad_depth, int B, int hp, int wp,
                       int H, int W, int stride, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * LocalStage CNN  (models/local_stage.py:30-73), inference (BatchNorm folded into the convs)
 * ------------------------------------------------------------------------------------------------- */

/* Kernel-layout weights.  Sizes in floats: be_local_stage_packed_floats(). One contiguous device buffer. */
size_t be_local_stage_packed_floats(void);

/* The reference state-dict (100 entries) as raw device pointers, in state_dict() order,
 * skipping the int64 num_batches_tracked entries: for each of the 13 conv+BN pairs
 *   {conv.weight [Cout,Cin,k,k], conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var}
 * then fc.1.{weight [1024,2304],bias}, fc.2.{weight,bias,running_mean,running_var}, fc.4.{weight [10,1024],bias}
 * = 13*6 + 2 + 4 + 2 = 86 pointers. */
#define BE_LOCAL_STAGE_NTENSORS 86
int be_local_stage_pack_f32(const float* const* tensors_host /* [86] device ptrs */, float bn_eps,
                            float* packed, void* stream);

/* Per-call options of the forward (NULL = the defaults).  They travel with the call - no process-wide state, so two models
 * with different settings, or two host threads, do not interfere.
 *   winograd  1 (default): the 3x3 convolutions on the 6x6 maps (layers 1-3) run in Winograd form (be_wino_tile_rows()); 0: direct
 *             implicit GEMM (the form the split-bf16 experiment and A/B runs use).  Both read the same packed buffer.
 *   chunk     the batch is walked in sub-batches of this many patches (0 = default 8192) so that the workspace stays bounded
 *             whatever N is; the same value must be given to be_local_stage_workspace_bytes(). */
typedef struct be_local_stage_opts {
    int winograd;
    int chunk;
} be_local_stage_opts;

/* Workspace (activations) for a batch of n patches walked in sub-batches of `chunk` (0 = default), in bytes. */
size_t be_local_stage_workspace_bytes(int64_t n, int chunk);

/* LocalStage.forward in eval mode: x [N,3,21,21] (NCHW as the reference feeds it) -> out [N,10]. */
int be_local_stage_forward_f32(const float* packed, const float* x, float* out, int64_t n,
                               void* workspace, size_t workspace_bytes, const be_local_stage_opts* opts, void* stream);

/* The same forward with the 21x21 windows gathered from a view (e.g. straight from the image pair [2,3,H,W]) in
 * place of nn.Unfold + permute (blurry_edges_test.py:120-121); patch numbering as be_render_colors_view_f32. */
int be_local_stage_forward_view_f32(const float* packed, const be_patch_view* view_host, int64_t patches_per_image,
                                    float* out, int64_t n, void* workspace, size_t workspace_bytes,
                                    const be_local_stage_opts* opts, void* stream);

/* Layer-level entry points (used by the layer-by-layer parity tests and by the training path).
 * Activations are NHWC.  act: 0 none, 1 Smish (models/local_stage.py:4-6), 2 ReLU (GlobalStage FFN). */
typedef struct be_conv_desc {
    int n, h, w;          /* batch, spatial size (stride 1, "same" padding)      */
    int cin, cout;        /* cin % 32 == 0 (conv1: cin = 4, NHWC4 input)         */
    int ksize;            /* 1, 3, or 7 (7: the conv1 row-gather mode: cin = 4 = the staging's channels, of which the fourth is
                           * zero padding - weights are packed from cin <= 3)  */
    int act;
} be_conv_desc;
size_t be_conv_packed_floats(int cout, int cin, int ksize);
/* Fold conv bias + eval-mode BatchNorm into packed weights/bias.  bn_* may be NULL (plain conv/linear).
 * layout_chw_hw: 0, or H*W (9) when the input features were flattened from (C,H,W) (fc.1, :45-46). */
int be_conv_pack_f32(const float* weight_oihw, const float* bias, const float* bn_gamma, const float* bn_beta,
                     const float* bn_mean, const float* bn_var, float bn_eps, int cout, int cin, int ksize,
                     int layout_chw_hw, float* packed_w, float* packed_bias, void* stream);
/* A table of pack jobs in ONE launch: job i = be_conv_pack_f32 (dgrad = 0; BatchNorm pointers may be NULL) or
 * be_conv_pack_dgrad_f32 (dgrad = 1: bias / BatchNorm pointers ignored) with the same argument meaning and the same
 * preconditions (the caller checks them: the table lives in DEVICE memory and is not inspected on the host).  A training step
 * re-packs every layer for the forward and the data-gradient convolution (local_training.py:103-106 under autograd). */
typedef struct be_pack_job {
    const float *weight, *bias, *bn_gamma, *bn_beta, *bn_mean, *bn_var;
    float* packed_w;
    float* packed_bias;
    float bn_eps;
    int cout, cin, ksize, layout_chw_hw, dgrad;
} be_pack_job;
int be_conv_pack_jobs_f32(const be_pack_job* jobs_device, int njobs, void* stream);
/* y[n,h,w,cout] = act(conv(x) + bias (+ residual)); residual may be NULL; ldy = row stride of y in floats. */
int be_conv_nhwc_f32(const be_conv_desc* desc_host, const float* x, const float* packed_w,
                     const float* packed_bias, const float* residual, float* y, int ldy, void* stream);
/* The same convolution with scratch lent by the caller: launches with few output tiles (batch-64 training: M = 2304
 * rows) split their K loop over up to 8 slices (partial sums in scratch, summed in a fixed order with the bias /
 * residual / activation applied by a second small kernel); large launches behave exactly as be_conv_nhwc_f32.
 * scratch: 8 * M * cout_pad32 * 4 bytes is always enough (fewer slices are used when it is smaller). */
int be_conv_nhwc_splitk_f32(const be_conv_desc* d, const float* x, const float* packed_w, const float* packed_bias,
                            const float* residual, float* y, int ldy, void* scratch, size_t scratch_bytes, void* stream);
/* ResidualBlock tail in ONE launch (models/local_stage.py:22-27): y = act(conv_kxk(x) + conv_1x1(x2) + bias) where the
 * second branch is the block's downsample; its 1x1 conv is appended to the K loop of the first (no residual tensor
 * in HBM).  Weights: both branches (each with its own folded BatchNorm) packed side by side, biases summed. */
size_t be_conv_fused2_packed_floats(int cout, int cin, int ksize, int cin2);
int be_conv_pack_fused2_f32(const float* weight_oihw, const float* bias, const float* bn_gamma, const float* bn_beta,
                            const float* bn_mean, const float* bn_var, const float* weight2_oi, const float* bias2,
                            const float* bn2_gamma, const float* bn2_beta, const float* bn2_mean, const float* bn2_var,
                            float bn_eps, int cout, int cin, int ksize, int cin2, float* packed_w, float* packed_bias,
                            void* stream);
int be_conv_nhwc_fused2_f32(const be_conv_desc* desc_host, const float* x, const float* x2, int cin2,
                            const float* packed_w, const float* packed_bias, float* y, int ldy, void* stream);
/* nbatch independent problems of the same shape in one launch (grid.z): batch b reads x + b*x_stride, packed_w +
 * b*w_stride and writes y + b*y_stride (strides in floats, multiples of 4).  packed_bias may be NULL (no bias) - here
 * and in be_conv_nhwc_f32.  Used for the per-position GEMMs of the Winograd convolutions below. */
int be_conv_nhwc_batched_f32(const be_conv_desc* d, const float* x, const float* packed_w, const float* packed_bias, float* y,
                             int ldy, int nbatch, int64_t x_stride, int64_t w_stride, int64_t y_stride, void* stream);

/* Winograd form of the 3x3 'same' convolutions on 6x6 maps (LocalStage layers 1-3): exact fp32 products and accumulation with
 * fewer multiplies than the direct form.  Tile shape (compile-time, be_wino_tile_rows()): 6 = F(6,3) along the rows x F(3,3) along
 * the columns - two 8x5 tiles per map, 40 transform positions, 80 multiplies per map and channel pair instead of 324 (the default
 * since round 4); 3 = F(3,3) x F(3,3) - four 5x5 tiles, 25 positions, 100 multiplies (rounds 1-3).  be_wino_pack_f32 folds an
 * optional eval BatchNorm like be_conv_pack_f32 and writes U [positions][cout_pad32][cin] + bias [cout_pad32];
 * be_wino_conv3x3_6x6_f32 runs input transform, `positions` batched GEMMs and output transform (+ bias, residual, activation).
 * workspace: be_wino_workspace_floats(n, cin, cout) floats.  cin %% 32 == 0, cout %% 4 == 0. */
int be_wino_tile_rows(void);      /* 6 or 3: output rows per Winograd tile; positions = 5 (rows + 2), tiles per map = 12 / rows */
size_t be_wino_packed_floats(int cout, int cin);
int be_wino_pack_f32(const float* w_oihw, const float* bias, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                     const float* bn_var, float bn_eps, int cout, int cin, float* packed_w, float* packed_bias, void* stream);
size_t be_wino_workspace_floats(int64_t n, int cin, int cout);
int be_wino_conv3x3_6x6_f32(const float* x, const float* packed_w, const float* packed_bias, const float* residual, float* y,
                            int64_t n, int cin, int cout, int act, float* workspace, size_t workspace_floats, void* stream);

/* Two chained 3x3 convolutions (conv1 -> act1 -> conv2 -> + residual -> act2: a residual block of LocalStage) with the
 * intermediate 6x6 map kept in registers between conv1's output transform and conv2's input transform. */
size_t be_wino_pair_workspace_floats(int64_t n, int cin, int cmid, int cout);
int be_wino_conv3x3_pair_6x6_f32(const float* x, const float* packed_w1, const float* packed_bias1, int act1,
                                 const float* packed_w2, const float* packed_bias2, const float* residual, int act2, float* y,
                                 int64_t n, int cin, int cmid, int cout, float* workspace, size_t workspace_floats, void* stream);

/* nn.MaxPool2d(k, stride, pad) on NHWC (models/local_stage.py:42-43). */
int be_maxpool_nhwc_f32(const float* x, float* y, int n, int h, int w, int c, int k, int stride, int pad,
                        void* stream);
/* The same pooling reading an input whose rows are ldx floats apart (a channel slice of a wider NHWC tensor). */
int be_maxpool_nhwc_ld_f32(const float* x, int ldx, float* y, int n, int h, int w, int c, int k, int stride, int pad,
                           void* stream);
/* [N,3,21,21] NCHW -> [N,21,21,4] NHWC with a zero 4th channel (input staging of conv1). */
int be_nchw3_to_nhwc4_f32(const float* x, float* y, int64_t n, int hw, void* stream);
/* The padded form of that staging for large batches: [N,h,wrow,4], image column c at padded column c + 3, zeros in the
 * other columns (wrow >= w + 7; LocalStage uses 28), so that the 8-pixel kernel rows of conv1 stay inside a row and the
 * pixel-major LDS-DMA kernel needs no zero-fill (be_conv_pm.hip).  models/local_stage.py:34-37. */
int be_nchw3_to_nhwc4p_f32(const float* x, float* y, int64_t n, int h, int w, int wrow, void* stream);
int be_view_to_nhwc4p_f32(const be_patch_view* view_host, int64_t patches_per_image, int64_t first, float* y, int64_t n,
                          int wrow, void* stream);
/* conv1 (7x7, pad 3, cin 4 = rgb + zero, cout 64) + folded BatchNorm + activation on that padded staging; weights as packed
 * by be_conv_pack_f32(ksize 7).  Bit-identical to be_conv_nhwc_f32 on the unpadded staging. */
int be_conv7x7_nhwc4p_f32(const be_conv_desc* d, const float* x, int wrow, const float* packed_w, const float* packed_bias,
                          float* y, int ldy, void* stream);
/* conv1 + folded BatchNorm + Smish + MaxPool2d(3, 2, 1) of LocalStage's head (models/local_stage.py:34-37,42,64-65) in ONE launch,
 * image-major: x4p [n,21,28,4] (the padded staging of be_nchw3_to_nhwc4p_f32 / be_view_to_nhwc4p_f32, wrow = 28), packed_w / bias =
 * the 7x7 pack of be_conv_pack_f32 (cout 64, K = 224), y [n,11,11,64].  A workgroup keeps whole images in LDS, the weights in
 * registers, and writes only the pooled map.  Bit-identical to be_conv7x7_nhwc4p_f32 (act 1) followed by be_maxpool_nhwc_f32(3,2,1). */
int be_conv7x7_pool_nhwc4p_f32(const float* x4p, int64_t n, const float* packed_w, const float* packed_bias, float* y, void* stream);
/* Patches [first, first+n) of a view -> [n,21,21,4] (the staging be_local_stage_forward_view_f32 uses). */
int be_view_to_nhwc4_f32(const be_patch_view* view_host, int64_t patches_per_image, int64_t first, float* y,
                         int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * LocalStage training kernels (models/local_stage.py under autograd; local_training.py:103-108)
 *   Activations are NHWC matrices [M = N*H*W][C].  scratch: be_train_scratch_bytes() bytes, 16-byte aligned,
 *   reused by every call (stream-ordered).  All reductions are two-stage in a fixed order (no float atomics).
 * ------------------------------------------------------------------------------------------------- */
size_t be_train_scratch_bytes(void);
/* nn.BatchNorm2d/1d in train mode (+ optional residual add, + optional Smish): batch mean / biased variance per
 * channel, running stats updated with `momentum` and the unbiased variance (run_* may be NULL).
 * out = act((y-mean)*invstd*gamma + beta (+res)); s_in (may be NULL) receives the Smish input for the backward. */
int be_bn_train_fwd_f32(const float* y, const float* gamma, const float* beta, const float* res, float eps,
                        float momentum, float* run_mean, float* run_var, float* mean, float* invstd, float* s_in,
                        float* out, int M, int C, int act, void* scratch, size_t scratch_bytes, void* stream);
/* Backward of the above: ds = dout * smish'(s_in) (s_in NULL: ds = dout), dgamma = sum ds*xhat, dbeta = sum ds,
 * dy = gamma*invstd*(ds - dbeta/M - xhat*dgamma/M).  ds may alias dout; ds is also the gradient of `res`. */
int be_bn_train_bwd_f32(const float* dout, const float* s_in, const float* y, const float* mean, const float* invstd,
                        const float* gamma, float* ds, float* dy, float* dgamma, float* dbeta, int M, int C,
                        void* scratch, size_t scratch_bytes, void* stream);
/* out[c] = sum_r a[r][c] (conv / linear bias gradients). */
int be_col_sum_f32(const float* a, float* out, int M, int C, void* scratch, size_t scratch_bytes, void* stream);
/* Backward of nn.MaxPool2d on NHWC: the gradient goes to the first maximum of each window (PyTorch's rule). */
int be_maxpool_nhwc_bwd_f32(const float* x, const float* dout, float* dx, int n, int h, int w, int c, int k, int stride,
                            int pad, void* stream);
/* Weight gradient of a stride-1 "same" conv / a Linear (ksize 1 on a 1x1 image), fp32 MFMA, written in the
 * reference's parameter layout dw [cout][cin][k][k] (ksize 7: x is the NHWC4 staging, dw is [cout][3][7][7];
 * layout_chw_hw as in be_conv_pack_f32). */
int be_conv_wgrad_f32(const float* x, const float* dy, float* dw, int n, int h, int w, int cin, int cout, int ksize,
                      int layout_chw_hw, void* scratch, size_t scratch_bytes, void* stream);
/* Data gradient = be_conv_nhwc_f32 with the transposed, tap-mirrored weights packed by this function
 * (dx has `cin` channels; run the conv with desc.cin = cout, desc.cout = cin, act 0).  packed_bias: cin_pad32 zeros. */
size_t be_conv_dgrad_packed_floats(int cout, int cin, int ksize);
int be_conv_pack_dgrad_f32(const float* weight_oihw, int cout, int cin, int ksize, int layout_chw_hw, float* packed_w,
                           float* packed_bias, void* stream);
/* Last Linear (K -> J, J small): dx [M,K], dw [J,K], db [J] from x [M,K], w [J,K], dy [M,J]. */
int be_linear_small_bwd_f32(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int M,
                            int K, int J, void* stream);

/* One training UNIT of LocalStage = nn.Conv2d / nn.Linear -> nn.BatchNorm (batch statistics) [-> + residual] [-> Smish]
 * (models/local_stage.py:11-17, 20-28, 44-50 in train mode; the forward half of local_training.py:103) in three launches:
 * the convolution (desc->act is ignored; its K loop may be split), one kernel that sums the split-K slices with the bias
 * into y [M,cout] and forms the per-row-block column sums of y and y^2, and one that finishes the batch statistics in its
 * prologue (fixed order), updates run_mean / run_var (momentum, unbiased variance; may be NULL), stores mean / invstd [cout]
 * and writes out = act((y-mean)*invstd*gamma + beta (+res)); s_in (may be NULL) keeps the Smish input for the backward.
 * packed_w / packed_bias: be_conv_pack_f32 WITHOUT BatchNorm (the conv bias alone).  cout %% 32 == 0, cout <= 1024.
 * scratch: be_train_scratch_bytes() bytes. */
int be_train_unit_fwd_f32(const be_conv_desc* desc_host, const float* x, const float* packed_w, const float* packed_bias,
                          const float* gamma, const float* beta, const float* res, float eps, float momentum, float* run_mean,
                          float* run_var, float* y, float* mean, float* invstd, float* s_in, float* out, int act,
                          void* scratch, size_t scratch_bytes, void* stream);
/* Backward of that unit (the autograd graph under loss.backward(), local_training.py:106) in four launches (the weight-gradient
 * GEMM and the data-gradient convolution share one grid): ds = dout *
 * smish'(s_in) (s_in NULL: ds = dout; ds is also the residual's gradient) with its column sums; dgamma / dbeta and
 * dy = gamma*invstd*(ds - dbeta/M - xhat*dgamma/M) with the column sums of dy; the weight-gradient GEMM over x (desc = the
 * FORWARD shape; ksize 7: x is the NHWC4 staging and dw is [cout][3][7][7]; layout_chw_hw as in be_conv_pack_f32); the
 * data-gradient convolution through dgrad_packed_w / dgrad_packed_bias (be_conv_pack_dgrad_f32; both NULL with dx NULL when
 * no input gradient is wanted); and one kernel that sums the weight-gradient slices into dw (reference layout), the bias
 * gradient into db and the data-gradient slices (+ dx_add [M,cin] if not NULL: the other branch of a residual block) into dx. */
int be_train_unit_bwd_f32(const be_conv_desc* desc_host, const float* x, const float* dout, const float* s_in, const float* y,
                          const float* mean, const float* invstd, const float* gamma, const float* dgrad_packed_w,
                          const float* dgrad_packed_bias, const float* dx_add, int layout_chw_hw, float* ds, float* dy,
                          float* dgamma, float* dbeta, float* dw, float* db, float* dx, void* scratch, size_t scratch_bytes,
                          void* stream);
/* The same two entry points for TWO units that are independent of each other and share their input - the 3x3 convolution
 * (unit a) and the 1x1 downsample (unit b) of a ResidualBlock (models/local_stage.py:20-28): every launch of the single form
 * carries both units' workgroups (the two convolutions in one grid, the BatchNorm kernels with one grid slice per unit, one
 * closing kernel), each unit in its own half of the scratch regions.  Results are those of the single calls, bit for bit.
 * Forward: both units produce the same [n,h,w,cout].  Backward: the units also share x and cin; a->dx receives the SUM of
 * both input gradients (what ResidualBlock's autograd node accumulates), b->dx [M,cin] is working space (contents
 * unspecified afterwards), dx_add must be NULL in both.  Shapes the merged launches do not take run one unit after the
 * other inside the call. */
typedef struct be_train_unit_fwd {
    be_conv_desc desc;
    const float *x, *packed_w, *packed_bias, *gamma, *beta, *res;
    float *run_mean, *run_var, *y, *mean, *invstd, *s_in, *out;
    int act;
} be_train_unit_fwd;
typedef struct be_train_unit_bwd {
    be_conv_desc desc;
    const float *x, *dout, *s_in, *y, *mean, *invstd, *gamma, *dgrad_packed_w, *dgrad_packed_bias, *dx_add;
    int layout_chw_hw;
    float *ds, *dy, *dgamma, *dbeta, *dw, *db, *dx;
} be_train_unit_bwd;
int be_train_unit_pair_fwd_f32(const be_train_unit_fwd* a, const be_train_unit_fwd* b, float eps, float momentum,
                               void* scratch, size_t scratch_bytes, void* stream);
int be_train_unit_pair_bwd_f32(const be_train_unit_bwd* a, const be_train_unit_bwd* b, void* scratch, size_t scratch_bytes,
                               void* stream);
/* Round 6: the backward of the units with channel counts that are multiples of 128 on maps of at most 11 x 11 (layers 1-3 of
 * models/local_stage.py:38-41 at batch 64 k, local_training.py:103-106) runs its weight-gradient GEMM and data-gradient
 * convolution as ONE persistent, balanced launch (csrc/be_train_sk.h: every problem lays its tiles end to end on an axis of K
 * chunks, workgroup g takes the chunks [g Q, (g + 1) Q)); BE_NO_TRAIN_SK=1 in the environment selects rounds 3-5's equal-slice
 * launch instead.  This call is the HOST-ONLY inspection hook of that arithmetic (no GPU, no launch): the plan for one unit
 * [n,h,w,cin] -> cout, ksize 1 | 3, with `workgroups` slots of which the fraction w_share goes to the weight gradient.  Per
 * segment six ints are written to seg (at most cap segments): problem (0 weight gradient, 1 convolution), tile, first chunk,
 * last chunk + 1, slice, slices of that tile.  Returns the segment count, < 0 on bad arguments. */
int be_train_sk_plan_debug(int n, int h, int w, int cin, int cout, int ksize, int workgroups, double w_share, int* seg, int cap);

/* Parameter gradients of y = x W^T + b over many rows (the linears of GlobalStage in training, global_training.py:207-213 under
 * autograd): dw [cout,cin] = dy^T x and db [cout] = column sums of dy, in two launches (weight-gradient tiles + column-sum
 * workgroups in one grid, then one reduction kernel; fixed order).  cin, cout multiples of 128, M >= 256 rows. */
int be_linear_param_grads_f32(const float* x, const float* dy, float* dw, float* db, int M, int cin, int cout, void* scratch,
                              size_t scratch_bytes, void* stream);
/* Last Linear forward (K -> J, J small; models/local_stage.py:50): y [M,J] = x [M,K] w [J,K]^T + b, raw parameter layout. */
int be_linear_small_fwd_f32(const float* x, const float* w, const float* b, float* y, int M, int K, int J, void* stream);
/* nn.MaxPool2d forward on NHWC that also records the winning window element (dy*k + dx of the FIRST maximum, one byte per
 * output value; idx 4-byte aligned, c %% 4 == 0), and the backward that routes dout through those bytes. */
int be_maxpool_nhwc_fwd_idx_f32(const float* x, float* y, unsigned char* idx, int n, int h, int w, int c, int k, int stride,
                                int pad, void* stream);
int be_maxpool_nhwc_bwd_idx_f32(const unsigned char* idx, const float* dout, float* dx, int n, int h, int w, int c, int k,
                                int stride, int pad, void* stream);

/* Tail of a training step (local_training.py:107-108): torch.nn.utils.clip_grad_norm_(max_norm, 2) + torch.optim.AdamW.step()
 * for parameters whose gradients are slices of ONE flat buffer (what the LocalStage backward writes), in two launches: squared-
 * norm partials (fp64; this launch also counts the step), the update.  table (device): one entry per workgroup, at most be_adam_chunk() elements
 * each: parameter / exp_avg / exp_avg_sq pointers, offset of the slice in grad_flat, count.  grad_scale multiplies the gradient
 * first (1/world after a summing all-reduce).  max_norm <= 0: no clipping.  step_device: float scalar = steps taken so far
 * (incremented here); grad_norm_out (may be NULL): the norm before clipping.  write_back: store the scaled + clipped gradient
 * back (clip_grad_norm_ modifies .grad in place).  partial: npartial_cap >= ceil(n_flat / be_adam_chunk()) doubles. */
typedef struct be_adam_entry {
    float *p, *m, *v;
    int64_t goff;
    int n;
} be_adam_entry;
int be_adam_chunk(void);
int be_clip_adamw_f32(const be_adam_entry* table_device, int nentries, float* grad_flat, int64_t n_flat, double* partial,
                      int npartial_cap, float max_norm, float grad_scale, double lr, double beta1, double beta2, double eps,
                      double weight_decay, float* step_device, float* grad_norm_out, int write_back, void* stream);
/* Final reduction of be_local_loss_f32's per-patch partials [B,3] into the scalar of LocalLoss.forward (local_training.py:47-52):
 * loss = S0/(441 B) + beta_bndry S1/(441 B) + beta_smooth S2/(361 B), sums in fp64 in patch order. */
int be_local_loss_finish_f32(const float* partial, int B, float beta_bndry, float beta_smooth, float* loss_out, void* stream);

/* eval_depth (utils/metrics.py:3-20) on the device: pred, gt [B,H,W]; a pixel counts when mask_src > 0 (the scripts pass
 * the estimated depth map itself, blurry_edges_test.py:148); crop pixels dropped on every side; out5 (device, float64) =
 * delta1, delta2, delta3, RMSE in cm, AbsRel in cm, summed jointly over the batch as the reference does. */
int be_eval_depth_f32(const float* pred, const float* gt, const float* mask_src, int B, int H, int W, int crop, float tau_n,
                      float z_min, float z_max, double* out5, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Depth-completion U-Net glue ('--densify pp', models/depth_completion_unet.py:79-113; blurry_edges_test.py:141-142).
 * Its convolutions are be_conv_nhwc_f32 launches (3x3 + folded BatchNorm + ReLU; ConvTranspose2d(k=2,s=2) as a 1x1
 * convolution with 4*cout outputs ordered (dy,dx,co)); these two do the layout work around them.
 * ------------------------------------------------------------------------------------------------- */
/* x [N,C,HW] NCHW -> y [N,HW,cpad] NHWC, channels C..cpad-1 zero. */
int be_nchw_to_nhwc_pad_f32(const float* x, float* y, int64_t n, int c, int64_t hw, int cpad, void* stream);
/* Pixel shuffle of the transposed convolution into the decoder's concatenated input: t [N,h,w,4*cout] ->
 * y[n, 2i+dy+top, 2j+dx+left, ch_off+co] of an [N,oh,ow,ldy] tensor (top/left = F.pad offsets diff//2, :63-65;
 * positions outside are dropped; the caller zero-fills the padding border once). */
int be_upconv2x2_scatter_f32(const float* t, float* y, int64_t n, int h, int w, int cout, int oh, int ow, int top, int left,
                             int ldy, int ch_off, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * GlobalStage encoder pieces, inference (nn.TransformerEncoderLayer of models/global_stage.py:28-32; the linears
 * run on be_conv_nhwc_f32 as 1x1 convs)
 * ------------------------------------------------------------------------------------------------- */
/* Multi-head self-attention with head dim 16: qkv [B*L, 3*H*16] (rows = tokens, columns q|k|v as
 * nn.MultiheadAttention's in_proj produces them) -> out [B*L, H*16] = softmax(q k^T / 4) v per head.
 * workspace: be_attention_workspace_floats(B,L,H) floats.  L % 128 == 0; l_valid in (L - 128, L]: the number of real
 * tokens when the caller padded the sequence up to L (keys >= l_valid get probability exactly 0; output rows >= l_valid
 * are computed and meaningless).  nn.TransformerEncoder takes any L <= max_len^2 (models/global_stage.py:18-20). */
size_t be_attention_workspace_floats(int B, int L, int H);
int be_attention_f32(const float* qkv, float* out, float* workspace, int B, int L, int l_valid, int H, void* stream);
/* y = LayerNorm(x + res) over the last dimension D in {64,128,192,256}; res may be NULL; y may alias x. */
int be_add_layernorm_f32(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                         int64_t rows, int D, float eps, void* stream);
/* x[b] += pe for b < batches (PositionalEncoding.forward, models/global_stage.py:18-20); per_batch = L*D floats. */
int be_add_pe_f32(float* x, const float* pe, int64_t batches, int64_t per_batch, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * GlobalStage encoder, training (model.train() of global_training.py:207-213: dropout 0.1 at the four sites of
 * nn.TransformerEncoderLayer, autograd through every layer).  Dropout is counter-based: element i of site `site` is
 * kept iff hash(i, seed, site) >= p * 2^32, so the backward kernels re-derive the masks and nothing is stored.
 * ------------------------------------------------------------------------------------------------- */
/* Attention forward for training: out as be_attention_f32 with dropout on the probabilities (site = batch*H + head,
 * element = query*L + key), lse [B*H, L] = log2-sum-exp of each score row (saved for the backward).  The workspace keeps, for
 * the backward call of the same layer, the split q / k / v and one keep bit per probability (B*H*L*L/8 bytes, from float
 * offset be_attention_train_keep_offset_floats on: [B*H][L/32 query tiles][L/32 key blocks][64 lanes] halfwords). */
size_t be_attention_train_workspace_floats(int B, int L, int H);
int be_attention_train_fwd_f32(const float* qkv, float* out, float* lse, float* workspace, int B, int L, int l_valid, int H,
                               float dropout_p, uint32_t seed, void* stream);
/* Attention backward: dout [B*L, H*16] -> dqkv [B*L, 3*H*16]; probabilities are recomputed from qkv and lse
 * (no [L,L] float tensor is ever stored); deterministic (no atomics).  Same workspace size as the forward; `scratch`
 * (be_attention_bwd_scratch_floats floats, contents irrelevant before and after the call: one buffer can serve every layer)
 * receives the partial dQ sums of the key groups.
 * operands_ready != 0: `workspace` is the buffer the forward call of this layer used and nothing wrote to it since
 * (its split q/k/v and keep bits are reused); 0: they are produced again from qkv and the seed.  l_valid as in be_attention_f32: rows >= l_valid of
 * dout must be zero (they are when the caller slices the padded output), and dqkv comes out zero there. */
size_t be_attention_bwd_scratch_floats(int B, int L, int H);
int be_attention_bwd_f32(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv,
                         float* workspace, float* scratch, int operands_ready, int B, int L, int l_valid, int H, float dropout_p,
                         uint32_t seed, void* stream);
/* The keep mask the two functions above apply, [B*H, L, L] in {0,1} (test hook; small L only). */
int be_attention_dropout_mask_f32(float* mask, int B, int L, int H, float dropout_p, uint32_t seed, void* stream);
/* The same decisions in the packed layout the forward leaves in its workspace (B*H*L*L/16 halfwords), from the formula:
 * what be_attention_bwd_f32 regenerates when operands_ready == 0, and what the tests compare the forward's stores with. */
size_t be_attention_train_keep_offset_floats(int B, int L, int H);
int be_attention_keep_bits_u16(uint16_t* keep, int B, int L, int H, float dropout_p, uint32_t seed, void* stream);
/* y = dropout(x) = x * keep / (1-p); with gate != NULL additionally zero where gate <= 0, which makes it the
 * backward of dropout(relu(.)) given the ReLU output.  y may alias x.  n < 2^32. */
int be_dropout_f32(const float* x, const float* gate, float* y, int64_t n, float dropout_p, uint32_t seed, uint32_t site,
                   void* stream);
/* v = res + dropout(x) (stored), y = LayerNorm(v); D = 128; res may be NULL. */
int be_add_layernorm_train_f32(const float* x, const float* res, const float* gamma, const float* beta, float* v, float* y,
                               int64_t rows, int D, float eps, float dropout_p, uint32_t seed, uint32_t site, void* stream);
/* LayerNorm backward from the stored v: dv (gradient of v, hence of the residual input; NULL to skip),
 * dx = dropout'(dv) (gradient of the dropped branch; NULL to skip), partial [be_layernorm_bwd_partial_floats]:
 * per-block sums of (dy*xhat | dy) as rows of 2*D floats - be_col_sum_f32 over them gives dgamma | dbeta. */
size_t be_layernorm_bwd_partial_floats(int64_t rows, int D);
int be_layernorm_bwd_f32(const float* dy, const float* v, const float* gamma, float* dv, float* dx, float* partial,
                         int64_t rows, int D, float eps, float dropout_p, uint32_t seed, uint32_t site, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Synthetic-shape training data (train_val_data_generator.py:31-275).  float64 throughout, the
 * reference's array layouts.  shape [N,maxo,10] int32 = (kind 0 circle | 1 rectangle | 2 triangle, vertex count,
 * x0,y0..x3,y3; circle: x0,y0 centre, x1 radius), prop [N,maxo,4] = (depth, c0,c1,c2), objects far -> near,
 * nobj [N], bg [N,3], sig [N,maxo,2] PSF radius in pixels per aperture (utils/data_generator.py:16-17).
 * N < 65536, maxo <= 32.
 * ------------------------------------------------------------------------------------------------- */
/* Rasterisation (cv2.circle / cv2.drawContours of train_val_data_generator.py:58-76, thickness -1 and 1): two bit planes per
 * object, FILL and RING, masks [N][maxo][2][H][ceil(W/32)] uint32 (be_datagen_raster_words of them; bit x & 31 of word x >> 5).
 * The planes follow the algorithms those calls run in OpenCV 4.x modules/imgproc/src/drawing.cpp: Circle() (midpoint walk),
 * Line() = clipLine() + the 8-connected LineIterator from the left end point, CollectPolyEdges() + FillEdgeCollection() (16.16
 * fixed-point edges, active for y0 <= y < y1, runs from ceil(x_left) to floor(x_right)).  Objects >= nobj[i] get empty planes. */
size_t be_datagen_raster_words(int n, int H, int W, int maxo);
int be_datagen_raster_u32(const int* shape, const int* nobj, int n, int H, int W, int maxo, uint32_t* masks, void* stream);
/* aif [N,H,W,3] (colour / 255, as images_aif is stored, :137), bloc [N,H,W] (0/255), idep, bdep [N,H,W]  (:36-41,77-85,100-103);
 * masks: the planes be_datagen_raster_u32 wrote. */
int be_datagen_scene_f64(const uint32_t* masks, const double* prop, const int* nobj, const double* bg, int n, int H, int W,
                         int maxo, double z_far, double* aif, double* bloc, double* idep, double* bdep, void* stream);
/* imgs [N,2,H,W,3]: background, then every object (its FILL plane) blurred with its PSF per aperture and alpha-composited (:87-94). */
size_t be_datagen_blur_scratch_bytes(int n, int H, int W);
int be_datagen_blur_composite_f64(const uint32_t* masks, const double* prop, const int* nobj, const double* bg, const double* sig,
                                  int n, int H, int W, int maxo, int max_nobj, double* imgs, void* scratch,
                                  size_t scratch_bytes, void* stream);
/* imgs <- round(imgs); bdist [N,H,W] city-block distance to the nearest bloc > 0 pixel (ones when there is none);
 * deri [N,2,H,W,3] Sobel magnitude / 255 with reflected borders (:105-123).  scratch >= N*H*W*4 bytes. */
int be_datagen_finish_f64(double* imgs, const double* bloc, double* bdist, double* deri, int n, int H, int W, void* scratch,
                          size_t scratch_bytes, void* stream);
/* gt = imgs/255*alpha, ny = round(clip(Poisson(gt) + sigma*N(0,1), 0, alpha)) (:165-182); alpha [n], per_sample =
 * elements per alpha; counter-based streams keyed by (seed, element). */
int be_datagen_noise_f64(const double* imgs, const double* alpha, double sigma, uint32_t seed, int64_t n, int64_t per_sample,
                         double* gt, double* ny, void* stream);
/* cand [N,H,W] uint8: 1 where a boundary pixel lies within `reach` (Chebyshev) and the pixel is >= margin from every
 * image border (:214-218). */
int be_datagen_candidates_f64(const double* bloc, unsigned char* cand, int n, int H, int W, int reach, int margin, void* stream);
/* R x R crops centred on flat pixel indices pick[i] (into [N,H,W]; must respect the margin R/2) of aperture aper[i]:
 * in6 = aif [N,H,W,3], gt, ny, deri [N,2,H,W,3], idep, bdep [N,H,W]; out9 = aif, gt, ny, deri [P,R,R,3], idep, bdep,
 * bloc, bdist (in-patch city-block distance) [P,R,R], alpha [P]  (:226-252). */
int be_datagen_crop_f64(const double* const* in6, const double* bloc, const double* alpha, const int64_t* pick, const int* aper,
                        int64_t n_patch, int n, int H, int W, int R, double* const* out9, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py's roofline leg): opt-in hipEvent pair around every matrix-kernel launch and around the
 * HBM-bound kernels of the LocalStage / pass-A step, recorded on
 * the launch stream.  be_profile_enable(0) turns it off and frees the events.  Not thread-safe.
 * ------------------------------------------------------------------------------------------------- */
#define BE_KERNEL_CONV_128x128      0   /* k_conv_igemm<2,2,2,2,TAPS>: layers 1-3 + fc.1 (the dominant kernel) */
#define BE_KERNEL_CONV_128x96       1   /* layer0                                                              */
#define BE_KERNEL_CONV_128x64       2
#define BE_KERNEL_CONV_128x32       3   /* fc.4                                                                */
#define BE_KERNEL_CONV_ROW8_128x64  4   /* conv1 (7x7 row-gather)                                              */
#define BE_KERNEL_CONV_SMALL        5   /* 64x64 / 128x32 tiles for small M (training batches)                 */
#define BE_KERNEL_WINO_GEMM         6   /* k_wino_gemm_ws / k_wino_gemm: the transform-domain GEMMs (one per position) of a Winograd layer */
#define BE_KERNEL_GEMM_ROWS         7   /* k_wino_gemm as a row GEMM: 1x1 convolutions / linears, large batches */
/* HBM-bound kernels: `bytes` = the algorithmic bytes the launch has to move, `flops` = 0 */
#define BE_KERNEL_WINO_TRANSFORM    8   /* k_wino_in / k_wino_out / k_wino_out_in / k_wino_out_pool2 */
#define BE_KERNEL_MAXPOOL           9   /* k_maxpool_nhwc */
#define BE_KERNEL_RENDER_COLORS    10   /* k_render_colors (pass A: wedges + ridge colour solve) */
#define BE_KERNEL_STAGING          11   /* NCHW / image view -> NHWC4 staging of conv1's input */
#define BE_KERNEL_TRAIN_BWD_GEMMS  12   /* k_unit_gemms / k_unit_gemms_sk: training units' weight-gradient GEMMs + data-gradient convolutions (one or two units), a block's two forward convolutions: one launch */
#define BE_KERNEL_CONV1_POOL       13   /* k_conv1_pool: conv1 7x7 + Smish + the first max-pool, image-major (bytes: staging in, pooled 11x11x64 map out) */
int be_profile_enable(int max_launches);
int be_profile_reset(void);
/* Waits for the recorded events; fills up to cap records (launch order); returns the number filled (>= 0)
 * or a negative error.  flops/bytes = ALGORITHMIC work of the launch (the convolution's 2*MAC with the unpadded K,
 * zero-padding taps included as the reference counts them; in+weights+out bytes); flops_executed (may be NULL) =
 * the MFMA work actually issued (tile padding included, taps outside the image skipped). */
int be_profile_read(int* kernel_id_host, double* flops_host, double* bytes_host, double* flops_executed_host,
                    float* ms_host, int cap);

#ifdef __cplusplus
}
#endif
#endif /* BLURRY_EDGES_HIP_H */
