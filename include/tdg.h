/* tdg.h -- C ABI of lib3dgan_hip.so: the MI355X (gfx950) kernels behind the
 * 3dgan training hot path (models/gan.py + ops/layers.py of algoterranean/3dgan).
 *
 * The reference has no native boundary: every op below is a TensorFlow-1.x op that the
 * reference's Python graph builders instantiate, and `sess.run` executes inside the TF C++
 * runtime.  Each entry point cites the reference call site (file:line under /root/reference)
 * whose TF op (forward and/or the autodiff ops TF derives from it) it replaces.
 *
 * Conventions
 *  - Plain C: pointers are device pointers (HBM) owned by the caller; the library allocates
 *    nothing persistent and never synchronises.  All work is enqueued on `stream`
 *    (a hipStream_t passed as void*).
 *  - Every function returns 0 on success, a negative TDG_E* code otherwise, and never
 *    throws; `tdg_last_error()` returns a thread-local message.
 *  - Activations are NHWC with an explicit channel stride `cs` (elements) >= channels;
 *    padding channels must hold zeros.  dtype selects the storage/compute type of
 *    activations and packed filters: TDG_F32 (exact f32 MFMA, parity path) or TDG_BF16
 *    (bf16 MFMA, f32 accumulate).  Master weights, gradients, optimizer state, biases,
 *    BN parameters and all reductions are always f32.
 */
#ifndef TDG_H_
#define TDG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDG_OK 0
#define TDG_EINVAL (-1)      /* invalid descriptor / argument            */
#define TDG_EUNSUPPORTED (-2)/* valid but not implemented for this shape */
#define TDG_EHIP (-3)        /* a HIP runtime call failed                */
#define TDG_EWORKSPACE (-4)  /* workspace too small                      */

#define TDG_F32 0
#define TDG_BF16 1

/* activations (fused epilogues / elementwise) */
#define TDG_ACT_NONE 0
#define TDG_ACT_RELU 1    /* models/gan.py:245 tf.nn.relu     */
#define TDG_ACT_LRELU 2   /* ops/activations.py:28 max(leak*x, x) */
#define TDG_ACT_TANH 3    /* models/gan.py:252 tf.tanh        */
#define TDG_ACT_SIGMOID 4 /* models/gan.py:275 tf.nn.sigmoid  */

/* mask applied by an epilogue: out *= act'(mask_src) evaluated from the POST-activation
 * (lrelu) or PRE-activation (relu after BN) tensor stored at the same offsets as `out`. */
#define TDG_MASK_NONE 0
#define TDG_MASK_LRELU 1  /* slope 1 where mask_src > 0 else leak (TF MaximumGrad, App. A-6) */
#define TDG_MASK_RELU 2   /* 1 where mask_src > 0 else 0 */

/* One strided 2-D convolution with TF 'SAME'/'VALID' geometry between a "big" tensor
 * x [n, h, w, c] (conv input / conv2d_transpose output) and a "small" tensor
 * y [n, oh, ow, k] (conv output / conv2d_transpose input), filter master layout
 * [kh, kw, c, k] f32 (HWIO for conv2d, ops/layers.py:96; [k,k,Cout,Cin] for deconv2d,
 * ops/layers.py:135 -- same memory order with c = deconv Cout, k = deconv Cin). */
typedef struct TdgConvDesc {
  int32_t n, h, w, c, cs;   /* big side: dims, channels, channel stride   */
  int32_t oh, ow, k, ks;    /* small side                                  */
  int32_t kh, kw, stride;
  int32_t pad_t, pad_l;     /* TF pad_before             */
  int32_t dtype;            /* TDG_F32 | TDG_BF16                          */
} TdgConvDesc;

/* Fused epilogue of the two conv GEMM forms. */
typedef struct TdgEpilogue {
  const float* bias;        /* per output channel, may be NULL (tf.nn.bias_add, ops/layers.py:102,143) */
  int32_t act;              /* TDG_ACT_*                                     */
  float leak;               /* lrelu leak                                    */
  int32_t mask_mode;        /* TDG_MASK_*                                    */
  const void* mask_src;     /* tensor with the geometry of the output, dtype = desc.dtype */
  int32_t accumulate;       /* 1: out = (act(acc + bias) + out) * mask  -- sums a second gradient path (U-Net skips) */
  /* Column partials of the STORED tile, emitted by the epilogue that holds it anyway (instead of a separate pass over
   * the tensor): per workgroup row tile b,  col_partial[(b*2 + 0)*N + n] = sum over its rows,  [(b*2 + 1)*N + n] = the
   * second moment (TDG_COL_BN).  TDG_COL_SUM: sum of the stored values (after activation and mask) = the bias gradient
   * of the layer that owns the tensor (autodiff of tf.nn.bias_add, ops/layers.py:102).  TDG_COL_BN: sum (v - bias[n]) and
   * sum (v - bias[n])^2 of the stored pre-activation = the batch statistics of tf.contrib.layers.batch_norm
   * (ops/layers.py:103,144).  Only rows of images < col_images count (0: all).  The launch reports how many row
   * tiles it wrote in *col_nblk_out (host memory, written before the call returns); 0 = this launch's kernel
   * variant does not provide partials (f32 tiles, accumulating epilogues, thin layers): run the separate pass.
   * Finish with tdg_col_finalize_sum / tdg_bn_fwd_from_partials. */
  float* col_partial;
  size_t col_partial_bytes;
  int32_t col_mode;         /* TDG_COL_*                                     */
  int32_t col_images;
  int32_t* col_nblk_out;
  /* Optional scratch for split-K (caller-owned, like every buffer): a forward-type GEMM whose grid would leave most CUs
   * idle (few output rows, long K: pix2pix's 1x1 ... 8x8 bottleneck layers) is cut along K into f32 partial tiles here
   * and finished by a second kernel in a fixed order.  NULL: never split. */
  void* splitk_ws;
  size_t splitk_ws_bytes;
} TdgEpilogue;
enum { TDG_COL_NONE = 0, TDG_COL_SUM = 1, TDG_COL_BN = 2 };

const char* tdg_last_error(void);
int tdg_version(void);
/* Name of the GEMM kernel variant chosen by this thread's most recent tdg_conv2d_* call (profiling aid:
 * lets a caller attribute HIP-event timings to the kernel symbols rocprofv3 reports). */
const char* tdg_last_kernel(void);

/* Diagnostics (bench.py's roofline leg): between tdg_timing_begin() and tdg_timing_end() every conv GEMM kernel
 * launch is bracketed by HIP events on its own stream; tdg_timing_end() waits for them and returns one record per
 * launch: the kernel (as rocprofv3 names it, template arguments included), its duration and the algorithmic FLOPs
 * of the columns/rows it covered.  `count` receives the number of launches even when it exceeds `capacity`.
 * Not for use during stream capture. */
typedef struct TdgLaunchRecord {
  char kernel[64];
  double ms;
  double flops;
} TdgLaunchRecord;
int tdg_timing_begin(void);
int tdg_timing_end(TdgLaunchRecord* out, int capacity, int* count);

/* ---- filter packing: f32 master -> GEMM operand layout in desc.dtype ------------------
 * FWD form  : rows = k (small-side channels), K = (tap, c)        -> used by tdg_conv2d_fwd
 * BWD form  : per output-parity class, rows = c, K = (tap', k)    -> used by tdg_conv2d_bwd_data
 * Sizes in bytes via the *_bytes queries. */
size_t tdg_packed_filter_fwd_bytes(const TdgConvDesc* d);
size_t tdg_packed_filter_bwd_bytes(const TdgConvDesc* d);
int tdg_pack_filter_fwd(const TdgConvDesc* d, const float* w, void* packed, void* stream);
int tdg_pack_filter_bwd(const TdgConvDesc* d, const float* w, void* packed, void* stream);

/* Every filter of a network in one call (after an optimizer step the reference's variables change all at
 * once: tf.train.Optimizer.apply_gradients, models/gan.py:80-81): same results as calling the two functions
 * above per job, in as few launches as the jobs fit.  All jobs must share one dtype.  A null destination
 * skips that form. */
typedef struct TdgPackJob {
  TdgConvDesc desc;
  const float* w;          /* f32 master filter [kh][kw][c][k] */
  void* packed_fwd;
  void* packed_bwd;
} TdgPackJob;
int tdg_pack_filters(const TdgPackJob* jobs, int n_jobs, void* stream);

/* ---- conv GEMMs -------------------------------------------------------------------------
 * tdg_conv2d_fwd      : y = epi(conv2d(x, W))            tf.nn.conv2d, ops/layers.py:101;
 *                       also d(conv2d_transpose)/d(input) (autodiff of ops/layers.py:142)
 *                       and the tangent pass of the gradient penalty (models/gan.py:228)
 * tdg_conv2d_bwd_data : x = epi(conv2d_backprop_input(W, y))   autodiff of ops/layers.py:101;
 *                       also IS tf.nn.conv2d_transpose, ops/layers.py:142
 * tdg_conv2d_bwd_filter: dW (+)= conv2d_backprop_filter(x, y)  autodiff of ops/layers.py:101,142
 *                       dw = beta*dw + sum over rows; deterministic two-stage reduction.
 * `n_images` lets a call cover a leading sub-batch of the buffers (desc.n is the capacity
 * used for bounds only).  */
int tdg_conv2d_fwd(const TdgConvDesc* d, int n_images, const void* x, const void* w_packed_fwd,
                   void* y, const TdgEpilogue* epi, void* stream);
int tdg_conv2d_bwd_data(const TdgConvDesc* d, int n_images, const void* y, const void* w_packed_bwd,
                        void* x, const TdgEpilogue* epi, void* stream);
size_t tdg_conv2d_bwd_filter_workspace_bytes(const TdgConvDesc* d, int n_images);
int tdg_conv2d_bwd_filter(const TdgConvDesc* d, int n_images, const void* x, const void* y,
                          float* dw, float beta, void* workspace, size_t workspace_bytes, void* stream);
/* The same sum with the big-side rows coming from TWO tensors: images [0, n_first) from x, images [n_first, n_images)
 * from x2 (its image 0 first); y holds all n_images.  One launch instead of two plus one slab reduction less: the
 * critic's filter gradient of the iwgan D step = first-order rows (layer inputs of D(x), D(g)) + tangent-pass rows
 * (tangents of the penalty's double backward, models/gan.py:228) against one delta tensor. */
int tdg_conv2d_bwd_filter2(const TdgConvDesc* d, int n_images, const void* x, int n_first, const void* x2, const void* y,
                           float* dw, float beta, void* workspace, size_t workspace_bytes, void* stream);

/* ---- dense fc2-style row ops (tf.matmul with one output unit, ops/layers.py:57 via
 *      models/gan.py:285, and its autodiff) --------------------------------------------- */
/* out[r] = act(dot(x[r, :cols], w) + bias[0]) */
int tdg_rowdot(int dtype, const void* x, int rows, int cols, const float* w, const float* bias,
               int act, float* out, void* stream);
/* dx[r, c] = (dout[r] * w[c]) * mask(mask_src[r, c]) */
int tdg_rowouter(int dtype, const float* dout, const float* w, int rows, int cols, int mask_mode,
                 float leak, const void* mask_src, void* dx, void* stream);
/* Finish column partials written by a conv epilogue (TdgEpilogue.col_partial, nblk row tiles, c columns):
 * out[n] = beta*out[n] + sum_b partial[(b*2)*c + n], fixed summation order. */
int tdg_col_finalize_sum(const float* partial, int nblk, int c, float* out, float beta, void* stream);
/* dw[c] = beta*dw[c] + sum_r coef[r] * x[r, c]   (coef NULL -> 1); deterministic */
int tdg_colsum_weighted(int dtype, const void* x, int rows, int cols, int cs, const float* coef,
                        float* dw, float beta, void* workspace, size_t workspace_bytes, void* stream);
size_t tdg_colsum_workspace_bytes(int rows, int cols);

/* ---- batch norm, training mode, no gamma (tf.contrib.layers.batch_norm defaults,
 *      ops/layers.py:58,103,144) ------------------------------------------
 * fwd : pre = (u - mean) * rsqrt(var + eps) + beta ; h = act(pre)   (u over `rows` x c)
 *       stats[0:c] = mean, stats[c:2c] = rstd (saved for backward).  u/pre use channel stride cs, h uses
 *       h_cs (h may be a channel window of a wider concat buffer); likewise dh uses dh_cs in bwd.
 * bwd : dpre = dh * act'(pre); du = rstd * (dpre - mean(dpre) - xhat * mean(dpre*xhat));
 *       dbeta = beta_acc*dbeta + sum(dpre)                                                */
size_t tdg_bn_workspace_bytes(int rows, int c);
int tdg_bn_fwd(int dtype, const void* u, int rows, int c, int cs, const float* beta, float eps,
               int act, float leak, void* pre, void* h, int h_cs, float* stats, void* workspace,
               size_t workspace_bytes, void* stream);
/* tdg_bn_fwd over `ngroups` batches of rows_per_group rows that lie behind each other in u / pre / h, each normalised with
 * ITS OWN batch statistics (stats [ngroups][2][c]; workspace >= ngroups * tdg_bn_workspace_bytes): the generator passes of
 * the n_disc_train critic runs of one iteration (models/gan.py:150-155,169-173 -- `sess.run(d_train_op)` n_disc_train times,
 * each drawing its own z through the same generator variables) taken as ONE pass; batch_norm (ops/layers.py:103,144)
 * normalises per run, i.e. per group.  pre may be null (no backward pass follows). */
int tdg_bn_fwd_groups(int dtype, const void* u, int rows_per_group, int ngroups, int c, int cs, const float* beta,
                      float eps, int act, float leak, void* pre, void* h, int h_cs, float* stats, void* workspace,
                      size_t workspace_bytes, void* stream);
/* ---- instance norm of the gen-2 layer surface (hem/ops/images.py:73-89; `use_instance_norm=True`,
 *      hem/ops/layers.py:123,200): per (image, channel) moments over the hw positions, biased variance, eps 1e-3,
 *      learned per-channel scale (init 1) and shift (init 0); fused with the layer's activation.
 *      stats [n][2][c] = (mu, rstd) are kept for the backward. */
int tdg_instance_norm_fwd(int dtype, const void* u, int n, int hw, int c, int cs, const float* scale, const float* shift,
                          float eps, int act, float leak, void* h, int h_cs, float* stats, void* stream);
/* du = d/du of act(scale * xhat + shift) given dh; dscale / dshift = beta * old + sums over images and positions
 * (deterministic); workspace >= n * 2 * c floats. */
int tdg_instance_norm_bwd(int dtype, const void* dh, int dh_cs, const void* u, int n, int hw, int c, int cs, const float* scale,
                          const float* shift, const float* stats, int act, float leak, void* du, float* dscale, float* dshift,
                          float beta, void* workspace, size_t workspace_bytes, void* stream);
/* out = act(a + b), flat, one layout: the shortcut sum of `residual` (hem/ops/layers.py:297-304); TDG_ACT_NONE = add */
int tdg_add_act(int dtype, const void* a, const void* b, size_t n, int act, float leak, void* out, void* stream);
/* tdg_bn_fwd with the batch statistics taken from TDG_COL_BN partials of the producing conv (pivot = that conv's
 * bias, may be NULL) instead of a pass over u: stats, then pre = (u - mean) * rstd + beta and h = act(pre).
 * pre may be NULL when no backward pass will follow this forward pass (the generator pass of a critic step): only h is written. */
int tdg_bn_fwd_from_partials(int dtype, const void* u, int rows, int c, int cs, const float* beta, float eps, int act,
                             float leak, void* pre, void* h, int h_cs, float* stats, const float* partial, int nblk,
                             const float* pivot_bias, void* stream);
/* dbias (optional): += / = the column sums of the stored du, i.e. the bias gradient of the conv in front of the batch
 * norm (BiasAddGrad of ops/layers.py:102 under :103), taken from the pass that writes du instead of a pass of its own. */
int tdg_bn_bwd(int dtype, const void* dh, int dh_cs, const void* pre, int rows, int c, int cs,
               const float* beta, const float* stats, int act, float leak, void* du, float* dbeta,
               float beta_acc, float* dbias, float dbias_acc, void* workspace, size_t workspace_bytes, void* stream);

/* ---- elementwise ------------------------------------------------------------------------ */
/* y = act(x + bias[c]) over rows x c (tf.nn.bias_add + activation) */
int tdg_bias_act(int dtype, const void* x, int rows, int c, int cs, const float* bias, int act,
                 float leak, void* y, void* stream);
/* dx = dy * act'(.) given the post-activation tensor (tanh/sigmoid/lrelu) */
int tdg_act_bwd(int dtype, const void* dy, const void* post, size_t n, int act, float leak, void* dx,
                void* stream);
/* out = scale * (in + shift), f32 in -> dtype out   (models/gan.py:50: 2*(x-0.5)) */
int tdg_affine_cast(int dtype, const float* in, size_t n, float scale, float shift, void* out, void* stream);
/* Two 4-channel activations from two compact f32 row sets (ca + cb == 4): out_ab[r] = scale * ([a[r] | b[r]] + shift) and
 * out_a0[r] = the same with zeros in b's channels -- pix2pix's critic inputs [x | y] and [x | G(x) to come]
 * (hem/models/pix2pix.py:103-104 rescale + the concat of :236) in one pass. */
int tdg_affine_cast_pair(int dtype, const float* a, int ca, const float* b, int cb, size_t rows, float scale, float shift,
                         void* out_ab, void* out_a0, void* stream);
/* out[r*cs + ch] = scale * (in[r*c + ch] + shift), ch < c: compact f32 rows -> channel-padded activation */
int tdg_affine_cast_rows(int dtype, const float* in, int rows, int c, int cs, float scale, float shift, void* out,
                         void* stream);
int tdg_cast_to_f32(int dtype, const void* in, size_t n, float* out, void* stream);
int tdg_cast_from_f32(int dtype, const float* in, size_t n, void* out, void* stream);
/* xhat[r,:] = x[r,:] + alpha[r] * (g[r,:] - x[r,:])   (models/gan.py:225-226) */
int tdg_gp_interp(int dtype, const void* x, const void* g, const float* alpha, int rows, int cols,
                  void* xhat, void* stream);
/* acc[0] = beta*acc[0] + sum(x^2) over n elements (models/gan.py:229); deterministic.
 * ONE-STREAM RULE: the kernel hands its last block a ticket from a per-process device-global slot, so launches of this entry point must not overlap on the device; the library returns TDG_EINVAL if it is launched on a second (non-capturing) stream of the process. */
int tdg_sumsq(int dtype, const void* x, size_t n, float* acc, float beta, void* workspace,
              size_t workspace_bytes, void* stream);
size_t tdg_reduce_workspace_bytes(size_t n);
/* out[0] = mean(x[0:n]) f32 input (tf.reduce_mean of D outputs, models/gan.py:196-204) */
int tdg_mean_f32(const float* x, int n, float* out, void* stream);
/* out[0] = beta * out[0] + sum of n floats, one launch (the bias gradient of a dense layer with one output: ops/layers.py:56-57) */
int tdg_sum_f32(const float* x, int n, float* out, float beta, void* stream);
/* out[s] = mean(x[s*seglen : (s+1)*seglen]) for s < nseg, one launch (the means of D(x) and D(g), :196-197) */
int tdg_mean_segments_f32(const float* x, int nseg, int seglen, float* out, void* stream);
/* Vanilla-GAN losses on post-sigmoid scores (models/gan.py:193-194) and their gradients w.r.t. the LOGITS:
 *   scal[0] = d_loss = mean(-log(dr+1e-8) - log(1-df+1e-8)),  scal[1] = g_loss = mean(-log(df+1e-8))
 *   seed_real = d d_loss/d logit_real, seed_fake_d = d d_loss/d logit_fake, seed_fake_g = d g_loss/d logit_fake */
int tdg_gan_logloss(const float* d_real, const float* d_fake, int n, float* seed_real, float* seed_fake_d,
                    float* seed_fake_g, float* scal, void* stream);
/* ---- pix2pix losses (hem/models/pix2pix.py:263-304) -------------------------------------------------
 * Sigmoid cross-entropy terms on the PatchGAN logits (channel 0, row stride cs) of the real pass
 * (rows [0, rows)) and the fake pass (rows [rows, 2*rows)):
 *   scal[0] = d_real = mean xent(z_real, 1), scal[1] = d_fake = mean xent(z_fake, 0),
 *   scal[2] = g_fake = mean xent(z_fake, 1)      (xent(z,l) = max(z,0) - z*l + log(1+exp(-|z|)), App. A-7)
 * mode 1 (D step): seed = d(d_real + d_fake)/dz for both passes; mode 2 (G step): seed_fake = d g_fake/dz,
 * seed_real = 0; mode 0: losses only.  seed has the layout of logits. */
int tdg_p2p_xent(int dtype, const void* logits, int rows, int cs, int mode, void* seed, float* scal, void* stream);
/* y, g: channel 0 of [-1,1] tensors with row stride cs.  scal[0] = l1 = mean|y01 - g01|, scal[1] = rmse (:285,299,
 * hem/ops/losses.py:10-11) after rescaling both to [0,1].  If dg != NULL: dg[r*dgs] += weight * d l1 / d g. */
int tdg_p2p_l1(int dtype, const void* y, const void* g, int rows, int cs, float weight, void* dg, int dgs, float* scal,
               void* workspace, size_t workspace_bytes, void* stream);
/* ---- VAE pieces (models/vae.py:66-90,113-129) -------------------------------------------------------
 * heads = [z_mean | z_stddev] rows of 2L (channel stride hs); z = mean + stddev * eps (models/vae.py:128) */
int tdg_vae_reparam(int dtype, const void* heads, int hs, const void* eps, int es, int rows, int L, void* z, int zs,
                    void* stream);
/* dheads = [dz | dz * eps] */
int tdg_vae_reparam_bwd(int dtype, const void* dz, int zs, const void* eps, int es, int rows, int L, void* dheads, int hs,
                        void* stream);
/* --vae_full_elbo (SURVEY App. C-7 opt-in: the reference differentiates the reconstruction term alone, models/vae.py:41):
 * dheads = [dz + w*m | dz*eps + w*(s - s / (1e-8 + s^2))], the gradient of decoder_loss + w * latent_loss (:80-81) */
int tdg_vae_reparam_bwd_kl(int dtype, const void* dz, int zs, const void* eps, int es, const void* heads, int hs_in,
                           float kl_weight, int rows, int L, void* dheads, int hs, void* stream);
/* scal[0] = latent_loss = 0.5 * sum(mean^2 + std^2 - log(1e-8 + std^2) - 1)  (models/vae.py:80-81) */
int tdg_vae_kl(int dtype, const void* heads, int hs, int rows, int L, float* scal, void* workspace, size_t workspace_bytes,
               void* stream);
/* scal[0] = decoder_loss = -sum(x log(1e-8+d) + (1-x) log(1e-8+1-d)) (models/vae.py:76-77);
 * seed = d decoder_loss / d d.  x: compact f32 [rows, c] in [0,1]; d, seed: [rows, cs] in dtype. */
int tdg_vae_bce(int dtype, const float* x, const void* d, int rows, int c, int cs, void* seed, float* scal, void* workspace,
                size_t workspace_bytes, void* stream);
/* Convolutional autoencoder loss (models/cnn.py:31,75-79): with xs = scale * (x + shift) (the rescale to [-1,1]),
 * scal[0] = mean |xs - d|, seed = d loss / d d = sign(d - xs) / (rows * c)  (sign(0) = 0, TF AbsGrad).
 * x: compact f32 [rows, c]; d, seed: [rows, cs] in dtype. */
int tdg_l1_loss(int dtype, const float* x, const void* d, int rows, int c, int cs, float scale, float shift, void* seed,
                float* scal, void* workspace, size_t workspace_bytes, void* stream);
/* tf.nn.dropout(y, keep_prob) (hem/ops/layers.py:207) given the uniform draws u [rows, c] (compact f32):
 * y = y * floor(keep_prob + u) / keep_prob, in place on [rows, ycs]-strided y.  Applied to the incoming gradient with
 * the same u it is the layer's backward. */
int tdg_dropout(int dtype, void* y, int rows, int c, int ycs, const float* u, float keep, void* stream);
/* GP scalars from sumsq (device-resident, no host sync): slopes = sqrt(ss);
 * scal[0] = penalty = (slopes-1)^2 ; scal[1] = lambda * 2*(slopes-1)/slopes            */
int tdg_gp_scalars(const float* sumsq, float lambda, float* scal, void* stream);
/* tdg_sumsq (beta = 0) and tdg_gp_scalars in one launch (models/gan.py:229-230: slopes over the WHOLE batch tensor, penalty) */
int tdg_gp_sumsq(int dtype, const void* x, size_t n, float* sumsq, float lambda, float* scal, void* workspace,
                 size_t workspace_bytes, void* stream);
/* out = coef[0] * in   (u = d penalty / d v, coefficient read from device memory) */
int tdg_scale_by_dev(int dtype, const void* in, size_t n, const float* coef, void* out, void* stream);
/* --gp_per_sample (SURVEY App. C-4 opt-in: the reference takes ONE norm over the whole batch tensor, models/gan.py:229):
 * per row r of v [rows, cols]: slopes_r = |v_r|, pen_rows[r] = (slopes_r - 1)^2 and
 * u_r = lambda * 2 (slopes_r - 1) / (slopes_r * rows) * v_r  =  d(lambda * mean_r pen_r) / d v_r. */
int tdg_gp_rows(int dtype, const void* v, int rows, int cols, float lambda, float* pen_rows, void* u, void* stream);
/* x[i] = value for i < n (f32) */
int tdg_fill_f32(float* x, size_t n, float value, void* stream);
/* column sums of a rows x c activation: db[c] = beta*db[c] + sum_r dy[r,c] (bias gradient) */
int tdg_bias_grad(int dtype, const void* dy, int rows, int c, int cs, float* db, float beta,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- optimizers on flat f32 buckets (tf.train.*Optimizer via util.py:150-183) ----------- */
/* Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller */
int tdg_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float beta1,
                  float beta2, float eps, float grad_scale, void* stream);
/* Same update with the step count t kept in device memory (t_dev[0] = number of steps already applied):
 * lr_t is derived in-kernel, so a captured hipGraph can be replayed without re-baking arguments, and the kernel itself
 * counts the step (t_dev[0] += 1, by its last block).
 * ONE-STREAM RULE: the kernel hands its last block a ticket from a per-process device-global slot, so launches of this entry point must not overlap on the device; the library returns TDG_EINVAL if it is launched on a second (non-capturing) stream of the process. */
int tdg_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                      float beta2, float eps, float grad_scale, int32_t* t_dev, void* stream);
int tdg_add_i32(int32_t* x, int32_t inc, void* stream);
/* RMSProp (rms slot initialised to 1 by the caller), optional momentum, not centered */
int tdg_rmsprop_step(float* p, const float* g, float* rms, float* mom, size_t n, float lr,
                     float decay, float momentum, float eps, float grad_scale, void* stream);
/* RMSProp with centered=True (util.py:161-164 `centered = args.centered`): mg slot starts at 0 */
int tdg_rmsprop_centered_step(float* p, const float* g, float* mg, float* rms, float* mom, size_t n,
                              float lr, float decay, float momentum, float eps, float grad_scale,
                              void* stream);
/* tf.train.AdagradOptimizer / ProximalAdagradOptimizer with zero regularisation (util.py:167-168,
 * :173-174): acc slot starts at 0.1 */
int tdg_adagrad_step(float* p, const float* g, float* acc, size_t n, float lr, float grad_scale,
                     void* stream);
/* tf.train.AdadeltaOptimizer (util.py:165-166): rho 0.95, eps 1e-8 by default, slots start at 0 */
int tdg_adadelta_step(float* p, const float* g, float* acc, float* acc_update, size_t n, float lr,
                      float rho, float eps, float grad_scale, void* stream);
/* tf.train.FtrlOptimizer (util.py:182-183), lr_power -0.5: acc slot starts at 0.1, linear at 0 */
int tdg_ftrl_step(float* p, const float* g, float* acc, float* linear, size_t n, float lr, float l1,
                  float l2, float grad_scale, void* stream);
int tdg_sgd_momentum_step(float* p, const float* g, float* acc, size_t n, float lr, float momentum,
                          float grad_scale, void* stream);
/* p = clamp(p, lo, hi)  (models/gan.py:142-143; never executed by the reference, App. C-3) */
int tdg_clamp(float* p, size_t n, float lo, float hi, void* stream);
/* flag[0] = 1 if any element is NaN/Inf (hem/util/training.py:52-53 tf.check_numerics) */
int tdg_check_finite(const float* x, size_t n, int* flag, void* stream);

/* ---- RNG: Philox4x32-10 counter streams (tf.random_normal / tf.random_uniform,
 *      models/gan.py:246,224) -------------------------------------------------------------- */
int tdg_random_normal(int dtype, uint64_t seed, uint64_t stream_id, uint64_t offset, size_t n,
                      void* out, void* stream);
int tdg_random_uniform_f32(uint64_t seed, uint64_t stream_id, uint64_t offset, size_t n, float* out,
                           void* stream);
/* Graph-replayable forms: the counter offset is ((draw_dev[0] + 1) << 24), read from device memory, and the kernel
 * itself counts the draw (draw_dev[0] += 1, by its last block), so a replay of the same launch is a fresh draw.
 * ONE-STREAM RULE: the kernel hands its last block a ticket from a per-process device-global slot, so launches of this entry point must not overlap on the device; the library returns TDG_EINVAL if it is launched on a second (non-capturing) stream of the process. */
int tdg_random_normal_dev(int dtype, uint64_t seed, uint64_t stream_id, int32_t* draw_dev, size_t n,
                          void* out, void* stream);
int tdg_random_uniform_f32_dev(uint64_t seed, uint64_t stream_id, int32_t* draw_dev, size_t n, float* out,
                               void* stream);

/* ---- input pipeline, host side (no GPU work): PNG scanline reconstruction (filter types 0-4 of the PNG
 *      specification) after the caller has inflated the IDAT stream -- what tf.image.decode_png does for the
 *      reference's nyuv2 / floorplan records (hem/data/nyuv2.py:152-153, data.py:15).  `filtered` holds `rows`
 *      scanlines of 1 + row_bytes bytes, `out` receives rows * row_bytes bytes; bpp = bytes per complete pixel. */
int tdg_png_unfilter(const unsigned char* filtered, int rows, int row_bytes, int bpp, unsigned char* out);

/* ---- tdg_jpeg_info / tdg_jpeg_decode: a JPEG file's bytes -> width x height x 3 RGB bytes (host memory, no device work).
 *      Replaces `tf.image.decode_image(..., channels=3)` on the reference's floorplan records, whose `image` feature is
 *      the raw bytes of the source file (data/floorplan_tfrecords.py:26-41 writes them, data.py:15 decodes them).
 *      Baseline and extended-sequential Huffman files, 8 bit, 1 (grayscale, replicated to 3 channels) or 3 components,
 *      sampling factors 1 and 2, restart intervals; libjpeg's default arithmetic (accurate integer inverse DCT, triangle
 *      chroma upsampling, 16-bit fixed-point YCbCr -> RGB).  Progressive / arithmetic / 12-bit / CMYK files: TDG_EINVAL
 *      with the reason in tdg_last_error(). */
int tdg_jpeg_info(const unsigned char* data, size_t nbytes, int* width, int* height, int* components);
int tdg_jpeg_decode(const unsigned char* data, size_t nbytes, unsigned char* rgb, size_t rgb_bytes);

/* ---- tdg_shuffle_draw / tdg_gather_rows: the host side of the streaming input pipeline (no device work) --------------
 *      `d.repeat().shuffle(buffer_size)` of the reference (data.py:56-57, train.py:171-174: buffer_size 10000) on example
 *      INDICES: `buf[0 .. buf_len)` holds the indices currently in the shuffle buffer, `*next_in` the next index of the
 *      repeated stream 0, 1, ..., n_total-1, 0, 1, ... to enter it.  Each of the `count` draws picks a uniformly random slot
 *      (xoshiro256** on state[4], Lemire's unbiased range reduction), writes its index to `out` and refills the slot from
 *      the stream -- tf.data's ShuffleDataset semantics: an example can appear at most ~buf_len draws before its stream
 *      position, never later than the buffer lets it wait.  The caller fills buf with the first min(buffer_size, ...)
 *      stream indices and sets *next_in behind them.
 *      tdg_gather_rows: out[i] = src[idx[i]] for rows of `row_bytes` bytes (the batch assembled in pinned host memory). */
int tdg_shuffle_draw(int64_t* buf, int64_t buf_len, uint64_t* state, int64_t* next_in, int64_t n_total, int64_t count,
                     int64_t* out);
int tdg_gather_rows(const unsigned char* src, int64_t n_rows, size_t row_bytes, const int64_t* idx, int64_t count,
                    unsigned char* out);

#ifdef __cplusplus
}
#endif
#endif /* TDG_H_ */
