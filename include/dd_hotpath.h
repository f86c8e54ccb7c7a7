/* dd_hotpath.h -- C ABI of the MI355X-native multi-camera -> BEV training hot path.
 *
 * The reference (annikabrundyn/driving-dirty) has no FFI: its hot path is reached through
 * the PyTorch-Lightning Python surface and ATen ops.  Each entry point below replaces the
 * ATen op(s) that one reference line invokes; the Python host (driving-dirty_amd/ops.py)
 * binds them with ctypes and owns every tensor.  See INTEGRATION.md for the binding.
 *
 * Conventions
 *   - plain pointers to DEVICE memory, sizes as int32/int64, no torch types;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - no allocation, no host synchronisation, no global state besides the thread-local
 *     error string; scratch memory is a caller-provided workspace sized by *_workspace_bytes;
 *   - return 0 on success, a DD_ERR_* code otherwise (unsupported shapes are refused,
 *     never silently approximated); dd_last_error() describes the last failure of the
 *     calling thread;
 *   - activations between conv layers are NHWC fp32 (channels innermost); parameters
 *     stay in PyTorch layouts (Conv2d OIHW, Linear [out,in]) so state_dict round-trips.
 */
#ifndef DD_HOTPATH_H
#define DD_HOTPATH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DD_ABI_VERSION 3

enum {
  DD_OK = 0,
  DD_ERR_UNSUPPORTED = 1, /* shape / option outside what the kernels implement */
  DD_ERR_BAD_ARG = 2,     /* NULL pointer, non-positive size, misaligned pointer */
  DD_ERR_LAUNCH = 3,      /* hipLaunch / hipFuncSetAttribute failed */
  DD_ERR_WORKSPACE = 4    /* workspace too small */
};

/* Epilogues of the conv kernels. */
enum {
  DD_EPI_NONE = 0,      /* y = conv(x)                                   */
  DD_EPI_BIAS = 1,      /* y = conv(x) + bias                            */
  DD_EPI_BIAS_RELU = 2, /* y = relu(conv(x) + bias)   (components.py:41-43: F.relu(self.cN(x))) */
  DD_EPI_RELU_MASK = 3  /* y = conv(x) * (mask > 0)   (autograd of the PREVIOUS layer's ReLU, fused) */
};

/* 3x3 convolution, padding 1, stride 1 or 2, Cout = 32.
 * cin_real: channels of the PyTorch weight (3 or 32); cin_store: channels of the NHWC
 * activation buffer (4 when cin_real == 3, the 4th channel being zero; else 32).
 * Replaces nn.Conv2d c1/c2/c3 of reference src/autoencoder/components.py:19-21. */
typedef struct dd_conv_desc {
  int32_t batch, height, width; /* input spatial size */
  int32_t cin_real, cin_store, cout;
  int32_t ksize, stride, pad;
  int32_t rows_per_task; /* 0 = library default; tuning knob, never changes results */
} dd_conv_desc;

int dd_abi_version(void);
const char* dd_last_error(void);

/* Measurement aid (no reference counterpart): ONE wave writes `nsamples` pairs (shader-clock counter, 100 MHz reference counter)
 * into samples[2 * nsamples] (device memory), sleeping `spin` x 127 x 64 clocks between two samples.  Launched on its own stream it
 * records the shader clock the part delivers while kernels on other streams run -- the power-management effect that per-kernel
 * profiler counters (which serialise kernels) cannot show for overlapped launches.  tools/clock_probe.py. */
int dd_clock_probe(uint64_t* samples, int32_t nsamples, int32_t spin, void* stream);

/* The conv kernels launch exactly as many workgroups as fit on the chip at once and give every wave an equal,
 * contiguous share of the work.  When another kernel must run BESIDE them for milliseconds (the RCCL all-reduce
 * of data-parallel training: its workgroups need LDS the conv workgroups do not leave free), hand it a few
 * compute units: with fewer than 256 the conv grids shrink accordingly instead of spilling into a second round.
 * Process-wide; 1..256 (default 256).  Never changes results (weight-gradient partial sums are reduced in a fixed
 * order for any grid). */
int dd_set_cu_budget(int32_t compute_units);
int dd_get_cu_budget(void);
/* Persistent workgroups per CU of the optimizer kernels (dd_adam_step, dd_adam_step_rankb; 1..8, default 1).  ONE when the pass runs
 * beside the fp32 conv backward -- a second workgroup per CU would be queued ahead of the next conv kernel, which then waits for the
 * whole pass to drain; FOUR when it runs by itself (bf16 models: nothing MFMA-bound to hide under), where more workgroups in flight
 * stream faster.  Process-wide, read at launch; never changes results (the update is elementwise). */
int dd_set_adam_blocks_per_cu(int32_t blocks);
/* Compute units dd_adam_step_rankb leaves WITHOUT a workgroup of its own (0..128, default 0; the grid shrinks by that many workgroups per
 * dd_set_adam_blocks_per_cu).  A pass that runs beside the backward from the moment its factors exist (optim.HipAdam's "early" slot)
 * holds 16-24 KB of LDS on every CU for a millisecond: a single-workgroup kernel of the backward that needs nearly the whole LDS
 * (dd_mlp_tail_bwd: 154 KB) then waits for the first workgroup of the pass to retire -- the autoencoder's step lost 0.6 ms there.
 * With 8 spare units such kernels start at once.  Process-wide, read at launch; never changes results. */
int dd_set_adam_spare_cus(int32_t compute_units);

/* ---- layout: 6-view gather (K4) --------------------------------------------------
 * views [B,6,3,H,W] fp32 -> wide NHWC4 image [B,H,6W,4] with the reference's view order
 * [0,1,2,5,4,3] (roadmap_bce_v2.py:58-62, autoencoder.py:55-57); channel 3 is zero.
 * mask_slot in 0..5 blanks that slot of the wide image (autoencoder.py:59-67) and, when
 * target != NULL, copies its original content to target [B,3,H,W] (NCHW); -1 = no mask.
 * wide_nchw (optional, may be NULL) also receives the stitched image as [B,3,H,6W] NCHW,
 * which is what the reference's six_to_one_task / wide_stitch_six_images return. */
int dd_stitch6(const float* views, float* wide_nhwc4, float* wide_nchw, float* target,
               int32_t batch, int32_t height, int32_t width, int32_t mask_slot, void* stream);

/* The same gather from a HOST array of `batch` per-sample DEVICE base pointers (each [6,3,H,W] contiguous; the table
 * travels in the kernel arguments, 64 samples per launch): the reference
 * receives the batch as a tuple of tensors (helper.py:22-23) and torch.stack()s it first (roadmap_bce_v2.py:55);
 * this skips that copy. */
int dd_stitch6_ptrs(const float* const* sample_ptrs, float* wide_nhwc4, int32_t batch, int32_t height, int32_t width,
                    void* stream);

/* Box rasteriser for a whole batch: boxes_to_binary_map (src/utils/bb_to_img.py:5-20) as called per sample by
 * bb_coord_to_map (src/bounding_box_model/spatial_bb/spatial_w_rm.py:85-95).  `boxes` = the samples' [n,2,4] corner
 * tensors concatenated on the DEVICE (boxes_dtype 0 = f64 as the dataset holds them, data_helper.py:129; 1 = f32);
 * `sample_offsets` = HOST array of batch+1 box indices (sample s owns boxes [off[s], off[s+1]); empty samples give an
 * all-zero map).  maps = [batch,800,800] fp32 0/1 with the vertical flip applied, bit-identical to
 * Pillow 12.2's ImageDraw.polygon fill. */
int dd_boxes_to_binary_map(const void* boxes, int32_t boxes_dtype, const int32_t* sample_offsets, float* maps,
                           int32_t batch, void* stream);

/* Input pipeline variant: frames [B,6,H,W,3] uint8 as a JPEG decoder emits them -> the same wide
 * NHWC4 fp32 image, with torchvision ToTensor's /255 (reference autoencoder.py:133, data_helper.py:63-68) fused. */
int dd_stitch6_u8(const unsigned char* frames, float* wide_nhwc4, int32_t batch, int32_t height, int32_t width,
                  void* stream);
/* The same from a HOST array of `batch` per-sample DEVICE base pointers (each [6,H,W,3] uint8 contiguous: the collate's tuple of
 * decoded frames, helper.py:22-23 over data_helper.py:63-68 without ToTensor), with BasicAE.six_to_one_task's masked-view task
 * (autoencoder.py:59-73) optional as in dd_stitch6: `mask_slot` in [0,5] blanks that slot of the wide image and writes the view, as
 * fp32 NCHW [B,3,H,W], to `target` (may be NULL); -1 = no mask.  The division by 255 is a true fp32 division (ToTensor's). */
int dd_stitch6_u8_ptrs(const unsigned char* const* sample_ptrs, float* wide_nhwc4, float* target, int32_t batch, int32_t height,
                       int32_t width, int32_t mask_slot, void* stream);

/* NCHW [B,C,H,W] <-> NHWC [B,H,W,Cs] (Cs >= C; extra channels written as zero / ignored). */
int dd_nchw_to_nhwc(const float* src, float* dst, int32_t batch, int32_t c, int32_t h, int32_t w,
                    int32_t c_store, void* stream);
int dd_nhwc_to_nchw(const float* src, float* dst, int32_t batch, int32_t c, int32_t h, int32_t w,
                    int32_t c_store, void* stream);

/* ---- conv 3x3 (K1) ----------------------------------------------------------------
 * Packed weights: the MFMA B-operand image consumed by the kernels (dd_conv_packed_floats
 * floats).  kind: 0 = forward, 1 = data-gradient of a stride-1 conv (flipped taps,
 * Cin<->Cout swapped), 2 = data-gradient of a stride-2 conv. */
int64_t dd_conv_packed_floats(const dd_conv_desc* d, int32_t kind);
int dd_conv_pack(const float* w_oihw, float* packed, const dd_conv_desc* d, int32_t kind, void* stream);

/* y[B,Ho,Wo,32] = epilogue(conv3x3(x[B,H,W,cin_store], W) ...).  `mask` is only read by
 * DD_EPI_RELU_MASK and has y's shape. */
int dd_conv_fwd(const float* x, const float* packed_fwd, const float* bias, const float* mask,
                float* y, const dd_conv_desc* d, int32_t epilogue, void* stream);

/* dx[B,H,W,32] = conv_transpose(dy[B,Ho,Wo,32], W) * (relu_src > 0 if relu_src != NULL).
 * relu_src is the conv's INPUT activation (the previous layer's ReLU output), so the previous
 * ReLU's backward is fused here.  Needs packed kind 1 (stride 1) or 2 (stride 2). */
int dd_conv_dgrad(const float* dy, const float* packed_dgrad, const float* relu_src, float* dx,
                  const dd_conv_desc* d, void* stream);

/* The ReLU sign as ONE BIT per activation: dd_conv_fwd_relu_bits = dd_conv_fwd(DD_EPI_BIAS_RELU) that also writes
 * relu_bits[B,Ho,Wo] (uint32: bit c = y[..., c] > 0); dd_conv_dgrad_relu_bits = dd_conv_dgrad whose fused ReLU mask
 * is read from such a bit plane (4 bytes per pixel instead of 128): the HBM-bound stride-2 data gradient stops
 * re-reading a 1.9 GB activation only for its signs. */
int dd_conv_fwd_relu_bits(const float* x, const float* packed_fwd, const float* bias, float* y, uint32_t* relu_bits,
                          const dd_conv_desc* d, void* stream);
int dd_conv_dgrad_relu_bits(const float* dy, const float* packed_dgrad, const uint32_t* relu_bits, float* dx,
                            const dd_conv_desc* d, void* stream);

/* ---- Conv -> BatchNorm2d -> ReLU variant (reference src/autoencoder/components_v2.py:19-24,43-46) -----------------
 * The conv writes the PRE-normalisation tensor u = conv(x) + bias and gathers the batch statistics in its epilogue
 * (`stats`: dd_conv_stats_floats() floats, zeroed by the call).  dd_bn2d_finalize turns them into batch mean / inv-std,
 * updates the running statistics like torch.nn.BatchNorm2d (momentum, unbiased variance) and writes `affine`[128]:
 * [0:32) scale, [32:64) shift (as INPUT transform of the next layer) and the same pair at [64:128) (as MASK transform).
 * Whoever reads u applies y = relu(u*scale + shift) on the fly: the next conv (`in_affine`), the pool, the ReLU mask
 * of the data gradient, the weight gradient's input -- the normalised activation is never written to memory. */
int64_t dd_conv_stats_floats(void);
int dd_conv_fwd_stats(const float* x, const float* packed_fwd, const float* bias, const float* in_affine, float* u,
                      float* stats, const dd_conv_desc* d, void* stream);
int dd_bn2d_finalize(const float* stats, int64_t count, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, int32_t training, float* affine, float* save_mean,
                     float* save_invstd, void* stream);
/* Stand-alone batch statistics of a 32-channel NHWC tensor u [npix, 32] into the same `stats` table (the v2 Decoder's
 * ConvTranspose2d -> BatchNorm2d, components_v2.py:73-78,96-98: its convs run on the generic kernels, which have no
 * statistics epilogue); wavefront-shuffle reductions, deterministic. */
int dd_bn2d_stats(const float* u, float* stats, int64_t npix, void* stream);
/* y = relu(u*scale + shift), u/y [npix, 32] (only when a caller wants the activation itself, e.g. the c3_only exit) */
int dd_bn2d_apply_relu(const float* u, const float* affine, float* y, int64_t npix, void* stream);
/* BatchNorm2d backward on g = dL/d(BN output) (already ReLU-masked): dgamma, dbeta (wavefront-shuffle reduction) and
 * du = gamma*invstd * (g - dbeta/N - xhat*dgamma/N)  (training) or gamma*invstd*g (eval). */
int64_t dd_bn2d_workspace_bytes(void);
int dd_bn2d_bwd(const float* g, const float* u, const float* gamma, const float* save_mean, const float* save_invstd,
                float* du, float* dgamma, float* dbeta, int64_t npix, int32_t training, void* workspace, void* stream);
/* conv data / weight gradients whose ReLU mask / input is a pre-BN tensor seen through `affine` */
int dd_conv_dgrad_bn(const float* dy, const float* packed_dgrad, const float* pre_bn, const float* affine, float* dx,
                     const dd_conv_desc* d, void* stream);
int dd_conv_wgrad_bn(const float* pre_bn, const float* affine, const float* dy, float* dw_oihw, float* dbias, void* workspace,
                     int64_t workspace_bytes, const dd_conv_desc* d, void* stream);
/* NCHW-order max_pool1d(4) over relu(u*scale+shift) and its backward (gradient w.r.t. the BN output), C = 32, H*W % 4 == 0 */
int dd_pool4_bn_fwd(const float* u, const float* affine, float* pooled, int32_t batch, int32_t h, int32_t w, void* stream);
int dd_pool4_bn_bwd(const float* dpooled, const float* u, const float* affine, float* dfeat, int32_t batch, int32_t h, int32_t w,
                    void* stream);

/* dw_oihw[32,cin_real,3,3], dbias[32] from x[B,H,W,cin_store] and dy[B,Ho,Wo,32]
 * (dy already multiplied by this layer's ReLU mask).  Deterministic: per-wave partial sums
 * in `workspace`, then a fixed-order reduction. */
int64_t dd_conv_wgrad_workspace_bytes(const dd_conv_desc* d);
int dd_conv_wgrad(const float* x, const float* dy, float* dw_oihw, float* dbias, void* workspace,
                  int64_t workspace_bytes, const dd_conv_desc* d, void* stream);

/* out = dy * (y > 0), n elements (ReLU backward as a stand-alone pass). */
int dd_relu_bwd(const float* dy, const float* y, float* out, int64_t n, void* stream);

/* Two helpers that let a padding-0 3x3 layer with 32 channels on both sides (SpatialMappingCNN.out_conv, spatial_bb/components.py:24,73:
 * [B,258,258,32] -> [B,256,256,32]) run on the c2 layer's Winograd kernels below, which compute the padding-1 convolution of the same
 * input: its outputs are the interior of theirs.
 *   dd_relu_sign_bits: bits[p] bit c = x[p][c] > 0 for an NHWC activation of 32 channels -- the sign words dd_conv_*_fwd_relu_bits write
 *     beside their own output, for an activation some other kernel produced (the data gradient's ReLU mask);
 *   dd_relu_bwd_pad_bits: out_pad [B,h+2,w+2,32] = dy [B,h,w,32] where the sign word bits_pad [B,h+2,w+2] of that pixel has the channel's
 *     bit set, zero elsewhere and on the border ring (outputs that do not exist in the padding-0 layer carry no gradient); dy's 32 channels
 *     may be a slice [dy_coff, +32) of dy_cstore stored ones (the concat buffer's gradient, read where it lies). */
int dd_relu_sign_bits(const float* x, uint32_t* bits, int64_t npix, void* stream);
int dd_relu_bwd_pad_bits(const float* dy, const uint32_t* bits_pad, float* out_pad, int32_t batch, int32_t h, int32_t w, int32_t dy_cstore,
                         int32_t dy_coff, void* stream);

/* ---- NCHW-order max_pool1d(4) on an NHWC feature (K6) -----------------------------------
 * feat [B,H,W,C] NHWC.  The reference flattens the NCHW tensor and pools windows of 4 along
 * that vector (components.py:46-47); pooled[b, g] = max_{i<4} feat_nchw_flat[b, 4g+i],
 * g < floor(C*H*W/4).  Backward routes dpooled to the first maximum of each window and
 * multiplies by (feat > 0), i.e. it also applies the backward of the ReLU that produced feat. */
int dd_pool4_fwd(const float* feat, float* pooled, int32_t batch, int32_t h, int32_t w, int32_t c,
                 void* stream);
int dd_pool4_relu_bwd(const float* dpooled, const float* feat, float* dfeat, int32_t batch, int32_t h,
                      int32_t w, int32_t c, void* stream);
/* The same pooling with the backward's routing decided in the forward: `idx` receives one uint16 per thread-quad
 * (4 bits per window: index of the first maximum | (max > 0) << 2; dd_pool4_idx_elems of them, -1 when H*W or C
 * is not a multiple of 4), and dd_pool4_idx_relu_bwd scatters dpooled from those codes alone -- the 481 MB c3
 * feature (components.py:43) is neither kept for nor re-read by the backward. */
/* The joint roadmap + box model feeds the c3 feature to the pool AND to the box heads (joint_model: both losses on one encoder pass):
 * dfeat = (feat > 0) * (gfeat + routed dpooled) in ONE pass -- dd_relu_bwd(gfeat, feat) + dd_pool4_relu_bwd(dpooled, feat) + dd_add, same
 * arithmetic.  C == 32, H*W % 4 == 0, 16-byte aligned tensors. */
int dd_pool4_relu_bwd_add(const float* dpooled, const float* feat, const float* gfeat, float* dfeat, int32_t batch, int32_t h, int32_t w,
                          int32_t c, void* stream);
int64_t dd_pool4_idx_elems(int32_t batch, int32_t h, int32_t w, int32_t c);
int dd_pool4_fwd_idx(const float* feat, float* pooled, uint16_t* idx, int32_t batch, int32_t h, int32_t w, int32_t c,
                     void* stream);
int dd_pool4_idx_relu_bwd(const float* dpooled, const uint16_t* idx, float* dfeat, int32_t batch, int32_t h, int32_t w,
                          int32_t c, void* stream);

/* ---- dense head pieces (K8, K9) ---------------------------------------------------------
 * BatchNorm1d (batch statistics when training != 0, else running statistics) -> ReLU ->
 * dropout with a caller-supplied keep mask (components.py:104-109).  x,y,keep: [rows, feat].
 * keep may be NULL (no dropout); scale = 1/(1-p).  save_mean/save_invstd [feat] are written in
 * training mode and consumed by the backward.  Running stats are updated in place with
 * `momentum` using the unbiased batch variance, as torch.nn.BatchNorm1d does; num_batches_tracked (device
 * int64, may be NULL) is incremented in training mode, as the module's forward does. */
int dd_bn_relu_drop_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, const float* keep, float* y, float* save_mean,
                        float* save_invstd, int32_t rows, int32_t feat, float eps, float momentum,
                        float scale, int32_t training, int64_t* num_batches_tracked, void* stream);
int dd_bn_relu_drop_bwd(const float* dy, const float* x, const float* y, const float* gamma,
                        const float* keep, const float* save_mean, const float* save_invstd,
                        const float* running_mean, const float* running_var, float* dx, float* dgamma, float* dbeta, int32_t rows,
                        int32_t feat, float eps, float scale, int32_t training, void* stream);

/* ---- the encoder's small tail as ONE launch each way (components.py:48-51 with DenseBlock.forward :104-109):
 *   lin1 [m,h1] (output of the big fc1 Linear) -> BatchNorm1d -> ReLU -> dropout -> Linear(h1->h2) -> BatchNorm1d ->
 *   ReLU -> dropout -> Linear(h2->l) -> z [m,l].
 * Same arithmetic and argument meaning as dd_bn_relu_drop_* and dd_linear_* chained (13 + 8 launches); one 256-thread
 * workgroup, operands in LDS.  Limits: m <= 32, h1, h2, l <= 128 and multiples of 4
 * (dd_mlp_tail_supported returns 0 otherwise and the entry points DD_ERR_UNSUPPORTED: chain the separate kernels).
 * Forward outputs kept for the backward: y1 [m,h1], lin2 [m,h2], y2 [m,h2], save_mean/invstd 1, 2. */
int dd_mlp_tail_supported(int32_t m, int32_t h1, int32_t h2, int32_t l);
int dd_mlp_tail_fwd(const float* lin1, const float* gamma1, const float* beta1, float* running_mean1, float* running_var1,
                    int64_t* num_batches_tracked1, const float* keep1, const float* w2, const float* bias2,
                    const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                    int64_t* num_batches_tracked2, const float* keep2, const float* wz, const float* bz, float* y1,
                    float* lin2, float* y2, float* z, float* save_mean1, float* save_invstd1, float* save_mean2,
                    float* save_invstd2, int32_t m, int32_t h1, int32_t h2, int32_t l, float eps1, float eps2,
                    float momentum1, float momentum2, float scale1, float scale2, int32_t training, void* stream);
int dd_mlp_tail_bwd(const float* dz, const float* lin1, const float* y1, const float* lin2, const float* y2,
                    const float* gamma1, const float* gamma2, const float* keep1, const float* keep2, const float* w2,
                    const float* wz, const float* save_mean1, const float* save_invstd1, const float* save_mean2,
                    const float* save_invstd2, const float* running_mean1, const float* running_var1,
                    const float* running_mean2, const float* running_var2, float* dlin1, float* dgamma1, float* dbeta1,
                    float* dw2, float* dbias2, float* dgamma2, float* dbeta2, float* dwz, float* dbz, int32_t m, int32_t h1,
                    int32_t h2, int32_t l, float eps1, float eps2, float scale1, float scale2, int32_t training,
                    void* stream);

/* Mean binary cross-entropy with logits over n elements (roadmap_bce_v2.py:106), one pass:
 * loss_out[0] = mean(max(z,0) - z*t + log1p(exp(-|z|))); dlogits (optional) = (sigmoid(z) - t) *
 * grad_scale / n; probs (optional) = sigmoid(z).  target is fp32 0/1.  partials: workspace of
 * dd_loss_workspace_bytes(n) bytes. */
int64_t dd_loss_workspace_bytes(int64_t n);
int dd_bce_logits(const float* logits, const float* target, float* loss_out, float* dlogits,
                  float* probs, int64_t n, float grad_scale, void* workspace, void* stream);
/* The same with the target as bytes 0/1 (the dataset's bool road_image, data_helper.py:137-139; n % 4 == 0 or tail
 * handled): skips the .float() pass of roadmap_bce_v2.py:87. */
int dd_bce_logits_u8(const float* logits, const unsigned char* target, float* loss_out, float* dlogits, float* probs,
                     int64_t n, float grad_scale, void* workspace, void* stream);
/* The same with the masks as the collate hands them over -- a tuple of per-sample bool tensors (helper.py:22-23), which the
 * reference stacks first (roadmap_bce_v2.py:87): target_ptrs is a HOST array of `batch` (<= 64) device pointers to
 * per_sample bytes each (per_sample % 4 == 0); logits / dlogits / probs are [batch * per_sample]. */
int dd_bce_logits_u8_ptrs(const float* logits, const unsigned char* const* target_ptrs, int32_t batch, int64_t per_sample,
                          float* loss_out, float* dlogits, float* probs, float grad_scale, void* workspace, void* stream);
/* x[0..n) *= *scalar (a DEVICE float), skipped entirely on the device when *scalar == 1: the upstream gradient of the scalar
 * loss in `loss.backward()` (roadmap_bce_v2.py:106 / Lightning's backward) is 1, and the loss kernels above have already
 * written d(loss)/d(input); autograd's generic `grad * dz` would spend a full pass on multiplying by one. */
int dd_scale_by_device_scalar(float* x, const float* scalar, int64_t n, void* stream);

/* probs = sigmoid(logits) (roadmap_bce_v2.py:81), n % 4 == 0 */
int dd_sigmoid(const float* z, float* p, int64_t n, void* stream);
/* dlogits = dprobs * probs * (1 - probs): autograd of the sigmoid INSIDE RoadMap.forward (roadmap_pretrain_ae.py:76,
 * the MSE twin takes its loss from the probabilities), n % 4 == 0 */
int dd_sigmoid_bwd(const float* dprobs, const float* probs, float* dlogits, int64_t n, void* stream);
/* Mean squared error mean((a-b)^2) with optional da = 2(a-b)*grad_scale/n (autoencoder.py:91). */
int dd_mse(const float* a, const float* b, float* loss_out, float* da, int64_t n, float grad_scale,
           void* workspace, void* stream);

/* ---- generic NHWC convolution (K1/K2/K3/K5 of SURVEY 2.3: every other Conv2d / ConvTranspose2d of the path) ---
 * One implicit-GEMM kernel family for the decoder (components.py:70-73,89-92) and the spatial bounding-box
 * heads (spatial_bb/components.py:18-26,129-139): any kernel size, stride, dilation and padding, Cin a multiple
 * of 4 channels taken from a channel slice of the input buffer, Cout <= 64 (<= 96 for the weight gradient) written into a channel slice and a
 * strided / offset pixel lattice of the output buffer.  The same forward kernel also runs
 *   - ConvTranspose2d stride 1 (dilated): a convolution with flipped taps and pad = dil*(k-1) - pad;
 *   - ConvTranspose2d k2 s2: four 1x1 convolutions, one per output phase (ostride 2, ooff = phase);
 *   - the data gradient of a stride-1 conv (flipped taps, channels swapped) and of a strided conv
 *     (div_h/div_w = stride: a tap contributes only where the coordinate divides);
 *   - rot90 / flip / mosaic tiling / channel concat of SpatialMappingCNN and *MergingCNN (components.py:43-73,159):
 *     the views are re-laid once by dd_view_to_nhwc4 and every conv writes straight into its tile / channel slice.
 * Input coordinate of output pixel (yo, xo) and tap (ky, kx):  num = yo*stride + k*dil - pad; with div > 1 the tap
 * is skipped unless num % div == 0, and the coordinate is num / div.  Out-of-image coordinates read zeros. */
enum { DD_EPI_BIAS_SIGMOID = 4 };

typedef struct dd_gconv_desc {
  int32_t batch;
  int32_t in_h, in_w, in_cstore, in_coff, cin;          /* input buffer [B,in_h,in_w,in_cstore]; channels [in_coff, in_coff+cin) */
  int32_t out_h, out_w;                                 /* output pixels computed per image */
  int32_t omem_h, omem_w, out_cstore, out_coff, cout;   /* output buffer [B,omem_h,omem_w,out_cstore]; channels [out_coff, out_coff+cout) */
  int32_t kh, kw, stride_h, stride_w, dil_h, dil_w, pad_h, pad_w;
  int32_t div_h, div_w;                                 /* 1 = ordinary convolution */
  int32_t ostride_h, ostride_w, ooff_h, ooff_w;         /* output pixel (yo,xo) lands at (yo*ostride_h+ooff_h, xo*ostride_w+ooff_w) */
  int32_t mask_pass_lo, mask_pass_hi;                   /* DD_EPI_RELU_MASK: output-BUFFER channels [lo,hi) are written unmasked (a channel
                                                           slice of a concat buffer that did not come out of a ReLU; 0,0 = mask everything) */
} dd_gconv_desc;

/* Weight image for the kernels: element (n, c, tap) is read from w[w_off + n*sn + c*sc + (flip ? T-1-tap : tap)];
 * n < n_real output channels, c < c_real input channels (zero beyond).  (sn, sc) express OIHW, IOHW, and the
 * transposed views needed by data gradients without copying the parameter. */
int64_t dd_gconv_packed_floats(const dd_gconv_desc* d);
int dd_gconv_pack(const float* w, float* packed, const dd_gconv_desc* d, int64_t w_off, int64_t sn, int64_t sc,
                  int32_t flip, int32_t n_real, int32_t c_real, void* stream);
/* y = epilogue(conv(x)); `mask` (DD_EPI_RELU_MASK only) has the output buffer's geometry. */
int dd_gconv_fwd(const float* x, const float* packed, const float* bias, const float* mask, float* y,
                 const dd_gconv_desc* d, int32_t epilogue, void* stream);
/* Weight gradient of y = conv(x): dw[w_off + n*sn + c*sc + (flip ? T-1-tap : tap)] = sum over pixels of
 * dy[..., n] * x[tap-shifted ..., c]  (dy has the OUTPUT buffer geometry of `d`);  dbias[n] = sum dy (may be NULL).
 * accumulate: bit 0 adds into dw, bit 1 adds into dbias instead of overwriting (the four phases of a k2 s2
 * transposed conv write disjoint weights but share one bias). */
int64_t dd_gconv_wgrad_workspace_bytes(const dd_gconv_desc* d);
int dd_gconv_wgrad(const float* x, const float* dy, float* dw, float* dbias, const dd_gconv_desc* d, int64_t w_off,
                   int64_t sn, int64_t sc, int32_t flip, int32_t n_real, int32_t c_real, int32_t accumulate,
                   void* workspace, int64_t workspace_bytes, void* stream);

/* The dilated stride-1 layers of the box heads (ConvTranspose2d k7 d7 / k7 d3 / k8 d8 / k6 d6, spatial_bb/components.py:90-92,
 * 135-138: forward = flipped-tap gather with pad d(k-1), data gradient = plain gather) on a phase-decomposed, LDS-staged
 * kernel: rows of equal residue mod d share their input rows, the input patch of an 8-channel chunk sits in LDS, a wave
 * owns 64 pixels x all Cout.  Same descriptor, weight addressing (w_off, sn, sc, flip) and epilogues (NONE, BIAS, BIAS_RELU,
 * RELU_MASK) as dd_gconv_*; needs stride 1, div 1, ostride 1, Cin % 8 == 0, Cout <= 96.  dd_dconv_supported() says
 * whether a descriptor qualifies (callers fall back to dd_gconv_fwd otherwise). */
int32_t dd_dconv_supported(const dd_gconv_desc* d);
int64_t dd_dconv_packed_floats(const dd_gconv_desc* d);
int dd_dconv_pack(const float* w, float* packed, const dd_gconv_desc* d, int64_t w_off, int64_t sn, int64_t sc,
                  int32_t flip, int32_t n_real, int32_t c_real, void* stream);
int dd_dconv_fwd(const float* x, const float* packed, const float* bias, const float* mask, float* y,
                 const dd_gconv_desc* d, int32_t epilogue, void* stream);

/* dd_dconv_fwd that ALSO leaves in colsum[0 .. cout) the per-channel sums of what it wrote: the data gradient of up_conv_(k+1) is the
 * output gradient of up_conv_k, whose bias gradient (autograd's sum over batch and pixels, spatial_bb/components.py:135-137) is exactly
 * those sums -- a separate pass over the tensor otherwise.  Epilogues NONE / RELU_MASK; only layers dd_dconv_colsum_supported accepts
 * (the data gradients of the k7 d7 up-convs with 17-64 output channels). */
int32_t dd_dconv_colsum_supported(const dd_gconv_desc* d, int32_t epilogue, int32_t has_mask);
int64_t dd_dconv_colsum_workspace_bytes(void);
int dd_dconv_fwd_colsum(const float* x, const float* packed, const float* mask, float* y, float* colsum, const dd_gconv_desc* d,
                        int32_t epilogue, void* workspace, int64_t workspace_bytes, void* stream);

/* EXPERIMENT, off by default (csrc/dconv_split.hip): the forward of the k7 d7 dilated ConvTranspose2d layers with Cout > 16
 * (up_conv_1 / up_conv_2, spatial_bb/components.py:135-136) with fp32-equivalent products on the bf16 matrix pipe: every fp32
 * operand is split exactly into three bf16 pieces (hi + mid + lo, truncation: 24 significant bits) and the six cross products
 * >= 2^-23 of the full product are issued on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Same descriptor, packing
 * parameters and epilogues (NONE / BIAS / BIAS_RELU) as dd_dconv_pack + dd_dconv_fwd; the operands are split once, outside:
 *   dd_dconv_split_input   x (NHWC fp32, channels [in_coff, +cin)) -> xs, dd_dconv_split_input_bytes(d) bytes
 *   dd_dconv_split_pack    weights -> packed, dd_dconv_split_packed_bytes(d) bytes
 *   dd_dconv_fwd_split     y = epilogue(conv_transpose(x, w)) -- and, for a descriptor in gather form (pad 0, out = in - d(k-1):
 *                          what dd_dconv_fwd takes for the data gradient), the DATA GRADIENT of the same layers, epilogue NONE or
 *                          RELU_MASK (`mask` = the layer's input, as for dd_dconv_fwd; mask_pass_lo/hi honoured)
 * dd_dconv_split_supported(d): 1 for the descriptors it is built for (k7 d7, cin % 16 == 0, 16 < cout <= 96; forward: full transposed
 * form, in_w <= 320; data gradient: out_w <= 320). */
int32_t dd_dconv_split_supported(const dd_gconv_desc* d);
int64_t dd_dconv_split_input_bytes(const dd_gconv_desc* d);
int64_t dd_dconv_split_packed_bytes(const dd_gconv_desc* d);
int dd_dconv_split_input(const float* x, void* xs, const dd_gconv_desc* d, void* stream);
int dd_dconv_split_pack(const float* w, void* packed, const dd_gconv_desc* d, int64_t w_off, int64_t sn, int64_t sc, int32_t flip,
                        int32_t n_real, int32_t c_real, void* stream);
int dd_dconv_fwd_split(const void* xs, const void* packed, const float* bias, const float* mask, float* y, void* y_planes,
                       const dd_gconv_desc* d, int32_t epilogue, void* stream);      /* y_planes (may be NULL): also write the split image of the
                       OUTPUT, [batch][out_h][cout / 16][out_w][112 B] as dd_dconv_split_rows would make it of y -- the next layer's operand */
/* ... and the WEIGHT GRADIENT of the 96->64 / 64->32 layers from the same split images (xs of the layer's input, gs of dL/dy: produced
 * by dd_dconv_split_input or, without a descriptor, dd_dconv_split_rows -- `rows` image rows of `w` pixels, channels [coff, coff + c) of
 * `cstore`, rows * (c / 16) * w * 112 bytes): dw [cin][cout][7][7], per-workgroup partials in `workspace`, fixed-order fp64 second
 * stage; same contract as dd_dconv_wgrad. */
int dd_dconv_split_rows(const float* x, void* xs, int64_t rows, int32_t w, int32_t cstore, int32_t coff, int32_t c, void* stream);
int32_t dd_dconv_wgrad_split_supported(int32_t k, int32_t dil, int32_t cin, int32_t cout);
int64_t dd_dconv_wgrad_split_workspace_bytes(int32_t cin, int32_t cout);
int dd_dconv_wgrad_split(const void* xs, const void* gs, float* dw, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t gh, int32_t gw,
                         int32_t cout, int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* Weight gradient of the same layers (ConvTranspose2d stride 1, dilation `dil`, kernel k x k, no padding):
 *   dw[c][o][ky][kx] (IOHW, PyTorch's layout) (+)= sum over images and pixels of x[iy][ix][c] * g[iy + dil*ky][ix + dil*kx][o]
 * x [batch,h,w,x_cstore] (channels [x_coff,+cin)), g = dL/dy [batch,gh,gw,g_cstore] (channels [g_coff,+cout)) with
 * gh >= h + dil*(k-1) (output_padding rows are simply never read).  LDS-staged operands, per-workgroup partials in
 * `workspace`, fp64 fixed-order second stage.  Built for the seven (k, dil, cin, cout) combinations of the box heads:
 * dd_dconv_wgrad_supported() says which; the bias gradient is dd_channel_sum of g. */
int32_t dd_dconv_wgrad_supported(int32_t k, int32_t dil, int32_t cin, int32_t cout);
int64_t dd_dconv_wgrad_workspace_bytes(int32_t k, int32_t dil, int32_t cin, int32_t cout);
int dd_dconv_wgrad(const float* x, const float* g, float* dw, int32_t batch, int32_t h, int32_t w, int32_t x_cstore, int32_t x_coff,
                   int32_t cin, int32_t gh, int32_t gw, int32_t g_cstore, int32_t g_coff, int32_t cout, int32_t k, int32_t dil,
                   int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* out[c] (+)= sum over the npix pixels of buf[p, coff + c], c < cout <= 128 (bias gradient of a transposed conv whose
 * weight gradient is taken in the role-swapped form, see gconv.py). */
int64_t dd_channel_sum_workspace_bytes(void);
int dd_channel_sum(const float* buf, float* out, int64_t npix, int32_t cstore, int32_t coff, int32_t cout, int32_t accumulate,
                   void* workspace, void* stream);

/* Last layer of the box heads, ConvTranspose2d(8 -> 1, k2 s2) + sigmoid (spatial_bb/components.py:139,168):
 * x [B,h,w,8] NHWC, wt [8,1,2,2] (IOHW), probs [B,2h,2w].  The backward takes dL/dprobs, applies the sigmoid's
 * derivative, returns dx already masked by (x > 0) (x is a ReLU output), dwt and dbias (deterministic). */
int64_t dd_deconv2x2_c1_workspace_bytes(int32_t c);
int dd_deconv2x2_c1_fwd(const float* x, const float* wt, const float* bias, float* probs, int32_t batch, int32_t h, int32_t w,
                        int32_t c, void* stream);
int dd_deconv2x2_c1_bwd(const float* x, const float* wt, const float* probs, const float* dprobs, float* dx, float* dwt,
                        float* dbias, int32_t batch, int32_t h, int32_t w, int32_t c, void* workspace, void* stream);

/* The decoder's tail (components.py:72-73,91-92): dc3 = relu(ConvTranspose2d(32 -> 32, k2 s2)) in ONE launch -- the four output
 * phases are four column tiles of one GEMM, the input is read once -- x [batch,h,w,32] NHWC, wt [32,32,2,2] (IOHW, the parameter),
 * out [batch,2h,2w,32] NHWC, w >= 32; and dc4 = ConvTranspose2d(32 -> 3, k1) written as the NCHW image the decoder returns:
 * x [batch,h,w,32] NHWC, wt [32,3,1,1], out [batch,3,h,w]. */
int dd_deconv2x2_c32_fwd(const float* x, const float* wt, const float* bias, float* out, int32_t batch, int32_t h, int32_t w,
                         int32_t relu, void* stream);
/* The same into channels [out_coff, out_coff + 32) of an out_cstore-channel NHWC buffer (ss_deconv of the merging heads writes its slice of
 * the 96-channel concat buffer, spatial_bb/components.py:147,159: one launch instead of four phase launches of the generic engine). */
int dd_deconv2x2_c32_fwd_slice(const float* x, const float* wt, const float* bias, float* out, int32_t batch, int32_t h, int32_t w, int32_t relu,
                               int32_t out_cstore, int32_t out_coff, void* stream);

/* Weight (and bias) gradient of ConvTranspose2d(32, 32, k2, s2) -- ss_deconv (spatial_bb/components.py:89,130), the decoder's dc3
 * (components.py:72,91) -- in one launch + a fixed-order second stage: x [batch, h, w, 32] dense, g = channels [g_coff, g_coff + 32) of a
 * [batch, 2h, 2w, g_cstore] NHWC buffer, dw [32, 32, 2, 2] (the layer's weight layout), db [32] or NULL. */
int64_t dd_deconv2x2_c32_wgrad_workspace_bytes(void);
int dd_deconv2x2_c32_wgrad(const float* x, const float* g, float* dw, float* db, int32_t batch, int32_t h, int32_t w, int32_t g_cstore,
                           int32_t g_coff, void* workspace, int64_t workspace_bytes, void* stream);

/* Data gradient of ss_conv -- Conv2d(32, 32, (1, 24), stride (1, 7)), spatial_bb/components.py:88,129 (what autograd computes for
 * F.conv2d's input there) -- in one launch: g [batch, h, gw, 32], w [32, 32, 1, 24] (the layer's weight as it is), dx [batch, h, xw, 32],
 * all dense NHWC fp32, gw = (xw - 24) / 7 + 1 <= 128.  Every element of dx is written (pixels no tap reaches get 0). */
int32_t dd_ssconv_dgrad_supported(int32_t h, int32_t gw, int32_t xw);
int dd_ssconv_dgrad(const float* g, const float* w, float* dx, int32_t batch, int32_t h, int32_t gw, int32_t xw, void* stream);
/* The layer's forward (F.conv2d(x, w, b, stride=(1, 7)) [+ ReLU], spatial_bb/components.py:129,147): x [batch, h, xw, 32], y [batch, h, gw, 32]
 * dense NHWC fp32, gw = (xw - 24) / 7 + 1; bias may be NULL. */
int dd_ssconv_fwd(const float* x, const float* w, const float* bias, float* y, int32_t batch, int32_t h, int32_t xw, int32_t gw, int32_t relu,
                  void* stream);
int dd_conv1x1_c32_c3_nchw(const float* x, const float* wt, const float* bias, float* out, int32_t batch, int32_t h, int32_t w,
                           void* stream);

/* ---- the six strip convolutions of SpatialMappingCNN, one launch each way (spatial_bb/components.py:18-24 the layers -- four
 * Conv2d(3, 32, (1, 50), stride (3, 2)) and two Conv2d(3, 32, (52, 1), stride (3, 2), padding 1) --, :34-65 their inputs: raw views,
 * views rotated by +-90 degrees, views flipped in H and W, :70-73 the 3 x 2 mosaic they are concatenated into).  Every one of them is
 * a 1-D convolution along a row of the RAW view, so rotation, flip and mosaic placement are index arithmetic inside the kernel: the
 * views are read where they lie (`sample_ptrs`: HOST array of `batch` DEVICE pointers, one per sample: fp32 [6][3][H][W], or with
 * u8 != 0 uint8 [6][H][W][3] decoded frames, ToTensor's /255 fused), nothing is re-laid.  Tile order of `weights` / `biases` /
 * `dweights` / `dbiases` (HOST arrays of 6 DEVICE pointers): bl_conv, fl_conv, b_conv, f_conv, br_conv, fr_conv; weights in the
 * modules' own layout [32][3][kh][kw].
 *   fwd    mosaic [batch][3 th][2 tw][32] NHWC fp32 = ReLU(conv + bias) of every tile (th = (H - 1) / 3 + 1, tw = (W - 50) / 2 + 1)
 *          relu_bits (may be NULL): the mosaic's sign words [batch][3 th][2 tw], bit c = channel c > 0 (as dd_relu_sign_bits would give)
 *   wgrad  g = dL/d(mosaic) (already masked by the mosaic's ReLU) -> the six weight and bias gradients; per-wave partial sums in
 *          `workspace` (dd_strip6_wgrad_workspace_bytes()), added in a fixed order: deterministic
 * dd_strip6_supported: the six tiles must come out equal (H even, H % 3 != 0, W = H + 50; the reference's 256 x 306) and W <= 320. */
int32_t dd_strip6_supported(int32_t height, int32_t width);
int dd_strip6_fwd(const void* const* sample_ptrs, int32_t u8, const float* const* weights, const float* const* biases, float* mosaic,
                  uint32_t* relu_bits, int32_t batch, int32_t height, int32_t width, void* stream);
int64_t dd_strip6_wgrad_workspace_bytes(void);
int dd_strip6_wgrad(const void* const* sample_ptrs, int32_t u8, const float* g, float* const* dweights, float* const* dbiases, int32_t batch,
                    int32_t height, int32_t width, void* workspace, int64_t workspace_bytes, void* stream);

/* One camera view of views[B,6,3,H,W] -> NHWC4 [B,H',W',4] with the geometric transform SpatialMappingCNN applies
 * before its strip convs (spatial_bb/components.py:43-65): 0 = none, 1 = rot90(k=1, dims [2,3]) (view 4, "b"),
 * 2 = rot90(k=1, dims [3,2]) (view 1, "f"), 3 = flip([2,3]) (views 5 and 2).  Rotations swap H and W. */
int dd_view_to_nhwc4(const float* views, float* out, int32_t batch, int32_t height, int32_t width, int32_t view,
                     int32_t transform, void* stream);
/* The same from a HOST array of `batch` per-sample DEVICE base pointers (each [6,3,H,W] contiguous), as dd_stitch6_ptrs:
 * BBSpatialRoadMap._run_step receives the collate's tuple (helper.py:22-23) and torch.stack()s it (spatial_w_rm.py:100-103);
 * this skips that 180 MB copy. */
int dd_view_to_nhwc4_u8_ptrs(const unsigned char* const* sample_ptrs, float* out, int32_t batch, int32_t height, int32_t width,
                             int32_t view, int32_t transform, void* stream);      /* per-sample [6,H,W,3] uint8 frames, /255 fused */
int dd_view_to_nhwc4_ptrs(const float* const* sample_ptrs, float* out, int32_t batch, int32_t height, int32_t width,
                          int32_t view, int32_t transform, void* stream);
/* rm_conv_1 (spatial_bb/components.py:80, Conv2d(1, 32, 7, stride=3, dilation=3, padding=1)) reads only the road-map pixels
 * (3u - 1, 3v - 1): dst[b][u][v] = (src[b][stride*u + offset][stride*v + offset], 0, 0, 0) (zero outside the image) is the
 * NHWC4 image on which the same taps form a DENSE k x k convolution -- same products, same order, 1/9 of the bytes.
 * src [batch,h,w] (one channel), dst [batch,oh,ow,4]. */
int dd_subsample_nhwc4(const float* src, float* dst, int32_t batch, int32_t h, int32_t w, int32_t oh, int32_t ow,
                       int32_t stride, int32_t offset, void* stream);
/* The same from a HOST array of `batch` per-sample DEVICE pointers to the road masks as the dataset hands them over --
 * bool / uint8 [h,w] (data_helper.py:137-139), nonzero = 1.0: replaces torch.stack(road_image).float() of
 * spatial_w_rm.py:105 (a stack and a 4x widening cast of the batch's masks) for the one layer that reads the mask. */
int dd_subsample_nhwc4_u8_ptrs(const unsigned char* const* mask_ptrs, float* dst, int32_t batch, int32_t h, int32_t w,
                               int32_t oh, int32_t ow, int32_t stride, int32_t offset, void* stream);
/* ... and that dense convolution itself, 1 -> 32 channels, 7x7: y [batch,sh-6,sw-6,32] = (relu)(conv(taps4 channel 0) + bias),
 * w [32,1,7,7] (the parameter as it is), sw <= 320.  The taps are the K dimension of the GEMM (28 MFMAs per 32 x 32 tile
 * instead of the 98 of a channel padded to 4).  dd_conv1ch_wgrad: dw [32,1,7,7] and dbias [32] (may be NULL) from
 * g = dL/dy [batch,sh-6,sw-6,32] (already masked by the ReLU); deterministic (fp64 fixed-order second stage). */
int dd_conv1ch_fwd(const float* taps4, const float* w, const float* bias, float* y, int32_t batch, int32_t sh, int32_t sw,
                   int32_t relu, void* stream);
/* The same two with rm_conv_1's OUTPUT (and its gradient) in the "phase-major" layout a dilation-3 consumer wants (rm_conv_2,
 * spatial_bb/components.py:81,131): the nine residue classes (i mod 3, j mod 3) of [B][oh][ow][32] as nine images,
 *   P[b * 9 + (i % 3) * 3 + j % 3][i / 3][j / 3][:] = T[b][i][j][:],   ph x pw = ceil(oh / 3) x ceil(ow / 3) pixels each, zero past a class's end.
 * On P a 3x3 convolution with dilation 3 is nine plain 3x3 convolutions, which run on the c2 layer's Winograd kernels (batch 9 B).
 * relu_bits (may be NULL): the sign words of y_phase in the same layout.  dd_phase3_scatter / dd_phase3_gather move a 32-channel slice of a
 * dense NHWC buffer to / from such images, `off` cells in from the corner (1: the interior of a padding-1 convolution's output; gather
 * writes every cell of dst_phase, zero where no pixel maps). */
int dd_conv1ch_fwd_phase3(const float* taps4, const float* w, const float* bias, float* y_phase, uint32_t* relu_bits, int32_t batch,
                          int32_t sh, int32_t sw, int32_t relu, void* stream);
int dd_conv1ch_wgrad_phase3(const float* taps4, const float* g_phase, float* dw, float* dbias, int32_t batch, int32_t sh, int32_t sw,
                            void* workspace, void* stream);
int dd_phase3_scatter(const float* src_phase, float* dst, int32_t batch, int32_t oh, int32_t ow, int32_t src_ph, int32_t src_pw, int32_t off,
                      int32_t dst_cstore, int32_t dst_coff, void* stream);
int dd_phase3_gather(const float* src, float* dst_phase, int32_t batch, int32_t oh, int32_t ow, int32_t dst_ph, int32_t dst_pw, int32_t off,
                     int32_t src_cstore, int32_t src_coff, void* stream);
int64_t dd_conv1ch_wgrad_workspace_bytes(void);
int dd_conv1ch_wgrad(const float* taps4, const float* g, float* dw, float* dbias, int32_t batch, int32_t sh, int32_t sw,
                     void* workspace, void* stream);
/* dst[p][dst_coff + c] = src[p][src_coff + c], c < channels, p < npix: a channel slice of one NHWC buffer into a channel slice of
 * another -- the `torch.cat((ssr, space_rep, rm), dim=1)` of the merging heads (components.py:109,159) and the slice of its
 * gradient; everything a multiple of 4 channels. */
int dd_copy_channels(const float* src, float* dst, int64_t npix, int32_t channels, int32_t src_cstore, int32_t src_coff,
                     int32_t dst_cstore, int32_t dst_coff, void* stream);
/* The same between rectangular windows of the two buffers: src [B,src_mem_h,src_mem_w,src_cstore], window origin (src_y0, src_x0), and
 * likewise dst; h x w pixels, `channels` channels from src_coff to dst_coff (the interior of a padded activation into a concat slice). */
int dd_copy_channels_window(const float* src, float* dst, int32_t batch, int32_t h, int32_t w, int32_t channels, int32_t src_mem_h,
                            int32_t src_mem_w, int32_t src_y0, int32_t src_x0, int32_t src_cstore, int32_t src_coff, int32_t dst_mem_h,
                            int32_t dst_mem_w, int32_t dst_y0, int32_t dst_x0, int32_t dst_cstore, int32_t dst_coff, void* stream);
/* out[i] = a[i] + b[i] (gradient fan-in of the shared views / feature), n % 4 == 0. */
int dd_add(const float* a, const float* b, float* out, int64_t n, void* stream);
/* Mean binary cross-entropy on PROBABILITIES (spatial_w_rm.py:131 F.binary_cross_entropy; log clamped at -100 like
 * torch) with optional dprobs = (-(t/p) + (1-t)/(1-p)) * grad_scale / n. */
int dd_bce_probs(const float* probs, const float* target, float* loss_out, float* dprobs, int64_t n, float grad_scale,
                 void* workspace, void* stream);

/* ---- skinny GEMMs of the dense head (K7): nn.Linear with the weight kept as [out=N, in=K] ------------
 * Replace F.linear / its autograd for DenseBlock.fc1 (components.py:105), Encoder.fc_z_out (components.py:51)
 * and the roadmap head (roadmap_bce_v2.py:75).  M = batch rows (<= 64), N and K multiples of 4.
 *   fwd   y[M,N]  = x[M,K] w[N,K]^T + bias[N]     (bias may be NULL)
 *   dgrad dx[M,K] = dy[M,N] w[N,K]
 *   wgrad dw[N,K] = dy[M,N]^T x[M,K];  dbias[N] = sum_m dy[m,:]   (dbias may be NULL)
 * fwd/dgrad split the long contraction over workgroups and reduce the partial slabs in a fixed order
 * (deterministic); `workspace` must hold dd_linear_workspace_bytes(m, n, k) bytes. */
int64_t dd_linear_workspace_bytes(int32_t m, int32_t n, int32_t k);
int dd_linear_fwd(const float* x, const float* w, const float* bias, float* y, int32_t m, int32_t n, int32_t k,
                  void* workspace, int64_t workspace_bytes, void* stream);
int dd_linear_dgrad(const float* dy, const float* w, float* dx, int32_t m, int32_t n, int32_t k, void* workspace,
                    int64_t workspace_bytes, void* stream);
int dd_linear_wgrad(const float* dy, const float* x, float* dw, float* dbias, int32_t m, int32_t n, int32_t k,
                    void* stream);
/* dbias[N] = sum_m dy[m,:] alone: the bias gradient of a layer whose weight gradient is formed inside its optimizer pass
 * (dd_adam_step_rankb) while its bias is updated the ordinary way (components.py:105, roadmap_bce_v2.py:75). */
int dd_column_sum(const float* dy, float* dbias, int32_t m, int32_t n, void* stream);

/* Threat score tp / (sum a + sum b - tp), tp = sum a*b (reference src/utils/helper.py:74-77); round_b != 0 scores
 * round(b) (roadmap_bce_v2.py:140).  One pass, deterministic. */
int64_t dd_threat_score_workspace_bytes(void);
int dd_threat_score(const float* a, const float* b, float* out, int64_t n, int32_t round_b, void* workspace, void* stream);

/* ---- optimizer -----------------------------------------------------------------------------
 * torch.optim.Adam step (autoencoder.py:119-120, roadmap_bce_v2.py:154-157; no weight decay,
 * no amsgrad) over one flat fp32 buffer: p, g, m, v of n elements; step >= 1.  m and v are computed exactly as torch does
 * (multiplies and fused multiply-adds); the update term m_hat / (sqrt(v_hat) + eps) uses the hardware square root and
 * reciprocal (1 ulp each): p is within the rounding of a term 2e-7 relative off torch's (tests: 1e-6). */
int dd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, int32_t step, float grad_scale, void* stream);
/* The same update for `count` tensors in one launch (the model's many small parameters: biases, BatchNorm vectors,
 * 3x3 filters -- torch.optim.Adam loops over them, roadmap_bce_v2.py:154-157).  `tensors` is a HOST array; all share
 * `step`.  Meant for tensors of up to a few thousand elements (one element per thread). */
typedef struct dd_adam_tensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} dd_adam_tensor;
int dd_adam_step_multi(const dd_adam_tensor* tensors, int32_t count, float lr, float beta1, float beta2, float eps,
                       int32_t step, float grad_scale, void* stream);

/* Rank-B form of the same update for an nn.Linear weight p [n][k] (reference: DenseBlock.fc1 components.py:27,70,105; the road-map head
 * roadmap_bce_v2.py:50,75; optimizer autoencoder.py:119-120, roadmap_bce_v2.py:154-157): the gradient is NOT an input.  It is formed
 * inside the pass, dW[o][i] = sum_b dy[b][o] x[b][i], from the layer's output gradient dy [rows][n] and input x [rows][k] (rows = the
 * batch, or world x batch for all-gathered factors) on the fp32 matrix cores, and consumed by the update of the same lane: p, m, v are
 * read and written once each, no dW tensor exists (dd_linear_wgrad + dd_adam_step: two more passes over the tensor).  bias_p / bias_m /
 * bias_v (all three or none): the layer's bias [n] and its moments, updated in the same launch from the column sums of dy.
 * k % 4 == 0; p, m, v, x 16-byte aligned; same arithmetic per element as dd_adam_step (gradient: fp32 MFMA accumulation over rows). */
int dd_adam_step_rankb(float* p, float* m, float* v, const float* dy, const float* x, int32_t rows, int32_t n, int32_t k,
                       float* bias_p, float* bias_m, float* bias_v, float lr, float beta1, float beta2, float eps, int32_t step,
                       float grad_scale, void* stream);

/* ---- Winograd F(2,3) along x for the 32 -> 32 stride-1 layer (c2, components.py:20): the same outputs as
 * dd_conv_fwd_relu_bits / dd_conv_dgrad_relu_bits from 2/3 of the multiplies (4 per output-pixel pair and tap row
 * instead of 6), still exact fp32 arithmetic (the transforms are fp32 adds and one halving; results differ from the
 * direct form by summation order only, ~1e-6 relative).  kind 0 = forward, 1 = stride-1 data gradient. */
int64_t dd_conv_wino_packed_floats(const dd_conv_desc* d);
int dd_conv_wino_pack(const float* w_oihw, float* packed, const dd_conv_desc* d, int32_t kind, void* stream);
int dd_conv_wino_fwd_relu_bits(const float* x, const float* packed, const float* bias, float* y, uint32_t* relu_bits,
                               const dd_conv_desc* d, void* stream);
int dd_conv_wino_dgrad_relu_bits(const float* dy, const float* packed, const uint32_t* relu_bits, float* dx,
                                 const dd_conv_desc* d, void* stream);
/* The 2-D form F(2x2,3x3) of the forward and data gradient: 16 multiplies per 2x2 output tile instead of 36 (one wave
 * per SIMD: four ring rows and 64 KB of transformed weights per workgroup).  Same contract as the 1-D entry points. */
int64_t dd_conv_wino2_packed_floats(const dd_conv_desc* d);
int dd_conv_wino2_pack(const float* w_oihw, float* packed, const dd_conv_desc* d, int32_t kind, void* stream);
int dd_conv_wino2_fwd_relu_bits(const float* x, const float* packed, const float* bias, float* y, uint32_t* relu_bits,
                                const dd_conv_desc* d, void* stream);
int dd_conv_wino2_dgrad_relu_bits(const float* dy, const float* packed, const uint32_t* relu_bits, float* dx,
                                  const dd_conv_desc* d, void* stream);
/* The same data gradient CONSUMED IN PLACE by the 3 -> 32 layer's weight gradient (components.py:19,41: c1): instead of
 * writing g1 = dL/d(c1 output) (1.93 GB at bs 32) for dd_conv_wgrad to read back, the kernel multiplies every masked
 * output tile with the matching patch of the stitched input x_nhwc4 [B,H,W,4] while it is in registers and returns
 * dW1 [32,3,3,3] (OIHW) and dbias1 [32].  Same arithmetic as dd_conv_wino2_dgrad_relu_bits + dd_conv_wgrad up to the
 * order of the fp32 sums. */
int64_t dd_conv_wino2_dgrad_w1_workspace_bytes(const dd_conv_desc* d);
int dd_conv_wino2_dgrad_w1(const float* dy, const float* packed, const uint32_t* relu_bits, const float* x_nhwc4,
                           float* dw1_oihw, float* dbias1, void* workspace, int64_t workspace_bytes,
                           const dd_conv_desc* d, void* stream);
/* ... and of the weight gradient: F(3x3,2x2), 16 accumulators S[u][v] over tiles, dW = A^T S A in the reduce kernel */
int64_t dd_conv_wino2_wgrad_workspace_bytes(const dd_conv_desc* d);
int dd_conv_wino2_wgrad(const float* x, const float* dy, float* dw_oihw, float* dbias, void* workspace,
                        int64_t workspace_bytes, const dd_conv_desc* d, void* stream);
/* The same in two calls, so that the second (two small reduce kernels over the per-workgroup partials in `workspace`) can
 * run on another stream beside whatever follows the first: nothing in the backward waits for dW. */
int dd_conv_wino2_wgrad_partials(const float* x, const float* dy, void* workspace, int64_t workspace_bytes,
                                 const dd_conv_desc* d, void* stream);
int dd_conv_wino2_wgrad_finish(void* workspace, int64_t workspace_bytes, float* dw_oihw, float* dbias,
                               const dd_conv_desc* d, void* stream);
/* weight + bias gradient of the same layer by F(3,2) along x (as dd_conv_wgrad: deterministic two-stage reduction) */
int64_t dd_conv_wino_wgrad_workspace_bytes(const dd_conv_desc* d);
int dd_conv_wino_wgrad(const float* x, const float* dy, float* dw_oihw, float* dbias, void* workspace,
                       int64_t workspace_bytes, const dd_conv_desc* d, void* stream);

/* ---- bf16 mixed precision (BASELINE config 5: 6x3x512x612 inputs, bf16) ---------------------------------
 * The encoder conv stack (components.py:19-21,41-43) with bf16 operands on the bf16 matrix cores and fp32
 * accumulation: activations and activation gradients are NHWC bf16 (raw uint16 bit patterns in this ABI), rounded
 * to nearest even exactly once where they are written; weights, biases and their gradients stay fp32.  The FC tail,
 * BatchNorm, losses and Adam are the fp32 entry points above.  The reference has no mixed-precision mode: the
 * contract is "torch autocast equivalent" -- a conv evaluated on bf16-rounded inputs and weights with fp32
 * accumulation, its output rounded to bf16 (oracle: oracle/bf16_parts.py). */
int dd_stitch6_bf16(const float* views, uint16_t* wide_nhwc4, int32_t batch, int32_t height, int32_t width, void* stream);
/* The same from a HOST array of `batch` per-sample DEVICE base pointers (each [6,3,H,W] fp32 contiguous), as dd_stitch6_ptrs:
 * no torch.stack of the collate's tuple (roadmap_bce_v2.py:55) in front of the bf16 path either. */
int dd_stitch6_bf16_u8_ptrs(const unsigned char* const* sample_ptrs, uint16_t* wide_nhwc4, int32_t batch, int32_t height, int32_t width,
                            void* stream);      /* per-sample [6,H,W,3] uint8 frames: /255 (true division), then the bf16 rounding */
int dd_stitch6_bf16_ptrs(const float* const* sample_ptrs, uint16_t* wide_nhwc4, int32_t batch, int32_t height, int32_t width,
                         void* stream);
int64_t dd_conv_bf16_packed_elems(const dd_conv_desc* d);                 /* uint16 elements of an operand image */
/* kind as dd_conv_pack: 0 forward, 1 stride-1 data gradient, 2 stride-2 data gradient */
int dd_conv_bf16_pack(const float* weight, const dd_conv_desc* d, int32_t kind, uint16_t* packed, void* stream);
/* y = bf16(relu(conv(x) + bias)); relu_bits (nullable) [B,Ho,Wo] uint32 = mask of (y > 0) over the 32 channels */
int dd_conv_bf16_fwd(const uint16_t* x, const uint16_t* packed, const float* bias, uint16_t* y, uint32_t* relu_bits,
                     const dd_conv_desc* d, void* stream);
/* dx = bf16(dgrad(dy) * bit(channel) of relu_bits[pixel]) with relu_bits [B,H,W] of the conv's INPUT activation */
int dd_conv_bf16_dgrad(const uint16_t* dy, const uint16_t* packed, const uint32_t* relu_bits, uint16_t* dx,
                       const dd_conv_desc* d, void* stream);
int64_t dd_conv_bf16_wgrad_workspace_bytes(const dd_conv_desc* d);
/* dweight [32,Cin,3,3], dbias [32] fp32 from bf16 x and dy, fp32 accumulation, deterministic two-stage reduction */
int dd_conv_bf16_wgrad(const uint16_t* x, const uint16_t* dy, float* dweight, float* dbias, const dd_conv_desc* d,
                       void* workspace, int64_t workspace_bytes, void* stream);
/* max_pool1d(4) over the NCHW-flattened feature (components.py:46-47) from NHWC bf16; H*W % 4 == 0, C % 4 == 0 */
int dd_pool4_bf16_fwd(const uint16_t* feat, float* pooled, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);
/* its backward with the ReLU in front of the pool fused (feat > 0), gradient written in bf16 */
int dd_pool4_relu_bf16_bwd(const float* dpooled, const uint16_t* feat, uint16_t* dfeat, int32_t batch, int32_t h,
                           int32_t w, int32_t c, void* stream);
/* The same pair for C == 32 with the backward's routing decided in the forward, as dd_pool4_fwd_idx / dd_pool4_idx_relu_bwd do for fp32:
 * idx holds dd_pool4_bf16_idx_elems() 16-bit words, one per (window, 4 channels), 4 bits per channel = first maximum | (max > 0) << 2
 * (torch keeps the earliest index on ties; the ReLU in front of the pool is the "> 0" bit).  The backward reads dpooled and idx only --
 * the feature map is not kept for it -- and both kernels move 64-window tiles through LDS so that each side of the NHWC <-> NCHW-flat
 * change of order is read and written in whole 128-byte lines.  idx may be NULL in the forward (no codes written).  feat / dfeat
 * 16-byte aligned.  Results equal dd_pool4_bf16_fwd / dd_pool4_relu_bf16_bwd bit for bit. */
int64_t dd_pool4_bf16_idx_elems(int32_t batch, int32_t h, int32_t w, int32_t c);
int dd_pool4_bf16_fwd_idx(const uint16_t* feat, float* pooled, uint16_t* idx, int32_t batch, int32_t h, int32_t w, int32_t c,
                          void* stream);
int dd_pool4_idx_relu_bf16_bwd(const float* dpooled, const uint16_t* idx, uint16_t* dfeat, int32_t batch, int32_t h, int32_t w,
                               int32_t c, void* stream);
int dd_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);     /* n % 4 == 0, round to nearest even */
int dd_bf16_to_f32(const uint16_t* src, float* dst, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DD_HOTPATH_H */
