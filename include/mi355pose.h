/*
 * mi355pose.h — C ABI of libmi355pose.so: the MI355X (gfx950) kernels underneath the
 * domain-adaptive hand-pose training / evaluation hot path.
 *
 * The reference (CVlab315/Domain-Adaptative-Hand-Pose-Estimation) has NO native / FFI
 * boundary: it is 100 % Python on torch.nn.  Its drop-in surface is the
 * Python API (train1.py / test.py CLIs, uda.model class names, state_dict layout), which the
 * host package mirrors.  This header is the NEW internal boundary beneath that Python API:
 * one entry point per fused op and direction, each replacing the ATen op sequence the
 * reference issues at the cited file:line (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers + sizes; every buffer is DEVICE memory allocated by the caller;
 *     the library never allocates, never synchronises, never owns memory;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default);
 *     every entry point is hipGraph-capturable;
 *   - returns 0 on success, a negative MI355_E* code otherwise; text via mi355_last_error()
 *     (thread-local); no exceptions cross the ABI; re-entrant (autograd worker threads);
 *   - activations are NHWC ("rows x channels"), dtype MI355_F32 or MI355_BF16;
 *     heat-maps (21 channels) are NCHW fp32 rows of H*W, as the reference's losses see them;
 *   - scalars that change every iteration (GL lambda, learning rate, upstream loss grad)
 *     are read from DEVICE floats so that a captured graph replays with fresh values.
 *   - "conv-form": every conv-like layer is described as the convolution
 *         y[N,Ho,Wo,Co] = conv(x[N,Hi,Wi,Ci], w[Co][kh][kw][Ci], stride, pad)
 *     nn.Conv2d uses it directly; nn.ConvTranspose2d(Cin,Cout) is the ADJOINT of the
 *     conv-form with Co=Cin_t, Ci=Cout_t (its forward is conv_dgrad, its input-gradient is
 *     conv_fwd, its weight-gradient is conv_wgrad with the roles of x/dy swapped).
 */
#ifndef MI355POSE_H
#define MI355POSE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_F32 0
#define MI355_BF16 1
#define MI355_FP8 2   /* conv descriptors of the *_fp8 entry points only: fp8 operands, fp32 accumulate, bf16 results */

#define MI355_OK 0
#define MI355_EINVAL (-1)   /* bad argument / unsupported shape */
#define MI355_ELAUNCH (-2)  /* HIP launch error */
#define MI355_EWORKSPACE (-3)

typedef struct {
  int N, Hi, Wi, Ci; /* conv-form input  (NHWC)                       */
  int Ho, Wo, Co;    /* conv-form output (NHWC); = (Hi + 2 pad - kh) / stride + 1, or -- unit stride, mi355_conv_fwd* and
                        mi355_conv_wgrad only -- any smaller size: the top-left Ho x Wo crop of that output */
  int kh, kw, stride, pad;
  int dtype;         /* MI355_F32 | MI355_BF16 (activations+packed weights) | MI355_FP8 (mi355_conv_*_fp8 only) */
} mi355_conv_desc;

int mi355_version(void);
const char* mi355_last_error(void);

/* ---------------------------------------------------------------- convolutions (MFMA implicit GEMM)
 * Replace torch.nn.Conv2d / ConvTranspose2d forward+backward issued by
 *   uda/model/resnet.py:23-38 (torchvision ResNet stem/blocks), uda/model/pose_resnet2.py:33-41
 *   (3x ConvTranspose2d 4x4 s2 p1), uda/model/regda_7.py:4906-4929 (_make_head),
 *   :4513-4514,4551,4561 (make_head), :4588-4589,4627,4637 (make_head2).
 * w  : packed weights [Co][kh][kw][Ci]  in `dtype`     (forward operand)
 * wT : packed weights [Ci][kh][kw][Co]  in `dtype`     (dgrad operand)
 * bias: fp32 [Co] or NULL.  residual: tensor shaped like y (added before the store) or NULL.
 * scale_dev: NULL or device float multiplied into the result (folds utils/gl.py:18, grad*coeff).
 * accumulate!=0 : dx += result (several consumers of one tensor), else dx = result.
 * dw: fp32 [Co][kh][kw][Ci]; accumulate!=0 adds into dw.  ws: scratch of at least
 * mi355_conv_wgrad_workspace() bytes.
 */
int mi355_conv_fwd(const mi355_conv_desc* d, const void* x, const void* w, const float* bias,
                   const void* residual, void* y, void* stream);
int mi355_conv_dgrad(const mi355_conv_desc* d, const void* dy, const void* wT, const float* bias,
                     const float* scale_dev, int accumulate, void* dx, void* stream);
/* 1x1 / unit-stride bf16 convs (forward and input gradient) can run on a second kernel, the persistent pipelined GEMM of
 * csrc/pgemm.hip (same results bit for bit): mode 0 never (default: end to end it measured no gain), 1 where
 * it was faster per layer in isolation, 2 wherever the launch fits it (tests / A-B runs), -1 back to the environment's choice
 * (MI355_PGEMM).  Returns the previous setting. */
int mi355_set_pgemm(int mode);
/* Inference forms: y = act(conv(x) + bias + residual), dx = act(dgrad(dy) + bias) (the ConvTranspose2d forward), act = ReLU when
 * relu != 0.  The caller folds an eval-mode BatchNorm that follows into the operands (w * gamma / sqrt(var + eps) per output
 * channel, bias = beta - mean * that scale): conv -> BN -> (+identity) -> ReLU of resnet.py / pose_resnet2.py:33-41 /
 * regda_7.py:4906-4929 becomes one launch in test.py's forward-only path. */
int mi355_conv_fwd_act(const mi355_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual, int relu,
                       void* y, void* stream);
int mi355_conv_dgrad_act(const mi355_conv_desc* d, const void* dy, const void* wT, const float* bias, int relu, void* dx,
                         void* stream);
/* dx <- result + (bit of acc_mask set ? dx : 0).  acc_mask: the ReLU bit mask of mi355_bn_train_fwd over a tensor shaped like
 * dx.  The fork of a residual block (torchvision Bottleneck / BasicBlock: out = relu(bn(...) + identity)): dx holds the
 * gradient of the block's OUTPUT ReLU input, still unmasked, and the conv is the block's first one (same input as the
 * identity branch); replaces the masked copy mi355_bn_bwd would write as `dresidual` and the add autograd would run. */
int mi355_conv_dgrad_masked_acc(const mi355_conv_desc* d, const void* dy, const void* wT, const float* scale_dev, void* dx,
                                const void* acc_mask, void* stream);
/* Convolution / transposed-convolution forward with the BatchNorm batch statistics of its OUTPUT computed in the
 * epilogue (the nn.BatchNorm2d that follows every conv of resnet.py / pose_resnet2.py:33-41 / regda_7.py:4906-4929 in
 * training mode): partial[slice][Co|Ci][n, mean, M2] per output-row tile, *nslices = slices written (0: this launch
 * could not fuse them -- the caller then runs mi355_bn_train_fwd).  partial_bytes >= mi355_conv_stats_bytes(rows, C). */
size_t mi355_conv_stats_bytes(long rows, int C);
int mi355_conv_fwd_stats(const mi355_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                         float* partial, size_t partial_bytes, int* nslices, void* stream);
int mi355_conv_dgrad_stats(const mi355_conv_desc* d, const void* dy, const void* wT, void* dx, float* partial,
                           size_t partial_bytes, int* nslices, void* stream);
/* Concatenated-K forward: y = conv(x, w) + x2 * w2^T + bias + bias2 as ONE implicit GEMM (K = kh*kw*Ci + c2).  Replaces the
 * `heatmap_conv(heatmap) + feature_conv(feature)` pair at the entry of the multiscale-fusion heads (reference
 * uda/model/regda_7.py:4573-4581 make_head.forward, :4649-4662 make_head2.forward) by one launch: x2 [N*Ho*Wo][c2] is the
 * heat-map operand re-laid as NHWC at the OUTPUT resolution (mi355_nchw_to_nhwc, channels zero-padded to c2), w2 [Co][c2]
 * the 1x1 heat-map weights, both in the descriptor's dtype (bf16 / fp32); c2 a multiple of the 16-byte chunk, <= 64 (bf16) /
 * 32 (fp32).  bias / bias2 nullable.  partial / nslices: nullable pair, BatchNorm statistics of y as in mi355_conv_fwd_stats. */
int mi355_conv_fwd_cat(const mi355_conv_desc* d, const void* x, const void* w, const float* bias, const void* x2,
                       const void* w2, const float* bias2, int c2, void* y, float* partial, size_t partial_bytes,
                       int* nslices, void* stream);
/* The same idea for the BACKWARD pass: a GEMM whose output is the dy of a BatchNorm (the input gradient of the conv
 * that consumed the BatchNorm's output) leaves that BatchNorm's backward reduction (sum dy_eff, sum dy_eff * xhat) per
 * output-row tile in partial[slice][C][2]; mi355_bn_bwd_partials then skips its reduction pass.  bn: the BatchNorm's
 * saved forward state; relu != 0 masks dy with y > 0 (y given) or with the mask recomputed from x (y NULL). */
typedef struct mi355_bn_bwd_src {
  const void* x; const void* y; const float* gamma; const float* beta; const float* save_mean; const float* save_invstd;
  int relu;
} mi355_bn_bwd_src;
int mi355_conv_dgrad_bnbwd(const mi355_conv_desc* d, const void* dy, const void* wT, const float* scale_dev, int accumulate,
                           void* dx, const mi355_bn_bwd_src* bn, float* partial, size_t partial_bytes, int* nslices,
                           void* stream);
int mi355_conv_fwd_bnbwd(const mi355_conv_desc* d, const void* x, const void* w, void* y, const mi355_bn_bwd_src* bn,
                         float* partial, size_t partial_bytes, int* nslices, void* stream);
/* ---- fp8 operand path (BASELINE config 5: "fp8 conv MFMA path"; reference layers: the K-heavy 3x3 / 4x4 convolutions
 * of uda/model/regda_7.py:4906-4929, pose_resnet2.py:33-41 and the torchvision Bottleneck 3x3).  OCP formats:
 * fmt 0 = e4m3 (activations, weights), fmt 1 = e5m2 (gradients).  Per-tensor scaling state = 4 floats on the device:
 * {scale, descale = 1/scale, amax bits (max |x| seen since the last update), descale of the latest weight pack made with it}.
 * A copy must be descaled with the scale it was MADE with: the state is refreshed at every optimizer step and a copy may be
 * consumed after that, so activation copies carry their own record (write = 2 below) and weight packs leave theirs in state[3].
 *
 * mi355_fp8_quantize: q[i] = saturate_fmt(x[i] * state[0]) for i < n (n a multiple of 16; x bf16 or fp32), and
 *   state[2] = max(state[2], max |x|) -- with write = 0 only the amax is taken (just-in-time scaling: amax, update, quantise);
 *   write = 2: q has 16 more bytes behind its n, which receive this copy's own record {scale, descale, 0, 0} (usable wherever
 *   a state's descale is read).
 * mi355_fp8_update_scale: for each of n states (stride_floats apart): scale = 2^floor(log2(fmt_max / (amax * 2^margin)))
 *   (kept when no amax was recorded; 1 if never set), descale = 1/scale, amax = 0.
 * mi355_pack_weights_fp8: fp32 master [O][T][I] -> e4m3 wf [O][T][I] and wt [I][T][O] (O, I multiples of 32).  margin >= 0:
 *   just-in-time per-tensor scale from amax(|w|), left in `state`; margin < 0: the scale already in `state` (delayed
 *   scaling).  Either way amax(|w|) is recorded in state[2] for the next mi355_fp8_update_scale, and state[3] = the descale
 *   this pack was made with (what its consumers must read). */
int mi355_fp8_quantize(const void* x, void* q, float* state, long n, int src_dtype, int fmt, int write, void* stream);
int mi355_fp8_update_scale(float* states, int n, int stride_floats, int fmt, int margin, void* stream);
int mi355_pack_weights_fp8(const float* w, void* wf, void* wt, float* state, int O, int T, int I, int margin, void* stream);
/* The same for many weights in one launch with the scales already in their states (delayed scaling): items in DEVICE memory,
 * blk0 = first block of the item = sum over earlier items of (O/32)*(I/32)*T. */
typedef struct mi355_pack8_item { const float* w; void* wf; void* wt; float* state; int O, T, I, blk0; } mi355_pack8_item;
int mi355_pack_weights_fp8_batched(const mi355_pack8_item* items_dev, int nitems, int total_blocks, void* stream);
/* mi355_conv_fwd / mi355_conv_dgrad with fp8 operands (d->dtype = MI355_FP8; channels contracted over: a multiple of 128):
 *   y  (bf16) = (conv(x8, w8)  * *descale_x  * *descale_w + bias) [+ residual]
 *   dx (bf16) = (dgrad(dy8, wT8) * *descale_dy * *descale_w) * (*scale_dev, optional) [+ dx when accumulate]
 * partial / nslices (nullable): BatchNorm statistics of the result from the epilogue, as mi355_conv_fwd_stats. */
int mi355_conv_fwd_fp8(const mi355_conv_desc* d, const void* x8, int x_fmt, const void* w8, const float* descale_x,
                       const float* descale_w, const float* bias, const void* residual, void* y, float* partial,
                       size_t partial_bytes, int* nslices, void* stream);
int mi355_conv_dgrad_fp8(const mi355_conv_desc* d, const void* dy8, int dy_fmt, const void* wT8, const float* descale_dy,
                         const float* descale_w, const float* scale_dev, int accumulate, void* dx, float* partial,
                         size_t partial_bytes, int* nslices, void* stream);
/* The 3x3 / unit-stride fp8 launches (forward and input gradient) use the variant that stages one operand tile per kernel ROW
 * and reads it shifted for the three taps (a third fewer bytes per MFMA) from `min_tiles` 128x128 output tiles on: 0 never,
 * 1 wherever the shape allows (tests), -1 back to the environment's choice (MI355_FP8_KW3, default 1024).  Returns the previous
 * setting.  Same arithmetic in another summation order (fp32 accumulate). */
long mi355_set_fp8_kw3(long min_tiles);
/* Weight gradient of the K-heavy convs from the fp8 copies of their operands ('fp8' mode; replaces the bf16 mi355_conv_wgrad
 * for the layers of resnet.py:92-107 (Bottleneck conv2, stride 1 and 2), pose_resnet2.py:33-41 (4x4 transposed convs, in
 * conv-form: x = the output gradient, dy = the input) and regda_7.py:4906-4929 whose forward and input gradient already run on
 * fp8 operands): dw[Co][kh][kw][Ci] (fp32) (+)= descale_x * descale_dy * sum_m dy8[m][o] * x8[pixel(m, tap)][c].
 * x8: [N][Hi][Wi][Ci], dy8: [N][Ho][Wo][Co], each e4m3 (format 0) or e5m2 (1); `d` an fp8 descriptor that is either 3x3 /
 * stride 1 / pad 1 with a power-of-two width >= 8, or 3x3 / 4x4 / stride 2 / pad 1 with even extents and a power-of-two output
 * width in [8, 64]; channel counts multiples of 16.  Workspace / accumulate as mi355_conv_wgrad. */
size_t mi355_conv_wgrad_fp8_workspace(const mi355_conv_desc* d);
int mi355_conv_wgrad_fp8(const mi355_conv_desc* d, const void* x8, int x_fmt, const void* dy8, int dy_fmt, const float* descale_x,
                         const float* descale_dy, float* dw, int accumulate, void* ws, size_t ws_bytes, void* stream);
size_t mi355_conv_wgrad_workspace(const mi355_conv_desc* d);
int mi355_conv_wgrad(const mi355_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate,
                     void* ws, size_t ws_bytes, void* stream);
/* Many weight gradients in one call (the layers of one ResNet stage, collected during its backward pass).  Small problems --
 * each offers too few output tiles to fill 256 CUs, so mi355_conv_wgrad splits its pixel range 16..48 ways into fp32 slabs -- are
 * launched TOGETHER (up to 24 per launch, split counts chosen for the group, one grouped slab reduction); problems that have a
 * specialised kernel or fill the chip alone go through mi355_conv_wgrad unchanged.  items: HOST array; results as mi355_conv_wgrad.
 * NOTE: the non-grouped problems reuse `ws` one after the other (stream order), the grouped ones get disjoint regions. */
typedef struct mi355_wgrad_item { mi355_conv_desc d; const void* x; const void* dy; float* dw; int accumulate; int pad_; } mi355_wgrad_item;
size_t mi355_conv_wgrad_grouped_workspace(const mi355_wgrad_item* items, int n);
int mi355_conv_wgrad_grouped(const mi355_wgrad_item* items, int n, void* ws, size_t ws_bytes, void* stream);
/* fp32 master [O][T][I] -> packed `dtype` copies: wf [O][T][Ipad] (cast) and/or wt [Ipad][T][O] (transposed);
 * channels I..Ipad-1 are zero (the 3-channel stem is padded to one 16-byte chunk). */
int mi355_pack_weights(const float* w, void* wf, void* wt, int O, int T, int I, int Ipad, int dtype, void* stream);
/* The same for many weights in one launch (all convs of an optimizer group, right after its step).  items: DEVICE array;
 * blk0 = first block of the item = sum over earlier items of ceil(Ipad/32)*ceil(O/32)*T; total_blocks = that sum over all. */
typedef struct mi355_pack_item { const float* w; void* wf; void* wt; int O, T, I, Ipad, blk0, pad_; } mi355_pack_item;
int mi355_pack_weights_batched(const mi355_pack_item* items_dev, int nitems, int total_blocks, int dtype, void* stream);
/* dbias[C] (=|+=) column sums of dy[rows][C] (bias gradient of the biased head convs). */
size_t mi355_colsum_workspace(long rows, int C);
int mi355_colsum(const void* dy, float* out, long rows, int C, int dtype, int accumulate, void* ws,
                 size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- BatchNorm2d (+ReLU, +residual add)
 * Replace nn.BatchNorm2d (+nn.ReLU, + the residual add of torchvision blocks): 66 instances for
 * ResNet-50; train mode: batch statistics, running-stat update with momentum 0.1 and
 * the UNBIASED variance, eps 1e-5.
 *   y = relu?( (x-mean)*invstd*gamma + beta + residual? )
 * save_mean / save_invstd: fp32 [C] outputs consumed by mi355_bn_bwd.
 * stat_updates: how many times the running-stat momentum update (and num_batches_tracked += 1) is applied: 1 normally;
 *   2 when one forward stands for two identical forwards of the reference (train1.py:405 and :441 run the unchanged
 *   backbone / neck / main head twice on the same target batch).
 * bwd: dy_eff = relu ? dy*(y>0) : dy ; dx = gamma*invstd*(dy_eff - mean(dy_eff) - xhat*mean(dy_eff*xhat));
 *      dresidual (nullable) = dy_eff ; dgamma/dbeta (=|+=).
 *      With relu and y == NULL the mask is recomputed from x (valid when the forward had no residual): one
 *      tensor read less in each backward pass.
 * relu_mask (nullable, all four entry points): [rows][C / chunk] bytes, chunk = 8 (bf16) / 4 (fp32) channels, bit e of a
 *      byte = (y > 0) of channel chunk*chunk_size + e.  The forward writes it; a backward given the mask takes the ReLU
 *      mask from it and reads neither y nor beta: 1/16 of the bytes of y in each of the two backward passes of a
 *      BatchNorm + residual + ReLU (the last BatchNorm of every residual block).
 * q8_out / q8_state (nullable, both or none; bf16 only; 'fp8' compute mode): an fp8 copy of the result written on the side --
 *      forward: q8 = e4m3(y * q8_state[0]), backward: q8 = e5m2(dx * q8_state[0]) -- and max |.| recorded in q8_state[2]
 *      (mi355_fp8_quantize semantics): the consuming fp8 convolution then needs no quantisation pass of its own.
 */
size_t mi355_bn_workspace(long rows, int C);
int mi355_bn_train_fwd(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked,
                       float* save_mean, float* save_invstd, long rows, int C, float eps, float momentum,
                       int stat_updates, int relu, int dtype, void* ws, size_t ws_bytes, void* relu_mask, void* q8_out,
                       float* q8_state, void* stream);
/* mi355_bn_train_fwd without its statistics pass: the partials come from mi355_conv_fwd_stats / _dgrad_stats.
 * scale_shift: 2*C floats of scratch. */
int mi355_bn_train_fwd_partials(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                float* save_mean, float* save_invstd, long rows, int C, float eps, float momentum,
                                int stat_updates, int relu, int dtype, const float* partial, int nslices,
                                float* scale_shift, void* relu_mask, void* q8_out, float* q8_state, void* stream);
int mi355_bn_eval_fwd(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, long rows, int C, float eps,
                      int relu, int dtype, void* stream);
int mi355_bn_bwd_partials(const void* dy, const void* x, const void* y, const float* gamma, const float* beta,
                          const float* save_mean, const float* save_invstd, void* dx, void* dresidual, float* dgamma,
                          float* dbeta, int accumulate, long rows, int C, int relu, int dtype, const float* partial,
                          int nslices, float* coeff, const void* relu_mask, void* q8_out, float* q8_state, void* stream);
int mi355_bn_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* beta,
                 const float* save_mean, const float* save_invstd, void* dx, void* dresidual, float* dgamma, float* dbeta,
                 int accumulate, long rows, int C, int relu, int dtype, void* ws, size_t ws_bytes,
                 const void* relu_mask, void* q8_out, float* q8_state, void* stream);
/* mi355_bn_bwd takes ONE launch when x and dy fit the chip's LDS together (<= ~16.8 MB each on 256 CUs), C is a multiple of
 * 64 (bf16) / 32 (fp32) and the mask does not come from y: each CU keeps its tile of both tensors in LDS between the
 * reduction and the apply pass (3 tensor passes over HBM instead of 5); the blocks exchange their partial sums inside the
 * launch behind a bounded spin.  MI355_BN_RESIDENT=0 keeps the three-launch form.  The one-launch form is only taken when the
 * occupancy query admits one block per CU at the largest LDS size (else the three-launch form runs).  A block whose spin
 * gives up (~0.3 s: another kernel kept a block of the grid off the chip) FAILS LOUDLY: it writes NaN into its rows of dx
 * (and dgamma / dbeta if it owns them) and counts the event.  *out = give-ups since the library was loaded or
 * mi355_bn_resident_reset.  Synchronises.  Callers poll it where they synchronise anyway (train1.py: once per epoch;
 * mi355.ops.bn_resident_check raises). */
int mi355_bn_resident_timeouts(unsigned* out);
/* Clears the give-up count and the arrival counters after a give-up has been handled.  Synchronises the device. */
int mi355_bn_resident_reset(void);
/* TEST HOOK: grid-barrier poll iterations before a block gives up (0 restores the default, 1 << 19 ~ 0.3 s).  Used by the
 * tests to provoke a give-up and check that it is loud. */
int mi355_bn_resident_set_spin_limit(unsigned limit);
/* Run-time switch of the one-launch backward (1 on, 0 off, -1 back to the environment's choice); returns the previous value.
 * Switch it off while another kernel runs beside the backward on the same device (e.g. a collective overlapped with it): all
 * blocks of the one-launch form must be resident at once.  For the same reason two PROCESSES that share a GPU must not both use it
 * (train1.py / bench.py set MI355_BN_RESIDENT=0 when ranks share a device); a launch whose blocks could not all become resident
 * gives up after about 0.3 s, poisons its outputs with NaN and is counted by mi355_bn_resident_timeouts. */
int mi355_bn_set_resident(int on);
/* g <- (bit of relu_mask set) ? g : 0 in place; g [rows][C] bf16 / fp32, relu_mask as above.  The stand-alone form of what
 * mi355_conv_dgrad_masked_acc and mi355_bn_bwd (relu_mask) do on the fly: a residual block's last BatchNorm hands the
 * UNMASKED dy on to the other branch (no dresidual write) together with its bit mask (mi355/nn.py _LAZY_MASK). */
int mi355_apply_relu_mask(void* g, const void* relu_mask, long rows, int C, int dtype, void* stream);

/* Stem: BatchNorm (statistics partials from the conv epilogue, as mi355_bn_train_fwd_partials) + ReLU + MaxPool2d(3, 2, 1) in
 * one pass over the conv output: resnet.py:27-28 (`bn1`, `relu`, `maxpool` of the torchvision stem).  y_pool [N][Ho][Wo][C],
 * argidx as mi355_maxpool_fwd writes it (its backward: mi355_maxpool_bwd, then mi355_bn_bwd with relu = 1 and y = NULL). */
int mi355_bn_relu_maxpool_fwd_partials(const void* x, void* y_pool, uint8_t* argidx, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, int64_t* nbt, float* save_mean,
                                       float* save_invstd, int N, int H, int W, int C, float eps, float momentum,
                                       int stat_updates, int dtype, const float* partial, int nslices, float* scale_shift,
                                       void* stream);

/* ---------------------------------------------------------------- stem max-pool 3x3 s2 p1
 * Replaces nn.MaxPool2d(3,2,1) of the torchvision stem (uda/model/resnet.py:28).  argidx: uint8 window
 * position (first maximum in (kh,kw) scan order, as ATen) kept for the backward gather. */
int mi355_maxpool_fwd(const void* x, void* y, uint8_t* argidx, int N, int H, int W, int C, int dtype,
                      void* stream);
int mi355_maxpool_bwd(const void* dy, const uint8_t* argidx, void* dx, int N, int H, int W, int C, int dtype,
                      void* stream);

/* ---------------------------------------------------------------- layout changes at the API edge */
/* image / feature NCHW fp32 -> NHWC `dtype` with channels zero-padded to Cpad (stem input). */
int mi355_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype, void* stream);
/* The stem (resnet.py:23-28: Conv2d(3, 64, 7, stride 2, padding 3)) as a 4x4 / unit-stride conv over the image folded 2x2 into
 * channels: y [N][H/2][W/2][16] `dtype`, folded channel (dy*2 + dx)*4 + c (c = 3 zero), H and W even.  The conv then runs through
 * mi355_conv_fwd* / mi355_conv_wgrad with the descriptor {Hi = H/2, Wi = W/2, Ci = 16, kh = kw = 4, stride 1, pad 2, Ho = H/2,
 * Wo = W/2}: those entry points accept the top-left Ho x Wo crop of a unit-stride conv's output.  K = 256 instead of 49 x 8 = 392. */
int mi355_nchw_to_s2d(const float* x, void* y, int N, int H, int W, int dtype, void* stream);
/* stem weights fp32 [Co][7][7][3] -> folded forward operand `dtype` [Co][4][4][16]; and the folded weight gradient fp32
 * [Co][4][4][16] back onto g fp32 [Co][7][7][3] (g = / += per `accumulate`). */
int mi355_stem_s2d_pack(const float* w, void* out, int Co, int dtype, void* stream);
int mi355_stem_s2d_unpack_grad(const float* gs, float* g, int Co, int accumulate, void* stream);
/* NHWC `dtype` -> NCHW fp32 (the feature map `f` returned by PoseResNetx9.forward, regda_7.py:4944). */
int mi355_nhwc_to_nchw(const void* x, float* y, int N, int C, int H, int W, int dtype, void* stream);

/* ---------------------------------------------------------------- 21-channel pointwise convs
 * The 1x1 convs touching the K=21 heat-map tensors (regda_7.py:4916-4922 heads' last conv,
 * :4513 / :4588 heatmap_conv).  Heat-maps y are NCHW fp32 [N][K][HW]; features x are NHWC [N*HW][C].
 *  c2k : y[n][k][p] (=) bias[k] + sum_c x[n*HW+p][c] * w[k][c]          (w fp32 [K][C])
 *  k2c : out[n*HW+p][c] (=|+=residual) bias[c] + sum_k y[n][k][p]*w[c][k], times *scale_dev (w fp32 [C][K])
 *  wgrad: dw[K][C] (kc_layout=1) or dw[C][K] (kc_layout=0) (=|+=) sum_{n,p} y[n][k][p] * x[n*HW+p][c]
 *  rowsum: out[k] (=|+=) sum_{n,p} y[n][k][p]                             (bias gradient)
 *  w_transposed!=0: w is stored the other way round ([C][K] for c2k, [K][C] for k2c), which is how the
 *  input-gradient of each op reuses the other op with the same weight tensor.
 */
/* MFMA form of c2k (forward of the heads' last conv): w is the packed `dtype` copy [K][C]; K <= 32, C/chunk a power of 2. */
int mi355_conv1x1_heatmap(const void* x, const void* w, const float* bias, float* y, int N, int HW, int C, int K,
                          int dtype, void* stream);
int mi355_pw_c2k(const void* x, const float* w, const float* bias, float* y, int N, int HW, int C, int K,
                 int w_transposed, int dtype, void* stream);
int mi355_pw_k2c(const float* y, const float* w, const float* bias, const void* residual,
                 const float* scale_dev, void* out, int N, int HW, int C, int K, int w_transposed, int dtype,
                 void* stream);
/* mi355_pw_k2c with the BatchNorm statistics of its output from the epilogue (the BN in front of `last_lay`, regda_7.py:4551-4561):
 * partial[N * ceil(HW/64)][C][n, mean, M2], consumed by mi355_bn_train_fwd_partials */
int mi355_pw_k2c_stats(const float* y, const float* w, const float* bias, const void* residual, const float* scale_dev,
                       void* out, int N, int HW, int C, int K, int w_transposed, int dtype, float* partial,
                       size_t partial_bytes, int* nslices, void* stream);
size_t mi355_pw_wgrad_workspace(int N, int HW, int C, int K);
int mi355_pw_wgrad(const void* x, const float* y, float* dw, int kc_layout, int accumulate, int N, int HW,
                   int C, int K, int dtype, void* ws, size_t ws_bytes, void* stream);
int mi355_hm_rowsum(const float* y, float* out, int accumulate, int N, int K, int HW, void* ws, size_t ws_bytes,
                    void* stream); /* ws: >= N*K floats */

/* ---------------------------------------------------------------- heat-map decode / losses (rows = B*K maps)
 * argmax2d: utils/keypoint_detection.py:7-35 get_max_preds — first maximum (np.argmax tie rule, NaN
 *   counts as maximum), x = idx % W, y = floor(idx / W), both zeroed when max <= 0.  Bit-exact.
 * softargmax: utils/keypoint_detection.py:209-239 — softmax(beta*hm), (E[col], E[row]) * out_scale.
 */
int mi355_argmax2d(const float* hm, int32_t* idx, float* xy, float* maxval, int rows, int H, int W,
                   void* stream);
int mi355_softargmax(const float* hm, float* uv, int rows, int H, int W, float beta, float out_scale,
                     void* stream);
/* KL loss, uda/model/loss.py:145-158: per row r
 *   logp = log_softmax(pred[r]); t = (target[r]+eps)/sum(target[r]+eps);
 *   loss_rows[r] = weight[r] * sum_j xlogy-style t_j*(log t_j - logp_j)   (0 where t_j==0)
 *   unit_grad[r][j] (nullable) = weight[r]*inv_count * (softmax_j*sum(t) - t_j)   = d(mean loss)/d pred
 */
int mi355_kl_heatmap(const float* pred, const float* target, const float* weight, float eps,
                     float* loss_rows, float* unit_grad, int rows, int HW, float inv_count, void* stream);
/* out[0] = scale * sum(in[0..n)) in a fixed order (deterministic). */
int mi355_reduce_sum(const float* in, float* out, int n, float scale, void* stream);
/* out[i] = in[i] * (*g_dev) : backward of the KL loss (upstream scalar gradient on device). */
int mi355_scale_by_dev(const float* in, const float* g_dev, float* out, long n, void* stream);
/* out[i] = in[i] * (*g_dev) on a feature tensor (dtype MI355_F32 / MI355_BF16, n a multiple of one 16-byte chunk):
 * backward of utils/gl.py:8-18 (GradientFunction: grad * coeff) for consumers that cannot fold lambda into their own
 * input-gradient epilogue. */
int mi355_scale_feature(const void* in, const float* g_dev, void* out, long n, int dtype, void* stream);
/* Pseudo labels from arg-max coordinates xy[B*K][2] (of the 64x64-level main prediction):
 *   centre = trunc(xy / div); gt = clipped Gaussian patch (patch[(2r+1)^2], host table) at centre on an
 *   S x S map.  regda_4.py:76-86 (div 1, r 6), regda_7.py:3026-3039 (div 4, r 3), :3188-3201 (div 2, r 4).
 * kind: 0 base  : gf = clip(sum_{j!=k} gt_j, 0, 1)                                   (regda_4.py:83-84)
 *       1 x1/x5 : gf = clip(1 - 10 gt, 0, 1)                                         (regda_7.py:3255-3256)
 *       2 x6    : gf = clip(clip(sum_k gt_k,0,1) - 10 gt, 0, 1)                      (regda_7.py:3614-3616)
 * extra (nullable, [B*K][S*S]): gf = clip(gf + extra - 100 gt, 0, 1)                 (:3542-3544, :3618-3620)
 * normalise==1 : gf /= max(gf) per map (0/0 -> NaN as in the reference)              (:3546-3548, :3623-3625)
 * normalise==2 : the same, but a map whose maximum is 0 is left at zero (synthetic-noise benchmarks: collapsed target
 *                predictions make such maps after a few dozen iterations; not the reference's behaviour)
 * gt / gf outputs nullable.
 */
int mi355_pseudo_label(const float* xy, const float* patch, int radius, int div, int S, int kind,
                       const float* extra, int normalise, float* gt, float* gf, int B, int K, void* stream);
/* nn.Upsample(size, mode='bilinear') (align_corners=False) on [rows][h][w] -> [rows][H][W];
 * out = alpha*up(in) + (accumulate ? out : 0)   (train1.py:410-424: target5 = 0.5*up(adv3) + up(adv2)). */
int mi355_bilinear_up(const float* in, float* out, int rows, int h, int w, int H, int W, float alpha,
                      int accumulate, void* stream);
/* PCK pieces of utils/keypoint_detection.py:38-92 on device: dist[B*K] = |pred-tgt| / (side/10) or -1. */
int mi355_pck_dists(const float* pred_xy, const float* tgt_xy, float* dists, int rows, float norm_x,
                    float norm_y, void* stream);

/* ---------------------------------------------------------------- optimiser
 * torch.optim.SGD(momentum, weight_decay, nesterov=True) of train1.py:141-148 over a flat fp32 range:
 *   g' = g + wd*p ; buf = momentum*buf + g' ; p -= lr * (nesterov ? g' + momentum*buf : buf)
 * lr read from *lr_dev.  p_lowp (nullable): bf16 copy of the updated parameters (same flat layout).
 */
int mi355_sgd_nesterov(float* p, const float* g, float* buf, long n, const float* lr_dev, float momentum,
                       float wd, int nesterov, void* p_lowp, void* stream);
int mi355_cast_f32(const float* in, void* out, long n, int dtype, void* stream);

/* ---------------------------------------------------------------- in-library kernel timing (bench.py roofline)
 * on = 1: every launch of the MFMA conv family is bracketed by hipEvents on its stream; on = 2: the BatchNorm kernels and
 * the weight-gradient slab reductions as well (family 1 / 2; they never enter the totals below).  mi355_prof_read
 * synchronises the recorded events and returns the conv family's totals since the last reset. */
int mi355_prof_enable(int on);
/* The launches logged since the last reset, one by one in launch order: family (0 conv MFMA, 1 BatchNorm, 2 other), event-timed
 * duration in microseconds (contains the dispatch latency mi355_prof_event_overhead_us measures), algorithmic FLOPs and bytes,
 * and a label naming the layer ("fwd k3s1 256>256 @64x64 n64 +stats", "bn_bwd_res rows262144 C256 ...").  Each entry is ONE
 * kernel launch, so the list joins in order with a rocprofv3 kernel trace of the same iteration (profiles/insitu_table.py). */
int mi355_prof_launch_count(long* n);
int mi355_prof_read_launch(long i, int* family, double* us, double* flops, double* bytes, char* label, int label_cap);
/* idle spin of `us` microseconds on the stream (measurement aid: queue launches behind it so the GPU never waits for the host) */
int mi355_spin_us(long us, void* stream);
/* the timed launches split at an arithmetic intensity (FLOP / byte): out[0..3] = {ms, flops, bytes, launches} below it, out[4..7] above */
int mi355_prof_read_split(double flop_per_byte, double* out);
/* mean reading of an event pair around an empty kernel (the dispatch latency contained in every event-timed launch) */
int mi355_prof_event_overhead_us(int n, void* stream, double* us);
int mi355_prof_reset(void);
int mi355_prof_read(double* total_ms, long* launches, double* flops, double* algorithmic_bytes);

#ifdef __cplusplus
}
#endif
#endif
