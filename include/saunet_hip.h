/* libsaunet_hip.so -- C ABI of the MI355X-native SAUNet hot path (gfx950, wave64, MFMA).
 *
 * The reference (sunjesse/shape-attentive-unet) has no FFI: its seam is the Python
 * torch.nn.Module API.  Every entry point below replaces the ATen op (or
 * group of ops) that a reference module executes; the citation after each declaration is
 * the reference call site it stands in for (paths relative to /root/reference).
 *
 * Conventions
 *  - plain pointers and sizes only; the CALLER owns all memory (PyTorch allocates), including
 *    workspaces; nothing here allocates, synchronises or calls back to the host.
 *  - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream and
 *    re-entrant (forward runs on the Python thread, backward on autograd's thread).
 *  - activations are NHWC ("channels_last"); a tensor view is (ptr, pixels/N,H,W, C, ld) where
 *    ld >= C is the channel stride in ELEMENTS (so a channel slice of a wider buffer is a view:
 *    this is how DenseNet's concat is made free).
 *  - dtype: SAUNET_F32 (0) or SAUNET_BF16 (1) = storage type of activations / packed weights;
 *    accumulation, statistics, losses and parameters are always float32 (statistics sums float64).
 *  - return 0 on success, <0 = saunet_status; saunet_last_error() gives a thread-local message.
 *    Nothing throws across the ABI.
 */
#ifndef SAUNET_HIP_H
#define SAUNET_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum saunet_status { SAUNET_OK = 0, SAUNET_BAD_SHAPE = -1, SAUNET_BAD_DTYPE = -2, SAUNET_LAUNCH_FAILED = -3,
                     SAUNET_BAD_ALIGN = -4, SAUNET_UNSUPPORTED = -5 };
enum saunet_dtype { SAUNET_F32 = 0, SAUNET_BF16 = 1 };
/* weight packings produced by saunet_pack_weight (K = reduction index, contiguous) */
enum saunet_pack_mode {
    SAUNET_PACK_FWD = 0,       /* Conv2d [Co,Ci,kh,kw]        -> [Co][kh][kw][Ci]                         */
    SAUNET_PACK_DGRAD = 1,     /* Conv2d [Co,Ci,kh,kw]        -> [Ci][kh'][kw'][Co], taps flipped (stride 1) */
    SAUNET_PACK_CONVT_FWD = 2, /* ConvTranspose2d [Ci,Co,4,4] -> [ph][pw][Co][th][tw][Ci]  (k4 s2 p1 phases) */
    SAUNET_PACK_CONVT_DGRAD = 3/* ConvTranspose2d [Ci,Co,4,4] -> [Ci][kh][kw][Co]  (a stride-2 conv over dy) */
};

typedef struct saunet_conv_desc {
    int32_t dtype;
    int32_t N, H, W, Cin, ldx;     /* input view  [N,H,W,Cin]  stride ldx  */
    int32_t Ho, Wo, Cout, ldy;     /* output view [N,Ho,Wo,Cout] stride ldy */
    int32_t KH, KW, stride, pad;
    int32_t transposed;            /* 1: ConvTranspose2d(k=4,s=2,p=1); x is the LOW-res input, y the 2x output */
    int32_t pro_relu;              /* prologue: a = x*pro_scale[c]+pro_shift[c], then max(a,0) if set */
    /* Statistic accumulators may be REPLICATED: workgroup b adds into replica (b % stat_replicas) at
     * base + replica*stat_rstride (elements).  Thousands of workgroups hitting the same 2*C float64 addresses serialise
     * in the cross-XCD atomic path (+40 us on a 4096-block launch); 16 replicas remove that.  0/1 = single copy. */
    int32_t stat_replicas, stat_rstride;
    int32_t epi_relu;              /* epilogue: y = max(conv + bias, 0) -- inference with BatchNorm folded into weights and bias */
    /* Optional caller-owned scratch for the forward entry points (saunet_conv2d_forward / _ex / _bnpro): at least
     * saunet_conv2d_forward_workspace(d) bytes, 16-byte aligned, private to this call until it has completed on `stream`.  NULL / too small:
     * the launch uses a kernel selection that needs none (same result up to the float32 summation order). */
    void* workspace; int64_t workspace_bytes;
} saunet_conv_desc;

const char* saunet_last_error(void);
/* ABI version of the structs and entry points in this header.  History: 1 = rounds 1-4; 2 = round 5's layout (saunet_conv_desc grew
 * `workspace` / `workspace_bytes`, the saunet_dense_layer_* entries) -- that round still answered 1; 3 = round 6.  A caller built
 * against another header must refuse to run:  if (saunet_version() != SAUNET_ABI_VERSION) abort();  -- the library reads every descriptor
 * field of ITS header, a shorter struct from an older header would be read past its end. */
#define SAUNET_ABI_VERSION 4
int saunet_version(void);
/* names of the kernels the calling thread's API calls have launched since the previous call of this function, joined by '+' (a name is the
 * kernel's symbol without "_kernel", e.g. "conv_igemm_fwd", "bn_bwd_correct_ab+dense_dgrad3"); thread-local, valid until the next call.
 * Measurement aid (bench.py attributes HIP-event times to kernel families with it); no effect on any launch. */
const char* saunet_launch_log(void);
/* number of compute units seen on `device` (sanity / grid sizing); <0 on error */
int saunet_init(int device);

/* ---- convolution family ------------------------------------------------------------------
 * replaces nn.Conv2d / nn.ConvTranspose2d forward, e.g. models/models.py:118-123, :203-237,
 * attention_blocks.py:150-151,179-186,215-220, GSConv.py:40-42,56-57, resnet.py:24-27 and the
 * DenseNet-121 convs of torchvision (models/models.py:271). */
int saunet_pack_weight(int mode, int dtype, const float* w, int Co, int Ci, int KH, int KW, void* out, void* stream);
/* the same for up to 64 weights in ONE launch (all packings of a training step are refreshed right after the
 * optimiser update instead of 300+ tiny launches) */
typedef struct saunet_pack_list { int32_t count; int32_t mode[64]; int32_t dims[64][4]; const void* src[64]; void* dst[64]; } saunet_pack_list;
int saunet_pack_weight_multi(const saunet_pack_list* pl, int dtype, void* stream);
/* y = conv(prologue(x), w) (+bias).  If stat_sum/stat_sumsq are non-NULL, per-output-channel
 * sums of the un-biased accumulator and its square are ATOMICALLY added (float64) -- the batch
 * statistics BatchNorm needs, taken in the producer's epilogue. */
/* bytes of workspace the forward of `d` can use (0 = none).  Today only the 8 x 8 `center` geometry asks for one: its 3x3 convolution and
 * data gradient (models/models.py:316) split K over workgroups and keep the float32 partials there (csrc/conv_mm.hip, cell mode). */
int64_t saunet_conv2d_forward_workspace(const saunet_conv_desc* d);
int saunet_conv2d_forward(const saunet_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                          const float* pro_scale, const float* pro_shift, void* y,
                          double* stat_sum, double* stat_sumsq, void* stream);
/* Same, with an optional fused BatchNorm-backward epilogue for the case "this convolution IS the dgrad that produces
 * the gradient of a BN(+ReLU) output": instead of y = conv(...), store g = y * [bn_x*scale+shift > 0] and add
 * sum g, sum g*xhat (xhat = (bn_x-mean)*invstd) to sums[0:C], sums[C:2C] -- the reduction pass of BN backward, done
 * while the tile is still on chip.  bn_x is the tensor that BN normalised, same pixels/channels as y. */
typedef struct saunet_bn_epilogue {
    const void* bn_x; int32_t ld_bn_x; int32_t relu;
    int32_t accumulate;   /* 1: y += scale[c]*g instead of y = g ("linear" BN backward, see saunet_bn_backward_coeff); 2: y = scale[c]*g (the same
                           * term written by the FIRST consumer of a buffer, without reading y); 1x1 path only */
    int32_t reserved;
    const float* scale; const float* shift; const float* mean; const float* invstd;
    double* sums;
    int32_t sums_replicas, sums_rstride;   /* replicated like the statistics (0/1 = single copy) */
    const uint8_t* relu_mask;              /* ABI 4: optional ReLU decisions as bits (saunet_affine_act_mask layout, pixels * Cout / 8 bytes) used INSTEAD of
                                            * bn_x*scale+shift > 0 -- the output of a residual block, whose pre-activation also has the skip tensor in it
                                            * (/root/reference/models/resnet.py:54-59).  1x1 data gradients on the implicit-GEMM path, bf16 only. */
} saunet_bn_epilogue;
int saunet_conv2d_forward_ex(const saunet_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                             const float* pro_scale, const float* pro_shift, void* y,
                             double* stat_sum, double* stat_sumsq, const saunet_bn_epilogue* epi, void* stream);
/* saunet_bn_epilogue with bn_x == NULL and accumulate == 1 asks saunet_conv2d_forward_ex for  y += conv(x, w)  with no BatchNorm epilogue:
 * the data gradient of a residual block's first convolution lands on top of the skip branch's gradient instead of being added by a separate
 * pass (BasicBlock.forward `out += residual`, /root/reference/models/resnet.py:30-59 as used at models/models.py:316,322).  Only some
 * geometries have a kernel for it; this returns 1 when `d` does (saunet_conv2d_forward_ex fails with SAUNET_UNSUPPORTED otherwise). */
int saunet_conv2d_accumulate_supported(const saunet_conv_desc* d);
/* Consumer-side BatchNorm finalize: y = conv(relu?(BN(x)), w) where the BatchNorm coefficients are derived INSIDE the convolution kernel from
 * the raw batch statistics its producer accumulated -- no saunet_bn_finalize launch in between (a DenseNet layer is then two launches
 * instead of four; torchvision _DenseLayer norm1/norm2 as used at /root/reference/models/models.py:306-313).
 *   channels [c_lo, Cin): mean / variance from (sum, sumsq, count) exactly as saunet_bn_finalize computes them (float64), and -- by one
 *       workgroup -- written to xhat (rows xs = invstd, xt = -mean*invstd, mean, invstd, biased variance) for later consumers of the same
 *       channels (a dense block's concat channels are normalised by up to 24 later norm1 layers: same statistics, different gamma / beta);
 *   channels [0, c_lo): read from xhat.
 * One workgroup also writes what saunet_bn_finalize would have returned -- params [4][Cin] = scale, shift, mean, invstd (the backward pass
 * needs them) -- and updates running_mean / running_var (momentum, unbiased variance) for all Cin channels.  Training mode only. */
typedef struct saunet_bn_prologue {
    const double* sum; const double* sumsq;    /* accumulators indexed by input channel, replicated like saunet_conv_desc.stat_* */
    int32_t replicas, rstride;
    double count;
    float eps, momentum;
    int32_t c_lo, ld_xhat;
    float* xhat;                                /* [5][ld_xhat]; may be NULL when c_lo == 0 and nobody else needs the rows */
    const float* gamma; const float* beta;      /* [Cin] */
    float* params;                              /* [4][Cin] out */
    float* running_mean; float* running_var;    /* [Cin] in/out, may be NULL */
} saunet_bn_prologue;
int saunet_conv2d_forward_bnpro(const saunet_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                const saunet_bn_prologue* pro, void* y, double* stat_sum, double* stat_sumsq, void* stream);
/* the xhat rows of C channels from their statistics (the channels a dense block starts with): xhat[5][ld] as above */
int saunet_bn_xhat(int C, const double* sum, const double* sumsq, int replicas, int rstride, double count, float eps, float* xhat, int ld, void* stream);
/* dw[...] += sum_pixels dy (x) prologue(x).  dw is the float32 gradient in the PARAMETER's own
 * layout ([Co,Ci,kh,kw], or [Ci,Co,4,4] when d->transposed); it must be zero-initialised by the
 * caller (split-K partial sums are added atomically).  replaces autograd's convolution_backward
 * (weight part) behind loss.backward() at train.py:104. */
int saunet_conv2d_wgrad(const saunet_conv_desc* d, const void* x, const void* dy,
                        const float* pro_scale, const float* pro_shift, float* dw,
                        void* workspace, int64_t workspace_bytes, void* stream);
/* The same with the cross-workgroup reduction of the tiled kernels DEFERRED: when the shape runs on them only the per-group partial
 * gradients are written to `workspace` and *pending describes the reduction still to do (pending->groups > 0; the workspace must stay
 * untouched until saunet_wgrad_reduce_multi has run); otherwise the gradient is complete on return and pending->groups == 0.
 * saunet_wgrad_reduce_multi performs up to SAUNET_WGRAD_REDUCE_MAX pending reductions in ONE launch (dw[i] += sum over groups, no atomics)
 * -- a DenseNet block's backward issues two weight gradients per layer, i.e. 116 tiny reduce launches per step otherwise. */
#define SAUNET_WGRAD_REDUCE_MAX 64
typedef struct saunet_wgrad_pending { const float* ws; float* dw; int64_t wsize; int32_t groups, taps; } saunet_wgrad_pending;   /* taps = 9 / 16: partials are [tap][wsize / taps], dw is [wsize / taps][tap]; 0: same layout (filled in by the library) */
typedef struct saunet_wgrad_reduce_list { int32_t count, reserved; saunet_wgrad_pending item[SAUNET_WGRAD_REDUCE_MAX]; } saunet_wgrad_reduce_list;
int saunet_conv2d_wgrad_deferred(const saunet_conv_desc* d, const void* x, const void* dy,
                                 const float* pro_scale, const float* pro_shift, float* dw,
                                 void* workspace, int64_t workspace_bytes, saunet_wgrad_pending* pending, void* stream);
int saunet_wgrad_reduce_multi(const saunet_wgrad_reduce_list* l, void* stream);
/* Weight AND bias gradient of a few-output pointwise convolution in one pass (Cout <= 4: the C -> 1 side outputs c3 / c4 / c5 / phi / cw, `fuse`,
 * `final`, /root/reference/models/models.py:286-301,324 and attention_blocks.py:150-151): dbias[co] += sum_p dy[p][co] rides as a "ones" input
 * channel of the row-sum kernel, so the separate pass over dy (saunet_channel_sum) and its float64 -> float32 cast disappear.  dw and dbias
 * are ACCUMULATED into (hand in zeroed buffers); workspace as for saunet_conv2d_wgrad (saunet_conv2d_wgrad_workspace(d) bytes: per-split
 * partials + an ordered reduce, no float atomics; without it: one float atomic per value and workgroup).
 * saunet_conv2d_wgrad_bias_supported(d) == 1 tells whether the geometry is served. */
int saunet_conv2d_wgrad_bias_supported(const saunet_conv_desc* d);
int saunet_conv2d_wgrad_bias(const saunet_conv_desc* d, const void* x, const void* dy, const float* pro_scale, const float* pro_shift,
                             float* dw, float* dbias, void* workspace, int64_t workspace_bytes, void* stream);
/* Weight gradients of up to SAUNET_WGRAD_GROUP_MAX stride-1 convolutions of ONE geometry (same N x H x W map, same kernel size 1x1 pad 0 or
 * 3x3 pad 1, same dtype; per-problem channel counts, operands, prologue vectors and gradient buffers) in ONE launch (+ one reduction launch
 * when the pixel tiles are split over several workgroups).  A DenseNet block's backward (torchvision _DenseBlock as used at
 * /root/reference/models/models.py:306-313) defers the two weight gradients of each of its 6-24 layers to the end of the block: on the
 * low-resolution blocks one layer's problem has 32-128 pixel tiles, and filling 256 CUs took 32-128 pixel groups each writing a full partial
 * gradient; all layers together fill the chip with 2-8 groups.  dw[i] is complete on return; it is STORED when one workgroup owns the
 * whole problem and ACCUMULATED (dw += sum of partials) otherwise, so hand in zeroed buffers.
 * saunet_conv2d_wgrad_grouped_workspace: bytes of scratch the call needs (>= 0), or a negative saunet_status when the geometry is not
 * served by the tiled kernels (the caller then issues saunet_conv2d_wgrad per problem). */
#define SAUNET_WGRAD_GROUP_MAX 32
typedef struct saunet_wgrad_group_item {
    const void* x; const void* dy; float* dw; const float* pro_scale; const float* pro_shift;
    int32_t Cin, ldx, Cout, lddy;
} saunet_wgrad_group_item;
typedef struct saunet_wgrad_group {
    int32_t dtype, N, H, W, KH, pad, pro_relu, count;
    saunet_wgrad_group_item item[SAUNET_WGRAD_GROUP_MAX];
} saunet_wgrad_group;
int64_t saunet_conv2d_wgrad_grouped_workspace(const saunet_wgrad_group* g);
int saunet_conv2d_wgrad_grouped(const saunet_wgrad_group* g, void* workspace, int64_t workspace_bytes, void* stream);
/* ---- DenseNet layer backward, fused (round 5; bf16 storage, training-mode BatchNorm) ---------------------------------------------------
 * One torchvision _DenseLayer (norm1-relu-conv1(1x1 -> 128)-norm2-relu-conv2(3x3 -> 32), used at /root/reference/models/models.py:306-313)
 * inside a block whose concat is ONE buffer; replaces autograd's convolution_backward (input part) + native_batch_norm_backward of both
 * BatchNorms behind loss.backward() (train.py:104) with TWO launches per layer (four until round 4):
 *   saunet_dense_layer_backward_conv2:  g = dbuf[:, Cin:Cin+32] - (A + B * xhat)   (the deferred correction of the block's "linear" BN1 backward,
 *       A, B = ab / count, applied as the operand is loaded and written to dz2 for the weight gradient)
 *       G = conv2-dgrad(g) * [relu mask of norm2(z1)]  -> g_out,   sums2 += (sum G, sum G * xhat(z1))
 *     (maps with >= 256 16 x 16 tiles use the LDS-DMA staged kernel and a separate correction pass: two launches inside the call)
 *   saunet_dense_layer_backward_conv1:  dz1 = scale2 * (G - mean(G) - xhat(z1) * mean(G * xhat))  from sums2, applied as G is loaded, written to dz1;
 *       D = conv1-dgrad(dz1) * [relu mask of norm1(buf)],  dbuf[:, :Cin] += scale1 * D,  sums1 += (sum D, sum D * xhat),
 *       ab[.., :Cin] += scale1 * (sum D, sum D * xhat)   (what saunet_bn_backward_coeff did in its own launch),  dgamma2 / dbeta2 = sums2.
 * dz2 == NULL: the chunk needs no correction (the block's last layer: nothing was accumulated into ab for it); conv2 then reads dbuf directly.
 * All accumulators are float64, replicated [R][2][C] like the statistics, zero-initialised by the caller. */
typedef struct saunet_dense_layer_bwd {
    int32_t N, H, W, Cin, Ctot;
    int32_t c_begin;                            /* _conv1 only: 0, or the first channel this call touches (layer pairs below) */
    const void* buf; void* dbuf;                /* [P][Ctot] concat activations / gradient (channel stride Ctot) */
    const float* xhat; int32_t ld_xhat, reserved2;   /* [5][ld_xhat]: xs, xt, mean, invstd, var of the concat channels (saunet_bn_xhat / _bnpro) */
    double* ab; int32_t ab_replicas, ab_rstride;     /* [R][2][Ctot] */
    double count;
    const void* z1; void* g; void* dz1; void* dz2;   /* [P][128] x3, [P][32]; g is scratch between the two calls */
    const void* w2_dgrad; const void* w1_dgrad;     /* SAUNET_PACK_DGRAD packings of conv2 / conv1 */
    const float* p1; const float* p2;               /* [4][Cin], [4][128]: scale, shift, mean, invstd (saunet_conv2d_forward_bnpro params) */
    double* sums2; int32_t sums2_replicas, sums2_rstride;   /* [R][2][128] */
    double* sums1; int32_t sums1_replicas, sums1_rstride;   /* [R][2][Cin] */
    float* dgamma2; float* dbeta2;                  /* [128] out */
} saunet_dense_layer_bwd;
int saunet_dense_layer_backward_conv2(const saunet_dense_layer_bwd* l, void* stream);
int saunet_dense_layer_backward_conv1(const saunet_dense_layer_bwd* l, void* stream);
/* Layer PAIRS (round 5): the conv1 data gradient is bound by the read of buf[:, :Cin] and the read-modify-write of dbuf[:, :Cin]; two
 * consecutive layers touch the same rows.  Only the top 32-channel chunk of layer l is needed before layer l - 1 can start (its conv2 data
 * gradient reads that chunk of dbuf), so a pair runs as
 *     _conv2(hi);  _conv1(hi with c_begin = hi.Cin - 32);  _conv2(lo);  _conv1_pair(hi, lo)
 * where _conv1_pair adds BOTH layers' contributions to dbuf[:, :lo.Cin] in one pass (lo.Cin == hi.Cin - 32; hi.dz1 is read, lo.dz1 written,
 * sums1 / ab of both layers updated for those channels).  _pair_supported: 1 when the geometry runs the LDS-staged kernels that implement
 * the window and the pair (low-resolution maps), else 0 -- the caller then issues the two-launch sequence per layer. */
int saunet_dense_layer_backward_pair_supported(const saunet_dense_layer_bwd* hi);
int saunet_dense_layer_backward_conv1_pair(const saunet_dense_layer_bwd* hi, const saunet_dense_layer_bwd* lo, void* stream);
/* ab[0][c] += scale[c] * S1[c],  ab[0][ab_half + c] += scale[c] * S2[c]  (S = the replicated BatchNorm-backward sums of a consumer that stored
 * y = scale * g: the transition behind a dense block), dgamma = S2, dbeta = S1: the coefficient half of the linear BN backward in the running
 * float64 form the fused dense-layer backward reads. */
int saunet_bn_backward_coeff_ab(int C, const double* sums, int sums_replicas, int sums_rstride, const float* scale, double* ab, int ab_half,
                                float* dgamma, float* dbeta, void* stream);
/* y = d - (A + B * (x*xs + xt)) with A, B = ab / count (replicated float64 sums, rows ab and ab + ab_half): the deferred correction of the linear
 * BN1 backward for C channels (C <= 256); y may alias d. */
int saunet_bn_backward_correct_ab(int dtype, const void* d, int ldd, const void* x, int ldx, void* y, int ldy, const double* ab, int ab_replicas,
                                  int ab_rstride, int ab_half, double count, const float* xs, const float* xt, int64_t pixels, int C, void* stream);
/* dgamma / dbeta of every norm1 of a dense block from the per-layer sums (one launch per block instead of one per layer) */
#define SAUNET_DENSE_LAYERS_MAX 64
typedef struct saunet_dense_bn1_list {
    int32_t count, replicas;
    const double* sums[SAUNET_DENSE_LAYERS_MAX]; int32_t rstride[SAUNET_DENSE_LAYERS_MAX]; int32_t cin[SAUNET_DENSE_LAYERS_MAX];
    float* dgamma[SAUNET_DENSE_LAYERS_MAX]; float* dbeta[SAUNET_DENSE_LAYERS_MAX];
} saunet_dense_bn1_list;
int saunet_dense_bn1_grads(const saunet_dense_bn1_list* l, void* stream);
/* bytes of caller-owned scratch saunet_conv2d_wgrad needs for this shape (0 = none; <0 = saunet_status).
 * The tiled kernels write per-block partial gradients there with plain stores and reduce them afterwards
 * (cross-XCD float atomics on the same addresses are ~10x more expensive than the stores + one reduce pass). */
int64_t saunet_conv2d_wgrad_workspace(const saunet_conv_desc* d);
/* db[c] = sum_pixels dy[:,c]  (float64 atomics into a zeroed buffer, then cast by the caller) */
int saunet_channel_sum(int dtype, const void* dy, int64_t pixels, int C, int ld, double* out, void* stream);

/* ---- batch normalisation -------------------------------------------------------------------
 * replaces nn.BatchNorm2d / SynchronizedBatchNorm2d (single-device path = F.batch_norm,
 * lib/nn/modules/batchnorm.py:58-61) forward+backward; models/norm.py:16-22. */
int saunet_bn_stats(int dtype, const void* x, int64_t pixels, int C, int ld, double* sum, double* sumsq,
                    int replicas, int rstride, void* stream);
/* base[i] = sum_r base[r*rstride + i], i < n: collapse replicated accumulators into replica 0 */
int saunet_sum_replicas(double* base, int n, int replicas, int rstride, void* stream);
/* training=1: mean/var from (sum,sumsq,count) [+conv_bias], updates running stats with `momentum`
 * (unbiased var), writes scale=gamma*invstd, shift=beta-mean*scale, mean, invstd.
 * training=0: scale/shift from the running statistics. */
int saunet_bn_finalize(int C, const double* sum, const double* sumsq, int replicas, int rstride, double count, const float* conv_bias,
                       const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* scale, float* shift,
                       float* mean, float* invstd, int training, void* stream);
/* SynchronizedBatchNorm2d across data-parallel replicas (lib/nn/modules/batchnorm.py:118-139, _compute_mean_std):
 * (sum, sumsq, count) are the GLOBAL (all-reduced) statistics; inv_std = clamp(biased var, eps)^-1/2; the moving average is
 * the reference's accumulator pair  tmp = tmp*(1-momentum) + stat,  iter = iter*(1-momentum) + 1,  running = tmp / iter
 * (unbiased variance).  tmp_running_mean == NULL skips the running-statistic update. */
int saunet_syncbn_finalize(int C, const double* sum, const double* sumsq, int replicas, int rstride, double count,
                           const float* gamma, const float* beta, float eps, float momentum,
                           float* tmp_running_mean, float* tmp_running_var, float* running_iter,
                           float* running_mean, float* running_var, float* scale, float* shift,
                           float* mean, float* invstd, void* stream);
/* y = act(x*scale+shift (+residual)) */
int saunet_affine_act(int dtype, const void* x, int ldx, const float* scale, const float* shift,
                      const void* residual, int ldr, int relu, void* y, int ldy, int64_t pixels, int C, void* stream);
/* y = act(x*scale+shift) AND pooled[n][c] = mean over the image's H*W pixels of y (the SE squeeze, attention_blocks.py:32,50) in the
 * same pass; pooled [pixels/HW][C] float32 is zeroed here.  Vector path only (C, strides multiples of 8 bf16 / 4 f32 elements). */
int saunet_affine_act_pool(int dtype, const void* x, int ldx, const float* scale, const float* shift, int relu, void* y, int ldy,
                           int64_t pixels, int C, float* pooled, int HW, void* stream);
/* g = dy * [out>0] where out = x*scale+shift(+residual) (if relu);  sums[0:C] += sum g,
 * sums[C:2C] += sum g*xhat  with xhat = (x-mean)*invstd   (float64 atomics, zeroed by caller) */
int saunet_bn_backward_reduce(int dtype, const void* dy, int lddy, const void* x, int ldx, const void* residual, int ldr,
                              const float* scale, const float* shift, const float* mean, const float* invstd,
                              int relu, double* sums, int replicas, int rstride, int64_t pixels, int C, void* stream);
/* dx (+)= scale*(g - sum_g/count - xhat*sum_gxhat/count) (training) or scale*g (eval);
 * optionally dres = g.  dgamma/dbeta are written from `sums` (float32) when non-NULL. */
int saunet_bn_backward_apply(int dtype, const void* dy, int lddy, const void* x, int ldx, const void* residual, int ldr,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             int relu, const double* sums, int sums_replicas, int sums_rstride, double count, int training, int accumulate,
                             void* dx, int lddx, void* dres, int lddres, float* dgamma, float* dbeta,
                             int64_t pixels, int C, void* stream);
/* Consumer-side BatchNorm finalize for the conv -> BN -> act layers (models/models.py:118-123, attention_blocks.py:215-220, resnet.py:54-59):
 * y = act(BN(x) (+residual)) with scale / shift derived INSIDE the pass from the raw batch statistics of x's producer, exactly as
 * saunet_bn_finalize computes them (conv_bias: that entry's `conv_bias`, may be NULL; c_lo must be 0; xhat may be NULL).  One workgroup writes pro->params ([4][C]: scale, shift,
 * mean, invstd -- what the backward entries take) and updates the running statistics.  Training mode, vector path only (SAUNET_UNSUPPORTED
 * otherwise: run saunet_bn_finalize + saunet_affine_act).  relu_mask: optional, see saunet_affine_act_mask.  _pool_bn: the same for
 * saunet_affine_act_pool. */
int saunet_affine_act_bn(int dtype, const void* x, int ldx, const saunet_bn_prologue* pro, const float* conv_bias, const void* residual, int ldr, int relu,
                         void* y, int ldy, int64_t pixels, int C, uint8_t* relu_mask, void* stream);
int saunet_affine_act_pool_bn(int dtype, const void* x, int ldx, const saunet_bn_prologue* pro, const float* conv_bias, int relu, void* y, int ldy,
                              int64_t pixels, int C, float* pooled, int HW, void* stream);
/* DenseNet transition with the average pool IN FRONT of its 1x1 convolution (torchvision _Transition: norm -> relu -> conv1x1 -> AvgPool2d(2, 2),
 * /root/reference/models/models.py:271 as sliced at :306-313).  A pointwise convolution and an average pool commute, so
 *   y = conv1x1(saunet_bn_relu_avgpool2(x))
 * equals pool(conv1x1(relu(bn(x)))) up to rounding, with the convolution, its data gradient and its weight gradient on a quarter of the pixels.
 *   forward : y[n, oy, ox, c] = 1/4 * sum over the 2x2 window of relu(x*scale+shift)          x [N,H,W,C] -> y [N,H/2,W/2,C]
 *   backward: g = [x*scale+shift > 0] * 1/4 * da[n, h/2, w/2, c];  sums[0:C] += sum g, sums[C:2C] += sum g*xhat (float64, zeroed by the caller:
 *             what saunet_bn_backward_reduce would have produced);  dx = g (scaled = 0: continue with saunet_bn_backward_apply on pre-masked
 *             sums) or scale*g (scaled = 1: the dense block's linear form, saunet_bn_backward_coeff_ab).
 * Even H and W, C <= 2048, vector path only (SAUNET_UNSUPPORTED otherwise). */
int saunet_bn_relu_avgpool2(int dtype, const void* x, int ldx, const float* scale, const float* shift, void* y, int ldy,
                            int N, int H, int W, int C, void* stream);
int saunet_bn_relu_avgpool2_backward(int dtype, const void* da, int ldda, const void* x, int ldx, const float* scale, const float* shift,
                                     const float* mean, const float* invstd, int scaled, void* dx, int lddx,
                                     double* sums, int replicas, int rstride, int N, int H, int W, int C, void* stream);
/* Residual blocks (y = relu(x*scale+shift + residual), /root/reference/models/resnet.py:54-59): the backward pass needs the ReLU decision of
 * every element, and recomputing it means re-reading the skip tensor in both the reduce and the apply pass.  saunet_affine_act_mask also
 * writes the decisions as bits -- relu_mask[pixel * C/8 + c/8] bit (c % 8), dense, pixels * C / 8 bytes -- and the _masked backward entries
 * read those instead of the residual (one full-resolution tensor read less per pass).  bf16, C and strides multiples of 8, 16-byte aligned
 * views (SAUNET_UNSUPPORTED otherwise); same sums / outputs as the unmasked entries with relu = 1. */
int saunet_affine_act_mask(int dtype, const void* x, int ldx, const float* scale, const float* shift,
                           const void* residual, int ldr, void* y, int ldy, int64_t pixels, int C, uint8_t* relu_mask, void* stream);
int saunet_bn_backward_reduce_masked(int dtype, const void* dy, int lddy, const void* x, int ldx, const uint8_t* relu_mask,
                                     const float* scale, const float* shift, const float* mean, const float* invstd,
                                     double* sums, int replicas, int rstride, int64_t pixels, int C, void* stream);
int saunet_bn_backward_apply_masked(int dtype, const void* dy, int lddy, const void* x, int ldx, const uint8_t* relu_mask,
                                    const float* scale, const float* shift, const float* mean, const float* invstd,
                                    const double* sums, int sums_replicas, int sums_rstride, double count, int training, int accumulate,
                                    void* dx, int lddx, void* dres, int lddres, float* dgamma, float* dbeta,
                                    int64_t pixels, int C, void* stream);

/* "Linear" form of the BatchNorm backward used inside DenseNet, where one concat channel feeds many BatchNorms:
 *   dx_c = sum_k s_kc*g_k  -  (A_c + B_c*xhat_c),   A_c = sum_k s_kc*mean(g_k),  B_c = sum_k s_kc*mean(g_k*xhat)
 * The first term is accumulated by the dgrad epilogue (accumulate=1); `coeff` folds one consumer's reduction into A/B
 * (and emits that BatchNorm's dgamma/dbeta); `correct` applies -(A + B*xhat) once per channel chunk, in place. */
int saunet_bn_backward_coeff(int C, const double* sums, int sums_replicas, int sums_rstride, double count, const float* scale, float* A, float* B,
                             float* dgamma, float* dbeta, int training, void* stream);
int saunet_bn_backward_correct(int dtype, void* dx, int lddx, const void* x, int ldx, const float* A, const float* B,
                               const float* xhat_scale, const float* xhat_shift, int64_t pixels, int C, void* stream);
/* `coeff` of one consumer (all C channels it normalises; training mode) and `correct` of the channel chunk [c_lo, c_hi) (<= 256 channels of
 * those C) in ONE launch: inside a dense block's backward the two always follow each other.  A_in/B_in -> A_out/B_out are distinct
 * (ping-pong) buffers; dx / x point at the chunk's first channel; xhat_scale / xhat_shift are indexed by absolute channel. */
int saunet_bn_backward_coeff_correct(int dtype, int C, const double* sums, int sums_replicas, int sums_rstride, double count, const float* scale,
                                     const float* A_in, const float* B_in, float* A_out, float* B_out, float* dgamma, float* dbeta,
                                     void* dx, int lddx, const void* x, int ldx, int c_lo, int c_hi,
                                     const float* xhat_scale, const float* xhat_shift, int64_t pixels, void* stream);

/* ---- resampling / pooling ------------------------------------------------------------------
 * F.interpolate(mode='bilinear', align_corners=True) models/models.py:337-356,372-374,386-389;
 * nn.MaxPool2d(2,2) :270,376; AvgPool2d(2,2) in the DenseNet transitions. */
int saunet_bilinear_forward(int dtype, const void* x, int N, int H, int W, int C, int ldx, void* y, int Ho, int Wo, int ldy, void* stream);
int saunet_bilinear_backward(int dtype, const void* dy, int N, int Ho, int Wo, int C, int lddy, void* dx, int H, int W, int lddx, int accumulate, void* stream);
int saunet_pool2x2_forward(int dtype, int is_max, const void* x, int N, int H, int W, int C, int ldx, void* y, int ldy, void* stream);
int saunet_pool2x2_backward(int dtype, int is_max, const void* x, const void* dy, int N, int H, int W, int C, int ldx, int lddy, void* dx, int lddx, int accumulate, void* stream);

/* out[(n,oh,ow)][(kh,kw,c)] = x[n,oh*s-p+kh,ow*s-p+kw,c] (zero outside); C a multiple of the 16-byte chunk.  Lowers the
 * DenseNet stem conv0 (7x7 stride 2 on the 8-channel padded image, models/models.py:304) to a K=392 pointwise GEMM. */
int saunet_im2col(int dtype, const void* x, int N, int H, int W, int C, int ldx, int KH, int KW, int stride, int pad,
                  void* out, int ldo, void* stream);

/* ---- element-wise glue ----------------------------------------------------------------------*/
/* dst[:, :C] (ld ldd) = src[:, :C] (ld lds) -- the only "cat" there is: writing a channel slice */
int saunet_copy_channels(int dtype_src, int dtype_dst, const void* src, int lds, void* dst, int ldd, int64_t pixels, int C, int accumulate, void* stream);
/* y = sigmoid(x) ; dx = dy*y*(1-y) */
int saunet_sigmoid_forward(int dtype, const void* x, int ldx, void* y, int ldy, int64_t pixels, int C, void* stream);
int saunet_sigmoid_backward(int dtype, const void* y, int ldyy, const void* dy, int lddy, void* dx, int lddx, int64_t pixels, int C, int accumulate, void* stream);
/* gate multiply  y = x * (alpha + 1), alpha is a 1-channel map (GSConv.py:55) */
int saunet_gate_mul_forward(int dtype, const void* x, int ldx, const void* alpha, void* y, int ldy, int64_t pixels, int C, void* stream);
int saunet_gate_mul_backward(int dtype, const void* x, int ldx, const void* alpha, const void* dy, int lddy,
                             void* dx, int lddx, void* dalpha, int64_t pixels, int C, void* stream);

/* ---- selectable global pooling (models/adaptive_avgmax_pool.py:19-40 adaptive_avgmax_pool2d, :43-74 AdaptiveAvgMaxPool2d; and
 * nn.AdaptiveAvgPool2d(1) of SEModule, models/attention_blocks.py:32) --------------------------------------------------------------
 * mode: 0 'avg', 1 'max', 2 'avgmax' = 0.5*(avg+max), 3 'avgmaxc' = [avg | max] (2C values per image).  One pass over x [N, HW, C]
 * (NHWC, row stride ldx); out is float32 [N, C] (or [N, 2C]); argmax (optional, int32 [N, C]) receives the FIRST pixel index of every
 * channel maximum (what F.max_pool2d's backward routes the gradient to).  Workspace: saunet_global_pool_workspace(N, HW, C) bytes.
 * backward: dx[n,p,c] = davg/HW + [p == argmax[n,c]] * dmax with (davg, dmax) read from dy according to the mode. */
int64_t saunet_global_pool_workspace(int N, int HW, int C);
int saunet_global_pool_forward(int dtype, int mode, const void* x, int N, int HW, int C, int ldx, float* out, int* argmax,
                               void* workspace, int64_t workspace_bytes, void* stream);
int saunet_global_pool_backward(int dtype, int mode, const float* dy, const int* argmax, int N, int HW, int C, void* dx, int lddx, void* stream);

/* ---- fused GatedSpatialConv2d (models/GSConv.py:16-57; call sites models/models.py:341-352) ----------------------
 * Replaces  cat -> BN(C+1) -> conv1x1(C+1,C+1) -> relu -> conv1x1(C+1,1) -> BN(1) -> sigmoid -> x*(alpha+1) -> conv1x1(C,C)
 * for C = 8/16/32 feature channels + 1 gating channel, bf16 storage, training-mode batch norm.  The chain is recomputed per
 * pixel (a wave owns 32 pixels per tile; the (C+1)x(C+1) products run on the matrix cores with the pixel's NHWC row as the
 * B operand and the float32 weights as bf16 hi + lo A fragments); intermediate maps never reach memory; backward recomputes
 * them.  Parameter vectors are float32:
 *   bn0 [4][C+1] = scale, shift, mean, invstd of BN(C+1) (saunet_bn_finalize layout);  bn1 [4][1] likewise for BN(1);
 *   w1 [C+1][C+1], b1 [C+1], w2 [C+1], b2 [1] the two gate convolutions;  wm [C][C] the module weight (no bias).
 * forward_z:   z[p] = w2 . relu(w1 . bn0(cat_p) + b1) + b2 (float32) and its sum / sum of squares (replicated float64
 *              accumulators, zeroed by the caller) for BN(1).
 * forward_out: alpha = sigmoid(bn1(z)) ; y = wm . (feat * (alpha + 1)).
 * backward_q:  q = dL/d bn1(z) per pixel; dwm [C][C], dbn1 = {dgamma1, dbeta1}, K[3] (dz = K0*q + K1 + K2*z).
 * backward_sums: dw1, db1, dw2, db2, dbn0 = {dgamma0 [C+1], dbeta0 [C+1]}, E [3][C+1] (dcat = E0*da0 + E1 + E2*cat).
 * backward_apply: dfeat [P][C], dgate [P].
 * workspace: saunet_gate_backward_workspace(pixels) bytes, shared by backward_q and backward_sums. */
int saunet_gate_forward_z(int dtype, int C, const void* feat, int ldf, const void* gate, int ldg, int64_t pixels, const float* bn0,
                          const float* w1, const float* b1, const float* w2, const float* b2, float* z, double* zsum, double* zsq,
                          int replicas, int rstride, void* stream);
int saunet_gate_forward_out(int dtype, int C, const void* feat, int ldf, const float* z, int64_t pixels, const float* bn1, const float* wm,
                            void* y, int ldy, void* alpha, void* stream);
int64_t saunet_gate_backward_workspace(int64_t pixels);
int saunet_gate_backward_q(int dtype, int C, const void* dy, int lddy, const void* feat, int ldf, const float* z, const void* dalpha,
                           int64_t pixels, const float* bn1, const float* wm, float* q, float* dwm, float* dbn1, float* K,
                           void* workspace, int64_t workspace_bytes, void* stream);
int saunet_gate_backward_sums(int dtype, int C, const void* feat, int ldf, const void* gate, int ldg, const float* q, const float* z,
                              int64_t pixels, const float* K, const float* bn0, const float* w1, const float* b1, const float* w2,
                              float* dw1, float* db1, float* dw2, float* db2, float* dbn0, float* E,
                              void* workspace, int64_t workspace_bytes, void* stream);
int saunet_gate_backward_apply(int dtype, int C, const void* dy, int lddy, const void* feat, int ldf, const void* gate, int ldg,
                               const float* q, const float* z, int64_t pixels, const float* K, const float* E, const float* bn0,
                               const float* w1, const float* b1, const float* w2, const float* bn1, const float* wm,
                               void* dfeat, int lddf, void* dgate, int lddg, void* stream);

/* ---- fused Conv2d(1,C,1x1) -> BatchNorm2d(C) -> ReLU on a one-channel float32 map (SAUNet.expand, models/models.py:316,367) ----
 * y_c = w_c*a + b_c makes the batch statistics of y analytic in those of a, so the layer is out_c = relu(A_c*a + B_c).
 * expand_coeff: from sum / sum-of-squares of a (replicated float64 accumulators as written by saunet_bn_stats) builds
 *   coef [4][C] = {A, B, k, gamma*invstd} and mu_var = {mean, biased variance of a}; updates the running statistics.
 *   Eval mode (training = 0) uses the running statistics instead (sum / sumsq may be NULL).
 * expand_forward: y [P][ldy] (float32 or bf16) from a [P].
 * expand_backward (training-mode statistics): sums = zeroed [replicas][2][C] float64 scratch; writes dw [C], db [C] (zeros; may be
 *   NULL), dgamma [C], dbeta [C], D [2] (scratch) and da [P] (float32). */
int saunet_expand_coeff(int C, const double* sum, const double* sumsq, int replicas, int rstride, double count, const float* w, const float* b,
                        const float* gamma, const float* beta, float eps, float momentum, float* rmean, float* rvar, float* coef, float* mu_var,
                        int training, void* stream);
int saunet_expand_forward(int dtype, const float* a, int64_t pixels, int C, const float* coef, void* y, int ldy, int relu, void* stream);
int saunet_expand_backward(int dtype, const void* dy, int lddy, const float* a, int64_t pixels, int C, const float* coef, const float* mu_var, int relu,
                           double* sums, int replicas, int rstride, float* dw, float* db, float* dgamma, float* dbeta, float* D, float* da, void* stream);

/* ---- dual attention tail (attention_blocks.py:50-57,165-173,237) ------------------------------
 * pooled[n,c] = mean_hw F ;  out = (S+1) * F * se[n,c] */
int saunet_global_avgpool(int dtype, const void* x, int N, int HW, int C, int ldx, float* pooled, void* stream);
int saunet_se_excite(const float* pooled, int N, int C, int Cr, const float* w1, const float* b1,
                     const float* w2, const float* b2, float* hidden, float* se, void* stream);
int saunet_se_excite_backward(const float* pooled, const float* hidden, const float* se, const float* dse, int N, int C, int Cr,
                              const float* w1, const float* w2, float* dpooled, float* dw1, float* db1, float* dw2, float* db2, void* stream);
int saunet_att_combine_forward(int dtype, const void* F, int ldf, const void* S, const float* se, void* out, int ldo,
                               int N, int HW, int C, void* stream);
/* dF = dout*(S+1)*se + dpooled/HW (second pass) ; dS = sum_c dout*F*se ; dse[n,c] = sum_hw dout*(S+1)*F */
int saunet_att_combine_backward(int dtype, const void* F, int ldf, const void* S, const float* se, const void* dout, int lddo,
                                void* dF, int lddf, void* dS, float* dse, int N, int HW, int C, void* stream);
int saunet_add_pooled_grad(int dtype, void* dF, int lddf, const float* dpooled, int N, int HW, int C, void* stream);

/* ---- dual-task loss + metrics (loss.py:124-159, :51-88; models/models.py:51-74) ---------------
 * one pass over logits [P,4] + edge [P,1]: partial sums -> sums[32] (float64, zeroed by caller):
 *  [0] sum w*nll  [1] sum w  [2..5] I_c  [6..9] K_c  [10] sum bce  [11..13] acc_num, acc_den, (unused)
 *  [14..16] |P_c & Y_c| c=1..3   [17..19] |Y_c|   [20..22] |P_c| */
int saunet_dual_loss_forward(int dtype, const void* logits, int ldl, const void* edge, const int64_t* seg_t, const float* edge_t,
                             int64_t pixels, double* sums, void* stream);
/* loss[0] = dice + ce + bce ; metrics[0]=acc, [1..3]=jaccard */
int saunet_dual_loss_finalize(const double* sums, int64_t pixels, float* loss, float* metrics, void* stream);
int saunet_dual_loss_backward(int dtype, const void* logits, int ldl, const void* edge, const int64_t* seg_t, const float* edge_t,
                              int64_t pixels, const double* sums, const float* dloss, void* dlogits, int lddl, void* dedge, void* stream);
/* inference head (models/models.py:96-109 `softmax(pred, dim=1)`; train.py:47 argmax): prob [P][C] float32 (row stride ldp) and / or
 * label [P] int64 = first maximum; either output may be NULL.  C in {2, 4, 8}. */
int saunet_softmax_argmax(int dtype, const void* logits, int ldl, int64_t pixels, int C, float* prob, int ldp, int64_t* label, void* stream);
/* SegmentationModuleBase.pixel_acc(pred, label, num_class) (/root/reference/models/models.py:51-74) for a caller that holds a prediction tensor:
 * pred = class scores of N x HW pixels, element (n, c, p) at pred[n*stride_n + c*stride_c + p*stride_p] (NCHW: HW*C, HW, 1; NHWC: HW*C, 1, C),
 * kind 0 float32 / 1 bf16 / 2 int64 / 3 uint8; the class of a pixel is the FIRST maximum (torch.max).  label [N*HW] int64.
 * counts: 2 + 3*(C-1) zero-initialised uint64 (exact integer counts: deterministic).  out[0] = acc over the pixels with label >= 1,
 * out[c] = Jaccard of class c = 1 .. C-1, computed in float32 with the reference's +1e-10.  C <= 16. */
int saunet_pixel_metrics(int kind, const void* pred, int64_t stride_n, int64_t stride_c, int64_t stride_p, const int64_t* label, int64_t N, int64_t HW,
                         int C, void* counts, float* out, void* stream);
/* SegmentationModuleBase.jaccard(pred, label) (models/models.py:76-78): sum(long(pred) & label) / (sum(pred) + sum(label) - sum(long(pred) & label)) over n
 * elements.  sums: 24 zero-initialised bytes (two int64 + one float64).  out[0] float32 (0/0 -> nan, like the reference). */
int saunet_binary_jaccard(int kind, const void* pred, const int64_t* label, int64_t n, void* sums, float* out, void* stream);


/* ---- Canny branch on device (models/models.py:359-363; replaces the host cv2.Canny round trip) --
 * out[n,h,w] in {0,255} as dtype.  work: int32 [N][3][H][W] scratch. */
int saunet_canny(int dtype, const float* image_nchw3, int N, int H, int W, int low, int high, void* out, int32_t* work, void* stream);

/* edge ground truth [N,1,H,W] in {0,1} from labels [N,H,W] (int64): radius-2 distance-transform edges of classes
 * 1..num_classes, bit-identical to data/ac17_dataloader.py:231-258 (mask_to_onehot + onehot_to_binary_edges). */
int saunet_mask_to_edges(const int64_t* seg, int N, int H, int W, int num_classes, float* edge, void* stream);
/* test-set post-processing (test_and_pack.py:31-76: undo_crop + order-0 resize to the original grid) as one gather:
 * out[z][Y][X] = p[floor((Y+.5)*h/H)][floor((X+.5)*w/W)],  p[y][x] = pred[z][by0+y-top][bx0+x-left] inside the cw x ch window, else 0.
 * pred int64 [Z][th][tw] (argmax labels), out uint8 [Z][H][W]; the geometry comes from the host (postprocess.undo_crop_geometry). */
int saunet_labels_uncrop_resize(const int64_t* pred, int Z, int th, int tw, int bx0, int by0, int cw, int ch, int left, int top,
                                int w, int h, int W, int H, unsigned char* out, void* stream);

/* ---- training-time augmentation on the device (data/augmentations.py:223-264, 308-331, 392-412; data/ac17_dataloader.py:22-57,
 * 139-150, 196-216, 260-287) over a zero-padded batch of raw slices [B][Hm][Wm] (float32 image, float32 mask) -------------------------
 * geometric: PaddingCenterCrop(S) + flips + rotation as ONE gather per output pixel (image bilinear, mask nearest, fill 0).  params is a
 *   device array of B records {int32 h, w, oy, ox, hflip, vflip, rotate; float cos, sin}: (h, w) the slice size, (oy, ox) the crop/pad offset
 *   (source = output + offset), rotate = 0 skips the affine map.
 * gamma_zscore: per slice x <- zscore(((x-min)/(max-min+1e-7))^gamma * (max-min) + min), population std (+1e-10); gamma <= 0: z-score only.
 * uniform_noise / gauss_blur / elastic_warp: displacement = gaussian_filter(2u-1, sigma, zero boundary) * alpha (weights = normalised half
 *   kernel [0..radius]); out(r,c) = in(r+drow, c+dcol), order 1, edge replication; apply[b] = 0 copies slice b through; mask outputs:
 *   seg_f (interpolated), seg_long = trunc, seg_edge = the value where it is an exact class id 1..3 else 0 (input of saunet_mask_to_edges). */
int saunet_augment_geometric(const float* img, const float* seg, int B, int Hm, int Wm, const void* params, int S, float* out_img, float* out_seg, void* stream);
int saunet_augment_gamma_zscore(float* x, int B, int npix, const float* gamma, void* stream);
int saunet_uniform_noise(uint64_t seed, float* out, int64_t n, void* stream);
int saunet_gauss_blur(const float* in, float* tmp, float* out, int B, int H, int W, const float* weights, int radius, int affine_2u_minus_1, float scale, void* stream);
int saunet_elastic_warp(const float* img, const float* seg, const float* drow, const float* dcol, const int* apply, int B, int H, int W,
                        float* out_img, float* seg_f, int64_t* seg_long, int64_t* seg_edge, void* stream);



/* ---- optimiser (train.py:166-216, radam.py:5-78) as multi-tensor kernels ------------------------*/
typedef struct saunet_tensor_list { int32_t count; const void* ptrs[4][96]; int64_t numel[96]; } saunet_tensor_list;
/* hyper-parameters live in a DEVICE float array so a captured hipGraph can be replayed after the host
 * rewrites the learning rate:
 *   SGD   hyper = {lr, momentum, weight_decay, first_step(0/1), grad_scale}
 *   RAdam hyper = {beta1, beta2, eps, weight_decay*lr, step_size, rectified(0/1), grad_scale}  (radam.py:52-75)
 *   Adam  hyper = {beta1, beta2, eps, weight_decay, lr/(1-beta1^t), 1/sqrt(1-beta2^t), grad_scale}  (torch.optim.Adam as built at
 *                 train.py:197-201: amsgrad off, L2 weight decay) */
int saunet_sgd_step(const saunet_tensor_list* tl /*0:param 1:grad 2:momentum*/, const float* hyper, void* stream);
int saunet_radam_step(const saunet_tensor_list* tl /*0:param 1:grad 2:exp_avg 3:exp_avg_sq*/, const float* hyper, void* stream);
int saunet_adam_step(const saunet_tensor_list* tl /*0:param 1:grad 2:exp_avg 3:exp_avg_sq*/, const float* hyper, void* stream);
/* flat[offset_i : offset_i+n_i] = grad_i (pack=1) or grad_i = flat[...]*scale (pack=0): all-reduce buckets */
int saunet_bucket_copy(const saunet_tensor_list* tl /*0:tensor 1:flat+offset*/, int pack, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
