/*
 * ffwm_hip.h -- C ABI of libffwm_hip.so, the MI355X (gfx950) implementation of the
 * flow-guided feature-warping hot path of csyxwei/FFWM.
 *
 * This is the drop-in boundary: each entry point below replaces one native function the
 * reference binds through pybind11 (citations are into the reference repository).  The
 * reference passes `at::Tensor&`; here every tensor is a raw device pointer to a
 * CONTIGUOUS NCHW buffer plus its sizes, so the library has no PyTorch dependency:
 *
 *   ffwm_block_extractor_forward    <- block_extractor_cuda.forward     cuda/block_extractor/block_extractor_cuda.cc:5-12
 *   ffwm_block_extractor_backward   <- block_extractor_cuda.backward    cuda/block_extractor/block_extractor_cuda.cc:14-25
 *   ffwm_local_attn_reshape_forward <- local_attn_reshape_cuda.forward  cuda/local_attn_reshape/local_attn_reshape_cuda.cc:5-11
 *   ffwm_local_attn_reshape_backward<- local_attn_reshape_cuda.backward cuda/local_attn_reshape/local_attn_reshape_cuda.cc:13-21
 *   ffwm_resample2d_forward         <- resample2d_cuda.forward          cuda/resample2d_package/resample2d_cuda.cc:6-14
 *   ffwm_resample2d_backward        <- resample2d_cuda.backward         cuda/resample2d_package/resample2d_cuda.cc:16-26
 *   ffwm_warp_forward / _backward   <- WarpNet.forward (F.grid_sample)  models/base_networks.py:168-173, fused with the
 *                                      flip + concat of FFWM.forward    models/base_networks.py:326-329
 *   ffwm_guided_filter_*            <- GuidedFilter.forward             models/external_function.py:239-277
 *   ffwm_affine_regularization      <- AffineRegularizationLoss.__call__ models/losses.py:200-219
 *   ffwm_correlation_colmax         <- max(bmm(source, target), dim=1)  models/losses.py:347-353
 *   ffwm_block_attention_*          <- avg_pool2d(BlockExtractor * LocalAttnReshape)  (composition of the ops above)
 *   ffwm_spectral_norm_*            <- torch.nn.utils.spectral_norm hooks models/base_networks.py:5,218-264,381-413
 *   ffwm_conv3x3_wgrad[_block]      <- convolution_backward grad_weight / grad_bias of the nn.Conv2d(., ., 3, 1, 1) layers
 *   ffwm_bn_lrelu_*                 <- nn.BatchNorm2d (training) + nn.LeakyReLU of the conv blocks  models/base_networks.py:12-31
 *   ffwm_mfm_*                      <- mfm.forward (split + max)        lightcnn/light_cnn.py
 *   ffwm_bias_relu_forward          <- conv bias add + nn.ReLU of VGG19  models/losses.py:398-519
 *   ffwm_adam_step                  <- torch.optim.Adam.step            models/ffwm_model.py:46-49,151-160
 *   ffwm_l1_multi                   <- the w * F.l1_loss(x * m, y * m) terms of backward_G   models/ffwm_model.py:107-139
 *   ffwm_conv2d_wgrad[_tiled]       <- convolution_backward grad_weight of the stride-2 / transposed / small-plane convs
 *
 * Conventions
 *   - dtype: FFWM_F32 or FFWM_F64 (the reference dispatches AT_DISPATCH_FLOATING_TYPES).
 *   - stream: a hipStream_t passed as void* (NULL = the null stream).  Launches are
 *     asynchronous, never synchronise, never allocate; the library keeps no tensor state.
 *   - The kernels run on the CURRENT HIP device of the calling thread; the caller selects the
 *     device that owns the pointers (one process per GPU in this project).
 *   - Ownership: the caller allocates every output.  Forward outputs are fully overwritten
 *     (no zero-fill needed).  Gradient outputs marked "+=" are accumulated into, exactly as
 *     the reference's atomicAdd kernels do, so the caller zero-fills them first
 *     (models/external_function.py:49-50,96,137-138).  A NULL gradient pointer skips that
 *     gradient.
 *   - Return: 0 on success, a negative ffwm_status otherwise; ffwm_last_error() returns a
 *     thread-local message.  (The reference returns 1 unconditionally and checks nothing.)
 *   - Index arithmetic is 64-bit for flat offsets (the reference overflows 32-bit `int`
 *     above 2^31 elements); a single H*W plane must stay below 2^31 elements.
 */
#ifndef FFWM_HIP_H_
#define FFWM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFWM_ABI_VERSION 5

typedef enum {
    FFWM_OK = 0,
    FFWM_ERR_ARG = -1,     /* NULL pointer, non-positive size, bad kernel_size/dilation */
    FFWM_ERR_DTYPE = -2,   /* dtype is not FFWM_F32 / FFWM_F64 */
    FFWM_ERR_SIZE = -3,    /* a plane exceeds 2^31 elements */
    FFWM_ERR_LAUNCH = -4   /* HIP reported a launch error */
} ffwm_status;

typedef enum { FFWM_F32 = 0, FFWM_F64 = 1 } ffwm_dtype;

int ffwm_abi_version(void);
const char* ffwm_last_error(void);

/* out[B,C,k*Hf,k*Wf][b,c,yf*k+i,xf*k+j] = bilinear(source[b,c], yf + flow[b,1,yf,xf] + i - k/2,
 *                                                  xf + flow[b,0,yf,xf] + j - k/2)
 * index-clamped border, weights from the unclamped fraction.
 * source[B,C,Hs,Ws], flow_field[>=B,2,Hf,Wf] (pixel units, ch0 = x, ch1 = y).
 * Reference: block_extractor_kernel.cu:21-85. */
int ffwm_block_extractor_forward(const void* source, const void* flow_field, void* output,
                                 int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf,
                                 int64_t Wf, int kernel_size, int dtype, void* stream);

/* grad_source[B,C,Hs,Ws] += 4-tap scatter of grad_output; grad_flow_field[B,2,Hf,Wf] += sum over
 * c and the k x k window of grad_output * d(bilinear)/d(x,y).  Either may be NULL.
 * Reference: block_extractor_kernel.cu:89-170. */
int ffwm_block_extractor_backward(const void* source, const void* flow_field,
                                  const void* grad_output, void* grad_source,
                                  void* grad_flow_field, int64_t B, int64_t C, int64_t Hs,
                                  int64_t Ws, int64_t Hf, int64_t Wf, int kernel_size,
                                  int dtype, void* stream);

/* out[B,1,k*H,k*W][b,0,y,x] = inputs[B,k*k,H,W][b,(y%k)*k + x%k, y/k, x/k].
 * Reference: local_attn_reshape_kernel.cu:21-61. */
int ffwm_local_attn_reshape_forward(const void* inputs, void* output, int64_t B, int64_t H,
                                    int64_t W, int kernel_size, int dtype, void* stream);

/* Inverse permutation.  accumulate != 0: grad_inputs += (reference semantics, atomicAdd into a
 * zero-filled buffer, local_attn_reshape_kernel.cu:66-108); accumulate == 0: grad_inputs is
 * overwritten (no zero-fill, half the traffic -- identical result on a zero-filled buffer). */
int ffwm_local_attn_reshape_backward(const void* grad_output, void* grad_inputs, int64_t B,
                                     int64_t H, int64_t W, int kernel_size, int accumulate,
                                     int dtype, void* stream);

/* grad_output through its ELEMENT STRIDES (ABI 5).  The reference's kernels index gradOutput with DIM3_INDEX and the tensor's own strides
 * (cuda/block_extractor/block_extractor_kernel.cu:8-15, local_attn_reshape_kernel.cu:8-15, resample2d_kernel.cu:8-15) and its Functions
 * throw the result of grad_output.contiguous() away (models/external_function.py:46-47,93-94,132-133), so a pybind module that replaces
 * the reference's must take an expanded / permuted / sliced gradient as it comes.  grad_output_strides[4] = (batch, channel, row, column)
 * strides in elements, all >= 0, column stride != 0; NULL or the contiguous NCHW strides = the entry points above (the tuned kernels).
 * Any other layout runs the per-element kernels: the reference's own decomposition -- correct for every view, not a tuned path.
 * Everything else as in the plain entry points (ffwm_block_extractor_backward, ffwm_local_attn_reshape_backward, ffwm_resample2d_backward). */
int ffwm_block_extractor_backward_strided(const void* source, const void* flow_field, const void* grad_output,
                                          const int64_t* grad_output_strides, void* grad_source, void* grad_flow_field, int64_t B,
                                          int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int kernel_size, int dtype,
                                          void* stream);
int ffwm_local_attn_reshape_backward_strided(const void* grad_output, const int64_t* grad_output_strides, void* grad_inputs, int64_t B,
                                             int64_t H, int64_t W, int kernel_size, int accumulate, int dtype, void* stream);
int ffwm_resample2d_backward_strided(const void* input1, const void* input2, const void* grad_output,
                                     const int64_t* grad_output_strides, void* grad_input1, void* grad_input2, int64_t B, int64_t C,
                                     int64_t Hi, int64_t Wi, int64_t H, int64_t W, int kernel_size, int dilation,
                                     int reference_quirk, int dtype, void* stream);

/* Gaussian-weighted kernel_size x kernel_size tap resampler.  input1[>=B,C,Hi,Wi],
 * input2[B,3,H,W] = (dx, dy, sigma) in pixels, out[B,C,H,W].
 * Reference: resample2d_kernel.cu:21-95.  kernel_size even >= 2, dilation >= 1. */
int ffwm_resample2d_forward(const void* input1, const void* input2, void* output, int64_t B,
                            int64_t C, int64_t Hi, int64_t Wi, int64_t H, int64_t W,
                            int kernel_size, int dilation, int dtype, void* stream);

/* grad_input1[B,C,Hi,Wi] += scatter (resample2d_kernel.cu:98-202); grad_input2[B,3,H,W] is
 * OVERWRITTEN (resample2d_kernel.cu:204-330).  reference_quirk != 0 keeps the reference's
 * `alpha = xf - int(xf)` truncation in the grad_input1 weights (:137-138); 0 uses floor (the
 * true gradient).  reference_quirk bit 1 (value 2, ABI 5): grad_input1 arrives UNINITIALISED and is overwritten -- the
 * owned-tile kernels store every cell exactly once (no zero-fill by the caller, no atomics on the regular path), every
 * other path clears the buffer itself first.  Either gradient may be NULL. */
int ffwm_resample2d_backward(const void* input1, const void* input2, const void* grad_output,
                             void* grad_input1, void* grad_input2, int64_t B, int64_t C,
                             int64_t Hi, int64_t Wi, int64_t H, int64_t W, int kernel_size,
                             int dilation, int reference_quirk, int dtype, void* stream);

/* WarpNet: out[b,c,y,x] = grid_sample(feat[B,C,Hi,Wi], flow[B,2,H,W]) bilinear, zeros padding,
 * align_corners=False; flow = normalised absolute coordinates, ch0 = x, ch1 = y.
 * flipcat != 0: out is [B,2C,H,W] = cat(w, flip(w, dim 3)) (one read, two writes instead of
 * grid_sample + flip + cat).  Reference: models/base_networks.py:168-173,326-329. */
int ffwm_warp_forward(const void* feat, const void* flow, void* output, int64_t B, int64_t C,
                      int64_t Hi, int64_t Wi, int64_t H, int64_t W, int flipcat, int dtype,
                      void* stream);

/* grad_feat[B,C,Hi,Wi] += ; grad_flow[B,2,H,W] += .  Either may be NULL.  grad_output is
 * [B,C,H,W] or, with flipcat bit 0, [B,2C,H,W].  flipcat bit 1 (value 2, ABI 4): grad_feat is UNINITIALISED memory and is produced
 * whole -- the caller's zero-fill is saved (and, on the owned-tile path of planes beyond LDS, the read of it); grad_flow still
 * accumulates. */
int ffwm_warp_backward(const void* feat, const void* flow, const void* grad_output,
                       void* grad_feat, void* grad_flow, int64_t B, int64_t C, int64_t Hi,
                       int64_t Wi, int64_t H, int64_t W, int flipcat, int dtype, void* stream);

/* Several independent warps in one call (at most a few launches): FFWM issues its warps in groups -- the eight 32 x 32
 * part crops of models/ffwm_model.py:84-88, the three illumination warps of models/losses.py:149, the three
 * warp-attention levels of models/base_networks.py:323-333 -- whose members are individually launch-bound.  Same
 * semantics per problem as ffwm_warp_forward / ffwm_warp_backward (grad_feat / grad_flow accumulate, either may be NULL;
 * forward ignores the three gradient fields).  All problems share flipcat and dtype. */
typedef struct ffwm_warp_problem {
    const void* feat;        /* [B,C,Hi,Wi] */
    const void* flow;        /* [B,2,H,W] */
    void* output;            /* forward: [B,C,H,W] or [B,2C,H,W] */
    const void* grad_output; /* backward */
    void* grad_feat;         /* backward, may be NULL */
    void* grad_flow;         /* backward, may be NULL */
    int64_t B, C, Hi, Wi, H, W;
} ffwm_warp_problem;
int ffwm_warp_multi_forward(const ffwm_warp_problem* problems, int n, int flipcat, int dtype, void* stream);
int ffwm_warp_multi_backward(const ffwm_warp_problem* problems, int n, int flipcat, int dtype, void* stream);

/* ---- the L1 terms of the generator loss in one launch (csrc/l1_loss.hip) ------------------------
 * models/ffwm_model.py:107-139 (backward_G) sums ~25 terms  w * F.l1_loss(x * m, y * m)  -- pixel loss :112-115, PerceptualLoss
 * (models/losses.py:293-320), MSL1Loss (:130-157), IdentityLoss (:76-112).  A problem is one term:
 *     out[slot] += scale * sum_i | x[i] * m[mi] - y[i] * m[mi] |      (scale = w / numel; mask NULL: m = 1)
 * with the mask broadcast over the channels: x [B, C, H, W] (n = B*chw elements, chw = C*H*W), mask [B, 1, H, W] (hw = H*W).
 * Forward (grad_out NULL): out[n_slots] is accumulated into (zero-fill it).  Backward (grad_out[n_slots] given):
 * grad_x = grad_out[slot] * scale * sign(x m - y m) * m is written for every problem (y is data: no gradient). */
typedef struct ffwm_l1_problem {
    const void* x;
    const void* y;
    const void* mask;       /* may be NULL */
    void* grad_x;           /* backward only */
    int64_t n, chw, hw;
    double scale;
    int slot;
} ffwm_l1_problem;
int ffwm_l1_multi(const ffwm_l1_problem* problems, int n, void* out, const void* grad_out, int n_slots, int dtype, void* stream);

/* ---- batched spectral normalisation of conv weights (netG / netD) -----------------------------
 * Replaces the per-layer hook of torch.nn.utils.spectral_norm that the reference wraps around every
 * convolution of FFWM and MSDiscriminator (models/base_networks.py:5,218-264,381-413):
 *     power_iterations x { v = normalize(W^T u); u = normalize(W v) };  sigma = u . (W v);
 *     weight_sn = W / sigma                         with W = weight.view(rows = Cout, cols = -1)
 * Three launches handle FFWM_SN_MAX_LAYERS layers (grids over (layer, chunk) pairs).  power_iterations
 * is 0 or 1 (the hook's default).  u[rows] and v[cols] are
 * updated IN PLACE (as the hook does in training mode) and copied to u_saved / v_saved, because a
 * later forward call may overwrite them before this call's backward runs (the hook clones them for
 * the same reason); power_iterations = 0 is the eval-mode behaviour (stored u, v).  wv[rows] is caller-provided scratch, sigma[1] receives sigma (needed by
 * the backward).  The `layers` array lives in HOST memory; all pointers inside are device pointers. */
#define FFWM_SN_MAX_LAYERS 32
typedef struct {
    const void* weight;   /* [rows, cols] */
    void* u;              /* [rows]  in/out */
    void* v;              /* [cols]  in/out */
    void* wv;             /* [rows]  scratch */
    void* weight_sn;      /* [rows, cols] out */
    void* sigma;          /* [1] out */
    void* u_saved;        /* [rows] out: the u this call normalised with (for its backward); may be NULL */
    void* v_saved;        /* [cols] out: likewise */
    int rows, cols;
} ffwm_sn_layer;

int ffwm_spectral_norm_forward(const ffwm_sn_layer* layers, int n_layers, int power_iterations,
                               double eps, int dtype, void* stream);

/* grad_weight = grad_weight_sn / sigma - (<grad_weight_sn, weight> / sigma^2) u v^T   (OVERWRITTEN;
 * u, v are constants, exactly as in the hook, where the power iteration runs under no_grad). */
typedef struct {
    const void* weight;          /* [rows, cols] */
    const void* u;               /* [rows] */
    const void* v;               /* [cols] */
    const void* sigma;           /* [1] */
    const void* grad_weight_sn;  /* [rows, cols] */
    void* grad_weight;           /* [rows, cols] out */
    void* partials;              /* scratch, ceil(rows*cols / 8192) elements */
    int rows, cols;
} ffwm_sn_grad_layer;

int ffwm_spectral_norm_backward(const ffwm_sn_grad_layer* layers, int n_layers, int dtype,
                                void* stream);

/* ---- guided filter (illumination-adaption path) ------------------------------------------------
 * out = GuidedFilter(r, eps)(x, y) of models/external_function.py:239-277 (box filters as cumsum
 * differences, :164-193), per [H, W] plane; x, y, out are [planes = B*C, H, W] contiguous with
 * c_x == c_y (the only way FFWM calls it: models/ffwm_model.py:57-59,81,104-105).  H, W <= 128,
 * H > 2r+1, W > 2r+1.  `saved` [5, planes, H, W] receives mean_x, mean_y, A, var_x+eps, mean_A for the
 * backward.  Four launches (column pass / row pass + pointwise stage, twice), each over planes x W/32 x quantities or
 * planes x H/4 workgroups; the PyTorch module issues ~100. */
int ffwm_guided_filter_forward(const void* x, const void* y, void* output, void* saved,
                               int64_t planes, int64_t H, int64_t W, int r, double eps, int dtype,
                               void* stream);

/* grad_x (OVERWRITTEN) of the above; y is data (the ground-truth image) and gets no gradient.
 * `workspace`: [2, planes, H, W] elements of the tensors' dtype, contents irrelevant on entry and exit. */
int ffwm_guided_filter_backward(const void* x, const void* y, const void* saved,
                                const void* grad_output, void* grad_x, void* workspace,
                                int64_t planes, int64_t H, int64_t W, int r, int dtype, void* stream);

/* ---- fused affine regularisation (FlowNet pre-training) -----------------------------------------
 * AffineRegularizationLoss.__call__ of models/losses.py:200-219 for one flow scale in ONE launch:
 *   loss_sum[0] += sum over b, both coordinate grids g = (flow + 1) / 2 * 128 and every kz x kz window p
 *                  of g of  p^T (K^T K) p            (the caller multiplies by loss_scale = 1 / (B h' w'))
 *   grad_flow  += d(loss_scale * that sum) / d flow   (may be NULL)
 * flow[B,2,h,w]; ktk[kz^2, kz^2] = the reference's `self.kernel` (K^T K, losses.py:192-198) in the
 * tensors' dtype; kernel_size in {3, 5, 7} (models/flownet_model.py:31).  The reference reaches the same
 * numbers through conv2d -> local_attn_reshape -> block_extractor -> avg_pool2d (those entry points remain;
 * ffwm_amd/losses.py composes them like the reference when `fused=False`). */
int ffwm_affine_regularization(const void* flow, const void* ktk, void* loss_sum, void* grad_flow,
                               int64_t B, int64_t h, int64_t w, int kernel_size, double loss_scale,
                               int dtype, void* stream);

/* ---- correlation column maximum (correctness loss of FlowNet pre-training) ----------------------
 * out[b, j] = max_i sum_k source[b, i, k] * target[b, k, j]   -- torch.max(torch.bmm(source_norm,
 * target_norm), dim=1)[0] of PerceptualCorrectness.calculate_loss (models/losses.py:347-353) without
 * the [B, N, N] matrix (1 GiB per sample at relu1_1), on fp32-in / fp32-accumulate MFMA.
 * source[B,N,C], target[B,C,N] contiguous float32, C in {64, 128, 256}, out[B,N]. */
int ffwm_correlation_colmax(const void* source, const void* target, void* out, int64_t B, int64_t N,
                            int64_t C, int dtype, void* stream);

/* Fused extractor + attention consumer (the GFLA-style local attention the cfg-5 shape
 * models): what the reference API can only express as
 *     avg_pool2d(BlockExtractor(source, flow, k) * LocalAttnReshape(weights, k), k, k)
 * (models/external_function.py:14-101 + F.avg_pool2d, the same composition models/losses.py:211-219 uses),
 * without the k^2-fold expanded tensors:
 *     out[b,c,y,x] = (sum_ij sample_ij(b,c,y,x) * weights[b, i*k + j, y, x]) / k^2,
 * sample_ij = the block_extractor value at (y*k + i, x*k + j).  Products and the row-major sum are rounded
 * separately and divided once, like the three ATen/extension kernels of the composition.
 * source[B,C,Hs,Ws], flow_field[>=B,2,Hf,Wf], weights[B,k*k,Hf,Wf], output[B,C,Hf,Wf] (overwritten). */
int ffwm_block_attention_forward(const void* source, const void* flow_field, const void* weights, void* output,
                                 int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf,
                                 int kernel_size, int dtype, void* stream);

/* grad_source[B,C,Hs,Ws] +=, grad_flow_field[B,2,Hf,Wf] +=, grad_weights[B,k*k,Hf,Wf] += for
 * grad_output[B,C,Hf,Wf]; any of the three may be NULL.  The extractor's grad_output window is
 * (grad_output / k^2) * weights_ij -- a channel-independent window times one number per channel, so (round 6) grad_source comes from
 * per-pixel cell coefficients Wy^T w Wx scaled per channel, grad_flow_field and grad_weights from P = sum_c (g_c / k^2) S_c over the
 * (k+1)^2 source neighbourhood: two launches for fp32 / k = 3 with grad_source wanted (csrc/block_extractor.hip: ba_bwd_src_kernel,
 * ba_bwd_pix_kernel), the per-element kernel otherwise. */
int ffwm_block_attention_backward(const void* source, const void* flow_field, const void* weights,
                                  const void* grad_output, void* grad_source, void* grad_flow_field,
                                  void* grad_weights, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf,
                                  int64_t Wf, int kernel_size, int dtype, void* stream);

/* Weight (and bias) gradient of a 3x3 / stride 1 / pad 1 / dilation 1 / groups 1 convolution (the layer type of
 * the FFWM and FlowNet conv stacks, models/base_networks.py:59-165,274-347; the grad_weight / grad_bias outputs of
 * ATen's convolution_backward) on fp32-in / fp32-accumulate MFMA, NCHW, no transposes:
 *     grad_weight[K,C,3,3][k,c,r,s] += sum_{b,y,x} grad_output[b,k,y,x] * input[b,c,y+r-1,x+s-1]
 *     grad_bias[K][k]               += sum_{b,y,x} grad_output[b,k,y,x]        (NULL: skipped)
 * input[B,C,H,W], grad_output[B,K,H,W] contiguous float32, W a multiple of 64; the caller zero-fills
 * grad_weight and grad_bias (pixel slices are combined with atomics). */
int ffwm_conv3x3_wgrad(const void* input, const void* grad_output, void* grad_weight, void* grad_bias, int64_t B,
                       int64_t C, int64_t K, int64_t H, int64_t W, int dtype, void* stream);

/* The same, restricted to the block grad_weight[k_begin:k_end, c_begin:c_end] and grad_bias[k_begin:k_end] (tensor
 * extents and strides are still K and C). */
int ffwm_conv3x3_wgrad_block(const void* input, const void* grad_output, void* grad_weight, void* grad_bias, int64_t B,
                             int64_t C, int64_t K, int64_t H, int64_t W, int64_t k_begin, int64_t k_end, int64_t c_begin,
                             int64_t c_end, int dtype, void* stream);

/* Training-mode BatchNorm2d fused with the LeakyReLU that follows it (the conv blocks of
 * models/base_networks.py:12-31,207-264,381-413): y = leaky_relu(F.batch_norm(x, running_mean, running_var, weight,
 * bias, training=True, momentum, eps), negative_slope).  x, y [B,C,H,W] contiguous float32 (16-byte aligned),
 * HW = H*W; weight / bias [C] or NULL; running_mean / running_var [C] updated in place (both NULL: not tracked);
 * save_mean / save_invstd [C] are written for the backward pass.  scratch: NULL, or 2*C + (C+1)/2 ZERO-FILLED doubles (two sums
 * per channel, then one 32-bit arrival counter per channel) -- with it a channel with few workgroups' worth of parallelism
 * (C < 1024) is split over several workgroups (two launches).  The kernels leave the scratch ZERO-FILLED again (the last
 * workgroup of a channel to read the sums clears them): one buffer serves every later call on the same stream. */
int ffwm_bn_lrelu_forward(const void* x, const void* weight, const void* bias, void* running_mean, void* running_var,
                          void* y, void* save_mean, void* save_invstd, void* scratch, int64_t B, int64_t C, int64_t HW,
                          double eps, double momentum, double negative_slope, int dtype, void* stream);

/* grad_x [B,C,H,W], grad_weight [C], grad_bias [C] are OVERWRITTEN (each channel is produced by one workgroup); any
 * of them may be NULL.  The activation mask is recomputed from x, the forward output is not needed. */
int ffwm_bn_lrelu_backward(const void* x, const void* grad_out, const void* weight, const void* bias,
                           const void* save_mean, const void* save_invstd, void* grad_x, void* grad_weight,
                           void* grad_bias, void* scratch, int64_t B, int64_t C, int64_t HW, double negative_slope,
                           int dtype, void* stream);

/* The tail of a residual block, activ(blocks(x) + input(x)) with blocks ending in a training-mode BatchNorm2d and input = a 1x1
 * convolution (models/base_networks.py:207-233), as one kernel per direction: y = act(BatchNorm(x) + res + rbias[c]) where res is the
 * shortcut convolution's output WITHOUT its bias and rbias that bias (NULL: none); act 0 = LeakyReLU(negative_slope), 1 = sigmoid.
 * Statistics, running buffers, save_mean / save_invstd and scratch as in ffwm_bn_lrelu_forward.  Replaces, per block and direction,
 * the library's BatchNorm kernel, the GEMM path's separate bias add and the add + activation pass. */
int ffwm_bn_res_act_forward(const void* x, const void* weight, const void* bias, void* running_mean, void* running_var,
                            const void* res, const void* rbias, void* y, void* save_mean, void* save_invstd, void* scratch,
                            int64_t B, int64_t C, int64_t HW, double eps, double momentum, double negative_slope, int act,
                            int dtype, void* stream);
/* grad_res = grad_out * act'(.) (taken from the saved output y; also the gradient of `input(x)`), grad_x = BatchNorm backward of
 * it, grad_weight / grad_bias = the BatchNorm's affine gradients (grad_bias is the gradient of rbias too).  grad_x, grad_res,
 * grad_weight, grad_bias may be NULL. */
int ffwm_bn_res_act_backward(const void* x, const void* y, const void* grad_out, const void* weight, const void* save_mean,
                             const void* save_invstd, void* grad_x, void* grad_res, void* grad_weight, void* grad_bias,
                             void* scratch, int64_t B, int64_t C, int64_t HW, double negative_slope, int act, int dtype,
                             void* stream);

/* LightCNN's max-feature-map activation (lightcnn/light_cnn.py `mfm.forward`: torch.split + torch.max):
 * y[B,C,HW] = max(x[B,0:C,HW] + bias[0:C], x[B,C:2C,HW] + bias[C:2C]); bias [2C] or NULL (the bias of the layer in front,
 * folded in: bit-identical to adding it first); backward grad_x[B,2C,HW] (overwritten) with ATen's tie rule for
 * `maximum` (equal halves share the gradient).  Contiguous float32. */
int ffwm_mfm_forward(const void* x, const void* bias, void* y, int64_t B, int64_t C, int64_t HW, int dtype, void* stream);
int ffwm_mfm_backward(const void* x, const void* bias, const void* grad_y, void* grad_x, int64_t B, int64_t C, int64_t HW,
                      int dtype, void* stream);

/* ---- residual-block tails and the warp-attention gate of netG (models/base_networks.py:207-233, 326-333) ----------------
 * ResidualBlock.forward = activ(blocks(x) + input(x)) and FFWM.forward's `skip = skip * att_i(skip)` (att_i ends in a sigmoid
 * ResidualBlock) as one pass each instead of 2-3 element-wise launches.  Contiguous float32, n elements, 16-byte aligned.
 *   ffwm_add_act_forward:       y = act(a + b), act 1 = LeakyReLU(negative_slope > 0), 3 = sigmoid (y may alias a or b)
 *   ffwm_add_act_backward:      grad_z = grad_y * act'(a + b), formed from y alone (grad_z may alias grad_y); it is the gradient of
 *                               a AND of b
 *   ffwm_sigmoid_gate_forward:  att = sigmoid(a + b), y = x * att
 *   ffwm_sigmoid_gate_backward: grad_z = grad_y * x * att * (1 - att) (gradient of a and of b), grad_x = grad_y * att
 * ATen's operation order throughout (results equal the PyTorch composition's bit for bit up to expf). */
int ffwm_add_act_forward(const void* a, const void* b, void* y, int64_t n, int act, double negative_slope, int dtype, void* stream);
int ffwm_add_act_backward(const void* y, const void* grad_y, void* grad_z, int64_t n, int act, double negative_slope, int dtype,
                          void* stream);
int ffwm_sigmoid_gate_forward(const void* a, const void* b, const void* x, void* att, void* y, int64_t n, int dtype, void* stream);
int ffwm_sigmoid_gate_backward(const void* x, const void* att, const void* grad_y, void* grad_z, void* grad_x, int64_t n, int dtype,
                               void* stream);

/* y[B,C,HW] = relu(h[B,C,HW] + bias[C]) -- the bias add and ReLU behind the frozen VGG19 convs (models/losses.py:398-519)
 * as one pass; y may alias h. */
int ffwm_bias_relu_forward(const void* h, const void* bias, void* y, int64_t B, int64_t C, int64_t HW, int dtype, void* stream);

/* ---- FlowNet eval forward (models/base_networks.py:59-165, BASELINE configs[1]): the layers around the dense convolutions
 * once BatchNorm is folded into the conv weights (ffwm_amd/flownet_eval.py).  Contiguous float32.
 *
 * ffwm_bias_act_forward: y = act(h[B,C,HW] + bias[C]) (bias may be NULL); act 0 = none, 1 = LeakyReLU(negative_slope),
 * 2 = tanh.  Two optional destinations (either may be NULL, y may alias h): channel c of sample b is written at
 * y + b * y_batch_stride + c * HW, so a destination can be a channel slice of a wider concatenation buffer
 * (torch.cat of base_networks.py:133-151 without its copy kernel). */
int ffwm_bias_act_forward(const void* h, const void* bias, void* y, void* y2, int64_t B, int64_t C, int64_t HW,
                          int64_t y_batch_stride, int64_t y2_batch_stride, int act, double negative_slope, int dtype,
                          void* stream);
/* FlowNet's thin-channel full-resolution 3 x 3 layers (models/base_networks.py:64-112: conv0 6 -> 64, inter_conv0 18 -> 16) by a direct
 * kernel: a lane owns one pixel and 8 / 16 output channels, weights as scalar operands.  weight_ctk = the layer's weights re-arranged by
 * the host from Conv2d's [K][C][3][3] to [C][3][3][K]; K % 8 == 0.  Conv2d(C, K, 3, 1, 1) -> y[B,K,H,W] contiguous; act = 1:
 * LeakyReLU(negative_slope) after the bias (bias may be NULL).  fp32. */
int ffwm_conv_thin_forward(const void* x, const void* weight_ctk, const void* bias, void* y, int64_t B, int64_t C, int64_t H, int64_t W,
                           int64_t K, int act, double negative_slope, int dtype, void* stream);

/* predict_flow* (base_networks.py:45-49): y[B,2,H,W] = tanh(conv2d(x[B,C,H,W], weight[2,C,3,3], bias[2], stride 1, pad 1)) */
int ffwm_flow_head_forward(const void* x, const void* weight, const void* bias, void* y, int64_t B, int64_t C, int64_t H,
                           int64_t W, int dtype, void* stream);
/* upsampled_flow*_to_* (base_networks.py:104-109): out = conv_transpose2d(flow[B,2,H,W], weight[2,2,4,4], bias[2], stride 2,
 * pad 1) -> [B,2,2H,2W], sample b written at out + b * out_batch_stride (a channel slice of the concatenation buffer). */
int ffwm_flow_up_forward(const void* flow, const void* weight, const void* bias, void* out, int64_t B, int64_t H, int64_t W,
                         int64_t out_batch_stride, int dtype, void* stream);

/* Training: the backward of the two launch-lean layers above (models/base_networks.py:45-49,104-109).
 * ffwm_flow_head_backward: y = the head's output, grad_y [B, 2, H, W] contiguous; writes grad_z = grad_y * (1 - y^2) [B, 2, H, W] (the
 * row operand of the weight gradient, ffwm_conv2d_wgrad_tiled(grad_z, x, ...), whose fused row sum is the bias gradient) and
 * grad_x [B, C, H, W].  ffwm_flow_up_backward: grad_x [B, 2, H, W] from grad_out [B, 2, 2H, 2W] read with its batch stride (a
 * channel slice of the decoder concatenation's gradient needs no copy). */
int ffwm_flow_head_backward(const void* y, const void* grad_y, const void* weight, void* grad_z, void* grad_x, int64_t B, int64_t C,
                            int64_t H, int64_t W, int dtype, void* stream);
int ffwm_flow_up_backward(const void* grad_out, const void* weight, void* grad_x, int64_t B, int64_t H, int64_t W,
                          int64_t grad_out_batch_stride, int dtype, void* stream);

/* fp32 MFMA implicit-GEMM convolution forward, NCHW, for the layers the vendor library wraps in layout transposes:
 * nn.Conv2d(C, K, kernel 3 or 4, stride 1 or 2, pad) or -- transposed == 1 -- nn.ConvTranspose2d(C, K, 4, 2, 1)
 * (weight [C, K, 4, 4]) of models/base_networks.py:12-31,64-112; transposed == 2 / 3: the data gradient of Conv2d(3, 2, 1) /
 * Conv2d(3, 1, 1) with `input` = grad_output and the layer's own weight tensor.  output sample b starts at
 * output + b * out_batch_stride (a channel slice of a concatenation buffer is a valid destination); output2 (NULL: none) is a
 * second destination that receives the same values (the decoder's concatenation slice next to the skip tensor).  act: 0 none,
 * 1 LeakyReLU(negative_slope), 2 tanh, applied after the bias.  Exact fp32 (v_mfma_f32_32x32x2_f32).
 * workspace (ABI 5; NULL: never split): a layer with so few output pixels that its tiles do not fill the chip is cut along its
 * REDUCTION when `workspace_bytes >= ffwm_conv2d_forward_workspace(...)` (16-byte aligned, contents irrelevant, nothing to
 * zero-fill): every slice stores its partial sums into its own slot and a second launch adds the slots in slice order and
 * applies bias / activation -- no float atomics, so the output is bit-reproducible.  (ABI <= 4 added the slices atomically into
 * a caller-zeroed output and left bias / activation to ffwm_bias_act_forward.) */
int64_t ffwm_conv2d_forward_workspace(int64_t B, int64_t C, int64_t H, int64_t W, int64_t K, int kernel, int stride, int pad,
                                      int transposed);
int ffwm_conv2d_forward(const void* input, const void* weight, const void* bias, void* output, void* output2, int64_t B, int64_t C,
                        int64_t H, int64_t W, int64_t K, int kernel, int stride, int pad, int transposed, int64_t out_batch_stride,
                        int64_t out2_batch_stride, int act, double negative_slope, void* workspace, int64_t workspace_bytes,
                        int dtype, void* stream);

/* fp32 MFMA weight gradient of the convolutions ffwm_conv2d_forward serves, one launch, no layout transposes:
 *   grad_weight[k][(c, r, s)] += sum_{b, oy, ox} rows[b, k, oy, ox] * gathered[b, c, oy * stride + r - pad, ox * stride + s - pad]
 * nn.Conv2d(C, K, kernel, stride, pad):   rows = grad_output [B,K,Ho,Wo], gathered = input [B,C,H,W]         -> [K, C, k, k]
 * nn.ConvTranspose2d(Ci, Co, 4, 2, 1):    rows = input [B,Ci,H,W],        gathered = grad_output [B,Co,2H,2W] -> [Ci, Co, 4, 4]
 * (kernel = 4, stride = 2, pad = 1).  grad_weight must be ZERO-FILLED (or hold the value to accumulate into): the pixel
 * slices of the reduction are added atomically. */
int ffwm_conv2d_wgrad(const void* rows, const void* gathered, void* grad_weight, int64_t B, int64_t K, int64_t Ho, int64_t Wo,
                      int64_t C, int64_t H, int64_t W, int kernel, int stride, int pad, int dtype, void* stream);

/* The same weight gradient on the tiled kernel (128 x 128 tiles of grad_weight, pixels linearised over the batch, one barrier per
 * 32 pixels; csrc/conv_bwd.hip "tiled variant"): grad_weight is OVERWRITTEN -- the library zero-fills it itself when the pixel
 * range is cut into slices that meet by atomics -- and grad_bias[K] (NULL: not wanted; Conv2d only: the sum of `rows` over batch
 * and pixels = convolution_backward's grad_bias) comes out of the same pass.  Needs Ho * Wo % 4 == 0 and a 16-byte aligned `rows`
 * (status FFWM_ERR_ARG otherwise: use ffwm_conv2d_wgrad).  prezeroed != 0 (ABI 5): grad_weight / grad_bias arrive ZEROED (slices of a gradient
 * arena cleared by one launch per step) and a sliced launch skips its own zero-fill -- an argument since ABI 5, the process-global
 * option of ABI 4 was not thread-safe. */
int ffwm_conv2d_wgrad_tiled(const void* rows, const void* gathered, void* grad_weight, void* grad_bias, int64_t B, int64_t K,
                            int64_t Ho, int64_t Wo, int64_t C, int64_t H, int64_t W, int kernel, int stride, int pad, int prezeroed,
                            int dtype, void* stream);

/* 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2, 3x3) on the fp32 MFMA units (csrc/conv_winograd.hip): the
 * forward (data_gradient = 0: weight [K, C, 3, 3]) or the data gradient (data_gradient = 1: input = grad_output with C =
 * the layer's OUTPUT channels, K = its input channels, weight = the layer's own [C, K, 3, 3]) of Conv2d(.., 3, 1, 1)
 * (models/base_networks.py:207-233: the residual blocks of netG; :59-112 FlowNet's conv*_1 / inter_conv*).
 * output [B, K, H, W] = conv + bias (NULL: none), then LeakyReLU(slope) when act = 1 (slope 0: ReLU).  workspace: device
 * memory of ffwm_conv3x3_winograd_workspace_bytes(K, C) bytes (the transformed weights; rewritten by every call).
 * data_gradient + 2: the workspace still holds the transformed weights of an earlier call with the same weight values,
 * direction, K, C and (W % 4 == 0) -- frozen networks (VGG19, LightCNN: models/losses.py:398-519) skip the transform. */
int64_t ffwm_conv3x3_winograd_workspace_bytes(int64_t K, int64_t C);
/* In how many pieces ffwm_conv3x3_winograd_forward will cut the reduction (input channels) of this call: 1, 2 or 4.  A call whose
 * (64 tiles, 64 output channels) pairs fill less than half the CUs (256 -> 256 at 32 x 32, batch 8) is cut so that one round of
 * persistent workgroups covers the chip; the pieces' partial outputs meet by atomics in the output, which the library zero-fills
 * itself.  Only for act = 0.  The caller's routing policy (ffwm_amd/conv.py) uses it to price a call. */
int ffwm_conv3x3_winograd_splits(int64_t B, int64_t C, int64_t H, int64_t W, int64_t K, int act);

/* The weight transforms of several Winograd calls in ONE launch.  Item i fills `workspace` (ffwm_conv3x3_winograd_workspace_bytes(K, C)
 * bytes) exactly as ffwm_conv3x3_winograd_forward(.., K, C, data_gradient, ..) would for an input whose width is (or is not) a
 * multiple of 4; that call is then made with data_gradient + 2 (reuse).  K / C in the CALL's terms: for the data gradient K = the
 * layer's input channels, C = its output channels, weight = the layer's own [C, K, 3, 3].  (ffwm_amd/spectral_norm.py: all layers of
 * a spectrally normalised network right after its batched normalisation.) */
typedef struct ffwm_wino_weights {
    const void* weight;
    void* workspace;
    int64_t K, C;
    int data_gradient;            /* 0 = forward, 1 = data gradient */
    int width_multiple_of_4;      /* of the planes the transform will be used on (selects the thin-tail split of K) */
} ffwm_wino_weights;
int ffwm_conv3x3_winograd_weights_multi(const ffwm_wino_weights* items, int n, int dtype, void* stream);
int ffwm_conv3x3_winograd_forward(const void* input, const void* weight, const void* bias, void* output, void* workspace,
                                  int64_t B, int64_t C, int64_t H, int64_t W, int64_t K, int data_gradient, int act,
                                  double slope, int dtype, void* stream);

/* One Adam step (no weight decay, no amsgrad: torch.optim.Adam as models/ffwm_model.py:46-49 and
 * models/flownet_model.py:33 construct it) over FLAT float32 arrays of n elements, 16-byte aligned: parameters,
 * gradients, first and second moments.  `step` is the 1-based step count (bias corrections 1 - beta^step). */
int ffwm_adam_step(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, int64_t n, double lr,
                   double beta1, double beta2, double eps, int64_t step, int dtype, void* stream);

/* The same step with the step counter in DEVICE memory (a step inside a captured hipGraph: a replay runs no host code).  state:
 * FOUR doubles (ABI 3; three before): state[0] = steps taken so far (start it at 0), state[1] and state[2] are scratch,
 * state[3] = learning-rate override -- when >= 0 it replaces `lr`, so that a learning-rate schedule reaches a captured step
 * (`lr` is baked into the graph as a kernel argument; the host writes state[3] between replays); start it at a NEGATIVE value for
 * "no override" (ABI 4; ABI 3 took 0 for "none", so a schedule that reached 0.0 could not freeze the weights).  Two launches. */
int ffwm_adam_step_device(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, int64_t n, double lr, double beta1,
                          double beta2, double eps, void* state, int dtype, void* stream);

/* ---- built-in per-kernel timing (HIP events on the launch stream) ---------------------------
 * ffwm_prof_enable(1) brackets every kernel launch of this library with a pair of HIP events
 * recorded on the stream the kernel is launched on.  ffwm_prof_collect() waits for the recorded
 * events, folds them into per-kernel totals and returns the number of distinct kernels seen;
 * ffwm_prof_get() reads one row.  ffwm_prof_reset() drops everything. */
int ffwm_prof_enable(int on);
int ffwm_prof_collect(void);
int ffwm_prof_get(int row, char* name, int name_len, int64_t* launches, double* total_ms,
                  double* algorithmic_bytes);
/* total algorithmic flops of row `row` (non-zero for the MFMA kernels: conv3x3_wgrad, correlation_colmax) */
int ffwm_prof_get_flops(int row, double* algorithmic_flops);
/* Sum over the row's launches of max(algorithmic bytes / 8 TB/s, algorithmic flops / 157.3 TFLOP/s), in ms: the time the BINDING
 * roofline of each launch allows.  A scope that serves many shapes (a weight gradient from 128 x 128 planes down to 2 x 2) is
 * MFMA-bound on some launches and weight-streaming on others; total_ms / this = how far the scope is from its own rooflines. */
int ffwm_prof_get_bound(int row, double* roofline_ms);
int ffwm_prof_reset(void);

/* Tuning/ablation switches (bench and tests only): returns the previous value, or
 * FFWM_ERR_ARG for an unknown key.  Keys: "be_fwd_variant", "be_bwd_variant",
 * "channel_slab", "xcd_remap", "ablate", "rows_per_thread",
 * "scatter_variant". */
int ffwm_set_option(const char* key, int value);

/* Zero-fill `bytes` bytes (a multiple of 4) at the 4-byte aligned device address `p` with a KERNEL on `stream` -- what the
 * caller-zero-fills contract of the backward entry points (external_function.py:49-50,96,137-138) needs under hipGraph capture, where
 * hipMemsetAsync becomes a memset node: on ROCm 7.0 such nodes were executed with a corrupted fill pattern when several graphs with
 * side-stream branches were replayed back to back (profiles/r05_wgrad_nan_root_cause.txt). */
int ffwm_zero_fill(void* p, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FFWM_HIP_H_ */
