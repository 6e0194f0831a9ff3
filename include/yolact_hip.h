/* yolact_hip.h — C-ABI of libyolact_hip.so, the MI355X (gfx950) hot path of YOLACT.
 *
 * The reference (feiyuhuahuo/Yolact_minimal) has NO native interface for this path: every tensor
 * op goes through torch.nn / ATen, and its only native file is the Cython greedy NMS
 * (cython_nms.pyx:24-74).  So the entry points below are what a maintainer would bind *instead of*
 * the ATen calls at the cited reference lines.  Conventions:
 *   - extern "C", raw DEVICE pointers + explicit sizes + a hipStream_t (passed as void*);
 *   - fp32 everywhere, activations NHWC ([B][H][W][C], C contiguous), weights [Cout][KH][KW][Cin];
 *   - every call is asynchronous on `stream`; nothing allocates, frees or synchronises;
 *   - scratch memory is supplied by the caller (ask ym_*_workspace_bytes first);
 *   - return 0 on success, a negative YM_E* code otherwise; ym_last_error() gives the text.
 * No torch types appear here.  The Python host (yolact_minimal_amd/hip.py) binds these with ctypes.
 */
#ifndef YOLACT_HIP_H
#define YOLACT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YM_OK 0
#define YM_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define YM_ENOSPC (-2)   /* workspace too small */
#define YM_ELAUNCH (-3)  /* hipGetLastError() after a launch was not hipSuccess */

#define YM_ACT_NONE 0
#define YM_ACT_RELU 1
#define YM_ACT_TANH 2
#define YM_ACT_GELU 3   /* exact erf GELU (Swin Mlp, modules/swin_transformer.py:88) — forward only */

typedef void* ym_stream_t; /* hipStream_t */

int ym_abi_version(void);
const char* ym_last_error(void);

/* ---- layout / parameter preparation ----------------------------------------------------------- */

/* NCHW fp32 image -> NHWC with C padded to 4 (pad lanes = 0).  Replaces the implicit layout of
 * `self.backbone(img)` input, reference modules/yolact.py:142. in:[B][C][H][W] (C<=4) out:[B][H][W][4] */
int ym_nchw_to_nhwc4(const float* in, float* out, int B, int C, int H, int W, ym_stream_t s);

/* OIHW fp32 weight (torch state-dict layout) -> [Cout][KH][KW][cin_pad] with zero padding of the
 * channel dim, rows zero-extended to k_pad floats (k_pad >= KH*KW*cin_pad, multiple of 32). */
int ym_pack_conv_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW,
                        int cin_pad, int k_pad, ym_stream_t s);

/* Eval-mode BatchNorm folded to y = x*scale + shift (reference modules/resnet.py:24,28,32 in eval):
 * scale = gamma / sqrt(var + eps), shift = beta - mean*scale. */
int ym_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
               float* scale, float* shift, int C, ym_stream_t s);

/* ---- fused convolution (implicit GEMM on the f32 MFMA pipe) ------------------------------------- */

typedef struct {
    int n_begin, n_end;      /* output-channel range [n_begin, n_end) routed to this segment      */
    float* out;              /* element (b, pixel, n) lives at out[b*batch_stride + pixel*pitch + (n-n_begin)] */
    int64_t batch_stride;    /* floats */
    int32_t pitch;           /* floats between consecutive output pixels */
    int32_t act;             /* YM_ACT_* applied to this segment */
} ym_conv_seg;

typedef struct {
    const float* in;         /* NHWC [B][H][W][Cin]; for Cin==4 ("stem" mode) the packed weight uses cin_pad=4 */
    const float* weight;     /* packed by ym_pack_conv_weight: [Cout][k_pad] */
    const float* scale;      /* [Cout] or NULL (=1) : folded BN scale */
    const float* shift;      /* [Cout] or NULL (=0) : folded BN shift or conv bias */
    const float* residual;   /* NULL or NHWC [B][Ho][Wo][Cout] added before the activation */
    int32_t B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, k_pad;
    int32_t nseg;            /* 1..3 */
    ym_conv_seg seg[3];
    /* tuning knobs (0 = library heuristic); yolact_minimal_amd/tuned_gfx950.json holds measured choices */
    int32_t tile_m, tile_n;  /* workgroup kernel: 128/64 (workgroup tile); wave kernel: 64/32 (per-wave tile) */
    int32_t ksplit;          /* workgroup kernel: K slices across the grid (slices > 1 need workspace) */
    int32_t kwaves;          /* 0 = LDS-tiled workgroup kernel; 1/2/4/8 = wave-private kernel with that many
                                waves of one workgroup splitting K for each output tile (tile_m x tile_n = the WAVE's tile, 32 / 64).
                                With stages = 22 / 23 / 24: the DMA-ring variant (kwaves 1 / 2 / 4, tile 32x32, 64x32 or 32x64, Cin % 32
                                == 0): every wave streams its operands through a private LDS ring of 2 / 3 / 4 K tiles filled by
                                global->LDS DMA; with tile 32x32, kwaves 4 it also takes tail_tiles / tail_ksplit (<= 8 slices) */
    int32_t transposed;      /* 0 = convolution.  1 = DATA GRADIENT of that convolution (conv-transpose gather):
                                `in` is dy NHWC [B][H][W][Cin] (H,W = the forward OUTPUT size, Cin = forward Cout
                                padded to a multiple of 32), the result is dx [B][Ho][Wo][Cout] (Ho,Wo = forward
                                INPUT size, Cout = forward Cin), stride/pad/KH/KW are the forward conv's, and
                                `weight` comes from ym_pack_conv_weight_dgrad.  Autograd counterpart of every
                                nn.Conv2d on the path (loss_total.backward(), reference train.py:126). */
    int32_t stages;          /* workgroup kernel operand staging: 0/2 = registers, double buffer; 3 = registers, loads two K tiles
                                ahead (64-wide tiles); 22/23/24 = direct global->LDS DMA, ring of 2/3/4 (24: 64x64 tile only);
                                42/43/44/46/48 = PERSISTENT direct-to-LDS kernel, ring of 2/3/4/6/8 (64x64 tile, plain NHWC
                                output, ReLU or no activation, no bn_sum; anything else falls back to 22/23/24): grid_wgs workgroups walk
                                the (tile, K slice) items and the operand stream runs on across item boundaries */
    double* bn_sum;          /* optional [Cout] fp64 accumulators (zeroed by the caller): the epilogue adds the */
    double* bn_sumsq;        /* per-channel sum / sum of squares of the conv OUTPUT (train-mode BN statistics).  */
                             /* Only when ym_conv2d_fuses_bn_stats(desc) == 1 (plain NHWC output).               */
    int32_t* tile_counters;  /* optional, >= ym_conv2d_tile_counters(desc) int32, ZERO before the first launch (the  */
                             /* kernel leaves them zero): with a K split and a plain NHWC output, the LAST workgroup  */
                             /* to finish an output tile sums the slices (in slice order: deterministic) and runs the  */
                             /* epilogue in the same launch; NULL = separate reduce launch.  One buffer may be shared  */
                             /* by launches that are ordered on one stream, never by concurrent ones.                  */
    int32_t nlevels;         /* 0 = one image size.  1..5 = "pyramid" input: `in` holds nlevels feature maps of DIFFERENT sizes back   */
    int32_t level_h[5];      /* to back, level-major ([level][B][h][w][Cin]); the same filter runs over all of them in ONE launch   */
    int32_t level_w[5];      /* (the shared PredictionModule of modules/yolact.py:149-157 is one conv per level in the reference). */
                             /* Stride 1, pad = K/2 only; H/W/Ho/Wo of the descriptor are ignored, M = sum_l B*h_l*w_l.  A plain   */
                             /* output (one segment, batch_stride 0) keeps the input's row order; a segmented output places level l's pixel p of image b at        */
                             /* out[b*batch_stride + (sum_{j<l} h_j*w_j + p)*pitch] = the reference's cat over levels (:155-157).     */
    int32_t tail_tiles;      /* workgroup-quantisation fix (needs tile_counters, ksplit <= 1, plain NHWC output): the   */
    int32_t tail_ksplit;     /* LAST tail_tiles output tiles are each split into tail_ksplit K slices, so that e.g. 580  */
                             /* tiles on 256 CUs become 512 whole tiles + 68x4 quarter tiles instead of 2-or-3 per CU.   */
    int32_t mma;             /* matrix pipe of the workgroup kernel.  0 = f32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, the    */
                             /* parity mode).  3 / 6 = "split bf16": every fp32 operand is split on its way into LDS into two / three  */
                             /* bf16 terms and the 3 / 6 most significant cross products run on v_mfma_f32_32x32x16_bf16 with fp32     */
                             /* accumulation (relative error per product ~2^-17 / ~2^-23; tensors in HBM stay fp32).  Needs Cin % 32   */
                             /* == 0, kwaves == 0, no pyramid input; stages is ignored (register-staged double buffer).               */
    int32_t bnb_relu;        /* Train-mode BatchNorm BACKWARD statistics, fused.  A data-gradient launch (transposed = 1) writes the   */
    const float* bnb_y;      /* gradient dout[M][Cout] of the PREVIOUS layer's BN output; with bnb_y != NULL its epilogue also adds    */
    const float* bnb_out;    /* that BN's two backward sums to bn_sum / bn_sumsq (zeroed by the caller): bn_sum[c] += sum_m dz,        */
    const float* bnb_mean;   /* bn_sumsq[c] += sum_m dz * xhat, where xhat = (y - mean) * invstd, y = bnb_y = that layer's raw conv    */
    const float* bnb_invstd; /* output [M][Cout], and dz = dout masked by the layer's ReLU when bnb_relu: out > 0 with                 */
    const float* bnb_gamma;  /* bnb_out = its saved output, or (bnb_out == NULL) xhat * gamma + beta > 0 re-derived exactly as the     */
    const float* bnb_beta;   /* forward pass computed it.  Same sums as the first pass of ym_bn_train_bwd, which                       */
                             /* ym_bn_train_bwd_apply then skips.  Needs ym_conv2d_fuses_bn_stats(desc) == 1.                          */
    int32_t grid_wgs;        /* persistent kernel (stages 4x): workgroups to launch; 0 = as many as the CUs hold (LDS-limited, at most  */
                             /* 4 per CU), never more than there are work items.  Wave kernel with DMA rings (kwaves > 0, stages 22-24):  */
                             /* waves per workgroup, 0 = 4 (1 / 2 with kwaves <= that: single tiles are balanced over the CUs)            */
} ym_conv_desc;

/* y = act(conv(x, w) * scale + shift + residual), one launch (plus a reduce launch if K is split).
 * Replaces: conv+BN+ReLU(+add) of Bottleneck.forward (modules/resnet.py:20-40), the stem
 * (modules/resnet.py:88-90), conv+bias+ReLU of FPN / ProtoNet (modules/yolact.py:62-68,37-47) and,
 * with three segments, the bbox/conf/coef convs + tanh + permute/reshape/cat of
 * PredictionModule.forward + Yolact.forward (modules/yolact.py:27-30,155-157). */
size_t ym_sizeof_conv_desc(void);                      /* for bindings: must equal their mirror of ym_conv_desc */
size_t ym_conv2d_workspace_bytes(const ym_conv_desc* d);
int ym_conv2d_tile_counters(const ym_conv_desc* d);   /* output tiles of the chosen plan (0 if K is not split) */
int ym_conv2d_fuses_bn_stats(const ym_conv_desc* d);
int ym_conv2d_fwd(const ym_conv_desc* d, void* workspace, size_t workspace_bytes, ym_stream_t s);

/* ---- Swin-T backward + AdamW (row a18 under loss.backward(); reference train.py:62-63,126) ---------------------------
 * LayerNorm backward over the last dim of x [M][C] (statistics recomputed): dx, dgamma, dbeta (overwritten). */
size_t ym_layernorm_bwd_workspace_bytes(int C);
int ym_layernorm_bwd(const float* dy, const float* x, const float* gamma, float eps, int64_t M, int C, float* dx, float* dgamma,
                     float* dbeta, void* workspace, size_t workspace_bytes, ym_stream_t s);
/* Backward of ym_patch_merge_layernorm: dy [B*ceil(H/2)*ceil(W/2)][4C], x NHWC [B][H][W][C] -> dx (same shape as x; every
 * element written), dgamma / dbeta [4C].  workspace >= ym_layernorm_bwd_workspace_bytes(4*C). */
int ym_patch_merge_layernorm_bwd(const float* dy, const float* x, int B, int H, int W, int C, const float* gamma, float eps,
                                 float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, ym_stream_t s);
/* DropPath + residual add in one pass (modules/swin_transformer.py:71-82 with the call sites :285,288): out = res + (y / keep) *
 * floor(keep + rnd[b]) in the reference's operation order; rnd [B] is the raw torch.rand draw, res / y / out are [B][per_sample]
 * (per_sample % 4 == 0).  Backward: the residual's gradient is dout, dy = (dout * mask) / keep. */
int ym_drop_path_add(const float* res, const float* y, const float* rnd, float keep, float* out, int B, int64_t per_sample,
                     ym_stream_t s);
int ym_drop_path_bwd(const float* dout, const float* rnd, float keep, float* dy, int B, int64_t per_sample, ym_stream_t s);
/* Exact (erf) GELU of Mlp.forward (modules/swin_transformer.py:92-96) on a saved pre-activation; n % 4 == 0. */
int ym_gelu_fwd(const float* z, float* out, int64_t n, ym_stream_t s);
int ym_gelu_bwd(const float* dy, const float* z, float* dz, int64_t n, ym_stream_t s);
/* Backward of ym_swin_window_attention: dout [B*H*W][C] -> dqkv [B*H*W][3C] (every element written), and ACCUMULATED into
 * caller-zeroed buffers: dqkv_bias_pad [3C] (gradient reaching the qkv bias through the padded tokens, to be added to the
 * qkv Linear's bias gradient) and dtable [(2*window-1)^2][heads] (relative-position-bias table). */
int ym_swin_window_attention_bwd(const float* qkv, const float* qkv_bias, const float* rel_bias_table, const float* dout, int B,
                                 int H, int W, int C, int heads, int window, int shift, float* dqkv, float* dqkv_bias_pad,
                                 float* dtable, ym_stream_t s);
/* torch.optim.AdamW (amsgrad=False) on flat fp32 buffers, step >= 1 (reference train.py:63: lr, weight_decay=0.05). */
int ym_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, ym_stream_t s);

/* ---- train_aug pixel work (next row f4; reference utils/augmentations.py:60-77,138-216,230-252) ---------------------------
 * The random decisions of one sample (drawn on the host in the reference's `random` call order):
 *   source H x W -> mirror -> crop (cx, cy, cw, ch) -> pad to the q x q square at (px, py), border = norm_mean -> bilinear resize
 *   to r x r -> final_mode 0: r == S | 1: paste at (fx, fy) into S x S (border = norm_mean) | 2: crop S x S at (fx, fy).
 * photometric: optional brightness add / contrast scale (each clipped to 0..255), then BGR->HSV, S *= saturation, H += hue
 * (wrapped to 0..360), HSV->BGR, clip.  mean / std in BGR order (config.py:66-67). */
typedef struct {
    int32_t H, W, mirror, cx, cy, cw, ch, q, px, py, r, S, final_mode, fx, fy, has_brightness, has_contrast;
    float brightness, contrast, saturation, hue, mean[3], std[3];
} ym_aug_plan;
/* img HWC BGR (uint8 if is_u8 else float32) -> out [3][S][S] normalised RGB planes, one launch. */
int ym_train_aug_image(const void* img_hwc_bgr, int is_u8, const ym_aug_plan* plan, float* out_chw, ym_stream_t s);
/* masks [n][H][W] (uint8 / float32), keep [k] indices of the surviving instances -> out [k][S][S] (bilinear, border 0).  uint8 masks
 * of an already-square crop (plan->cw == plan->ch) take OpenCV's 8-bit fixed-point resize, as in the reference (its masks stay uint8
 * there: utils/augmentations.py:138-141,180-181): a {0,1} mask comes out {0,1}. */
int ym_train_aug_masks(const void* masks, int is_u8, const int32_t* keep, int k, const ym_aug_plan* plan, float* out,
                       ym_stream_t s);

/* ---- evaluation inner products right after after_nms (next row f2) -----------------------------------------------
 * mask_iou (utils/box_utils.py:189-200): masks_a [n][P], masks_b [g][P] fp32 in {0,1} (P = img_h*img_w < 2^24) ->
 * iou [n][g] = inter / ((area_a + area_b) - inter), the reference's fp32 matmul evaluated as popcounts of bit rows
 * (exact, so bit-identical; 0/0 -> NaN).  Every mask is read from HBM once. */
size_t ym_mask_iou_workspace_bytes(int n, int g, int64_t P);
int ym_mask_iou(const float* masks_a, int n, const float* masks_b, int g, int64_t P, float* iou, void* workspace,
                size_t workspace_bytes, ym_stream_t s);
/* box_iou (utils/box_utils.py:8-37) of corner boxes [n][4] x [g][4] -> [n][g]. */
int ym_box_iou(const float* boxes_a, int n, const float* boxes_b, int g, float* iou, ym_stream_t s);
/* The matching loop of prep_metrics (utils/common_utils.py:186-216) for both IoU types and every threshold at once:
 * matched[type][t][i] = 1 iff prediction i (visited in order, class pred_cls[i]) found an unused gt of its class with
 * IoU > thresholds[t] (largest such IoU; double comparison like the python floats).  g <= 512. */
int ym_match_detections(const float* iou_box, const float* iou_mask, const int32_t* pred_cls, const int32_t* gt_cls, int n,
                        int g, const double* thresholds, int T, int num_classes, uint8_t* matched, ym_stream_t s);

/* COCO RLE of n binary masks [n][H][W] (fp32, nonzero = foreground), next row f3: what pycocotools.mask.encode(
 * np.asfortranarray(mask)) + .decode('ascii') yield in MakeJson.add_mask (utils/common_utils.py:88-96).  Per mask i:
 * counts[i*cap_runs .. +nruns[i]) = column-major run lengths starting with the zeros run, and the compressed ASCII string
 * str[i*cap_str .. +str_len[i]).  If a mask needs more than cap_runs runs (or cap_str bytes) nruns[i] still holds the
 * number of runs and str_len[i] = -1: call again with larger buffers.  W <= 4096; workspace >= n*cap_runs*4 bytes. */
int ym_rle_encode(const float* masks, int n, int H, int W, uint32_t* counts, int cap_runs, int32_t* nruns, uint8_t* str,
                  int cap_str, int32_t* str_len, void* workspace, size_t workspace_bytes, ym_stream_t s);

/* COCO annotation -> dense instance masks, next row f4 (the reader): what `self.coco.annToMask(aa)` yields per annotation in
 * COCODetection.__getitem__ (utils/coco.py:96) = pycocotools annToRLE (frPyObjects over the polygon list, merge = union) +
 * decode, i.e. cocoapi maskApi.c rleFrPoly / rleMerge / rleDecode.
 * ym_poly_to_mask: xy = every polygon's vertices (x0, y0, x1, y1, ... as in the JSON, float64) back to back; polygon p owns the
 *   vertices [poly_off[p], poly_off[p+1]) (counted in (x, y) pairs); annotation a owns the polygons [ann_off[a], ann_off[a+1]).
 * ym_runs_to_mask: the uncompressed-RLE form ({'counts': [...]}): annotation a owns counts[run_off[a] .. run_off[a+1]).
 * masks: [n][H][W] uint8 {0, 1}, 4-byte aligned.  W <= 4096.  Masks whose two bitmaps (see ym_ann_to_mask_workspace_bytes) do
 * not fit the LDS need that much workspace; otherwise workspace may be NULL. */
size_t ym_ann_to_mask_workspace_bytes(int n, int H, int W);
int ym_poly_to_mask(const double* xy, const int32_t* poly_off, const int32_t* ann_off, int n, int H, int W, uint8_t* masks,
                    void* workspace, size_t workspace_bytes, ym_stream_t s);
int ym_runs_to_mask(const uint32_t* counts, const int32_t* run_off, int n, int H, int W, uint8_t* masks, void* workspace,
                    size_t workspace_bytes, ym_stream_t s);

/* ---- training: weight gradient, batch-norm with batch statistics, small backward ops, SGD -------------------
 * These replace what autograd + ATen/cuDNN execute for `loss_total.backward()` / `optimizer.step()`
 * (reference train.py:124-130) and nn.BatchNorm2d in train mode (modules/resnet.py:10-14,46). */

/* OIHW weight -> [Cin][KH][KW][cout_pad] for the transposed gather above (cout_pad % 32 == 0, zero padded). */
int ym_pack_conv_weight_dgrad(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, int cout_pad,
                              ym_stream_t s);

/* All the weight packings of a training step in one launch (after the optimizer step every layer's forward and dgrad image is
 * stale at once).  items_dev: DEVICE array, one entry per destination image, ordered by first_chunk:
 *   kind 0: ym_pack_conv_weight   into dst [rows][pad_b]          (pad_a = cin_pad, pad_b = k_pad, rows >= cout zero padded)
 *   kind 1: ym_pack_conv_weight_dgrad into dst [cin][kh][kw][pad_a] (pad_a = cout_pad)
 * Chunks of an item: kind 0: ceil(rows * pad_b / 1024); kind 1: ceil(cin*kh*kw / 32) * (pad_a / 32) (32 x 32 transpose tiles).
 * first_chunk = running sum of the chunks of the preceding items; total_chunks = the sum over all items. */
typedef struct {
    const float* src;        /* OIHW weight */
    float* dst;
    int32_t cout, cin, kh, kw, pad_a, pad_b, rows, kind;
    uint32_t first_chunk;
    uint32_t reserved;
} ym_pack_item;
int ym_pack_conv_weights_batch(const ym_pack_item* items_dev, int n_items, int total_chunks, ym_stream_t s);

typedef struct {
    const float* x;          /* forward input NHWC [B][H][W][Cin] (Cin = padded channel pitch; 4 for the stem) */
    const float* dy;         /* output gradient [B][Ho][Wo][Cout] (Cout = channel pitch, multiple of 4) */
    float* dw;               /* OIHW [Cout_real][Cin_real][KH][KW], overwritten */
    int32_t B, H, W, Cin, Cin_real, Cout, Cout_real, KH, KW, stride, pad, Ho, Wo;
    int32_t msplit;          /* 0 = heuristic; pixel-range slices across the grid */
    int32_t accumulate;      /* 1: dw += gradient (a weight shared by several convs of one step, e.g. the prediction head over the */
                             /* five FPN levels, modules/yolact.py:149-153); 0: overwrite                                         */
    int32_t row_end[2];      /* output-channel ranges routed to separate OIHW tensors (the head's conf | bbox | coef convs run as   */
    float* dw_seg[2];        /* ONE 351-channel conv): rows [0,row_end[0]) -> dw, [row_end[0],row_end[1]) -> dw_seg[0],            */
                             /* [row_end[1],Cout_real) -> dw_seg[1].  row_end[0] = 0 means a single tensor (dw).                 */
    int32_t lds_buffers;     /* tuning knob: 0 / 2 = double-buffered pixel steps (2 workgroups per CU), 1 = single buffer (3 per CU);   */
                             /* 22 / 23 / 24 = operands DMA'd global -> LDS (no staging registers): 32 pixels x ring of 2, 16 pixels x  */
                             /* ring of 3 / 4 (more than 64 output channels only; Cin % 32 != 0 runs as 2).  Same products in the same  */
                             /* order: for a given msplit every variant returns the same bits.                                           */
} ym_wgrad_desc;
size_t ym_conv2d_wgrad_workspace_bytes(const ym_wgrad_desc* d);
int ym_conv2d_wgrad(const ym_wgrad_desc* d, void* workspace, size_t workspace_bytes, ym_stream_t s);
/* The two halves of ym_conv2d_wgrad apart, so that the slab reductions of MANY layers run as one launch (a res101 step has 104
 * backbone weight gradients; reference: one `at::conv_backward` weight kernel per layer under `loss.backward()`, train.py:125).
 * ym_conv2d_wgrad_slabs: first pass only -- the msplit partial gradients stay in `workspace` (which must then live, unshared, until
 * the batched reduction has run) and *item describes the pending reduction (d->row_end[0] must be 0: one destination tensor).
 * ym_wgrad_reduce_batch: `items_dev` = n_items such records in DEVICE memory, `first_block` of each set by the caller to the sum of
 * `blocks` of the records before it, total_blocks = the sum over all; sums every item's slabs in slab order (the same bits as
 * ym_conv2d_wgrad) into its OIHW tensor.  Items of one batch must not share a destination. */
typedef struct {
    const float* slabs;      /* = workspace of the slab pass */
    float* dw;               /* OIHW destination (d->dw) */
    uint32_t first_block;    /* IN: set by the caller when it builds the table */
    uint32_t blocks;         /* OUT: 256-thread blocks this item occupies in the batched launch */
    uint32_t plan[14];       /* OUT: opaque (sizes, slab stride, division constants) */
} ym_wgrad_reduce_item;
int ym_conv2d_wgrad_slabs(const ym_wgrad_desc* d, void* workspace, size_t workspace_bytes, ym_wgrad_reduce_item* item, ym_stream_t s);
int ym_wgrad_reduce_batch(const ym_wgrad_reduce_item* items_dev, int n_items, uint32_t total_blocks, ym_stream_t s);

/* BatchNorm2d forward in TRAIN mode on a conv output y [M][C] (C % 4 == 0): batch mean / biased variance
 * (fp64 accumulation), running stats updated in place with `momentum` (unbiased variance, torch semantics),
 * out = relu?( (y-mean)*invstd*gamma + beta + residual? ).  save_mean/save_invstd [C] feed the backward.
 * workspace >= 16*C bytes. */
int ym_bn_train_fwd(const float* y, int64_t M, int C, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, const float* residual, int relu, float* out,
                    float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, ym_stream_t s);
/* Same, but the first 16*C bytes of `workspace` already hold the fp64 sum[C] | sumsq[C] of y (accumulated by
 * ym_conv2d_fwd through ym_conv_desc.bn_sum / bn_sumsq): skips the statistics pass over y. */
int ym_bn_train_fwd_stats(const float* y, int64_t M, int C, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, const float* residual, int relu,
                          float* out, float* save_mean, float* save_invstd, const void* stats, ym_stream_t s);

/* Backward of the above: dz = dout * (out > 0 if relu); dres (optional) = dz; dgamma/dbeta [C];
 * out may be NULL with relu = 1 when the forward had NO residual and `beta` is given: the mask is then re-derived from y with the
 * forward's exact affine (one pass less over HBM); beta is otherwise unused and may be NULL.
 * dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)).  workspace >= 16*C bytes; with
 * ym_bn_train_bwd_workspace_bytes(M, C) the column sums use per-workgroup partials (larger grid, no atomics, ordered sum). */
size_t ym_bn_train_bwd_workspace_bytes(int64_t M, int C);
int ym_bn_train_bwd(const float* dout, const float* out, const float* y, int64_t M, int C, const float* gamma,
                    const float* beta, const float* save_mean, const float* save_invstd, int relu, float* dy, float* dres,
                    float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, ym_stream_t s);

/* Second pass of ym_bn_train_bwd alone: `stats` = fp64 sum_m dz [C] | sum_m dz*xhat [C], already accumulated by the epilogue of the
 * data-gradient conv that produced `dout` (ym_conv_desc.bnb_*).  Every other argument as in ym_bn_train_bwd. */
int ym_bn_train_bwd_apply(const float* dout, const float* out, const float* y, int64_t M, int C, const float* gamma,
                          const float* beta, const float* save_mean, const float* save_invstd, int relu, float* dy, float* dres,
                          float* dgamma, float* dbeta, const void* stats, ym_stream_t s);

/* Gradient of the fused prediction-head output w.r.t. the 351(+pad)-channel conv output, for all FPN levels in one launch:
 * the loss hands back dclass [B][N][nc], dbox [B][N][4], dcoef [B][N][cd] (N = anchors of all levels, anchor = (pixel, a) level by
 * level); row r of dz (level-major "pyramid" order: level l owns rows lev_row[l] .. lev_row[l+1), each B*hw_l rows = (image, pixel))
 * gets [na*nc | na*4 | na*cd * (1 - coef^2) | 0-pad] = the reference's permute/reshape/cat/tanh backward (modules/yolact.py:27-30,
 * 155-157).  lev_row / lev_anchor: int32[nlev+1] on the HOST (row offsets, anchor offsets of the levels).  g_scale: optional device
 * float[3] multiplying the class / box / coef parts (the upstream gradients of the loss terms), NULL = 1. */
int ym_head_grad_gather(const float* dclass, const float* dbox, const float* dcoef, const float* coef, int B, int N, int nc, int cd,
                        int na, int nlev, const int32_t* lev_row, const int32_t* lev_anchor, int pitch, const float* g_scale,
                        float* dz, ym_stream_t s);
/* dst_i[0..n_i) = src[off_i .. off_i + n_i) for up to three destinations (or += with accumulate): splits a concatenated bias /
 * statistics vector back into the parameters' own gradient slots in one launch. */
int ym_scatter3(const float* src, float* d0, int n0, float* d1, int n1, float* d2, int n2, int accumulate, ym_stream_t s);

/* Backward of a fused conv epilogue `y = act(conv + bias)`: dz = dy * act'(y) (dz may be NULL or == dy for
 * YM_ACT_NONE), dbias[C] = column sums of dz (optional).  workspace >= 8*C bytes when dbias != NULL; with
 * >= 16*C + min(1024, ceil(M/32..)) * 16*C bytes (ym_bn_train_bwd_workspace_bytes(M, C) always suffices) the sums use per-workgroup
 * partials + an ordered finish instead of fp64 atomics. */
int ym_act_bias_bwd(const float* dy, const float* y, int64_t M, int C, int act, float* dz, float* dbias, void* workspace,
                    size_t workspace_bytes, ym_stream_t s);

/* lincomb_mask_loss (modules/yolact.py:241-291) for ONE image, forward and backward in one pass on the f32 MFMA:
 *   loss += sum_p (wscale/area_p) * sum_pix BCE(crop_p(sigmoid(proto[pix] . coef_p)), gt_masks_ds[gt_idx[p]][pix])
 * and, with gscale = d(total)/d(loss_i) (= mask_alpha/Hp/Wp/total_pos), dproto [Hp*Wp][32] (overwritten) and
 * dcoef_full[anchor_idx[p]][32] (rows of the positives overwritten).  proto [Hp*Wp][32], coef_pos [n][32], box_pos [n][4]
 * (matched gt boxes: crop window, padding 1, and area), gt_masks_ds [n_gt][Hp*Wp] in {0,1}, n <= 128.
 * loss_accum is a device fp64 scalar the caller zeroes once per batch. */
size_t ym_mask_loss_workspace_bytes(void);
int ym_mask_loss_fwd_bwd(const float* proto, const float* coef_pos, const float* box_pos, const int32_t* gt_idx,
                         const float* gt_masks_ds, const int64_t* anchor_idx, int n, int Hp, int Wp, float wscale, float gscale,
                         double* loss_accum, float* dproto, float* dcoef_full, void* workspace, size_t workspace_bytes,
                         ym_stream_t s);
/* The same for a batch in one launch pair (workgroup row = image), reading the positives' coefficients / matched boxes /
 * matched gt index straight from the per-image full tensors through anchor_idx (no gathered copies): items is a HOST array. */
typedef struct {
    const float* proto;          /* [Hp*Wp][32] */
    const float* coef_full;      /* [N][32] coefficient predictions of the image */
    const float* anchor_box;     /* [N][4]  matched gt box per anchor (ym_match_anchors) */
    const int64_t* anchor_gt;    /* [N]     matched gt index per anchor */
    const float* gt_masks_ds;    /* [n_gt][Hp*Wp] in {0,1} */
    const int64_t* anchor_idx;   /* [n]     anchors trained on (the positives, sub-sampled to masks_to_train) */
    int32_t n;                   /* 0 <= n <= 128; 0: the item is skipped */
    float wscale;                /* positives / n */
    const int32_t* n_dev;        /* NULL, or a device int32 holding the image's positive count: the kernel trains on the first
                                    min(*n_dev, n) entries of anchor_idx and uses wscale = *n_dev / that (the host never reads it) */
    float* dproto;               /* [Hp*Wp][32] overwritten (untouched when n == 0) */
    float* dcoef_full;           /* [N][32] rows anchor_idx overwritten */
} ym_mask_loss_item;
size_t ym_mask_loss_batch_workspace_bytes(int B);
/* total_pos_dev: NULL, or a device int32 by which gscale is divided (gscale = mask_alpha / Hp / Wp then). */
int ym_mask_loss_batch(const ym_mask_loss_item* items, int B, int Hp, int Wp, float gscale, const int32_t* total_pos_dev,
                       double* loss_accum, void* workspace, size_t workspace_bytes, ym_stream_t s);

/* match() for ONE image (utils/box_utils.py:57-83, encode :104-114): gt_boxes_cls [g][5] = (x1,y1,x2,y2,class) in [0,1]
 * coordinates, anchors [N][4] (cx,cy,w,h).  Writes offsets [N][4], conf [N] int64 (class+1 / 0 background / -1 neutral),
 * anchor_box [N][4] (matched gt corners), anchor_gt [N] int64 (matched gt index).  First maximum wins a tie (torch.max), the
 * last gt wins a shared best anchor (the sequential loop :72-73).  1 <= g <= 256; workspace >= 4*N bytes. */
int ym_match_anchors(const float* gt_boxes_cls, int g, const float* anchors, int N, float pos_thre, float neg_thre,
                     float* offsets, int64_t* conf, float* anchor_box, int64_t* anchor_gt, void* workspace,
                     size_t workspace_bytes, ym_stream_t s);

/* The same for a batch in one launch (workgroup = image): gt_boxes_cls / g are HOST arrays of B device pointers / gt counts;
 * the outputs are the [B][N]... batch buffers; workspace >= 4*B*N bytes. */
int ym_match_anchors_batch(const float* const* gt_boxes_cls, const int32_t* g, int B, const float* anchors, int N, float pos_thre,
                           float neg_thre, float* offsets, int64_t* conf, float* anchor_box, int64_t* anchor_gt, void* workspace,
                           size_t workspace_bytes, ym_stream_t s);

/* category_loss (modules/yolact.py:205-232: OHEM hard negatives at neg_pos_ratio, softmax CE summed / total positives) and
 * box_loss (:234-239: smooth-L1 over positives / total positives) for a batch, with their gradients (d total / d input):
 * class_p [B][N][C], box_p/offsets [B][N][4], conf [B][N] -> dclass, dbox (fully overwritten), num_pos int32 [B+1]
 * (per image, then the total) and the two device fp64 scalars loss_c, loss_b (already scaled by conf_alpha / bbox_alpha).
 * Equal OHEM marks are ranked by anchor index (torch.sort leaves their order unspecified).  No host synchronisation. */
size_t ym_loss_workspace_bytes(int B, int N);
int ym_class_box_loss(const float* class_p, const float* box_p, const float* offsets, const int64_t* conf, int B, int N, int C,
                      float conf_alpha, float bbox_alpha, int neg_pos_ratio, float* dclass, float* dbox, int32_t* num_pos,
                      double* loss_c, double* loss_b, void* workspace, size_t workspace_bytes, ym_stream_t s);

/* The positives the mask loss trains on (modules/yolact.py:255-267): conf [B][N] (> 0 = positive), num_pos [B] their counts (from
 * ym_class_box_loss), keys [B][N] iid uniform [0,1) draws.  idx[b][0 .. min(count, cap)) = all positives of image b in anchor order
 * when count <= cap, else the `cap` positives with the largest keys (a uniformly random subset, the reference's randperm[:cap]),
 * in anchor order; equal keys by anchor index.  No host synchronisation. */
int ym_select_positives(const int64_t* conf, const float* keys, int B, int N, int cap, const int32_t* num_pos, int64_t* idx,
                        ym_stream_t s);

/* semantic_seg_loss (modules/yolact.py:293-313) for ONE image: seg_nhwc [P][pitch] logits (channels >= num_classes are
 * padding), gt_masks_ds [g][P] in {0,1} (down-sampled + binarised gt masks), gt_cls[j*gt_cls_stride] the class of gt j.
 * loss_accum += coeff * sum BCE-with-logits(seg, target) with target[c][pix] = max over gts of class c; dseg [P][pitch] =
 * coeff * (sigmoid - target), zero in the padding channels.  coeff = semantic_alpha / H / W / B. */
int ym_semantic_loss(const float* seg_nhwc, int P, int pitch, int num_classes, const float* gt_masks_ds, const int64_t* gt_cls,
                     int gt_cls_stride, int g, float coeff, float* dseg, double* loss_accum, ym_stream_t s);
/* The same for the batch in one launch: seg_nhwc / dseg [B][P][pitch]; gt_masks_ds / gt_cls / g are HOST arrays of B device
 * pointers / gt counts (g[i] = 0: no gt, pointers ignored). */
int ym_semantic_loss_batch(const float* seg_nhwc, int B, int P, int pitch, int num_classes, const float* const* gt_masks_ds,
                           const int64_t* const* gt_cls, int gt_cls_stride, const int32_t* g, float coeff, float* dseg,
                           double* loss_accum, ym_stream_t s);

int ym_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, ym_stream_t s);
/* The training pair that carries the argmax instead of re-deriving it: fwd_idx also writes idx [B][Ho][Wo][C] uint8 (window
 * position 3*dy + dx of the winner under ATen's rule, 4-byte aligned), bwd_idx gathers dx from idx + dy alone (x is not read). */
int ym_maxpool3x3s2_fwd_idx(const float* in, float* out, uint8_t* idx, int B, int H, int W, int C, ym_stream_t s);
int ym_maxpool3x3s2_bwd_idx(const uint8_t* idx, const float* dy, float* dx, int B, int H, int W, int C, ym_stream_t s);
int ym_bilinear2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, int align_corners, ym_stream_t s);

/* torch.optim.SGD(momentum, weight_decay) on one flat fp32 buffer (reference train.py:61,130). */
int ym_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                float weight_decay, int first_step, ym_stream_t s);

/* ---- small NHWC ops ------------------------------------------------------------------------------ */

/* MaxPool2d(3, stride 2, pad 1), reference modules/resnet.py:91. C % 4 == 0. */
int ym_maxpool3x3s2_fwd(const float* in, float* out, int B, int H, int W, int C, ym_stream_t s);
/* The ResNet stem in eval mode as ONE launch (modules/resnet.py:86-91: conv1 7x7/2 pad 3 -> bn1 -> relu -> maxpool 3x3/2 pad 1):
 * img NCHW [B][3][H][W] (read directly: no NHWC4 copy), w_packed = ym_pack_conv_weight(conv1.weight, cin_pad 4, k_pad 224)
 * [64][224], scale / shift [64] from ym_fold_bn, out NHWC [B][Hp][Wp][64] with Hp = ((H-1)/2+1 + 1)/2 ... (the two layers'
 * usual output sizes).  The 64-channel conv output never reaches HBM.  Same MFMA order and epilogue arithmetic as
 * ym_nchw_to_nhwc4 + ym_conv2d_fwd (stem mode) + ym_maxpool3x3s2_fwd: the same bits. */
int ym_stem_conv_bn_relu_maxpool(const float* img_nchw, const float* w_packed, const float* scale, const float* shift, float* out,
                                 int B, int H, int W, int k_pad, ym_stream_t s);

/* Bilinear x2 upsample, align_corners 0 (FPN, modules/yolact.py:70-71) or 1 (ProtoNet, :43). C % 4 == 0. */
int ym_bilinear2x_fwd(const float* in, float* out, int B, int H, int W, int C, int align_corners, ym_stream_t s);

/* softmax over the last dim of [rows][C] (F.softmax(class_pred, -1), modules/yolact.py:163). in may equal out. */
int ym_softmax_rows(const float* in, float* out, int64_t rows, int C, ym_stream_t s);

/* ---- Swin-T blocks (modules/swin_transformer.py); Linear layers are 1x1 ym_conv2d_fwd on [tokens][C] ---------- */

/* nn.LayerNorm over the last dim of [M][C] (C % 4 == 0, C <= 1536); in-place allowed. (:225,228,297,425,:470) */
int ym_layernorm(const float* x, const float* gamma, const float* beta, float eps, float* out, int64_t M, int C,
                 ym_stream_t s);

/* PatchMerging gather + norm (:299-323): x NHWC [B][H][W][C] -> out [B*ceil(H/2)*ceil(W/2)][4C] =
 * LayerNorm(concat(x[2i,2j], x[2i+1,2j], x[2i,2j+1], x[2i+1,2j+1])), zero padded for odd H/W. */
int ym_patch_merge_layernorm(const float* x, int B, int H, int W, int C, const float* gamma, const float* beta, float eps,
                             float* out, ym_stream_t s);

/* (Shifted-)window multi-head self-attention, everything between the qkv and proj Linears of one block
 * (WindowAttention.forward :172-199 + pad/roll/window_partition/window_reverse/un-roll/crop of :249-283):
 * qkv [B*H*W][3C] (q|k|v, head-major inside each), qkv_bias [3C] (value of padded tokens), rel_bias_table
 * [(2*window-1)^2][heads]; window must be 7 and C/heads == 32; shift 0 or window/2.  out [B*H*W][C]. */
int ym_swin_window_attention(const float* qkv, const float* qkv_bias, const float* rel_bias_table, int B, int H, int W, int C,
                             int heads, int window, int shift, float* out, ym_stream_t s);

/* ---- pre-processing (SURVEY.md §8f "next" row 1) ------------------------------------------------------------- */

/* `val_aug(img, val_size)` (utils/augmentations.py:219-227 = pad_to_square :138-165 + cv2.resize :188 +
 * normalize_and_toRGB :212-216) on the device: HWC BGR image (uint8 if is_uint8 else float32, DEVICE pointer) ->
 * out [3][S][S] float32 RGB, (x - mean) / std.  mean_bgr / std_bgr are HOST pointers to 3 floats (config.py:66-67). */
int ym_val_preprocess(const void* img_hwc_bgr, int is_uint8, int H, int W, int S, const float* mean_bgr,
                      const float* std_bgr, float* out_chw_rgb, ym_stream_t s);

/* ---- detection post-processing (utils/output_utils.py) ---------------------------------------------- */

typedef struct {
    int32_t num_anchors;     /* N */
    int32_t num_classes;     /* C including background (81) */
    int32_t coef_dim;        /* 32 */
    int32_t top_k;           /* cfg.top_k = 200 (<= 256) */
    int32_t max_det;         /* cfg.max_detections = 100 (<= 128) */
    float score_thre;        /* cfg.nms_score_thre = 0.05 */
    float iou_thre;          /* cfg.nms_iou_thre = 0.5 */
    float img_size;          /* cfg.img_size, used by the greedy ("traditional") path only */
} ym_nms_cfg;

size_t ym_nms_workspace_bytes(const ym_nms_cfg* cfg);

/* `nms()` with fast_nms (utils/output_utils.py:126-163 + :11-43 + box_iou utils/box_utils.py:8-37)
 * for ONE image.  Inputs: class_pred [N][C] (softmaxed), box_pred [N][4], coef_pred [N][coef_dim],
 * anchors [N][4] (cx,cy,w,h).  Outputs (device): out_count int32[1] (n <= max_det; 0 means the
 * reference returns five Nones), out_ids int64[max_det], out_scores f32[max_det],
 * out_boxes f32[max_det][4] (x1,y1,x2,y2 in 0..1), out_coefs f32[max_det][coef_dim]. */
int ym_detect_fast_nms(const float* class_pred, const float* box_pred, const float* coef_pred,
                       const float* anchors, const ym_nms_cfg* cfg, int32_t* out_count, int64_t* out_ids,
                       float* out_scores, float* out_boxes, float* out_coefs, void* workspace,
                       size_t workspace_bytes, ym_stream_t s);

/* The same for a batch of B images in ONE launch set (grid row = image; SURVEY.md §0.3: the reference's nms() is batch-1 only,
 * eval.py loops over images): class_pred [B][N][C], box_pred [B][N][4], coef_pred [B][N][coef_dim]; outputs out_count int32[B],
 * out_ids [B][max_det], out_scores [B][max_det], out_boxes [B][max_det][4], out_coefs [B][max_det][coef_dim] (rows past an
 * image's count are unspecified).  Per image the result equals ym_detect_fast_nms on that image.
 * workspace >= ym_nms_batch_workspace_bytes(cfg, B). */
size_t ym_nms_batch_workspace_bytes(const ym_nms_cfg* cfg, int B);
int ym_detect_fast_nms_batch(const float* class_pred, const float* box_pred, const float* coef_pred, const float* anchors,
                             const ym_nms_cfg* cfg, int B, int32_t* out_count, int64_t* out_ids, float* out_scores,
                             float* out_boxes, float* out_coefs, void* workspace, size_t workspace_bytes, ym_stream_t s);

/* Same contract, greedy per-class NMS: replaces traditional_nms (utils/output_utils.py:84-123) and the
 * Cython kernel it calls (cython_nms.pyx:24-74) without the 80 D2H/H2D round trips. */
int ym_detect_greedy_nms(const float* class_pred, const float* box_pred, const float* coef_pred,
                         const float* anchors, const ym_nms_cfg* cfg, int32_t* out_count, int64_t* out_ids,
                         float* out_scores, float* out_boxes, float* out_coefs, void* workspace,
                         size_t workspace_bytes, ym_stream_t s);

/* y[i] = exp(x[i]) rounded to nearest float — the exp of the box decode (utils/output_utils.py:150, `torch.exp`), exposed so
 * that the parity suite can compare it with oracle/expf_cr.c on arbitrary inputs (bit-identical by construction: same IEEE
 * double operation sequence).  The reference's own torch.exp is MKL VML (1 ulp off this value in 1.1 % of inputs). */
int ym_expf_cr(const float* x, float* y, int64_t n, ym_stream_t s);

/* Drop-in for `cython_nms.nms(dets, thresh)` (cython_nms.pyx:24) on device data: dets [n][5]
 * (x1,y1,x2,y2,score), "+1" areas, suppress ovr >= thresh; keep_mask uint8[n] (1 = kept), in original
 * index order like np.where(suppressed == 0).  workspace >= ym_greedy_nms_workspace_bytes(n). */
size_t ym_greedy_nms_workspace_bytes(int n);
int ym_greedy_nms(const float* dets, int n, float thresh, uint8_t* keep_mask, int32_t* out_count,
                  void* workspace, size_t workspace_bytes, ym_stream_t s);

/* Mask assembly: sigmoid(proto[Hp*Wp][K] @ coef[n][K]^T) cropped to the box window (padding 1) ->
 * out [n][Hp][Wp].  Replaces utils/output_utils.py:217-222 (+ crop, utils/box_utils.py:147-168) and
 * the same product in lincomb_mask_loss (modules/yolact.py:275-276).  K must be 32. do_crop=0 skips crop. */
int ym_mask_assemble(const float* proto, const float* coefs, const float* boxes, int n, int Hp, int Wp,
                     int K, int do_crop, float* out, ym_stream_t s);

/* F.interpolate(masks, (S,S), bilinear, align_corners=False) -> gt(0.5) -> slice to [n][img_h][img_w]
 * (utils/output_utils.py:224-228), S = max(img_h, img_w).  out values are exactly 0.0f / 1.0f. */
int ym_mask_resize_binarize(const float* masks, int n, int Hp, int Wp, int img_h, int img_w, float* out,
                            ym_stream_t s);

/* box_p *= S; box_p.int()  (utils/output_utils.py:230-231): boxes_f is scaled IN PLACE like the reference. */
int ym_boxes_to_pixels(float* boxes_f, int32_t* boxes_px, int n, float S, ym_stream_t s);

/* after_nms (utils/output_utils.py:200-233) for a batch of B images in one launch set: prototype x coefficient product, sigmoid,
 * crop, bilinear resize to S = max(img_h, img_w), > 0.5, slice — fused, the [n][Hp][Wp] soft masks never reach HBM — plus the
 * in-place box scaling + truncation.  proto [B][Hp][Wp][32]; coefs [B][max_det][32], boxes [B][max_det][4] (scaled IN PLACE),
 * counts int32[B] on the DEVICE (NULL = every image has max_det detections; B = 1 with max_det = n is the single-image call);
 * masks [B][max_det][img_h][img_w] (only slots < count are written), boxes_px int32 [B][max_det][4].
 * workspace >= ym_after_nms_batch_workspace_bytes(...) (0 unless the image is much smaller than the prototype map). */
size_t ym_after_nms_batch_workspace_bytes(int max_det, int Hp, int Wp, int img_h, int img_w);
int ym_after_nms_batch(const float* proto, const float* coefs, float* boxes, const int32_t* counts, int B, int max_det, int Hp,
                       int Wp, int K, int img_h, int img_w, int do_crop, float* masks, int32_t* boxes_px, void* workspace,
                       size_t workspace_bytes, ym_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* YOLACT_HIP_H */
