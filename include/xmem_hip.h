/*
 * xmem_hip.h - C-ABI of the MI355X (gfx950) kernels behind XMem++'s per-frame
 * space-time memory path.
 *
 * The reference (mbzuai-metaverse/XMem2) has no FFI or operator registry: its hot
 * path is a chain of stock ATen calls issued from Python.  This header is therefore
 * the boundary a maintainer would bind in place of those ATen call sites; every
 * entry point cites the reference lines it replaces (paths relative to the
 * reference checkout).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless
 *     the name ends in _host;  `stream` is a hipStream_t passed as void*.
 *   - every function returns 0 on success or a negative xmem_status code; nothing
 *     throws across the ABI; no function allocates or frees caller-visible memory
 *     (scratch is passed in, sized by the matching *_workspace_bytes function).
 *   - no global mutable state: safe to call concurrently on different streams/devices.
 *   - activations are NHWC fp32: [B][H][W][C] with an explicit pixel stride `ld*`
 *     (in floats) so a tensor may live inside a wider (concatenated) buffer.
 *   - memory elements are rows: keys [N][C_k], values [N][C_v], shrinkage [N].
 */
#ifndef XMEM_HIP_H
#define XMEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    XMEM_OK = 0,
    XMEM_ERR_BAD_ARG = -1,      /* null pointer / non-positive size / unsupported dimension */
    XMEM_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels implement (message says which) */
    XMEM_ERR_WORKSPACE = -3,    /* workspace too small */
    XMEM_ERR_LAUNCH = -4,       /* hipLaunch / hipGetLastError failure */
    XMEM_ERR_TOPK = -5          /* fewer memory elements than top_k (torch.topk raises, memory_util.py:46) */
} xmem_status;

/* ABI version of this header.  2 (round 4): xmem_conv_desc grew (in_half / out_half / w_half), storage-typed `_t` entry points, plan
 * tiles 23..40.  3 (round 5): no layout change, but the MEANING of w_winograd4 / w_winograd4_split changed - the F(4x4) transforms use the
 * interpolation points (0, +-3/4, +-3/2, inf), a caller must form G g G^T with the matching G (see xmem_conv_desc.w_winograd4).  A caller compiled against another version must not pass structs: check xmem_version() == XMEM_ABI_VERSION at load. */
#define XMEM_ABI_VERSION 3
int xmem_version(void);
const char* xmem_last_error_string(int code); /* static string for a status code */

/* Measurement aid: launches the empty kernel `xmem_trace_marker_kernel` on `stream`.  A rocprofv3 kernel trace of a
 * process that brackets a region with two markers can be cut to exactly that region (bench.py does; the reference
 * brackets the same region with perf_counter(), inference/run_on_video.py:106-113). */
int xmem_trace_marker(int tag, void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolution (implicit GEMM on v_mfma_f32_32x32x2_f32) with fused epilogue.
 * Replaces nn.Conv2d (+ eval BatchNorm2d + ReLU + residual add) call sites:
 *   model/resnet.py:59-75,95-114 (BasicBlock/Bottleneck), model/modules.py:166-175,
 *   135-142, 207-211, 186-191, 229-250, model/group_modules.py:25-52 (GConv2D/GroupResBlock).
 *
 *   out[b,oh,ow,n] = act( (sum_{kh,kw,c} in'[b,oh*s-p+kh,ow*s-p+kw,c] * w[n,kh,kw,c]) * scale[n] + shift[n]
 *                         + res[b,oh,ow,n] )
 *   in' = relu(in) if relu_in.  scale/shift hold the folded BatchNorm (scale = gamma/sqrt(var+eps),
 *   shift = beta - mean*scale) or (1, bias).  res may be NULL.  Weights are [Cout][KH][KW][Cin]
 *   (Cin contiguous, Cin % 4 == 0; pad with zero channels if needed).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const float* in;    int B, H, W, Cin, ldin;
    const float* w;     int Cout, KH, KW, stride, pad;
    const float* scale; const float* shift;
    const float* res;   int ldres;
    float* out;         int ldout;
    int relu_in, relu_out;
    int plan_tile;      /* 0 = built-in heuristic; 1..6 = {128x128, 128x64, 64x64} x {BK 32, BK 64} (autotuner override);
                           7..12 = the same GEMM tiles inside the Winograd F(2x2,3x3) path (needs w_winograd);
                           13..15 = fused Winograd GEMM + output transform, tiles {128x64, 64x64, 64x128};
                           16 = REDUCED PRECISION (opt-in mode only, never the default): Winograd path whose transformed
                                operands are stored in fp16 and multiplied on v_mfma_f32_32x32x16_f16 with fp32 accumulation
                                (needs w_winograd_f16, Cin % 64 == 0) - the counterpart of the reference's
                                torch.cuda.amp.autocast loop, inference/run_on_video.py:76 */
    int plan_splitk;    /* 0 = heuristic; >0 = number of K splits */
    const float* w_winograd; /* optional [16][Cout][Cin]: G g G^T of the 3x3 filter (3x3 / stride 1 / pad 1 only) */
    int res_broadcast;  /* 1: res is ONE image [Ho][Wo][ldres] added to every batch element (the per-object halves of the
                           fuser convolutions share the f16 half, model/modules.py:31-41 on cat([x, g])) */
    const void* w_winograd_f16; /* optional [16][Cout][Cin] IEEE half: the same G g G^T rounded to fp16 (plan_tile 16 only) */
    const float* w_winograd4;   /* optional [36][Cout][Cin]: G g G^T of Winograd F(4x4,3x3) (plan_tile 17..28 = the GEMM tiles of
                                   plans 7..12 / the streaming GEMM inside the F(4x4) path; fp32).  INTERPOLATION POINTS
                                   p = (0, 3/4, -3/4, 3/2, -3/2, inf) since ABI version 3 (version 2: 0, +-1, +-2, inf): row i of G is
                                   (1, p_i, p_i^2) / N_i with N_i = prod_{k != i} (p_i - p_k) over the finite points, the row of
                                   infinity (0, 0, 1):  64/81 0 0 | -128/243 -32/81 -8/27 | -128/243 32/81 -8/27 | 32/243 16/81 8/27 |
                                   32/243 -16/81 8/27 | 0 0 1.  Form it in fp64 and round once (xmem2_amd.ops.winograd4_weights).
                                   Error against a fp64 convolution ~3e-6 of the output scale (the textbook points: ~1e-5). */
    /* SPLIT-OPERAND ARITHMETIC (opt-in mode 'fp32x', never the default): arith = 1 and w_split != NULL run every GEMM of the
     * call on v_mfma_f32_32x32x16_f16 with each fp32 operand carried as two halfs (x = hi + lo, relative representation error
     * <= 2^-21; the four partial products are accumulated in fp32).  The split weights have the SHAPE of their fp32
     * counterparts ([Cout][KH][KW][Cin], [16][Cout][Cin], [36][Cout][Cin]) with every group of four input channels stored as
     * eight halfs [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] (16 bytes, like the four floats), pre-multiplied by a power of two
     * 2^s that the caller folds into `scale` (scale * 2^-s: exact).  Activations stay fp32 in memory; the Winograd-domain
     * intermediate V is written in the same group format.  Same call sites as above; arith = 0 ignores these fields. */
    int arith;
    const void* w_split;
    const void* w_winograd_split;
    const void* w_winograd4_split;
    /* FP16 LOOP (config['precision'] = 'fp16', opt-in, never the default; mirrors the reference's GPU mode: torch.cuda.amp.autocast
     * around the frame loop, inference/run_on_video.py:76, fp32 preload :59-66).  in_half = 1: `in` is [B][H][W][ldin] IEEE halfs
     * (Cin % 8 == 0, ldin % 8 == 0, 16-byte aligned) and w_half holds the weights [Cout][KH][KW][Cin] as halfs.  ZERO-PADDING
     * CONTRACT: the kernel reads all Cin (the layer's channel count padded to 8) halfs of every pixel; a caller whose layer has
     * fewer true channels keeps the padding channels of `in` ZERO (finite is not enough of a promise: the zero weights there would turn
     * an Inf / NaN into NaN), and a channel slice of a wider buffer holds a multiple of 8 channels and ends inside its pixel.  The contraction runs
     * on v_mfma_f32_32x32x16_f16 in the DIRECT form (no Winograd: plan_tile 1..3 / 0) with fp32 accumulation and an fp32 epilogue.
     * out_half = 1: `out` and `res` are halfs too (ldout / ldres count halfs; the result is rounded once, to nearest even);
     * out_half = 0 stores fp32 (key projection, mask head).  plan_tile 23..40, arith and the Winograd operands are ignored. */
    int in_half, out_half;
    const void* w_half;
} xmem_conv_desc;

size_t xmem_conv2d_workspace_bytes(const xmem_conv_desc* d);
int xmem_conv2d_nhwc(const xmem_conv_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Deterministic permanent-memory augmentations on the device: every augmented uint8 frame and float mask
 * of one annotated frame in ONE launch.  Replaces the per-annotation loop of inference/run_on_video.py:231-242 over
 * get_determenistic_augmentations(subset) (inference/frame_selection/frame_selection_utils.py:50-218): ColorJitter brightness
 * (1.5 / 0.5), Grayscale(3), RandomPosterize(3), RandomAdjustSharpness(16), gaussian_blur(7) on the image with the mask kept;
 * RandomAffine rotations / scalings / shears / translations on image (PIL: nearest, zero fill) AND mask (tensor branch:
 * affine grid + grid_sample nearest).  img [H][W][3] uint8 (decoded frame at working size), mask [K][H][W] float or NULL,
 * out_img [n_aug][H][W][3], out_mask [n_aug][K][H][W] or NULL.  The arithmetic of every step is restated in the precision
 * the host libraries use (csrc/augment.hip); a load-time call (it synchronises the stream once), not for graph capture.
 * ------------------------------------------------------------------------------------------ */
enum { XMEM_AUG_BRIGHTNESS = 0, XMEM_AUG_POSTERIZE = 1, XMEM_AUG_GRAY = 2, XMEM_AUG_SHARPNESS = 3, XMEM_AUG_BLUR7 = 4, XMEM_AUG_AFFINE = 5 };
#define XMEM_AUG_MAX 32
typedef struct {
    int type;                 /* XMEM_AUG_* */
    float factor;             /* brightness / sharpness factor, posterize bits */
    double image_matrix[6];   /* AFFINE, image side: PIL's inverse matrix (output pixel centre -> input coordinate),
                                 torchvision _get_inverse_affine_matrix about the image centre */
    float mask_grid[6];       /* AFFINE, mask side: (theta^T / [W/2, H/2]) as r00 r10 r20 r01 r11 r21 of the tensor branch's
                                 _gen_affine_grid (matrix about the origin) */
} xmem_aug_desc;
size_t xmem_augment_workspace_bytes(int n_aug, int H, int W);
int xmem_augment_frames(const uint8_t* img, const float* mask, int H, int W, int K, const xmem_aug_desc* descs, int n_aug,
                        uint8_t* out_img, float* out_mask, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pooling / resampling / gating kernels of the encoders and the decoder.
 * ------------------------------------------------------------------------------------------ */
/* nn.MaxPool2d(3, stride 2, pad 1), model/resnet.py:123; in [B][H][W][C] -> out [B][Ho][Wo][C] */
int xmem_maxpool3x3s2(const float* in, float* out, int B, int H, int W, int C, void* stream);

/* F.interpolate(scale 2, bilinear, align_corners=False) of g [B][h][w][C] plus the broadcast skip
 * feature [2h][2w][C]: UpsampleBlock, model/modules.py:186-190 + group_modules.py:15-23 */
int xmem_upsample2x_add(const float* g, const float* skip, float* out, int B, int h, int w, int C, void* stream);

/* F.interpolate(mode='area') by an integer ratio r (2 or 4), group_modules.py:22-23;
 * in [B][H][W][C] (pixel stride ldin) -> out [B][H/r][W/r][C] (pixel stride ldout) */
int xmem_area_downsample(const float* in, int ldin, float* out, int ldout, int B, int H, int W, int C, int r, void* stream);

/* dst[b][p][dst_off + c] = src[(b % srcB)][p][src_off + c]: builds the channel concatenations of
 * MainToGroupDistributor (group_modules.py:55-82) and torch.cat([g, h], 2) (modules.py:59,89). */
int xmem_copy_channels(const float* src, int ldsrc, int srcB, float* dst, int lddst, int B, int P, int C, void* stream);

/* The input of HiddenUpdater's three pointwise convolutions as one concatenated tensor, in one launch (model/modules.py:49-57:
 * g16_conv(g[0]) + g8_conv(downsample_groups(g[1], 1/2)) + g4_conv(downsample_groups(g[2], 1/4)), g[2] = cat(g4, logits)):
 * out [K][h][w][ldout] <- [ g16 [K][h][w][c16] | area2(g8 [K][2h][2w][c8]) | area4(g4 [K][4h][4w][c4]) | area4(logits [K][4h][4w][1]) ];
 * channels past c16 + c8 + c4 + 1 are not written.  Same bits as xmem_copy_channels + three xmem_area_downsample calls. */
int xmem_hidden_update_gather(const float* g16, int c16, const float* g8, int c8, const float* g4, int c4, const float* logits,
                              float* out, int ldout, int K, int h, int w, void* stream);

/* CBAM (model/cbam.py:21-77) on g [B][P=H*W][C] and the residual add of FeatureFusionBlock
 * (modules.py:36-39): out = g + CBAM(g).  mlp weights as in the checkpoint: w1 [C/16][C], b1, w2 [C][C/16], b2;
 * spatial 7x7 conv weight sw [2][7][7] (channel 0 = max, 1 = mean), bias sb[1].
 * workspace: xmem_cbam_workspace_bytes(B, H*W, C). */
size_t xmem_cbam_workspace_bytes(int B, int P, int C);
int xmem_cbam_residual(const float* g, float* out, int B, int H, int W, int C,
                       const float* w1, const float* b1, const float* w2, const float* b2,
                       const float* sw, const float* sb, void* workspace, size_t workspace_bytes, void* stream);

/* GRU-like gate shared by HiddenUpdater / HiddenReinforcer (modules.py:63-72, 93-99):
 * values [B][P][3*Ch] = (forget | update | new), h [B][P][Ch] -> new_h [B][P][Ch] */
int xmem_gru_gate(const float* values, const float* h, float* new_h, int B, int P, int Ch, void* stream);

/* y = a + b + c elementwise (n floats), HiddenUpdater sum modules.py:56-57 */
int xmem_add3(const float* a, const float* b, const float* c, float* y, size_t n, void* stream);

/* STORAGE-TYPED VARIANTS (the fp16 loop, config['precision'] = 'fp16': activations live in HBM as IEEE halfs, as the tensors of
 * the reference's autocast frame loop do, inference/run_on_video.py:76).  Same kernels, same fp32 arithmetic; `*_half` flags give
 * the storage type of each tensor (0 = float, 1 = half; leading dimensions count elements of that type); a stored value is
 * rounded once, to nearest even.  The plain functions above are the all-float instantiations.  The GRU state stays fp32. */
int xmem_maxpool3x3s2_t(const void* in, int in_half, void* out, int out_half, int B, int H, int W, int C, void* stream);
int xmem_upsample2x_add_t(const void* g, const void* skip, void* out, int half, int B, int h, int w, int C, void* stream);
int xmem_area_downsample_t(const void* in, int in_half, int ldin, void* out, int out_half, int ldout, int B, int H, int W, int C, int r, void* stream);
int xmem_copy_channels_t(const void* src, int src_half, int ldsrc, int srcB, void* dst, int dst_half, int lddst, int B, int P, int C, void* stream);
int xmem_cbam_residual_t(const void* g, void* out, int half, int B, int H, int W, int C,
                         const float* w1, const float* b1, const float* w2, const float* b2,
                         const float* sw, const float* sb, void* workspace, size_t workspace_bytes, void* stream);
int xmem_gru_gate_t(const void* values, int values_half, const float* h, float* new_h, int B, int P, int Ch, void* stream);

/* image [3][H][W] (NCHW, unpadded) -> [Hp][Wp][4] NHWC, zero padded as pad_divide_by
 * (util/tensor_util.py:47-61): left/top pads lw, lh; 4th channel zero. */
int xmem_pack_image(const float* img, float* out, int H, int W, int Hp, int Wp, int lh, int lw, void* stream);

/* Frame ingest on the device: decoded uint8 image [H][W][3] (RGB, HWC) -> the same padded NHWC4
 * tensor, after transforms.ToTensor + im_normalization (inference/data/video_reader.py:61-76,
 * dataset/range_transform.py:5-8): ((x / 255) - mean[c]) / std[c] in fp32, in that operation order.
 * mean3_host / std3_host: 3 floats each, HOST pointers. */
int xmem_pack_image_u8(const uint8_t* img, float* out, int H, int W, int Hp, int Wp, int lh, int lw,
                       const float* mean3_host, const float* std3_host, void* stream);

/* value-encoder input, model/network.py:73-81 + modules.py:126-131: per object k the 5 channels
 * (r,g,b,mask_k,sum_{j!=k} mask_j) padded to 8: image4 [Hp][Wp][4], masks [K][Hp][Wp] -> out [K][Hp][Wp][8] */
int xmem_pack_value_input(const float* image4, const float* masks, float* out, int K, int Hp, int Wp, void* stream);

/* KeyProjection activations, modules.py:207-211: proj [P][ldp] = (key C_k | d 1 | e C_k | pad) ->
 * key [P][C_k], shrinkage [P] = d^2+1, selection [P][C_k] = sigmoid(e) */
int xmem_key_post(const float* proj, int ldp, float* key, float* shrinkage, float* selection, int P, int Ck, void* stream);

/* Decoder tail + segment(): model/modules.py:247-248 (x4 bilinear), network.py:111-115 (sigmoid),
 * model/aggregate.py:6-17 (soft aggregation + softmax), util/tensor_util.py:63-77 (unpad).
 * logits [K][h4][w4] -> prob [K+1][H][W] (NCHW, crop offsets lh, lw inside the padded 4*h4 x 4*w4 frame);
 * prob_padded (nullable) receives the uncropped [K+1][4*h4][4*w4]. */
int xmem_logits_to_prob(const float* logits, float* prob, float* prob_padded, int K, int h4, int w4,
                        int H, int W, int lh, int lw, void* stream);

/* aggregate() of given masks (inference_core.py:128,163): masks [K][H][W] -> prob [K+1][H][W] */
int xmem_aggregate_masks(const float* masks, float* prob, int K, int H, int W, void* stream);

/* Given-mask / prediction merge of InferenceCore.step (inference/inference_core.py:117-127):
 * region = sum_k mask_k > 0.5; out_k = mask_k if bit k of valid_bits else (region ? 0 : pred_k).  [K][H][W] each. */
int xmem_merge_masks(const float* pred_no_bg, const float* mask, uint64_t valid_bits, float* out, int K, int H, int W, void* stream);

/* F.interpolate(prob, shape, mode='bilinear', align_corners=False) of _post_process (inference/run_on_video.py:166-168);
 * in [C][Hi][Wi] -> out [C][Ho][Wo] */
int xmem_resize_bilinear(const float* in, float* out, int C, int Hi, int Wi, int Ho, int Wo, void* stream);

/* torch.argmax(prob, dim=0) -> uint8, inference/run_on_video.py:170-172; prob [C][H][W] */
int xmem_argmax_u8(const float* prob, uint8_t* out, int C, int H, int W, void* stream);

/* NHWC [B][P][C] (pixel stride ld) <-> NCHW [B][C][P] layout transposes for the Python surface */
int xmem_nhwc_to_nchw(const float* in, int ld, float* out, int B, int P, int C, void* stream);
int xmem_nchw_to_nhwc(const float* in, float* out, int ld, int B, int P, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Memory readout: fused anisotropic-L2 similarity + streaming top-k + softmax.
 * Replaces get_similarity + do_softmax (model/memory_util.py:7-65) as called from
 * MemoryManager.match_memory (inference/memory_manager.py:82-120,143-177) without ever
 * materialising the N x HW matrix, and `_readout` (memory_manager.py:57-59,185-188).
 * ------------------------------------------------------------------------------------------ */
#define XMEM_MAX_SEGMENTS 4
typedef struct {
    const float* key;        /* [n][C_k] rows */
    const float* shrinkage;  /* [n] or NULL (treated as 1, memory_util.py:36-37) */
    int n;                   /* elements in this segment (may be 0) */
    const void* rows16;      /* optional [n][XMEM_ROWS16_HALFS] IEEE halfs: the operand rows of the fp16 filter for these keys, kept
                                by the caller across calls (xmem_affinity_rows16 makes them; they depend on key and shrinkage only).
                                NULL: the call derives them into its workspace.  Never changes a result. */
} xmem_key_segment;
#define XMEM_ROWS16_HALFS 144

/* Filter operand rows of `n` memory elements (csrc/affinity_common.hpp: [ms/8 x^2 | ms/8 x | 16 augmentation terms] in fp16,
 * 288 bytes per element).  A store calls it once per appended / replaced block and hands the rows to xmem_affinity_topk_hinted
 * through xmem_key_segment.rows16 - the per-call rows kernel and its N x 288 bytes of writes disappear from the frame loop. */
int xmem_affinity_rows16(const float* key, const float* shrinkage, int n, void* rows16, void* stream);

/* segments are searched as one virtual concatenation (long | temporary | permanent in the reference's
 * order, memory_manager.py:82-83); out indices are positions in that concatenation.
 * qk [HW][C_k]; qe [HW][C_k] or NULL.  top_k in [1, 64], sum(n) >= top_k else XMEM_ERR_TOPK.
 * out_w [HW][top_k] softmax weights exp(v)/sum exp(v) (no max shift, memory_util.py:48-49), sorted by
 * descending similarity; out_idx [HW][top_k]; out_sim (nullable) [HW][top_k] raw similarities. */
size_t xmem_affinity_topk_workspace_bytes(int n_total, int HW, int top_k);
int xmem_affinity_topk(const xmem_key_segment* segs_host, int n_seg,
                       const float* qk, const float* qe, int Ck, int HW, int top_k,
                       float* out_w, int32_t* out_idx, float* out_sim,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Diagnostics for tools (not needed by a caller): byte offsets, inside a workspace sized for (n_total, HW), of the per-query
 * candidate counts [HW] int32, the per-128-query-tile flags of the two filter passes [2][ceil(HW/128)] int32 and the per-query
 * lower bounds [HW] float of the last xmem_affinity_topk_hinted call that took the fp16-filter path. */
int xmem_affinity_debug_offsets(int n_total, int HW, size_t* count_off, size_t* flag_off, size_t* bound_off);

/* Measurement aid: two HIP events (hipEvent_t, created with timing enabled) that the following xmem_affinity_topk_hinted calls
 * record on their stream right before and right after the pass-1 launch of the fp16 filter kernel - the kernel bench.py's
 * `roofline` object is about (the reference times the whole step with perf_counter, inference/run_on_video.py:106-113).
 * NULL, NULL turns it off.  The pair is held per calling host thread (thread_local): only that thread's later calls record it, so the
 * library keeps no process-wide mutable state; no effect on results. */
int xmem_affinity_profile_events(void* before_filter, void* after_filter);

/* Same function with an optional HINT: `idx` are the out_idx [HW][top_k] of an earlier call on the same list of stores (the
 * previous frame of the video), `seg_n` the segment sizes of that call, `grid_w` the width of the stride-16 query grid (0: do not
 * use grid neighbours).  The hint only bounds the k-th similarity from below (any k distinct elements give a valid bound; the
 * previous frame's matches give a tight one): results are bit-identical with and without it.  With a bound (and >= 8192 memory
 * elements) the N x HW contraction runs on the fp16 matrix pipe as a rigorously bounded filter and only the surviving
 * candidates are evaluated in fp32, with the arithmetic of the fp32 select (csrc/affinity_filter.hip).
 * hint == NULL behaves as xmem_affinity_topk.  The reference recomputes everything per frame (memory_manager.py:82-120). */
typedef struct {
    const int32_t* idx; int top_k;
    int n_seg; int seg_n[XMEM_MAX_SEGMENTS];
    int grid_w;
} xmem_affinity_hint;
int xmem_affinity_topk_hinted(const xmem_key_segment* segs_host, int n_seg,
                              const float* qk, const float* qe, int Ck, int HW, int top_k,
                              const xmem_affinity_hint* hint,
                              float* out_w, int32_t* out_idx, float* out_sim,
                              void* workspace, size_t workspace_bytes, void* stream);

/* usage = affinity.sum(dim=2) (memory_util.py:62-63) restricted to [first, first+count) of the index space,
 * accumulated order-independently (64-bit fixed point) and then folded into the store counters as
 * KeyValueMemoryStore.update_usage does (kv_memory_store.py:96-103): use_count += usage; life_count += 1.
 * fx_scratch: count uint64, zeroed by the call. */
int xmem_usage_update(const float* w, const int32_t* idx, int HW, int top_k, int first, int count,
                      float* use_count, float* life_count, uint64_t* fx_scratch, void* stream);

typedef struct {
    const float* value;  /* [n][C_v] rows of ONE object */
    int n;
} xmem_value_segment;

/* out[obj][q][c] = sum_s w[q][s] * V_obj[idx[q][s]][c]   (sparse form of v @ affinity).
 * vsegs_host: n_obj * n_seg entries, object-major; all objects share the index space of the group.
 * out [n_obj][HW][ldout] NHWC rows (pixel stride ldout >= C_v). */
int xmem_readout_sparse(const xmem_value_segment* vsegs_host, int n_obj, int n_seg,
                        const float* w, const int32_t* idx, int HW, int top_k, int Cv,
                        float* out, int ldout, size_t obj_stride, void* stream);
/* The same readout with the output stored as IEEE halfs when out_half = 1 (ldout / obj_stride then count halfs): the fp16 loop's
 * decoder input.  Memory values stay fp32 (the permanent memory is preloaded in fp32, inference/run_on_video.py:59-66). */
int xmem_readout_sparse_t(const xmem_value_segment* vsegs_host, int n_obj, int n_seg,
                          const float* w, const int32_t* idx, int HW, int top_k, int Cv,
                          void* out, int out_half, int ldout, size_t obj_stride, void* stream);

/* Dense similarity (no top-k): sim[n][p], n over one segment, p over P queries.  Used by the long-term
 * consolidation (memory_manager.py:368) where the softmax runs over the candidate axis. out [P][n] (query-major). */
int xmem_similarity_dense(const float* key, const float* shrinkage, int n,
                          const float* qk, const float* qe, int P, int Ck, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Long-term consolidation (memory_manager.py:349-390) and eviction (kv_memory_store.py:160-181).
 * ------------------------------------------------------------------------------------------ */
/* usage[i] = use[i] / life[i] (kv_memory_store.py:183-189) */
int xmem_usage_ratio(const float* use, const float* life, float* usage, int n, void* stream);

/* torch.topk(values, k, largest, sorted=True) over a 1-D array by exact ranking (ties -> lower index first).
 * out_idx [k], out_val [k]. */
int xmem_topk_1d(const float* values, int n, int k, int largest, int32_t* out_idx, float* out_val, void* stream);

/* dst[i][0:C] = src[index[i]][0:C], i < n  (prototype gather, memory_manager.py:362-363) */
int xmem_gather_rows(const float* src, int C, const int32_t* index, int n, float* dst, void* stream);

/* in-place softmax over the last `count` entries of each row of sim [P][n] (do_softmax with top_k=None on
 * similarity[:, -count:], memory_util.py:55-60); entries before are zeroed. */
int xmem_softmax_rows_suffix(float* sim, int P, int n, int count, void* stream);

/* in-place top-k softmax of each row of sim [P][n], zeros elsewhere (do_softmax with top_k, memory_util.py:41-54: topk, exp
 * without max shift, / sum, scatter into zeros) on a materialised similarity; exact ties at the k-th value -> lowest indices. */
int xmem_softmax_rows_topk(float* sim, int P, int n, int k, void* stream);

/* out[p][c] = sum_i aff[p][n - count + i] * V[i][c], i < count: prototype values / shrinkage
 * (memory_manager.py:382-388).  V [count][C]. out [P][C]. */
int xmem_weighted_rows(const float* aff, int P, int n, int count, const float* V, int C, float* out, void* stream);

/* stream compaction for remove_obsolete_features: keep[i] = usage[i] > threshold (kv_memory_store.py:165);
 * out_index receives the kept indices in order, *out_count (device int) their number. */
int xmem_select_greater(const float* usage, int n, const float* threshold_dev, int32_t* out_index, int32_t* out_count, void* stream);

/* ------------------------------------------------------------------------------------------
 * Annotation-candidate selector (inference/frame_selection/frame_selection.py:99-244).
 * ------------------------------------------------------------------------------------------ */
/* Per-frame preparation (frame_selection.py:156-186): nearest-resize the C x H x W mask to h x w (as
 * torchvision Resize(NEAREST) on a tensor = F.interpolate(mode='nearest')), take the max over its channels,
 * form the composite key  c = (key * m) * alpha + key * one_minus_alpha  and expand it into the two
 * K = 2*C_k operands of the similarity: Mexp [HW][2C_k] = [c^2, c], Qexp [HW][2C_k] = [-e, 2 c e],
 * bsq [HW] = sum_c e c^2 (memory_util.py:20-27).  key / sel are [HW][C_k] rows.  mask may be NULL (c = key).
 * presence (device int32, nullable) receives #{pixels of the FULL-RES mask with max_c mask > eps}
 * (frame_selection.py:161-163). */
int xmem_selector_prepare(const float* key, const float* sel, const float* mask, int C, int H, int W,
                          int h, int w, int Ck, float alpha, float one_minus_alpha, float eps,
                          float* Mexp, float* Qexp, float* bsq, int32_t* presence, void* stream);

/* Cycle dissimilarity of every frame f against frame `chosen` (frame_selection.py:218-226):
 *   out[f] = sum_{i,j} relu( S(mem=c_chosen[i], ms=s_chosen[i]; q=c_f[j], qe=e_f[j])
 *                          - S(mem=c_f[i],      ms=s_f[i];      q=c_chosen[j], qe=e_chosen[j]) ) / HW^2
 * Mexp/Qexp [n_frames][HW][2C_k], bsq/shrinkage [n_frames][HW] from xmem_selector_prepare; valid [n_frames]
 * uint8 (nullable; 0 => out[f] = 0 without computing, frame_selection.py:201-203).  out [n_frames] double
 * (deterministic fixed-order reduction).  C_k = 64. */
size_t xmem_cycle_dissimilarity_workspace_bytes(int n_frames, int HW);
int xmem_cycle_dissimilarity(const float* Mexp, const float* Qexp, const float* bsq, const float* shrinkage,
                             int n_frames, int HW, int Ck, int chosen, const uint8_t* valid, double* out,
                             void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XMEM_HIP_H */
