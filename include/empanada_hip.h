/*
 * empanada_hip.h -- C ABI of libempanada_hip.so, the MI355X (gfx950) engine
 * behind the empanada panoptic-inference hot path.
 *
 * The reference (volume-em/empanada-napari) has no FFI: its boundary for this
 * path is the Python API of empanada/inference/engines.py and
 * empanada_napari/inference.py.  Each entry point below replaces the work one
 * of those Python functions hands to torch / torch.jit / numba, and cites it.
 * The Python mirror of the reference classes (empanada-napari_amd/engines.py,
 * inference.py) binds these symbols with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer owned by the caller;
 *     h_* is a HOST pointer; `stream` is a hipStream_t passed as void*.
 *   - return value: 0 on success, a negative emp_status otherwise; no C++
 *     exception crosses the ABI; emp_last_error() returns a static message.
 *   - nothing here synchronises the device except where stated (functions that
 *     return a host count).
 *   - dense float outputs use the reference's NCHW fp32 layout; label maps are
 *     int32 or int64 as stated per function.
 */
#ifndef EMPANADA_HIP_H
#define EMPANADA_HIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define EMP_API __attribute__((visibility("default")))
#else
#define EMP_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  EMP_OK = 0,
  EMP_ERR_INVALID = -1,   /* bad argument / unsupported shape */
  EMP_ERR_HIP = -2,       /* a HIP runtime call failed */
  EMP_ERR_STATE = -3,     /* call order (missing parameter, not finalized) */
  EMP_ERR_NOMEM = -4
} emp_status;

EMP_API const char* emp_last_error(void);
EMP_API int emp_abi_version(void);
/* number of visible HIP devices, or a negative emp_status */
EMP_API int emp_device_count(void);

/* ------------------------------------------------------------------------
 * 1. Network forward (hot loop 1)
 *    replaces: model(image, render_steps, interpolate_ins) as called from
 *    PanopticDeepLabRenderEngine.infer, empanada/inference/engines.py:248-255,
 *    i.e. QuantizablePanopticDeepLabPR.forward,
 *    empanada/models/quantization/panoptic_deeplab.py:238-250, or (arch 1)
 *    QuantizablePanopticBiFPNPR.forward, quantization/panoptic_bifpn.py:147-161
 *    (H and W must then be multiples of 128).
 * ---------------------------------------------------------------------- */
typedef struct emp_pdl emp_pdl_t;

typedef struct {
  int32_t num_classes;            /* semantic channels C (1 = binary/sigmoid)          */
  int32_t stage4_stride;          /* 16 (layer4 dilated) or 32                         */
  int32_t decoder_channels;       /* 256                                               */
  int32_t aspp_channels;          /* 0 -> decoder_channels                             */
  int32_t n_stages;               /* number of low-level decoder stages (<=3)          */
  int32_t low_level_stages[3];    /* pyramid index per stage (1..3)                    */
  int32_t low_level_proj_sem[3];  /* projected channels, semantic decoder              */
  int32_t low_level_proj_ins[3];  /* projected channels, instance decoder              */
  int32_t atrous_rates[3];
  int32_t ins_decoder;            /* 1: separate instance decoder                      */
  int32_t num_fc;                 /* PointRend MLP depth (3)                           */
  int32_t subdivision_num_points; /* 8192                                              */
  int32_t arch;                   /* 0 PanopticDeepLabPR (MitoNet_v1), 1 PanopticBiFPNPR (MitoNet_v1_mini)  */
  int32_t fpn_dim;                /* BiFPN width (128)                                 */
  int32_t fpn_layers;             /* BiFPN depth (3)                                   */
  /* Round 4 -- the encoder.  0: ResNet50 (encoders/resnet.py:143-215).  1: a four-stage RegNet
   * (encoders/regnet.py:38-160; quantization/encoders/__init__.py exports regnetx_6p4gf and regnety_6p4gf): 3x3
   * stride-2 stem of rn_stem channels, stage i = rn_depths[i] bottleneck blocks of width rn_widths[i] with
   * rn_groups[i] groups in the 3x3, the first block of a stage at stride rn_strides[i], rn_se != 0 with the reference's
   * per-pixel squeeze-excite gate (blocks.py:35-50).  A RegNet network starts, like every network since round 6, in the
   * fp16x3 mode (emp_pdl_precision reports 2: heads within 1e-3 of the reference's fp32 forward in the max norm;
   * emp_pdl_set_precision(net, 1) is the exact fp32 mode: 1e-4); emp_pdl_set_precision(net, 0) or EMP_PRECISION=fp16
   * put it on the fp16 engine -- generic implicit-GEMM convolutions, the grouped 3x3 one launch per group, 4-5x the
   * rate, heads within ~0.6e-3 .. 1.4e-3 (rms) of the fp32 forward. */
  int32_t encoder;
  int32_t rn_stem;
  int32_t rn_widths[4];
  int32_t rn_depths[4];
  int32_t rn_groups[4];
  int32_t rn_strides[4];
  int32_t rn_se;
} emp_pdl_config;

EMP_API int emp_pdl_create(const emp_pdl_config* cfg, emp_pdl_t** out);
EMP_API void emp_pdl_destroy(emp_pdl_t* net);

/* Folded fp32 parameter of one convolution, by the reference's module path
 * (e.g. "encoder.layer1.0.conv1"); h_w is OIHW (or (O,I,1) for the PointRend
 * Conv1d layers), h_b has shape[0] entries or is NULL (zero bias).  BatchNorm
 * is already folded by the host (weights.fold_state_dict). */
EMP_API int emp_pdl_set_param(emp_pdl_t* net, const char* name, const float* h_w,
                      const int64_t* shape, int ndim, const float* h_b);
/* Packs fp16 weights on the device; fails if a parameter is missing. */
EMP_API int emp_pdl_finalize(emp_pdl_t* net);
/* Precision of the forward, chosen BEFORE emp_pdl_finalize.  ROUND 6: the default is 2, the fp16x3 MODE -- the one that
 * meets the contract (float heads within 1e-3 of the reference's fp32 forward, max norm); 0, the fp16 ENGINE, is the explicit
 * throughput opt-in (emp_pdl_set_precision(net, 0) or EMP_PRECISION=fp16: 3x the rate, ~1e-3 rms / ~5e-3 max), and it is
 * what bench.py's headline `value` measures because BASELINE's metric is quoted in fp16.
 * Round 4 -- 0 the fp16 engine; 1 the fp32 REFERENCE
 * MODE (csrc/ref32.hip): every map and weight fp32, products on the exact fp32 matrix pipe (v_mfma_f32_32x32x2_f32), no
 * layer fusion.  The reference runs this path in fp32 (empanada/inference/engines.py:248-255: the model in eval, no
 * autocast); in this mode the float heads meet the north star's 1e-3 in the MAX norm (~1e-5 measured), at roughly a
 * tenth of the fp16 engine's rate -- it is the device-side comparator, not the bench.  Same entry points and output
 * tensors; emp_pdl_tap is not available (emp_pdl_tap_raw hands out every fp32 NHWC map of the last forward by name).
 * The environment variable EMP_PRECISION=fp32 selects it for networks that do not call this.
 * Round 5 (ABI version 3) -- 2: the fp16x3 MODE: the fp32 mode's graph, maps and weights, every convolution on the FP16
 * matrix pipe with both operands split into fp16 pairs and three MFMAs per product into an fp32 accumulator
 * (csrc/conv16x3.hip; weights split once at emp_pdl_finalize).  Heads within 1e-3 of the reference's fp32 forward in the
 * MAX norm on every tile and weight draw tested (2.4e-5 worst on the centre map), at ~2.5x the fp32 mode's rate: the mode
 * for results that must meet the tolerance as written, and since round 6 the default.  EMP_PRECISION=fp16 / fp32 / fp16x3
 * selects a mode for networks that do not call this. */
EMP_API int emp_pdl_set_precision(emp_pdl_t* net, int precision);
EMP_API int emp_pdl_precision(const emp_pdl_t* net);
/* Number of parameters the network expects, and the name of the i-th one. */
EMP_API int emp_pdl_num_params(const emp_pdl_t* net);
EMP_API const char* emp_pdl_param_name(const emp_pdl_t* net, int i);

/* (Re)allocates the activation arena for batches up to N x H x W.  Called
 * implicitly by emp_pdl_forward when the shape grows; call it explicitly
 * before capturing the forward into a hipGraph. */
EMP_API int emp_pdl_reserve(emp_pdl_t* net, int N, int H, int W);
EMP_API size_t emp_pdl_arena_bytes(const emp_pdl_t* net);

typedef enum { EMP_IMG_F32 = 0, EMP_IMG_U8 = 1, EMP_IMG_U16 = 2 } emp_image_dtype;

/* d_image: (N,H,W) single channel, H % 16 == 0 and W % 16 == 0
 *   EMP_IMG_F32: already normalised (Preprocessor + factor_pad output);
 *   EMP_IMG_U8/U16: raw tile, normalised in the stem as (x - sub) * mul
 *   (empanada_napari/utils.py:153-165).
 * Outputs (fp32, NCHW):
 *   d_sem_logits (N, C, H*2^(rs-2), W*2^(rs-2))
 *   d_ctr_hmp    (N, 1, h, w), d_offsets (N, 2, h, w) with h,w = H/4,W/4, or
 *   H,W when interpolate_ins != 0 (bilinear x4, align_corners=True).  */
EMP_API int emp_pdl_forward(emp_pdl_t* net, const void* d_image, int image_dtype,
                    float sub, float mul, int N, int H, int W,
                    int render_steps, int interpolate_ins,
                    float* d_sem_logits, float* d_ctr_hmp, float* d_offsets,
                    void* stream);

/* Same forward with `factor_pad` (postprocess.py:25-36) fused into the stem: d_image is the tight (N,vh,vw) image,
 * the network runs at the padded size H x W (multiples of 16, H >= vh, W >= vw) and pixels outside vh x vw are zero
 * AFTER normalisation, as in the reference.  Outputs have the padded size. */
EMP_API int emp_pdl_forward_padded(emp_pdl_t* net, const void* d_image, int image_dtype,
                    float sub, float mul, int N, int vh, int vw, int H, int W,
                    int render_steps, int interpolate_ins,
                    float* d_sem_logits, float* d_ctr_hmp, float* d_offsets,
                    void* stream);

/* Algorithmic FLOPs (2*MAC) of one forward at this shape: conv/GEMM work only. */
EMP_API double emp_pdl_flops(const emp_pdl_t* net, int N, int H, int W, int render_steps);

/* Live timing of the dominant kernel class (the 256x256 implicit-GEMM conv tile): while enabled every such
 * launch is bracketed by HIP events on the forward's stream; emp_pdl_profile_read waits for them, returns the summed
 * duration, the algorithmic FLOPs (2*MAC) and the number of launches since the last read, and resets.  Used by
 * bench.py for the roofline block; no reference counterpart. */
EMP_API int emp_pdl_profile(emp_pdl_t* net, int enable);
EMP_API int emp_pdl_profile_read(emp_pdl_t* net, double* ms_total, double* flops_total, int* launches);

/* Parity/debug taps: keep every intermediate activation addressable by name
 * after a forward ("stem", "encoder.layer1.0", ..., "semantic_x").  Returns
 * the device pointer (fp16 NHWC) and shape {N,H,W,C,ld}. */
EMP_API int emp_pdl_tap(emp_pdl_t* net, const char* name, void** d_ptr, int64_t shape5[5]);
EMP_API int emp_pdl_num_taps(const emp_pdl_t* net);
/* the fp32 buffers of the last forward that are not NHWC fp16 maps ("semantic_head.out": the coarse logits the
 * PointRend subdivision starts from, (N,C,H/4,W/4); "ins_center.out", "ins_xy.out" when interpolate_ins): device
 * pointer + size in bytes */
EMP_API int emp_pdl_tap_raw(emp_pdl_t* net, const char* name, void** d_ptr, int64_t* bytes);
/* device-to-device copy on `stream` (lets a host language without a HIP binding read a tap) */
EMP_API int emp_copy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream);
EMP_API const char* emp_pdl_tap_name(const emp_pdl_t* net, int i);

/* ------------------------------------------------------------------------
 * 2. Building-block operators (used by the network, exported for parity
 *    tests and for callers that run their own graph)
 * ---------------------------------------------------------------------- */

/* NHWC fp16 implicit-GEMM convolution on MFMA, fp32 accumulate, fused
 * bias + per-image bias + residual + ReLU.  replaces: nn.Conv2d(+folded
 * BatchNorm)(+ReLU)(+skip add), e.g. encoders/resnet.py:109-129.
 *   d_in  : (N,H,W,in_ld) fp16, the conv reads channels [0,Cin); Cin % 64 == 0
 *   d_w   : (Cout, KH*KW, Cin) fp16
 *   d_bias: (Cout) fp32 or NULL;  d_bias_n: (N,Cout) fp32 or NULL
 *   d_res : (N,Ho,Wo,res_ld) fp16 or NULL
 *   d_out : (N,Ho,Wo,out_ld) fp16; writes channels [0,Cout); Cout % 8 == 0
 *   variant: 0 = automatic.  Otherwise staging (bits 0-3: 1 registers | 2 LDS-DMA | 3 LDS-DMA + LDS-transposed epilogue)
 *     + 16 * tile (1 128x128 | 2 128x64 | 3 64x64 | 4 256x256 | 5 half tile 128x256 / 256x128 | 6 64->64 3x3 with
 *     register weights | 7 64x64 with a deep LDS-DMA ring, the batch-1 path) + 256 * K-walk group; every tile gives
 *     bit-identical results (same K order), the parity tests force each one.
 *     + (1 << 20): d_w is the 256x256 tile's PACKED weight image (emp_conv256_pack_weights, round 5) -- that tile, its
 *     default K walk, the same bits out. */
EMP_API int emp_conv2d_nhwc_f16(const void* d_in, int N, int H, int W, int Cin, int in_ld,
                        const void* d_w, const float* d_bias, const float* d_bias_n,
                        const void* d_res, int res_ld,
                        void* d_out, int out_ld, int Cout,
                        int KH, int KW, int stride, int pad, int dil, int relu,
                        int variant, void* stream);

/* Packed weight image for the 256x256 convolution tile (round 5; csrc/conv_igemm256.hip pack256_kernel): the network
 * (emp_pdl_*) makes one per layer at the first launch that takes the tile.  Every 1 KiB piece an LDS-DMA instruction moves
 * (16 cout rows x 32 channels of one K step) is contiguous in the image, in the kernel's K-walk order, so the instruction
 * fetches whole 128-byte lines.  replaces: nothing in the reference (nn.Conv2d weights, resnet.py:109-129, re-laid out).
 *   d_w      : (Cout, KT * Cin + Cin2) fp16 -- the weights of emp_conv2d_nhwc_f16 (KT = KH*KW; Cin2 = the K-concatenated
 *              second source's channels, 0 if none); Cout % 256 == 0, Cin % 32 == 0, Cin2 % 32 == 0
 *   d_packed : the same number of halfs */
EMP_API int emp_conv256_pack_weights(const void* d_w, void* d_packed, int Cout, int KT, int Cin, int Cin2, void* stream);

/* The convolution of the fp32 REFERENCE MODE (round 4, csrc/ref32.hip; emp_pdl_set_precision): NHWC fp32 maps, fp32
 * weights (Cout, KH*KW, Cin) with Cin % 16 == 0 (pad with zero weights), every product on the exact fp32 matrix pipe
 * (v_mfma_f32_32x32x2_f32).  replaces nn.Conv2d(+folded BatchNorm)(+ReLU / SiLU)(+skip add) exactly as the reference
 * computes it (fp32, engines.py:248-255).  act: 0 none, 1 ReLU, 2 SiLU; other arguments as emp_conv2d_nhwc_f16. */
EMP_API int emp_conv2d_nhwc_f32(const float* d_in, int N, int H, int W, int Cin, int in_ld,
                        const float* d_w, const float* d_bias, const float* d_bias_n,
                        const float* d_res, int res_ld,
                        float* d_out, int out_ld, int Cout,
                        int KH, int KW, int stride, int pad, int dil, int act, void* stream);

/* The same convolution in the `fp16x3` precision mode (round 5, csrc/conv16x3.hip; emp_pdl_set_precision(net, 2)): the
 * same fp32 operands, every operand split into an fp16 pair x = hi + lo on its way into LDS and every product computed as
 * w_lo.x_hi + w_hi.x_lo + w_hi.x_hi on the FP16 matrix pipe into an fp32 accumulator -- the reference's fp32 result
 * (engines.py:248-255) to ~2^-22 relative per term at three fp16 MFMAs per product instead of sixteen fp16-MFMA times on
 * the fp32 pipe.  groups > 1: Cout and Cin are per group (Cin = cin_g padded to 16), as emp_conv2d_grouped_nhwc_f32;
 * groups == 1: cin_g = 0.  Values must fit fp16's range. */
EMP_API int emp_conv2d_nhwc_f16x3(const float* d_in, int N, int H, int W, int Cin, int in_ld,
                        const float* d_w, const float* d_bias, const float* d_bias_n,
                        const float* d_res, int res_ld,
                        float* d_out, int out_ld, int Cout,
                        int KH, int KW, int stride, int pad, int dil, int act,
                        int groups, int cin_g, void* stream);

/* Round 6 (ABI version 5) -- every variant of the fp16x3 convolution that the network reaches and the entry above cannot
 * express, for op-level tests and tuning (replaces the same nn.Conv2d, resnet.py:109-129 / aspp.py:51-103 / heads.py:12-15):
 *   wmode    : 0 fp32 weights split in the kernel; 1 weights pre-split into fp16 pairs (what emp_pdl_finalize makes);
 *              2 pairs + the split-role kernel's LDS-DMA weight image (long K, Cin % 32 == 0);
 *              + 4: split-K allowed -- a long-K launch of fewer than 128 workgroups (few pixels: ONE 1024^2 tile's ASPP branch) runs
 *              S <= 8 workgroups per tile over K / S each, fp32 partial sums added in ascending order by a finish pass (what the
 *              network does at small batches; the result differs from the unsplit one by fp32 summation order only)
 *   d_in2 .. : optional second source K-concatenated behind the first -- a 1x1 convolution of d_in2 (N, H2, W2, in2_ld) at
 *              stride2 summed into the same accumulators (a bottleneck's projection shortcut folded into conv3); d_w rows are
 *              then [KH*KW*Cin | Cin2] long
 *   d_head_w : optional fused 1x1 head (heads.py:14): (head_c, Cout) fp32; the activation map is not stored, d_head_out
 *              receives (N, head_c, Ho*Wo) = head_w . relu(conv) + d_head_b; head_c <= 4, act must be 1
 *   out_fmt  : 0 fp32 rows; 1 `hl32` rows (below).
 * Allocates and frees its temporaries and synchronises the stream: not a hot-path call. */
EMP_API int emp_conv2d_nhwc_f16x3_ex(const float* d_in, int N, int H, int W, int Cin, int in_ld,
                        const float* d_w, const float* d_bias, const float* d_bias_n,
                        const float* d_res, int res_ld,
                        void* d_out, int out_ld, int out_fmt, int Cout,
                        int KH, int KW, int stride, int pad, int dil, int act, int wmode,
                        const float* d_in2, int H2, int W2, int Cin2, int in2_ld, int stride2,
                        const float* d_head_w, const float* d_head_b, int head_c, float* d_head_out,
                        void* stream);

/* Round 6 -- the `hl32` map format and the 256 x 256 fp16x3 convolution over it (csrc/conv16x3p.hip).  An hl32 map keeps a
 * value x of the fp32 graph as the fp16 pair the fp16x3 product needs -- hi = fp16(x), lo = fp16(x - hi), x ~ hi + lo to 22
 * bits -- split ONCE by the producer: a row of C channels (C % 32 == 0) is C / 32 blocks of 128 bytes, each 32 hi halfs
 * followed by 32 lo halfs; rows are 2 * ld halfs apart (the fp32 map's 4 bytes per element).  The networks of precision 2 keep
 * the stride-16 region (ResNet layer3 / layer4, ASPP: resnet.py:109-129, aspp.py:51-103) in this format, so that the 128 bytes a
 * pixel contributes to a K step of 32 channels are one cache line and one LDS row, fetched by LDS-DMA.
 *   emp_hl32_from_f32 / emp_hl32_to_f32 : (rows, C) fp32 rows of in_ld floats <-> hl32 rows (ld in channels)
 *   emp_x3p_pack_weights : (Cout, K) fp32, K = KH*KW*Cin walked tap-major, Cout % 256 == 0, K % 32 == 0 -> the kernel's packed
 *                          image of 2 * Cout * K halfs ([cout tile][K step][lo pieces | hi pieces], rows permuted and chunks
 *                          swizzled as they lie in LDS)
 *   emp_conv2d_hl32_f16x3 : the convolution; d_in hl32, d_wimg the packed image, res_fmt / out_fmt 0 fp32 rows or 1 hl32
 *                          rows; Cout % 256 == 0, Cin % 32 == 0, KH*KW*Cin >= 128; the same three fp16 MFMAs per product in the
 *                          same K order as emp_conv2d_nhwc_f16x3: bit-identical results on the same operands. */
EMP_API int emp_hl32_from_f32(const float* d_in, void* d_out, int64_t rows, int C, int in_ld, int out_ld, void* stream);
EMP_API int emp_hl32_to_f32(const void* d_in, float* d_out, int64_t rows, int C, int in_ld, int out_ld, void* stream);
EMP_API int emp_x3p_pack_weights(const float* d_w, void* d_img, int Cout, int K, void* stream);
EMP_API int emp_conv2d_hl32_f16x3(const void* d_in, int N, int H, int W, int Cin, int in_ld,
                        const void* d_wimg, const float* d_bias, const float* d_bias_n,
                        const void* d_res, int res_ld, int res_fmt,
                        void* d_out, int out_ld, int out_fmt, int Cout,
                        int KH, int KW, int stride, int pad, int dil, int act, void* stream);
/* The same convolution with split-K allowed (what the network does below the plane region's batch threshold: the merged 3x3 ASPP
 * branches of ONE 1024^2 tile are 32 workgroups over 576 K steps): with d_scratch of scratch_bytes (S * M * Cout * 4 needed; 64 MiB
 * covers every launch the rule splits) a launch of fewer than 128 workgroups runs S <= 8 workgroups per tile over K / S each (raw fp32
 * partial sums), and a finish pass adds them in ascending order, then bias / bias_n / residual / activation -- the unsplit result up to
 * fp32 summation order.  d_out2 (optional): couts [split2, Cout) go there (split2 % 256 == 0), as the network's merged launches.
 * Launches the rule does not split run exactly as emp_conv2d_hl32_f16x3.  No allocation, no synchronisation. */
EMP_API int emp_conv2d_hl32_f16x3_ksplit(const void* d_in, int N, int H, int W, int Cin, int in_ld,
                        const void* d_wimg, const float* d_bias, const float* d_bias_n,
                        const void* d_res, int res_ld, int res_fmt,
                        void* d_out, int out_ld, int out_fmt, void* d_out2, int out2_ld, int split2, int Cout,
                        int KH, int KW, int stride, int pad, int dil, int act,
                        void* d_scratch, int64_t scratch_bytes, void* stream);

/* Round 6 -- the depthwise-separable block of the fp16x3 mode as ONE launch (csrc/sepconv_x3.hip): replaces
 * nn.Conv2d(C,C,k,groups=C,padding=k/2,bias=False) -> nn.Conv2d(C,Cout,1) -> folded BatchNorm -> ReLU / SiLU (models/blocks.py:15-33;
 * decoders/panoptic_deeplab.py:68-80, heads.py:12-15) and, with d_head_w, the head's final 1x1 (heads.py:14) on fp32 NHWC maps.
 * The depthwise half runs in fp32 (taps ky-major / kx-minor, an fmaf chain from 0); its result goes to LDS as the fp16 pair
 * hi + lo and meets the pointwise weights (hi + lo) in three fp16 MFMAs per product, fp32 accumulate from the bias.
 *   emp_sepconv_x3_pack : d_dw (k*k, C) fp32 taps, d_pw (Cout, C) fp32 -> d_dw_packed (k*k*C floats), d_pw_packed (2*C*Cout halfs)
 *   emp_sepconv_x3_nhwc_f32 : C % 32 == 0, 64 <= C <= 448, Cout 128 or 256, k 5 (or 3 without a head), act 0 / 1 / 2;
 *                          exactly one of d_out (N,H,W,out_ld) and the head output d_head_out (N, head_c, H*W), head_c <= 2 */
EMP_API int emp_sepconv_x3_pack(const float* d_dw, const float* d_pw, int ks, int C, int Cout,
                        float* d_dw_packed, void* d_pw_packed, void* stream);
EMP_API int emp_sepconv_x3_nhwc_f32(const float* d_in, int N, int H, int W, int C, int in_ld,
                        const float* d_dw_packed, const void* d_pw_packed, const float* d_bias, int Cout, int act,
                        float* d_out, int out_ld,
                        const float* d_head_w, const float* d_head_b, int head_c, float* d_head_out,
                        int ks, void* stream);

/* The same convolution with `groups` groups (nn.Conv2d(groups=g), the 3x3 of a RegNet bottleneck:
 * empanada/models/encoders/regnet.py:51-77 via blocks.py:134-153): group g reads input channels [g * cin_g, g * cin_g +
 * Cin16) -- Cin16 = cin_g padded to a multiple of 16, the padding meeting zero weights, so a row of the input must hold
 * (groups - 1) * cin_g + Cin16 floats -- and writes output channels [g * cout_g, (g + 1) * cout_g); weights
 * (groups * cout_g, KH*KW, Cin16), bias (groups * cout_g). */
EMP_API int emp_conv2d_grouped_nhwc_f32(const float* d_in, int N, int H, int W, int groups, int cin_g, int Cin16, int in_ld,
                        const float* d_w, const float* d_bias, float* d_out, int out_ld, int cout_g,
                        int KH, int KW, int stride, int pad, int dil, int act, void* stream);

/* out = act( in . W[:, :Cin] + in2(strided) . W[:, Cin:] + bias ): a 1x1 convolution whose reduction continues over a
 * second tensor sampled with stride2.  replaces the tail of a ResNet bottleneck with a projection shortcut,
 * `out = relu(bn3(conv3(x)) + downsample(identity))`, empanada/models/encoders/resnet.py:109-129 with
 * downsample = Conv2d(1x1, stride) + BN (:176-181): both branches accumulate in fp32 in one launch, the shortcut map is
 * never written.  d_w (Cout, Cin + Cin2) fp16, rows K-contiguous; in (N,H,W,in_ld), in2 (N,H2,W2,in2_ld) with
 * (H-1)*stride2 < H2; Cin, Cin2 multiples of 64. */
EMP_API int emp_conv1x1_dual_nhwc_f16(const void* d_in, int N, int H, int W, int Cin, int in_ld, const void* d_in2,
                              int H2, int W2, int Cin2, int in2_ld, int stride2, const void* d_w,
                              const float* d_bias, void* d_out, int out_ld, int Cout, int relu, int variant,
                              void* stream);

/* NHWC fp16 depthwise KxK convolution (K in {3,5}, stride 1, pad K/2, no bias), fp32 accumulate.
 * replaces nn.Conv2d(C, C, K, groups=C, bias=False): the first half of the decoder's separable convs
 * (models/blocks.py separable conv, models/decoders/bifpn.py SeparableConv2d).
 *   d_w: (K*K, C) fp16;  C % 64 == 0 */
EMP_API int emp_dwconv_nhwc_f16(const void* d_in, int N, int H, int W, int C, int in_ld,
                        const void* d_w, int K, void* d_out, int out_ld, void* stream);

/* Fused separable convolution: y = act(pointwise(depthwise5x5(x)) + bias), and optionally the
 * 1x1 head on top of it (head_c in 1..4) so that y never reaches HBM.
 * replaces the 'depthwise_separable_conv' blocks of the Panoptic-DeepLab decoder / heads
 * (models/decoders/panoptic_deeplab.py fuse convs, models/heads/panoptic_deeplab.py head.0 -> head.1).
 *   d_in    : (N,H,W,in_ld) fp16, channels [0,C), C % 64 == 0, 128 <= C <= 512
 *   d_dw_w  : (25, C) fp16;  d_bias: (Cout) fp32 or NULL;  Cout in {128,256}
 *   d_pw_w  : the (Cout, C) pointwise weights re-ordered by emp_sepconv5x5_pack_pw (MFMA fragment order, C*Cout fp16)
 *   act     : 0 none | 1 ReLU | 2 SiLU
 *   head_c == 0: d_out (N,H,W,out_ld) fp16 receives y
 *   head_c  > 0: d_head_out (N,head_c,H,W) fp32 receives d_head_w (head_c,Cout) . y + d_head_b; d_out unused */
EMP_API int emp_sepconv5x5_pack_pw(const void* d_pw_w, int pw_ld, int C, int Cout, void* d_packed, void* stream);
EMP_API int emp_sepconv5x5_nhwc_f16(const void* d_in, int N, int H, int W, int C, int in_ld,
                        const void* d_dw_w, const void* d_pw_w, const float* d_bias,
                        int Cout, int act, void* d_out, int out_ld,
                        const float* d_head_w, const float* d_head_b, int head_c, float* d_head_out,
                        void* stream);
/* The same with a 3x3 depthwise kernel (d_dw_w [9][C]) and no head mode: depthwise 3x3 -> pointwise -> BN -> SiLU, the
 * `SeparableConv2d` after every BiFPN fusion node, empanada/models/decoders/bifpn.py:24-45 as used at :63-69,128-134. */
EMP_API int emp_sepconv3x3_nhwc_f16(const void* d_in, int N, int H, int W, int C, int in_ld, const void* d_dw_w,
                            const void* d_pw_w, const float* d_bias, int Cout, int act, void* d_out, int out_ld,
                            void* stream);

/* The separable block with an exact depthwise half (round 3; csrc/sepconv_precise.hip): where the block above rounds
 * to fp16 inside -- depthwise taps, depthwise result, pointwise weights -- was a large part of the centre heat-map's
 * distance to the fp32 reference forward (models/quantization/panoptic_deeplab.py:238-250 run in fp32).  Here the taps
 * stay fp32 on the vector pipe and the depthwise result goes to the matrix pipe as an fp16 hi + lo pair (two MFMAs per
 * product, fp32 accumulation); the pointwise weights stay fp16 (their hi + lo form bought 0.5 % of the error for a third
 * more matrix work: measured, removed).  x and y are fp16.
 *   K       : depthwise kernel size, 5 or 3 (3: no head mode)
 *   d_dw_w  : the (K*K, C) fp32 taps re-ordered by emp_sepconvp_pack_dw (chunk-major [C/64][K*K][64] fp32)
 *   d_pw_w  : the (Cout, C) fp32 pointwise weights (row stride pw_ld) rounded to fp16 and re-ordered by
 *             emp_sepconvp_pack_pw (C*Cout fp16, MFMA fragment order)
 *   C % 64 == 0, 128 <= C <= 512, Cout in {128,256}, head_c in 0..2; other arguments as emp_sepconv5x5_nhwc_f16 */
EMP_API int emp_sepconvp_pack_dw(const void* d_dw_w, int K, int C, void* d_packed, void* stream);
EMP_API int emp_sepconvp_pack_pw(const void* d_pw_w, int pw_ld, int C, int Cout, void* d_packed, void* stream);
EMP_API int emp_sepconvp_nhwc_f16(const void* d_in, int N, int H, int W, int C, int in_ld, int K,
                        const void* d_dw_w, const void* d_pw_w, const float* d_bias,
                        int Cout, int act, void* d_out, int out_ld,
                        const float* d_head_w, const float* d_head_b, int head_c, float* d_head_out,
                        void* stream);
/* Round 4 -- the same block with the POINTWISE WEIGHTS as an fp16 hi + lo pair too (a third MFMA per product, w_lo * x_hi;
 * Cout == 128 only): the 3x3 node blocks, the decoder's fusion conv and the centre head of PanopticBiFPNPR
 * (models/quantization/panoptic_bifpn.py:147-161, decoders/bifpn.py:35-134), whose pointwise weight roundings were 65 %
 * of the weight-side error of the centre heat-map (tools/error_budget.py --arch bifpn).  emp_sepconvp_ws_pack_pw writes
 * 2*C*Cout fp16 (the hi fragments, then the lo fragments); every other argument as above. */
EMP_API int emp_sepconvp_ws_pack_pw(const void* d_pw_w, int pw_ld, int C, int Cout, void* d_packed, void* stream);
EMP_API int emp_sepconvp_ws_nhwc_f16(const void* d_in, int N, int H, int W, int C, int in_ld, int K,
                        const void* d_dw_w, const void* d_pw_w, const float* d_bias,
                        int Cout, int act, void* d_out, int out_ld,
                        const float* d_head_w, const float* d_head_b, int head_c, float* d_head_out,
                        void* stream);

/* ------------------------------------------------------------------------
 * 3. Instance post-processing (hot loop 2), one launch group per batch
 * ---------------------------------------------------------------------- */

/* logits -> probabilities: sigmoid (C==1) or channel softmax (C>1).
 * replaces logits_to_prob, engines.py:22-30.  (N,C,H,W) fp32 in/out. */
EMP_API int emp_logits_to_prob(const float* d_logits, float* d_prob, int N, int C, int H, int W, void* stream);

/* Recursive per-pixel median over a ks-deep queue of probability maps.
 * replaces _MedianQueue.get_median, engines.py:59-66 (torch.cat + torch.median).
 * d_slices: array of ks device pointers (host array), each (C,H,W) fp32;
 * d_out may alias d_slices[mid] (that is what the reference does, :76-84). */
EMP_API int emp_median_slices(const float* const* h_slice_ptrs, int ks, float* d_out,
                      size_t count, void* stream);

/* The same filter over a run of consecutive slices in one launch (batched 3-D path): the recursion of
 * _MedianQueue.get_next (engines.py:76-84) is per pixel, so one thread carries the filtered history.
 *   d_hist (mid,count) filtered maps preceding the run; d_raw (n_raw,count) raw maps of the run plus
 *   mid look-ahead maps; d_out (n_out,count): out[j] = median(filtered[j-mid..j-1], raw[j..j+mid]).
 *   ks odd, 3..15 (the widget offers up to 11, _volume_inference.py:388).  d_out may be d_raw (in place). */
EMP_API int emp_median_recursive(const float* d_hist, const float* d_raw, int n_raw, int ks, int n_out,
                         float* d_out, size_t count, void* stream);

/* Centre NMS + voting + nearest upsampling: get_instance_cells,
 * engines.py:257-275 (find_instance_center postprocess.py:38-76 and
 * group_pixels :78-169).  Batched over N images.
 *   d_ctr_hmp (N,1,h,w) fp32, d_offsets (N,2,h,w) fp32
 *   step: 4 (coarse boundaries) or 1;  up = upsampling*step (nearest factor)
 *   d_cells (N, h*up, w*up) int32: 0 where the image has no centre
 *   d_centers (N, max_centers, 2) int32 (y,x) in row-major order,
 *   d_num_centers (N) int32 (clamped to max_centers; overflow is reported by
 *   a negative return of emp_instance_cells_check).
 *   d_work: scratch of emp_instance_cells_work_bytes(N,h,w) bytes (the NMS bit mask + a per-image grid over the centres).
 * Round 6: an image with 192 .. 16 384 centres is voted through a uniform grid over its centres (bins of ~2+ centres, rings of bins
 * searched outwards from the voted position until the next ring cannot hold a centre at the best distance found) instead of the
 * scan over every centre per pixel: the scan's cells bit for bit (same fp32 expression per candidate, lowest index among the minima
 * of the rounded distance, 1e5 start value, non-finite votes -> 0) at pixels x ~tens instead of pixels x centres evaluations
 * (1024^2 tile, 4 500 centres: 1.56 -> 0.15 ms).  EMP_VOTE_GRID=0 (read per call): the scan for every image. */
EMP_API size_t emp_instance_cells_work_bytes(int N, int h, int w);
EMP_API int emp_instance_cells(const float* d_ctr_hmp, const float* d_offsets, int N, int h, int w,
                       float nms_threshold, int nms_kernel, int step, int up,
                       int32_t* d_cells, int32_t* d_centers, int32_t* d_num_centers,
                       int max_centers, void* d_work, void* stream);

/* harden + thing-mask + majority vote + per-class renumbering + stuff filter:
 * PanopticDeepLabRenderEngine.postprocess, engines.py:277-298 with
 * merge_semantic_and_instance, postprocess.py:223-296.  Batched over N.
 *   d_sem (N,C,H,W) fp32 probabilities; d_cells (N,H,W) int32
 *   thing_list: host array of n_things class ids
 *   d_pan (N,H,W) int64 (reference dtype) ; max_ids: upper bound on cell ids
 *   d_work: scratch of emp_panoptic_merge_work_bytes(N, C, max_ids) bytes */
EMP_API size_t emp_panoptic_merge_work_bytes(int N, int C, int max_ids);
EMP_API int emp_panoptic_merge(const float* d_sem, const int32_t* d_cells, int N, int C, int H, int W,
                       float confidence_thr, const int32_t* h_thing_list, int n_things,
                       int64_t label_divisor, int64_t stuff_area, int64_t void_label,
                       int max_ids, int64_t* d_pan, void* d_work, void* stream);

/* ------------------------------------------------------------------------
 * 4. Label map <-> run-length encoding (3-D stitching path)
 * ---------------------------------------------------------------------- */

/* 8-connected components of EQUAL non-zero label, numbered 1..K per image in raster order of each
 * component's first pixel.  replaces rle.connected_components (skimage.measure.label / cc3d),
 * empanada/inference/rle.py:18-24, used by pan_seg_to_rle_seg (:66-69) and
 * Engine2d.force_connected (empanada_napari/inference.py:263-279).
 *   d_in, d_out (N,H,W) int32; d_num (N) int32 components per image (may be NULL). */
EMP_API size_t emp_ccl8_work_bytes(int N, int H, int W);
EMP_API int emp_ccl8(const int32_t* d_in, int N, int H, int W, int32_t* d_out, int32_t* d_num,
             void* d_work, void* stream);

/* Engine2d.force_connected, empanada_napari/inference.py:263-279, for a batch of label maps in one call: for each
 * thing class in list order, the ids in [class*divisor, (class+1)*divisor) are replaced by their 8-connected
 * components numbered in raster order + class*divisor (later classes see the earlier classes' relabelling, as the
 * reference edits pan_seg in place).  d_pan (N,H,W) int64 (the Render engine's output); d_out (N,H,W) int32 (what
 * Engine2d.infer returns after .astype(np.int32), inference.py:325); h_thing_list is a HOST array. */
EMP_API size_t emp_force_connected_work_bytes(int N, int H, int W);
EMP_API int emp_force_connected(const int64_t* d_pan, int N, int H, int W, const int32_t* h_thing_list, int n_things,
                        int64_t label_divisor, int32_t* d_out, void* d_work, void* stream);

/* 26-connected components of EQUAL non-zero label of ONE volume (skimage.measure.label on a 3-D array, default
 * full connectivity), numbered in raster order of first voxels.  replaces filters.connected_components,
 * empanada/inference/filters.py:14-20, as used by filters.pan_seg_to_rle_seg (:58-116) after erode / dilate /
 * fill_holes.  d_in, d_out (D,H,W) int32, D*H*W < 2^30; work buffer: emp_ccl8_work_bytes(1, D*H, W). */
EMP_API int emp_ccl26(const int32_t* d_in, int D, int H, int W, int32_t* d_out, int32_t* d_num,
              void* d_work, void* stream);

/* One grey erosion (op 0) / dilation (op 1) of a uint32 label volume with the 3-D cross footprint, border mode
 * 'reflect'.  replaces skimage.morphology.erosion / dilation as called by filters.erode / filters.dilate,
 * empanada/inference/filters.py:154-176 (one call per iteration; d_out != d_in). */
EMP_API int emp_morph_cross3d(const uint32_t* d_in, uint32_t* d_out, int D, int H, int W, int op, void* stream);

/* (host) filters.fill_holes_in_segmentation, empanada/inference/filters.py:178-210, on a host uint32 label volume in
 * place: per slice, per label in ascending order, the label's original bounding box is cut out of the current
 * slice and scipy.ndimage.binary_fill_holes of the cut-out (any label counts as foreground) is written back as
 * that label.  Slices run on threads. */
EMP_API int emp_fill_holes_slices(uint32_t* h_vol, int64_t D, int64_t H, int64_t W);

/* Runs of equal non-zero label over the raveled image, in raster order.  replaces
 * regionprops(...).coords -> rle_encode, rle.py:73-81 + array_utils.py:213-239.
 *   d_runs (N, max_runs, 3) int32 {start, length, label}; d_num_runs (N) int32 (un-clamped). */
EMP_API size_t emp_rle_extract_work_bytes(int N, int H, int W);
EMP_API int emp_rle_extract(const int32_t* d_labels, int N, int H, int W, int32_t* d_runs,
                    int32_t* d_num_runs, int max_runs, void* d_work, void* stream);

/* Round 6 -- emp_ccl8 / emp_ccl26 and emp_rle_extract with the class isolation folded into the kernels' reads: the labels of
 * [lo, hi) of an int32 (in_bytes 4) or int64 (in_bytes 8) panoptic map are kept, everything else reads as background --
 * pan_seg_to_rle_seg's `instance_seg[outside the class range] = 0` (empanada/inference/rle.py:46-48) and force_connected's
 * (empanada_napari/inference.py:270-272) without a select pass (and an int64 -> int32 pass) per class in front of the kernels.
 *   emp_ccl_range : depth 0: N images of H x W, 8-connected (work: emp_ccl8_work_bytes(N, H, W)); depth > 0 (N == 1): one
 *                   volume depth x H x W, 26-connected (work: emp_ccl8_work_bytes(1, depth * H, W)); d_num may be NULL
 *   emp_rle_extract_range : as emp_rle_extract; the run labels are the map's own values (hi < 2^31) */
EMP_API int emp_ccl_range(const void* d_in, int in_bytes, int N, int depth, int H, int W, int64_t lo, int64_t hi,
                        int32_t* d_out, int32_t* d_num, void* d_work, void* stream);
EMP_API int emp_rle_extract_range(const void* d_labels, int in_bytes, int N, int H, int W, int64_t lo, int64_t hi,
                        int32_t* d_runs, int32_t* d_num_runs, int max_runs, void* d_work, void* stream);

/* Run list -> dense volume of elem_bytes-wide integers.  replaces numpy_fill_instances,
 * array_utils.py:754-766 (runs must not overlap: later-overwrites-earlier is not defined here). */
EMP_API int emp_rle_fill(const int64_t* d_starts, const int64_t* d_lens, const int64_t* d_vals,
                 int64_t nruns, void* d_volume, int64_t size, int elem_bytes, void* stream);

/* The same with the reference's overwrite order for overlapping objects: where runs overlap, the run with the highest
 * d_order (position of its instance in the dict) wins, as numpy_fill_instances' sequential fill does.  d_prio: scratch
 * of `size` int32, initialised by the call. */
EMP_API int emp_rle_fill_ordered(const int64_t* d_starts, const int64_t* d_lens, const int64_t* d_vals,
                 const int32_t* d_order, int64_t nruns, void* d_volume, int64_t size, int elem_bytes,
                 int32_t* d_prio, void* stream);

/* HOST: intersections of pairs of run-length objects stored CSR-style (object k owns runs
 * [h_off[k], h_off[k+1])).  replaces rle_intersection, array_utils.py:344-407. */
EMP_API int emp_rle_pair_intersections(const int64_t* h_starts, const int64_t* h_runs, const int64_t* h_off,
                               const int64_t* h_pairs, int64_t n_pairs, int64_t* h_out);
/* HOST: k-of-n vote over (n,2) ranges -> maximal ranges covered >= thr times (thr 1 = union).
 * replaces vote_by_ranges / rle_voting / join_ranges, array_utils.py:461-699. */
EMP_API int emp_ranges_vote(const int64_t* h_ranges, int64_t n, int thr, int64_t* h_out, int64_t* n_out);
/* HOST: copies n_seg segments of two parallel int64 arrays (run starts, run lengths): segment j = h_cnt[j] elements of
 * source array h_src_id[j] from element h_src_off[j] on, to h_out_* + h_out_off[j]; worker threads.  The concatenation of
 * the per-slab partial trackers into one tracker on rank 0 (empanada_napari/multigpu.py:240-252 builds that tracker by
 * feeding every slice to InstanceTracker.update in one process, tracker.py:61-100). */
EMP_API int emp_gather_segments_i64(const int64_t* const* h_src_a, const int64_t* const* h_src_b, const int32_t* h_src_id,
                            const int64_t* h_src_off, const int64_t* h_cnt, const int64_t* h_out_off, int64_t n_seg,
                            int64_t* h_out_a, int64_t* h_out_b);

/* ------------------------------------------------------------------------
 * 5. HOST: slice-to-slice matching + instance tracking of ONE class over a
 *    stack of slices.  replaces RLEMatcher / rle_matcher (matcher.py:136-326),
 *    forward_matching / backward_matching / apply_matchers (patterns.py:55-121)
 *    and InstanceTracker.update / finish (tracker.py:61-123).  The assignment on
 *    the IoU matrix stays with the caller (scipy.optimize.linear_sum_assignment,
 *    matcher.py:218):  step_begin -> [assignment] -> step_apply, per slice.
 * ---------------------------------------------------------------------- */
typedef struct emp_stack_matcher emp_stack_matcher_t;
EMP_API emp_stack_matcher_t* emp_sm_create(int64_t class_id, int64_t label_divisor, double iou_thr, double ioa_thr,
                                   int do_match);
EMP_API void emp_sm_destroy(emp_stack_matcher_t* h);
/* append a slice: (n,3) {start, length, label} runs in raster order (emp_rle_extract output), plane width, id offset */
EMP_API int emp_sm_push_slice_runs(emp_stack_matcher_t* h, const int64_t* h_runs, int64_t n, int64_t width, int64_t id_offset);
/* `count` slices at once (one launch group of the run extractor), built on the library's worker threads (EMP_SM_THREADS,
 * default 4; emp_sm_prepare uses the same threads for the pair tables) */
EMP_API int emp_sm_push_slices_runs(emp_stack_matcher_t* h, int64_t count, const int64_t* const* runs, const int64_t* n,
                                    int64_t width, int64_t id_offset);
/* append a slice as objects: labels (n), boxes (n,4), CSR offsets (n+1), starts, runs */
EMP_API int emp_sm_push_slice_objects(emp_stack_matcher_t* h, int64_t n, const int64_t* labels, const int64_t* boxes,
                              const int64_t* off, const int64_t* starts, const int64_t* runs);
EMP_API int64_t emp_sm_num_slices(const emp_stack_matcher_t* h);
/* Slab-wise matching (one matcher per rank: multigpu.py).  The forward / backward passes of patterns.py:68-121 are a chain
 * along the axis, but what a slice hands to its neighbour is small: the grouping of its components into labelled objects
 * (dict order) and RLEMatcher.next_label (matcher.py:254-268).  A rank pushes its own slab plus the neighbours' boundary
 * slices ("ghosts": same runs, hence same component indices), builds all pair tables up front with emp_sm_prepare (the
 * label-independent bulk: run intersections of neighbouring slices), imports the state of a ghost slice as its target and
 * runs emp_sm_run over its own slices only; emp_sm_track takes the GLOBAL slice position.
 *   emp_sm_prepare     : pair tables of local slices (from, to]
 *   emp_sm_state_size / emp_sm_export_state : objects of local slice idx as labels[n_obj], CSR off[n_obj+1], members[n_mem]
 *                        (component indices), and the label counter
 *   emp_sm_import_state: slice idx gets these objects and becomes the target; next_label < 0 keeps the counter;
 *                        assign_new 1 = forward pass, 0 = backward pass (patterns.py:102-109) */
/* scipy.optimize.linear_sum_assignment(cost, maximize=True), as matcher.py:218 calls it, on a dense row-major (nr, nc)
 * float64 matrix: min(nr, nc) pairs with ascending rows.  The library's own solver (scipy's shortest-augmenting-path
 * algorithm restated, ties included); emp_sm_run uses it for the assignment blocks unless EMP_SM_SCIPY=1. */
EMP_API int emp_lsa_maximize(const double* cost, int64_t nr, int64_t nc, int64_t* rows, int64_t* cols);
/* the same assignment from the matrix's non-zero entries only (er[k], ec[k]) -> ew[k] (an IoU matrix is almost all
 * zeros): the algorithm, its floating-point values and its tie-breaking are those of the dense form, the per-step scan of
 * all columns is replaced by a segment tree -- what the slice matcher calls (csrc/matcher.hip lsa_maximize_sparse) */
EMP_API int emp_lsa_maximize_sparse(int64_t nr, int64_t nc, int64_t nnz, const int64_t* er, const int64_t* ec, const double* ew,
                                    int64_t* rows, int64_t* cols);
EMP_API int emp_sm_prepare(emp_stack_matcher_t* h, int64_t from, int64_t to);
EMP_API int emp_sm_state_size(const emp_stack_matcher_t* h, int64_t idx, int64_t* n_obj, int64_t* n_mem);
EMP_API int emp_sm_export_state(const emp_stack_matcher_t* h, int64_t idx, int64_t* labels, int64_t* off, int64_t* members,
                                int64_t* next_label);
EMP_API int emp_sm_import_state(emp_stack_matcher_t* h, int64_t idx, int64_t n_obj, const int64_t* labels, const int64_t* off,
                                const int64_t* members, int64_t next_label, int assign_new);
EMP_API int emp_sm_begin_backward(emp_stack_matcher_t* h);
/* First half of RLEMatcher.__call__ (matcher.py:280-326) for slice idx.  nt / nm: number of target / slice objects;
 * nt < 0: nothing to assign (target initialised / class not matched).  The overlap matrix is held sparse (candidates by a
 * 64-pixel grid over the boxes, exact run intersections); the connected components of its graph that are a single
 * (target, object) pair are assigned by the library -- any optimal assignment contains them, every other entry of their
 * row and column is 0 and zero-IoU pairs never pass the threshold (matcher.py:226-229) -- and the rest forms the SOLVER
 * BLOCK: emp_sm_pending_shape gives its shape (0 x 0: nothing to solve), emp_sm_iou its dense float64 IoU matrix (rows /
 * columns in ascending original order).  The caller runs scipy.optimize.linear_sum_assignment(maximize=True) on the block,
 * as the reference does on the whole matrix (matcher.py:218), and passes the result, in BLOCK coordinates, to
 * emp_sm_step_apply (n = 0 when the block is empty). */
EMP_API int emp_sm_step_begin(emp_stack_matcher_t* h, int64_t idx, int* nt, int* nm);
EMP_API const double* emp_sm_iou(const emp_stack_matcher_t* h);          /* solver block, valid until the next step */
EMP_API int emp_sm_step_apply(emp_stack_matcher_t* h, const int64_t* rows, const int64_t* cols, int64_t n);
/* Runs `count` steps from slice idx in direction dir (+1 / -1), feeding the tracker after each when `track`, and stops
 * before the first slice with a non-empty solver block (*stopped_at = its index, step pending: solve emp_sm_iou with
 * linear_sum_assignment, emp_sm_step_apply, emp_sm_track, resume), or runs to the end (*stopped_at = -1).
 * Replaces the per-slice Python loop of forward_matching / backward_matching, patterns.py:68-121. */
EMP_API int emp_sm_run(emp_stack_matcher_t* h, int64_t idx, int dir, int64_t count, int track, int64_t* stopped_at);
/* shape of the pending step's solver block */
/* steps with competing overlaps solved so far by the sparse solver (default) / by a dense solver call on the IoU matrix
 * (EMP_SM_FULL_LSA=1 the whole matrix as the reference hands it to scipy, empanada/inference/matcher.py:216-218; =0 the conflict block) */
EMP_API int emp_sm_solver_stats(const emp_stack_matcher_t* h, int64_t* sparse_steps, int64_t* dense_steps);
EMP_API int emp_sm_pending_shape(const emp_stack_matcher_t* h, int* nt, int* nm);
EMP_API int emp_sm_tracker_init(emp_stack_matcher_t* h, int axis /* 0 xy, 1 xz, 2 yz */, int64_t D, int64_t H, int64_t W);
EMP_API int emp_sm_track(emp_stack_matcher_t* h, int64_t idx, int64_t index2d);
/* InstanceTracker.update over local slices last, last-1, ..., first at positions global_first + (i - first) (the order
 * backward_matching feeds the tracker: empanada/inference/patterns.py:102-134, tracker.py:61-97), every track sized once. */
EMP_API int emp_sm_track_range(emp_stack_matcher_t* h, int64_t first, int64_t last, int64_t global_first);
EMP_API int emp_sm_tracker_finish(emp_stack_matcher_t* h);
EMP_API int64_t emp_sm_num_tracks(const emp_stack_matcher_t* h);
EMP_API int emp_sm_track_info(const emp_stack_matcher_t* h, int64_t k, int64_t* label, int64_t* box6, int64_t* n_runs);
EMP_API int emp_sm_track_runs(const emp_stack_matcher_t* h, int64_t k, int64_t* starts, int64_t* runs);
/* the same for ALL tracks in one call each (tracker.instances as flat arrays: labels (T), boxes (T,6), run counts (T) --
 * any of the three may be NULL -- and the run lists back to back in track order) */
EMP_API int emp_sm_tracks_info(const emp_stack_matcher_t* h, int64_t* labels, int64_t* boxes6, int64_t* counts, int64_t* total_runs);
EMP_API int emp_sm_tracks_runs(const emp_stack_matcher_t* h, int64_t* starts, int64_t* runs);
EMP_API int64_t emp_sm_slice_num_objects(const emp_stack_matcher_t* h, int64_t idx);
EMP_API int emp_sm_slice_object_info(const emp_stack_matcher_t* h, int64_t idx, int64_t k, int64_t* label, int64_t* box4,
                             int64_t* n_runs);
EMP_API int emp_sm_slice_object_runs(const emp_stack_matcher_t* h, int64_t idx, int64_t k, int64_t* starts, int64_t* runs);

#ifdef __cplusplus
}
#endif
#endif /* EMPANADA_HIP_H */
