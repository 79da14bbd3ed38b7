/* studiosr_hip.h -- C ABI of libstudiosr_hip.so (MI355X / gfx950 hot path of veritross/studiosr).
 *
 * The reference (pure Python) has no FFI; its hot path is the closed set of ATen ops reached from
 * studiosr/models/{common,swinir,hat,edsr,rcan}.py.  Each entry point below
 * replaces one group of those ops and cites the reference lines it stands in for.  All pointers are
 * DEVICE pointers on the current HIP device; `stream` is a hipStream_t passed as void*; functions
 * enqueue work only (no allocation, no synchronisation, graph-capture safe) and return 0 on success
 * or a negative SR_E* code (sr_last_error() gives the text).  Layout everywhere: NHWC with the
 * channel count padded to a multiple of 32 (pad lanes hold zeros), "T" = bf16 or fp32.
 *
 * Packed weights ("Wp"): fragment order [n_tile][k_chunk][lane 0..63][8] of T, element
 *   (n = 16*n_tile + (lane & 15), k = 32*k_chunk + 8*(lane >> 4) + j); for 3x3 convs
 *   k = (ky*3 + kx) * Cin_p + c.  studiosr_amd/packing.py builds them from a state_dict.
 */
#ifndef STUDIOSR_HIP_H
#define STUDIOSR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SR_ABI_VERSION 11

enum { SR_OK = 0, SR_EINVAL = -1, SR_ELAUNCH = -2, SR_EUNSUPPORTED = -3 };
enum { SR_F32 = 0, SR_BF16 = 1 };                                 /* element types */
/* compute_dtype only (sr_gemm, sr_conv3x3): fp32 tensors, every contraction as hi*hi + hi*lo + lo*hi on the bf16 matrix cores
 * (split operands, fp32 accumulate): fp32-class accuracy at 3 bf16 MFMAs per K-chunk.  Packed weights: per lane 8 hi then 8 lo. */
enum { SR_BF16X3 = 2 };
enum { SR_ACT_NONE_ = 0, SR_ACT_RELU_ = 1, SR_ACT_LRELU_ = 2, SR_ACT_GELU_ = 3 };
enum { SR_PAD_NONE = 0, SR_PAD_EVAL_MIRROR = 1, SR_PAD_REFLECT = 2 };
enum { SR_MAP_IDENTITY = 0, SR_MAP_WINDOW = 1 };
enum { SR_EPI_STD = 0, SR_EPI_QKV = 1, SR_EPI_QKV_OCA = 2 };
enum { SR_OUT_NHWC = 0, SR_OUT_PIXEL_SHUFFLE = 1, SR_OUT_FINAL_NCHW = 2 };
/* Row half of the cyclic shift (torch.roll over H, swinir.py:154,168) and of the shift mask (common.py:250-274):
 *   SR_Y_ROLL        whole image on this device: rows are rolled by `shift` in addressing, full mask      (default)
 *   SR_Y_STRIP       row strip of a larger image whose rows were rolled by the host's halo exchange (the buffer already
 *                    holds rows [r0+shift, r1+shift) of the image): no row roll in addressing, no row mask
 *   SR_Y_STRIP_LAST  as SR_Y_STRIP for the strip that ends with the wrapped window row: the row mask applies
 * The column half (`shift`) is unchanged in every mode. */
enum { SR_Y_ROLL = 0, SR_Y_STRIP = 1, SR_Y_STRIP_LAST = 2 };

int sr_abi_version(void);
const char* sr_last_error(void);

/* Model.inference ingest + SwinIR/HAT padding + Normalizer.normalize / MeanShift(sub)
 * (studiosr/models/common.py:42-43,108-121,228-230,277-282; swinir.py:249-255).
 * x: fp32 NCHW [B,C,H,W] -> out: NHWC [B,Hp,Wp,Cp] of out_dtype, out = x*scale[c] + bias[c],
 * rows/cols >= H/W filled by pad_mode, channels >= C zero. */
int sr_ingest_nchw(const float* x, void* out, int out_dtype, int B, int C, int H, int W, int Hp, int Wp,
                   int Cp, int pad_mode, const float* scale, const float* bias, void* stream);

/* nn.LayerNorm over the channel axis (swinir.py:26,313; hat.py:460): fp32 rows [M, Cp] -> [M, Cp]. */
int sr_layernorm(const float* x, float* y, const float* gamma, const float* beta, int M, int C, int Cp,
                 float eps, void* stream);
/* the same with a bf16 or fp32 result (y_dtype = SR_*): the consumer is a conv that rounds to bf16 anyway (hat.py:165-170) */
int sr_layernorm_to(const float* x, void* y, int y_dtype, const float* gamma, const float* beta, int M, int C, int Cp, float eps, void* stream);

typedef struct SrGemm {
    /* y = epilogue( prologue(A) @ W^T ): nn.Linear qkv/proj/fc1/fc2 with fused LayerNorm prologue,
     * bias, GELU, residual and the window partition/reverse + cyclic shift folded into row addressing
     * (swinir.py:69-71,80,103,151-172; common.py:184-195,236-247; hat.py:76-78,165-193,216,233). */
    const void* A;        /* [rows, lda] a_dtype */
    const void* Wp;       /* packed weights, compute dtype */
    const float* bias;    /* [N] or NULL */
    const float* ln_gamma;/* [K] fp32 or NULL: LayerNorm prologue over the first k_real channels (a_dtype must be f32) */
    const float* ln_beta;
    void* out;            /* STD: [rows, ldo] out_dtype.  QKV: q buffer */
    void* out_k;          /* QKV only */
    void* out_vt;         /* QKV only */
    const float* skip;    /* fp32 residual [rows, ldskip] added after act/scale, or NULL */
    int M, K, N;          /* K, N padded (K % 32 == 0, N % 64 == 0) */
    int k_real;           /* LayerNorm width */
    int lda, ldo, ldskip;
    int a_dtype, out_dtype, compute_dtype;
    int act;
    float out_scale;
    int a_map, o_map;     /* SR_MAP_*: rows are in window order on that side */
    int H, W, ws, shift;  /* geometry of the window map */
    int epi;              /* SR_EPI_* */
    int heads, hd_p, ntok;/* QKV epilogue: q,k -> [bwin][head][tok][hd_p], v -> [bwin][head][hd_p][ntok] */
    float ln_eps;
    int ln_norm_only;     /* 1: prologue is (x - mean) * rstd only (gamma/beta were folded into Wp/bias at pack time) */
    int oca_pad;          /* SR_EPI_QKV_OCA (HAT OCAB, hat.py:247-264): q -> window order as SR_EPI_QKV; k -> zero-bordered image
                           * [B][H+2p][W+2p][heads][hd_p]; v -> transposed zero-bordered [B][heads][hd_p][(H+2p)*(W+2p)]; p = oca_pad = physical border (multiple of 4).
                           * Rows must be in window order (a_map = SR_MAP_WINDOW, shift 0); the borders are zeroed by the caller. */
    int y_mode;           /* SR_Y_* (window maps only) */
    /* Optional gated second residual of the SR_EPI_STD epilogue (ABI v4; HAT, hat.py:192: x = shortcut + attn + conv_scale * CAB(x)):
     *   out[row][n] += skip2[row][n] * skip2_gate[row / gate_rows][n]
     * skip2 = the CAB conv output (NHWC, skip2_dtype, row stride ldskip2) addressed like `out`; skip2_gate = per-image channel gates
     * from sr_channel_gate (row stride ld_gate).  Replaces a separate sr_channel_attention pass over the stream. */
    const void* skip2;
    const float* skip2_gate;
    int skip2_dtype, ldskip2, gate_rows, ld_gate;
} SrGemm;
int sr_gemm(const SrGemm* a, void* stream);

typedef struct SrSwinBlock {
    /* The WHOLE SwinTransformerBlock (swinir.py:146-174: norm1, roll, window_partition, WindowAttention :78-105 with bias table and
     * calculate_mask, window_reverse, roll back, shortcut, norm2, Mlp common.py:173-195, shortcut) in ONE launch, ABI v5.  One window per
     * 4-wave workgroup; every GEMM stage is a run of uniform [64 tokens x 192 columns x 32 k] steps fed from ONE packed weight stream
     * (studiosr_amd/packing.py pack_swin_block_stream: 48 slots of 12 fragments in consumption order).  Every bias (q, v->proj, proj, fc1,
     * fc2), both LayerNorm affines, the attention scale and log2(e) (softmax runs on exp2) are folded into that stream: biases ride on
     * two constant-one pad channels (hi + lo bf16 split).  bf16 operands, fp32 stream / statistics / softmax; out may alias x. */
    const float* x;        /* [B,H,W,ldx] fp32 stream */
    float* out;
    const void* wstream;   /* 48 slots x 12 fragments x 64 lanes x 8 (bf16) or 16 (bf16x3) bf16 */
    const float* bias;     /* relative-position bias * log2(e) in fragment order [heads][qt][kt][lane][4] */
    int B, H, W, C, Cp, ldx, heads, hd_p, ws, shift, Hp;
    float eps;
    int y_mode;            /* SR_Y_* */
    int compute_dtype;     /* SR_BF16: bf16 operands (wstream 48 * 12 * 64 * 8 bf16); SR_BF16X3: split operands hi + lo (precision "fp32x3": fp32-class
                            * accuracy, wstream 48 * 12 * 64 * 16 bf16 = per lane 8 hi | 8 lo, erf GELU, bias pre-scaled by log2(e) as for bf16) */
    int max_workgroups;    /* ABI v10.  0 (and -1): one workgroup per window (default: fastest at every size measured); > 0: at most this many PERSISTENT workgroups,
                            * each walking windows b, b + grid, ... with the next window's rows fetched under the current result stores; -2: as many as the
                            * device holds at once (bf16: 3 per CU).  Results do not depend on it (a window's arithmetic is the same whoever computes it). */
} SrSwinBlock;
int sr_swin_block_supported(int C, int Cp, int heads, int hd_p, int ws, int Hp, int compute_dtype);
int sr_swin_block(const SrSwinBlock* a, void* stream);

typedef struct SrSwinLight {
    /* The WHOLE SwinTransformerBlock (swinir.py:146-174) of the reference's lightweight SwinIR geometry (SwinIR.from_pretrained(light=True),
     * swinir.py:418-427: embed_dim 60, 6 heads of 10, window 8, hidden 120) in ONE launch, ABI v7: one window per 4-wave workgroup, padded sizes
     * 64 channels / 16 features per head / 128 hidden.  Weights in fragment order (packing.to_fragments), LayerNorm affines folded into the
     * following Linear, attention scale in the q rows: wqkv [288 x 64] with row part * 96 + head * 16 + d, wproj [64 x 96] (column head * 16 + d),
     * w1 [128 x 64], w2 [64 x 128]; biases fp32, zero in the pads; bias = relative-position bias in accumulator-fragment order
     * [heads][4][4][64][4] (packing.bias_fragments).  bf16 operands, fp32 stream / statistics / softmax; out may alias x. */
    const float* x; float* out;
    const void* wqkv; const float* bqkv; const void* wproj; const float* bproj; const void* w1; const float* b1; const void* w2; const float* b2;
    const float* bias;
    int B, H, W, C, ldx, shift;
    float eps;
    int y_mode;            /* SR_Y_* */
    int compute_dtype;     /* SR_BF16, or SR_BF16X3 (weights packed hi | lo, precision "fp32x3": fp32-class accuracy, erf GELU) */
} SrSwinLight;
int sr_swin_light_supported(int C, int Cp, int heads, int hd, int ws, int hidden, int compute_dtype);
int sr_swin_light(const SrSwinLight* a, void* stream);

typedef struct SrSwinQkv {
    /* q, k, v = qkv(LayerNorm1(x)) in front of sr_window_attention, ABI v6 (hat.py:164-176; swinir.py:146-160 for geometries sr_swin_block
     * does not cover): the stream-form replacement of sr_gemm's SR_EPI_QKV launch.  One workgroup per 64 consecutive window-order tokens
     * (roll + window_partition = its row gather); weights from ONE packed stream (packing.py pack_swin_qkv_stream: 18 slots; LayerNorm1
     * affine, attention scale and all three biases folded in).  Outputs bf16: q, k [B*nW][heads][ws*ws][hd_p], vt [B*nW][heads][hd_p][ws*ws]. */
    const float* x;        /* [B,H,W,ldx] fp32 stream */
    void* q; void* k; void* vt;
    const void* wstream;   /* 18 slots x 12 fragments x 64 lanes x 8 bf16 */
    int B, H, W, C, Cp, ldx, heads, hd_p, ws, shift;
    float eps;
    int y_mode;            /* SR_Y_* */
    int compute_dtype;     /* SR_BF16 */
    int frag_order;        /* 1: q, k, vt in the fragment order of SrWindowAttn.qkv_frag (oca_pad == 0, ws 16 only) */
    int oca_pad;           /* 0: k / vt in window order (above).  > 0 (HAT OCAB, hat.py:247-264; as sr_gemm's SR_EPI_QKV_OCA): k -> zero-bordered image order
                            * [B][H+2p][W+2p][heads][hd_p], vt -> transposed zero-bordered planes [B][heads][hd_p][(H+2p)(W+2p)], p = oca_pad (multiple of 4), shift 0 */
    void* n1;              /* optional side output (ABI v8; HAT: the input of the block's CAB convolutions, hat.py:165-170): LayerNorm1(x) * n1_gamma + n1_beta in image
                            * order [B,H,W,ldn], bf16 (SR_BF16) or fp32 (SR_BF16X3) -- saves the stand-alone sr_layernorm launch of a group's first block */
    const float* n1_gamma; const float* n1_beta;  /* [Cp], pads 0 */
    int ldn;
} SrSwinQkv;
int sr_swin_qkv_supported(int C, int Cp, int heads, int hd_p, int ws, int compute_dtype);
int sr_swin_qkv(const SrSwinQkv* a, void* stream);

typedef struct SrSwinTail {
    /* Everything of a window-attention block BEHIND its attention kernel in ONE launch, ABI v6 (hat.py:172-194 HAB, hat.py:286-293 OCAB;
     * swinir.py:169-174 for geometries sr_swin_block does not cover):
     *     x1  = x + proj(O) + bproj [ + y * gate[image] ]      out = x1 + fc2(GELU(fc1(LayerNorm2(x1))))
     * O = output of sr_window_attention / sr_oca: bf16 [B*H*W][heads*hd_p] in WINDOW order (window_reverse and the roll back are the row
     * gather of this kernel); x, y, out in image order.  Replaces sr_gemm (projection, SR_EPI_STD with skip / gated skip2) + sr_mlp_fused.
     * One workgroup per 64 consecutive window-order tokens; weights from ONE packed stream (packing.py pack_swin_tail_stream: 6 proj slots,
     * then slots 24..47 of the sr_swin_block stream: fc1 / fc2 with their biases on the constant-one pad channels, LayerNorm2 folded). */
    const float* x;        /* [B,H,W,ldx] fp32 stream (the shortcut) */
    float* out;            /* may alias x */
    const void* o;         /* bf16 [B*H*W][heads*hd_p] */
    const void* wstream;   /* 30 (48 with the fused next-block QKV, see q2) slots x 12 fragments x 64 lanes x 8 bf16 */
    const float* bproj;    /* [Cp] fp32, pad 0 */
    const void* y;         /* optional second residual bf16 [B,H,W,ldy] (HAT: the CAB convolution output) or NULL */
    const float* gate;     /* [B][ld_gate] fp32 per-image channel gates from sr_channel_gate (conv_scale folded in); read when y != NULL */
    int B, H, W, C, Cp, ldx, ldy, ld_gate, heads, hd_p, ws, shift, Hp;
    float eps;
    int y_mode;            /* SR_Y_* */
    int compute_dtype;     /* SR_BF16 */
    /* optional side output: n1 = LayerNorm(out) * n1_gamma + n1_beta as bf16 [B,H,W,ldn] (HAT: norm1 of the NEXT block, whose CAB
     * convolutions read it, hat.py:165-170 -- replaces that block's sr_layernorm_to launch).  NULL: not written. */
    void* n1;
    const float* n1_gamma; /* [Cp] fp32, pad 0 */
    const float* n1_beta;
    int ldn;
    /* optional in-kernel gate (replaces sr_channel_gate + `gate`): when pool_partial != NULL every workgroup recomputes
     * gate[c] = y_scale * sigmoid(W2 relu(W1 mean + b1) + b2) of its image from the per-tile channel sums of the CAB convolution
     * (same code and arithmetic as sr_channel_gate; fields as in SrChannelAttn). */
    const float* pool_partial; /* [B, ca_n_tiles, Cp] or NULL */
    const float* ca_w1; const float* ca_b1; const float* ca_w2; const float* ca_b2;
    int ca_Cr, ca_n_tiles;
    float y_scale;
    /* optional fused stage: q, k, v = qkv(LayerNorm1(out)) of the NEXT block (what sr_swin_qkv would compute from `out`), scattered into that
     * block's window order (its shift is shift2).  When q2 != NULL, wstream continues with the 18 slots of packing.pack_swin_qkv_stream of
     * the next block (48 slots in all).  Layouts as SrSwinQkv; y_mode must be SR_Y_ROLL; both shifts multiples of 4. */
    void* q2; void* k2; void* vt2;
    int shift2;
    int frag_order;        /* as SrSwinQkv.frag_order, for q2 / k2 / vt2 */
    int oca_pad2;          /* (ABI v9) > 0: the fused stage is the LayerNorm1 + QKV of the group's overlapping cross attention (hat.py:247-264): q2 in window order
                            * (shift2 = 0, row-major), k2 / vt2 in the zero-bordered layouts of SrSwinQkv.oca_pad = oca_pad2 (a multiple of 4) */
    int wg_tokens;         /* (ABI v9) tokens per workgroup: 64, 32 (SR_BF16 only: twice the workgroups of half the rows each -- for launches that would leave the chip at
                            * one 64-token workgroup per CU or less, where the kernel is one latency chain per CU), or 0 = the launcher decides (32 up to 128 workgroups of 64) */
} SrSwinTail;
int sr_swin_tail_supported(int C, int Cp, int heads, int hd_p, int ws, int Hp, int compute_dtype);
int sr_swin_tail(const SrSwinTail* a, void* stream);

typedef struct SrMlp {
    /* x_out = x + fc2(GELU(fc1(LayerNorm(x)))) in ONE kernel (common.py:173-195, swinir.py:172, hat.py:193,292):
     * the hidden activations never leave the CU.  bf16 operands / fp32 accumulate; out may alias x. */
    const float* x;       /* [M, ldx] fp32 stream */
    float* out;           /* [M, ldx] fp32 */
    const float* ln_gamma;/* [Cp], or NULL when gamma/beta are folded into w1p/b1 */
    const float* ln_beta; /* [Cp] or NULL */
    const void* w1p;      /* packed fc1 [Hp x Cp] bf16 */
    const float* b1;      /* [Hp] */
    const void* w2p;      /* packed fc2 [Cp x Hp] bf16 */
    const float* b2;      /* [Cp] */
    int M, C, Cp, Hp, ldx;
    float eps;
} SrMlp;
int sr_mlp_fused_supported(int Cp, int Hp, int compute_dtype); /* 1 if sr_mlp_fused covers this shape */
int sr_mlp_fused(const SrMlp* a, void* stream);

typedef struct SrConv3x3 {
    /* nn.Conv2d(k=3, s=1, p=1) as an im2col-free implicit GEMM on an LDS halo tile, fused bias /
     * ReLU / LeakyReLU / GELU / res_scale / residual and optionally nn.PixelShuffle or the final
     * unnormalise+crop+NCHW store (common.py:104-105,124-153,232-233; swinir.py:241,290,316-326,371-372;
     * edsr.py:34-48; rcan.py:11-36; hat.py:45-47,380,462-467). */
    const void* x;        /* NHWC [B,H,W,Cin_p] x_dtype */
    const void* Wp;       /* packed, compute dtype, K = 9*Cin_p */
    const float* bias;    /* [Cout_p] or NULL */
    void* out;
    const void* skip;     /* residual, same geometry as the NHWC output, skip_dtype, or NULL */
    float* pool_partial;  /* optional [B, n_tiles, Cout_p] per-tile channel sums (channel attention), or NULL */
    const float* fin_scale; /* FINAL_NCHW: out = (acc + bias) * fin_scale[c] + fin_bias[c] */
    const float* fin_bias;
    int B, H, W, Cin_p, Cout_p;
    int x_dtype, out_dtype, skip_dtype, compute_dtype;
    int act;
    float out_scale;
    int out_mode;         /* SR_OUT_* */
    int ps_r;             /* PIXEL_SHUFFLE factor r: packed channel n = (i*r + j)*Cps_p + c */
    int cps_p;            /* PIXEL_SHUFFLE: padded channel count of the shuffled output */
    int fin_c, fin_h, fin_w; /* FINAL_NCHW: real channels and cropped size */
    float act_slope;      /* SR_ACT_LRELU: negative slope; 0 = nn.LeakyReLU's default 0.01 (ABI v4; SwinFIR's SFB uses 0.2, swinfir.py:59) */
    int tile_rows;        /* bf16 tile height: 0 = chosen by the library (8, or 4 when the launch would leave the chip under-filled; always 8
                           * with pool_partial), 4 / 8 = as given -- with pool_partial the caller sizes it with sr_conv3x3_pool_tiles_rows */
} SrConv3x3;
int sr_conv3x3(const SrConv3x3* a, void* stream);
int sr_conv3x3_pool_tiles(int H, int W, int Cout_p, int compute_dtype); /* n_tiles of pool_partial for this geometry (tile_rows = 0) */
int sr_conv3x3_pool_tiles_rows(int H, int W, int Cout_p, int tile_rows); /* the same for an explicit bf16 tile height (4 or 8) */

typedef struct SrRcab {
    /* y = conv2(ReLU(conv1(x))): the conv-ReLU-conv body of an RCAB (rcan.py:11-24, common.py:140-153 without the residual) for 64
     * (padded) channels in ONE launch -- the intermediate stays in LDS -- plus per-tile channel sums of y for the channel-attention
     * gate (sr_channel_attention).  bf16 operands / fp32 accumulate; same bits as two sr_conv3x3 launches. */
    const void* x;        /* NHWC [B,H,W,64] x_dtype */
    const void* w1p;      /* packed conv1 weights (sr_conv3x3 layout), bf16 */
    const float* b1;      /* [64] */
    const void* w2p;      /* packed conv2 weights */
    const float* b2;      /* [64] */
    void* y;              /* NHWC [B,H,W,64] y_dtype */
    float* pool_partial;  /* optional [B, sr_rcab_pool_tiles(H, W), 64] */
    int B, H, W, C_p;     /* C_p must be 64 */
    int x_dtype, y_dtype;
    /* Optional gated input (ABI v4): the channel-attention tail of the PREVIOUS RCAB (rcan.py:21-24: res = CA(body(x)); res += x) is
     * folded into this launch's halo staging, so an RCAB is ONE launch instead of conv pair + sr_channel_attention:
     *   x_eff = x + gate * gate_y,  gate = sigmoid(W2 relu(W1 mean(gate_y) + b1) + b2)  (mean from gate_pool, as sr_channel_attention)
     * x_eff feeds conv1 and its tile interior is written to x_out (fp32 NHWC) -- the skip tensor of this RCAB.  x must be fp32,
     * gate_y has y_dtype; neither may alias y / x_out / pool_partial.  gate_y == NULL: plain conv pair.  gate_Cr <= 8. */
    const void* gate_y;        /* NHWC [B,H,W,64] y_dtype: conv-pair output of the previous RCAB */
    const float* gate_pool;    /* its pool partials [B, sr_rcab_pool_tiles(H, W), 64] */
    const float* gate_w1;      /* [Cr, C] */
    const float* gate_b1;      /* [Cr] */
    const float* gate_w2;      /* [C, Cr] */
    const float* gate_b2;      /* [C] */
    float* x_out;              /* NHWC [B,H,W,64] fp32 */
    int gate_C, gate_Cr;
    int compute_dtype;         /* ABI v11.  0 or SR_BF16: bf16 operands; SR_BF16X3: split operands hi + lo (precision "fp32x3": w1p / w2p packed hi | lo, x / y / gate_y fp32;
                                * 152 KB of LDS images, one workgroup per CU) -- the conv family of the reference-precision path (common.py:36-48) */
} SrRcab;
int sr_rcab_conv_pair(const SrRcab* a, void* stream);
int sr_rcab_pool_tiles(int H, int W); /* n_tiles of pool_partial */

typedef struct SrCab {
    /* y = conv2(GELU(conv1(x))): the body of HAT's CAB (hat.py:41-49: Conv2d(180, 60, 3), GELU, Conv2d(60, 180, 3)) in ONE launch, ABI v6 --
     * the intermediate stays in LDS -- plus per-tile channel sums of y for the ChannelAttention squeeze (sr_channel_gate / sr_swin_tail).
     * bf16 NHWC in and out, fp32 accumulate; weights packed as for sr_conv3x3 (packing.pack_conv3x3). */
    const void* x;        /* NHWC [B,H,W,Cin_p] bf16 (HAT: LayerNorm1 output) */
    const void* w1p;      /* conv1: Cout_p = Cmid_p rows, Cin_p input channels */
    const float* b1;      /* [Cmid_p] */
    const void* w2p;      /* conv2: Cout_p rows, Cmid_p input channels */
    const float* b2;      /* [Cout_p] */
    void* y;              /* NHWC [B,H,W,Cout_p] bf16; must not alias x */
    float* pool_partial;  /* optional [B, sr_cab_pool_tiles(H, W), Cout_p] */
    int B, H, W, Cin_p, Cmid_p, Cout_p;  /* 192, 64, 192 */
    int dtype;            /* SR_BF16; or SR_BF16X3 (ABI v11, sr_cab_fused only): split operands hi + lo (precision "fp32x3"): x, y fp32, weights packed hi | lo, erf GELU to 1.5e-7,
                           * 103 KB of LDS images on the two-phase K walk, one workgroup per CU; mid_pre must be NULL */
    void* mid_pre;        /* optional side output (ABI v8; training): conv1(x) + b1 BEFORE the GELU, NHWC [B,H,W,Cmid_p] bf16 -- what the backward needs for GELU' and for conv2's
                           * weight gradient, so that it does not run conv1 again (trainer.py:104 loss.backward() through hat.py:41-49) */
    int tile_rows;        /* (ABI v9) output rows per workgroup tile (14 columns): 0 or 6, or 8 in sr_hab_mid (large launches: fewer workgroups, less halo recomputation);
                           * pool_partial then has sr_cab_pool_tiles_rows(H, W, tile_rows) slots per image */
    /* Backward form (ABI v11, sr_cab_fused, bf16; trainer.py:104 loss.backward() through hat.py:41-49): the CAB's data gradient is the same shape of launch with the transposed
     * weights -- x = dy [.,Cin_p], w1p = conv2's flipped / transposed weights (Cin_p -> Cmid_p), w2p = conv1's (Cmid_p -> Cout_p), b1 = b2 = zeros -- and the pointwise step
     * mid = conv(x) * GELU'(bwd_pre) instead of GELU(conv(x) + b1).  bwd_pre = the forward's mid_pre; side outputs for the weight gradients, NHWC [B,H,W,Cmid_p] bf16:
     * bwd_dmid = mid, bwd_g = GELU(bwd_pre).  Replaces sr_conv3x3 -> sr_tr_gelu -> sr_conv3x3. */
    const void* bwd_pre;
    void* bwd_dmid;
    void* bwd_g;
} SrCab;
int sr_cab_supported(int Cin_p, int Cmid_p, int Cout_p, int dtype);
int sr_cab_pool_tiles(int H, int W);
int sr_cab_pool_tiles_rows(int H, int W, int tile_rows);
int sr_cab_fused(const SrCab* a, void* stream);

typedef struct SrWindowAttn {
    /* softmax(q k^T + bias[head] + shift mask) v per (window, head)
     * (swinir.py:83-102; common.py:250-274; hat.py:90-107).  q is pre-scaled (scale folded into Wq). */
    const void* q;        /* [bwin][head][ntok][hd_p] T */
    const void* k;        /* [bwin][head][ntok][hd_p] T */
    const void* vt;       /* [bwin][head][hd_p][ntok] T */
    const float* bias;    /* [heads][ntok][ntok] fp32 (table[rpi] gathered once per model) */
    void* out;            /* [bwin*ntok][heads*hd_p] T (window-order rows) */
    int n_bwin, heads, hd_p, ntok;
    int H, W, ws, shift;  /* mask geometry (shift == 0 -> no mask) */
    int dtype;             /* SR_F32 (exact fp32 MFMAs), SR_BF16, or -- ABI v11 -- SR_BF16X3: fp32 q / k / vt / out, every product as split-operand bf16 (precision "fp32x3"); needs
                            * bias_frag, ws % 4 == 0, hd_p 32, 64 / 256 tokens, row-major operands */
    int y_mode;           /* SR_Y_* */
    const float* bias_frag; /* optional: the same bias in accumulator-fragment order [heads][qt][kt][lane][4]
                             * (element = bias[h][16 qt + (lane & 15)][16 kt + 4 (lane >> 4) + r]); selects the flash-form kernel */
    int qkv_frag;           /* 1 (ABI v6; bf16, ntok 256, hd_p 32): q, k, vt are in FRAGMENT order as written by sr_swin_qkv / sr_swin_tail with frag_order = 1
                             * (q, k: [tile of 16 tokens][lane = 16 g + i][8] = token 16 tile + i, features 8 g ..; vt: [64-key block][d tile][32-key step][lane][8]
                             * = d 16 dt + i, keys 64 kb + 32 ks + 16 (e >> 2) + 4 g + (e & 3)): every operand fragment is one coalesced 1-KiB load */
    const float* bias_tiles; /* optional (ABI v8; bf16, 16 x 16 windows, hd_p 32): the bias as its 31 DISTINCT 16 x 16 tiles per head, [heads][31][lane][4] in the
                              * accumulator-fragment order of bias_frag -- valid when bias[h][q][k] depends on (q >> 4) - (k >> 4) and the in-row offsets only, as
                              * every relative-position bias does (tile d = (q >> 4) - (k >> 4) + 15).  Selects the LDS form: one (window, head) per workgroup with K,
                              * V^T and these tiles staged in LDS once (csrc/sr_wattn_lds_body.h) */
    const float* x;          /* optional (ABI v8; bf16, 16 x 16 windows, hd_p 32, heads 6, C 180 in 192 padded channels, bias_tiles set): the fp32 stream [B,H,W,ldx] -- the
                              * workgroup of a (window, head) then computes q, k, v ITSELF as qkv_h(LayerNorm1(x)) from wqkv (q / k / vt are not read; roll + window_partition
                              * = the row gather, shift / y_mode as SrSwinQkv) and keeps them in LDS (csrc/sr_wattn_qkv_body.h) */
    const void* wqkv;        /* sr_swin_qkv's 18-slot weight stream (packing.pack_swin_qkv_stream) */
    int ldx, C;
    float eps;
} SrWindowAttn;
int sr_window_attention(const SrWindowAttn* a, void* stream);

/* The two independent middle stages of a HAB as ONE launch, ABI v8 (hat.py:165-176: conv_x = conv_block(norm1(x)) beside attn_x = attn(x_windows)):
 * the first sr_cab_pool_tiles(H, W) * B workgroups run sr_cab_fused's tiles on `cab`, the others sr_window_attention's flash form on `attn` (16 x 16
 * windows, head_dim <= 32, bf16, attn->bias_frag required, attn->qkv_frag honoured).  Same outputs as the two separate calls (the CAB's conv1 sums its K in
 * two 96-channel phases: same products, another fp32 order); one launch instead of a fork / join across two queues of a captured graph. */
int sr_hab_mid_supported(int ntok, int hd_p, int ws, int attn_dtype, int Cin_p, int Cmid_p, int Cout_p, int cab_dtype);
int sr_hab_mid(const SrWindowAttn* attn, const SrCab* cab, void* stream);

typedef struct SrOcaAttn {
    /* HAT overlapping cross attention core (hat.py:266-283): every ws x ws query window attends to the
     * (ws+2p) x (ws+2p) neighbourhood around it (nn.Unfold with zero padding, :217-221,255): softmax(q k^T + bias) v.
     * Buffers as written by sr_gemm with SR_EPI_QKV_OCA. */
    const void* q;        /* [bwin][head][ws*ws][hd_p] T, pre-scaled */
    const void* k;        /* [B][H+2e][W+2e][heads][hd_p] T, zero border of e = `border` pixels (e >= pad, e % 4 == 0) */
    const void* vt;       /* [B][heads][hd_p][(H+2e)*(W+2e)] T, zero border */
    const float* bias;    /* [heads][ws*ws][nk_pad] fp32, nk_pad = keys padded to a multiple of 32 (pad columns ignored) */
    void* out;            /* [bwin*ws*ws][heads*hd_p] T (window-order rows) */
    int B, H, W, heads, hd_p, ws, pad, border, nk_pad;
    int dtype;
    const float* bias_frag; /* optional: the bias in accumulator-fragment order [heads][qt][nk_frag/16][lane][4] with the key dimension
                             * padded to nk_frag (multiple of 64) by -1e30 columns; selects the flash-form kernel */
    int nk_frag;
    const float* bias_rel;  /* optional (ABI v8; bf16, ws 16, pad == border == 4): the bias as its relative-position TABLE, [heads][1521] with
                             * bias[h][q][k] = bias_rel[h][(ky - qy + 15) * 39 + (kx - qx + 15)] (q = 16 qy + qx, k = 24 ky + kx) -- the reference's table rotated by 880
                             * entries, which is where its python-style negative indices land (hat.py:494-517, 276-279); packing.oca_bias_rel builds it from the
                             * gathered bias and verifies every entry.  Selects the LDS form: one (window, head) per workgroup with K, V^T and this table staged in
                             * LDS once (csrc/sr_oca_lds.hip) */
} SrOcaAttn;
int sr_oca_attention(const SrOcaAttn* a, void* stream);

/* Model.inference front / back end (common.py:42-45), batched: uint8 HWC [B,H,W,C] -> fp32 NCHW, out = u8 / divisor
 * (255 iff img_range == 1.0, common.py:39), and fp32 NCHW -> uint8 HWC, out = uint8(clip(round_half_even(x * mult), 0, 255)).
 * IEEE fp32 division / multiplication: bit-identical to the reference's numpy / torch ops. */
int sr_u8_to_nchw(const unsigned char* in, float* out, int B, int C, int H, int W, float divisor, void* stream);
int sr_nchw_to_u8(const float* in, unsigned char* out, int B, int C, int H, int W, float mult, void* stream);

/* nn.PixelShuffle (common.py:129,133,136) standalone: out[b,c,h*r+i,w*r+j] = in[b,c*r*r+i*r+j,h,w].
 * elem_size 2 or 4 bytes; tensors are plain NCHW. Bit-exact copy. */
int sr_pixel_shuffle_nchw(const void* in, void* out, int elem_size, int B, int C_out, int H, int W, int r,
                          void* stream);

/* ChannelAttention (common.py:156-170, hat.py:25-38) second half: given per-tile channel sums of y,
 * s = sigmoid(W2 relu(W1 mean + b1) + b2); out = y * s * y_scale + skip.  y/out/skip NHWC fp32 or bf16. */
typedef struct SrChannelAttn {
    const void* y;
    const float* pool_partial; /* [B, n_tiles, C_p] */
    const float* w1;      /* [Cr, C] */
    const float* b1;      /* [Cr] */
    const float* w2;      /* [C, Cr] */
    const float* b2;      /* [C] */
    const void* skip;     /* or NULL */
    void* out;
    int B, H, W, C, C_p, Cr, n_tiles;
    int y_dtype, skip_dtype, out_dtype;
    float y_scale;
    const void* skip2;    /* optional second residual (HAT: shortcut + attn + conv_scale*cab) or NULL */
    int skip2_dtype;
} SrChannelAttn;
int sr_channel_attention(const SrChannelAttn* a, void* stream);
/* The squeeze half alone: gate[b][c] = y_scale * sigmoid(W2 relu(W1 mean_b + b1) + b2) for c < C, 0 for C <= c < C_p (fp32 [B, C_p]);
 * same arithmetic as sr_channel_attention.  Consumed by sr_gemm's gated second residual.  Only pool_partial, w1, b1, w2, b2, B, H, W, C,
 * C_p, Cr, n_tiles and y_scale of the struct are read. */
int sr_channel_gate(const SrChannelAttn* a, float* gate, void* stream);


/* ------------------------------------------------------------------------------------------------------------------
 * Training engine (ABI v3): the kernels behind studiosr_amd/autograd.py, i.e. forward + backward of every op the
 * reference's training step reaches (studiosr/engine/trainer.py:97-109: autocast forward, L1 loss, backward, Adam).
 * All tensors fp32, UNPADDED and in the reference's own layouts (parameters / gradients exactly as in the state_dict).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct SrBgemm {
    /* C[b1,b2][m,n] (=, += or atomic +=) alpha * sum_k A[b1,b2][m,k] * B[b1,b2][k,n] + bias[n] on the exact-fp32 matrix
     * cores.  Every operand is addressed through element strides, so transposes, head slices of a packed qkv tensor and
     * NCHW / NHWC outputs are free.  Replaces torch's addmm / bmm / mkldnn_convolution (over an sr_im2col3x3 buffer) and their
     * autograd adjoints: nn.Linear (swinir.py:69-71, common.py:184-195), q k^T and attn v (swinir.py:85,102; hat.py:92,107,
     * 268,283), nn.Conv2d 3x3 / 1x1 (common.py:104-105,140-153,156-170; hat.py:25-52).
     * ksplit > 1 splits K over workgroups and accumulates with fp32 atomics (weight gradients: K = tokens); C must then hold
     * the value to accumulate onto (zeros for a fresh gradient). */
    const float* A; const float* B; float* C;
    const float* bias;                /* [N] or NULL */
    int M, N, K;
    long long sa_m, sa_k, sb_k, sb_n, sc_m, sc_n;                 /* element strides */
    int nb1, nb2;                                                 /* two batch levels (e.g. windows x heads) */
    long long sa_b1, sa_b2, sb_b1, sb_b2, sc_b1, sc_b2;
    float alpha;
    int accumulate;                   /* ksplit == 1: C += instead of C = */
    int ksplit;
    int compute_dtype;                /* SR_F32: exact fp32 MFMA.  SR_BF16: operands rounded to bf16 while staged, bf16 MFMA, fp32 accumulate
                                       * (what torch.autocast(bfloat16) does to the reference's matmuls, trainer.py:80,102); large shapes only,
                                       * small ones stay on the fp32 kernel */
} SrBgemm;
int sr_bgemm(const SrBgemm* g, void* stream);

/* col[m, tap*C + c] = x[b, y + tap/3 - 1, x + tap%3 - 1, c] (zero outside), m = (b*H + y)*W + x, tap = ky*3 + kx: reads and writes
 * are contiguous over c, so conv(x, w) = col @ w.permute(0, 2, 3, 1).view(Cout, 9*Cin)^T (common.py:104-105).  x is addressed by
 * element strides (NCHW or NHWC).  sr_col2im3x3 is the adjoint written as a gather: dx[m, c] = sum_tap dcol[m - tap offset, tap*C + c]. */
int sr_im2col3x3(const float* x, float* col, int B, int H, int W, int C, long long sb, long long sy, long long sx, long long sc, void* stream);
int sr_col2im3x3(const float* dcol, float* dx, int B, int H, int W, int C, void* stream);

/* rows r = (bw*heads + h)*Nq + i of S [.., Nk]: S <- softmax(S + bias[h,i,:] + mask[bw % nW, i, :]) in place
 * (swinir.py:92-100, hat.py:97-106,276-281; bias / mask may be NULL); backward: dP <- P * (dP - sum_j dP_j P_j). */
int sr_softmax_fwd(float* S, const float* bias, const float* mask, long long rows, int heads, int Nq, int Nk, int nW, void* stream);
int sr_softmax_bwd(const float* P, float* dP, long long rows, int Nk, void* stream);

/* nn.LayerNorm(C) forward saving (mean, rstd) per row, and its backward; dgamma / dbeta are ACCUMULATED (atomics). */
int sr_layernorm_fwd_train(const float* x, const float* gamma, const float* beta, float* y, float* stats, long long M, int C, float eps, void* stream);
int sr_layernorm_bwd(const float* x, const float* stats, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, long long M, int C, void* stream);

/* out[b][c] += alpha * sum_p x[b][p][c] (bias gradients, AdaptiveAvgPool2d(1): common.py:160, hat.py:31);
 * out[i] = sum_b x[b*stride_b + i] (relative-position-bias gradient summed over windows, deterministic). */
int sr_colsum(const float* x, float* out, int nb, long long P, int C, float alpha, void* stream);
int sr_batch_sum(const float* x, float* out, long long nb, long long n, long long stride_b, void* stream);

/* flat elementwise ops (forward and backward of GELU / ReLU / LeakyReLU / sigmoid, a x + b y, x*y, DropPath's per-sample scale
 * (swinir.py:137,171-172), the channel-attention gate x[b,p,c] * s[b,c] and its broadcast adjoint, per-channel affine) */
enum { SR_EW_GELU_FWD = 0, SR_EW_GELU_BWD = 1, SR_EW_RELU_FWD = 2, SR_EW_RELU_BWD = 3, SR_EW_LRELU_FWD = 4, SR_EW_LRELU_BWD = 5, SR_EW_AXPBY = 6,
       SR_EW_MUL = 7, SR_EW_SIGMOID_FWD = 8, SR_EW_SIGMOID_BWD = 9, SR_EW_SCALE_SAMPLE = 10, SR_EW_MUL_BC = 11, SR_EW_BCAST_BC = 12, SR_EW_AFFINE_C = 13 };
int sr_eltwise(int op, const float* x, const float* y, const float* s, float* out, long long n, long long inner, int C, float a, float b, void* stream);

/* index maps, each usable in both directions (the adjoint of a permutation is its inverse; overlapping OCA windows fold by gather):
 * window_partition(roll(x, -shift)) <-> image (swinir.py:154-158,164-168); nn.Unfold of OCAB (hat.py:217-221,255-263);
 * nn.PixelShuffle on NHWC (common.py:129,133,136); relative_position_bias_table[rpi] with python-style negative wrap
 * (swinir.py:86-91, hat.py:93-96,276-279; the adjoint scatter-adds into dtable); the model's NCHW output with Normalizer.unnormalize /
 * MeanShift(add) and the crop to [H*s, W*s] (common.py:232-233, swinir.py:372). */
int sr_window_copy(const float* src, float* dst, int B, int H, int W, int C, int ws, int shift, int to_windows, void* stream);
int sr_oca_unfold(const float* img, float* win, int B, int H, int W, int C, int ws, int wse, int forward, void* stream);
int sr_pixel_shuffle_nhwc(const float* src, float* dst, int B, int H, int W, int C, int r, int forward, void* stream);
int sr_bias_gather(const float* table, const long long* rpi, float* bias, float* dtable, int T, int heads, long long NN, int forward, void* stream);
int sr_nhwc_out(const float* src, float* dst, const float* scale, const float* shift, int B, int Hs, int Ws, int C, int Ho, int Wo, int forward, void* stream);


/* channel concat / split (swinfir.py:25,31,79; han.py:105,112): dst[r, off_d + c] (=|+=) src[r, off_s + c], c < n.
 * nn.Conv3d(1, 1, 3, padding=1) over the (C, H, W) volume of an NHWC tensor (HAN's CSAM, han.py:37-53): w = 27 taps [dc][dy][dx];
 * flip = 1 applies the adjoint (data gradient); sr_conv3d27_wgrad accumulates dw[27] and db[1]. */
int sr_copy_cols(const float* src, float* dst, long long rows, int n, int ld_s, int off_s, int ld_d, int off_d, int accumulate, void* stream);
int sr_conv3d27(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int C, int flip, void* stream);
int sr_conv3d27_wgrad(const float* x, const float* dy, float* dw, float* db, int B, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * One-time weight layout transforms, device side, enqueue-only (what studiosr_amd/packing.py does on the host; bit-identical).
 * A host in any language turns reference checkpoint tensors (fp32, state_dict layouts) into the packed operands above:
 *   sr_pack_matrix          nn.Linear weight [n_rows, n_cols] (row stride ld) -> fragments of out_dtype.  Padded row n takes
 *                           source row row_idx[n] (-1 = zero row; NULL = identity up to n_rows), padded column k takes col_idx[k].
 *                           row_scale[N_p] (attention scale hd^-0.5 on the q rows, swinir.py:83) and col_scale[n_cols]
 *                           (LayerNorm gamma when the affine is folded into the Linear) are optional single multiplies.
 *                           The qkv row order is part*heads*hd_p + head*hd_p + d; proj / fc use identity maps with zero pads.
 *   sr_pack_conv3x3         nn.Conv2d weight [Cout, Cin, 3, 3] -> fragments with k = (ky*3 + kx)*cin_p + c; row_idx as above
 *                           (a conv feeding nn.PixelShuffle(r) uses packed row (i*r + j)*cps_p + c <- channel c*r*r + i*r + j,
 *                           common.py:129,133,136).
 *   sr_pack_vector          bias / LayerNorm vector -> zero-padded fp32 [n_p] through the same index map (+ optional scale).
 *   sr_pack_bias_fragments  relative_position_bias_table [T, heads] gathered through rpi [Nq*Nk] (int64, negative indices wrap as
 *                           the reference's python indexing does: hat.py:494-517) into accumulator-fragment order
 *                           [heads][Nq/16][Nk/16][lane][4] for sr_swin_attn_fused / sr_window_attention / sr_oca_attention.
 * ------------------------------------------------------------------------------------------------------------------ */
int sr_pack_matrix(const float* w, long long ld, const int* row_idx, const int* col_idx, const float* row_scale, const float* col_scale, void* out, int out_dtype,
                   int N_p, int K_p, int n_rows, int n_cols, void* stream);
int sr_pack_conv3x3(const float* w, const int* row_idx, void* out, int out_dtype, int N_p, int Cout, int Cin, int cin_p, void* stream);
int sr_pack_vector(const float* b, const int* idx, const float* scale, float* out, int n_p, int n, void* stream);
int sr_pack_bias_fragments(const float* table, const long long* rpi, float* out, int T, int heads, int Nq, int Nk, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fast training path (ABI v7): the fused forward + backward of a training step under torch.autocast(bfloat16)
 * (studiosr/engine/trainer.py:97-109: autocast forward, L1 loss, loss.backward(), Adam) for the default HAT / SwinIR
 * block geometry (C 180, 6 heads, hidden 360, windows 8 / 16).  Activations between kernels are bf16 (what autocast
 * stores), the residual stream / LayerNorm statistics / accumulators / parameters / gradients are fp32.  The host side
 * is studiosr_amd/fasttrain.py; every other geometry and the exact-fp32 path stay on the generic engine above.
 * ------------------------------------------------------------------------------------------------------------------ */

/* Packed operand arena <- flat fp32 parameter buffer through host-built index maps (per optimizer step, ONE launch):
 *   v = scl[i] * (idx[i] >= 0 ? P[idx[i]] : 1) * (idx2 && idx2[i] >= 0 ? P[idx2[i]] : 1);
 *   mode[i] == 1: v = bf16(v) (leading part of a bias carried as hi + lo on the two constant-one channels), == 2: v - bf16(v);
 *   out[i] = (out_dtype) v.  Replaces studiosr_amd/packing.py's per-tensor torch ops in the training loop. */
int sr_tr_gather(const float* P, const int* idx, const int* idx2, const float* scl, const unsigned char* mode, void* out, int out_dtype, long long n, void* stream);
/* The adjoint: grad[i] = scale[i] * sum_{s < ns[i]} arena[src[i] + s * stride[i]]  (src[i] == -1: grad[i] = 0; src[i] == -2: grad[i] untouched); sums in slice order. */
int sr_tr_finalize(const float* arena, const long long* src, const int* stride, const int* ns, const float* scale, float* grad, long long n, void* stream);
/* (ABI v9) the same sums with the items in arena order: grad[dst[i]] = scale[i] * sum_{s < ns[i]} arena[src[i] + s * stride[i]] (the host sorts the items by src: coalesced
 * partial reads whatever the parameter order is; ns[i] = 0 writes a zero) */
int sr_tr_finalize_to(const float* arena, const long long* src, const int* dst, const int* stride, const int* ns, const float* scale, float* grad, long long n, void* stream);
int sr_tr_finalize_to8(const float* arena, const long long* src, const int* dst, const int* stride, const int* ns, const float* scale, float* grad, long long n, void* stream); /* ABI v11: eight lanes per item (SrTrFinalize.lanes) */

typedef struct SrTrWgradJob {
    /* dW[slice][tap][n][k] = sum over the slice's tokens t of A[t][n] * B[t'][k]: the weight gradient of nn.Linear (taps 1, t' = t;
     * swinir.py:69-71, common.py:184-195) or of a 3x3 nn.Conv2d (taps 9: t' = the pixel at offset (tap / 3 - 1, tap % 3 - 1) of t in its
     * H x W image, zero outside; A = dy, B = x, NHWC; common.py:104-105).  A, B: bf16 token-major [T, lda / ldb].  ones_col >= 0: column
     * ones_col of B reads as 1 (bias gradient as one more column).  out: fp32 [ks][taps][Np][Kp] partial sums (ks token slices). */
    const void* A; const void* B; float* out;
    int lda, ldb, Np, Kp, T, taps, H, W, ones_col, ks;
    int a_f32, b_f32;   /* 1: that operand is fp32 (rounded to bf16 while it is staged), 0: bf16 */
    int halo;           /* taps 9 only, H % 4 == 0, W % 8 == 0: steps are 4 x 8-pixel patches whose 6 x 10 halo of B is staged ONCE for all nine taps
                         * (same result; a sixth of the operand traffic) */
} SrTrWgradJob;
int sr_tr_wgrad(const SrTrWgradJob* jobs, int njobs, void* stream); /* jobs: HOST array, at most 8 per launch */
long long sr_tr_wgrad_out_floats(const SrTrWgradJob* j);

typedef struct SrTrAttnBwd {
    /* Backward of softmax(q k^T + bias + shift mask) v per (window, head), flash form (swinir.py:83-102, hat.py:90-107, 266-283): P is
     * recomputed from q, k, bias.  bf16 operands; q is pre-scaled.  Layouts: q, k, v, dq, dk, dv [bwin][head][N][32]; qT, kT, dOT
     * [bwin][head][32][N]; o, dO rows [bwin * Nq + tok][ldo] with head h at column 32 h; bias [heads][Nq][Nk], biasT [heads][Nk][Nq] fp32;
     * lse, delta [bwin][head][Nq] fp32 (written, then read by the second pass).  The relative_position_bias_table gradient leaves as
     * dtab_part [heads * groups * Nq / 64][Tpad] fp32: workgroup (head, group, 64 queries) sums dS over the group's windows and folds it through
     * rpi [Nq * Nk] (int32, negative entries wrap by T rows) into one table-sized partial; dtable[t][h] = sum of head h's groups * Nq / 64
     * consecutive partials (sr_tr_finalize).  Nk = 256 with toeplitz16 and groups * 4 == n_bwin (one partial per (head, window)): ONE launch with every operand
     * of the inner loops in LDS (csrc/sr_tr_attn_lds.hip) instead of the two register-only passes; same outputs. */
    const void* q; const void* qT; const void* k; const void* kT; const void* v;
    const void* o; const void* dO; const void* dOT;
    const float* bias; const float* biasT;
    void* dq; void* dk; void* dv;
    float* lse; float* delta; float* dtab_part; const int* rpi;
    int n_bwin, heads, hd_p, Nq, Nk, ldo, groups, T, Tpad;
    int toeplitz16;        /* 1: rpi is the standard window index (yq - yk + ws - 1) * (2 ws - 1) + (xq - xk + ws - 1) (hat.py:480-492, swinir.py:56-67): the fold runs on lane
                            * rotations.  Nq = Nk = 64 (8 x 8 windows, ABI v10): with it and groups * 4 == n_bwin ONE launch does everything (a wave per (window, head)); otherwise
                            * the two register passes with the index-map fold */
    int H, W, ws, shift;   /* mask geometry (shift == 0: no mask) */
    int oca_rel;           /* 1 (ABI v8; Nk = 576): bias[q][k] is a function of (ky - qy, kx - qx) only and rpi the overlapping-cross-attention index
                            * (ky - qy - 7) * 39 + (kx - qx - 7) with wrapping negatives (hat.py:494-517): pass Q then keeps the head's table in LDS beside the window's
                            * K / V / K^T fragments and folds the gradient through the index arithmetic instead of reading bias rows and rpi from memory */
    int lse_given;         /* 1 (ABI v11; with oca_rel): lse holds the FORWARD's log-sum-exp (SrTrAttnFwd.lse) and is only read: pass Q then forms P = exp(S - lse) tile by tile
                            * (two workgroups per CU instead of one wave per SIMD) and still writes delta */
} SrTrAttnBwd;
int sr_tr_attn_bwd(const SrTrAttnBwd* a, void* stream);

typedef struct SrTrAttnFwd {
    /* Training forward of HAT's overlapping cross attention core (hat.py:266-283) on the UNFOLDED keys / values: out rows
     * [bwin * Nq + tok][ldo] (head h at column 32 h) = softmax(q k^T + bias) v; q [bwin][head][Nq][32], k [bwin][head][Nk][32],
     * vT [bwin][head][32][Nk] bf16, bias [heads][Nq][Nk] fp32.  Nq 256, Nk 576. */
    const void* q; const void* k; const void* vT; const float* bias; void* out;
    int n_bwin, heads, hd_p, Nq, Nk, ldo;
    const float* bias_rel;  /* optional (ABI v8): as SrOcaAttn.bias_rel; selects the LDS form (csrc/sr_oca_lds.hip) */
    float* lse;             /* optional (ABI v8, LDS form only): log-sum-exp of every query row [bwin][head][Nq] */
} SrTrAttnFwd;
int sr_tr_attn_fwd(const SrTrAttnFwd* a, void* stream);

typedef struct SrTrOcaFold {
    /* nn.Unfold(kernel wse, stride 16, padding pad) of OCAB (hat.py:217-221,255-263) between the per-window layout k, v [bwin][head][256][32]
     * and the per-window neighbourhoods kwin, vwin [bwin][head][wse*wse][32] (+ transposes kwinT, vwinT [bwin][head][32][wse*wse]).
     * unfold = 1: k, v -> kwin, vwin, kwinT, vwinT (zeros outside the image); unfold = 0: the adjoint, k, v <- sum of the copies in kwin, vwin. */
    void* k; void* v; void* kwin; void* vwin; void* kwinT; void* vwinT;
    int B, nwy, nwx, heads, wse, pad;
} SrTrOcaFold;
int sr_tr_oca_fold(const SrTrOcaFold* a, int unfold, void* stream);

int sr_tr_block_supported(int C, int Cp, int heads, int hd_p, int ws, int Hp);

typedef struct SrTrQkvFwd {
    /* n1 = LayerNorm1(x) * gamma + beta (bf16, image order, channels 180 / 181 = 1; optional) and q, k, v = qkv(n1) for the window
     * attention kernels, each in both orientations (hat.py:164-176; swinir.py:146-160).  wstream: packing of sr_swin_qkv (18 slots) with
     * the UNFOLDED weights. */
    const float* x; const float* gamma; const float* beta; const void* wstream;
    void* q; void* qT; void* k; void* kT; void* v; void* vT; void* n1;
    int B, H, W, C, Cp, ldx, ldn, heads, hd_p, ws, shift;
    float eps;
} SrTrQkvFwd;
int sr_tr_qkv_fwd(const SrTrQkvFwd* a, void* stream);

typedef struct SrTrTailFwd {
    /* x1 = x + s_a[b] (proj(O) + bproj) + y * gate[b];  out = x1 + s_m[b] (fc2(GELU(fc1(LayerNorm2(x1)))) + b2)   (hat.py:172-194, 286-293;
     * swinir.py:169-174; s_a / s_m = DropPath's per-image scale or NULL).  x1 is stored for the backward; the gate is recomputed from the
     * CAB's pool partials (as sr_swin_tail) and optionally written to gate_out [B][Cp].  wstream: sr_swin_tail's 30 slots, unfolded weights,
     * hidden pad columns written as s_m by the kernel. */
    const float* x; float* out; float* x1; const void* o; const void* wstream; const float* bproj; const float* gamma; const float* beta;
    const void* y; const float* pool_partial; const float* ca_w1; const float* ca_b1; const float* ca_w2; const float* ca_b2; float* gate_out;
    const float* s_a; const float* s_m;
    int B, H, W, C, Cp, ldx, ldy, heads, hd_p, ws, shift, Hp, ca_Cr, ca_n_tiles;
    float eps, y_scale;
} SrTrTailFwd;
int sr_tr_tail_fwd(const SrTrTailFwd* a, void* stream);

typedef struct SrTrTailBwd {
    /* Adjoint of sr_tr_tail_fwd.  In: dout (gradient of out), x1, y, gate [B][Cp].  Out: dx1 (fp32, image order); bf16 token-major operands of
     * the weight-gradient GEMMs in WINDOW order: n2w [T][Cp] (LayerNorm2 output, ones in 180 / 181), doutw [T][Cp], gw [T][Hp] (s_m GELU(h),
     * s_m in columns 360 / 361), dhw [T][Hp], dx1sw [T][Cp] (s_a dx1); dOw [T][Cp] and dOT [bwin][head][32][ntok] = gradient of the attention
     * output; dyc (bf16, image order, ldy) = dx1 * gate; dgate_part [T / 64][Cp] = per-workgroup sums of dx1 * y; ln_part [T / 64][2][Cp] =
     * per-workgroup LayerNorm2 dgamma | dbeta.  wstream: 42 slots (fasttrain.py pack_tail_bwd). */
    const float* dout; const float* x1; const void* y; const float* gate; const float* gamma; const float* beta; const void* wstream;
    const float* s_a; const float* s_m;
    float* dx1; void* n2w; void* doutw; void* gw; void* dhw; void* dOw; void* dOT; void* dx1sw; void* dyc; float* dgate_part; float* ln_part;
    int B, H, W, C, Cp, ldx, ldy, heads, hd_p, ws, shift, Hp;
    float eps;
} SrTrTailBwd;
int sr_tr_tail_bwd(const SrTrTailBwd* a, void* stream);

typedef struct SrTrQkvBwd {
    /* Adjoint of sr_tr_qkv_fwd: dn1 = [dq | dk | dv] Wqkv (+ dn1c, the CAB branch's gradient w.r.t. n1, bf16 image order), LayerNorm1
     * backward, dx = dx1 + ...  Also writes n1w [T][Cp] (LayerNorm1 output in window order) and dqkvw [T][3 * heads * 32] (the same gradients
     * token-major): the operands of the qkv weight gradient, and ln_part [T / 64][2][Cp].  wstream: 18 slots (fasttrain.py pack_qkv_bwd). */
    const float* dx1; const float* x; const void* dq; const void* dk; const void* dv; const void* dn1c; const float* gamma; const float* beta; const void* wstream;
    float* dx; void* n1w; void* dqkvw; float* ln_part;
    int B, H, W, C, Cp, ldx, ldn, heads, hd_p, ws, shift;
    float eps;
} SrTrQkvBwd;
int sr_tr_qkv_bwd(const SrTrQkvBwd* a, void* stream);

typedef struct SrTrCaBwd {
    /* ChannelAttention backward of HAT's CAB (hat.py:25-38): from dgate_part (sr_tr_tail_bwd) the squeeze MLP's parameter gradients
     * (dparam_part [B][dparam_stride]: dw1 [Cr*C] | db1 [Cr] | dw2 [C*Cr] | db2 [C], one partial per image) and
     * dy[b][px][c] += dmean[b][c] / (H W) in place (dy = dyc of sr_tr_tail_bwd, bf16 image order). */
    const float* dgate_part; const float* pool_partial; const float* w1; const float* b1; const float* w2; const float* b2;
    void* dy; float* dparam_part;
    int B, H, W, C, Cp, Cr, n_tiles, parts, ld, dparam_stride;
    float y_scale;
} SrTrCaBwd;
int sr_tr_ca_bwd(const SrTrCaBwd* a, void* stream);

typedef struct SrTrLnBwd {
    /* nn.LayerNorm backward alone on the padded fp32 stream: dx = LN'(x)^T dy (+ dskip); ln_part [M / 64][2][Cp] dgamma | dbeta partials.
     * dy / dskip fp32 or bf16 (flags); rows in memory order, M % 64 == 0. */
    const float* x; const void* dy; const float* gamma; const void* dskip; float* dx; float* ln_part;
    long long M; int C, Cp, ld, dy_bf16, dskip_bf16; float eps;
} SrTrLnBwd;
int sr_tr_ln_bwd(const SrTrLnBwd* a, void* stream);
/* nn.PixelShuffle(r) backward on NHWC bf16 into the conv's packed row order n = (i r + j) cps + c; LeakyReLU backward from the output;
 * out = a + b (b fp32 or bf16, b_dtype = SR_*). */
int sr_tr_unshuffle(const void* src, void* dst, int B, int H, int W, int cps, int r, void* stream);
int sr_tr_lrelu_bwd(const void* dy, const void* y, void* dx, float slope, long long n, void* stream);
int sr_tr_add(const float* a, const void* b, int b_dtype, float* out, long long n, void* stream);
/* torch.optim.Adam's update (the reference Trainer's optimizer, trainer.py:133-139; L2 weight_decay, no amsgrad) on the flat fp32 parameter /
 * gradient / first- / second-moment buffers in ONE launch; step = 1-based update count (bias corrections computed in double on the host). */
int sr_tr_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps, float weight_decay, long long step, void* stream);

/* g = GELU(x) and / or dx = dg * GELU'(x), bf16 (the nn.GELU between the CAB's convolutions, hat.py:43); n elements, n % 8 == 0. */
int sr_tr_gelu(const void* x, const void* dg, void* g, void* dx, long long n, void* stream);
/* ABI v10: the positional launches of a training step with their arguments in a block (int f(const Block*, void* stream)): recordable in a launch plan.
 * Each forwards to the function of the same name without the suffix. */
typedef struct SrTrGelu { const void* x; const void* dg; void* g; void* dx; long long n; } SrTrGelu;
int sr_tr_gelu_args(const SrTrGelu* a, void* stream);
typedef struct SrTrAdd { const float* a; const void* b; float* out; long long n; int b_dtype; } SrTrAdd;
int sr_tr_add_args(const SrTrAdd* a, void* stream);
typedef struct SrTrFinalize { const float* arena; const long long* src; const int* dst; const int* stride; const int* ns; const float* scale; float* grad; long long n;
                              int lanes;  /* ABI v11: 0 / 1 = one thread per item; 8 = eight adjacent lanes share an item (slices j, j + 8, ...; their sums meet in a fixed xor tree): items
                                           * with MANY slices -- LayerNorm / bias-table partials, one per workgroup: 256 of them are 64 dependent rounds for one thread */
                            } SrTrFinalize;
int sr_tr_finalize_to_args(const SrTrFinalize* a, void* stream);
typedef struct SrTrUnshuffle { const void* src; void* dst; int B, H, W, cps, r; } SrTrUnshuffle;
int sr_tr_unshuffle_args(const SrTrUnshuffle* a, void* stream);
typedef struct SrTrLreluBwd { const void* dy; const void* y; void* dx; long long n; float slope; } SrTrLreluBwd;
int sr_tr_lrelu_bwd_args(const SrTrLreluBwd* a, void* stream);
typedef struct SrLayernorm { const float* x; void* y; const float* gamma; const float* beta; int y_dtype, M, C, Cp; float eps; } SrLayernorm;
int sr_layernorm_to_args(const SrLayernorm* a, void* stream);

/* Launch plans (ABI v10; csrc/sr_plan.cpp).  The reference's training loop (studiosr/engine/trainer.py:97-109) is a handful of ATen calls per layer; here a
 * training step is ~530 C-ABI launches whose arguments are static after the first step.  A plan is that sequence recorded once: sr_plan_create copies the
 * operations AND their argument blocks; sr_plan_run enqueues them in order, each on streams[op.stream] (slot 0 = the caller's current stream).
 *   SR_PLAN_CALL1  fn(arg, stream)            every entry point of the form int f(const Struct*, void* stream)
 *   SR_PLAN_CALL2  fn(arg, arg2, stream)      sr_hab_mid
 *   SR_PLAN_CALLI  fn(arg, ival, stream)      sr_tr_wgrad (job array, count), sr_tr_oca_fold (block, direction)
 *   SR_PLAN_EVENT_RECORD / SR_PLAN_STREAM_WAIT  hipEventRecord(event[ival], stream) / hipStreamWaitEvent(stream, event[ival]): the cross-stream edges
 * Enqueue-only like everything else here; a plan may be run any number of times, from one thread at a time. */
enum { SR_PLAN_CALL1 = 0, SR_PLAN_CALL2 = 1, SR_PLAN_CALLI = 2, SR_PLAN_EVENT_RECORD = 3, SR_PLAN_STREAM_WAIT = 4 };
typedef struct SrPlanOp {
    int kind;       /* SR_PLAN_* */
    int stream;     /* index into the stream table of sr_plan_run */
    int ival;       /* CALLI: the integer argument; EVENT_RECORD / STREAM_WAIT: event index in [0, n_events) */
    int arg_bytes;  /* bytes of *arg (copied by sr_plan_create) */
    int arg2_bytes; /* CALL2: bytes of *arg2 */
    int reserved;
    const void* fn;
    const void* arg;
    const void* arg2;
} SrPlanOp;
void* sr_plan_create(const SrPlanOp* ops, int n, int n_events);  /* NULL on error (sr_last_error) */
int sr_plan_streams(const void* plan);                           /* stream slots the plan uses */
int sr_plan_ops(const void* plan);
int sr_plan_run(const void* plan, void* const* streams, int n_streams);
void sr_plan_destroy(void* plan);

#ifdef __cplusplus
}
#endif
#endif
