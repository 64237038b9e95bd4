/* pssr_mi355.h — C ABI of libpssr_mi355.so, the MI355X (gfx950) kernels behind pssr2_amd.
 *
 * The reference (ucsdmanorlab/PSSR2) is pure Python and has no FFI layer; its hot path is the
 * torch.nn ops listed in SURVEY.md §8a.  Each entry point below replaces one (or a fused group) of
 * those ops; the reference file:line it stands for is cited next to it.  Conventions:
 *   - every pointer is a DEVICE pointer unless marked "host"; the caller owns all memory,
 *     the library allocates nothing persistent;
 *   - every call takes the HIP stream to launch on (pssr_stream_t == hipStream_t) and never
 *     synchronises it;
 *   - return value: 0 on success, a negative PSSR_ERR_* otherwise (never throws across the ABI);
 *     pssr_last_error() returns a thread-local message for the last failure;
 *   - activations are NHWC ("pixel-major") with an explicit channel stride so that an op can read
 *     or write a channel slice of a wider buffer (this is how torch.cat is elided);
 *   - dtype of activations / packed weights: PSSR_F32 (exact-f32 MFMA path, parity), PSSR_BF16 or PSSR_F16
 *     (16-bit storage, f32 accumulate; fp16 needs loss scaling by the caller).  Parameters, statistics and
 *     gradients are f32/f64.
 */
#ifndef PSSR_MI355_H
#define PSSR_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pssr_stream_t; /* hipStream_t */

enum { PSSR_F32 = 0, PSSR_BF16 = 1, PSSR_F16 = 2 };

enum {
    PSSR_OK = 0,
    PSSR_ERR_ARG = -1,     /* invalid shape / alignment / enum */
    PSSR_ERR_LAUNCH = -2,  /* HIP launch failure */
    PSSR_ERR_UNSUPPORTED = -3
};

int pssr_abi_version(void);
const char* pssr_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Weight packing.  torch keeps Conv2d weights OIHW f32 (pssr/models/_blocks.py:10-11,28,35); the
 * conv kernels consume a K-chunked, LDS-image-ordered copy in the compute dtype:
 *   packed[chunk][tap][n (padded to 128)][32 bytes = 2 swizzled 16-byte slots of K]
 * mode 0 (forward):  GEMM-K = input channels  [ci_begin, ci_begin+ci_count) of the OIHW tensor,
 *                    GEMM-N = output channels (optionally permuted by `n_perm`: n_src = n_perm[n]).
 * mode 1 (dgrad):    GEMM-K = output channels, GEMM-N = input channels of the range, taps flipped.
 * mode 2 (flat-K):   a KxK conv seen as 1x1 over im2col'ed channels: K index = (ci-ci_begin)*ks*ks + tap.
 * mode 3 (flat-K dgrad): the transpose of mode 2: GEMM-K = output channels, GEMM-N = flat index.
 * mode 4 (space-to-depth): a KxK stride-K conv (RDNet transition, pssr/models/_rdnet.py:54-62) seen as 1x1 over the
 *                    space-to-depth input written by pssr_layernorm2d_fwd(s2d=1): K index = tap*cpad + ci,
 *                    cpad = cin rounded up to 16.   mode 5: its transpose (dgrad).
 * `k_pad` = GEMM-K rounded up to 16, `n_pad` = GEMM-N rounded up to 128 (zero filled).
 */
int pssr_pack_conv_weight(const float* w_oihw, void* packed, int cout, int cin, int ks,
                          int ci_begin, int ci_count, int mode, const int32_t* n_perm,
                          int k_pad, int n_pad, int dtype, pssr_stream_t stream);
/* The same for many weights in one launch (a training step re-packs every conv weight, forward and dgrad form).
 * `items_dev` is a DEVICE array; fields as the arguments above; `center` != 0 (modes 2/3, ks = 3): `w_oihw` is a 1x1
 * weight [cout][cin] standing for the centre tap of a 3x3 kernel (ResBlock.respass consumed through the im2col'ed input). */
typedef struct pssr_pack_item {
    const float* w; void* packed; const int32_t* n_perm;
    int32_t cout, cin, ks, ci_begin, ci_count, mode, k_pad, n_pad, dtype, center;
} pssr_pack_item;
int pssr_pack_conv_weight_batch(const pssr_pack_item* items_dev, int n_items, pssr_stream_t stream);
/* bytes needed for a packed weight */
int64_t pssr_packed_weight_bytes(int taps, int k_pad, int n_pad, int dtype);

/* Inverse of mode 0 / mode 2 for gradients: wgrad writes dW as f32 [n][tap][k_pad]; this scatters
 * (adds when `accumulate`) into the OIHW f32 gradient of the parameter. */
int pssr_unpack_conv_wgrad(const float* dw_packed, float* dw_oihw, int cout, int cin, int ks,
                           int ci_begin, int ci_count, int mode, const int32_t* n_perm,
                           int k_pad, int accumulate, pssr_stream_t stream);
/* same, summing `parts` partial slabs laid out [parts][rows][taps][k_pad] (rows = the wgrad's cout) first */
int pssr_unpack_conv_wgrad_parts(const float* dw_parts, int parts, int rows, float* dw_oihw, int cout, int cin, int ks,
                                 int ci_begin, int ci_count, int mode, const int32_t* n_perm,
                                 int k_pad, int accumulate, pssr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution, stride 1, "same" zero padding, NHWC, MFMA 32x32 tiles.
 * Replaces nn.Conv2d forward (pssr/models/_blocks.py:16-17,39-40) and, with mode-1 packed
 * weights, its input-gradient; BatchNorm2d+ReLU of the *previous* layer is applied while the
 * input tile is staged (prologue), and BatchNorm statistics / residual tail / ReLU-mask are
 * applied while the output tile is written (epilogue), so no normalised tensor is materialised.
 */
enum { PSSR_PRO_NONE = 0, PSSR_PRO_BN_RELU = 1,
       PSSR_PRO_GELU = 2 /* in = gelu(in): nn.GELU between the two 1x1 convs of an RDNet block, pssr/models/_rdnet.py:185,200 */ };
enum {
    PSSR_EPI_STORE = 0,      /* out = acc + bias                                             */
    PSSR_EPI_TAIL = 1,       /* out = relu(acc + bias + aux*aux_scale + aux_shift)  (ResBlock tail, _blocks.py:40) */
    PSSR_EPI_DGRAD_MASK = 2, /* out = (aux*aux_scale+aux_shift > 0) ? acc : 0   (ReLU backward) */
    PSSR_EPI_DGRAD_GELU = 4, /* out = acc * gelu'(aux)   (GELU backward; with FLAG_STATS stats[0..cout) += sum out)  */
    PSSR_EPI_HEADQ = 5,      /* inference form of Reconstruction (_blocks.py:15-18) for 64 hidden channels and ONE output channel: the
                              * epilogue of `pre` (FLAG_RELU implied, channels sub-pixel major, cout = 16 * 64) does not store its
                              * activation but the nine per-tap dot products of relu(acc + bias) with Reconstruction.conv's weights:
                              * head_q[tap][sub][n][h][w] (f32) for sub-pixel sub of low-resolution pixel (n, y, x);
                              * pssr_head_q_gather sums, for every output pixel, the nine products of its shifted neighbours.
                              * 16-bit storage, the conv_v3 tiles (h, w >= 16); PSSR_ERR_UNSUPPORTED otherwise */
    PSSR_EPI_FINAL = 3       /* out_f32_nchw = (acc + bias)*out_scale + out_shift; any cout <= 32
                                (Reconstruction.conv + "x*128+128", _blocks.py:17, resunet.py:95)  */
};
/* Per-channel f64 statistics are accumulated into PSSR_STAT_STRIPES interleaved copies
 * (stats[stripe][2*C], stripe = workgroup index mod PSSR_STAT_STRIPES) so that thousands of workgroups do
 * not serialise on 2*C addresses; the finalisers sum the stripes. */
#define PSSR_STAT_STRIPES 32
/* Every statistic buffer has PSSR_STAT_ROWS = 2 x PSSR_STAT_STRIPES rows: a workgroup adds each of its f32 partial sums v to stripe
 * s = workgroup index mod PSSR_STAT_STRIPES as TWO exact pieces -- row s gets v rounded to a multiple of 2^-20, row
 * PSSR_STAT_STRIPES + s the remainder rounded to a multiple of 2^-64.  Sums of such pieces are exact in f64 (no rounding, hence
 * independent of the order in which the atomics arrive) while a row stays below 2^33 resp. holds fewer than 2^10 addends; the
 * consumers (pssr_bn_finalize, pssr_bn_bwd_coefs, pssr_f64_to_f32, ...) add the rows in a fixed order.  Training steps are therefore
 * bit-reproducible; beyond those magnitudes the sums merely round like ordinary f64 atomics again. */
#define PSSR_STAT_ROWS (2 * PSSR_STAT_STRIPES)

enum {
    PSSR_FLAG_RELU = 1,   /* EPI_STORE: relu after bias (Reconstruction.pre, _blocks.py:16)   */
    PSSR_FLAG_STATS = 2,  /* accumulate per-channel f64 sums into `stats`:
                             EPI_STORE: [sum v, sum v^2]; EPI_DGRAD_MASK: [sum g, sum g*xhat] */
    PSSR_FLAG_HEADQ = 8,  /* EPI_STORE (with FLAG_RELU): store the activation AND the tap planes of PSSR_EPI_HEADQ (same conditions, head_w /
                             head_q set): the training form -- `pre`'s activation is kept for the backward pass, Reconstruction.conv's
                             forward still needs no pass over it */
    PSSR_FLAG_AFFINE = 4, /* EPI_STORE, 16-bit storage, not with FLAG_STATS: out = (acc + bias) * aux_scale + aux_shift (then FLAG_RELU):
                             an eval-mode BatchNorm (+ ReLU) applied by the PRODUCING convolution on its f32 accumulators
                             (_blocks.py:28-32 in eval mode), so that the next layer's loader needs no prologue */
    PSSR_FLAG_SOLO = 32,  /* hint, any epilogue: nothing else runs beside this launch (a forward pass on one stream): tilings that fill the
                             chip by themselves are preferred -- 128 x 64 tiles where 128 x 128 would leave one workgroup per CU.  Launches
                             of a backward pass, which share the chip with the weight-gradient stream, measured better without it */
    PSSR_FLAG_SHUF2 = 16  /* EPI_STORE, 16-bit storage, cout % 32 == 0, not with FLAG_STATS / FLAG_HEADQ: F.pixel_shuffle(out, 2)
                             (resunet.py:82) done by the store.  `out` is the [n, 2h, 2w, out_cstride] buffer the shuffled map belongs
                             into (channels out_coff .. out_coff + cout / 4); the weights (and bias) arrive with their output channels
                             in sub-pixel-major order (packed row s * cout / 4 + c = torch channel 4 c + s, s = 2 i + j), and the
                             8-channel piece c of sub-pixel (i, j) of pixel (y, x) goes to pixel (2 y + i, 2 x + j), channel c */
};

typedef struct pssr_conv_desc {
    int32_t dtype;
    int32_t n, h, w;            /* batch and spatial size (input == output)                    */
    /* up to two input sources accumulated into one output; source 1 is optional (cin1 == 0)    */
    const void* in0; int32_t in0_cstride, in0_coff, cin0, taps0;   /* taps: 9 (3x3) or 1 (1x1)  */
    const void* w0;             /* packed (pssr_pack_conv_weight), K padded to 16               */
    const void* in1; int32_t in1_cstride, in1_coff, cin1, taps1;
    const void* w1;
    int32_t prologue;           /* applies to source 0 only                                    */
    const float* pro_scale;     /* [cin0] gamma*invstd                                         */
    const float* pro_shift;     /* [cin0] beta - mean*gamma*invstd                             */
    void* out; int32_t out_cstride, out_coff, cout, n_pad;
    const float* bias;          /* [cout] or NULL                                              */
    int32_t epilogue, flags;
    const void* aux; int32_t aux_cstride, aux_coff;
    const float* aux_scale; const float* aux_shift;      /* [cout]                              */
    const float* aux_mean; const float* aux_invstd;      /* [cout] (DGRAD_MASK + STATS)         */
    double* stats;              /* [PSSR_STAT_ROWS][2*cout], caller-zeroed                     */
    /* "blocked" pixel order (log2 r, 0 = plain NHWC): pixel (y,x) of an r-times upsampled image
     * lives at ((y/r*W/r + x/r)*r*r + (y%r)*r + x%r), i.e. F.pixel_shuffle (_blocks.py:17) of an
     * NHWC tensor whose channels were ordered sub-pixel-major needs no data movement at all.    */
    int32_t in0_blk, out_blk, aux_blk;
    float out_scale, out_shift; /* EPI_FINAL only                                              */
    /* optional scratch for split-K (layers whose tiles do not fill the chip): pssr_conv2d_workspace_bytes(desc) bytes,
     * uninitialised, caller-owned; NULL (or too small) simply disables the split                                   */
    void* workspace; int64_t workspace_bytes;
    /* EPI_HEADQ only (appended in ABI version 3; other epilogues never read them): Reconstruction.conv's weight [1][64][3][3] f32 and
     * the tap products [9][16][n][h][w] f32 */
    const float* head_w; float* head_q;
} pssr_conv_desc;

int pssr_conv2d(const pssr_conv_desc* desc, pssr_stream_t stream);
/* bytes of `workspace` with which pssr_conv2d would split K for this shape (0: no split); negative = PSSR_ERR_*.  Pointers
 * in `desc` are ignored. */
int64_t pssr_conv2d_workspace_bytes(const pssr_conv_desc* desc);
/* Deprecated (ABI version 1): selected the round-1 pipelined 256-pixel loop, which conv_v3_kernel replaced (tunable IGEMM_V3
 * below).  Kept so that old bindings load; ignores its argument and returns 0. */
int pssr_conv2d_pipeline_mode(int mode);

/* Kernel-selection tunables (no reference counterpart: the reference has no native code).  ONE process-wide table, filled once
 * from the environment variables PSSR_<NAME> on first use and changed afterwards only through pssr_set_option(); no launch
 * path reads the environment.  Names: IGEMM_V3 (1: LDS-DMA / counted-wait 3x3 loop for 16-bit layers with > 64 output
 * channels on >= 16x16 images when its 256-pixel tiles fill the chip; 2: whenever the shape allows; 0: the 128-pixel loop), IGEMM_FLAT, IGEMM_BIG, IGEMM_KSPLIT, CONV_EPI8, WGRAD_LEAN,
 * WGRAD_DMA, WGRAD_BLOCKS, WGRAD_BLOCKS_1X1, DWCONV_TILE, DWWG_BLOCKS, LN_BWD_BLOCKS (and the diagnostic IGEMM_DBG, LN_DBG,
 * V3_LDS_PAD: 0 in production).  pssr_set_option returns the previous value (>= 0) or PSSR_ERR_ARG
 * for an unknown name / out-of-range value; pssr_get_option returns the value or PSSR_ERR_ARG. */
int pssr_set_option(const char* name, int value);
int pssr_get_option(const char* name);

/* Weight gradient of the same convolution (autograd of nn.Conv2d.weight):
 *   dw[n][tap][k] = sum_pixels dy[p][n] * prologue(in)[p + tap][k]      (f32)
 * The pixel reduction is split over workgroups.  dw_parts > 0 (preferred): the caller provides
 * dw[dw_parts][cout][taps][cin_pad] (no initialisation needed), dw_parts = pssr_conv2d_wgrad_parts(desc); every
 * workgroup stores its partial slab once and pssr_unpack_conv_wgrad_parts sums the parts while scattering to OIHW.
 * dw_parts == 0: one caller-zeroed slab, partial sums combined with f32 atomics, then pssr_unpack_conv_wgrad.   */
typedef struct pssr_wgrad_desc {
    int32_t dtype;
    int32_t n, h, w;
    const void* dy; int32_t dy_cstride, dy_coff, dy_blk, cout;   /* cout*elemsize % 16 == 0       */
    const void* in; int32_t in_cstride, in_coff, in_blk, cin_pad; /* cin_pad % 16 == 0            */
    int32_t taps;
    int32_t prologue; const float* pro_scale; const float* pro_shift;
    float* dw;                                                   /* [max(dw_parts,1)][cout][taps][cin_pad] */
    int32_t dw_parts;
} pssr_wgrad_desc;

int pssr_conv2d_wgrad(const pssr_wgrad_desc* desc, pssr_stream_t stream);
/* number of partial slabs this shape is split into (> 0), or a negative PSSR_ERR_*; pointers in `desc` are ignored */
int pssr_conv2d_wgrad_parts(const pssr_wgrad_desc* desc);


/* ---------------------------------------------------------------------------------------------
 * Per-channel and pointwise kernels around the convolutions (all HBM-bound).
 * NHWC tensor slices are passed as (pointer, channel stride, channel offset); C % 4 == 0.
 */

/* (statistic buffer of PSSR_STAT_ROWS rows, see there) stats[0:c] += sum, stats[c:2c] += sum of squares of (x*pre_scale + pre_shift) over N,H,W of an
 * NCHW f32 tensor: batch statistics of ResUNet.norm on "x/128-1" (pssr/models/resunet.py:66-68). */
int pssr_channel_stats_nchw(const float* x, int n, int c, int64_t hw, float pre_scale, float pre_shift,
                            double* stats, pssr_stream_t stream);

/* nn.BatchNorm2d training-mode bookkeeping (pssr/models/_blocks.py:31; torch defaults eps=1e-5,
 * momentum=.1): from [sum, sumsq] -> scale=gamma*invstd, shift=beta-mean*scale, mean, invstd and the
 * running_mean / running_var update (unbiased variance).  running_* may both be NULL. */
int pssr_bn_finalize(const double* stats, double count, const float* gamma, const float* beta, float eps,
                     float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                     float* mean, float* invstd, int c, pssr_stream_t stream);
/* eval mode: scale/shift from the running statistics */
int pssr_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* scale, float* shift, int c,
                        pssr_stream_t stream);
/* a = relu(scale * y + shift) in the storage type (16-bit, power-of-two channel count): the activated input of the next convolution of a
 * ResBlock (pssr/models/_blocks.py:26-41, nn.BatchNorm2d + nn.ReLU) written out once per layer, bit-identical to what the convolution
 * loaders' BatchNorm+ReLU prologue stages, so that pssr_conv2d_wgrad can read it without a prologue (LDS-DMA kernel). */
int pssr_bn_relu_apply(const void* y, int y_cs, int y_co, const float* scale, const float* shift, void* a, int a_cs,
                       int a_co, int64_t npix, int c, int dtype, pssr_stream_t stream);
/* BatchNorm backward: with stats = [sum g, sum g*xhat] the input gradient is A*g + B*y + C per channel;
 * also emits dgamma = sum g*xhat and dbeta = sum g (either may be NULL). */
int pssr_bn_bwd_coefs(const double* stats, double count, const float* gamma, const float* mean,
                      const float* invstd, float* coef_a, float* coef_b, float* coef_c, float* dgamma,
                      float* dbeta, int c, pssr_stream_t stream);
int pssr_bn_bwd_apply(const void* g, int g_cs, int g_co, const void* y, int y_cs, int y_co,
                      const float* coef_a, const float* coef_b, const float* coef_c,
                      void* dy, int dy_cs, int dy_co, int64_t npix, int c, int dtype, pssr_stream_t stream);

/* Input stage of ResUNet.forward (resunet.py:66-70) fused with the im2col of the 3x3 neighbourhood:
 * xcol[n,y,x, ch*9+tap] = ((x[n,ch,y+ky-1,x+kx-1]*pre_scale+pre_shift)*scale[ch]+shift[ch]), zero
 * outside the image, channels [9c, xc) zero.  The first conv, the first respass and the input
 * channel of Reconstruction.pre then run as flat-K 1x1 convolutions over xcol. */
int pssr_input_im2col(const float* x_nchw, void* xcol, int n, int c, int h, int w, int xc,
                      float pre_scale, float pre_shift, const float* scale, const float* shift,
                      int dtype, pssr_stream_t stream);
/* backward of the above for the parameters of ResUNet.norm: folds d(xcol) (two optional sources)
 * onto the normalised input and accumulates stats = [sum dx0, sum dx0*xhat0] per input channel. */
int pssr_input_norm_bwd(const void* dxcol_a, const void* dxcol_b, int xc, const float* x_nchw,
                        float pre_scale, float pre_shift, const float* mean, const float* invstd,
                        int n, int c, int h, int w, double* stats, int dtype, pssr_stream_t stream);

/* F.max_pool2d(x, 2) (resunet.py:76) and its gradient; the gradient goes to the first maximum of
 * each window and is added to `dskip` (the skip-connection gradient of the same tensor, may be NULL). */
/* Same with a third, optional gradient source: the patchify stem of RDNet (pssr_input_patchify below).  Any of
 * dxcol_a / dxcol_b / dpatch may be NULL (not all three). */
int pssr_input_norm_bwd2(const void* dxcol_a, const void* dxcol_b, int xc, const void* dpatch, int pc, int patch,
                         const float* x_nchw, float pre_scale, float pre_shift, const float* mean, const float* invstd,
                         int n, int c, int h, int w, double* stats, int dtype, pssr_stream_t stream);

int pssr_maxpool2(const void* in, int in_cs, int in_co, void* out, int out_cs, int out_co,
                  int n, int h, int w, int c, int dtype, pssr_stream_t stream);
int pssr_maxpool2_bwd(const void* act, int act_cs, int act_co, const void* dpool, int dp_cs, int dp_co,
                      const void* dskip, int ds_cs, int ds_co, void* dout, int do_cs, int do_co,
                      int n, int h, int w, int c, int dtype, pssr_stream_t stream);

/* F.pixel_shuffle(x, r) (resunet.py:82) written straight into a channel slice of the concat buffer
 * (torch.cat at resunet.py:84 is never materialised separately); inverse=1 is the gradient. */
int pssr_pixel_shuffle(const void* lo, int lo_cs, int lo_co, void* hi, int hi_cs, int hi_co,
                       int n, int h, int w, int c_hi, int r, int inverse, int dtype, pssr_stream_t stream);

/* backward of "relu(bn(y) + r)" (ResBlock tail, _blocks.py:40): dz = dout*(out>0) and
 * stats += [sum dz, sum dz*xhat(y)] for the BatchNorm in front of the addition. */
int pssr_relu_bwd_stats(const void* dout, int do_cs, int do_co, const void* out, int o_cs, int o_co,
                        const void* y, int y_cs, int y_co, const float* mean, const float* invstd,
                        void* dz, int dz_cs, int dz_co, double* stats, int64_t npix, int c, int dtype,
                        pssr_stream_t stream);

/* The same with the gradient of the block output formed in the loader instead of read from a tensor (16-bit storage, c a power of two;
 * PSSR_ERR_UNSUPPORTED otherwise -- the caller then runs the separate kernels):
 *   _pool:      d(out) = dskip + route(dpool): the block output feeds F.max_pool2d(x, 2) (resunet.py:76, first maximum in row-major
 *               window order takes the gradient, as torch) and the decoder's skip connection (resunet.py:84); h, w even
 *   _unshuffle: d(out)[n, y, x, 4 ch + 2 i + j] = dhi[n, 2 y + i, 2 x + j, ch]: the block output went through F.pixel_shuffle(x, 2)
 *               (resunet.py:82) into the first c / 4 channels of the next level's concat buffer; c >= 32
 * dz is bit for bit what pssr_maxpool2_bwd / pssr_pixel_shuffle(inverse) followed by pssr_relu_bwd_stats write. */
int pssr_relu_bwd_stats_pool(const void* dpool, int dp_cs, int dp_co, const void* dskip, int ds_cs, int ds_co,
                             const void* out, int o_cs, int o_co, const void* y, int y_cs, int y_co,
                             const float* mean, const float* invstd, void* dz, int dz_cs, int dz_co, double* stats,
                             int n, int h, int w, int c, int dtype, pssr_stream_t stream);
int pssr_relu_bwd_stats_unshuffle(const void* dhi, int dh_cs, int dh_co, const void* out, int o_cs, int o_co,
                                  const void* y, int y_cs, int y_co, const float* mean, const float* invstd,
                                  void* dz, int dz_cs, int dz_co, double* stats, int n, int h, int w, int c, int dtype,
                                  pssr_stream_t stream);

/* out[row][c] += sum over pixels (bias gradients); out is a statistic buffer of PSSR_STAT_ROWS x c doubles */
int pssr_channel_sum_nhwc(const void* x, int cs, int co, int64_t npix, int c, double* out, int dtype,
                          pssr_stream_t stream);
/* f32 NCHW -> NHWC in the compute dtype, scaled, channels [c, out_cs) zero (gradient of the output) */
int pssr_nchw_to_nhwc(const float* in, void* out, int n, int c, int64_t hw, int out_cs, float scale,
                      int dtype, pssr_stream_t stream);
/* np.clip(x, 0, 255).astype(np.uint8): truncation toward zero (pssr/predict.py:245-246) */
int pssr_clip_u8(const float* in, uint8_t* out, int64_t n, pssr_stream_t stream);
/* out[i] (+)= sum over `stripes` copies in[k*n + i] */
int pssr_f64_to_f32(const double* in, float* out, int n, int accumulate, int stripes, pssr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SSIM / MS-SSIM + Gaussian-L1 loss (pssr/util.py:10-52; pytorch_msssim 1.0.0 algorithm) on f32
 * NCHW planes [planes = N*C][h][w].  `win_host` is a HOST array of k (odd, <= 33) filter taps.
 * One level forward: sums[plane*2+0] += sum of the cs map, sums[plane*2+1] += sum of the ssim map
 * over the (h-k+1)x(w-k+1) valid region; optional l1_sum += sum_q |x-y|(q) * S(q), S = zero-padded
 * window mass (the mean over the padded G (*) |x-y| map of pssr/util.py:50 without materialising it). */
int pssr_ssim_level_fwd(const float* x, const float* y, int planes, int h, int w, const float* win_host,
                        int k, float c1, float c2, double* sums, double* l1_sum, pssr_stream_t stream);
/* F.avg_pool2d(kernel 2, padding = size % 2) between MS-SSIM levels */
int pssr_avgpool2_planes(const float* in, float* out, int planes, int h, int w, pssr_stream_t stream);
/* loss value and the per-plane, per-level upstream weights of the map elements (device side, no
 * host sync): ms=1 -> prod_l relu(v_l)^w_l with v_l = cs mean (ssim mean at the last level);
 * ms=0 -> plain SSIM mean.  loss = mix*(1-mean) + (1-mix)*l1 (l1_sum may be NULL when mix == 1). */
int pssr_msssim_weights(const double* sums, int levels, int planes, const double* nvalid,
                        const float* level_weights, int ms, float mix, const double* l1_sum,
                        double l1_numel, const float* grad_out, float* loss_out, float* wts,
                        float* l1_coef, pssr_stream_t stream);
/* gradient of one level wrt x: transposed Gaussian filtering of the three adjoint maps, plus the
 * avg-pool gradient from the coarser level (dcoarse, may be NULL) and the L1 term (l1_coef, device
 * scalar, may be NULL). */
int pssr_ssim_level_bwd(const float* x, const float* y, int planes, int h, int w, const float* win_host,
                        int k, float c1, float c2, const float* wts, int use_ssim, const float* dcoarse,
                        int hc, int wc, const float* l1_coef, float* dx, pssr_stream_t stream);

/* The same pair with the per-position derivatives kept between the passes (training; 11-tap window only): the forward also stores
 * `adj` = [planes][3][h][w] f32 -- d(cs or ssim map)/d(mu_x, E[x^2], E[xy]) at every valid position, per unit of upstream weight
 * (use_ssim: the last MS-SSIM level / plain SSIM) -- and the backward filters those three maps instead of recomputing the five
 * forward maps on a 20-pixel-wider halo first (78 k instead of 295 k multiply-adds per 32 x 32 tile).  Same values as the pair above
 * up to the rounding of one reassociated product. */
int pssr_ssim_level_fwd_adj(const float* x, const float* y, float in_div, int planes, int h, int w, const float* win_host, int k,
                            float c1, float c2, int use_ssim, double* sums, double* l1_sum, int stripes, int64_t stripe_stride,
                            float* adj, pssr_stream_t stream);
/* `sums` / `l1_sum` of pssr_ssim_level_fwd_adj are 2 x `stripes` copies `stripe_stride` doubles apart (a workgroup adds the two exact
 * pieces of each of its sums -- see PSSR_STAT_ROWS -- to copy s and copy stripes + s:
 * a 512^2 x 32 level ends with 8192 workgroups); pssr_msssim_weights_striped folds them in a fixed order into `folded`
 * ([levels * planes * 2 + 1] doubles) before doing what pssr_msssim_weights does. */
int pssr_msssim_weights_striped(const double* sums, int stripes, int64_t stripe_stride, double* folded, int levels, int planes,
                                const double* nvalid, const float* level_weights, int ms, float mix, const double* l1_sum,
                                double l1_numel, const float* grad_out, float* loss_out, float* wts, float* l1_coef,
                                pssr_stream_t stream);
int pssr_ssim_level_bwd_adj(const float* x, const float* y, float in_div, const float* adj, int planes, int h, int w,
                            const float* win_host, int k, const float* wts, const float* dcoarse, int hc, int wc,
                            const float* l1_coef, float* dx, pssr_stream_t stream);
/* in_div (1 = none): the pair works on x / in_div and y / in_div (as torch evaluates `hr_hat / 255` of pssr/train.py:101: a
 * multiplication by the f32 reciprocal)
 * without those tensors existing, and dx is the gradient wrt the UNdivided x; pssr_avgpool2_planes_div makes the next level's inputs
 * from the undivided level-0 maps the same way. */
int pssr_avgpool2_planes_div(const float* in, float in_div, float* out, int planes, int h, int w, pssr_stream_t stream);
/* both pyramids of the loss (prediction and target) in one launch */
int pssr_avgpool2_pair_div(const float* x, const float* y, float in_div, float* xo, float* yo, int planes, int h, int w,
                           pssr_stream_t stream);

/* torch.optim.AdamW step (decoupled weight decay) over flat f32 buffers; `step` is 1-based. */
int pssr_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                    pssr_stream_t stream);
/* hipGraph-safe variant: `state` is a device int64[2] = {step, lr as f32 bits}; the call increments the
 * step on the stream and the update kernel reads both from memory (nothing is frozen at capture). */
int pssr_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, int64_t* state,
                        float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                        pssr_stream_t stream);
/* Dynamic loss scaling without the host (fp16 storage, BASELINE config 4; the reference trains in fp32: no upstream counterpart;
 * policy = torch.amp.GradScaler's).  `amp` is a device int32[4]: [0] loss scale (f32 bits) -- the caller multiplies the loss by it --,
 * [1] good steps in a row, [2] steps skipped, [3] a gradient of the current step is not finite.
 *   pssr_amp_check       sets amp[3] when any of the n gradients is inf / NaN.
 *   pssr_adamw_step_amp  pssr_adamw_step_dev with the gradients divided by the scale; when amp[3] is set neither the step count nor a
 *                        parameter moves; afterwards the scale is multiplied by `backoff` (skipped step) or, every `interval` good
 *                        steps, by `growth`, and amp[3] is cleared.  Everything in stream order: a whole fp16 step replays as a graph. */
int pssr_amp_check(const float* g, int64_t n, int32_t* amp, pssr_stream_t stream);
int pssr_adamw_step_amp(float* p, const float* g, float* m, float* v, int64_t n, int64_t* state,
                        float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                        int32_t* amp, float growth, float backoff, int interval, pssr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Pair generation / crappifiers (pssr/data.py:471-495, pssr/crappifiers.py).
 */
/* PIL.Image.resize(BILINEAR) of uint8 planes [planes][H][W] -> [planes][h][w], bit-exact with Pillow's
 * two-pass 22-bit fixed-point resampler (pssr/data.py:483).  `tmp` holds [planes][H][w] bytes. */
int pssr_bilinear_down_u8(const uint8_t* hr, uint8_t* tmp, uint8_t* lr, int planes, int H, int W,
                          int h, int w, pssr_stream_t stream);
int pssr_u8_to_f32(const uint8_t* in, float* out, int64_t n, pssr_stream_t stream);
/* flags: 1 = clip to [0,255] (MultiCrappifier, crappifiers.py:41-42); 2 = np.round (half-to-even) then
 * clip (data.py:487).  Noise comes from Philox4x32-10 keyed by (seed, tile_offset + tile) with the pixel
 * index as counter, so results do not depend on batching or on the number of GPUs.
 * AdditiveGaussian (crappifiers.py:62-64): out = in + N(gain, sigma), sigma = max(N(intensity, spread), 0)
 * per tile; `noise` (f64, may be NULL) injects a pre-drawn field instead (exact-parity tests). */
int pssr_crappify_gaussian(const float* in, float* out, int tiles, int64_t per_tile, float intensity,
                           float gain, float spread, uint64_t seed, uint64_t tile_offset,
                           const double* noise, int flags, const uint64_t* tile_counter,
                           pssr_stream_t stream);
/* Poisson (crappifiers.py:81-86): out = x*(1-i) + Poisson(max(x,0))*i + gain */
int pssr_crappify_poisson(const float* in, float* out, int tiles, int64_t per_tile, float intensity,
                          float gain, float spread, uint64_t seed, uint64_t tile_offset, int flags,
                          const uint64_t* tile_counter, pssr_stream_t stream);
/* The same mix with the Poisson samples handed in (f64 [n], e.g. numpy's legacy-stream draws of the reference run): numpy's
 * arithmetic of pssr/crappifiers.py:81-86 exactly (x*(1-i) in float32, y*i and the sums in float64), then flags (pssr/data.py:487).
 * For exact-parity tests, like `noise` of pssr_crappify_gaussian; `intensity` / `gain` are the Python floats, undrawn. */
int pssr_crappify_poisson_samples(const float* in, const double* samples, float* out, int64_t n, double intensity,
                                  double gain, int flags, pssr_stream_t stream);
/* `tile_counter` (device, may be NULL) is added to tile_offset inside the kernel, so a captured hipGraph
 * draws fresh noise on every replay; pssr_counter_add advances it on the stream. */
int pssr_counter_add(uint64_t* counter, uint64_t inc, pssr_stream_t stream);
/* Blur (crappifiers.py:122-124): per-plane separable Gaussian, edge replicate, radius int(4*sigma+.5) */
int pssr_gaussian_blur(const float* in, float* tmp, float* out, int planes, int h, int w, float sigma,
                       float gain, int flags, pssr_stream_t stream);

/* SaltPepper (pssr/crappifiers.py:88-105): clip(x + gain, 0, 255), then `amount` (a fraction, per-tile max(N(amount, spread), 0)) of
 * the pixels becomes 255 or 0 with equal probability; Philox stream as the other noises; flags as pssr_crappify_gaussian. */
int pssr_crappify_saltpepper(const float* in, float* out, int tiles, int64_t per_tile, float amount, float gain, float spread,
                             uint64_t seed, uint64_t tile_offset, int flags, const uint64_t* tile_counter, pssr_stream_t stream);
/* Blur(spread > 0) (pssr/crappifiers.py:107-124): every tile of planes_per_tile frames is blurred with its own
 * sigma = max(N(sigma, spread), 0) drawn from the tile's Philox stream (sigma <= 0: the tile is only offset by gain). */
int pssr_gaussian_blur_tiles(const float* in, float* tmp, float* out, int tiles, int planes_per_tile, int h, int w, float sigma,
                             float spread, float gain, uint64_t seed, uint64_t tile_offset, int flags, const uint64_t* tile_counter,
                             pssr_stream_t stream);
/* Geometry of _gen_pair (pssr/data.py:471-482) for a batch of uint8 stacks resident in HBM: centred square crop to at most
 * `res`, np.pad(mode="reflect") up to res at the bottom / right, np.rot90 in the image plane when rot != 0, np.flip along
 * flip_axis (0 frames, 1 rows, 2 columns, 3 rows and columns, -1 none).  The random draws (rot, flip_axis) stay on the host in the reference's
 * order; out is uint8 [n_items][c][res][res]. */
typedef struct pssr_gather_item { const uint8_t* src; int sh, sw, rot, flip_axis; } pssr_gather_item;
int pssr_gen_pair_geometry_u8(const pssr_gather_item* items_dev, int n_items, uint8_t* out, int c, int res, pssr_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * Atrous / PSP-pooling model variants (pssr/models/_blocks.py:43-92 ResBlockA, PSP_Pooling).
 * A dilated 3x3 convolution (Conv2d(kernel_size=3, padding="same", dilation=d), _blocks.py:55) = pssr_im2col_dil + the 1x1
 * kernels of pssr_conv2d / pssr_conv2d_wgrad over K = 9 * cp (weights packed with mode 4; mode 5 for the input gradient) +
 * pssr_col2im_dil.  Tensors are NHWC slices (pointer, channel stride, channel offset) in the compute dtype.
 */
/* out[n, y, x, ci] = x_nchw * pre_scale + pre_shift (ci < c), 0 for c <= ci < out_cs: the network input "x / 128 - 1"
 * (resunet.py:66) of the atrous models, which have no input BatchNorm (resunet.py:50) */
int pssr_input_plain(const float* x_nchw, void* out, int n, int c, int h, int w, int out_cs, float pre_scale, float pre_shift,
                     int dtype, pssr_stream_t stream);
/* col[pixel][t * cp + ci] = act(in[pixel + ((t / 3 - 1) * dil, (t % 3 - 1) * dil)][ci]), zero outside the image and for
 * c <= ci < cp; act = relu(v * scale[ci] + shift[ci]) rounded to the compute dtype (the pre-activation BatchNorm + ReLU of
 * ResBlockA, _blocks.py:52-55) or the identity when scale == shift == NULL.  col: [n][h][w][9 * cp]. */
int pssr_im2col_dil(const void* in, int in_cs, int in_co, int c, const float* scale, const float* shift, void* col, int cp,
                    int n, int h, int w, int dil, int dtype, pssr_stream_t stream);
/* Gradient of the above: out[pixel][ci] = sum_t dcol[pixel - off_t][t * cp + ci].  With y != NULL the ReLU mask of the
 * pre-activation (y * scale + shift > 0) is applied; with stats != NULL also stats[stripe][0:c] += sum g,
 * stats[stripe][c:2c] += sum g * (y - mean) * invstd (f64 statistic buffer of PSSR_STAT_ROWS rows, caller-zeroed): the inputs of
 * pssr_bn_bwd_coefs for the BatchNorm in front of the ReLU. */
int pssr_col2im_dil(const void* dcol, int cp, void* out, int out_cs, int out_co, int c, int n, int h, int w, int dil,
                    const void* y, int y_cs, int y_co, const float* scale, const float* shift, const float* mean,
                    const float* invstd, double* stats, int dtype, pssr_stream_t stream);
/* stats[stripe][0:c] += sum x, stats[stripe][c:2c] += sum x^2 over the pixels of an NHWC slice: batch statistics of a
 * BatchNorm that normalises an existing tensor (the first BatchNorm of every ResBlockA branch, the BatchNorms of PSP_Pooling) */
int pssr_channel_stats_nhwc(const void* x, int cs, int co, int c, int64_t npix, double* stats, int dtype, pssr_stream_t stream);
/* out = sum of n_in (1..8) NHWC slices, then ReLU if relu != 0: "relu(sum(branches) + respass)" (_blocks.py:67).  ins / in_cs /
 * in_co are HOST arrays. */
int pssr_sum_relu(const void* const* ins, const int* in_cs, const int* in_co, int n_in, void* out, int out_cs, int out_co,
                  int64_t npix, int c, int relu, int dtype, pssr_stream_t stream);
/* dz = dout where out > 0 else 0 (gradient of a ReLU from its output) */
int pssr_relu_mask(const void* dout, int do_cs, int do_co, const void* out, int o_cs, int o_co, void* dz, int dz_cs, int dz_co,
                   int64_t npix, int c, int dtype, pssr_stream_t stream);
/* out = relu(x * scale[c] + shift[c]): F.relu(BatchNorm(x)) of PSP_Pooling (_blocks.py:88,91) */
int pssr_affine_relu(const void* x, int cs, int co, const float* scale, const float* shift, void* out, int out_cs, int out_co,
                     int64_t npix, int c, int dtype, pssr_stream_t stream);
/* F.max_pool2d(x, kernel_size=k) (_blocks.py:87; floor mode: out is [n][h / k][w / k][c]) and its gradient (to the first
 * maximum of each window in row-major order, zero for pixels outside every window) */
int pssr_maxpool_k(const void* in, int in_cs, int in_co, void* out, int out_cs, int out_co, int n, int h, int w, int c, int k,
                   int dtype, pssr_stream_t stream);
int pssr_maxpool_k_bwd(const void* act, int act_cs, int act_co, const void* dpool, int dp_cs, int dp_co, void* dx, int dx_cs,
                       int dx_co, int n, int h, int w, int c, int k, int dtype, pssr_stream_t stream);
/* F.interpolate(x, size=(h, w), mode="bilinear") (align_corners=False, _blocks.py:87) from [n][hs][ws][c] and its gradient */
int pssr_bilinear_up(const void* in, int in_cs, int in_co, void* out, int out_cs, int out_co, int n, int hs, int ws, int h, int w,
                     int c, int dtype, pssr_stream_t stream);
int pssr_bilinear_up_bwd(const void* dout, int do_cs, int do_co, void* din, int di_cs, int di_co, int n, int hs, int ws, int h,
                         int w, int c, int dtype, pssr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * RDNet encoder of RDResUNet (pssr/models/_rdnet.py, pssr/models/rdresunet.py:84).  The 1x1 convolutions of its
 * blocks and transitions run on pssr_conv2d / pssr_conv2d_wgrad (taps = 1; PSSR_PRO_GELU / PSSR_EPI_DGRAD_GELU fuse
 * nn.GELU); the kernels below are the HBM-bound remainder.  torch.cat of the dense features (_rdnet.py:132-138,
 * 169-170) is elided: every op reads / writes a channel slice (pointer, channel stride, channel offset).
 */
/* PatchifyStem input (_rdnet.py:110-111 seen as a 1x1 conv over patches): xpatch[n, y/ps, x/ps, ci*ps*ps + (y%ps)*ps + x%ps]
 * = (x*pre_scale + pre_shift)*scale[ci] + shift[ci]  (input scaling + input BatchNorm, rdresunet.py:105-107); `pc` =
 * channel stride (>= c*ps*ps, multiple of 16, rest zero).  Pack the stem weight with mode 2. */
int pssr_input_patchify(const float* x_nchw, void* xpatch, int n, int c, int h, int w, int ps, int pc,
                        float pre_scale, float pre_shift, const float* scale, const float* shift, int dtype,
                        pssr_stream_t stream);
/* Depthwise 7x7 conv, stride 1, zero padding 3 (nn.Conv2d(C, C, groups=C, kernel_size=7, padding=3), _rdnet.py:182,197).
 * pssr_dwconv7_pack: torch weight [C,1,7,7] f32 -> [49][C] f32 (flip = 1: rotated by 180 degrees, for the input gradient).
 * pssr_dwconv7: out (+)= bias + conv(in, packed)  (accumulate = 1 adds into `out`: the gradient of a dense-stage input
 * that several blocks read).  pssr_dwconv7_wgrad: dw[C][49] += sum_pixels dy * shifted x (f32 atomics, caller zeroes).
 * pssr_dwconv7_wgrad_ws (w % 8 == 0): the same sums without atomics -- one [C][49] slab per workgroup in `workspace`
 * (pssr_dwconv7_wgrad_workspace_bytes), added to dw in a fixed order: reproducible bit for bit, and not bound by the
 * L2 atomic rate (2.7 M atomics on 21 k addresses cost ~150 us whatever the map size). */
int pssr_dwconv7_pack(const float* w, float* packed, int c, int flip, pssr_stream_t stream);
/* pssr_dwconv7_pack of up to PSSR_DWPACK_BATCH_MAX weights by one launch (every block's forward and 180-degree-rotated copy: the
 * depthwise weights of pssr/models/rdnet.py:71-73 change once per optimizer step). */
#define PSSR_DWPACK_BATCH_MAX 48
typedef struct pssr_dwpack_batch {
    const float* w[PSSR_DWPACK_BATCH_MAX];
    float* packed[PSSR_DWPACK_BATCH_MAX];
    int32_t c[PSSR_DWPACK_BATCH_MAX];
    int32_t flip[PSSR_DWPACK_BATCH_MAX];
} pssr_dwpack_batch;
int pssr_dwconv7_pack_batch(const pssr_dwpack_batch* items, int n_items, pssr_stream_t stream);
int pssr_dwconv7(const void* in, int in_cs, int in_co, const float* w_packed, const float* bias, void* out, int out_cs,
                 int out_co, int n, int h, int w, int c, int accumulate, int dtype, pssr_stream_t stream);
int pssr_dwconv7_wgrad(const void* dy, int dy_cs, int dy_co, const void* x, int x_cs, int x_co, float* dw, int n, int h,
                       int w, int c, int dtype, pssr_stream_t stream);
int64_t pssr_dwconv7_wgrad_workspace_bytes(int n, int h, int w, int c);
int pssr_dwconv7_wgrad_ws(const void* dy, int dy_cs, int dy_co, const void* x, int x_cs, int x_co, float* dw, int n, int h,
                          int w, int c, int dtype, void* workspace, int64_t workspace_bytes, pssr_stream_t stream);
/* timm LayerNorm2d (_rdnet.py:60,112,183,198): layer norm over the C channels of every pixel, eps inside the sqrt, affine.
 * Channels [c, c_pad) of the output are written as zeros (K padding of the following 1x1 conv).  s2d = 1 writes the
 * 2x2 space-to-depth layout out[n, y/2, x/2, ((y&1)*2 + (x&1))*c_pad + ch], which makes the stride-2 transition conv
 * (_rdnet.py:56-62) a 1x1 conv over 4*c_pad channels (weights packed with mode 4).  mean / rstd: f32 [n*h*w], kept for
 * the backward pass (NULL in inference).
 * Backward: dx (+)= rstd*(g*gamma - mean_c(g*gamma) - xhat*mean_c(g*gamma*xhat)); stats[stripe][0..C) += sum g*xhat
 * (dgamma), stats[stripe][C..2C) += sum g (dbeta), f64 statistic buffer of PSSR_STAT_ROWS rows, caller-zeroed. */
int pssr_layernorm2d_fwd(const void* in, int in_cs, int in_co, const float* gamma, const float* beta, float eps, void* out,
                         int out_cs, int out_co, int s2d, int c_pad, int n, int h, int w, int c, float* mean, float* rstd,
                         int dtype, pssr_stream_t stream);
int pssr_layernorm2d_bwd(const void* g, int g_cs, int g_co, int s2d, int c_pad, const void* x, int x_cs, int x_co,
                         const float* gamma, const float* mean, const float* rstd, void* dx, int dx_cs, int dx_co,
                         int accumulate, int n, int h, int w, int c, double* stats, int dtype, pssr_stream_t stream);
/* out[img][c] += scale * sum_{pixels of img} a*b (b may be NULL: plain sum): the spatial mean of timm's
 * EffectiveSEModule (x.mean((2,3))) and the per-image reductions of its backward.  One workgroup per (image, 32 channels) sums all the
 * image's pixels in a fixed order: the same bits on every run, nothing but `out` written. */
int pssr_image_channel_dot(const void* a, int a_cs, int a_co, const void* b, int b_cs, int b_co, int n, int hw, int c,
                           float scale, float* out, int dtype, pssr_stream_t stream);
/* Entry points of the earlier version that met through a workspace: the workspace size is now 0 and `workspace` is ignored. */
int64_t pssr_image_channel_dot_workspace_bytes(int n, int hw, int c);
int pssr_image_channel_dot_ws(const void* a, int a_cs, int a_co, const void* b, int b_cs, int b_co, int n, int hw, int c,
                              float scale, float* out, int dtype, void* workspace, int64_t workspace_bytes, pssr_stream_t stream);
/* EffectiveSEModule gate: u = fc(s) (1x1 conv on the [N, C] means), gate = hard_sigmoid(u) = relu6(u + 3)/6. */
int pssr_ese_gate(const float* s_mean, const float* w_fc, const float* b_fc, int n, int c, float* u, float* gate,
                  pssr_stream_t stream);
/* out[p, c] = t[p, c] * gate[img][c] * gamma[c] + add[img][c]  (gate / add may be NULL): ESE gating and the layer
 * scale of DenseBlock.forward (_rdnet.py:173-174) written at the block's channel offset of the stage buffer; with
 * `add` it is also the backward of both. */
int pssr_scale_nc(const void* t, int t_cs, int t_co, const float* gate, const float* gamma, const float* add, void* out,
                  int out_cs, int out_co, int n, int hw, int c, int dtype, pssr_stream_t stream);
/* Backward of gate + layer scale on the [N, C] side tensors, A[n][c] = sum_pixels dout*t (pssr_image_channel_dot):
 * dgamma, d fc.bias, d fc.weight and add[n][c] = (1/hw) * W^T du, the per-image term of dt.  gate == NULL: plain
 * layer scale (dgamma = sum_n A only). */
int pssr_ese_bwd(const float* A, const float* gate, const float* u, const float* gamma, const float* s_mean,
                 const float* w_fc, int n, int c, int hw, float* du, float* dgamma, float* db_fc, float* dw_fc, float* add,
                 pssr_stream_t stream);

/* Second half of PSSR_EPI_HEADQ / PSSR_FLAG_HEADQ: out_f32_nchw[n][0][Y][X] = (bias + sum over the 9 taps (ky, kx) of
 * q[tap] at high-resolution pixel (Y + ky - 1, X + kx - 1), zero outside the image) * out_scale + out_shift, high-resolution pixel
 * (4 y + i, 4 x + j) being sub-pixel 4 i + j of low-resolution pixel (y, x)  (F.pixel_shuffle + Reconstruction.conv, _blocks.py:17;
 * "x * 128 + 128", resunet.py:94).  h, w: the LOW-resolution size. */
int pssr_head_q_gather(const float* q, const float* bias, float* out_nchw, int n, int h, int w,
                       float out_scale, float out_shift, pssr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Reconstruction.conv (pssr/models/_blocks.py:11,17 + "x*128+128", pssr/models/resunet.py:95) for C_out = 1..3: dedicated
 * streaming kernels (csrc/head_conv.hip) for the conv that reads the largest tensor of the network; bf16 storage only.
 * `act` is the NHWC activation in blocked pixel order (log2 r = blk, see pssr_conv_desc), `g` the incoming f32 NCHW
 * gradient of the network output (scaled by g_scale = the output scale, 128).
 *   fwd:   out_nchw = (conv3x3(act, w) + bias) * out_scale + out_shift
 *   dgrad: dact = (act > 0) * conv3x3_transposed(g * g_scale, w)        (ReLU of Reconstruction.pre fused)
 *   wgrad: dw_oihw += sum_pixels (g * g_scale) (x) shifted act            (f32 atomics into the caller-zeroed gradient)
 */
int pssr_head_conv_fwd(const void* act, int act_cs, int act_co, int blk, const float* w_oihw, const float* bias,
                       float* out_nchw, int n, int h, int w, int cin, int cout, float out_scale, float out_shift,
                       int dtype, pssr_stream_t stream);
int pssr_head_conv_dgrad(const float* g_nchw, float g_scale, const float* w_oihw, const void* act, int act_cs, int act_co,
                         void* dact, int d_cs, int d_co, int blk, int n, int h, int w, int cin, int cout, int dtype,
                         pssr_stream_t stream);
int pssr_head_conv_wgrad(const float* g_nchw, float g_scale, const void* act, int act_cs, int act_co, int blk,
                         float* dw_oihw, int n, int h, int w, int cin, int cout, int dtype, pssr_stream_t stream);
/* dgrad and wgrad in ONE pass over the activation (each tile is read once and used as ReLU mask, as the dW operand and for
 * the optional bias sums): dact and dw_oihw as above (blk <= 2); bias_sum (or NULL) is a caller-zeroed f32
 * [4^blk * cin] that receives sum_pixels dact per (sub-pixel, channel) = per channel of the blocked NHWC tensor viewed at
 * low resolution, i.e. the bias gradient of the pixel-shuffle convolution in front (Reconstruction.pre). */
int pssr_head_conv_bwd(const float* g_nchw, float g_scale, const float* w_oihw, const void* act, int act_cs, int act_co,
                       void* dact, int d_cs, int d_co, int blk, float* dw_oihw, float* bias_sum, int n, int h, int w, int cin,
                       int cout, int dtype, pssr_stream_t stream);

/* The same with order-independent sums (bit-reproducible training): dw_rows [PSSR_STAT_ROWS][cout * cin * 9] and bias_rows (or NULL)
 * [PSSR_STAT_ROWS][4^blk * cin], caller-zeroed f64 statistic buffers; fold them with pssr_f64_to_f32(rows = PSSR_STAT_ROWS). */
int pssr_head_conv_bwd_rows(const float* g_nchw, float g_scale, const float* w_oihw, const void* act, int act_cs, int act_co,
                            void* dact, int d_cs, int d_co, int blk, double* dw_rows, double* bias_rows, int n, int h, int w,
                            int cin, int cout, int dtype, pssr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-sheet prediction on the device (BASELINE config 5).
 * pssr_sliding_tiles_u8: tiles [ntile][c][size][size] f32 = the row-major sliding windows tile0 .. tile0+ntile-1 of a uint8 sheet
 *   [c][h][w] (pssr/data.py:629-638 `_sliding_window` + `_tensor_ready`; stride = size - overlap; remainders dropped).
 * pssr_patch_tiles_u8: sheet [c][n_rows*step+overlap][n_cols*step+overlap] u8 = overlap-averaged stitching of u8 tiles
 *   [n_rows*n_cols][c][size][size] with `margin` pixels trimmed on inner edges, then the uint8 cast (truncation):
 *   pssr/util.py:116-137 `_patch_images` + pssr/util.py:101.  step = size - overlap (all in output pixels).
 */
int pssr_sliding_tiles_u8(const uint8_t* sheet, float* tiles, int c, int h, int w, int size, int stride, int tile0, int ntile,
                          pssr_stream_t stream);
int pssr_patch_tiles_u8(const uint8_t* tiles, uint8_t* sheet, int c, int n_rows, int n_cols, int size, int overlap, int margin,
                        pssr_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * normalize_preds (pssr/util.py:139-191) for uint8 image pairs of equal size resident in HBM: percentile window
 * of the ground truth, mean removal, covariance amplitude of the prediction, rescaling to the ground truth's intensity, clip
 * and uint8 cast -- bit-exact with the reference's numpy float32 / float64 arithmetic (float32 percentile lerp and pairwise
 * summation reproduced; see csrc/metrics.hip).  Images are [n_images][pixels_per_image]; workspace of
 * n_images * pssr_normalize_preds_workspace_bytes(pixels_per_image) bytes. */
int64_t pssr_normalize_preds_workspace_bytes(int64_t pixels_per_image);
int pssr_normalize_preds_u8(const uint8_t* hr, const uint8_t* hr_hat, uint8_t* hr_norm, uint8_t* hr_hat_norm, int n_images,
                            int64_t pixels_per_image, float pmin, float pmax, void* workspace, pssr_stream_t stream);
/* The same for a prediction [n_images][h][w] whose size differs from the ground truth's [n_images][H][W] (pssr/util.py:176-179): the
 * covariance amplitude is taken between the ground truth and skimage.transform.resize(prediction, (H, W)) -- order-1 interpolation at
 * (i + 0.5) * in / out - 0.5, mirror boundary, Gaussian pre-filter sigma = (in / out - 1) / 2 along shrinking axes (scikit-image
 * >= 0.19, i.e. scipy.ndimage.zoom(grid_mode=True) + gaussian_filter; restated and checked against scipy in oracle/metrics_ref.py;
 * scikit-image itself is absent: parity unpinned).  Outputs keep their own sizes, as upstream.  Workspace of
 * pssr_normalize_preds_resized_workspace_bytes(...) bytes. */
int64_t pssr_normalize_preds_resized_workspace_bytes(int n_images, int H, int W, int h, int w);
int pssr_normalize_preds_resized_u8(const uint8_t* hr, int H, int W, const uint8_t* hr_hat, int h, int w, uint8_t* hr_norm,
                                    uint8_t* hr_hat_norm, int n_images, float pmin, float pmax, void* workspace,
                                    pssr_stream_t stream);

/* Per-image restoration metrics of uint8 image pairs [n_images][h][w] resident in HBM, as pssr/predict.py:193-203 computes them
 * through skimage.metrics (peak_signal_noise_ratio, structural_similarity with data_range 255: uniform 7x7 window, K1 .01,
 * K2 .03, sample covariance, mean over the interior).  out[i] = { sum of squared differences (exact integer in a double),
 * mean SSIM }: mse = out[0] / (h w 255^2), psnr = 10 log10(255^2 h w / out[0]).  h, w >= 7 (skimage raises below that too).
 * workspace: pssr_image_metrics_workspace_bytes(n_images, h, w) bytes.  Reproducible bit for bit (fixed summation order). */
int64_t pssr_image_metrics_workspace_bytes(int n_images, int h, int w);
int pssr_image_metrics_u8(const uint8_t* hr, const uint8_t* hr_hat, double* out, int n_images, int h, int w, void* workspace,
                          pssr_stream_t stream);

/* Several small f32 device-to-device copies in one launch (host struct passed by value to the kernel): the hand-written
 * backward pass moves side results (dbeta doubling as a bias gradient, centre taps, ...) into the flat gradient buffer the
 * all-reduce and AdamW read (the reference leaves this to autograd's .grad accumulation). */
#define PSSR_COPY_BATCH_MAX 16
typedef struct pssr_copy_batch {
    float* dst[PSSR_COPY_BATCH_MAX];
    const float* src[PSSR_COPY_BATCH_MAX];
    int64_t n[PSSR_COPY_BATCH_MAX];
} pssr_copy_batch;
int pssr_copy_f32_batch(const pssr_copy_batch* items, int n_items, pssr_stream_t stream);
/* Several pssr_f64_to_f32 folds (striped f64 statistic rows [stripes][n] -> f32 [n], optionally accumulated) by one launch: the
 * bias / LayerNorm / layer-scale gradient sums of a backward pass (RDNet has ~115 per step) are parameter gradients nobody reads
 * before the pass ends, so the engine queues them and folds 16 at a time. */
typedef struct pssr_fold_batch {
    float* dst[PSSR_COPY_BATCH_MAX];
    const double* src[PSSR_COPY_BATCH_MAX];
    int32_t n[PSSR_COPY_BATCH_MAX];
    int32_t accumulate[PSSR_COPY_BATCH_MAX];
} pssr_fold_batch;
int pssr_f64_to_f32_batch(const pssr_fold_batch* items, int n_items, int stripes, pssr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PSSR_MI355_H */
