// C ABI of the implicit-GEMM convolution (forward and input-gradient): argument checks and dispatch on the storage type.
// The kernels live in conv_igemm_impl.h, compiled once per type by conv_igemm_{bf16,f16,f32}.hip.
#include "conv_igemm_args.h"

using pssr_conv::ConvArgs;

// ABI version 1 entry, kept so that old bindings still load: the round-1 pipelined loop it selected is gone (superseded by
// conv_v3_kernel, tunable IGEMM_V3); always reports mode 0 and changes nothing
extern "C" int pssr_conv2d_pipeline_mode(int mode) {
    (void)mode;
    return 0;
}

static int conv2d_entry(const pssr_conv_desc* d, pssr_stream_t stream, long* query_ws);

#ifdef PSSR_V3_STAMPS
static unsigned* g_stamp_buf = nullptr;
extern "C" void pssr_debug_stamp_buffer(void* p) { g_stamp_buf = (unsigned*)p; }
#endif

extern "C" int pssr_conv2d(const pssr_conv_desc* d, pssr_stream_t stream) { return conv2d_entry(d, stream, nullptr); }

extern "C" int64_t pssr_conv2d_workspace_bytes(const pssr_conv_desc* d) {
    long bytes = 0;
    const int rc = conv2d_entry(d, nullptr, &bytes);
    return rc != PSSR_OK ? (int64_t)rc : (int64_t)bytes;
}

static int conv2d_entry(const pssr_conv_desc* d, pssr_stream_t stream, long* query_ws) {
    PSSR_CHECK(d != nullptr, PSSR_ERR_ARG, "conv2d: null desc");
    PSSR_CHECK(d->dtype == PSSR_F32 || d->dtype == PSSR_BF16 || d->dtype == PSSR_F16, PSSR_ERR_ARG, "conv2d: bad dtype %d", d->dtype);
    const int kch = d->dtype == PSSR_F32 ? 8 : 16;
    const int esz = d->dtype == PSSR_F32 ? 4 : 2;
    PSSR_CHECK(d->n > 0 && d->h > 0 && d->w > 0, PSSR_ERR_ARG, "conv2d: bad shape %dx%dx%d", d->n, d->h, d->w);
    PSSR_CHECK(query_ws || (d->in0 && d->w0 && d->out), PSSR_ERR_ARG, "conv2d: null pointer");
    PSSR_CHECK(d->cin0 > 0 && d->cin0 % kch == 0, PSSR_ERR_ARG, "conv2d: cin0=%d must be a positive multiple of %d", d->cin0, kch);
    PSSR_CHECK(d->taps0 == 9 || d->taps0 == 1, PSSR_ERR_ARG, "conv2d: taps0=%d", d->taps0);
    PSSR_CHECK((d->in0_cstride * esz) % 16 == 0 && (d->in0_coff * esz) % 16 == 0, PSSR_ERR_ARG, "conv2d: in0 stride/offset not 16-byte aligned");
    PSSR_CHECK(d->in0_coff + d->cin0 <= d->in0_cstride, PSSR_ERR_ARG, "conv2d: in0 slice exceeds stride");
    if (d->cin1) {
        PSSR_CHECK(d->in1 && d->w1 && d->cin1 % kch == 0 && d->taps1 == 1, PSSR_ERR_ARG, "conv2d: source 1 must be a 1x1 (or flat-K) source");
        PSSR_CHECK((d->in1_cstride * esz) % 16 == 0 && (d->in1_coff * esz) % 16 == 0 && d->in1_coff + d->cin1 <= d->in1_cstride, PSSR_ERR_ARG, "conv2d: in1 stride/offset");
    }
    if (d->epilogue == PSSR_EPI_FINAL) {
        PSSR_CHECK(d->cout > 0 && d->cout <= 32 && (d->flags & ~PSSR_FLAG_SOLO) == 0, PSSR_ERR_ARG, "conv2d: EPI_FINAL needs 0 < cout <= 32 and no flags");
    } else {
        PSSR_CHECK(d->cout > 0 && d->cout % 4 == 0, PSSR_ERR_ARG, "conv2d: cout=%d must be a positive multiple of 4", d->cout);
        PSSR_CHECK(d->out_coff % 4 == 0 && d->out_cstride % 4 == 0 &&
                   d->out_coff + ((d->flags & PSSR_FLAG_SHUF2) ? d->cout / 4 : d->cout) <= d->out_cstride, PSSR_ERR_ARG, "conv2d: out stride/offset");
    }
    PSSR_CHECK(d->n_pad % 128 == 0 && d->n_pad >= d->cout, PSSR_ERR_ARG, "conv2d: n_pad=%d", d->n_pad);
    PSSR_CHECK(d->prologue >= 0 && d->prologue <= PSSR_PRO_GELU, PSSR_ERR_ARG, "conv2d: prologue=%d", d->prologue);
    PSSR_CHECK(d->prologue != PSSR_PRO_BN_RELU || (d->pro_scale && d->pro_shift), PSSR_ERR_ARG, "conv2d: prologue needs scale/shift");
    PSSR_CHECK(d->epilogue >= 0 && d->epilogue <= 5, PSSR_ERR_ARG, "conv2d: epilogue=%d", d->epilogue);
    if (d->flags & PSSR_FLAG_HEADQ)
        PSSR_CHECK(d->epilogue == PSSR_EPI_STORE && !(d->flags & (PSSR_FLAG_STATS | PSSR_FLAG_AFFINE)) && (d->flags & PSSR_FLAG_RELU), PSSR_ERR_ARG,
                   "conv2d: FLAG_HEADQ goes with EPI_STORE | FLAG_RELU only");
    if (d->epilogue == PSSR_EPI_HEADQ || (d->flags & PSSR_FLAG_HEADQ)) {
        PSSR_CHECK(query_ws || (d->head_w && d->head_q), PSSR_ERR_ARG, "conv2d: EPI_HEADQ needs head_w / head_q");
        PSSR_CHECK(esz == 2 && d->cout == 1024 && d->taps0 == 9 && d->w >= 16 && d->h >= 16 && d->prologue != PSSR_PRO_GELU && !(d->flags & PSSR_FLAG_STATS),
                   PSSR_ERR_UNSUPPORTED, "conv2d: EPI_HEADQ needs 16-bit storage, a 3x3 source, cout = 16 x 64 (4 x upscaling) and at least 16x16 pixels");
    }
    PSSR_CHECK(d->in0_blk >= 0 && d->out_blk >= 0 && d->aux_blk >= 0 && d->in0_blk <= 3 && d->out_blk <= 3 && d->aux_blk <= 3, PSSR_ERR_ARG, "conv2d: blocked order");
    {
        const int mb = d->in0_blk > d->out_blk ? (d->in0_blk > d->aux_blk ? d->in0_blk : d->aux_blk) : (d->out_blk > d->aux_blk ? d->out_blk : d->aux_blk);
        PSSR_CHECK(d->h % (1 << mb) == 0 && d->w % (1 << mb) == 0, PSSR_ERR_ARG, "conv2d: blocked order needs H,W multiples of r");
    }
    if (d->epilogue == PSSR_EPI_TAIL || d->epilogue == PSSR_EPI_DGRAD_MASK) {
        PSSR_CHECK(d->aux && d->aux_scale && d->aux_shift && d->aux_coff % 4 == 0 && d->aux_cstride % 4 == 0, PSSR_ERR_ARG, "conv2d: epilogue needs aux tensor");
    }
    if (d->epilogue == PSSR_EPI_DGRAD_GELU)
        PSSR_CHECK(d->aux && d->aux_coff % 4 == 0 && d->aux_cstride % 4 == 0, PSSR_ERR_ARG, "conv2d: GELU backward needs the pre-activation as aux");
    if (d->flags & PSSR_FLAG_SHUF2) {
        PSSR_CHECK(d->epilogue == PSSR_EPI_STORE && !(d->flags & (PSSR_FLAG_STATS | PSSR_FLAG_HEADQ)) && d->out_blk == 0, PSSR_ERR_ARG,
                   "conv2d: FLAG_SHUF2 goes with EPI_STORE, without statistics, tap planes or a blocked output order");
        PSSR_CHECK(esz == 2 && d->cout % 32 == 0 && d->out_coff % 8 == 0 && d->out_cstride % 8 == 0 && pssr_tunables().conv_epi8, PSSR_ERR_UNSUPPORTED,
                   "conv2d: FLAG_SHUF2 needs 16-bit storage, cout a multiple of 32 and 8-channel aligned outputs");
    }
    if (d->flags & PSSR_FLAG_AFFINE) {
        PSSR_CHECK(d->epilogue == PSSR_EPI_STORE && !(d->flags & PSSR_FLAG_STATS) && d->aux_scale && d->aux_shift, PSSR_ERR_ARG,
                   "conv2d: FLAG_AFFINE needs EPI_STORE without statistics and aux_scale / aux_shift");
        PSSR_CHECK(esz == 2 && d->cout % 8 == 0 && d->out_coff % 8 == 0 && d->out_cstride % 8 == 0 && pssr_tunables().conv_epi8, PSSR_ERR_UNSUPPORTED,
                   "conv2d: FLAG_AFFINE needs 16-bit storage and 8-channel aligned outputs");
    }
    if (d->flags & PSSR_FLAG_STATS) {
        PSSR_CHECK(d->stats != nullptr, PSSR_ERR_ARG, "conv2d: stats buffer missing");
        PSSR_CHECK(d->epilogue != PSSR_EPI_DGRAD_MASK || (d->aux_mean && d->aux_invstd), PSSR_ERR_ARG, "conv2d: mask stats need mean/invstd");
        PSSR_CHECK(d->epilogue != PSSR_EPI_TAIL, PSSR_ERR_ARG, "conv2d: no stats on tail");
    }
    ConvArgs a;
    a.N = d->n; a.H = d->h; a.W = d->w;
    a.in[0] = d->in0; a.in_cs[0] = d->in0_cstride; a.in_co[0] = d->in0_coff; a.nchunks[0] = d->cin0 / kch; a.taps[0] = d->taps0; a.w[0] = d->w0;
    a.in[1] = d->in1; a.in_cs[1] = d->in1_cstride; a.in_co[1] = d->in1_coff; a.nchunks[1] = d->cin1 ? d->cin1 / kch : 0; a.taps[1] = d->cin1 ? d->taps1 : 1; a.w[1] = d->w1;
    a.prologue = d->prologue; a.pro_scale = d->pro_scale; a.pro_shift = d->pro_shift;
    a.out = d->out; a.out_cs = d->out_cstride; a.out_co = d->out_coff; a.cout = d->cout; a.n_pad = d->n_pad;
    a.bias = d->bias; a.epi = d->epilogue; a.flags = d->flags;
    a.aux = d->aux; a.aux_cs = d->aux_cstride; a.aux_co = d->aux_coff;
    a.aux_scale = d->aux_scale; a.aux_shift = d->aux_shift; a.aux_mean = d->aux_mean; a.aux_invstd = d->aux_invstd;
    a.stats = d->stats;
    a.in0_blk = d->in0_blk; a.out_blk = d->out_blk; a.aux_blk = d->aux_blk;
    a.out_scale = d->out_scale; a.out_shift = d->out_shift;
    const bool headq = a.epi == PSSR_EPI_HEADQ || (a.flags & PSSR_FLAG_HEADQ);
    a.head_w = headq ? d->head_w : nullptr;
    a.head_q = headq ? d->head_q : nullptr;
    a.tiles_x = a.tiles_y = a.tiles_n = 0;
    a.epi8 = esz == 2 && d->epilogue != PSSR_EPI_FINAL && d->cout % 8 == 0 && d->out_coff % 8 == 0 && d->out_cstride % 8 == 0 &&
             (d->epilogue == PSSR_EPI_STORE || d->epilogue == PSSR_EPI_HEADQ || (d->aux_coff % 8 == 0 && d->aux_cstride % 8 == 0));
    if (!pssr_tunables().conv_epi8) a.epi8 = 0;
    a.dbg = pssr_tunables().igemm_dbg;
    a.stamps = nullptr;
#ifdef PSSR_V3_STAMPS
    a.stamps = g_stamp_buf;
#endif
    long ws_query = 0;
    if (query_ws) { a.ksplit = -1; a.ws = (float*)&ws_query; }
    else { a.ws = (float*)d->workspace; a.ksplit = d->workspace ? (int)(d->workspace_bytes / 1024) : 0; }
    hipStream_t s = (hipStream_t)stream;
    const int rc = d->dtype == PSSR_BF16 ? pssr_conv::launch_bf16(a, s) : d->dtype == PSSR_F16 ? pssr_conv::launch_f16(a, s) : pssr_conv::launch_f32(a, s);
    if (query_ws) *query_ws = ws_query;
    return rc;
}
