// Pipelined 3x3 implicit-GEMM convolution ("v3") for gfx950, 16-bit storage (bf16 / f16), forward and input-gradient.
// Included by conv_igemm_impl.h (after the epilogues).
//
// Why another main loop (round-1 counters of conv_igemm_kernel<.,128,geo0,9> on a K = 1152 layer): MFMA pipe 31 % busy,
// LDS 41 % busy with 24 % of its cycles bank conflicts, 5.6 vector instructions per MFMA, and -- from the byte
// counts -- the 128 x 128 tile stages 42.6 KB per K chunk through global -> VGPR -> ds_write for 4.7 MFLOP, i.e. the
// L2 -> CU path (~56 B/clk/CU) and the LDS store path, not the matrix pipe, bound the loop well below 50 %.
//
//   * tile = 16 x 16 output pixels x 128 channels per 256-thread workgroup; a wave owns 128 pixels x 64 channels
//     (4 x 2 accumulator tiles of 32 x 32): per MFMA 0.75 fragment reads instead of 1, and the weight slices of a
//     chunk (3/4 of the staged bytes) are shared by twice the pixels;
//   * weights never touch registers: the packed layout IS the LDS image, a stage (one kernel row = 3 taps of one
//     32-byte channel chunk, 12 KB) goes global -> LDS by LDS-DMA into a ring of three stage buffers, two stages ahead
//     of the multiply, retired by counted `s_waitcnt vmcnt(N)` + one raw `s_barrier` per stage;
//   * the input halo tile (18 x 18 pixels x 32 B) of the next chunk is requested at the first stage of a chunk and
//     committed (BatchNorm + ReLU prologue, zero padding) to the other of two images at its last stage;
//   * the image is two planes [half][pixel][16 B]: every fragment address is ONE per-wave base register plus an
//     immediate (tap, row tile), and with the row permutation below every ds_read_b128 of a fragment is conflict
//     free (the 18-pixel pitch of the old [pixel][32 B] image made the two image rows of a fragment collide);
//   * the multiply of a stage is one inline-asm block: fragment reads of tap t+1 are issued between the MFMAs of
//     tap t (two register sets), one `lgkmcnt(0)` per tap.  Left to hipcc this loop was serialised behind full waits
//     (the round-1 pipelined loop, since removed);
//   * 57.6 KB of LDS and <= 256 registers: two workgroups per CU, each covering the other's barriers and epilogue.
//
// Row permutation: ds_read_b128 is served in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32).  MFMA row r of a
// 32-row tile is pixel (x = r & 15, y = (r >> 4) ^ [4 <= (r & 15) < 12]) of the tile's 2 x 16 pixel block, so that
// each lane group reads 16 consecutive pixels of ONE image row = 16 distinct 16-byte slots.  The epilogue maps rows
// back with the same function (Cfg::PERM).
#pragma once

namespace {

template <int BN> struct Cfg3 {
    static_assert(BN == 128 || BN == 64, "v3 tiles: 16x16 pixels x 128 channels, or 16x32 pixels x 64 channels");
    // BN = 128: 16 x 16 pixels, waves 2 (pixels) x 2 (channels); BN = 64: 16 rows x 32 columns, the four waves stacked along the rows.
    // Either way a wave owns 128 pixels x 64 channels = 4 x 2 accumulator tiles and runs the same stage code.
    static constexpr int TWL = BN == 128 ? 4 : 5, THL = 4, TW = 1 << TWL, TH = 16, NI = 1;
    static constexpr int WM = BN == 128 ? 2 : 4, WN = 4 / WM, MI = 4, NJ = 2, MP = TW * TH;
    static constexpr int HW2 = TW + 2, HPI = (TH + 2) * HW2;
    static constexpr int PLANE = HPI * 16, A_BYTES = 2 * PLANE;          // 5184 / 9792: both = 64 mod 128, so the two halves of a commit store hit different banks
    static_assert(PLANE % 128 == 64, "plane size");
    static constexpr int A_ITEMS = (2 * HPI + 255) / 256;                 // 3 / 5
    static constexpr int B_SLICE = BN * 32, B_STAGE = 3 * B_SLICE;        // 4096, 12288 / 2048, 6144
    static constexpr int PPS = B_SLICE / 1024, NPIECE = 3 * PPS;          // 1-KiB LDS-DMA pieces per slice / per stage
    static constexpr int NP = (NPIECE + 3) / 4;                           // pieces a wave issues per stage (3 / 2; BN = 64: two of the 8 are duplicates)
    static constexpr int B0 = 2 * A_BYTES;
    static constexpr int MAIN_BYTES = B0 + 3 * B_STAGE;                   // 57600 / 57600
    static constexpr int E_BYTES = WM * 32 * (BN + 4) * 4, RED_BYTES = 4 * BN * 2 * 4;     // epilogue rows are BN + 4 floats apart (conv_epilogue8, TR)
    static constexpr int LDS_BYTES = (MAIN_BYTES > E_BYTES + RED_BYTES) ? MAIN_BYTES : (E_BYTES + RED_BYTES);
    static constexpr int PERM = BN == 128 ? 1 : 2;
    // byte offset (within a plane) of accumulator row tile mi of a wave: image rows 2 mi (BN = 128); row pair mi >> 1, column half mi & 1 (BN = 64)
    static constexpr int AMI1 = BN == 128 ? 2 * HW2 * 16 : 16 * 16;
    static constexpr int AMI2 = BN == 128 ? 4 * HW2 * 16 : 2 * HW2 * 16;
    static constexpr int AMI3 = BN == 128 ? 6 * HW2 * 16 : 2 * HW2 * 16 + 16 * 16;
    static constexpr int ROWB = HW2 * 16;                                  // bytes per halo row of a plane (one kernel row down)
};

// ---- the multiply of one stage as one asm statement.
// operands: %0-%7 accumulators [mi*2+nj]; %8-%13 fragment set 0 (A0-A3, B0, B1); %14-%19 set 1; %20 image address
// (per-wave base incl. the lane's half plane); %21 weight address; %22 image offset (kernel row); %23-%26 offsets of the
// wave's row tiles 0-3 (%23 = 0); %27 weight offset (ring slot); %28 bytes of one weight slice (tap)
#define V3_AMI_0 "%c23"
#define V3_AMI_1 "%c24"
#define V3_AMI_2 "%c25"
#define V3_AMI_3 "%c26"
#define V3_RA(D, MI_, KX) "ds_read_b128 %" #D ", %20 offset:%c22+" V3_AMI_##MI_ "+" #KX "*16\n\t"
#define V3_RB(D, NJ_, KX) "ds_read_b128 %" #D ", %21 offset:%c27+" #KX "*%c28+" #NJ_ "*1024\n\t"
// (the weights are the MFMA's A operand: the accumulators hold the transposed product, 4 consecutive channels of one pixel per
// register quad -- what conv_epilogue8<..., TR = true> writes to LDS with one ds_write_b128)
#define V3_MM(OP, ACC, A, B) OP " %" #ACC ", %" #B ", %" #A ", %" #ACC "\n\t"
#define V3_READ0(KXA, KXB) V3_RA(8, 0, KXA) V3_RB(12, 0, KXB) V3_RB(13, 1, KXB) V3_RA(9, 1, KXA) V3_RA(10, 2, KXA) V3_RA(11, 3, KXA)
// 8 MFMAs on set 0 with the reads of tap KX into set 1 between them (and vice versa)
#define V3_TAP01(OP, KX)                                                                     \
    V3_MM(OP, 0, 8, 12) V3_RA(14, 0, KX) V3_RB(18, 0, KX)                                    \
    V3_MM(OP, 1, 8, 13) V3_RB(19, 1, KX) V3_RA(15, 1, KX)                                    \
    V3_MM(OP, 2, 9, 12) V3_RA(16, 2, KX) V3_RA(17, 3, KX)                                    \
    V3_MM(OP, 3, 9, 13) V3_MM(OP, 4, 10, 12) V3_MM(OP, 5, 10, 13) V3_MM(OP, 6, 11, 12) V3_MM(OP, 7, 11, 13)
#define V3_TAP10(OP, KX)                                                                     \
    V3_MM(OP, 0, 14, 18) V3_RA(8, 0, KX) V3_RB(12, 0, KX)                                    \
    V3_MM(OP, 1, 14, 19) V3_RB(13, 1, KX) V3_RA(9, 1, KX)                                    \
    V3_MM(OP, 2, 15, 18) V3_RA(10, 2, KX) V3_RA(11, 3, KX)                                   \
    V3_MM(OP, 3, 15, 19) V3_MM(OP, 4, 16, 18) V3_MM(OP, 5, 16, 19) V3_MM(OP, 6, 17, 18) V3_MM(OP, 7, 17, 19)
#define V3_TAP0_LAST(OP)                                                                     \
    V3_MM(OP, 0, 8, 12) V3_MM(OP, 1, 8, 13) V3_MM(OP, 2, 9, 12) V3_MM(OP, 3, 9, 13)          \
    V3_MM(OP, 4, 10, 12) V3_MM(OP, 5, 10, 13) V3_MM(OP, 6, 11, 12) V3_MM(OP, 7, 11, 13)
#define V3_WAIT "s_waitcnt lgkmcnt(0)\n\t"
#define V3_ROW_TEXT(OP) V3_READ0(0, 0) V3_WAIT V3_TAP01(OP, 1) V3_WAIT V3_TAP10(OP, 2) V3_WAIT V3_TAP0_LAST(OP)
#define V3_ONE_TEXT(OP) V3_READ0(1, 0) V3_WAIT V3_TAP0_LAST(OP)
#define V3_OPERANDS(AOFF, BOFF)                                                                                           \
        : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]),           \
          "+v"(acc[3][0]), "+v"(acc[3][1]),                                                                               \
          "=&v"(fr[0]), "=&v"(fr[1]), "=&v"(fr[2]), "=&v"(fr[3]), "=&v"(fr[4]), "=&v"(fr[5]),                              \
          "=&v"(fr[6]), "=&v"(fr[7]), "=&v"(fr[8]), "=&v"(fr[9]), "=&v"(fr[10]), "=&v"(fr[11])                             \
        : "v"(a_addr), "v"(b_addr), "i"(AOFF), "i"(0), "i"(C::AMI1), "i"(C::AMI2), "i"(C::AMI3), "i"(BOFF), "i"(C::B_SLICE)   \
        : "memory"

template <typename T> struct V3Op;
template <> struct V3Op<bf16_t> { static constexpr bool BF = true; };
template <> struct V3Op<f16_t> { static constexpr bool BF = false; };

// one kernel row (3 taps) of the chunk staged in image `a_addr`, weights in ring slot BOFF
template <typename T, class C, int AOFF, int BOFF>
__device__ __forceinline__ void v3_row(f32x16 (&acc)[4][2], unsigned a_addr, unsigned b_addr) {
    u32x4 fr[12];
    if constexpr (V3Op<T>::BF) asm volatile(V3_ROW_TEXT("v_mfma_f32_32x32x16_bf16") V3_OPERANDS(AOFF, BOFF));
    else asm volatile(V3_ROW_TEXT("v_mfma_f32_32x32x16_f16") V3_OPERANDS(AOFF, BOFF));
}
// the centre tap only (1x1 second source, staged in the same halo geometry); the ring slot is part of b_addr
template <typename T, class C, int AOFF>
__device__ __forceinline__ void v3_one(f32x16 (&acc)[4][2], unsigned a_addr, unsigned b_addr) {
    u32x4 fr[12];
    if constexpr (V3Op<T>::BF) asm volatile(V3_ONE_TEXT("v_mfma_f32_32x32x16_bf16") V3_OPERANDS(AOFF, 0));
    else asm volatile(V3_ONE_TEXT("v_mfma_f32_32x32x16_f16") V3_OPERANDS(AOFF, 0));
}

// LDS-DMA of one 1-KiB piece: 64 lanes x 16 B from (sbase + voff) to LDS address lds (wave-uniform).  The s_nop 4 covers a
// scalar operand that came out of a v_readfirstlane
__device__ __forceinline__ void v3_dma(unsigned lds, unsigned voff, const char* sbase) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds), "v"(voff), "s"(sbase) : "memory");
}

template <typename T, int BN>
__global__ __launch_bounds__(256, 2) void conv_v3_kernel(const ConvArgs p) {
    using C = Cfg3<BN>;
    using X = TT<T>;
    constexpr int EPS = X::EPS, KCH = X::KCH, ESZ = (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int h = lane >> 5, r = lane & 31;
#ifdef PSSR_V3_STAMPS
    unsigned long long st_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0) :: "memory");
#endif

    // consecutive tiles (channel tiles of a pixel tile first, then its neighbour along x) on ONE XCD: conv_igemm_kernel
    int bid = blockIdx.x;
    if (p.ksplit <= 1 && !(p.dbg & 64)) {
        const int per = gridDim.x >> 3;
        if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
    }
    const int tn = bid % p.tiles_n;
    int tmi = bid / p.tiles_n;
    const int tile_x = tmi % p.tiles_x; tmi /= p.tiles_x;
    const int tile_y = tmi % p.tiles_y;
    const int img0 = tmi / p.tiles_y;
    const int x0 = tile_x << C::TWL, y0 = tile_y << 4, n0 = tn * BN;

    // ---- halo staging descriptors (the same pixels for every chunk): item = (halo pixel, 16-byte half)
    const long img_base = (long)img0 * p.H * p.W;
    unsigned a_off0[C::A_ITEMS], a_off1[C::A_ITEMS];
    int a_wr[C::A_ITEMS];
    bool a_ok[C::A_ITEMS];
    const int half = tid & 1;
#pragma unroll
    for (int it = 0; it < C::A_ITEMS; ++it) {
        const int idx = tid + it * 256;
        const int pp = idx >> 1;
        const int hy = pp / C::HW2, hx = pp % C::HW2;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool inb = pp < C::HPI && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        a_ok[it] = inb;
        a_off0[it] = inb ? (unsigned)((pix_index(img0, gy, gx, p.H, p.W, p.in0_blk) - img_base) * p.in_cs[0] * ESZ + half * 16) : 0u;
        a_off1[it] = inb ? (unsigned)(gy * p.W + gx) * (unsigned)(p.in_cs[1] * ESZ) + half * 16 : 0u;
        a_wr[it] = pp < C::HPI ? half * C::PLANE + pp * 16 : -1;
    }
    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.in[0] + (img_base * p.in_cs[0] + p.in_co[0]) * ESZ), 0, (int)0xfffffff0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.in[1] + (img_base * p.in_cs[1] + p.in_co[1]) * ESZ), 0, (int)0xfffffff0u, 0x00020000);

    // ---- fragment addresses: image row tile mi / tap and weight tile nj / tap are immediates
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int rx = r & 15, ry = (r >> 4) ^ ((rx >= 4 && rx < 12) ? 1 : 0);
    const int wrow = BN == 128 ? wm * 8 : wave * 4;                      // first image row of the wave's 128 pixels
    const unsigned a_addr0 = lds0 + h * C::PLANE + ((wrow + ry) * C::HW2 + rx) * 16;
    const unsigned b_addr0 = lds0 + C::B0 + (wn * 64 + r) * 32 + ((h ^ ((r >> 3) & 1)) << 4);

    // ---- LDS-DMA pieces of this wave: piece q = wave + 4 j of a 3-slice stage = (slice q >> 2, KiB q & 3 of the slice)
    const int tap_stride = p.n_pad * 32;
    // piece q = (wave + 4 j) mod NPIECE of a 3-slice stage = (slice q / PPS, KiB q % PPS of the slice); BN = 64 has 6 pieces for 8 issue
    // slots: two are issued twice (same bytes to the same place), so that every wave has the same number of operations in flight
    unsigned d_voff[C::NP];
    unsigned d_loff[C::NP];
#pragma unroll
    for (int j = 0; j < C::NP; ++j) {
        const int q = (wave + 4 * j) % C::NPIECE;
        d_voff[j] = (unsigned)((q / C::PPS) * tap_stride + (q % C::PPS) * 1024 + lane * 16);
        d_loff[j] = (unsigned)(q * 1024);
    }
    const unsigned d_lds = lds0 + C::B0;                                  // + slot * B_STAGE + piece * 1024
    const char* const w0n = (const char*)p.w[0] + (long)n0 * 32;
    const char* const w1n = (const char*)p.w[1] + (long)n0 * 32;

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

    // chunk range of source 0 (split-K: blockIdx.y owns a slice; the 1x1 source rides with the last slice)
    int cb = 0, ce = p.nchunks[0];
    if (p.ksplit > 1) {
        const int per = (ce + p.ksplit - 1) / p.ksplit;
        cb = blockIdx.y * per;
        ce = cb + per < ce ? cb + per : ce;
        if (cb > ce) cb = ce;
    }
    const int n1c = (p.nchunks[1] > 0 && (p.ksplit <= 1 || (int)blockIdx.y == p.ksplit - 1)) ? p.nchunks[1] : 0;
    const int nst = 3 * (ce - cb) + n1c;                 // stages of this workgroup
    const int nimg = (ce - cb) + n1c;                    // staged images

    // BatchNorm prologue coefficients of source 0 behind the tiles: [chunk][half][scale x EPS | shift x EPS]
    float* const tab = (float*)(smem + C::LDS_BYTES);
    const bool pro = p.prologue == PSSR_PRO_BN_RELU;
    if (pro) {
        const int cin0 = p.nchunks[0] * KCH;
        for (int i = tid; i < cin0; i += 256) {
            const int c = i / KCH, w_ = i % KCH;
            float* q = tab + ((c * 2 + w_ / EPS) * 2) * EPS + (w_ % EPS);
            q[0] = p.pro_scale[i]; q[EPS] = p.pro_shift[i];
        }
    }
    const float* const tab_t = tab + half * 2 * EPS;
    if constexpr (BN == 128) {
        if (p.epi == PSSR_EPI_HEADQ || (p.flags & PSSR_FLAG_HEADQ)) conv_headq_stage(p, tab, tid, n0);      // (published by the __syncthreads of the loop prologue)
    }

#ifdef PSSR_V3_STAMPS
    unsigned long long st_prev = 0;
    unsigned st_sum[5] = {0, 0, 0, 0, 0};
#define V3_STAMP(I)                                                                                               \
    {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        unsigned long long t_;                                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if ((I) >= 0) st_sum[(I) < 0 ? 0 : (I)] += (unsigned)(t_ - st_prev);                                      \
        st_prev = t_;                                                                                             \
    }
#else
#define V3_STAMP(I)
#endif
    u32x4 a_reg[C::A_ITEMS];
    // image k (0 .. nimg-1) is chunk cb + k of source 0, or chunk k - (ce - cb) of source 1
#define V3_LOAD_A(K)                                                                                              \
    {                                                                                                             \
        const int k_ = (K);                                                                                       \
        if (k_ < ce - cb) {                                                                                       \
            const int so_ = (cb + k_) * 32;                                                                       \
            _Pragma("unroll") for (int it = 0; it < C::A_ITEMS; ++it)                                             \
                a_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra0, (int)a_off0[it], so_, 0)); \
        } else {                                                                                                  \
            const int so_ = (k_ - (ce - cb)) * 32;                                                                \
            _Pragma("unroll") for (int it = 0; it < C::A_ITEMS; ++it)                                             \
                a_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra1, (int)a_off1[it], so_, 0)); \
        }                                                                                                         \
    }
#define V3_COMMIT_A(K)                                                                                            \
    {                                                                                                             \
        const int k_ = (K);                                                                                       \
        char* dst_ = smem + (k_ & 1) * C::A_BYTES;                                                                \
        if (pro && k_ < ce - cb) {                                                                                \
            float tsc_[EPS], tsh_[EPS];                                                                           \
            const float* tq_ = tab_t + (cb + k_) * (4 * EPS);                                                     \
            _Pragma("unroll") for (int e = 0; e < EPS; e += 4) { load4(tq_ + e, tsc_ + e); load4(tq_ + EPS + e, tsh_ + e); } \
            _Pragma("unroll") for (int it = 0; it < C::A_ITEMS; ++it) {                                           \
                if (a_wr[it] >= 0) {                                                                              \
                    u32x4 v = X::bn_relu(a_reg[it], tsc_, tsh_);                                                  \
                    if (!a_ok[it]) v = u32x4{0u, 0u, 0u, 0u};                                                     \
                    *(u32x4*)(dst_ + a_wr[it]) = v;                                                               \
                }                                                                                                 \
            }                                                                                                     \
        } else {                                                                                                  \
            _Pragma("unroll") for (int it = 0; it < C::A_ITEMS; ++it) {                                           \
                if (a_wr[it] >= 0) {                                                                              \
                    u32x4 v = a_reg[it];                                                                          \
                    if (!a_ok[it]) v = u32x4{0u, 0u, 0u, 0u};                                                     \
                    *(u32x4*)(dst_ + a_wr[it]) = v;                                                               \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
    }
    // weights of stage S (0 .. nst-1) -> ring slot S % 3
#define V3_DMA_B(S, SLOT)                                                                                         \
    {                                                                                                             \
        const int s_ = (S);                                                                                       \
        const unsigned dl_ = d_lds + (unsigned)(SLOT) * C::B_STAGE;                                               \
        if (s_ < 3 * (ce - cb)) {                                                                                 \
            const char* src_ = w0n + (long)((cb * 3 + s_) * 3) * tap_stride;       /* chunk cb + s/3, kernel row s % 3 */ \
            _Pragma("unroll") for (int j = 0; j < C::NP; ++j)                                                     \
                v3_dma(__builtin_amdgcn_readfirstlane(dl_ + d_loff[j]), d_voff[j], src_);                         \
        } else {      /* one slice: PPS pieces; waves past them re-issue piece 0 */                                \
            const char* src_ = w1n + (long)(s_ - 3 * (ce - cb)) * tap_stride;                                     \
            const unsigned pq_ = (unsigned)(wave % C::PPS);                                                       \
            v3_dma(dl_ + pq_ * 1024, pq_ * 1024 + lane * 16, src_);                                               \
        }                                                                                                         \
    }
    // end of a stage: the next stage's weights (everything but the youngest N vector-memory operations of this wave) have
    // landed and this wave's LDS stores are done; then all waves meet
#define V3_END_(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // KIND 0: everything landed; 1: the youngest NP operations (one stage of weights) stay in flight; 2: NP + A_ITEMS (+ the next image)
#define V3_END(KIND)                                                                                              \
    {                                                                                                             \
        if constexpr ((KIND) == 0) { V3_END_(0); }                                                                \
        else if constexpr (BN == 128) { if constexpr ((KIND) == 1) { V3_END_(3); } else { V3_END_(6); } }         \
        else { if constexpr ((KIND) == 1) { V3_END_(2); } else { V3_END_(7); } }                                  \
    }

    if (nst > 0) {
        const int ns0 = 3 * (ce - cb);      // stages of source 0
        // ---- prologue: image 0, stages 0 and 1
        V3_STAMP(-1)
        V3_LOAD_A(0)
        V3_DMA_B(0, 0)
        if (nst > 1) V3_DMA_B(1, 1)
        __syncthreads();                    // prologue table
        V3_COMMIT_A(0)
        if (ns0 > 1) { V3_END(1); }         // stage 1 is a full stage: it stays in flight
        else { V3_END(0); }

        V3_STAMP(4)
        unsigned a_cur = a_addr0;           // image of the current chunk
        int s = 0;
        for (int k = 0; k < ce - cb; ++k) {
            const bool more = k + 1 < nimg;             // another image follows
            const bool more0 = k + 1 < ce - cb;         // ... and it is a source-0 chunk (3-piece stages)
            // ---- kernel row 0: requests the next image and the weights of row 2 (always a 3-piece stage)
            if (more) V3_LOAD_A(k + 1)
            V3_DMA_B(s + 2, 2)
            V3_STAMP(0)
            if (!(p.dbg & 2)) v3_row<T, C, 0 * C::ROWB, 0 * C::B_STAGE>(acc, a_cur, b_addr0);
            __builtin_amdgcn_sched_barrier(0);
            V3_STAMP(1)
            if (more) { V3_END(2); } else { V3_END(1); }
            V3_STAMP(3)
            ++s;
            // ---- kernel row 1
            if (s + 2 < nst) V3_DMA_B(s + 2, 0)
            V3_STAMP(0)
            if (!(p.dbg & 2)) v3_row<T, C, 1 * C::ROWB, 1 * C::B_STAGE>(acc, a_cur, b_addr0);
            __builtin_amdgcn_sched_barrier(0);
            V3_STAMP(1)
            if (more0) { V3_END(1); } else { V3_END(0); }
            V3_STAMP(3)
            ++s;
            // ---- kernel row 2: commits the next image
            if (s + 2 < nst) V3_DMA_B(s + 2, 1)
            V3_STAMP(0)
            if (!(p.dbg & 2)) v3_row<T, C, 2 * C::ROWB, 2 * C::B_STAGE>(acc, a_cur, b_addr0);
            __builtin_amdgcn_sched_barrier(0);
            V3_STAMP(1)
            if (more) V3_COMMIT_A(k + 1)
            V3_STAMP(2)
            if (more0) { V3_END(1); } else { V3_END(0); }
            V3_STAMP(3)
            ++s;
            a_cur = a_addr0 + (unsigned)((k + 1) & 1) * C::A_BYTES;
        }
        // ---- 1x1 second source: one single-tap stage per chunk, centre tap of the same halo geometry
        int slot = 0;                       // == s % 3 (ns0 is a multiple of 3)
        for (int k = ce - cb; k < nimg; ++k) {
            const bool more = k + 1 < nimg;
            if (more) V3_LOAD_A(k + 1)
            if (s + 2 < nst) V3_DMA_B(s + 2, (slot + 2) % 3)
            v3_one<T, C, C::ROWB>(acc, a_cur, b_addr0 + (unsigned)slot * C::B_STAGE);
            __builtin_amdgcn_sched_barrier(0);
            if (more) V3_COMMIT_A(k + 1)
            V3_END(0);
            ++s;
            slot = slot == 2 ? 0 : slot + 1;
            a_cur = a_addr0 + (unsigned)((k + 1) & 1) * C::A_BYTES;
        }
    }
#undef V3_LOAD_A
#undef V3_COMMIT_A
#undef V3_DMA_B
#undef V3_END
#undef V3_END_
#undef V3_STAMP
#ifdef PSSR_V3_STAMPS
    if (p.stamps && lane == 0) {
        unsigned* q = p.stamps + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        for (int i = 0; i < 5; ++i) q[i] = st_sum[i];
        q[5] = nst;
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        (void)xcc;
        q[6] = (unsigned)(st_t0 & 0xffffffffu); q[7] = (unsigned)(st_prev & 0xffffffffu);      // kernel entry, end of the main loop
    }
#endif
    // the accumulators were last written by MFMAs inside an asm statement: cover their latency before anything reads them
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    if (p.dbg & 1) return;
    if (p.ksplit > 1) {
        float4* dst = (float4*)p.ws + ((long)blockIdx.x * p.ksplit + blockIdx.y) * (4 * C::MI * C::NJ) * 256 + tid;
#pragma unroll
        for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
            for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dst[((mi * C::NJ + nj) * 4 + q) * 256] = make_float4(acc[mi][nj][4 * q], acc[mi][nj][4 * q + 1], acc[mi][nj][4 * q + 2], acc[mi][nj][4 * q + 3]);
        return;
    }
    if constexpr (BN == 128) {
        if (p.epi == PSSR_EPI_HEADQ || (p.flags & PSSR_FLAG_HEADQ)) {
            conv_headq_epilogue<T, C>(p, acc, (const float*)(smem + C::LDS_BYTES), smem, tid, x0, y0, img0, n0);
            if (p.epi == PSSR_EPI_HEADQ) return;          // FLAG_HEADQ (training): the activation is stored as well
        }
    }
    conv_epilogue8_any<T, BN, C, true>(p, acc, smem, tid, x0, y0, img0, n0);
#ifdef PSSR_V3_STAMPS
    if (p.stamps) {                          // second half of the buffer: [0] epilogue instructions done, [1] its stores acknowledged
        unsigned long long t1, t2;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
        if (lane == 0) {
            unsigned* q = p.stamps + (long)gridDim.x * gridDim.y * 32 + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 2;
            q[0] = (unsigned)(t1 & 0xffffffffu); q[1] = (unsigned)(t2 & 0xffffffffu);
        }
    }
#endif
}

}  // namespace
