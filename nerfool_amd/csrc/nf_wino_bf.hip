// The 3x3 stride-1 convolutions of the ResUNet (a14) as Winograd F(2x2, 3x3) with the element-wise products on the BF16 matrix
// cores, operands split into bf16 parts.  ref: ibrnet/feature_network.py:28-36, 38-78, 127-151 -- the same convolutions as
// nf_wino.hip (same tile geometry, same window staging, same per-wave LDS weight ring, same output transform); what differs is
// the product  M_xi[k][tile] = sum_c U_xi[k][c] V_xi[c][tile]:
//   * v_mfma_f32_32x32x16_bf16 multiplies 16 input channels per instruction at 16x the rate of v_mfma_f32_32x32x2_f32, whose
//     rate equals the fp32 vector rate (MI355X_MICROARCH.md) and whose time ADDS to the vector instructions around it;
//   * NS = 3 ("bf16x3"): every fp32 operand is written as hi + mid + lo, three bf16 values of 8 significant bits each (the
//     transformed weights U on the host, the transformed inputs V in the kernel: x -> bf16(x), the exact remainder -> bf16, the
//     exact remainder of that -> bf16), and the product is the six cross terms of order <= 2^-16:
//         hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid          (fp32 accumulation inside the matrix instruction)
//     The dropped terms are <= 2^-24 of the product -- fp32 rounding level -- so this form is a drop-in for the fp32 kernel (the
//     float64 layer tests hold at the same 5e-6) at 6/16 of its matrix-pipe time;
//   * NS = 2 ("bf16x2", round 5 -- the BACKWARD-DATA passes only): hi + mid, the three cross terms of order <= 2^-8
//         hi.hi + hi.mid + mid.hi
//     i.e. operands of 16 significant bits; the dropped terms (hi.lo, lo.hi, mid.mid) are <= 3 x 2^-16 of a product, zero-mean.  The
//     forward pass decides ReLU signs and feeds the renderer, so it stays at NS = 3; the gradient has no discontinuity to flip and its
//     parity bar (1e-3 of the float64 gradient, measured 4e-5 .. 5e-4 before) absorbs 1e-5 per layer: half the matrix instructions,
//     two thirds of the streamed records, 7 instead of 11 vector instructions per split pair;
//   * NS = 1 ("bf16"): plain bf16 operands, one instruction per 16 channels: BASELINE config 5's opt-in precision
//     (ibrnet_precision = 'bf16'), 8 significant bits per operand, stated tolerance in DESIGN.md.
// Work split: wave w owns row w of the 4x4 transformed tile as in nf_wino.hip.  A chunk of 16 input channels is ONE k-block of the
// bf16 instruction: lane (tile t, half hh) supplies the 8 channels 2 j + hh, j = 0..7 -- the channels the fp32 kernel's 8 steps
// gave it one at a time.  Per chunk a wave reads its window rows once (8 channels x 4 row-combined values in registers), and
// then, for nu = 0..3: derives the 8 transformed values of column nu, splits them, and runs KB steps of NS (NS + 1) / 2
// instructions each against the A parts of (nu, kb) streamed through the ring (one 1 KB piece per part: 64 lanes x 8 bf16).
#include <string.h>

#include "nf_wino.h"

typedef __bf16 wb8 __attribute__((ext_vector_type(8)));
typedef unsigned wu4 __attribute__((ext_vector_type(4)));
#define WB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

static inline uint16_t wb_rne(float f) {      // HOST: round to nearest even (finite weights)
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float wb_up(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

extern "C" int64_t nf_wino_bf_pack_floats(int c_out, int c_in, int k_per_group, int n_split) {
    const int groups = (c_out + k_per_group - 1) / k_per_group, chunks = (c_in + WN_CC - 1) / WN_CC, KB = k_per_group / 32;
    // per (group, wave, chunk): 4 nu x KB steps of n_split pieces of 256 floats (64 lanes x 8 bf16); + room for the last prefetch
    return (int64_t)groups * 4 * chunks * 4 * KB * n_split * 256 + 4096;
}

/* HOST: weight [c_out][c_in][3][3] -> the bf16 parts of U = G g G^T as the four waves' streams
 * [group][wave][chunk][nu][kb][part][lane][8 bf16]; lane (k = kb 32 + (lane & 31), hh = lane >> 5), element j <-> channel 2 j + hh
 * of the chunk.  backward != 0: the backward-data convolution (rotated taps, roles of c_out / c_in swapped), as nf_wino_pack. */
extern "C" int nf_wino_bf_pack(const float* weight, int c_out, int c_in, int backward, int k_per_group, int n_split, float* out) {
    static const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int N = backward ? c_in : c_out, C = backward ? c_out : c_in;
    if (k_per_group % 32 != 0 || k_per_group < 32 || n_split < 1 || n_split > 3) return 1;
    const int KB = k_per_group / 32, groups = (N + k_per_group - 1) / k_per_group, chunks = (C + WN_CC - 1) / WN_CC;
    const int64_t total = nf_wino_bf_pack_floats(N, C, k_per_group, n_split);
    for (int64_t i = total - 4096; i < total; ++i) out[i] = 0.f;
    uint16_t* piece = reinterpret_cast<uint16_t*>(out);
    for (int g = 0; g < groups; ++g)
        for (int w = 0; w < 4; ++w)
            for (int ch = 0; ch < chunks; ++ch)
                for (int nu = 0; nu < 4; ++nu)
                    for (int kb = 0; kb < KB; ++kb, piece += (size_t)n_split * 512)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int k = g * k_per_group + kb * 32 + (lane & 31), c = ch * WN_CC + 2 * j + (lane >> 5);
                                float u = 0.f;
                                if (k < N && c < C) {
                                    for (int a = 0; a < 3; ++a)
                                        for (int b = 0; b < 3; ++b) {
                                            const float gv = backward ? weight[(((size_t)c * c_in + k) * 3 + (2 - a)) * 3 + (2 - b)]
                                                                      : weight[(((size_t)k * c_in + c) * 3 + a) * 3 + b];
                                            u += G[w][a] * gv * G[nu][b];
                                        }
                                }
                                float rem = u;
                                for (int p = 0; p < n_split; ++p) {
                                    const uint16_t b = wb_rne(rem);
                                    piece[(size_t)p * 512 + lane * 8 + j] = b;
                                    rem -= wb_up(b);          // exact: the remainder of a rounding to 8 significant bits
                                }
                            }
    return 0;
}

// two fp32 values -> their bf16 roundings (packed) and the exact remainders
__device__ __forceinline__ unsigned wb_split_pair(float& x0, float& x1) { return nf_split_pair_bf16(x0, x1); }

#ifndef WB_A_GLOBAL
#define WB_A_GLOBAL 0     // 1: A parts from global memory into registers one step ahead (no LDS ring) -- tuning form, see load_a
#endif
#ifndef WB_PIPE_E
#define WB_PIPE_E 1       // the next chunk's window values are read under the last steps of the current chunk
#endif
#define NF_LAUNDER_WB(x) asm volatile("" : "+v"(x))
#ifdef WB_PHASE_TIMERS
// TUNING BUILD ONLY (NF_VARIANT_SRC=nf_wino_bf.hip tools/build_variant.sh <name> -DWB_PHASE_TIMERS; tools/experimental/wino_phases.py):
// s_memtime ticks a wave spends in the phases of a chunk, one row per wave (no atomics: they would jam the memory path)
// [0] chunks, [1] whole chunk, [2] window reads + row combination (E), [3] waits for streamed records, [4] window hand-over + barrier,
// [6] one timer's own cost (two back-to-back readings), [7] prologue, [8] output stage
#define WB_PH_SLOTS 32768
__device__ unsigned long long wb_phase[WB_PH_SLOTS * 16];
__device__ __forceinline__ unsigned long long wb_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
extern "C" int nf_wino_bf_phase_read(unsigned long long* out16, int reset) {
    static unsigned long long host[WB_PH_SLOTS * 16];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(wb_phase), sizeof(host)) != hipSuccess) return 1;
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    for (int r = 0; r < WB_PH_SLOTS; ++r)
        for (int i = 0; i < 16; ++i) out16[i] += host[r * 16 + i];
    if (reset) {
        memset(host, 0, sizeof(host));
        if (hipMemcpyToSymbol(HIP_SYMBOL(wb_phase), host, sizeof(host)) != hipSuccess) return 1;
    }
    return 0;
}
#define WB_T(var) const unsigned long long var = wb_now()
#define WB_ACC(i, d) ph[i] += (d)
#else
#define WB_T(var)
#define WB_ACC(i, d)
#endif

template <int KB, int NS>
__global__ void __launch_bounds__(256, 2) k_wino3x3_bf(const float* __restrict__ rec, const float* __restrict__ x, WnTensor xi, int Hi, int Wi,
                                                        int pad, float* __restrict__ y, WnTensor yo, int Ho, int Wo, int C, int K, int groups, int n_img) {
    constexpr int WN_PR = 10, WN_CH = WN_PR * WN_PS, WN_BUF = WN_CC * WN_CH;
    constexpr int STEP = NS * 256;                    // floats of one step's records: NS pieces of 1 KB
    constexpr int NSTEP = 4 * KB;                     // steps per chunk: (nu, kb)
    constexpr int WN_SLOTS = 4, DIST = WN_SLOTS - 1;
    static_assert(NSTEP % WN_SLOTS == 0, "a step's slot is a compile-time constant");
#ifdef WB_EXP_ONE_PIECE
    constexpr int PCS = 1;        // TIMING EXPERIMENT (wrong results): one piece streamed per step whatever NS
#else
    constexpr int PCS = NS;
#endif
    HIP_DYNAMIC_SHARED(float, smem)
    float* ex = smem;
    const int lane = threadIdx.x & 63, w = wn_uniform(threadIdx.x >> 6);
    const int t = lane & 31, hh = lane >> 5, tr = t >> 3, tc = t & 7;
    const int tiles_x = (Wo + 15) >> 4, tiles_y = (Ho + 7) >> 3;
    const int n_tiles = tiles_x * tiles_y * n_img, per_xcd = (n_tiles + 7) >> 3;
    const int slot0 = blockIdx.x >> 3, tl = slot0 / groups, grp = slot0 - tl * groups;
    const int tile = (blockIdx.x & 7) * per_xcd + tl;
    if (tile >= n_tiles) return;                // whole workgroup, before any barrier
    const int n = tile / (tiles_x * tiles_y), trem = tile - n * (tiles_x * tiles_y);
    const int oy0 = (trem / tiles_x) * 8, ox0 = (trem % tiles_x) * 16;
    const int iy0 = oy0 - pad, ix0 = ox0 - pad;
    const int chunks = (C + WN_CC - 1) / WN_CC;
    const float* xn = x + n * xi.ns;
    const float* wsrc = rec + ((size_t)(grp * 4 + w) * chunks) * (NSTEP * STEP);
    float* ring = smem + 2 * WN_BUF + w * (WN_SLOTS * STEP);
    const int ra = w == 0 ? 0 : (w == 2 ? 2 : 1), rb = w == 0 ? 2 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
    const float sb = w == 1 ? 1.f : -1.f;
    const int lbase = (2 * tr) * WN_PS + 2 * tc;

    // ---- window staging: identical to nf_wino.hip (480 segments of 6 floats per chunk, always three 8-byte loads each)
    const int seg_a = threadIdx.x, seg_b = threadIdx.x + 256;
    const bool has_b = seg_b < WN_CC * WN_PR * 3;
    float pre[2][6];
    auto seg_geom = [&](int chunk, int sg, int& c, int& gy, int& gx0, bool& row_ok) {
        const int j = sg / 3, part = sg - 3 * j;
        const int ch = j / WN_PR, pr = j - ch * WN_PR;
        c = chunk * WN_CC + ch, gy = iy0 + pr, gx0 = ix0 + 6 * part;
        row_ok = c < C && gy >= 0 && gy < Hi;
    };
    auto fetch_seg = [&](int chunk, int sg, float (&raw)[6]) {
        int c, gy, gx0;
        bool row_ok;
        seg_geom(chunk, sg, c, gy, gx0, row_ok);
        const float* src = xn + (row_ok ? c : 0) * xi.cs + (row_ok ? gy : 0) * xi.rs;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int a = gx0 + 2 * q;
            const int start = a < 0 ? 0 : (a > Wi - 2 ? Wi - 2 : a);
            const w2f v = *reinterpret_cast<const w2f*>(src + start);
            raw[2 * q] = v[0];
            raw[2 * q + 1] = v[1];
        }
    };
    auto commit_seg = [&](int chunk, int sg, const float (&raw)[6]) {
        int c, gy, gx0;
        bool row_ok;
        seg_geom(chunk, sg, c, gy, gx0, row_ok);
        const int j = sg / 3, part = sg - 3 * j;
        const int ch = j / WN_PR, pr = j - ch * WN_PR;
        float* dst = smem + (chunk & 1) * WN_BUF + ch * WN_CH + pr * WN_PS + 6 * part;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int a = gx0 + 2 * q;
            const bool same = a >= 0 && a <= Wi - 2;
            const bool ok0 = row_ok && a >= 0 && a < Wi, ok1 = row_ok && a + 1 >= 0 && a + 1 < Wi;
            const float v0 = ok0 ? (same ? raw[2 * q] : raw[2 * q + 1]) : 0.f;
            const float v1 = ok1 ? (same ? raw[2 * q + 1] : raw[2 * q]) : 0.f;
            *reinterpret_cast<w2a*>(dst + 2 * q) = w2a{v0, v1};
        }
    };
    auto fetch = [&](int chunk) {
        fetch_seg(chunk, seg_a, pre[0]);
        if (has_b) fetch_seg(chunk, seg_b, pre[1]);
    };
    auto commit = [&](int chunk) {
        commit_seg(chunk, seg_a, pre[0]);
        if (has_b) commit_seg(chunk, seg_b, pre[1]);
    };

    w16 acc[4][KB];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][kb][r] = 0.f;
#ifdef WB_PHASE_TIMERS
    unsigned long long ph[9] = {};
#endif
    WB_T(t_begin);

    // ---- weight ring: a step = NS pieces of 1 KB under one M0 set-up, issued DIST steps ahead of their use
    const float* wnext = wsrc;
    auto issue_step = [&](int fixed) {
        float* dst = ring + (fixed % WN_SLOTS) * STEP;
#if defined(__HIP_DEVICE_COMPILE__)
        unsigned keep;
        const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)dst;
        if (PCS == 3)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                         "global_load_lds_dwordx4 %1, %3 offset:1024\n\tglobal_load_lds_dwordx4 %1, %3 offset:2048\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(lane * 16), "s"(lds), "s"(wnext)
                         : "memory");
        else
            wn_dma16xn<PCS>(wnext, dst, lane);
#else
        for (int j = 0; j < 4 * NS; ++j) dst[(j >> 2) * 256 + 4 * lane + (j & 3)] = wnext[(j >> 2) * 256 + 4 * lane + (j & 3)];
#endif
        wnext += STEP;
    };
    wu4 opa[2][NS];         // a step's A parts, read one step ahead of their use
    auto read_a = [&](wu4 (&wa)[NS], int s1) {
        const float* rs = ring + (s1 % WN_SLOTS) * STEP + 4 * lane;
#pragma unroll
        for (int p = 0; p < NS; ++p) wa[p] = *reinterpret_cast<const wu4*>(rs + p * 256);
    };
    // WB_A_GLOBAL (tuning form): the A parts of the NEXT step straight from global memory (L2) into the registers the LDS reads filled -- no
    // ring, no LDS-DMA, no LDS read; the compiler counts the loads (vmcnt) itself
    auto load_a = [&](wu4 (&wa)[NS]) {
        const float* src = wnext + 4 * lane;
#pragma unroll
        for (int p = 0; p < NS; ++p) wa[p] = *reinterpret_cast<const wu4*>(src + p * 256);
        wnext += STEP;
    };
    constexpr bool AG = WB_A_GLOBAL != 0;
    if (AG) {
        load_a(opa[0]);     // step 0 (older than the window fetch below: a wait for it does not wait for the window)
        fetch(0);
        commit(0);
        __syncthreads();
    } else {
#pragma unroll
        for (int q = 0; q < DIST; ++q) issue_step(q);     // (at least NSTEP >= 4 > DIST steps in the stream)
        fetch(0);
        commit(0);          // the compiler drains the VM counter for the fetched registers here: the first DIST steps have landed too
        __syncthreads();
        read_a(opa[0], 0);
    }
    WB_T(t_pro);
    WB_ACC(7, t_pro - t_begin);

    // ---- a chunk's window rows -> the four row-combined values E[j][0..3] of the lane's 8 channels (channel 2 j + hh).  The values of
    //      chunk c + 1 are read UNDER the last KB steps of chunk c (WB_PIPE_E): E is dead once column 3 has been split, so its registers take
    //      the next chunk's values -- four channels per step, raw window values in flight while the step's six products run -- and the
    //      hand-over (commit + barrier) moves in front of those steps.  (Read at the top of the chunk, right behind the barrier, they were
    //      17 % of a chunk with nothing of the wave's own to overlap: profiles/r04_wino_phase_timers.txt.)
    constexpr bool PIPE = WB_PIPE_E != 0;
    constexpr int HAND = PIPE ? NSTEP - 1 - KB : NSTEP - 1;        // the step the hand-over sits in front of
    float E[8][4];
    struct RawE { w2a a0, a1, b0, b1; };
    auto e_load = [&](const float* pbuf, int j, RawE& r) {
#ifdef WB_EXP_NO_E
        const float* pa = pbuf + hh * WN_CH;          // TIMING EXPERIMENT (wrong results): one channel's rows for all eight
#else
        const float* pa = pbuf + (2 * j + hh) * WN_CH;
#endif
        r.a0 = *reinterpret_cast<const w2a*>(pa + ra * WN_PS), r.a1 = *reinterpret_cast<const w2a*>(pa + ra * WN_PS + 2);
        r.b0 = *reinterpret_cast<const w2a*>(pa + rb * WN_PS), r.b1 = *reinterpret_cast<const w2a*>(pa + rb * WN_PS + 2);
    };
    auto e_combine = [&](int j, const RawE& r) {      // sb = +-1: the fused form is exact
        E[j][0] = fmaf(sb, r.b0[0], r.a0[0]), E[j][1] = fmaf(sb, r.b0[1], r.a0[1]), E[j][2] = fmaf(sb, r.b1[0], r.a1[0]), E[j][3] = fmaf(sb, r.b1[1], r.a1[1]);
    };
    if (PIPE) {
        const float* pbuf = smem + lbase;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            RawE r;
            e_load(pbuf, j, r);
            e_combine(j, r);
        }
    }

    auto run_chunk = [&](auto last_c, int chunk) {
        constexpr bool last = decltype(last_c)::value;
        const float* pbuf = smem + (chunk & 1) * WN_BUF + lbase;
        const float* pnext = smem + ((chunk + 1) & 1) * WN_BUF + lbase;
        WB_T(t_c0);
        {
            WB_T(t_c0b);
            WB_ACC(6, t_c0b - t_c0);
        }
        if (!PIPE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                RawE r;
                e_load(pbuf, j, r);
                e_combine(j, r);
            }
#ifdef WB_PHASE_TIMERS
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) NF_LAUNDER_WB(E[j][i]);
#endif
        }
        WB_T(t_e);
        WB_ACC(2, t_e - t_c0);
        // the next chunk's window travels global -> registers under this chunk's products, -> LDS in front of step HAND
        if (AG && NSTEP > 1) load_a(opa[1]);          // step 1's parts: issued in front of the window fetch (older than it)
#ifndef WB_EXP_NO_FETCH
        if (!last) fetch(chunk + 1);
#endif
        RawE nx[4];          // (PIPE) raw window values of four channels of the next chunk, in flight under a step's products
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            // column nu of B^T d B for the 8 channels, split into bf16 parts: part p of channel pair (2 q, 2 q + 1) -> bv[p][q]
            wu4 bv[NS];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v0, v1;
                if (nu == 0) { v0 = E[2 * q][0] - E[2 * q][2]; v1 = E[2 * q + 1][0] - E[2 * q + 1][2]; }
                else if (nu == 1) { v0 = E[2 * q][1] + E[2 * q][2]; v1 = E[2 * q + 1][1] + E[2 * q + 1][2]; }
                else if (nu == 2) { v0 = E[2 * q][2] - E[2 * q][1]; v1 = E[2 * q + 1][2] - E[2 * q + 1][1]; }
                else { v0 = E[2 * q][1] - E[2 * q][3]; v1 = E[2 * q + 1][1] - E[2 * q + 1][3]; }
#ifdef WB_EXP_NO_SPLIT
#pragma unroll
                for (int p = 0; p < NS; ++p) bv[p][q] = __builtin_bit_cast(unsigned, p == 0 ? v0 : v1);       // TIMING EXPERIMENT (wrong results)
#else
#pragma unroll
                for (int p = 0; p < NS; ++p) bv[p][q] = wb_split_pair(v0, v1);
#endif
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int s = nu * KB + kb;           // compile-time after unrolling
                // the chunk hand-over: in front of the last step (as in nf_wino.hip), or in front of column 3's steps (PIPE)
                if (s == HAND && !last) {
                    WB_T(t_h0);
#ifndef WB_EXP_NO_FETCH
                    commit(chunk + 1);
#endif
#ifndef WB_EXP_NO_BARRIER
                    __syncthreads();
#endif
                    WB_T(t_h1);
                    WB_ACC(4, t_h1 - t_h0);
                }
                if (AG) {
                    // the parts of step s + 1 (the first step of the next chunk behind the last one), one step ahead of their use
                    if (s >= 1 && (!last || s < NSTEP - 1)) load_a(opa[(s + 1) & 1]);
                }
                // [A] records of step s + DIST; never past the end of the stream
                if (!AG && (!last || s + DIST < NSTEP)) issue_step(s + DIST);
                if (!AG && (!last || s < NSTEP - 1)) {
                    WB_T(t_w0);
                    // [C] the records of step s + 1 were issued DIST - 1 steps ago; behind them in the counter: the steps issued since
                    // (fewer at the end of the stream) and, while the window fetch of this chunk is younger than them, its 6 loads (the
                    // fetch is issued at the top of the chunk, IN FRONT of step 0's refill: it is younger than the records of steps
                    // 1 .. DIST - 1 only)
                    if (!last) {
#ifdef WB_EXP_NO_FETCH
                        if (s < DIST - 1) wn_wait_vm<(DIST - 1) * PCS>();
#else
                        if (s < DIST - 1) wn_wait_vm<(DIST - 1) * PCS + WN_FETCH_OPS>();
#endif
                        else wn_wait_vm<(DIST - 1) * PCS>();
                    } else {
                        // (s is a constant after unrolling: the switch folds to one wait)
                        const int left = NSTEP - 2 - s < DIST - 1 ? NSTEP - 2 - s : DIST - 1;
                        static_assert(DIST - 1 == 2, "cases below");
                        if (left == 2) wn_wait_vm<2 * PCS>();
                        else if (left == 1) wn_wait_vm<PCS>();
                        else wn_wait_vm<0>();
                    }
                    WB_T(t_w1);
                    WB_ACC(3, t_w1 - t_w0);
                    // [D] next step's A parts
                    read_a(opa[(s + 1) & 1], s + 1);
                }
                if (PIPE && !last && nu == 3) {
                    // column 3 is split: E is free.  The raw values read under the previous step have arrived (the wait above): combine
                    // them, then start this step's four channels (KB = 1: all eight in two batches under the one step)
                    constexpr int per = 8 / KB;                    // channels per step
                    if (kb > 0) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) e_combine(per * (kb - 1) + jj, nx[jj]);
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) e_load(pnext, per * kb + jj, nx[jj]);
                    if (per == 8) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) e_combine(jj, nx[jj]);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) e_load(pnext, 4 + jj, nx[jj]);
                    }
                }
                // [E] this step's products: the cross terms of order <= 2^-16
                const wu4* a = opa[s & 1];
#ifdef WB_EXP_NO_MFMA
                // TIMING EXPERIMENT (wrong results): one vector instruction in place of each matrix instruction, operands kept alive
#define WB_PROD(pa, pb) acc[nu][kb][(pa) + 3 * (pb)] += __builtin_bit_cast(float, a[pa][0]) * __builtin_bit_cast(float, bv[pb][0])
#else
#define WB_PROD(pa, pb) acc[nu][kb] = WB_MFMA(__builtin_bit_cast(wb8, a[pa]), __builtin_bit_cast(wb8, bv[pb]), acc[nu][kb])
#endif
                WB_PROD(0, 0);
                if (NS >= 2) {
                    WB_PROD(0, 1);
                    WB_PROD(1, 0);
                }
                if (NS == 3) {
                    WB_PROD(0, 2);
                    WB_PROD(2, 0);
                    WB_PROD(1, 1);
                }
#undef WB_PROD
            }
        }
        if (PIPE && !last) {         // the last four channels, read under the last step
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) e_combine(4 + jj, nx[jj]);
        }
        WB_T(t_c1);
        WB_ACC(1, t_c1 - t_c0);
        WB_ACC(0, 1);
    };
    for (int chunk = 0; chunk + 1 < chunks; ++chunk) run_chunk(std::false_type{}, chunk);
    run_chunk(std::true_type{}, chunks - 1);
    WB_T(t_out0);
    __syncthreads();
    // ---- output transform Y = A^T M A: identical to nf_wino.hip
    const int kbase = grp * (32 * KB);
    float* yn = y + n * yo.ns;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            w4f c0, c1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * r4 + i;
                const float a0 = acc[0][kb][r], a1 = acc[1][kb][r], a2 = acc[2][kb][r], a3 = acc[3][kb][r];
                c0[i] = a0 + a1 + a2;
                c1[i] = a1 - a2 - a3;
            }
            float* e = ex + kb * 8192 + w * 2048 + r4 * 256 + 4 * lane;
            *reinterpret_cast<w4f*>(e) = c0;
            *reinterpret_cast<w4f*>(e + 1024) = c1;
        }
    __syncthreads();
    {
        const int oy = w & 1, rsel = 8 * (w >> 1);
        const int row = oy0 + 2 * tr + oy, col = ox0 + 2 * tc;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            float o[2][8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float* e = ex + kb * 8192 + j * 1024 + ((rsel >> 2) + hf) * 256 + 4 * lane;
                    const w4f m0 = *reinterpret_cast<const w4f*>(e), m1 = *reinterpret_cast<const w4f*>(e + 2048);
                    const w4f m2 = *reinterpret_cast<const w4f*>(e + 4096), m3 = *reinterpret_cast<const w4f*>(e + 6144);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[j][4 * hf + i] = oy == 0 ? m0[i] + m1[i] + m2[i] : m1[i] - m2[i] - m3[i];
                }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = kbase + 32 * kb + wn_nidx(rsel + q, hh);
#ifdef WB_EXP_NO_STORE
                if (o[0][q] == 123.456f && k < K && row < Ho) yn[k * yo.cs + row * yo.rs + col] = o[1][q];      // TIMING EXPERIMENT: (almost) no stores
#else
                if (k < K && row < Ho) {
                    float* yp = yn + k * yo.cs + row * yo.rs + col;
                    if (col + 1 < Wo) *reinterpret_cast<w2f*>(yp) = w2f{o[0][q], o[1][q]};
                    else if (col < Wo) yp[0] = o[0][q];
                }
#endif
            }
        }
    }
#ifdef WB_PHASE_TIMERS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WB_T(t_out1);
    WB_ACC(8, t_out1 - t_out0);
    if (lane == 0) {
        unsigned long long* row = wb_phase + (size_t)((blockIdx.x * 4 + w) % WB_PH_SLOTS) * 16;
        for (int i = 0; i < 9; ++i) row[i] += ph[i];
    }
#endif
}

template <int KB, int NS>
static void wb_launch(const float* records, const float* x, WnTensor xi, int Hi, int Wi, int pad, float* y, WnTensor yo, int Ho, int Wo,
                      int n_img, int c_in, int c_out, int groups, hipStream_t st) {
    const int n_tiles = ((Wo + 15) / 16) * ((Ho + 7) / 8) * n_img;
    dim3 grid((unsigned)(8 * ((n_tiles + 7) / 8) * groups));
    // two windows + four rings of 4 slots x NS KB; the output exchange (KB * 8192 floats) re-uses the same memory
    constexpr size_t staging = sizeof(float) * (2 * WN_CC * 10 * WN_PS + 4 * 4 * NS * 256);
    constexpr size_t smem = staging > sizeof(float) * KB * 8192 ? staging : sizeof(float) * KB * 8192;
    static_assert(2 * smem <= 160 * 1024, "two workgroups per CU");
    static bool once_on[NF_MAX_DEVICES] = {};
    bool& once = once_on[nf_current_device()];
    if (!once) {
        (void)hipFuncSetAttribute((const void*)k_wino3x3_bf<KB, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        once = true;
    }
    hipLaunchKernelGGL((k_wino3x3_bf<KB, NS>), grid, dim3(256), smem, st, records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, c_in, c_out, groups, n_img);
}

/* nf_conv3x3_wino with the products on the bf16 matrix cores: records from nf_wino_bf_pack(..., k_per_group, n_split); n_split = 3:
 * fp32-grade (six cross terms of the three-way operand split), n_split = 1: plain bf16 operands.  Same tensors, same geometry. */
extern "C" int nf_conv3x3_wino_bf(const float* records, int k_per_group, int n_split, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h,
                                  int Hi, int Wi, int pad, float* y, int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img,
                                  int c_in, int c_out, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && c_in >= 1 && c_out >= 1 && Hi >= 1 && Wi >= 2 && Ho >= 1 && Wo >= 1 && (k_per_group == 64 || k_per_group == 32) &&
                   n_split >= 1 && n_split <= 3,
               "nf_conv3x3_wino_bf: bad arguments (k_per_group %d, n_split %d)", k_per_group, n_split);
    const int groups = (c_out + k_per_group - 1) / k_per_group;
    const WnTensor xi = {xs_n, xs_c, xs_h}, yo = {ys_n, ys_c, ys_h};
    hipStream_t st = (hipStream_t)stream;
    if (k_per_group == 32) {
        if (n_split == 3) wb_launch<1, 3>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);
        else if (n_split == 2) wb_launch<1, 2>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);
        else wb_launch<1, 1>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);
    } else {
        if (n_split == 3) wb_launch<2, 3>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);
        else if (n_split == 2) wb_launch<2, 2>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);
        else wb_launch<2, 1>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);
    }
    NF_LAUNCH_CHECK("nf_conv3x3_wino_bf");
    return 0;
}
