// 3x3 stride-1 convolutions of the ResUNet (a14) as Winograd F(2x2, 3x3) on the fp32 matrix cores.
// ref: ibrnet/feature_network.py:28-36, 38-78, 127-151 (every 3x3 convolution of the network that has stride 1; the
// reflect padding is already in the input -- csrc/nf_cnn.hip writes pre-padded activations -- so the convolution itself
// runs with padding 0, and its backward-data pass is the same kernel on the zero-extended gradient with rotated weights).
//
//   Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A      per 4x4 input tile d -> 2x2 outputs, 16 element-wise products
// The 16 products are 16 independent GEMMs  M_xi[k][tile] = sum_c U_xi[k][c] V_xi[c][tile]  on v_mfma_f32_32x32x2_f32:
// rows = 32 output channels (A operand = transformed weights, host-packed records streamed from L2), columns = 32 tiles
// (a 4 x 8 block of tiles), k = input channels.  Wave w of the 4-wave workgroup owns row w of the
// 4x4 transformed tile (xi = (w, nu), nu = 0..3): it builds ITS rows of B^T d B straight from the raw input patch in LDS
// (two patch rows, one 4-wide column transform: 8 adds per 4 B operands) -- V never exists in memory -- and keeps
// 4 x KB accumulator tiles.  The column half of A^T M A is local to a wave, the row half crosses the waves through LDS.
// 2.25x fewer multiplies than the direct form.
// Data movement: the accumulators take 128 of a lane's registers, so nothing else may live in registers for long.
//   * weights: every wave streams ITS records from L2 into a private LDS ring by LDS-DMA (global_load_lds_dwordx4: 1 KB per
//     instruction, no register in between), three steps ahead, and reads a step's A operands back with ds_read_b128 -- one step
//     before the multiplications that use them (two register sets), like the B operands, so the LDS latency hides behind the
//     previous step's eight MFMAs;
//   * input: the next chunk's raw window is fetched into 12 registers at the top of a chunk and committed to the second LDS
//     buffer at its end (one barrier per chunk).
//   Both streams share the wave's in-order VM counter.  The DMA is issued from inline asm and waited for with exact counted
//   s_waitcnt vmcnt(N) (the window fetch is a FIXED six loads per wave, see fetch_seg), so a weight wait never has to sit out the
//   HBM latency of the window fetch issued behind it -- with the earlier register ring that coupling cost 12-23 % (measured by
//   dropping the fetch).  Two workgroups per CU (64 KB of LDS each).
// Measured and dropped (tools/bench_conv3x3.py, 64 -> 64 at 189 x 252): two tile blocks per wave at one workgroup per CU
// (+30 %), B operands loaded straight from global memory without LDS (+50 %), weights through a four-deep register ring (the
// previous form of this kernel), the window by LDS-DMA as well (15 single-float pieces per wave and chunk with per-lane
// addresses, nothing left for the compiler to drain: +3-6 % -- more address arithmetic and thinner memory requests than the six
// 8-byte loads), 32 output channels per workgroup at three workgroups per CU (+-2 %).
#include "nf_wino.h"

extern "C" int64_t nf_wino_pack_floats(int c_out, int c_in, int k_per_group) {
    const int groups = (c_out + k_per_group - 1) / k_per_group, chunks = (c_in + WN_CC - 1) / WN_CC;
    return (int64_t)groups * 4 * chunks * 8 * 4 * (k_per_group / 32) * 64 + 4096;      // + room for the last prefetch past the end
}

/* HOST: weight [c_out][c_in][3][3] -> U = G g G^T as the four waves' streams [group][wave][chunk][step][piece][lane][4].
 * backward != 0 packs the backward-data convolution: g'[c][k][a][b] = g[k][c][2-a][2-b], roles of c_out / c_in swapped
 * (the records then describe a convolution with `c_in` OUTPUT channels). */
extern "C" int nf_wino_pack(const float* weight, int c_out, int c_in, int backward, int k_per_group, float* out) {
    static const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int N = backward ? c_in : c_out, C = backward ? c_out : c_in;      // output / input channels of the packed convolution
    if (k_per_group % 32 != 0 || k_per_group < 32) return 1;
    const int KB = k_per_group / 32, groups = (N + k_per_group - 1) / k_per_group, chunks = (C + WN_CC - 1) / WN_CC;
    const int64_t total = nf_wino_pack_floats(N, C, k_per_group);
    for (int64_t i = total - 4096; i < total; ++i) out[i] = 0.f;
    // a step's 4 * KB records (index i = nu * KB + kb) travel as KB pieces of 1 KB: piece p = records 4p .. 4p + 3, lane-major
    // ([lane][4]) -- one global_load_lds_dwordx4 moves a piece, one ds_read_b128 hands a lane its four A operands
    float* step = out;
    for (int g = 0; g < groups; ++g)
        for (int w = 0; w < 4; ++w)
            for (int ch = 0; ch < chunks; ++ch)
                for (int s = 0; s < 8; ++s, step += 4 * KB * 64)
                    for (int nu = 0; nu < 4; ++nu)
                        for (int kb = 0; kb < KB; ++kb)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int k = g * k_per_group + kb * 32 + (lane & 31), c = ch * WN_CC + 2 * s + (lane >> 5);
                                float u = 0.f;
                                if (k < N && c < C) {
                                    for (int a = 0; a < 3; ++a)
                                        for (int b = 0; b < 3; ++b) {
                                            const float gv = backward ? weight[(((size_t)c * c_in + k) * 3 + (2 - a)) * 3 + (2 - b)]
                                                                      : weight[(((size_t)k * c_in + c) * 3 + a) * 3 + b];
                                            u += G[w][a] * gv * G[nu][b];
                                        }
                                }
                                const int i = nu * KB + kb;
                                step[((i >> 2) * 64 + lane) * 4 + (i & 3)] = u;
                            }
    return 0;
}

template <int KB>
__global__ void __launch_bounds__(256, KB == 1 ? 3 : 2) k_wino3x3(const float* __restrict__ rec, const float* __restrict__ x, WnTensor xi, int Hi, int Wi,
                                                    int pad, float* __restrict__ y, WnTensor yo, int Ho, int Wo, int C, int K, int groups, int n_img) {
    // LDS: two buffers of the raw input window of a 16-channel chunk | one weight ring per wave; after the last chunk the same
    // memory carries the row half of the output transform from wave to wave
    constexpr int WN_PR = 10, WN_CH = WN_PR * WN_PS, WN_BUF = WN_CC * WN_CH;
    constexpr int STEP = 4 * KB * 64;                 // floats of one step's records = KB pieces of 256
    constexpr int WN_SLOTS = wn_slots(KB);
    constexpr int DIST = WN_SLOTS - 1;
    HIP_DYNAMIC_SHARED(float, smem)
    float* ex = smem;
    const int lane = threadIdx.x & 63, w = wn_uniform(threadIdx.x >> 6);
    const int t = lane & 31, hh = lane >> 5, tr = t >> 3, tc = t & 7;
    // workgroup -> (tile, output-channel group).  The hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each
    // with its own L2: the ids that share an XCD (id % 8) walk a contiguous run of tiles in raster order, the `groups` channel
    // groups of a tile back to back -- the groups re-read the same input window, neighbouring tiles share its halo.
    const int tiles_x = (Wo + 15) >> 4, tiles_y = (Ho + 7) >> 3;
    const int n_tiles = tiles_x * tiles_y * n_img, per_xcd = (n_tiles + 7) >> 3;
    const int slot = blockIdx.x >> 3, tl = slot / groups, grp = slot - tl * groups;
    const int tile = (blockIdx.x & 7) * per_xcd + tl;
    if (tile >= n_tiles) return;                // whole workgroup, before any barrier
    const int n = tile / (tiles_x * tiles_y), trem = tile - n * (tiles_x * tiles_y);
    const int oy0 = (trem / tiles_x) * 8, ox0 = (trem % tiles_x) * 16;
    const int iy0 = oy0 - pad, ix0 = ox0 - pad;
    const int chunks = (C + WN_CC - 1) / WN_CC;
    const float* xn = x + n * xi.ns;
    // this wave's weight stream: [chunk][step] x STEP floats, consumed strictly in order
    const float* wsrc = rec + ((size_t)(grp * 4 + w) * chunks) * (8 * STEP);
    float* ring = smem + 2 * WN_BUF + w * (WN_SLOTS * STEP);
    // B^T d B, row w: which two window rows, and the sign of the second
    const int ra = w == 0 ? 0 : (w == 2 ? 2 : 1), rb = w == 0 ? 2 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
    const float sb = w == 1 ? 1.f : -1.f;
    const int lbase = (2 * tr) * WN_PS + 2 * tc;

    // staging: the 160 rows (16 channels x 10) of a chunk's window are cut into 480 segments of 6 floats; thread j owns segments
    // j and j + 256.  A segment is ALWAYS three 8-byte loads (the weight ring counts on it): a pair that straddles or leaves the
    // image is read from the nearest in-image pair position and its elements are picked / zeroed afterwards.
    const int seg_a = threadIdx.x, seg_b = threadIdx.x + 256;
    const bool has_b = seg_b < WN_CC * WN_PR * 3;
    float pre[2][6];
    // where a segment's three pairs come from / which of their elements are real (recomputed at commit time: nothing but the
    // raw pairs stays in registers while the loads are in flight, and nothing touches them before the commit)
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
            const bool same = a >= 0 && a <= Wi - 2;          // the pair was read where it lies
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
    auto commit = [&](int chunk) {          // into buffer chunk & 1
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

    // ring bookkeeping (all wave-uniform): slot to fill next, slot to read next, stream position of the next fill
    int wslot = 0, rslot = 0;
    const float* wnext = wsrc;
    // when the slot count divides the 8 steps of a chunk, a step's slot is a compile-time constant (`fixed`)
    constexpr bool WN_FIXED = 8 % WN_SLOTS == 0;
    auto issue_step = [&](int fixed) {
        const int slot = WN_FIXED ? fixed % WN_SLOTS : wslot;
        wn_dma16xn<KB>(wnext, ring + slot * STEP, lane);
        wnext += STEP;
        if (!WN_FIXED) wslot = wslot + 1 == WN_SLOTS ? 0 : wslot + 1;
    };
    // operands of a step, read from LDS ONE STEP AHEAD of the multiplications that use them (two register sets, parity of the
    // step): the LDS latency then hides behind the previous step's eight MFMAs instead of idling the matrix pipe
    float opa[2][4 * KB];
    w2a opb[2][4];
    auto read_ops = [&](float (&wa)[4 * KB], w2a (&pb)[4], const float* win, int s1) {
        const float* rs = ring + (WN_FIXED ? s1 % WN_SLOTS : rslot) * STEP + 4 * lane;
        if (!WN_FIXED) rslot = rslot + 1 == WN_SLOTS ? 0 : rslot + 1;
#pragma unroll
        for (int p = 0; p < KB; ++p) {
            const w4f v4 = *reinterpret_cast<const w4f*>(rs + p * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) wa[4 * p + j] = v4[j];
        }
        const float* pa = win + (2 * s1 + hh) * WN_CH;          // row w of B^T d B of the tile comes from two window rows
        // every offset here is even: 8-byte LDS reads (the 4-byte-aligned type made them ds_read2_b32 pairs, whose two halves of a
        // wave collide on the banks -- a third of the LDS cycles -- and whose 8-bit offsets cost an address add each)
        pb[0] = *reinterpret_cast<const w2a*>(pa + ra * WN_PS), pb[1] = *reinterpret_cast<const w2a*>(pa + ra * WN_PS + 2);
        pb[2] = *reinterpret_cast<const w2a*>(pa + rb * WN_PS), pb[3] = *reinterpret_cast<const w2a*>(pa + rb * WN_PS + 2);
    };
#pragma unroll
    for (int q = 0; q < DIST; ++q) issue_step(q);     // 8 steps at least in the stream, DIST < 8: never past its end
    fetch(0);
    commit(0);          // the compiler drains the VM counter for the fetched registers here: the first DIST steps have landed too
    __syncthreads();
    read_ops(opa[0], opb[0], smem + lbase, 0);
    // one chunk = eight steps; `last` (the final chunk: nothing to fetch, the stream ends) is a compile-time flag of the body
    auto run_chunk = [&](auto last_c, int chunk) {
        constexpr bool last = decltype(last_c)::value;
        const float* pbuf = smem + (chunk & 1) * WN_BUF + lbase;
        const float* pnext = smem + ((chunk + 1) & 1) * WN_BUF + lbase;
        auto step = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // the chunk hand-over sits in front of step 7: the next window (fetched at step 0) goes to the other buffer -- last read
            // one chunk ago, every wave has passed the barrier behind that -- and one barrier both publishes it and retires this
            // chunk's buffer, whose last operands (step 7's) were read during step 6
            if (s == 7 && !last) {
                commit(chunk + 1);
                __syncthreads();
            }
            // [A] records of step s + DIST into the slot whose operands were read two steps ago; never past the end of the stream
            if (!last || s + DIST < 8) issue_step(s + DIST);
            // [B] the next chunk's window: seven steps of cover until the commit
            if (s == 0 && !last) fetch(chunk + 1);
            if (!last || s < 7) {
                // [C] the records of step s + 1 were issued DIST - 1 steps ago; behind them in the counter: the steps issued since
                // (fewer at the end of the stream) and, for s < DIST, this chunk's window fetch
                if (!last) wn_wait_vm<(DIST - 1) * KB + (s < DIST ? WN_FETCH_OPS : 0)>();
                else wn_wait_vm<(6 - s < DIST - 1 ? 6 - s : DIST - 1) * KB>();
                // [D] next step's operands
                read_ops(opa[(s + 1) & 1], opb[(s + 1) & 1], s == 7 ? pnext : pbuf, (s + 1) & 7);
            }
            // [E] this step's multiplications
            const w2a a0 = opb[s & 1][0], a1 = opb[s & 1][1], b0 = opb[s & 1][2], b1 = opb[s & 1][3];
            // sb = +-1: the fused form is exact, i.e. the same value as a multiply and an add, in one instruction
            const float e0 = fmaf(sb, b0[0], a0[0]), e1 = fmaf(sb, b0[1], a0[1]), e2 = fmaf(sb, b1[0], a1[0]), e3 = fmaf(sb, b1[1], a1[1]);
            const float v[4] = {e0 - e2, e1 + e2, e2 - e1, e1 - e3};
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) acc[nu][kb] = WN_MFMA(opa[s & 1][nu * KB + kb], v[nu], acc[nu][kb]);
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
    };
    for (int chunk = 0; chunk + 1 < chunks; ++chunk) run_chunk(std::false_type{}, chunk);
    run_chunk(std::true_type{}, chunks - 1);
    __syncthreads();      // every wave is done with the windows and its ring: the same memory now carries the output transform
    // ---- output transform Y = A^T M A: columns (nu) inside the wave, rows (w) across the waves through LDS (all KB at once)
    const int kbase = grp * (32 * KB);
    float* yn = y + n * yo.ns;
    // exchange image: [kb][writer wave m][output column j][r / 4][lane][r % 4] -- a lane's four consecutive accumulator registers
    // travel as one 16-byte LDS access in both directions (48 LDS instructions per wave instead of 192 four-byte ones)
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
                if (k < K && row < Ho) {
                    float* yp = yn + k * yo.cs + row * yo.rs + col;
                    if (col + 1 < Wo) *reinterpret_cast<w2f*>(yp) = w2f{o[0][q], o[1][q]};
                    else if (col < Wo) yp[0] = o[0][q];
                }
            }
        }
    }
}

/* y[n, k, oy, ox] = sum_c sum_{a,b<3} x[n, c, oy + a - pad, ox + b - pad] W[k][c][a][b]  (zeros outside x), output Ho x Wo.
 * Forward of the network's padding-0 convolutions: pad = 0, Ho = Hi - 2.  Backward-data: x = d y, pad = 2, Ho = Hi + 2,
 * records packed with backward != 0.  x, y: element strides (image, channel, row), unit column stride. */
template <int KB>
static void wn_launch(const float* records, const float* x, WnTensor xi, int Hi, int Wi, int pad, float* y, WnTensor yo, int Ho, int Wo,
                      int n_img, int c_in, int c_out, int groups, hipStream_t st) {
    const int n_tiles = ((Wo + 15) / 16) * ((Ho + 7) / 8) * n_img;
    dim3 grid((unsigned)(8 * ((n_tiles + 7) / 8) * groups));
    // two windows + four rings; the output exchange (KB * 8192 floats) re-uses the same memory
    constexpr size_t staging = sizeof(float) * (2 * WN_CC * 10 * WN_PS + 4 * wn_slots(KB) * 4 * KB * 64);
    constexpr size_t smem = staging > sizeof(float) * KB * 8192 ? staging : sizeof(float) * KB * 8192;
    static bool once_on[NF_MAX_DEVICES] = {};   // more than the default 64 KB of dynamic LDS needs the attribute once per kernel and device
    bool& once = once_on[nf_current_device()];
    if (!once) {
        (void)hipFuncSetAttribute((const void*)k_wino3x3<KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        once = true;
    }
    hipLaunchKernelGGL((k_wino3x3<KB>), grid, dim3(256), smem, st, records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, c_in, c_out, groups, n_img);
}

extern "C" int nf_conv3x3_wino(const float* records, int k_per_group, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi,
                               int Wi, int pad, float* y, int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img,
                               int c_in, int c_out, int tile_blocks, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && c_in >= 1 && c_out >= 1 && Hi >= 1 && Wi >= 2 && Ho >= 1 && Wo >= 1 && (k_per_group == 64 || k_per_group == 32),
               "nf_conv3x3_wino: bad arguments (k_per_group %d)", k_per_group);
    const int groups = (c_out + k_per_group - 1) / k_per_group;
    const WnTensor xi = {xs_n, xs_c, xs_h}, yo = {ys_n, ys_c, ys_h};
    hipStream_t st = (hipStream_t)stream;
    (void)tile_blocks;      // one block of 32 tiles per wave (two blocks at one workgroup per CU measured 30 % slower)
    if (k_per_group == 32) wn_launch<1>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);
    else wn_launch<2>(records, x, xi, Hi, Wi, pad, y, yo, Ho, Wo, n_img, c_in, c_out, groups, st);
    NF_LAUNCH_CHECK("nf_conv3x3_wino");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The border ring of a backward-data convolution.  d(padded input) = full correlation of d y (H x W) with the rotated weights:
// (H + 2) x (W + 2) values -- for a 48 x 63 plane that is 7 x 5 output blocks of 8 x 16 where the forward pass of the same layer
// has 6 x 4: 46 % more matrix work for a ring one pixel wide (measured: 108 us against 69 us, 11 launches per step).  The ring
// only sees ONE row (column) of d y -- rows Y + a - 2 with Y = 0 leave a = 2 -- so it is four 1-D convolutions of 3 taps:
//   top    g[k][0][X]     = sum_c sum_b dy[c][0][X + b - 2]     W'[k][c][2][b]      W'[k][c][a][b] = weight[c][k][2 - a][2 - b]
//   bottom g[k][H + 1][X] = sum_c sum_b dy[c][H - 1][X + b - 2] W'[k][c][0][b]
//   left   g[k][Y][0]     = sum_c sum_a dy[c][Y + a - 2][0]     W'[k][c][a][2]
//   right  g[k][Y][W + 1] = sum_c sum_a dy[c][Y + a - 2][W - 1] W'[k][c][a][0]
// 0.14 GFLOP for layer3: plain vector FMAs (k_wino_ring below).  The host computes the rest -- rows / columns 1 .. -- with nf_conv3x3_wino at pad 1 on the shifted output
// (nerfool_amd/ops.conv3x3_wino_bwd_split), which also takes the bottom row / right column along when its blocks have room.
// ---------------------------------------------------------------------------------------------------------------------
#define WR_KB 64            // output channels per wave (lane = output channel); the packed weights are padded to a multiple of it
#define WR_WAVES 8          // input-channel split of a workgroup

extern "C" int64_t nf_wino_ring_pack_floats(int c_out, int c_in) { return (int64_t)4 * c_out * 3 * ((c_in + WR_KB - 1) / WR_KB * WR_KB); }

/* HOST: weight [c_out][c_in][3][3] -> ring weights [kind 4][c = c_out][tap 3][k = c_in rounded up to 64] of the backward-data pass */
extern "C" int nf_wino_ring_pack(const float* weight, int c_out, int c_in, float* out) {
    const int Kp = (c_in + WR_KB - 1) / WR_KB * WR_KB;
    for (int kind = 0; kind < 4; ++kind)
        for (int c = 0; c < c_out; ++c)
            for (int t = 0; t < 3; ++t)
                for (int k = 0; k < Kp; ++k) {
                    // W'[k][c][a][b] = weight[c][k][2 - a][2 - b];  top: a = 2, b = t;  bottom: a = 0, b = t;  left: a = t, b = 2;  right: a = t, b = 0
                    const int a = kind == 0 ? 2 : (kind == 1 ? 0 : t), b = kind == 2 ? 2 : (kind == 3 ? 0 : t);
                    out[(((size_t)kind * c_out + c) * 3 + t) * Kp + k] = k < c_in ? weight[(((size_t)c * c_in + k) * 3 + (2 - a)) * 3 + (2 - b)] : 0.f;
                }
    return 0;
}

#define WR_POS 8            // ring positions per workgroup
#define WR_NI 1             // images per workgroup (more would share the weight loads; one keeps every CU busy at 4 images: measured 16 us against 19 at two, 30 at four)
#define WR_XS 12            // floats per (image, channel) line piece in LDS: WR_POS + 2 taps, padded to three 16-byte reads
// lane = output channel (weights [kind][c][tap][k]: one coalesced 256-byte load per channel and tap, eight channels ahead in
// registers); the d y values of the workgroup's WR_POS positions -- the same for every lane -- are fetched ONCE by the whole
// workgroup (one load per value, whatever the stride of the line: the column lines have one cache line per element) into LDS and
// read back as broadcasts; the eight waves split the input channels and add up through LDS in a fixed order.
__global__ void __launch_bounds__(64 * WR_WAVES) k_wino_ring(const float* __restrict__ wr, const float* __restrict__ dy, WnTensor di, int H, int W,
                                                             float* __restrict__ g, WnTensor go, int C, int K, int kinds, int col_rows,
                                                             int n_img) {
    HIP_DYNAMIC_SHARED(float, xs)           // [WR_NI][C][WR_XS], later the partial sums [wave][WR_NI * WR_POS][64]
    const int lane = threadIdx.x & 63, wave = wn_uniform(threadIdx.x >> 6);
    const int Kp = (K + WR_KB - 1) / WR_KB * WR_KB, k = blockIdx.y * 64 + lane, n0 = blockIdx.z * WR_NI;
    // blockIdx.x -> (segment kind, WR_POS positions of it): the kinds asked for, rows first (length W + 2), then columns (rows 1 ..
    // col_rows: H, or H + 1 when the bottom row is not a ring segment of its own and the corners belong to the columns)
    int chunk = blockIdx.x, kind = -1;
    const int row_chunks = (W + 2 + WR_POS - 1) / WR_POS, col_chunks = (col_rows + WR_POS - 1) / WR_POS;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nq = q < 2 ? row_chunks : col_chunks;
        if (kind < 0 && (kinds >> q & 1)) {
            if (chunk < nq) kind = q;
            else chunk -= nq;
        }
    }
    const bool row_kind = kind < 2;
    const int p0 = chunk * WR_POS, len = row_kind ? W + 2 : col_rows;
    // the taps of position p lie at p - 2 .. p of the source line (rows: p = X) or p - 1 .. p + 1 (columns: p = Y - 1)
    const int64_t line = row_kind ? (int64_t)(kind == 0 ? 0 : H - 1) * di.rs : (kind == 2 ? 0 : W - 1);
    const int64_t step = row_kind ? 1 : di.rs;
    const int q_first = p0 + (row_kind ? -2 : -1), n_src = row_kind ? W : H;
    // (unconditional loads from clamped addresses, eight in flight per thread: a load behind a branch is waited for on the spot)
    for (int i0 = threadIdx.x; i0 < WR_NI * C * WR_XS; i0 += 8 * blockDim.x) {
        float v[8];
        bool ok[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * blockDim.x, ii = i < WR_NI * C * WR_XS ? i : 0;
            const int e = ii % WR_XS, ic = ii / WR_XS, c = ic % C, im = ic / C, q = q_first + e;
            ok[u] = e < WR_POS + 2 && q >= 0 && q < n_src && n0 + im < n_img;
            const int qc = q < 0 ? 0 : (q >= n_src ? n_src - 1 : q), nc = n0 + im < n_img ? n0 + im : n_img - 1;
            v[u] = dy[nc * di.ns + c * di.cs + line + qc * step];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < WR_NI * C * WR_XS) xs[i] = ok[u] ? v[u] : 0.f;
        }
    }
    __syncthreads();
    float acc[WR_NI][WR_POS];
#pragma unroll
    for (int im = 0; im < WR_NI; ++im)
#pragma unroll
        for (int j = 0; j < WR_POS; ++j) acc[im][j] = 0.f;
    const float* wk = wr + (size_t)kind * C * 3 * Kp + (k < Kp ? k : 0);
    constexpr int AHEAD = 8;         // (all 32 channels of a wave at once measured slower: 23 us against 16)
    for (int c0 = wave; c0 < C; c0 += AHEAD * WR_WAVES) {
        float wv[AHEAD][3];
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            const int c = c0 + u * WR_WAVES;
#pragma unroll
            for (int t = 0; t < 3; ++t) wv[u][t] = wk[((size_t)(c < C ? c : 0) * 3 + t) * Kp];
        }
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            const int c = c0 + u * WR_WAVES;
            if (c < C) {
#pragma unroll
                for (int im = 0; im < WR_NI; ++im) {
                    const float* xp = xs + (im * C + c) * WR_XS;
                    const w4f x0 = *reinterpret_cast<const w4f*>(xp), x1 = *reinterpret_cast<const w4f*>(xp + 4), x2 = *reinterpret_cast<const w4f*>(xp + 8);
                    const float x[12] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x2[0], x2[1], x2[2], x2[3]};
#pragma unroll
                    for (int j = 0; j < WR_POS; ++j)
#pragma unroll
                        for (int t = 0; t < 3; ++t) acc[im][j] = fmaf(wv[u][t], x[j + t], acc[im][j]);
                }
            }
        }
    }
    __syncthreads();        // everybody is done with the line pieces: the same memory takes the partial sums
    float* red = xs;
#pragma unroll
    for (int im = 0; im < WR_NI; ++im)
#pragma unroll
        for (int j = 0; j < WR_POS; ++j) red[(wave * (WR_NI * WR_POS) + im * WR_POS + j) * 64 + lane] = acc[im][j];
    __syncthreads();
    // wave w adds up four of the 32 (image, position) values of every lane (fixed order: the result does not depend on scheduling)
#pragma unroll
    for (int aa = 0; aa < WR_NI * WR_POS / WR_WAVES; ++aa) {
        const int a = wave * (WR_NI * WR_POS / WR_WAVES) + aa, im = a / WR_POS, j = a - im * WR_POS, p = p0 + j;
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < WR_WAVES; ++q) sum += red[(q * (WR_NI * WR_POS) + a) * 64 + lane];
        if (p < len && k < K && n0 + im < n_img) {
            const int Y = row_kind ? (kind == 0 ? 0 : H + 1) : p + 1, X = row_kind ? p : (kind == 2 ? 0 : W + 1);
            g[(n0 + im) * go.ns + k * go.cs + Y * go.rs + X] = sum;
        }
    }
}

/* ring of g = d(padded input) [N, K, H + 2, W + 2] of a 3x3 stride-1 convolution from dy [N, C, H, W]; `kinds`: bit 0 top row,
 * 1 bottom row, 2 left column, 3 right column; the columns cover rows 1 .. H, and row H + 1 as well when the bottom row is not
 * asked for (the corners must belong to somebody).  records: nf_wino_ring_pack(weight [C][K][3][3]). */
extern "C" int nf_conv3x3_bwd_ring(const float* ring_records, const float* dy, int64_t ds_n, int64_t ds_c, int64_t ds_h, int H, int W, float* g,
                                   int64_t gs_n, int64_t gs_c, int64_t gs_h, int n_img, int c_dy, int c_dx, int kinds, nf_stream_t stream) {
    NF_REQUIRE(n_img >= 1 && c_dy >= 1 && c_dx >= 1 && H >= 1 && W >= 1 && kinds > 0 && kinds < 16, "nf_conv3x3_bwd_ring: bad arguments (kinds %d)", kinds);
    const int col_rows = (kinds & 2) ? H : H + 1;
    const int row_chunks = (W + 2 + WR_POS - 1) / WR_POS, col_chunks = (col_rows + WR_POS - 1) / WR_POS;
    int chunks = 0;
    for (int q = 0; q < 4; ++q)
        if (kinds >> q & 1) chunks += q < 2 ? row_chunks : col_chunks;
    const WnTensor di = {ds_n, ds_c, ds_h}, go = {gs_n, gs_c, gs_h};
    // line pieces of WR_NI images, or the eight waves' partial sums, whichever is larger
    size_t smem = sizeof(float) * (size_t)WR_NI * c_dy * WR_XS;
    if (smem < sizeof(float) * WR_WAVES * WR_NI * WR_POS * 64) smem = sizeof(float) * WR_WAVES * WR_NI * WR_POS * 64;
    NF_REQUIRE(smem <= 64 * 1024, "nf_conv3x3_bwd_ring: at most %d gradient channels (got %d)", 64 * 1024 / (4 * WR_NI * WR_XS), c_dy);
    hipLaunchKernelGGL(k_wino_ring, dim3((unsigned)chunks, (unsigned)((c_dx + 63) / 64), (unsigned)((n_img + WR_NI - 1) / WR_NI)), dim3(64 * WR_WAVES),
                       smem, (hipStream_t)stream, ring_records, dy, di, H, W, g, go, c_dy, c_dx, kinds, col_rows, n_img);
    NF_LAUNCH_CHECK("nf_conv3x3_bwd_ring");
    return 0;
}
