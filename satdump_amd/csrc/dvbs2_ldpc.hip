// dvbs2_ldpc.hip -- DVB-S2 LDPC soft decoding on gfx950 (BASELINE.json configs[4]): the reference's layered
// offset-min-sum decoder on int8 soft bits (plugins/dvb_support/codings/dvb-s2/ldpc/layered_decoder.hh:29-79,152-169 with
// OffsetMinSumAlgorithm<SIMD<int8_t, W>, NormalUpdate, 2>, algorithms.hh:207-279, as BBFrameLDPC instantiates it,
// bbframe_ldpc.h:19-32), bit for bit: same check order semantics, same saturating int8 arithmetic, same early exit -- including the
// coupling of `batch` consecutive frames the reference's SIMD build decodes in one call (W = 16 with -msse4.1: a frame that has
// converged keeps being updated until all sixteen of its call have, or the trials are used up).
//
// Mapping. One WORKGROUP per frame, the frame's 64 800 (16 200) LLRs in LDS for the whole pass -- every Tanner-graph gather and
// scatter of the decoder is an LDS byte access; one THREAD per check node of a layer (M = 360 checks: the DVB-S2 matrices are
// quasi-cyclic with period 360, the reference walks its checks layer-major), the check-to-bit messages of the frame (`bnl`, 195 -
// 285 k per frame: the state that does NOT fit the 160 KB of LDS) packed four per dword in HBM as [layer][slot/4][check]: a wave's
// loads and stores of them are contiguous. HBM bytes per frame and iteration = 2 x messages + 2 x LLRs: the kernel's roofline.
// The reference runs the checks of a layer one after the other; they are independent unless two of them share a data bit (a bit
// group with two parity addresses in the same residue class mod q). Those pairs are found when the graph is built and such a layer
// is run in `phases`: a check waits for every lower-numbered check it shares a bit with -- the sequential result, exactly.
// One launch = one trial of every frame ([update if the frame's batch was still bad] + parity check): the batch coupling needs no
// inter-workgroup synchronisation, the host reads back one counter per trial to stop early.
#include "common.h"
#include "../../include/sdhip.h"
#include <algorithm>
#include <memory>
#include <vector>

namespace sdhip
{
#include "dvbs2_tables.inc"

    struct LdpcDev
    {
        int M, N, K, R, q, CNL, DQ;
        const unsigned *ndp;        // [q][2 * DQ][M] the nodes of links 2h (low half) and 2h + 1 (high half) of check (layer, check): its data bits in bit
                                    // order, then its own parity bit, then its predecessor's; 0xFFFF = no such link
        const unsigned char *cnc;   // [q] data bits per check of the layer
        const unsigned char *phase; // [q][M] phase of a check inside its layer (0 unless it shares a bit with a lower-numbered one)
        const unsigned char *nph;   // [q] phases of the layer
        // "narrow" layers: many phases of a few checks each (chains of checks sharing bits: 90 phases of 4 checks, 52 of 7, ...). A check per thread
        // makes every one of those phases cost a whole check's instruction stream (~400 dependent instructions on one wave); they run a LINK per
        // lane instead: G lanes per check (G = the power of two >= its degree), minima / sign parity by row reductions.
        int MB, G;                   // checks a layer's message block holds (>= M: a narrow layer's schedule is padded to width x phases); lanes per check
        const unsigned char *narrow; // [q] 0 = a check per thread; else W = checks per phase of the padded schedule
        const unsigned short *snode; // [q][MB][4 * DQ] node of link d of the check at schedule position k = phase * W + slot, 0xFFFF = no such link
        int packed;                  // wide layers: LdpcCheck::update_pk (two links per register half) instead of update (SDHIP_LDPC_PACKED=0: A/B)
    };

    __device__ __forceinline__ int q8(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }

    // one check node: OffsetMinSumAlgorithm::finalp + update + the add back, layered_decoder.hh:53-77. Split in three so that a layer
    // whose checks run in dependent phases (shared data bits) pays the HBM latency of the message words and the L2 latency of the bit
    // addresses ONCE, before its first phase, not once per phase: load (every thread of the layer at once) / update (the phase's
    // threads: LDS and registers only) / store.
    template <int DQ>
    struct LdpcCheck
    {
        unsigned bw[DQ];         // check-to-bit messages, four per dword
        unsigned nd[2 * DQ];     // node indices of the links, two per dword
        int deg;
        __device__ __forceinline__ void load(const LdpcDev &g, const unsigned *bnl_f, int i, int j)
        {
            const int cnt = g.cnc[i];
            deg = cnt + 2 - ((i | j) == 0 ? 1 : 0);
#pragma unroll
            for (int w = 0; w < DQ; w++)
                bw[w] = 4 * w < deg ? bnl_f[((size_t)i * DQ + w) * g.MB + j] : 0u;
#pragma unroll
            for (int h = 0; h < 2 * DQ; h++)
                nd[h] = g.ndp[((size_t)i * 2 * DQ + h) * g.M + j];
        }
        __device__ __forceinline__ int node(int d) const { return (int)((nd[d >> 1] >> (16 * (d & 1))) & 0xFFFFu); }
        __device__ __forceinline__ void update(signed char *llr)
        {
            int min0 = 255, min1 = 255;
            unsigned signs = 0;
            unsigned iw[DQ]; // the links' inputs, four per dword: the second loop takes them from here instead of reading and subtracting again
#pragma unroll
            for (int w = 0; w < DQ; w++)
            {
                iw[w] = 0;
#pragma unroll
                for (int b = 0; b < 4; b++)
                {
                    const int d = 4 * w + b;
                    if (d < deg)
                    {
                        const int inp = q8((int)llr[node(d)] - (int)(signed char)(bw[w] >> (8 * b)));
                        iw[w] |= ((unsigned)inp & 0xFFu) << (8 * b);
                        int mag = inp < -127 ? 127 : (inp < 0 ? -inp : inp); // vqabs
                        mag = mag > 0 ? mag - 1 : 0;                          // unsigned saturating - beta, beta = nearbyint(0.5 * 2) = 1
                        // mins[1] = min(mins[1], max(mins[0], mag)); mins[0] = min(mins[0], mag) (the first two: min / max of the pair)
                        const int hi = mag > min0 ? mag : min0;
                        min1 = hi < min1 ? hi : min1;
                        min0 = mag < min0 ? mag : min0;
                        signs ^= (unsigned)inp;
                    }
                }
            }
#pragma unroll
            for (int w = 0; w < DQ; w++)
            {
                unsigned nw = 0;
#pragma unroll
                for (int b = 0; b < 4; b++)
                {
                    const int d = 4 * w + b;
                    if (d < deg)
                    {
                        const int n = node(d);
                        const int inp = (int)(signed char)(iw[w] >> (8 * b));
                        int mag = inp < -127 ? 127 : (inp < 0 ? -inp : inp);
                        mag = mag > 0 ? mag - 1 : 0;
                        const int other = mag == min0 ? min1 : min0;
                        const bool neg = ((signs ^ (unsigned)inp) & 0x80u) != 0; // sign(other, (signs ^ link) | 127)
                        int out = neg ? -other : other;
                        out = out < -32 ? -32 : (out > 31 ? 31 : out); // update(): clamp to [-32, 31]
                        llr[n] = (signed char)q8(inp + out);
                        nw |= ((unsigned)out & 0xFFu) << (8 * b);
                    }
                }
                bw[w] = nw;
            }
        }
        // The same check node on PAIRS of links (round 6): the update above spends ~35 instructions a link on byte extraction, clamps and
        // selects; here links 2p and 2p + 1 ride in the two 16-bit halves of a register through v_pk_* arithmetic -- the int8 saturations are packed max / min,
        // |x| is max(x, -x), "minus beta, not below zero" a saturating packed subtract, the two smallest magnitudes are tracked per half and merged once (the two
        // smallest of a multiset do not depend on the order they are met in), "the other links' minimum" is min0 + [mag == min0] x (min1 - min0) without a compare,
        // the sign a packed arithmetic shift, and four new messages leave through one v_perm. Same integers at every point (every quantity is an exact small
        // integer: nothing rounds), the bits written in the same link order. ~19 instructions a link.
        __device__ __forceinline__ void update_pk(signed char *llr)
        {
            typedef short s2 __attribute__((ext_vector_type(2)));
            typedef unsigned short u2 __attribute__((ext_vector_type(2)));
            auto S = [](unsigned x) { return __builtin_bit_cast(s2, x); };
            auto U = [](unsigned x) { return __builtin_bit_cast(u2, x); };
            auto RS = [](s2 x) { return __builtin_bit_cast(unsigned, x); };
            auto RU = [](u2 x) { return __builtin_bit_cast(unsigned, x); };
            auto sat8 = [&](s2 x) { return __builtin_elementwise_min(__builtin_elementwise_max(x, s2{-128, -128}), s2{127, 127}); };
            constexpr int P = 2 * DQ;
            constexpr bool KEEP = DQ <= 3; // the magnitudes kept for the second loop, or formed again there (four instructions a pair against 2 DQ registers: DQ = 4
                                           // sits at the 168 registers that three waves per SIMD allow)
            auto vqabs_beta = [&](unsigned iv, unsigned vmask) -> unsigned {
                const s2 Iv = S(iv);
                u2 G = U(RS(__builtin_elementwise_min(__builtin_elementwise_max(Iv, s2{0, 0} - Iv), s2{127, 127}))); // vqabs
                G = __builtin_elementwise_sub_sat(G, u2{1, 1});                                                       // unsigned saturating - beta
                return (RU(G) & vmask) | (0x00FF00FFu & ~vmask);
            };
            unsigned inp[P], mag[KEEP ? P : 1];
            unsigned m0 = 0x00FF00FFu, m1 = 0x00FF00FFu, sg = 0u;
#pragma unroll
            for (int p = 0; p < P; p++)
            {
                inp[p] = 0u;
                if constexpr (KEEP)
                    mag[p] = 0x00FF00FFu;
                if (2 * p < deg)
                {
                    const bool v1 = 2 * p + 1 < deg;
                    const unsigned vmask = v1 ? 0xFFFFFFFFu : 0x0000FFFFu;
                    const unsigned n0 = nd[p] & 0xFFFFu, n1 = v1 ? nd[p] >> 16 : n0;
                    const int l0 = llr[n0], l1 = llr[n1];
                    const unsigned L = ((unsigned)l0 & 0xFFFFu) | ((unsigned)l1 << 16);
                    // the pair's messages, sign-extended: the bytes moved to the high byte of each half, then an arithmetic shift
                    const unsigned mb = __builtin_amdgcn_perm(0u, bw[p >> 1], (p & 1) ? 0x030c020cu : 0x010c000cu);
                    const s2 M = S(mb) >> 8;
                    const s2 I = sat8(S(L) - M);
                    const unsigned iv = RS(I) & vmask;
                    inp[p] = iv;
                    const unsigned gv = vqabs_beta(iv, vmask);
                    if constexpr (KEEP)
                        mag[p] = gv;
                    const u2 hi = __builtin_elementwise_max(U(gv), U(m0));
                    m1 = RU(__builtin_elementwise_min(hi, U(m1)));
                    m0 = RU(__builtin_elementwise_min(U(gv), U(m0)));
                    sg ^= iv;
                }
            }
            const unsigned a0 = m0 & 0xFFFFu, b0 = m0 >> 16, a1 = m1 & 0xFFFFu, b1 = m1 >> 16;
            const unsigned min0 = a0 < b0 ? a0 : b0, mx = a0 < b0 ? b0 : a0, mn1 = a1 < b1 ? a1 : b1;
            const unsigned min1 = mx < mn1 ? mx : mn1;
            const unsigned MIN0 = min0 | (min0 << 16), DELTA = (min1 - min0) | ((min1 - min0) << 16);
            const unsigned SGN = (((sg ^ (sg >> 16)) & 0x80u) != 0u) ? 0xFFFFFFFFu : 0u; // the product of all the links' signs
            unsigned olo = 0u;
#pragma unroll
            for (int p = 0; p < P; p++)
            {
                unsigned o = 0u;
                if (2 * p < deg)
                {
                    const bool v1 = 2 * p + 1 < deg;
                    const unsigned vmask = v1 ? 0xFFFFFFFFu : 0x0000FFFFu;
                    unsigned gv;
                    if constexpr (KEEP)
                        gv = mag[p];
                    else
                        gv = vqabs_beta(inp[p], vmask);
                    const u2 X = U(gv) - U(MIN0);                                   // >= 0: min0 is the smallest
                    const u2 E = u2{1, 1} - __builtin_elementwise_min(X, u2{1, 1}); // [mag == min0]
                    const s2 other = S(RU(E * U(DELTA) + U(MIN0)));
                    const s2 NEG = S(RS(S(inp[p]) >> 15) ^ SGN);                    // -1 where the sign of the product of the OTHER links' signs is negative
                    s2 out = (other ^ NEG) - NEG;
                    out = __builtin_elementwise_min(__builtin_elementwise_max(out, s2{-32, -32}), s2{31, 31}); // update(): clamp to [-32, 31]
                    const unsigned nv = RS(sat8(S(inp[p]) + out));
                    llr[nd[p] & 0xFFFFu] = (signed char)(nv & 0xFFu);
                    if (v1)
                        llr[nd[p] >> 16] = (signed char)((nv >> 16) & 0xFFu);
                    o = RS(out) & vmask;
                }
                if (p & 1)
                    bw[p >> 1] = __builtin_amdgcn_perm(o, olo, 0x06040200u); // four new messages: the low bytes of the four halves
                else
                    olo = o;
            }
        }
        __device__ __forceinline__ void store(const LdpcDev &g, unsigned *bnl_f, int i, int j) const
        {
#pragma unroll
            for (int w = 0; w < DQ; w++)
                if (4 * w < deg)
                    bnl_f[((size_t)i * DQ + w) * g.MB + j] = bw[w];
        }
    };
    // parity of one check (LDPCDecoder::bad, layered_decoder.hh:29-47): bad unless every connected LLR is nonzero and an even number negative
    __device__ __forceinline__ bool ldpc_check_bad(const LdpcDev &g, const signed char *llr, int i, int j)
    {
        const int cnt = g.cnc[i];
        int neg = 0;
        bool zero = false;
        auto take = [&](int v) {
            zero |= v == 0;
            neg ^= v < 0 ? 1 : 0;
        };
        const int nh = (cnt + 2 + 1) >> 1;
        for (int h = 0; h < nh; h++)
        {
            const unsigned w = g.ndp[((size_t)i * 2 * g.DQ + h) * g.M + j];
            const unsigned n0 = w & 0xFFFFu, n1 = w >> 16;
            if (n0 != 0xFFFFu)
                take(llr[n0]);
            if (n1 != 0xFFFFu)
                take(llr[n1]);
        }
        return zero || neg;
    }

    // the parity of check j of EVERY layer (what the loop over ldpc_check_bad computes, layer by layer with a way out at the first bad one): the node words of several
    // layers are fetched together -- a layer at a time, each one's loads waited for a memory round trip that nothing hid, q of them per trial (round 6)
    template <int DQ>
    __device__ __forceinline__ bool ldpc_checks_bad(const LdpcDev &g, const signed char *llr, int j)
    {
        constexpr int UN = DQ <= 3 ? 6 : (DQ <= 5 ? 4 : 2);
        bool bad = false;
        for (int i0 = 0; i0 < g.q && !bad; i0 += UN)
        {
            unsigned w[UN][2 * DQ];
#pragma unroll
            for (int u = 0; u < UN; u++)
#pragma unroll
                for (int h = 0; h < 2 * DQ; h++)
                    w[u][h] = i0 + u < g.q ? g.ndp[((size_t)(i0 + u) * 2 * DQ + h) * g.M + j] : 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < UN; u++)
            {
                int neg = 0;
                bool zero = false;
#pragma unroll
                for (int h = 0; h < 2 * DQ; h++)
                {
                    const unsigned n0 = w[u][h] & 0xFFFFu, n1 = w[u][h] >> 16;
                    if (n0 != 0xFFFFu)
                    {
                        const int v = llr[n0];
                        zero |= v == 0;
                        neg ^= v < 0 ? 1 : 0;
                    }
                    if (n1 != 0xFFFFu)
                    {
                        const int v = llr[n1];
                        zero |= v == 0;
                        neg ^= v < 0 ? 1 : 0;
                    }
                }
                bad |= zero || neg;
            }
        }
        return bad;
    }

    constexpr int LDPC_THREADS = 384; // 360 checks of a layer, six waves

    // allreduce over the G lanes of a group (G = 4, 8, 16: DPP inside a 16-lane row; 32: one cross-row exchange on top)
    template <int G, class Op>
    __device__ __forceinline__ unsigned group_allreduce(unsigned v, Op op)
    {
        v = op(v, dpp_mov<0xB1>(v)); // quad_perm [1,0,3,2]
        v = op(v, dpp_mov<0x4E>(v)); // quad_perm [2,3,0,1]
        if constexpr (G >= 8)
            v = op(v, dpp_mov<0x141>(v)); // row_half_mirror
        if constexpr (G >= 16)
            v = op(v, dpp_mov<0x140>(v)); // row_mirror
        if constexpr (G >= 32)
            v = op(v, (unsigned)__shfl_xor((int)v, 16));
        return v;
    }
    // One phase of a narrow layer for this lane's link: the check node update of LdpcCheck::update with the check's links spread over the
    // group's lanes. n = node (0xFFFF: no link), m = the link's check-to-bit message; returns the new message.
    template <int G>
    __device__ __forceinline__ int ldpc_link_update(signed char *llr, unsigned n, int m)
    {
        const bool has = n != 0xFFFFu;
        const int inp = has ? q8((int)llr[n] - m) : 0;
        int mag = inp < -127 ? 127 : (inp < 0 ? -inp : inp);
        mag = mag > 0 ? mag - 1 : 0;
        const unsigned key = has ? (unsigned)mag : 255u;
        auto mn = [](unsigned a, unsigned b) { return a < b ? a : b; };
        const unsigned min0 = group_allreduce<G>(key, mn);
        // the second of the sorted magnitudes: min0 again when two links hold it
        const unsigned long long lanes = __ballot(has && key == min0), negs = __ballot(has && inp < 0);
        const int base = (int)(threadIdx.x & 63u) & ~(G - 1);
        const unsigned gm = G >= 32 ? 0xFFFFFFFFu : ((1u << (G & 31)) - 1u);
        const unsigned at_min = (unsigned)(lanes >> base) & gm, neg_bits = (unsigned)(negs >> base) & gm;
        const unsigned next = group_allreduce<G>(key == min0 ? 255u : key, mn); // (unconditional: the lanes of a wave stay convergent)
        const unsigned min1 = __popc(at_min) >= 2 ? min0 : next;
        const int other = (unsigned)mag == min0 ? (int)min1 : (int)min0;
        const bool neg = ((__popc(neg_bits) & 1) != 0) != (inp < 0); // sign of the product of the OTHER links' signs
        int out = neg ? -other : other;
        out = out < -32 ? -32 : (out > 31 ? 31 : out);
        if (has)
            llr[n] = (signed char)q8(inp + out);
        __builtin_amdgcn_wave_barrier(); // (the next phase's reads come after every lane's write: lockstep on the hardware, a meeting point for the host twin)
        return out;
    }
    // a narrow layer: lane (group, link) walks the phases; node indices and messages of the phases ahead ride in a register queue (their
    // addresses depend on the phase alone), so that no global latency sits between two phases
    template <int G, int DQ>
    __device__ __forceinline__ void ldpc_narrow_layer(const LdpcDev &g, signed char *llr, unsigned *bnl_f, int i, int tid)
    {
        constexpr int D = 8, L = 4 * DQ;
        const int W = g.narrow[i], nph = g.nph[i];
        const int gid = tid / G, l = tid % G;
        const bool mine = gid < W && l < L;
        const bool multi = W * G > 64; // the phase's lanes span more than one wave: the phases need the workgroup's barrier
        const unsigned short *sn = g.snode + ((size_t)i * g.MB + gid) * L + l;
        signed char *mb = reinterpret_cast<signed char *>(bnl_f) + ((size_t)i * g.MB + gid) * L + l; // messages of a narrow layer: [position][link] bytes
        const size_t step = (size_t)W * L;
        unsigned nq[D];
        int mq[D];
#pragma unroll
        for (int u = 0; u < D; u++)
        {
            nq[u] = 0xFFFFu;
            mq[u] = 0;
            if (mine && u < nph)
            {
                nq[u] = sn[(size_t)u * step];
                mq[u] = mb[(size_t)u * step];
            }
        }
        for (int base = 0; base < nph; base += D)
        {
#pragma unroll
            for (int u = 0; u < D; u++)
            {
                const int ph = base + u;
                if (ph < nph)
                {
                    const unsigned n = nq[u];
                    const int m = mq[u];
                    nq[u] = 0xFFFFu;
                    if (mine && ph + D < nph)
                    {
                        nq[u] = sn[(size_t)(ph + D) * step];
                        mq[u] = mb[(size_t)(ph + D) * step];
                    }
                    if ((tid & ~63) < W * G) // (waves without a link skip the arithmetic)
                    {
                        const int out = ldpc_link_update<G>(llr, n, m);
                        if (n != 0xFFFFu)
                            mb[(size_t)ph * step] = (signed char)out;
                    }
                    if (multi)
                        __syncthreads();
                }
            }
        }
    }
    // trial t of every frame. bad_prev / bad_cur: per frame flags of trial t-1 / t; updates[f]: update passes run so far.
    // probe (SDHIP_LDPC_PROBE=1, measurements only): shader-clock ticks workgroup 0's first lane spends in {frame in, waiting for a layer's loads, the
    // phases, the stores, the parity check, frame out}, summed over the launches
#ifdef SDHIP_HOST_TWIN
    __device__ __forceinline__ long long ldpc_clock() { return 0; }
    __device__ __forceinline__ void ldpc_wait_loads() {}
#else
    __device__ __forceinline__ long long ldpc_clock() { return (long long)__builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void ldpc_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
    // three waves per SIMD (two workgroups of six waves on a CU, what the LDS allows) wherever the check's registers fit 168: the allocator is told so, DQ = 4 sits at the edge
    template <int DQ>
    struct LdpcWaves
    {
        static constexpr int v = DQ <= 4 ? 3 : 2;
    };
    template <int DQ>
    __global__ __launch_bounds__(LDPC_THREADS) __attribute__((amdgpu_waves_per_eu(LdpcWaves<DQ>::v, 4))) void k_ldpc_trial(LdpcDev g, signed char *frames, unsigned *bnl, int nframes, int batch, int t, const int *bad_prev,
                                                                 int *bad_cur, int *updates, int *any_bad, long long *probe)
    {
        __shared__ signed char llr[64800];
        // the per-layer bytes (data bits per check, phases, narrow width) in LDS: read in front of every layer, they each cost an L2 round trip
        // on the critical path when they come from memory
        __shared__ unsigned char s_layer[3][256];
        const int f = (int)blockIdx.x, tid = (int)threadIdx.x;
        if (tid < g.q)
        {
            s_layer[0][tid] = g.cnc[tid];
            s_layer[1][tid] = g.nph[tid];
            s_layer[2][tid] = g.narrow[tid];
        }
        g.cnc = s_layer[0];
        g.nph = s_layer[1];
        g.narrow = s_layer[2];
        const bool pr = probe != nullptr && f == 0 && tid == 0;
        long long pc[6] = {0, 0, 0, 0, 0, 0}, pt = probe ? ldpc_clock() : 0;
        auto lap = [&](int k) {
            if (probe)
            {
                const long long now = ldpc_clock();
                pc[k] += now - pt;
                pt = now;
            }
        };
        bool active = false;
        if (t > 0)
        { // while (bad(...) && --trials >= 0) update(...): the whole batch goes on while ANY of its frames is bad
            const int b0 = f / batch * batch;
            for (int k = 0; k < batch; k++)
                active |= bad_prev[b0 + k] != 0;
            if (!active)
            {
                if (tid == 0)
                    bad_cur[f] = 0;
                return;
            }
        }
        signed char *fr = frames + (size_t)f * g.N;
        // data bits as they are; parity bit q*j + i of the frame is node M*i + j of the decoder (layered_decoder.hh:157-159)
        for (int w = tid; w < g.K / 4; w += LDPC_THREADS)
            reinterpret_cast<unsigned *>(llr)[w] = reinterpret_cast<const unsigned *>(fr)[w];
        for (int p = tid; p < g.R; p += LDPC_THREADS)
        {
            const int j = p / g.q, i = p - j * g.q;
            llr[g.K + g.M * i + j] = fr[g.K + p];
        }
        unsigned *bnl_f = bnl + (size_t)f * g.q * DQ * g.MB;
        __syncthreads();
        lap(0);
        if (active)
        {
            // a wide layer's message words, node indices and phases are fetched one layer AHEAD (they do not depend on the layer in between: the
            // messages are the previous pass's), so that their HBM / L2 latency is not paid in front of every layer
            LdpcCheck<DQ> ck, nx;
            int my = -1, my_nx = -1;
            auto fetch = [&](int i, LdpcCheck<DQ> &c, int &m) {
                m = -1;
                if (i < g.q && !g.narrow[i] && tid < g.M)
                {
                    m = (int)g.phase[i * g.M + tid];
                    c.load(g, bnl_f, i, tid);
                }
            };
            auto layer = [&](int i, LdpcCheck<DQ> &c, int m) {
                const int nph = g.nph[i];
                if (g.narrow[i])
                { // many phases of a few checks: a link per lane (see LdpcDev)
                    switch (g.G)
                    {
                    case 4:
                        ldpc_narrow_layer<4, DQ>(g, llr, bnl_f, i, tid);
                        break;
                    case 8:
                        ldpc_narrow_layer<8, DQ>(g, llr, bnl_f, i, tid);
                        break;
                    case 16:
                        ldpc_narrow_layer<16, DQ>(g, llr, bnl_f, i, tid);
                        break;
                    default:
                        ldpc_narrow_layer<32, DQ>(g, llr, bnl_f, i, tid);
                        break;
                    }
                    __syncthreads();
                    lap(2);
                }
                else
                {
                    for (int ph = 0; ph < nph; ph++)
                    {
                        if (m == ph)
                        {
                            if (g.packed)
                                c.update_pk(llr);
                            else
                                c.update(llr);
                        }
                        __syncthreads();
                    }
                    lap(2);
                    if (m >= 0)
                        c.store(g, bnl_f, i, tid);
                    lap(3);
                }
            };
            // (two register sets taking turns: copying one into the other would wait for its loads)
            fetch(0, ck, my);
            for (int i = 0; i < g.q; i += 2)
            {
                fetch(i + 1, nx, my_nx);
                lap(1); // (issue only: what of the loads' latency is not hidden shows up in the phases)
                layer(i, ck, my);
                if (i + 1 < g.q)
                {
                    fetch(i + 2, ck, my);
                    lap(1);
                    layer(i + 1, nx, my_nx);
                }
            }
        }
        bool bad = false;
        if (tid < g.M)
        {
            if (g.packed)
                bad = ldpc_checks_bad<DQ>(g, llr, tid);
            else
                for (int i = 0; i < g.q && !bad; i++)
                    bad = ldpc_check_bad(g, llr, i, tid);
        }
        const int any = __syncthreads_or(bad ? 1 : 0);
        lap(4);
        if (active)
        {
            for (int w = tid; w < g.K / 4; w += LDPC_THREADS)
                reinterpret_cast<unsigned *>(fr)[w] = reinterpret_cast<const unsigned *>(llr)[w];
            for (int p = tid; p < g.R; p += LDPC_THREADS)
            {
                const int j = p / g.q, i = p - j * g.q;
                fr[g.K + p] = llr[g.K + g.M * i + j];
            }
        }
        if (tid == 0)
        {
            bad_cur[f] = any;
            if (active)
                updates[f] += 1;
            if (any)
                atomicOr(any_bad, 1);
        }
        if (pr)
        {
            lap(5);
            for (int k = 0; k < 6; k++)
                probe[k] += pc[k];
            probe[6] += active ? 1 : 0;
            probe[7] += 1;
        }
    }

    // trials_out[b] = BBFrameLDPC::decode's value for batch b: update passes run if the batch converged, -1 otherwise
    __global__ void k_ldpc_result(int nbatches, int batch, const int *bad, const int *updates, int *trials_out)
    {
        const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (b >= nbatches)
            return;
        int any = 0;
        for (int k = 0; k < batch; k++)
            any |= bad[b * batch + k];
        trials_out[b] = any ? -1 : updates[b * batch];
    }

    struct LdpcEngine
    {
        sdhip_ldpc_cfg cfg;
        hipStream_t stream = nullptr;
        const S2Table *tab = nullptr;
        LdpcDev g{};
        DevBuf<unsigned> d_ndp;
        DevBuf<unsigned char> d_cnc, d_phase, d_nph, d_narrow;
        DevBuf<unsigned short> d_snode;
        int narrow_layers = 0;
        DevBuf<unsigned> d_bnl;
        DevBuf<int> d_flags; // bad[2][nf] | updates[nf] | any
        DevBuf<signed char> d_frames;
        DevBuf<int> d_trials;
        int max_phases = 1, conflict_layers = 0;
        sdhip_ldpc_info info{};
        DevBuf<long long> d_probe; // SDHIP_LDPC_PROBE=1 (see k_ldpc_trial)

        explicit LdpcEngine(const sdhip_ldpc_cfg &c) : cfg(c)
        {
            // BBFrameLDPC::BBFrameLDPC, bbframe_ldpc.cpp:26-107: (framesize, rate) -> table; C7_8 has none
            static const int normal[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, -1, 9, 10}, shortf[12] = {11, 12, 13, 14, 15, 16, 17, 18, 19, -1, 20, -1};
            if (c.rate < 0 || c.rate > 11 || (c.framesize != 0 && c.framesize != 1))
                throw HipError("dvbs2 ldpc: unknown frame size / code rate");
            const int ti = c.framesize == 0 ? normal[c.rate] : shortf[c.rate];
            if (ti < 0)
                throw HipError("dvbs2 ldpc: no table for this frame size / code rate");
            if (c.batch < 1 || c.batch > 64)
                throw HipError("dvbs2 ldpc: batch must be 1..64 (the SIMD width of the reference build it mirrors)");
            SD_HIP(hipSetDevice(c.device));
            SD_HIP(hipStreamCreate(&stream));
            tab = &S2_TABLES[ti];
            const int M = tab->M, N = tab->N, K = tab->K, R = N - K, q = R / M, CNL = tab->links_max_cn - 2;
            // LDPCDecoder::init, layered_decoder.hh:118-151: every data bit's parity addresses (ldpc.hh:36-110: row r of a bit group is
            // its table row shifted by r*q mod R), collected per check in bit order
            std::vector<unsigned short> pos0((size_t)R * CNL, 0);
            std::vector<int> cn(R, 0);
            const unsigned short *grp = S2_GRP + 2 * tab->grp_off, *row = S2_POS + tab->pos_off;
            int bit = 0;
            for (int gi = 0; gi < tab->ngroups; gi++)
            {
                const int deg = grp[2 * gi], rows = grp[2 * gi + 1];
                for (int r = 0; r < rows; r++, row += deg)
                    for (int m = 0; m < M; m++, bit++)
                        for (int n = 0; n < deg; n++)
                        {
                            const int i = (row[n] + m * q) % R;
                            if (cn[i] >= CNL)
                                throw HipError("dvbs2 ldpc: table inconsistent (check degree)");
                            pos0[(size_t)CNL * i + cn[i]++] = (unsigned short)bit;
                        }
            }
            if (bit != K)
                throw HipError("dvbs2 ldpc: table inconsistent (bit count)");
            if (q > 256)
                throw HipError("dvbs2 ldpc: more layers than the kernel's per-layer table holds");
            std::vector<unsigned char> cnc(q), nph(q, 1), phase((size_t)q * M, 0);
            std::vector<unsigned short> pos((size_t)q * CNL * M, 0);
            for (int i = 0; i < q; i++)
            {
                cnc[i] = (unsigned char)cn[i];
                std::vector<int> last(K, -1); // highest phase among the checks of this layer that touch the bit so far
                for (int j = 0; j < M; j++)
                {
                    if (cn[q * j + i] != cn[i])
                        throw HipError("dvbs2 ldpc: table inconsistent (layer not regular)");
                    int ph = 0;
                    for (int cc = 0; cc < cn[i]; cc++)
                    {
                        const int b = pos0[(size_t)CNL * (q * j + i) + cc]; // check q*j + i of the matrix is check (layer i, j) of the decoder
                        pos[((size_t)i * CNL + cc) * M + j] = (unsigned short)b;
                        ph = std::max(ph, last[b] + 1);
                    }
                    for (int cc = 0; cc < cn[i]; cc++)
                        last[pos0[(size_t)CNL * (q * j + i) + cc]] = ph;
                    if (ph > 250)
                        throw HipError("dvbs2 ldpc: too many dependent checks in one layer");
                    phase[(size_t)i * M + j] = (unsigned char)ph;
                    nph[i] = (unsigned char)std::max<int>(nph[i], ph + 1);
                }
                max_phases = std::max<int>(max_phases, nph[i]);
                conflict_layers += nph[i] > 1;
                if (getenv("SDHIP_LDPC_DUMP") && nph[i] > 1)
                { // the layer's dependency structure: checks per phase
                    std::vector<int> cntp(nph[i], 0);
                    for (int j = 0; j < M; j++)
                        cntp[phase[(size_t)i * M + j]]++;
                    int mx = 0, mx1 = 0;
                    for (int p2 = 0; p2 < nph[i]; p2++)
                    {
                        mx = std::max(mx, cntp[p2]);
                        if (p2)
                            mx1 = std::max(mx1, cntp[p2]);
                    }
                    fprintf(stderr, "[sdhip] ldpc table %d layer %d: degree %d + 2, %d phases, phase 0 has %d checks, later phases at most %d\n", ti, i, cn[i], nph[i], cntp[0], mx1);
                }
            }
            const int DQ = (CNL + 2 + 3) / 4;
            // narrow layers (LdpcDev): schedule position k = phase * W + slot, W = the widest phase; a link per lane, G lanes per check
            int G = 4;
            while (G < CNL + 2)
                G *= 2;
            std::vector<unsigned char> narrow(q, 0);
            int MB = M;
            const bool use_narrow = !(getenv("SDHIP_LDPC_NARROW") && atoi(getenv("SDHIP_LDPC_NARROW")) == 0);
            std::vector<std::vector<int>> per_phase_count(q);
            for (int i = 0; i < q; i++)
            {
                std::vector<int> cntp(nph[i], 0);
                for (int j = 0; j < M; j++)
                    cntp[phase[(size_t)i * M + j]]++;
                const int W = *std::max_element(cntp.begin(), cntp.end());
                if (use_narrow && nph[i] >= 8 && W * G <= LDPC_THREADS && W <= 255 && G <= 32)
                {
                    narrow[i] = (unsigned char)W;
                    MB = std::max(MB, (int)nph[i] * W);
                    narrow_layers++;
                }
            }
            std::vector<unsigned short> snode((size_t)q * MB * 4 * DQ, 0xFFFFu);
            for (int i = 0; i < q; i++)
            {
                if (!narrow[i])
                    continue;
                const int W = narrow[i];
                std::vector<int> slot(nph[i], 0);
                for (int j = 0; j < M; j++)
                {
                    const int ph = phase[(size_t)i * M + j];
                    const size_t k = (size_t)ph * W + slot[ph]++;
                    unsigned short *e = &snode[((size_t)i * MB + k) * 4 * DQ];
                    const int cnt = cn[i];
                    for (int d = 0; d < cnt; d++)
                        e[d] = pos[((size_t)i * CNL + d) * M + j];
                    e[cnt] = (unsigned short)(K + M * i + j); // the check's own parity bit
                    if ((i | j) != 0)                        // and its predecessor's (layered_decoder.hh:157-159; check (0, 0) has none)
                        e[cnt + 1] = (unsigned short)(i ? K + M * (i - 1) + j : K + (q - 1) * M + j - 1);
                }
            }
            std::vector<unsigned> ndp((size_t)q * 2 * DQ * M, 0xFFFFFFFFu);
            for (int i = 0; i < q; i++)
                for (int j = 0; j < M; j++)
                {
                    const int cnt = cn[i];
                    auto put = [&](int d, unsigned n) {
                        unsigned &w = ndp[((size_t)i * 2 * DQ + (d >> 1)) * M + j];
                        w = (d & 1) ? ((w & 0x0000FFFFu) | (n << 16)) : ((w & 0xFFFF0000u) | n);
                    };
                    for (int d = 0; d < cnt; d++)
                        put(d, pos[((size_t)i * CNL + d) * M + j]);
                    put(cnt, (unsigned)(K + M * i + j));
                    if ((i | j) != 0)
                        put(cnt + 1, (unsigned)(i ? K + M * (i - 1) + j : K + (q - 1) * M + j - 1));
                }
            d_ndp.reserve(ndp.size());
            SD_HIP(hipMemcpy(d_ndp.p, ndp.data(), ndp.size() * sizeof(unsigned), hipMemcpyHostToDevice));
            d_narrow.reserve(q);
            d_snode.reserve(snode.size());
            SD_HIP(hipMemcpy(d_narrow.p, narrow.data(), q, hipMemcpyHostToDevice));
            SD_HIP(hipMemcpy(d_snode.p, snode.data(), snode.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
            d_cnc.reserve(q);
            d_nph.reserve(q);
            d_phase.reserve(phase.size());
            SD_HIP(hipMemcpy(d_cnc.p, cnc.data(), q, hipMemcpyHostToDevice));
            SD_HIP(hipMemcpy(d_nph.p, nph.data(), q, hipMemcpyHostToDevice));
            SD_HIP(hipMemcpy(d_phase.p, phase.data(), phase.size(), hipMemcpyHostToDevice));
            g = LdpcDev{M, N, K, R, q, CNL, DQ, d_ndp.p, d_cnc.p, d_phase.p, d_nph.p, MB, G, d_narrow.p, d_snode.p, (getenv("SDHIP_LDPC_PACKED") && atoi(getenv("SDHIP_LDPC_PACKED")) == 0) ? 0 : 1};
            info.code_len = N;
            info.data_len = K;
            info.layers = q;
            info.links_total = tab->links_total;
            info.max_phases = max_phases;
            info.layers_with_shared_bits = conflict_layers;
            info.msg_bytes_per_frame = (uint64_t)q * DQ * M * 4; // (what the decoder reads and writes; the allocation is padded to MB checks per layer)
        }
        ~LdpcEngine()
        {
            if (stream)
                (void)hipStreamDestroy(stream);
        }

        // BBFrameLDPC::decode over nframes / batch calls (bbframe_ldpc.cpp:114-124, module_dvbs2_demod.cpp:246-257), frames in place
        int decode_dev(signed char *d_fr, int nframes, int max_trials, int *d_trials_out)
        {
            SD_HIP(hipSetDevice(cfg.device));
            if (nframes <= 0)
                return 0;
            if (nframes % cfg.batch)
                throw HipError("dvbs2 ldpc: the frame count must be a multiple of the batch");
            if (max_trials < 0)
                max_trials = 0;
            if (getenv("SDHIP_LDPC_PROBE") && !d_probe.p)
            {
                d_probe.reserve(8);
                SD_HIP(hipMemset(d_probe.p, 0, 8 * sizeof(long long)));
            }
            const size_t per = (size_t)g.q * g.DQ * g.MB;
            d_bnl.reserve(per * nframes);
            d_flags.reserve(3 * (size_t)nframes + 8);
            SD_HIP(hipMemsetAsync(d_bnl.p, 0, per * nframes * sizeof(unsigned), stream)); // reset(): bnl = 0
            SD_HIP(hipMemsetAsync(d_flags.p, 0, (3 * (size_t)nframes + 8) * sizeof(int), stream));
            int *bad0 = d_flags.p, *bad1 = d_flags.p + nframes, *upd = d_flags.p + 2 * nframes, *any = d_flags.p + 3 * nframes;
            int ran = 0;
            for (int t = 0; t <= max_trials; t++)
            {
                int *prev = (t & 1) ? bad0 : bad1, *cur = (t & 1) ? bad1 : bad0;
                SD_HIP(hipMemsetAsync(any, 0, sizeof(int), stream));
                {
                    ProfScope _ps("k_ldpc_trial", stream);
                    launch_trial(d_fr, nframes, t, prev, cur, upd, any);
                }
                ran = t;
                int h_any = 0;
                SD_HIP(hipMemcpyAsync(&h_any, any, sizeof(int), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
                if (!h_any)
                    break;
            }
            const int nb = nframes / cfg.batch;
            hipLaunchKernelGGL(k_ldpc_result, dim3((nb + 63) / 64), dim3(64), 0, stream, nb, cfg.batch, (ran & 1) ? bad1 : bad0, upd, d_trials_out);
            SD_HIP(hipStreamSynchronize(stream));
            if (d_probe.p)
            {
                long long h[8];
                SD_HIP(hipMemcpy(h, d_probe.p, sizeof(h), hipMemcpyDeviceToHost));
                fprintf(stderr, "[sdhip] ldpc probe (ticks, workgroup 0, %lld launches of which %lld updating): frame in %lld, load wait %lld, phases %lld, stores %lld, parity check %lld, frame out %lld\n",
                        h[7], h[6], h[0], h[1], h[2], h[3], h[4], h[5]);
            }
            return ran;
        }
        void launch_trial(signed char *d_fr, int nframes, int t, const int *prev, int *cur, int *upd, int *any)
        {
            const dim3 grid((unsigned)nframes), block(LDPC_THREADS);
#define SD_LDPC_CASE(D)                                                                                                                      \
    case D:                                                                                                                                  \
        hipLaunchKernelGGL((k_ldpc_trial<D>), grid, block, 0, stream, g, d_fr, d_bnl.p, nframes, cfg.batch, t, prev, cur, upd, any, d_probe.p); \
        break;
            switch (g.DQ)
            {
                SD_LDPC_CASE(1)
                SD_LDPC_CASE(2)
                SD_LDPC_CASE(3)
                SD_LDPC_CASE(4)
                SD_LDPC_CASE(5)
                SD_LDPC_CASE(6)
                SD_LDPC_CASE(7)
                SD_LDPC_CASE(8)
            default:
                throw HipError("dvbs2 ldpc: check degree beyond 32");
            }
#undef SD_LDPC_CASE
        }
        int decode_host(signed char *frames, int nframes, int max_trials, int *trials_out)
        {
            SD_HIP(hipSetDevice(cfg.device));
            if (nframes <= 0)
                return 0;
            d_frames.reserve((size_t)nframes * g.N);
            d_trials.reserve(nframes);
            SD_HIP(hipMemcpyAsync(d_frames.p, frames, (size_t)nframes * g.N, hipMemcpyHostToDevice, stream));
            const int r = decode_dev(d_frames.p, nframes, max_trials, d_trials.p);
            SD_HIP(hipMemcpyAsync(frames, d_frames.p, (size_t)nframes * g.N, hipMemcpyDeviceToHost, stream));
            SD_HIP(hipMemcpyAsync(trials_out, d_trials.p, (size_t)(nframes / cfg.batch) * sizeof(int), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            return r;
        }
    };
} // namespace sdhip

using namespace sdhip;
#define SD_GUARD_BEGIN try {
#define SD_GUARD_END(ret)           \
    }                               \
    catch (const std::exception &e) \
    {                               \
        sdhip::set_error(e.what()); \
        return ret;                 \
    }

extern "C"
{
    void *sdhip_ldpc_create(const sdhip_ldpc_cfg *cfg)
    {
        SD_GUARD_BEGIN
        return new LdpcEngine(*cfg);
        SD_GUARD_END(nullptr)
    }
    void sdhip_ldpc_destroy(void *h) { delete static_cast<LdpcEngine *>(h); }
    int sdhip_ldpc_get_info(void *h, sdhip_ldpc_info *out)
    {
        SD_GUARD_BEGIN
        *out = static_cast<LdpcEngine *>(h)->info;
        return 0;
        SD_GUARD_END(-1)
    }
    int sdhip_ldpc_decode_dev(void *h, int8_t *d_frames, int nframes, int max_trials, int *d_trials)
    {
        SD_GUARD_BEGIN
        if (reinterpret_cast<uintptr_t>(d_frames) & 3u) // the kernel moves the frames as 32-bit words (code_len is a multiple of 4)
            throw HipError("dvbs2 ldpc: d_frames must be 4-byte aligned");
        return static_cast<LdpcEngine *>(h)->decode_dev(reinterpret_cast<signed char *>(d_frames), nframes, max_trials, d_trials);
        SD_GUARD_END(-1)
    }
    int sdhip_ldpc_decode(void *h, int8_t *frames, int nframes, int max_trials, int *trials)
    {
        SD_GUARD_BEGIN
        return static_cast<LdpcEngine *>(h)->decode_host(reinterpret_cast<signed char *>(frames), nframes, max_trials, trials);
        SD_GUARD_END(-1)
    }
}
