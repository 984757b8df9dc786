// fec_kernels.hip -- CCSDS FEC chain on gfx950 (MI355X): k=7 r=1/2 Viterbi (ACS + traceback),
// BER re-encode, lock search, ASM search, frame extraction, derandomiser, RS(255,223/239).
//
// Design:
//  * Viterbi ACS: ONE WAVEFRONT PER BLOCK, lane = trellis state. The lane<->state map rotates every
//    step (state held by lane l after t steps = rotl6(l, t)), so a butterfly needs ONE cross-lane
//    exchange with lane l ^ (32 >> (t % 6)); new metric = min(mine, other) and the decision bits fall
//    out of two v_cmp masks. Metrics are 8-bit values with the reference's wrap + per-step min
//    renormalisation (volk_k7_r2_generic_fixed.h:95-163) reproduced exactly.
//  * Decisions: one 64-bit lane-order ballot per step, gathered 64 steps at a time and written as one
//    coalesced 512-byte store to an HBM scratch; traceback is segment-parallel (64 lanes x F/64 steps,
//    each starting VIT_TB_OVERLAP steps early) with an exactness certificate and a serial fallback.
//  * Block -> block start-state chaining (cc_decoder.cpp:295-302) is speculated by replaying the last
//    VIT_PREPASS steps of the previous block; the host verifies and re-decodes on a miss.
//  * RS: one thread per codeword, GF(256) log/exp tables in LDS; Berlekamp-Massey / Chien / Forney mirror
//    libcorrect's control flow step for step (src-core/libs/correct/reed-solomon/decode.c).
#include "fec_kernels.h"

namespace sdhip
{
    // =============================================================================================
    // Symbol fetch: rotate_soft (rotation.cpp:4-63) + signed_soft_to_unsigned (viterbi/utils.cpp:3-12)
    // + the decoder's view of the buffer (viterbi_1_2.cpp:94-98) / MetOp depuncture (viterbi_3_4.cpp:84-105)
    // =============================================================================================
    struct SymFetch
    {
        VitCfg c;
        const int8_t *blk;
        int limit; // soft bytes visible to the decoder (B, or 2048 in the lock search)

        __device__ __forceinline__ unsigned u_at(int j) const
        {
            if (j < 0 || j >= limit)
                return 128u;
            const int q = j & ~1;
            int a = blk[q], b = blk[q + 1];
            if (a == -128)
                a = -127;
            if (b == -128)
                b = -127;
            if (c.pre_swap)
            {
                int t = a;
                a = b;
                b = t;
            }
            if (c.iq_swap)
            {
                int t = a;
                a = b;
                b = t;
            }
            switch (c.phase)
            {
            case 1:
            {
                int t = a;
                a = b;
                b = -t;
                break;
            }
            case 2:
                a = -a;
                b = -b;
                break;
            case 3:
            {
                int t = a;
                a = -b;
                b = t;
                break;
            }
            default:
                break;
            }
            const int v = (j & 1) ? b : a;
            unsigned u = (unsigned)(v + 127) & 255u;
            if (u == 128u)
                u = 127u;
            return u;
        }

        // symbol pair (s0 | s1 << 8) consumed by decoder step t; `tail` supplies symbols past `limit`
        template <class Tail>
        __device__ __forceinline__ unsigned pair(int t, const Tail &tail) const
        {
            unsigned s0, s1;
            if (c.mode == 2)
            { // raw unsigned symbols, tail included in the data (sdhip_op_ccdecoder)
                const unsigned char *ub = (const unsigned char *)blk;
                const int j = 2 * t;
                s0 = (j < limit) ? ub[j] : 128u;
                s1 = (j + 1 < limit) ? ub[j + 1] : 128u;
            }
            else if (c.mode == 0)
            {
                const int j = c.shift + 2 * t;
                s0 = (j < limit) ? u_at(j) : tail(j - limit);
                s1 = (j + 1 < limit) ? u_at(j + 1) : tail(j + 1 - limit);
            }
            else
            {
                const int m = t / 3, r = t - 3 * m;
                const int base = 4 * m;
                const int x = c.fy ? 1 : 0; // FengYun: the punctured pair is not swapped (viterbi_3_4.cpp:63-75 against :91-102)
                if (base >= limit) // past the depunctured data (limit is a multiple of 4)
                {
                    s0 = tail(0);
                    s1 = tail(0);
                }
                else if (c.shift == 0)
                {
                    if (r == 0)
                    {
                        s0 = u_at(base);
                        s1 = u_at(base + 1);
                    }
                    else if (r == 1)
                    {
                        s0 = 128u;
                        s1 = u_at(base + (3 ^ x));
                    }
                    else
                    {
                        s0 = u_at(base + (2 ^ x));
                        s1 = 128u;
                    }
                }
                else
                {
                    if (r == 0)
                    {
                        s0 = 128u;
                        s1 = u_at(base + (1 ^ x));
                    }
                    else if (r == 1)
                    {
                        s0 = u_at(base + (0 ^ x));
                        s1 = 128u;
                    }
                    else
                    {
                        s0 = u_at(base + 2);
                        s1 = u_at(base + 3);
                    }
                }
            }
            return s0 | (s1 << 8);
        }
    };

    struct TailErasure
    {
        __device__ __forceinline__ unsigned operator()(int) const { return 128u; }
    };

    // =============================================================================================
    // ACS forward pass (one wave). X = this lane's path metric. Sink receives the ballot of each step.
    // =============================================================================================
    struct AcsConsts
    {
        unsigned m0[6], m1[6];        // per-lane branch masks (0 / 255) for the 6 layout phases
        unsigned long long himask[6]; // wave masks: lane holds an "upper" old state (state >= 32)
    };

    __device__ __forceinline__ void acs_init_consts(AcsConsts &k)
    {
        const unsigned lane = (unsigned)lane_id();
#pragma unroll
        for (int p = 0; p < 6; p++)
        {
            const unsigned st = rotl6(lane, p);
            const unsigned i = st & 31u;
            // Branchtab, cc_decoder.cpp:116-123: polys {79, 109}
            k.m0[p] = parity32((2u * i) & 79u) ? 255u : 0u;
            k.m1[p] = parity32((2u * i) & 109u) ? 255u : 0u;
            k.himask[p] = __ballot((st >> 5) != 0);
        }
    }

    template <int P>
    __device__ __forceinline__ unsigned acs_step(unsigned X, unsigned s0, unsigned s1, const AcsConsts &k, unsigned long long &ballot)
    {
        // BFLY, volk_k7_r2_generic_fixed.h:95-134 (both lanes of a butterfly share the same branch metric)
        const unsigned metric = (1u + (s0 ^ k.m0[P]) + (s1 ^ k.m1[P])) >> 3;
        const unsigned mine = (X + metric) & 255u;
        const unsigned send = (X + 63u - metric) & 255u;
        const unsigned other = (unsigned)__shfl_xor((int)send, 32 >> P);
        const unsigned Y = min(mine, other);
        const unsigned long long gt = __ballot(mine > other);
        const unsigned long long eq = __ballot(mine == other);
        ballot = eq | (gt ^ k.himask[P]);
        // renormalize, volk_k7_r2_generic_fixed.h:80-92
        return Y - wave_min_u32(Y);
    }

    template <class Fetch, class Sink>
    __device__ __forceinline__ unsigned acs_forward(int nsteps, unsigned X, const AcsConsts &k, const Fetch &fetch, Sink &sink)
    {
        const int lane = lane_id();
        unsigned pairs = 0;
        int t = 0;
#define SD_ACS_STEP(P)                                                                   \
    {                                                                                    \
        if ((t & 63) == 0)                                                               \
            pairs = fetch(t + lane);                                                     \
        const unsigned pr = (unsigned)__builtin_amdgcn_readlane((int)pairs, t & 63);     \
        unsigned long long bal;                                                          \
        X = acs_step<P>(X, pr & 255u, (pr >> 8) & 255u, k, bal);                         \
        sink.put(t, bal);                                                                \
        t++;                                                                             \
    }
        while (t + 6 <= nsteps)
        {
            SD_ACS_STEP(0)
            SD_ACS_STEP(1)
            SD_ACS_STEP(2)
            SD_ACS_STEP(3)
            SD_ACS_STEP(4)
            SD_ACS_STEP(5)
        }
        const int rem = nsteps - t;
        if (rem > 0) SD_ACS_STEP(0)
        if (rem > 1) SD_ACS_STEP(1)
        if (rem > 2) SD_ACS_STEP(2)
        if (rem > 3) SD_ACS_STEP(3)
        if (rem > 4) SD_ACS_STEP(4)
#undef SD_ACS_STEP
        sink.finish(nsteps);
        return X;
    }

    // find_endstate (cc_decoder.cpp:192-209): first state index holding the minimum metric.
    __device__ __forceinline__ unsigned acs_endstate(unsigned X, int nsteps)
    {
        const unsigned st = rotl6((unsigned)lane_id(), (unsigned)(nsteps % 6));
        const unsigned mn = wave_min_u32(X);
        return wave_min_u32(X == mn ? st : 64u);
    }

    // decision bit of `state` (the NEW state) at step t, ballots are in lane order
    __device__ __forceinline__ unsigned dec_bit(unsigned long long ballot, int t, unsigned state)
    {
        const unsigned l = rotr6(state, (unsigned)((t + 1) % 6));
        return (unsigned)(ballot >> l) & 1u;
    }

    // Sink: 64 ballots gathered in registers, one coalesced store per 64 steps
    struct SinkGlobal
    {
        unsigned long long *dec;
        unsigned long long rec;
        __device__ __forceinline__ void put(int t, unsigned long long b)
        {
            if (lane_id() == (t & 63))
                rec = b;
            if ((t & 63) == 63)
                dec[(t - 63) + lane_id()] = rec;
        }
        __device__ __forceinline__ void finish(int nsteps)
        {
            const int tail = nsteps & 63;
            if (tail && lane_id() < tail)
                dec[(nsteps - tail) + lane_id()] = rec;
        }
    };
    struct SinkLds
    {
        unsigned long long *dec;
        __device__ __forceinline__ void put(int t, unsigned long long b)
        {
            if (lane_id() == (t & 63))
                dec[t] = b;
        }
        __device__ __forceinline__ void finish(int) {}
    };

    // =============================================================================================
    // k_vit_decode
    // =============================================================================================
    constexpr int PRE_LDS = VIT_PREPASS + 8;

    // `list` (optional): the nblk blocks to decode are list[0..nblk) (indices relative to first_block) instead of 0..nblk;
    // the decision scratch is always indexed by the position in the launch.
    __global__ __launch_bounds__(256) void k_vit_decode(VitCfg c, const int8_t *__restrict__ soft, long long first_block, int nblk, VitBlockIO *io,
                                                         unsigned long long *decisions, unsigned *vbits, int dstride, int wpb, const int *__restrict__ list)
    {
        __shared__ unsigned long long pre_ballots[4][PRE_LDS];
        const int wave = (int)(threadIdx.x >> 6);
        const int lane = lane_id();
        const int slot = (int)blockIdx.x * 4 + wave;
        if (slot >= nblk)
            return;
        const int j = list ? list[slot] : slot;
        const int F = c.F, nsteps = F + 6;
        AcsConsts k;
        acs_init_consts(k);
        const TailErasure erasure;

        // ---- start state -------------------------------------------------------------------------
        int start = io[j].start_in;
        if (start == -1)
        {
            // speculate: replay the tail of the previous block from neutral metrics, take the state 6 steps
            // before its end state (== CCDecoder::work's chained start state when survivor paths have merged)
            SymFetch pf{c, soft + (first_block + j - 1) * vit_stride(c), c.B};
            const int P = VIT_PREPASS < F ? VIT_PREPASS : F;
            const int t0 = F - P, n = P + 6;
            SinkLds sk{pre_ballots[wave]};
            auto fetch = [&](int tt) -> unsigned { return pf.pair(t0 + tt, erasure); };
            unsigned X = acs_forward(n, 0u, k, fetch, sk);
            unsigned st = acs_endstate(X, n);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (int tt = n - 1; tt >= n - 6; tt--)
            {
                const unsigned kb = dec_bit(pre_ballots[wave][tt], tt, st);
                st = (st >> 1) | (kb << 5);
            }
            start = (int)st;
        }

        // ---- forward pass over the block ------------------------------------------------------------
        SymFetch f{c, soft + (first_block + j) * vit_stride(c), c.B};
        unsigned long long *dec = decisions + (size_t)slot * dstride;
        unsigned X = (start == -2) ? 31u : ((lane == start) ? 0u : 63u); // init_viterbi(_unbiased), cc_decoder.cpp:159-190
        {
            SinkGlobal sk{dec, 0ull};
            auto fetch = [&](int tt) -> unsigned { return f.pair(tt, erasure); };
            X = acs_forward(nsteps, X, k, fetch, sk);
        }
        const unsigned endstate = acs_endstate(X, nsteps);
        __threadfence(); // decisions written by other lanes of this wave are read below

        // ---- chained start state for the next block: state after steps F+5..F (cc_decoder.cpp:250-260,275)
        unsigned ret = endstate;
        for (int t = F + 5; t >= F; t--)
        {
            const unsigned kb = dec_bit(dec[t], t, ret);
            ret = (ret >> 1) | (kb << 5);
        }

        // ---- segment-parallel traceback ---------------------------------------------------------------
        const int L = wpb / 2;                 // bits per lane = wpb*32/64
        const int wpl = L / 32;                // words per lane
        const int nseg = (F + L - 1) / L;
        int hi = 6 + (lane + 1) * L;
        if (hi > F + 6)
            hi = F + 6;
        hi -= 1;
        const bool active = lane < nseg;
        int tstart = hi + VIT_TB_OVERLAP;
        if (tstart > F + 5)
            tstart = F + 5;
        unsigned st = (tstart == F + 5) ? endstate : 0u;
        unsigned entry = 0, exitst = 0;
        unsigned *vb = vbits + (size_t)j * wpb + (size_t)lane * wpl;
        if (active)
        {
            for (int t = tstart; t > hi; t--) // overlap: converge onto the survivor path
            {
                const unsigned kb = dec_bit(dec[t], t, st);
                st = (st >> 1) | (kb << 5);
            }
            entry = st;
            for (int w = wpl - 1; w >= 0; w--)
            {
                unsigned word = 0;
                for (int b = 31; b >= 0; b--)
                {
                    const int n = lane * L + w * 32 + b;
                    if (n < F)
                    {
                        const int t = n + 6;
                        const unsigned kb = dec_bit(dec[t], t, st);
                        st = (st >> 1) | (kb << 5);
                        word |= kb << (31 - b);
                    }
                }
                vb[w] = word;
            }
            exitst = st;
        }
        else
        {
            for (int w = 0; w < wpl; w++)
                vb[w] = 0;
        }
        // certificate: every segment must end where the one below it assumed it starts
        const unsigned exit_above = (unsigned)__shfl_down((int)exitst, 1);
        // lane l's ENTRY state (top of its segment) must equal lane l+1's EXIT state
        const bool mismatch = active && (lane + 1 < nseg) && (entry != exit_above);
        int fallback = 0;
        if (__ballot(mismatch) != 0ull)
        {
            fallback = 1;
            if (lane == 0)
            { // serial chainback, cc_decoder.cpp:228-276
                unsigned s = endstate; // decoded bit n is the decision read at step n + 6, starting at step F + 5
                unsigned *vball = vbits + (size_t)j * wpb;
                unsigned word = 0;
                for (int n = F - 1; n >= 0; n--)
                {
                    const int t = n + 6;
                    const unsigned kb = dec_bit(dec[t], t, s);
                    s = (s >> 1) | (kb << 5);
                    word |= kb << (31 - (n & 31));
                    if ((n & 31) == 0)
                    {
                        vball[n >> 5] = word;
                        word = 0;
                    }
                }
            }
        }
        if (lane == 0)
        {
            io[j].start_used = start;
            io[j].ret_state = (int)ret;
            io[j].end_state = (int)endstate;
            io[j].tb_fallback = fallback;
        }
    }

    void launch_vit_decode(const VitCfg &cfg, const int8_t *soft, int64_t first_block, int nblk, VitBlockIO *io, uint64_t *decisions, uint32_t *vbits,
                           hipStream_t st, const int *list)
    {
        if (nblk <= 0)
            return;
        const int dstride = (cfg.F + 6 + 63) / 64 * 64;
        const int wpb = vit_words_per_block(cfg.F);
        ProfScope _ps("k_vit_decode", st);
        hipLaunchKernelGGL(k_vit_decode, dim3((nblk + 3) / 4), dim3(256), 0, st, cfg, soft, (long long)first_block, nblk, io,
                           (unsigned long long *)decisions, vbits, dstride, wpb, list);
    }

    // =============================================================================================
    // k_vit2_*: lane-per-segment decoder, 64 path metrics packed two per VGPR
    // =============================================================================================
    // Register r (0..31) holds the metrics of states r (low half) and 63-r (high half). This pairing is the one that is
    // invariant under the trellis shift: butterfly i (inputs X[i], X[i+32]; outputs Y[2i], Y[2i+1]) and butterfly 31-i
    // are evaluated by the same packed instructions, and their four outputs land again as (Y[s], Y[63-s]) pairs:
    //   R1 = R[i]        = (X[i],    X[63-i]) = (A_i, B_i')        R2 = swap(R[31-i]) = (X[i+32], X[31-i]) = (B_i, A_i')
    //   T1 = R1 + (M, M')       = (m0, m3')      T2 = R2 + (63-M, 63-M')   = (m1, m2')    E = min(T1,T2) = (Y[2i],   Y[63-2i])
    //   T3 = R1 + (63-M, 63-M') = (m2, m1')      T4 = R2 + (M, M')         = (m3, m0')    O = min(T3,T4) = (Y[2i+1], Y[62-2i])
    // (M, M' = branch metrics of butterflies i and 31-i; their branch-table bits are complements of each other.)
    // Values live in the HIGH BYTE of the 16-bit fields: a packed 16-bit add then wraps exactly like the reference's unsigned char arithmetic
    // (volk_k7_r2_generic_fixed.h:118-121) with no masking, and the low byte is free for a TAG that does two jobs at once. Every register carries
    // tag 1 in its low half (0 in its high half; swapped for R2), branch constants carry none, so of the two candidates of a compare-select exactly one
    // has low byte 1: (a) on a tie of the values the untagged candidate wins the 16-bit minimum -- placed so that it is the one the reference's
    // `decision = (m0 - m1) >= 0` picks (:123-127) --, and (b) the low byte of the minimum says which candidate won: the (inverted) decision bit
    // comes out of the select itself, no subtraction. One v_and_or puts the tag back on a new metric; v_perm gathers the four low bytes of (E, O) and
    // v_lshl_or shifts them into the decision words. The per-step renormalisation (subtract the minimum, :80-92) is folded into the next step's
    // branch constants. ~12 instructions per butterfly pair (4 add, 2 min, 2 and_or, 2 running min, perm, shift-or); the doubled-value form with
    // masks and sign extraction this replaces took 17.
    typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ v2u16 v2v(unsigned x) { return __builtin_bit_cast(v2u16, x); }
    __device__ __forceinline__ unsigned v2u(v2u16 x) { return __builtin_bit_cast(unsigned, x); }
    __device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) { return v2u(v2v(a) + v2v(b)); }
    __device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) { return v2u(v2v(a) - v2v(b)); }
    __device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b) { return v2u(__builtin_elementwise_min(v2v(a), v2v(b))); }
    __device__ __forceinline__ unsigned pk_swap(unsigned a)
    {
        const v2u16 v = v2v(a);
        return v2u(__builtin_shufflevector(v, v, 1, 0));
    }
    constexpr unsigned V2_MASK = 0xFF00FF00u; // the metric bytes
    constexpr unsigned V2_TAG = 0x00000001u;  // low half tagged, high half not
    // Constants that feed three-operand ops ((x & m) | y = one v_and_or_b32) must live in registers: VOP3 encodings take no literals on
    // gfx9 and one scalar operand at most. The asm keeps the compiler from folding them back into literals.
    struct V2Consts
    {
        unsigned tag;  // VGPR
        unsigned mask; // SGPR
    };
    __device__ __forceinline__ void v2_consts(V2Consts &k)
    {
        asm volatile("v_mov_b32 %0, 0x1" : "=v"(k.tag));
        asm volatile("s_mov_b32 %0, 0xff00ff00" : "=s"(k.mask));
    }

    struct V2State
    {
        unsigned R[32]; // (Y[r] << 8 | 1, Y[63-r] << 8), not yet renormalised
        unsigned C2;    // min(Y) << 8 in both halves: true metric = (half - C) >> 8 (mod 256)
    };
    __device__ __forceinline__ void v2_init_neutral(V2State &s)
    {
#pragma unroll
        for (int r = 0; r < 32; r++)
            s.R[r] = V2_TAG;
        s.C2 = 0u;
    }
    // init_viterbi / init_viterbi_unbiased, cc_decoder.cpp:159-190
    __device__ __forceinline__ void v2_init_start(V2State &s, int start)
    {
        const unsigned all = ((start == -2) ? 31u : 63u) << 8;
#pragma unroll
        for (int r = 0; r < 32; r++)
        {
            unsigned lo = all, hi = all;
            if (start == r)
                lo = 0u;
            if (start == 63 - r)
                hi = 0u;
            s.R[r] = lo | (hi << 16) | V2_TAG;
        }
        s.C2 = 0u;
    }
    // normalised packed metrics (for the certificate and the end state): the values in the high bytes, low bytes clear
    __device__ __forceinline__ unsigned v2_norm(const V2State &s, int r) { return pk_sub(s.R[r], s.C2) & V2_MASK; }

    // find_endstate (cc_decoder.cpp:192-209): first state index holding the minimum metric
    __device__ __forceinline__ unsigned v2_endstate(const V2State &s)
    {
        unsigned best = 0xFFFFFu, idx = 0;
#pragma unroll
        for (int st = 0; st < 64; st++)
        {
            const int r = st < 32 ? st : 63 - st;
            const unsigned v = v2_norm(s, r);
            const unsigned x = st < 32 ? (v & 0xFFFFu) : (v >> 16);
            if (x < best)
            {
                best = x;
                idx = (unsigned)st;
            }
        }
        return idx;
    }

    // decision bit of NEW state st in the two decision words of a step
    __device__ __forceinline__ unsigned v2_dec_bit(unsigned w0, unsigned w1, unsigned st)
    {
        const unsigned r = st < 32u ? st : 63u - st;
        const unsigned i = r >> 1;
        const unsigned byte = (st < 32u ? 0u : 1u) + ((r & 1u) << 1);
        const unsigned w = (i & 8u) ? w1 : w0;
        return (w >> (byte * 8u + 7u - (i & 7u))) & 1u;
    }

    // one trellis step on symbol pair sym = s0 | s1 << 8
    template <bool DEC>
    __device__ __forceinline__ void v2_step(V2State &s, const V2Consts &kc, unsigned sym, unsigned &w0, unsigned &w1)
    {
        const unsigned s0 = sym & 255u, s1 = (sym >> 8) & 255u;
        const unsigned a[2] = {s0, s0 ^ 255u}, c[2] = {s1, s1 ^ 255u};
        // BFLY metric (volk_k7_r2_generic_fixed.h:110-113) in the high byte: class = b0*2 + b1 (Branchtab bits of the butterfly)
        unsigned D[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            D[k] = ((a[k >> 1] + c[k & 1] + 1u) << 5) & 0x3F00u; // ((sum + 1) >> 3) << 8
        const unsigned Q = pk_sub(0x3F003F00u, s.C2);
        unsigned K1[4], K2[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const unsigned PD = D[k] | (D[3 - k] << 16); // (this butterfly's class, the complementary class of butterfly 31-i)
            K1[k] = pk_sub(PD, s.C2);
            K2[k] = pk_sub(Q, PD);
        }
        unsigned Rn[32];
        unsigned mnA = 0xFFFFFFFFu, mnB = 0xFFFFFFFFu;
        unsigned acc0 = 0u, acc1 = 0u;
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            // Branchtab (cc_decoder.cpp:116-123, polys 79 / 109): b0 = parity(2i & 79) = i0^i1^i2, b1 = parity(2i & 109) = i1^i2^i4
            const int b0 = (i ^ (i >> 1) ^ (i >> 2)) & 1, b1 = ((i >> 1) ^ (i >> 2) ^ (i >> 4)) & 1;
            const int k = b0 * 2 + b1;
            // tags: R1 = (.. | 1, .. | 0), R2 = (.. | 0, .. | 1). Low half: a tie goes to R2's candidate (m1 / m3: decision 1), high half (the mirror
            // butterfly, candidates in the other order) to R1's: each time the reference's choice. Low byte of the minimum = 1 <=> decision 0.
            const unsigned R1 = s.R[i], R2 = pk_swap(s.R[31 - i]);
            const unsigned T1 = pk_add(R1, K1[k]);
            const unsigned T2 = pk_add(R2, K2[k]);
            const unsigned T3 = pk_add(R1, K2[k]);
            const unsigned T4 = pk_add(R2, K1[k]);
            const unsigned E = pk_min(T1, T2), O = pk_min(T3, T4);
            Rn[2 * i] = (E & kc.mask) | kc.tag;
            Rn[2 * i + 1] = (O & kc.mask) | kc.tag;
            mnA = pk_min(mnA, E);
            mnB = pk_min(mnB, O);
            if (DEC)
            {
                // bytes: [E.lo, E.hi, O.lo, O.hi] low bytes, each 0 or 1 (= the inverted decision)
                const unsigned p = __builtin_amdgcn_perm(O, E, 0x06040200u);
                if (i < 8)
                    acc0 = (acc0 << 1) | p;
                else
                    acc1 = (acc1 << 1) | p;
            }
        }
#pragma unroll
        for (int r = 0; r < 32; r++)
            s.R[r] = Rn[r];
        const unsigned mn = pk_min(mnA, mnB);
        const unsigned m1 = min(mn & 0xFFFFu, mn >> 16) & 0xFF00u;
        s.C2 = m1 | (m1 << 16);
        if (DEC)
        {
            w0 = ~acc0;
            w1 = ~acc1;
        }
    }

    // ---- symbol staging: row j = [VIT2_WARM last steps of block j-1 | steps 0..F+5 of block j | erasures]
    // The soft bytes a block of 256 threads (2048 steps) needs from block j are consecutive: they are staged into LDS with coalesced
    // dword loads first (at the byte alignment they have in memory), and the per-symbol fetch -- rotation, IQ swap, shift,
    // MetOp depuncture: SymFetch -- then reads bytes from LDS instead of issuing two byte loads per symbol to HBM (this kernel ran
    // at 1.7 TB/s of its 7.5 GB; MetOp 4.4 ms per step). Only the VIT2_WARM prologue steps of a row's first tile still read the
    // previous block directly.
    constexpr int V2P_STEPS = 2048;              // steps per thread block (256 threads x 8)
    constexpr int V2P_LDS = 2 * V2P_STEPS + 64;  // staged bytes: rate 1/2 needs 2 per step (+ shift, + alignment slack)
    // MODE / PHASE >= 0: that value of the decoder configuration at compile time (the kernel tests the run-uniform mode and phase per soft pair, nine pairs a
    // thread: scalar compares and branches were as many as its vector instructions); -1: read from the configuration, as before -- for anything the host does
    // not dispatch. Same code, same results.
    template <int MODE, int PHASE>
    __global__ __launch_bounds__(256) void k_vit2_prep(VitCfg c_in, const int8_t *__restrict__ soft, long long first_block, int nblk, unsigned short *symu, int SU)
    {
        VitCfg c = c_in;
        if constexpr (MODE >= 0)
            c.mode = MODE;
        if constexpr (PHASE >= 0)
            c.phase = PHASE;
        // thread = 8 consecutive steps of one row (one 16-byte store); one thread block per row, looping over its tiles of 2048 steps
        // (a block per tile -- 700 k blocks of 4 KB each on a MetOp batch -- was bound by the rate blocks can be dispatched at)
        __shared__ __attribute__((aligned(16))) unsigned char stage[V2P_LDS];
        const int groups = SU / 8, tiles = (groups + 255) / 256;
        const int j = (int)blockIdx.x;
        if (j >= nblk)
            return;
        const TailErasure erasure;
        const int nsteps = c.F + 6;
        const int8_t *cur = soft + (first_block + j) * vit_stride(c);
        auto do_tile = [&](const int tile) {
        const int gi = tile * (int)blockDim.x + (int)threadIdx.x;
        // steps of block j this tile covers, and the byte range SymFetch will touch for them
        const int r0 = tile * V2P_STEPS, r1 = r0 + V2P_STEPS;
        int t0 = r0 - VIT2_WARM, t1 = r1 - VIT2_WARM;
        t0 = t0 < 0 ? 0 : t0;
        t1 = t1 > nsteps ? nsteps : t1;
        int lo = 0, hi = 0;
        if (t1 > t0)
        {
            if (c.mode == 1)
            {
                lo = 4 * (t0 / 3);
                hi = 4 * ((t1 - 1) / 3) + 4;
            }
            else
            {
                const int sh = c.mode == 0 ? c.shift : 0;
                lo = (sh + 2 * t0) & ~1;
                hi = ((sh + 2 * (t1 - 1) + 1) | 1) + 1;
            }
            hi = hi > c.B ? c.B : hi;
        }
        const int a = (int)(reinterpret_cast<uintptr_t>(cur + lo) & 3u); // byte alignment of the range in memory
        if (hi > lo)
        {
            const int8_t *ab = cur + lo - a; // dword aligned
            const int ndw = (a + (hi - lo) + 3) / 4;
            for (int d = (int)threadIdx.x; d < ndw; d += 256)
            {
                const int first = lo - a + 4 * d; // byte index within the block of this dword's first byte
                unsigned w;
                if (first >= 0 && first + 4 <= c.B)
                    w = *reinterpret_cast<const unsigned *>(ab + 4 * d);
                else
                {
                    w = 0;
                    for (int q = 0; q < 4; q++)
                        if (first + q >= 0 && first + q < c.B)
                            w |= (unsigned)(unsigned char)cur[first + q] << (8 * q);
                }
                *reinterpret_cast<unsigned *>(stage + 4 * d) = w;
            }
        }
        __syncthreads();
        if (gi >= groups)
            return;
        const SymFetch pf{c, soft + (first_block + j - 1) * vit_stride(c), c.B};
        const SymFetch f{c, reinterpret_cast<const int8_t *>(stage) + a - lo, c.B}; // f.blk[i] = byte i of block j for lo <= i < hi
        const bool have_prev = first_block + j > 0;
        unsigned v[8];
        // Fast path (rate 1/2, the thread's eight steps wholly inside this block's staged bytes): the 16 or 18 soft bytes -- eight I/Q
        // pairs, nine when the BPSK shift makes a step straddle two pairs -- come out of LDS as six dwords realigned to the first
        // pair, and rotation / swap / offset-binary conversion (SymFetch::u_at) run on registers. The generic path below issues four
        // byte reads per step and carries every mode's code.
        {
            const int rt = gi * 8;
            const int j0 = c.shift + 2 * (rt - VIT2_WARM), jb = j0 & ~1, odd = c.shift & 1;
            if (c.mode == 0 && rt >= VIT2_WARM && rt + 8 <= VIT2_WARM + nsteps && jb >= lo && jb + 16 + 2 * odd <= hi)
            {
                const int ob = jb - lo + a;
                const unsigned *w32 = reinterpret_cast<const unsigned *>(stage) + (ob >> 2);
                const unsigned sh = 8u * (unsigned)(ob & 3);
                unsigned w[6], d[5];
#pragma unroll
                for (int i = 0; i < 6; i++)
                    w[i] = w32[i];
#pragma unroll
                for (int i = 0; i < 5; i++)
                    d[i] = (unsigned)(((((unsigned long long)w[i + 1]) << 32) | w[i]) >> sh);
                const bool swap = (c.pre_swap != 0) != (c.iq_swap != 0); // two swaps cancel
                unsigned ua[9], ub[9];
#pragma unroll
                for (int pi = 0; pi < 9; pi++)
                {
                    const unsigned word = d[pi >> 1] >> (16 * (pi & 1));
                    int av = (int)(signed char)(word & 0xffu), bv = (int)(signed char)((word >> 8) & 0xffu);
                    av = av == -128 ? -127 : av;
                    bv = bv == -128 ? -127 : bv;
                    if (swap)
                    {
                        const int t = av;
                        av = bv;
                        bv = t;
                    }
                    if (c.phase == 1)
                    {
                        const int t = av;
                        av = bv;
                        bv = -t;
                    }
                    else if (c.phase == 2)
                    {
                        av = -av;
                        bv = -bv;
                    }
                    else if (c.phase == 3)
                    {
                        const int t = av;
                        av = -bv;
                        bv = t;
                    }
                    unsigned x = (unsigned)(av + 127) & 255u, y = (unsigned)(bv + 127) & 255u;
                    ua[pi] = x == 128u ? 127u : x;
                    ub[pi] = y == 128u ? 127u : y;
                }
#pragma unroll
                for (int q = 0; q < 8; q++)
                    v[q] = odd ? (ub[q] | (ua[q + 1] << 8)) : (ua[q] | (ub[q] << 8));
                uint4 o;
                o.x = v[0] | (v[1] << 16);
                o.y = v[2] | (v[3] << 16);
                o.z = v[4] | (v[5] << 16);
                o.w = v[6] | (v[7] << 16);
                *reinterpret_cast<uint4 *>(symu + (size_t)j * SU + (size_t)gi * 8) = o;
                return;
            }
        }
        // The same for MetOp's rate 3/4 (SymFetch::pair, mode 1): three steps consume four soft bytes (one full pair, then one symbol
        // of the next pair with the other erased, then its partner), so eight steps touch the 12 or 16 bytes of three or four groups.
        // Which of the three positions a thread's first step has decides the whole pattern: three compile-time variants.
        {
            const int rt = gi * 8, t0s = rt - VIT2_WARM;
            if (c.mode == 1 && rt >= VIT2_WARM && rt + 8 <= VIT2_WARM + nsteps)
            {
                const int m0 = t0s / 3, r0 = t0s - 3 * m0, jb = 4 * m0;
                if (jb >= lo && 4 * ((t0s + 7) / 3) + 4 <= hi)
                {
                    const int ob = jb - lo + a;
                    const unsigned *w32 = reinterpret_cast<const unsigned *>(stage) + (ob >> 2);
                    const unsigned sh = 8u * (unsigned)(ob & 3);
                    unsigned w[5], d[4];
#pragma unroll
                    for (int i = 0; i < 5; i++)
                        w[i] = w32[i];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        d[i] = (unsigned)(((((unsigned long long)w[i + 1]) << 32) | w[i]) >> sh);
                    const bool swap = (c.pre_swap != 0) != (c.iq_swap != 0);
                    unsigned ua[8], ub[8]; // pair 2g = bytes (4g, 4g + 1), pair 2g + 1 = bytes (4g + 2, 4g + 3) of group g
#pragma unroll
                    for (int pi = 0; pi < 8; pi++)
                    {
                        const unsigned word = d[pi >> 1] >> (16 * (pi & 1));
                        int av = (int)(signed char)(word & 0xffu), bv = (int)(signed char)((word >> 8) & 0xffu);
                        av = av == -128 ? -127 : av;
                        bv = bv == -128 ? -127 : bv;
                        if (swap)
                        {
                            const int t = av;
                            av = bv;
                            bv = t;
                        }
                        if (c.phase == 1)
                        {
                            const int t = av;
                            av = bv;
                            bv = -t;
                        }
                        else if (c.phase == 2)
                        {
                            av = -av;
                            bv = -bv;
                        }
                        else if (c.phase == 3)
                        {
                            const int t = av;
                            av = -bv;
                            bv = t;
                        }
                        unsigned x = (unsigned)(av + 127) & 255u, y = (unsigned)(bv + 127) & 255u;
                        ua[pi] = x == 128u ? 127u : x;
                        ub[pi] = y == 128u ? 127u : y;
                    }
                    const bool sh0 = c.shift == 0;
                    if (c.fy)
                    { // FengYun keeps the punctured pair's order: exchanging the two bytes of those pairs here gives the MetOp pattern below that
#pragma unroll
                        for (int pi = 0; pi < 8; pi++)
                            if ((pi & 1) == (sh0 ? 1 : 0))
                            {
                                const unsigned t = ua[pi];
                                ua[pi] = ub[pi];
                                ub[pi] = t;
                            }
                    }
                    auto emit = [&](auto r0c) {
                        constexpr int R0 = decltype(r0c)::value;
#pragma unroll
                        for (int q = 0; q < 8; q++)
                        {
                            const int mm = (R0 + q) / 3, r = (R0 + q) % 3;
                            unsigned s0, s1;
                            if (sh0)
                            {
                                s0 = r == 0 ? ua[2 * mm] : (r == 1 ? 128u : ua[2 * mm + 1]);
                                s1 = r == 0 ? ub[2 * mm] : (r == 1 ? ub[2 * mm + 1] : 128u);
                            }
                            else
                            {
                                s0 = r == 0 ? 128u : (r == 1 ? ua[2 * mm] : ua[2 * mm + 1]);
                                s1 = r == 0 ? ub[2 * mm] : (r == 1 ? 128u : ub[2 * mm + 1]);
                            }
                            v[q] = s0 | (s1 << 8);
                        }
                    };
                    if (r0 == 0)
                        emit(std::integral_constant<int, 0>{});
                    else if (r0 == 1)
                        emit(std::integral_constant<int, 1>{});
                    else
                        emit(std::integral_constant<int, 2>{});
                    uint4 o;
                    o.x = v[0] | (v[1] << 16);
                    o.y = v[2] | (v[3] << 16);
                    o.z = v[4] | (v[5] << 16);
                    o.w = v[6] | (v[7] << 16);
                    *reinterpret_cast<uint4 *>(symu + (size_t)j * SU + (size_t)gi * 8) = o;
                    return;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; q++)
        {
            const int r = gi * 8 + q;
            unsigned x = 128u | (128u << 8);
            if (r < VIT2_WARM)
            {
                if (have_prev)
                    x = pf.pair(nsteps - VIT2_WARM + r, erasure);
            }
            else if (r < VIT2_WARM + nsteps)
                x = f.pair(r - VIT2_WARM, erasure);
            v[q] = x & 0xFFFFu;
        }
        uint4 o;
        o.x = v[0] | (v[1] << 16);
        o.y = v[2] | (v[3] << 16);
        o.z = v[4] | (v[5] << 16);
        o.w = v[6] | (v[7] << 16);
        *reinterpret_cast<uint4 *>(symu + (size_t)j * SU + (size_t)gi * 8) = o;
        };
        for (int tile = 0; tile < tiles; tile++)
        {
            do_tile(tile);
            __syncthreads(); // the next tile's staging overwrites the bytes this one read
        }
    }

    // ---- the same symbol pairs formed in the forward pass's own loads (round 6): k_vit2_prep exists only to materialise the rotated / exchanged /
    // depunctured stream -- 2 bytes per trellis step written and read again, 4.3 GB and 2.3 ms of a MetOp step -- and the forward pass's lanes each stream through
    // their own segment anyway. V2Fetch gives a lane the eight pairs of row positions [rt, rt + 8) (what symu[j][rt ..] holds: rt < VIT2_WARM = the last steps of
    // block j - 1, then steps rt - VIT2_WARM of block j, then erasures) from the soft bytes themselves: load() issues the (at most six) dword loads that contain the
    // group's 12 .. 18 soft bytes -- one group ahead of the arithmetic, like the row loads they replace --, decode() turns them into pairs on registers: the
    // arithmetic of k_vit2_prep's fast paths, with the depuncture phase (step mod 3) a per-lane value. A group that touches a block's tail steps, the first block's
    // missing predecessor or the last 8 bytes of a block (the dwords loaded must lie inside it) is evaluated step by step through SymFetch::pair like the generic
    // path of k_vit2_prep: a handful of groups per lane.
    template <int MODE, int PHASE>
    struct V2Fetch
    {
        VitCfg c;
        const int8_t *cur, *prev; // block j, block j - 1 (nullptr: the stream starts with block j)
        int nsteps;
        struct Raw
        {
            unsigned w[6];
            int rt, t0, sh; // row position; first step within its block; bit shift that aligns the first soft byte; sh < 0: not a fast group
        };
        __device__ __forceinline__ void init(const VitCfg &c_in, const int8_t *soft, long long first_block, int j)
        {
            c = c_in;
            if constexpr (MODE >= 0)
                c.mode = MODE;
            if constexpr (PHASE >= 0)
                c.phase = PHASE;
            nsteps = c.F + 6;
            cur = soft + (first_block + j) * vit_stride(c);
            prev = (first_block + j > 0) ? cur - vit_stride(c) : nullptr;
        }
        __device__ __forceinline__ Raw load(int rt) const
        { // (rt and VIT2_WARM are multiples of 8: a group lies in ONE block)
            Raw r;
#pragma unroll
            for (int i = 0; i < 6; i++)
                r.w[i] = 0u;
            r.rt = rt;
            r.sh = -1;
            const int8_t *blk = rt >= VIT2_WARM ? cur : prev;
            const int t0 = rt >= VIT2_WARM ? rt - VIT2_WARM : nsteps - VIT2_WARM + rt;
            r.t0 = t0;
            if (blk == nullptr || t0 < 0 || t0 + 8 > nsteps)
                return r;
            int jb, need;
            if (c.mode == 0)
            {
                jb = (c.shift + 2 * t0) & ~1;
                need = 16 + 2 * (c.shift & 1);
            }
            else if (c.mode == 1)
            {
                jb = 4 * (t0 / 3);
                need = 4 * ((t0 + 7) / 3) + 4 - jb;
            }
            else
                return r;
            if (jb + need + 8 > c.B)
                return r;
            const int8_t *p = blk + jb;
            const unsigned a = (unsigned)(reinterpret_cast<uintptr_t>(p) & 3u);
            const unsigned *w32 = reinterpret_cast<const unsigned *>(p - a);
#pragma unroll
            for (int i = 0; i < 6; i++)
                r.w[i] = w32[i];
            r.sh = (int)(8u * a);
            return r;
        }
        // Four soft bytes -- two (I, Q) pairs as they lie in memory -- to the four unsigned symbols SymFetch::u_at gives, all four at once: rotate_soft's -128 -> -127,
        // the exchange (pre_swap xor iq_swap) and the quarter turns (an exchange for the odd ones, some bytes negated), signed_soft_to_unsigned's + 127 with 128
        // (reserved for an erasure) -> 127. Byte-parallel in a dword: "equal to a constant" is a zero-byte test ((y & 0x7f..) + 0x7f.. | y has bit 7 clear exactly in
        // the zero bytes of y); s + 127 = (s ^ 0x80) - 1 and 127 - s = s ^ 0x7f bytewise, and the one subtraction cannot borrow across bytes (its bytes are >= 1
        // where 1 is subtracted: s >= -127 by then). 17 instructions for four symbols; a pair at a time took ~20 for two.
        __device__ __forceinline__ static unsigned eq80(unsigned x)
        { // 1 in the bytes of x that are 0x80, 0 elsewhere
            const unsigned y = x ^ 0x80808080u;
            const unsigned z = ((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y;
            return (~z & 0x80808080u) >> 7;
        }
        __device__ __forceinline__ unsigned conv4(unsigned x, bool swap) const
        {
            x |= eq80(x); // -128 -> -127
            const bool ex = swap != ((c.phase & 1) != 0);
            if (ex)
                x = __builtin_amdgcn_perm(0u, x, 0x02030001u); // (a, b) -> (b, a) in both pairs
            // bytes negated by the turn: 90 deg (a, b) -> (b, -a): the second of a pair; 180: both; 270 (a, b) -> (-b, a): the first
            const unsigned ng = c.phase == 1 ? 0xFF00FF00u : (c.phase == 2 ? 0xFFFFFFFFu : (c.phase == 3 ? 0x00FF00FFu : 0u));
            unsigned u = (x ^ (0x80808080u ^ ng)) - (0x01010101u & ~ng);
            u -= eq80(u); // 128 -> 127
            return u;
        }
        __device__ __forceinline__ uint4 decode(const Raw &r) const
        {
            const bool swap = (c.pre_swap != 0) != (c.iq_swap != 0); // two swaps cancel
            if (r.sh >= 0 && c.mode == 0)
            { // rate 1/2: the symbols of eight steps are 16 consecutive converted bytes, from the pair boundary or (BPSK's odd shift) one byte on
                unsigned d[5];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    d[i] = conv4((unsigned)(((((unsigned long long)r.w[i + 1]) << 32) | r.w[i]) >> r.sh), swap);
                if (!(c.shift & 1))
                    return make_uint4(d[0], d[1], d[2], d[3]);
                d[4] = conv4((unsigned)(((((unsigned long long)r.w[5]) << 32) | r.w[4]) >> r.sh), swap);
                uint4 o;
                o.x = (d[0] >> 8) | (d[1] << 24);
                o.y = (d[1] >> 8) | (d[2] << 24);
                o.z = (d[2] >> 8) | (d[3] << 24);
                o.w = (d[3] >> 8) | (d[4] << 24);
                return o;
            }
            else if (r.sh >= 0)
            { // MetOp / FengYun rate 3/4 (SymFetch::pair, mode 1): four soft bytes (A0 B0 A1 B1) make three steps = six symbols -- shift 0: A0 B0 | E B1 | A1 E, shift 1:
              // E B0 | A0 E | A1 B1 (E = 128, the punctured symbol; FengYun keeps the punctured pair's order: B1 and A1, resp. B0 and A0, exchanged). The four groups a
              // lane's eight steps can touch are expanded to their 24 symbols by v_perm (the same for every lane), and the lane takes its 16 from symbol 2 x (first step mod 3) on.
                unsigned gq[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    gq[i] = conv4((unsigned)(((((unsigned long long)r.w[i + 1]) << 32) | r.w[i]) >> r.sh), swap);
                const bool sh0 = c.shift == 0, fy = c.fy != 0;
                // selectors of the three dwords two groups expand to (bytes 0-3: the first group, 4-7: the second, 0x0c: zero -- the erasure is or-ed in)
                const unsigned s0 = sh0 ? (fy ? 0x020c0100u : 0x030c0100u) : (fy ? 0x0c01000cu : 0x0c00010cu);
                const unsigned s1 = sh0 ? (fy ? 0x05040c03u : 0x05040c02u) : (fy ? 0x040c0302u : 0x050c0302u);
                const unsigned s2 = sh0 ? (fy ? 0x0c03020cu : 0x0c02030cu) : (fy ? 0x03020c01u : 0x03020c00u);
                const unsigned e0 = sh0 ? 0x00800000u : 0x80000080u, e1 = sh0 ? 0x00008000u : 0x00800000u, e2 = sh0 ? 0x80000080u : 0x00008000u;
                unsigned O[6];
                O[0] = __builtin_amdgcn_perm(0u, gq[0], s0) | e0;
                O[1] = __builtin_amdgcn_perm(gq[1], gq[0], s1) | e1;
                O[2] = __builtin_amdgcn_perm(0u, gq[1], s2) | e2;
                O[3] = __builtin_amdgcn_perm(0u, gq[2], s0) | e0;
                O[4] = __builtin_amdgcn_perm(gq[3], gq[2], s1) | e1;
                O[5] = __builtin_amdgcn_perm(0u, gq[3], s2) | e2;
                const int r0 = r.t0 % 3;
                unsigned B[5];
#pragma unroll
                for (int k = 0; k < 5; k++)
                    B[k] = r0 == 2 ? O[k + 1] : O[k];
                const unsigned bs = r0 == 1 ? 16u : 0u;
                uint4 o;
                o.x = (unsigned)(((((unsigned long long)B[1]) << 32) | B[0]) >> bs);
                o.y = (unsigned)(((((unsigned long long)B[2]) << 32) | B[1]) >> bs);
                o.z = (unsigned)(((((unsigned long long)B[3]) << 32) | B[2]) >> bs);
                o.w = (unsigned)(((((unsigned long long)B[4]) << 32) | B[3]) >> bs);
                return o;
            }
            else
            { // (a loop, not eight copies of SymFetch::pair, and no indexed array: the pairs are or-ed into the four words)
                const SymFetch pf{c, prev, c.B}, f{c, cur, c.B};
                const TailErasure erasure;
                unsigned o0 = 0u, o1 = 0u, o2 = 0u, o3 = 0u;
#pragma unroll 1
                for (int q = 0; q < 8; q++)
                {
                    const int rr = r.rt + q;
                    unsigned x = 128u | (128u << 8);
                    if (rr < VIT2_WARM)
                    {
                        if (prev)
                            x = pf.pair(nsteps - VIT2_WARM + rr, erasure);
                    }
                    else if (rr < VIT2_WARM + nsteps)
                        x = f.pair(rr - VIT2_WARM, erasure);
                    const unsigned xs = (x & 0xFFFFu) << (16 * (q & 1));
                    const int k = q >> 1;
                    o0 |= k == 0 ? xs : 0u;
                    o1 |= k == 1 ? xs : 0u;
                    o2 |= k == 2 ? xs : 0u;
                    o3 |= k == 3 ? xs : 0u;
                }
                return make_uint4(o0, o1, o2, o3);
            }
        }
    };

    // ---- forward pass: one lane per (block, segment)
    template <bool DEC>
    __device__ __forceinline__ void v2_group(V2State &s, const V2Consts &kc, const uint4 q, unsigned (&w)[8][2])
    {
        const unsigned d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const unsigned sym = (d[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            v2_step<DEC>(s, kc, sym, w[k][0], w[k][1]);
        }
    }

    #ifndef V2_WAVES
#define V2_WAVES 2
#endif
    __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(V2_WAVES, V2_WAVES))) void k_vit2_acs(int F, int S, int NSEG, int nblk, const unsigned short *__restrict__ symu, int SU, VitBlockIO *io,
                                                      unsigned long long *dec, long long U64, unsigned *specx, unsigned *endx)
    {
        const int u = (int)(blockIdx.x * 64 + threadIdx.x);
        if (u >= nblk * NSEG)
            return;
        const int j = u / NSEG, g = u - j * NSEG;
        const uint4 *row = reinterpret_cast<const uint4 *>(symu + (size_t)j * SU + (size_t)g * S); // 8 steps per uint4
        V2State s;
        v2_init_neutral(s);
        V2Consts kc;
        v2_consts(kc);
        unsigned w[8][2];
        // ---- warm-up over the VIT2_WARM steps in front of the segment
        constexpr int WG = VIT2_WARM / 8;
        uint4 q = row[0];
        for (int grp = 0; grp < WG - 1; grp++)
        {
            const uint4 nq = row[grp + 1];
            v2_group<false>(s, kc, q, w);
            q = nq;
        }
        {
            const uint4 nq = row[WG];
            v2_group<true>(s, kc, q, w); // the last 6 decisions are the chained start state's chainback (g == 0)
            q = nq;
        }
        if (g == 0)
        {
            int start = io[j].start_in;
            if (start == -1)
            {
                // state 6 steps before the end state of the previous block (cc_decoder.cpp:250-260, 295-302)
                unsigned st = v2_endstate(s);
#pragma unroll
                for (int k = 7; k >= 2; k--)
                {
                    const unsigned kb = v2_dec_bit(w[k][0], w[k][1], st);
                    st = (st >> 1) | (kb << 5);
                }
                start = (int)st;
            }
            v2_init_start(s, start);
            io[j].start_used = start;
        }
        else
        {
#pragma unroll
            for (int r = 0; r < 32; r++)
                specx[(size_t)u * 32 + r] = v2_norm(s, r);
        }
        // ---- the segment itself: S steps (the last segment of a block: S + 6)
        unsigned long long *d = dec + u;
        for (int grp = 0; grp < S / 8; grp++)
        {
            const uint4 nq = row[WG + grp + 1];
            v2_group<true>(s, kc, q, w);
            q = nq;
#pragma unroll
            for (int k = 0; k < 8; k++)
                d[(size_t)(grp * 8 + k) * U64] = (unsigned long long)w[k][0] | ((unsigned long long)w[k][1] << 32);
        }
        if (g == NSEG - 1)
        {
            unsigned wl[6][2];
            const unsigned dd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 6; k++)
            {
                const unsigned sym = (dd[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
                v2_step<true>(s, kc, sym, wl[k][0], wl[k][1]);
                d[(size_t)(S + k) * U64] = (unsigned long long)wl[k][0] | ((unsigned long long)wl[k][1] << 32);
            }
            const unsigned endstate = v2_endstate(s);
            unsigned ret = endstate; // chained start state for the next block: state after steps F+5..F (cc_decoder.cpp:250-260,275)
#pragma unroll
            for (int k = 5; k >= 0; k--)
            {
                const unsigned kb = v2_dec_bit(wl[k][0], wl[k][1], ret);
                ret = (ret >> 1) | (kb << 5);
            }
            io[j].end_state = (int)endstate;
            io[j].ret_state = (int)ret;
        }
        else
        {
#pragma unroll
            for (int r = 0; r < 32; r++)
                endx[(size_t)u * 32 + r] = v2_norm(s, r);
        }
    }

    // ---- traceback: one lane per (block, segment); bits [g*S, (g+1)*S) <- decisions at steps n + 6
    __global__ __launch_bounds__(64) void k_vit2_tb(int F, int S, int NSEG, int nblk, const VitBlockIO *io, const unsigned long long *__restrict__ dec, long long U64,
                                                     unsigned *vbits, int wpb, int *entry, int *exitst)
    {
        const int u = (int)(blockIdx.x * 64 + threadIdx.x);
        if (u >= nblk * NSEG)
            return;
        const int j = u / NSEG, g = u - j * NSEG;
        const int t_hi = (g + 1) * S + 5;
        int tstart = t_hi + VIT_TB_OVERLAP;
        if (tstart > F + 5)
            tstart = F + 5;
        unsigned st = (tstart == F + 5) ? (unsigned)io[j].end_state : 0u;
        const long long ubase = (long long)j * NSEG;
        auto rec = [&](int t) -> unsigned long long {
            int gg = t / S;
            if (gg > NSEG - 1)
                gg = NSEG - 1;
            return dec[(size_t)(t - gg * S) * U64 + ubase + gg];
        };
        for (int t = tstart; t > t_hi; t--) // overlap: converge onto the survivor path
        {
            const unsigned long long r = rec(t);
            const unsigned kb = v2_dec_bit((unsigned)r, (unsigned)(r >> 32), st);
            st = (st >> 1) | (kb << 5);
        }
        entry[u] = (int)st;
        unsigned *vb = vbits + (size_t)j * wpb + (size_t)g * (S / 32);
        for (int wd = S / 32 - 1; wd >= 0; wd--)
        {
            unsigned long long r[32];
            const int tb = g * S + wd * 32 + 6;
#pragma unroll
            for (int b = 0; b < 32; b++)
                r[b] = rec(tb + b);
            unsigned word = 0;
#pragma unroll
            for (int b = 31; b >= 0; b--)
            {
                const unsigned kb = v2_dec_bit((unsigned)r[b], (unsigned)(r[b] >> 32), st);
                st = (st >> 1) | (kb << 5);
                word |= kb << (31 - b);
            }
            vb[wd] = word;
        }
        exitst[u] = (int)st;
    }

    // ---- the same lane-per-segment decoder with PATH HISTORIES instead of per-step decision words (k_vit2h_*) -------------------------------------
    // The tag that breaks a tie and names the winner of a compare-select (above) does not have to be put back on the metric after every step: within a
    // group of 8 steps, step k carries it in BIT k of the low byte -- on the branch constants (K | 1 << k for the two candidates that must lose a tie),
    // not on the registers -- and the low byte of a surviving metric is left alone. The bits of earlier steps sit below bit k, so exactly one of two
    // candidates has bit k set and the 16-bit minimum still resolves ties the reference's way; and after 8 steps the low byte of state s holds the 8
    // (inverted) decisions along the path that SURVIVES in s. That byte is all a traceback needs: 8 decoded bits at a time, and the state 8 steps
    // earlier is the bit reversal of the first six of them (each step back shifts the state right and enters the decision at bit 5, cc_decoder.cpp:
    // 250-260). Per step this drops the two v_and_or (tag back on), the v_perm and the v_lshl_or (decision words) of every butterfly pair -- 13 -> 9
    // instructions, + 16 v_or per step for the tagged constants, + 48 per group to gather and clear the bytes -- and the traceback makes one dependent
    // byte fetch per 8 steps. Storage is what it was: 64 bytes per 8 steps and lane, as four uint4 planes [group][v][unit].
    __device__ __forceinline__ void v2h_init_neutral(V2State &s)
    {
#pragma unroll
        for (int r = 0; r < 32; r++)
            s.R[r] = 0u;
        s.C2 = 0u;
    }
    __device__ __forceinline__ void v2h_init_start(V2State &s, int start)
    { // init_viterbi / init_viterbi_unbiased, cc_decoder.cpp:159-190
        const unsigned all = ((start == -2) ? 31u : 63u) << 8;
#pragma unroll
        for (int r = 0; r < 32; r++)
        {
            unsigned lo = all, hi = all;
            if (start == r)
                lo = 0u;
            if (start == 63 - r)
                hi = 0u;
            s.R[r] = lo | (hi << 16);
        }
        s.C2 = 0u;
    }
    // one trellis step, the KSTEP-th of its group; DEC = false: no tags (the low bytes stay as they are -- clear, in a warm-up)
    template <int KSTEP, bool DEC>
    __device__ __forceinline__ void v2h_step(V2State &s, unsigned sym)
    {
        const unsigned s0 = sym & 255u, s1 = (sym >> 8) & 255u;
        const unsigned a[2] = {s0, s0 ^ 255u}, c[2] = {s1, s1 ^ 255u};
        unsigned D[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            D[k] = ((a[k >> 1] + c[k & 1] + 1u) << 5) & 0x3F00u; // ((sum + 1) >> 3) << 8
        const unsigned Q = pk_sub(0x3F003F00u, s.C2);
        constexpr unsigned TL = DEC ? (1u << KSTEP) : 0u, TH = TL << 16;
        // candidates (see the register map above): T1 = (m0, m3'), T2 = (m1, m2'), T3 = (m2, m1'), T4 = (m3, m0'); a tie goes to m1 / m3 / m3' / m1'
        // (decision = (m0 - m1) >= 0 picks the second), so m0, m2 (low halves of T1, T3) and m2', m0' (high halves of T2, T4) carry the tag
        unsigned K1a[4], K1b[4], K2a[4], K2b[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const unsigned PD = D[k] | (D[3 - k] << 16);
            const unsigned K1 = pk_sub(PD, s.C2), K2 = pk_sub(Q, PD);
            K1a[k] = K1 | TL;
            K1b[k] = K1 | TH;
            K2a[k] = K2 | TH;
            K2b[k] = K2 | TL;
        }
        unsigned Rn[32];
        unsigned mnA = 0xFFFFFFFFu, mnB = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            const int b0 = (i ^ (i >> 1) ^ (i >> 2)) & 1, b1 = ((i >> 1) ^ (i >> 2) ^ (i >> 4)) & 1;
            const int k = b0 * 2 + b1;
            const unsigned R1 = s.R[i], R2 = pk_swap(s.R[31 - i]);
            const unsigned E = pk_min(pk_add(R1, K1a[k]), pk_add(R2, K2a[k]));
            const unsigned O = pk_min(pk_add(R1, K2b[k]), pk_add(R2, K1b[k]));
            Rn[2 * i] = E;
            Rn[2 * i + 1] = O;
            mnA = pk_min(mnA, E);
            mnB = pk_min(mnB, O);
        }
#pragma unroll
        for (int r = 0; r < 32; r++)
            s.R[r] = Rn[r];
        const unsigned mn = pk_min(mnA, mnB);
        const unsigned m1 = min(mn & 0xFFFFu, mn >> 16) & 0xFF00u;
        s.C2 = m1 | (m1 << 16);
    }
    template <bool DEC>
    __device__ __forceinline__ void v2h_group(V2State &s, const uint4 q)
    {
        v2h_step<0, DEC>(s, q.x & 0xFFFFu);
        v2h_step<1, DEC>(s, q.x >> 16);
        v2h_step<2, DEC>(s, q.y & 0xFFFFu);
        v2h_step<3, DEC>(s, q.y >> 16);
        v2h_step<4, DEC>(s, q.z & 0xFFFFu);
        v2h_step<5, DEC>(s, q.z >> 16);
        v2h_step<6, DEC>(s, q.w & 0xFFFFu);
        v2h_step<7, DEC>(s, q.w >> 16);
    }
    // the 64 history bytes of a group: dword q = states (2q, 63 - 2q, 2q + 1, 62 - 2q), byte 0 first
    __device__ __forceinline__ void v2h_gather(const V2State &s, unsigned (&rec)[16])
    {
#pragma unroll
        for (int q = 0; q < 16; q++)
            rec[q] = __builtin_amdgcn_perm(s.R[2 * q + 1], s.R[2 * q], 0x06040200u);
    }
    __device__ __forceinline__ void v2h_clear(V2State &s)
    {
#pragma unroll
        for (int r = 0; r < 32; r++)
            s.R[r] &= V2_MASK;
    }
    __device__ __forceinline__ void v2h_where(unsigned st, unsigned &q, unsigned &byte)
    {
        const unsigned r = st < 32u ? st : 63u - st;
        q = r >> 1;
        byte = ((r & 1u) << 1) + (st < 32u ? 0u : 1u);
    }
    // decisions along the path that ends in state st, step k of the group at bit k (a register-resident record: once per lane)
    __device__ __forceinline__ unsigned v2h_path(const unsigned (&rec)[16], unsigned st)
    {
        unsigned q, byte, w = 0;
        v2h_where(st, q, byte);
#pragma unroll
        for (int i = 0; i < 16; i++)
            w = (q == (unsigned)i) ? rec[i] : w;
        return ~(w >> (8u * byte)) & 255u;
    }
    __device__ __forceinline__ unsigned v2h_back6(unsigned six) { return __brev(six & 63u) >> 26; } // the state in front of six steps whose decisions are `six` (first step at bit 0)

#ifndef V2H_WAVES
#define V2H_WAVES 2
#endif
    // FUSED: the symbol pairs come from the soft bytes (V2Fetch; MODE / PHASE as for k_vit2_prep); otherwise from the rows k_vit2_prep wrote
    template <int MODE, int PHASE, bool FUSED>
    __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(V2H_WAVES, V2H_WAVES))) void k_vit2h_acs(VitCfg c_in, const int8_t *__restrict__ soft, long long first_block, int F, int S, int NSEG, int nblk,
                                                      const unsigned short *__restrict__ symu, int SU, VitBlockIO *io, uint4 *dec4, long long U64, unsigned *specx, unsigned *endx)
    {
        const int u = (int)(blockIdx.x * 64 + threadIdx.x);
        if (u >= nblk * NSEG)
            return;
        const int j = u / NSEG, g = u - j * NSEG;
        const uint4 *row = FUSED ? nullptr : reinterpret_cast<const uint4 *>(symu + (size_t)j * SU + (size_t)g * S); // 8 steps per uint4
        V2Fetch<MODE, PHASE> fx;
        if constexpr (FUSED)
            fx.init(c_in, soft, first_block, j);
        typedef typename V2Fetch<MODE, PHASE>::Raw Raw;
        const int rbase = g * S;
        // group `grp` of the lane's row: fetched one group ahead of its use (ld), turned into pairs behind the arithmetic in between (dc)
        auto ld = [&](int grp) -> Raw {
            if constexpr (FUSED)
                return fx.load(rbase + 8 * grp);
            else
            {
                const uint4 t = row[grp];
                Raw r;
                r.w[0] = t.x;
                r.w[1] = t.y;
                r.w[2] = t.z;
                r.w[3] = t.w;
                return r;
            }
        };
        auto dc = [&](const Raw &r) -> uint4 {
            if constexpr (FUSED)
            {
                __builtin_amdgcn_sched_barrier(0); // (the conversion stays behind the group it was fetched across)
                return fx.decode(r);
            }
            else
                return make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]);
        };
        V2State s;
        v2h_init_neutral(s);
        unsigned rec[16];
        // ---- warm-up over the VIT2_WARM steps in front of the segment: only its last group keeps histories (the chained start state, g == 0)
        constexpr int WG = VIT2_WARM / 8;
        uint4 q = dc(ld(0));
        for (int grp = 0; grp < WG - 1; grp++)
        {
            const Raw nr = ld(grp + 1);
            v2h_group<false>(s, q);
            q = dc(nr);
        }
        {
            const Raw nr = ld(WG);
            v2h_group<true>(s, q);
            q = dc(nr);
        }
        if (g == 0)
        {
            int start = io[j].start_in;
            if (start == -1)
            { // state 6 steps before the end state of the previous block (cc_decoder.cpp:250-260, 295-302): the decisions of the group's steps 7 .. 2
                v2h_gather(s, rec);
                start = (int)v2h_back6(v2h_path(rec, v2_endstate(s)) >> 2);
            }
            v2h_init_start(s, start);
            io[j].start_used = start;
        }
        else
        {
#pragma unroll
            for (int r = 0; r < 32; r++)
                specx[(size_t)u * 32 + r] = v2_norm(s, r);
            v2h_clear(s);
        }
        // ---- the segment itself: S steps (the last segment of a block: S + 6)
        uint4 *d = dec4 + u;
        for (int grp = 0; grp < S / 8; grp++)
        {
            const Raw nr = ld(WG + grp + 1);
            v2h_group<true>(s, q);
            q = dc(nr);
            v2h_gather(s, rec);
            v2h_clear(s);
#pragma unroll
            for (int v = 0; v < 4; v++)
                d[(size_t)(grp * 4 + v) * U64] = make_uint4(rec[4 * v], rec[4 * v + 1], rec[4 * v + 2], rec[4 * v + 3]);
        }
        if (g == NSEG - 1)
        {
            v2h_step<0, true>(s, q.x & 0xFFFFu);
            v2h_step<1, true>(s, q.x >> 16);
            v2h_step<2, true>(s, q.y & 0xFFFFu);
            v2h_step<3, true>(s, q.y >> 16);
            v2h_step<4, true>(s, q.z & 0xFFFFu);
            v2h_step<5, true>(s, q.z >> 16);
            v2h_gather(s, rec);
#pragma unroll
            for (int v = 0; v < 4; v++)
                d[(size_t)((S / 8) * 4 + v) * U64] = make_uint4(rec[4 * v], rec[4 * v + 1], rec[4 * v + 2], rec[4 * v + 3]);
            const unsigned endstate = v2_endstate(s);
            io[j].end_state = (int)endstate;
            io[j].ret_state = (int)v2h_back6(v2h_path(rec, endstate)); // chained start state for the next block: the state in front of steps F .. F+5 (cc_decoder.cpp:250-260,275)
        }
        else
        {
#pragma unroll
            for (int r = 0; r < 32; r++)
                endx[(size_t)u * 32 + r] = v2_norm(s, r);
        }
    }

    // ---- traceback on the histories: one lane per (block, segment), a group of 8 steps per hop. Group G of a block = steps 8G .. 8G + 7 (the block's
    // last one holds 6); bit n of the block is the decision of step n + 6, so output word w of the segment is steps 32w + 6 .. 32w + 37 of it.
    __global__ __launch_bounds__(64) void k_vit2h_tb(int F, int S, int NSEG, int nblk, const VitBlockIO *io, const uint4 *__restrict__ dec4, long long U64, unsigned *vbits,
                                                      int wpb, int *entry, int *exitst)
    {
        const int u = (int)(blockIdx.x * 64 + threadIdx.x);
        if (u >= nblk * NSEG)
            return;
        const int j = u / NSEG, g = u - j * NSEG;
        const int GPS = S / 8;
        const long long ubase = (long long)j * NSEG;
        auto path = [&](int Gb, unsigned st) -> unsigned { // decisions of group Gb along the path into st (step k at bit k)
            int gg = Gb / GPS;
            if (gg > NSEG - 1)
                gg = NSEG - 1;
            const int ql = Gb - gg * GPS;
            unsigned q, byte;
            v2h_where(st, q, byte);
            const unsigned *p = reinterpret_cast<const unsigned *>(dec4 + ((size_t)(ql * 4 + (int)(q >> 2)) * U64 + ubase + gg));
            return ~(p[q & 3u] >> (8u * byte)) & 255u;
        };
        const int Gtop = (g + 1) * GPS, Glow = g * GPS;
        unsigned st;
        if (g == NSEG - 1)
            st = (unsigned)io[j].end_state; // the state behind step F + 5
        else
        {
            st = 0u; // overlap: converge onto the survivor path
            for (int Gb = Gtop + VIT_TB_OVERLAP / 8; Gb > Gtop; Gb--)
                st = v2h_back6(path(Gb, st));
        }
        entry[u] = (int)st; // the state behind the last step of group Gtop
        unsigned *vb = vbits + (size_t)j * wpb + (size_t)g * (S / 32);
        unsigned long long acc = 0ull; // steps in time order from bit 63 down, the group just walked first
        for (int Gb = Gtop; Gb >= Glow; Gb--)
        {
            if (Gb == Glow)
                exitst[u] = (int)st; // the state behind the last step of group Glow = the previous segment's Gtop
            const unsigned D = path(Gb, st);
            acc = (acc >> 8) | ((unsigned long long)(__brev(D) >> 24) << 56);
            const int ql = Gb - Glow;
            if ((ql & 3) == 0 && ql < GPS)
                vb[ql >> 2] = (unsigned)(acc >> 26);
            st = v2h_back6(D);
        }
    }

    // ---- certificates: segment g's warm-up metrics == segment g-1's end metrics; traceback entry[g] == exit[g+1]
    __global__ __launch_bounds__(64) void k_vit2_cert(int NSEG, int nblk, const unsigned *__restrict__ specx, const unsigned *__restrict__ endx,
                                                       const int *__restrict__ entry, const int *__restrict__ exitst, VitBlockIO *io)
    {
        const int j = (int)(blockIdx.x * 64 + threadIdx.x);
        if (j >= nblk)
            return;
        int fail = 0;
        for (int g = 1; g < NSEG; g++)
        {
            const size_t u = (size_t)j * NSEG + g;
            for (int r = 0; r < 32; r++)
                fail |= specx[u * 32 + r] != endx[(u - 1) * 32 + r];
            fail |= entry[u - 1] != exitst[u];
        }
        io[j].tb_fallback = fail ? 2 : 0;
    }

    void launch_vit_decode2(const VitCfg &cfg, const int8_t *soft, int64_t first_block, int nblk, VitBlockIO *io, uint32_t *vbits, Vit2Work &w, hipStream_t st)
    {
        if (nblk <= 0)
            return;
        // segment length: 512 steps; two or four times that when the batch still fills the chip with >= 2 waves per SIMD (the 200-step
        // warm-up then costs 20 % / 10 % instead of 39 %)
        const int F = cfg.F;
        int S = VIT2_SEG;
        {
            const char *e = getenv("SDHIP_VIT2_SEG");
            if (e && atoi(e) >= 128 && atoi(e) % 32 == 0 && F % atoi(e) == 0 && F / atoi(e) >= 2)
                S = atoi(e);
            else
                for (int m = 4; m >= 2; m >>= 1) // 2048, then 1024 steps per lane (measured, MetOp / NPP 17 GB: 12.8 / 11.2 ms at 1024, 12.5 / 10.7 at 2048, 14.4 at 512)
                    if (F % (m * VIT2_SEG) == 0 && F / (m * VIT2_SEG) >= 2 && (long long)nblk * (F / (m * VIT2_SEG)) >= 2 * 65536)
                    {
                        S = m * VIT2_SEG;
                        break;
                    }
        }
        const int NSEG = F / S;
        const int SU = (VIT2_WARM + F + 6 + 8 + 7) / 8 * 8; // one spare group: the forward pass prefetches 8 steps ahead
        const long long U = (long long)nblk * NSEG, U64 = (U + 63) / 64 * 64;
        w.dec.reserve((size_t)(S + 8) * U64);
        w.specx.reserve((size_t)U * 32);
        w.endx.reserve((size_t)U * 32);
        w.entry.reserve((size_t)U);
        w.exitst.reserve((size_t)U);
        const int wpb = vit_words_per_block(F);
        const bool hist = !(getenv("SDHIP_VIT2_HIST") && atoi(getenv("SDHIP_VIT2_HIST")) == 0); // A/B switch: 0 = per-step decision words (k_vit2_acs / k_vit2_tb)
        const bool fused = hist && !(getenv("SDHIP_VIT2_FUSED") && atoi(getenv("SDHIP_VIT2_FUSED")) == 0); // A/B switch: 0 = k_vit2_prep writes the rows first
        if (!fused)
        {
            w.symu.reserve((size_t)nblk * SU + 64);
            ProfScope _ps("k_vit2_prep", st);
            auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), 0, st, cfg, soft, (long long)first_block, nblk, w.symu.p, SU); };
            static const bool templ = !(getenv("SDHIP_VIT2_PREP_TEMPL") && atoi(getenv("SDHIP_VIT2_PREP_TEMPL")) == 0);
            const int key = (templ && (cfg.mode == 0 || cfg.mode == 1) && cfg.phase >= 0 && cfg.phase <= 3) ? cfg.mode * 4 + cfg.phase : -1;
            switch (key)
            {
            case 0: go(k_vit2_prep<0, 0>); break;
            case 1: go(k_vit2_prep<0, 1>); break;
            case 2: go(k_vit2_prep<0, 2>); break;
            case 3: go(k_vit2_prep<0, 3>); break;
            case 4: go(k_vit2_prep<1, 0>); break;
            case 5: go(k_vit2_prep<1, 1>); break;
            case 6: go(k_vit2_prep<1, 2>); break;
            case 7: go(k_vit2_prep<1, 3>); break;
            default: go(k_vit2_prep<-1, -1>); break;
            }
        }
        if (hist)
        {
            {
                ProfScope _ps("k_vit2_acs", st);
                auto go = [&](auto kern) {
                    hipLaunchKernelGGL(kern, dim3((unsigned)(U64 / 64)), dim3(64), 0, st, cfg, soft, (long long)first_block, F, S, NSEG, nblk, w.symu.p, SU, io, (uint4 *)w.dec.p, U64, w.specx.p,
                                       w.endx.p);
                };
                const int key = !fused ? -2 : ((cfg.mode == 0 || cfg.mode == 1) && cfg.phase >= 0 && cfg.phase <= 3) ? cfg.mode * 4 + cfg.phase : -1;
                switch (key)
                {
                case 0: go(k_vit2h_acs<0, 0, true>); break;
                case 1: go(k_vit2h_acs<0, 1, true>); break;
                case 2: go(k_vit2h_acs<0, 2, true>); break;
                case 3: go(k_vit2h_acs<0, 3, true>); break;
                case 4: go(k_vit2h_acs<1, 0, true>); break;
                case 5: go(k_vit2h_acs<1, 1, true>); break;
                case 6: go(k_vit2h_acs<1, 2, true>); break;
                case 7: go(k_vit2h_acs<1, 3, true>); break;
                case -1: go(k_vit2h_acs<-1, -1, true>); break;
                default: go(k_vit2h_acs<-1, -1, false>); break;
                }
            }
            {
                ProfScope _ps("k_vit2_tb", st);
                hipLaunchKernelGGL(k_vit2h_tb, dim3((unsigned)(U64 / 64)), dim3(64), 0, st, F, S, NSEG, nblk, io, (const uint4 *)w.dec.p, U64, vbits, wpb, w.entry.p, w.exitst.p);
            }
        }
        else
        {
            {
                ProfScope _ps("k_vit2_acs", st);
                hipLaunchKernelGGL(k_vit2_acs, dim3((unsigned)(U64 / 64)), dim3(64), 0, st, F, S, NSEG, nblk, w.symu.p, SU, io, (unsigned long long *)w.dec.p, U64,
                                   w.specx.p, w.endx.p);
            }
            {
                ProfScope _ps("k_vit2_tb", st);
                hipLaunchKernelGGL(k_vit2_tb, dim3((unsigned)(U64 / 64)), dim3(64), 0, st, F, S, NSEG, nblk, io, (const unsigned long long *)w.dec.p, U64, vbits, wpb,
                                   w.entry.p, w.exitst.p);
            }
        }
        {
            ProfScope _ps("k_vit2_cert", st);
            hipLaunchKernelGGL(k_vit2_cert, dim3((nblk + 63) / 64), dim3(64), 0, st, NSEG, nblk, w.specx.p, w.endx.p, w.entry.p, w.exitst.p, io);
        }
    }

    // =============================================================================================
    // packed-bit helpers: bit n of a block lives in word n>>5 at position 31-(n&31)
    // =============================================================================================
    __device__ __forceinline__ unsigned getbit(const unsigned *w, int n) { return (w[n >> 5] >> (31 - (n & 31))) & 1u; }

    // up to 32 bits starting at bit `off` (MSB-first), n in [1,32]; may read word (off>>5)+1
    __device__ __forceinline__ unsigned peek_bits(const unsigned *w, long long off, int n)
    {
        const long long wi = off >> 5;
        const int sh = (int)(off & 31);
        const unsigned long long v = ((unsigned long long)w[wi] << 32) | (unsigned long long)w[wi + 1];
        return (unsigned)((v << sh) >> (64 - n));
    }

    // =============================================================================================
    // k_vit_ber: re-encode (cc_encoder.cpp:92-104) + get_ber (viterbi_1_2.cpp:37-50)
    // =============================================================================================
    __global__ __launch_bounds__(256) void k_vit_ber(VitCfg c, const int8_t *__restrict__ soft, long long first_block, int nblk, const unsigned *__restrict__ vbits,
                                                      int wpb, unsigned enc_state_in, VitBlockIO *io)
    {
        const int wave = (int)(threadIdx.x >> 6);
        const int lane = lane_id();
        const int j = (int)blockIdx.x * 4 + wave;
        if (j >= nblk)
            return;
        const int nber = c.nber, nenc = c.nenc > 0 ? c.nenc : c.nber;
        const unsigned *vb = vbits + (size_t)j * wpb;
        const unsigned *vprev = (j > 0) ? vbits + (size_t)(j - 1) * wpb : nullptr;
        // Eight bits a turn (round 6; a bit a turn with SymFetch::pair per bit made this kernel 2.1 ms of an NPP step): the received pairs of eight steps come
        // out of V2Fetch (the forward pass's own fetch: a few dword loads and register arithmetic), the encoder's windows out of one 14-bit piece of the decoded
        // stream -- bits i - 6 .. i + 7, MSB first --, of which step k's shift register (bit i + k - d at position d) is bits 7 - k .. 13 - k.
        V2Fetch<-1, -1> fx;
        fx.init(c, soft, first_block, j);
        const int ngroups = (nber + 7) / 8;
        unsigned err = 0, tot = 0;
        for (int gi = lane; gi < ngroups; gi += 64)
        {
            const auto raw = fx.load(VIT2_WARM + 8 * gi);
            const uint4 q = fx.decode(raw);
            const unsigned qq[4] = {q.x, q.y, q.z, q.w};
            const int i0 = 8 * gi;
            unsigned win;
            if (gi > 0)
                win = peek_bits(vb, (long long)i0 - 6, 14);
            else
            { // the six bits in front of the block: the previous block's encoder stopped behind its nenc-th bit; in front of the first block, the carried register
                const unsigned six = vprev ? peek_bits(vprev, (long long)nenc - 6, 6) : (enc_state_in & 63u);
                win = (six << 8) | peek_bits(vb, 0, 8);
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                if (i0 + k >= nber)
                    break;
                const unsigned stw = (win >> (7 - k)) & 127u; // bits i, i-1, ..., i-6 at positions 0..6
                const unsigned o0 = parity32(stw & 79u), o1 = parity32(stw & 109u);
                const unsigned pr = (qq[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                const unsigned s0 = pr & 255u, s1 = pr >> 8;
                if (s0 != 128u)
                {
                    tot++;
                    err += ((s0 > 127u) ? 1u : 0u) != o0;
                }
                if (s1 != 128u)
                {
                    tot++;
                    err += ((s1 > 127u) ? 1u : 0u) != o1;
                }
            }
        }
        err = wave_sum_u32(err);
        tot = wave_sum_u32(tot);
        if (lane == 0)
        {
            io[j].ber_err = (int)err;
            io[j].ber_tot = (int)tot;
            unsigned e = 0; // encoder register after this block: last 6 encoded bits, newest at bit 0
            for (int d = 0; d < 6; d++)
                e |= getbit(vb, nenc - 1 - d) << d;
            io[j].pad = (int)e;
        }
    }

    void launch_vit_ber(const VitCfg &cfg, const int8_t *soft, int64_t first_block, int nblk, const uint32_t *vbits, unsigned enc_state_in, VitBlockIO *io,
                        hipStream_t st)
    {
        if (nblk <= 0)
            return;
        ProfScope _ps("k_vit_ber", st);
        hipLaunchKernelGGL(k_vit_ber, dim3((nblk + 3) / 4), dim3(256), 0, st, cfg, soft, (long long)first_block, nblk, vbits, vit_words_per_block(cfg.F),
                           enc_state_in, io);
    }

    // =============================================================================================
    // fengyun_ahrpt_decoder: the rail split in front of the two Viterbi3_4 decoders and the differential decoder behind them
    // =============================================================================================
    __global__ __launch_bounds__(256) void k_fy_rails(const int8_t *__restrict__ soft, long long first_block, int nblk, int shift, int invert_second, int8_t *rail0,
                                                       int8_t *rail1, int mpt)
    { // thread = four consecutive symbols of one block: two dword loads (+ one for the shifted pair), one dword store per rail
        const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
        if (g >= (long long)nblk * 2048)
            return;
        const long long j = g >> 11;
        const int i0 = (int)(g & 2047) * 4;
        const unsigned *w = reinterpret_cast<const unsigned *>(soft + (first_block + j) * 16384);
        unsigned r0 = 0, r1 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int p = i0 + q + shift; // pair index inside the block
            int a = 0, b = 0;
            if (p < 8192)
            {
                const unsigned d = w[p >> 1] >> (16 * (p & 1));
                a = (int)(signed char)(d & 0xffu);
                b = (int)(signed char)((d >> 8) & 0xffu);
                a = a == -128 ? -127 : a;
                b = b == -128 ? -127 : b;
            }
            // iq swap: soft_buffer[2p] = b, soft_buffer[2p + 1] = a
            int v0 = b, v1 = invert_second ? ~a : a;
            int qq = q;
            if (mpt)
            { // the rails' own rotate_soft(.., PHASE_0, true): -128 -> -127 (~127), then the bytes of a pair exchanged (i0 is a multiple of 4: pairs (0,1), (2,3))
                v1 = v1 == -128 ? -127 : v1;
                qq = q ^ 1;
            }
            r0 |= ((unsigned)v0 & 0xffu) << (8 * qq);
            r1 |= ((unsigned)v1 & 0xffu) << (8 * qq);
        }
        reinterpret_cast<unsigned *>(rail0 + j * 8192)[i0 >> 2] = r0;
        reinterpret_cast<unsigned *>(rail1 + j * 8192)[i0 >> 2] = r1;
    }
    void launch_fy_rails(const int8_t *soft, int64_t first_block, int nblk, int shift, int invert_second, int8_t *rail0, int8_t *rail1, hipStream_t st, int mpt)
    {
        if (nblk <= 0)
            return;
        ProfScope _ps("k_fy_rails", st);
        const long long threads = (long long)nblk * 2048;
        hipLaunchKernelGGL(k_fy_rails, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, soft, (long long)first_block, nblk, shift, (invert_second || mpt) ? 1 : 0, rail0, rail1, mpt);
    }

    __global__ __launch_bounds__(256) void k_fy_diff(const unsigned *__restrict__ x, const unsigned *__restrict__ y, int nblk, int bits_per_rail, int wpb_rail,
                                                      unsigned x_prev, unsigned y_prev, unsigned *out, int wpb_out)
    { // thread = one output word = 16 rail positions: half a word of each rail and the bit in front of it
        const int wout = 2 * bits_per_rail / 32;
        const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
        if (g >= (long long)nblk * wout)
            return;
        const int j = (int)(g / wout), w = (int)(g - (long long)j * wout);
        const int k0 = 16 * w; // first rail position of this word
        const unsigned *xb = x + (size_t)j * wpb_rail, *yb = y + (size_t)j * wpb_rail;
        const unsigned xw = xb[k0 >> 5], yw = yb[k0 >> 5];
        unsigned xp, yp; // the rail bits in front of position k0
        if (k0 & 31)
        {
            xp = (xw >> (32 - (k0 & 31))) & 1u;
            yp = (yw >> (32 - (k0 & 31))) & 1u;
        }
        else if (k0 > 0)
        {
            xp = xb[(k0 >> 5) - 1] & 1u;
            yp = yb[(k0 >> 5) - 1] & 1u;
        }
        else if (j > 0)
        {
            xp = (xb - wpb_rail)[(bits_per_rail - 1) >> 5] >> (31 - ((bits_per_rail - 1) & 31)) & 1u;
            yp = (yb - wpb_rail)[(bits_per_rail - 1) >> 5] >> (31 - ((bits_per_rail - 1) & 31)) & 1u;
        }
        else
        {
            xp = x_prev & 1u;
            yp = y_prev & 1u;
        }
        unsigned o = 0;
#pragma unroll
        for (int q = 0; q < 16; q++)
        {
            const int k = k0 + q;
            const unsigned xi = (xw >> (31 - (k & 31))) & 1u, yi = (yw >> (31 - (k & 31))) & 1u;
            const unsigned dx = xi ^ xp, dy = yi ^ yp;
            // diff.cpp:61-74: X != Y -> (high, low) = (dy, dx), else (dx, dy)
            const unsigned hi = (xi ^ yi) ? dy : dx, lo = (xi ^ yi) ? dx : dy;
            o = (o << 2) | (hi << 1) | lo;
            xp = xi;
            yp = yi;
        }
        out[(size_t)j * wpb_out + w] = o;
    }
    void launch_fy_diff(const uint32_t *x, const uint32_t *y, int nblk, int bits_per_rail, int wpb_rail, unsigned x_prev, unsigned y_prev, uint32_t *out, int wpb_out,
                        hipStream_t st)
    {
        if (nblk <= 0)
            return;
        ProfScope _ps("k_fy_diff", st);
        const long long threads = (long long)nblk * (2 * bits_per_rail / 32);
        hipLaunchKernelGGL(k_fy_diff, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, x, y, nblk, bits_per_rail, wpb_rail, x_prev, y_prev, out, wpb_out);
    }

    // =============================================================================================
    // Generic punctured rates (conv_rate 2/3 .. 7/8): viterbi::puncturing::Depunc23/34/56/78, depunc.h:21-430.
    // The lock search and the call-by-call path (k_punc_static / k_punc_cont: one call of 8192 symbols per launch) follow the
    // reference step by step; a SYNCED run of calls is depunctured by ONE launch (k_punc_batch) and decoded as a batch of
    // overlapping blocks by the rate 1/2 engine's own kernels (FecEngine::punc_run).
    // =============================================================================================
    // Output index of input i is a closed form of the pattern: j = i + pos0 inputs from a period start, q = j / n whole periods of
    // P = n + popcount(two) outputs, plus what positions pos0 .. (j % n) - 1 of the current period emit. One block, a thread per
    // input (strided): the same bytes the reference's sequential loops write.
    __device__ __forceinline__ int punc_pre(const PuncPat &pat, int r)
    { // outputs of pattern positions 0 .. r-1
        return r + __popc(pat.two & ((1u << r) - 1u));
    }
    __device__ __forceinline__ void punc_scatter(const PuncPat &pat, const SymFetch &f, int n_in, int pos0, int lead, unsigned char *dst)
    {
        const int P = pat.n + __popc(pat.two);
        const int base = lead - punc_pre(pat, pos0);
        for (int i = (int)threadIdx.x; i < n_in; i += (int)blockDim.x)
        {
            const int j = i + pos0, q = j / pat.n, r = j - q * pat.n;
            int oo = base + q * P + punc_pre(pat, r);
            const unsigned u = f.u_at(i);
            if (!((pat.two >> r) & 1u))
                dst[oo] = (unsigned char)u;
            else if ((pat.lead128 >> r) & 1u)
            {
                dst[oo] = 128;
                dst[oo + 1] = (unsigned char)u;
            }
            else
            {
                dst[oo] = (unsigned char)u;
                dst[oo + 1] = 128;
            }
        }
    }
    // depunc_static(in, out, n_in, shift) on the rotated / converted first n_in symbols of a block (lock search)
    __global__ __launch_bounds__(256) void k_punc_static(VitCfg c, const int8_t *blk, PuncPat pat, int shift, int n_in, unsigned char *out)
    {
        const SymFetch f{c, blk, n_in};
        const int lead = shift > pat.n - 1 ? 1 : 0;
        if (lead && threadIdx.x == 0)
            out[0] = 128;
        punc_scatter(pat, f, n_in, shift % pat.n, lead, out);
    }
    // depunc_cont: appended at dst; `lead` = is_first || got_extra (the carried byte goes first), pos0 = changing_shift % n.
    // An odd count (total: the host computes it, it is a function of the pattern alone) leaves its last symbol in *carry.
    __global__ __launch_bounds__(256) void k_punc_cont(VitCfg c, const int8_t *blk, int n_in, PuncPat pat, int pos0, int lead, int total, unsigned char *carry,
                                                       unsigned char *dst)
    {
        const SymFetch f{c, blk, n_in};
        // An odd count: the last depunctured symbol is carried to the next call, and ViterbiSlidingBuffer::add copies only the even
        // part (viterbi_punc.cpp:108-118, viterbi_buffer.h:26-30): the buffer slot behind it keeps what it held -- cc_decoder reads
        // a few symbols past the block, so that slot is visible to the chainback's end state. Scatter, take the carry, put the old
        // byte back.
        unsigned char old = 0;
        if ((total & 1) && threadIdx.x == 0)
            old = dst[total - 1];
        __syncthreads();
        if (lead && threadIdx.x == 0)
            dst[0] = *carry;
        punc_scatter(pat, f, n_in, pos0, lead, dst);
        __syncthreads();
        if ((total & 1) && threadIdx.x == 0)
        {
            *carry = dst[total - 1];
            dst[total - 1] = old;
        }
    }
    __global__ __launch_bounds__(256) void k_punc_batch(VitCfg c, const int8_t *soft, long long first_block, int n_in, PuncPat pat, const PuncDesc *desc,
                                                        unsigned char *lin)
    {
        const int b = (int)blockIdx.x;
        const SymFetch f{c, soft + (first_block + b) * (long long)n_in, n_in};
        punc_scatter(pat, f, n_in, desc[b].pos0, 0, lin + desc[b].off);
    }
    void launch_punc_batch(const VitCfg &c, const int8_t *soft, long long first_block, int nblk, int n_in, const PuncPat &pat, const PuncDesc *desc, unsigned char *lin,
                           hipStream_t st)
    {
        if (nblk <= 0)
            return;
        ProfScope _ps("k_punc_batch", st);
        hipLaunchKernelGGL(k_punc_batch, dim3((unsigned)nblk), dim3(256), 0, st, c, soft, first_block, n_in, pat, desc, lin);
    }
    void launch_punc_static(const VitCfg &c, const int8_t *blk, const PuncPat &pat, int shift, int n_in, unsigned char *out, hipStream_t st)
    {
        ProfScope _ps("k_punc_static", st);
        hipLaunchKernelGGL(k_punc_static, dim3(1), dim3(256), 0, st, c, blk, pat, shift, n_in, out);
    }
    void launch_punc_cont(const VitCfg &c, const int8_t *blk, int n_in, const PuncPat &pat, int pos0, int lead, int total, unsigned char *carry,
                          unsigned char *dst, hipStream_t st)
    {
        ProfScope _ps("k_punc_cont", st);
        hipLaunchKernelGGL(k_punc_cont, dim3(1), dim3(256), 0, st, c, blk, n_in, pat, pos0, lead, total, carry, dst);
    }

    // =============================================================================================
    // k_vit_search: the IDLE branch of Viterbi1_2::work / Viterbi3_4::work on one block (one wave)
    // =============================================================================================
    constexpr int SEARCH_MAX_BITS = 1536;

    struct SearchCand
    {
        int s, phase, shift;
    };

    __global__ __launch_bounds__(64) void k_vit_search(VitCfg c, const int8_t *__restrict__ soft, long long block, int n_swap, int ph0, int ph1, int ph2, int ph3,
                                                        int nphases, VitSearchState *S)
    {
        __shared__ unsigned long long ballots[SEARCH_MAX_BITS + 8];
        __shared__ unsigned char bits[SEARCH_MAX_BITS];
        __shared__ unsigned char tailb[16];
        const int lane = lane_id();
        const int nb = c.nber, nsteps = nb + 6;
        const int phases[4] = {ph0, ph1, ph2, ph3};
        AcsConsts k;
        acs_init_consts(k);
        int ber_first = S->ber_first, ber_start = S->ber_start;
        unsigned enc = S->enc_state;
        if (lane < 16)
            tailb[lane] = S->tail[lane];
        __syncthreads();
        const int8_t *blk = soft + block * vit_stride(c);
        int cand = 0;
        const int nsw = (c.mode == 0) ? n_swap : 1;
        const int nph = (c.mode == 0) ? nphases : (c.fy ? 1 : 2);
        for (int s = 0; s < nsw; s++)
            for (int pi = 0; pi < nph; pi++)
                for (int shift = 0; shift < 2; shift++)
                {
                    VitCfg cc = c;
                    cc.iq_swap = (c.mode == 0) ? s : 0;
                    cc.phase = (c.mode == 0) ? phases[pi] : pi;
                    cc.shift = shift;
                    SymFetch f{cc, blk, 2048};
                    // mode 0: symbols past ber_soft_buffer come from ber_decoded_buffer (previous candidate's
                    // decoded bits, viterbi_1_2.h:39-42); mode 1: never-written zero bytes of ber_depunc_buffer
                    auto tail = [&](int idx) -> unsigned { return (cc.mode == 0) ? (unsigned)tailb[idx & 15] : 0u; };
                    auto fetch = [&](int tt) -> unsigned { return f.pair(tt, tail); };
                    unsigned X = ber_first ? 31u : ((lane == ber_start) ? 0u : 63u);
                    SinkLds sk{ballots};
                    X = acs_forward(nsteps, X, k, fetch, sk);
                    unsigned st = acs_endstate(X, nsteps);
                    __syncthreads();
                    ber_first = 0;
                    for (int n = nb - 1; n >= 0; n--) // chainback_viterbi, cc_decoder.cpp:228-276: bit n <- step n + 6
                    {
                        const int t = n + 6;
                        const unsigned kb = dec_bit(ballots[t], t, st);
                        st = (st >> 1) | (kb << 5);
                        if (lane == 0)
                            bits[n] = (unsigned char)kb;
                        if (n == nb - 6)
                            ber_start = (int)st; // chained start state of the next cc_decoder_ber.work
                    }
                    __syncthreads();
                    if (lane < 16)
                        tailb[lane] = bits[lane]; // ber_decoded_buffer now holds the new bits
                    __syncthreads();
                    // cc_encoder_ber.work + get_ber (the BER compare sees the UPDATED ber_decoded tail)
                    const int per = (nb + 63) / 64;
                    unsigned err = 0, tot = 0;
                    for (int q = 0; q < per; q++)
                    {
                        const int i = lane * per + q;
                        if (i >= nb)
                            break;
                        unsigned stw = 0;
                        for (int d = 0; d < 7; d++)
                        {
                            const int n = i - d;
                            const unsigned bit = (n >= 0) ? bits[n] : ((enc >> (-n - 1)) & 1u);
                            stw |= bit << d;
                        }
                        const unsigned o0 = parity32(stw & 79u), o1 = parity32(stw & 109u);
                        const unsigned pr = f.pair(i, tail);
                        const unsigned s0 = pr & 255u, s1 = pr >> 8;
                        if (s0 != 128u)
                        {
                            tot++;
                            err += ((s0 > 127u) ? 1u : 0u) != o0;
                        }
                        if (s1 != 128u)
                        {
                            tot++;
                            err += ((s1 > 127u) ? 1u : 0u) != o1;
                        }
                    }
                    err = wave_sum_u32(err);
                    tot = wave_sum_u32(tot);
                    unsigned e = 0;
                    for (int d = 0; d < 6; d++)
                        e |= (unsigned)bits[nb - 1 - d] << d;
                    enc = e;
                    if (lane == 0)
                    {
                        S->err[cand] = (int)err;
                        S->tot[cand] = (int)tot;
                    }
                    cand++;
                    __syncthreads();
                }
        if (lane == 0)
        {
            S->ber_first = ber_first;
            S->ber_start = ber_start;
            S->enc_state = enc;
            S->ncand = cand;
        }
        if (lane < 16)
            S->tail[lane] = tailb[lane];
    }

    void launch_vit_search(const VitCfg &cfg, const int8_t *soft, int64_t block, int n_swap, const int *phases, int nphases, VitSearchState *d_state, hipStream_t st)
    {
        int ph[4] = {0, 0, 0, 0};
        for (int i = 0; i < nphases && i < 4; i++)
            ph[i] = phases[i];
        ProfScope _ps("k_vit_search", st);
        hipLaunchKernelGGL(k_vit_search, dim3(1), dim3(64), 0, st, cfg, soft, (long long)block, n_swap, ph[0], ph[1], ph[2], ph[3], nphases, d_state);
    }

    // =============================================================================================
    // k_hard_bits: ccsds_simple_psk_decoder's bit slicers (module_ccsds_simple_psk_decoder.cpp:141-262)
    // =============================================================================================
    // All of the module's per-buffer transformations have a memory of at most two symbols (OQPSK one-symbol I delay,
    // delay_one of method 2/3, the differential decoder), so every output bit is a function of at most three
    // consecutive symbols of the raw stream: data-parallel, thread per 32-bit output word.
    struct HardSym
    {
        int a, b; // soft_buffer[2i], soft_buffer[2i+1] after the optional OQPSK delay and I/Q swap
    };
    __device__ __forceinline__ int hard_raw(const HardCfg &hc, const int8_t *soft, long long k)
    { // soft byte k of this call; k in [-4, 0) = the carried tail
        return k >= 0 ? (int)soft[k] : hc.tail[4 + k];
    }
    __device__ __forceinline__ HardSym hard_sym(const HardCfg &hc, const int8_t *soft, long long i)
    { // symbol i of this call (i >= -1)
        int I = hc.oqpsk_delay ? hard_raw(hc, soft, 2 * (i - 1)) : hard_raw(hc, soft, 2 * i); // :151-160
        int Q = hard_raw(hc, soft, 2 * i + 1);
        if (hc.swap_iq) // rotate_soft(.., PHASE_0, true), :162-163
        {
            const int t = I;
            I = Q;
            Q = t;
        }
        return HardSym{I, Q};
    }
    __global__ __launch_bounds__(256) void k_hard_bits(HardCfg hc, const int8_t *__restrict__ soft, int nblk, int which, unsigned *vbits, int wpb)
    {
        const int tiles = (wpb + 255) / 256; // 1-D grid (grid.y is limited to 65535 blocks)
        const int j = (int)(blockIdx.x / tiles);
        const int w = (int)((blockIdx.x % tiles) * blockDim.x + threadIdx.x);
        if (j >= nblk || w >= wpb)
            return;
        const int F = hc.F;
        unsigned word = 0;
        for (int bb = 0; bb < 32; bb++)
        {
            const int n = w * 32 + bb;
            if (n >= F)
                break;
            unsigned bit;
            if (!hc.qpsk)
                bit = soft[(long long)j * F + n] > 0; // :143-144 (NRZ-M is applied by the stream reader)
            else
            {
                const long long i = (long long)j * (F / 2) + (n >> 1); // symbol of this call holding output bit n
                const int second = n & 1;
                if (hc.nrzm)
                { // soft_demod + QPSKDiff (qpsk_diff.cpp:5-55): the very first two symbols of the stream produce nothing, so
                  // block 0 of the stream holds F-4 decoded bits followed by 4 never-written (zero) entries of bits_out
                    long long cur = i;
                    bool valid = true;
                    if (hc.blocks_done == 0 && j == 0)
                    {
                        cur = i + 2;
                        valid = n < F - 4;
                    }
                    bit = 0;
                    if (valid)
                    {
                        const HardSym p = hard_sym(hc, soft, cur - 1), c = hard_sym(hc, soft, cur);
                        const unsigned Xin_1 = p.b > 0, Yin_1 = p.a > 0, Xin = c.b > 0, Yin = c.a > 0;
                        unsigned ou;
                        if ((Xin ^ Yin) == 1u)
                            ou = ((Yin_1 ^ Yin) << 1) + (Xin_1 ^ Xin);
                        else
                            ou = ((Xin_1 ^ Xin) << 1) + (Yin_1 ^ Yin);
                        const unsigned first = hc.swap_diff ? (ou & 1u) : (ou >> 1), sec = hc.swap_diff ? (ou >> 1) : (ou & 1u);
                        bit = second ? sec : first;
                    }
                }
                else
                {
                    const HardSym c = hard_sym(hc, soft, i);
                    // bits_out[2i] = sym >> 1 = (sample[1] > 0), bits_out[2i+1] = sym & 1 = (sample[0] > 0); PHASE_90: (a, b) -> (b, -a)
                    unsigned b0, b1;
                    const bool m23 = hc.method2 || hc.method3;
                    if (which == 0)
                    { // main deframer
                        if (m23 && hc.method3)
                        { // unrotated soft_buffer, :236-241
                            b0 = c.b > 0;
                            b1 = c.a > 0;
                        }
                        else
                        { // soft_buffer rotated by 90 degrees, :186-193 / :213-220
                            b0 = c.a < 0;
                            b1 = c.b > 0;
                        }
                    }
                    else
                    { // deframer_qpsk
                        if (!m23)
                        { // 0 degrees, :176-183
                            b0 = c.b > 0;
                            b1 = c.a > 0;
                        }
                        else
                        {
                            const int ad = hard_sym(hc, soft, i - 1).a; // soft_buffer2: I delayed by one symbol, :197-203
                            if (!hc.method3)
                            {
                                b0 = c.b > 0;
                                b1 = ad > 0;
                            }
                            else
                            { // delayed copy rotated by 90 degrees: (ad, b) -> (b, -ad), :231-232
                                b0 = ad < 0;
                                b1 = c.b > 0;
                            }
                        }
                    }
                    bit = second ? b1 : b0;
                }
            }
            word |= bit << (31 - bb);
        }
        vbits[(size_t)j * wpb + w] = word;
    }
    void launch_hard_bits(const HardCfg &hc, const int8_t *soft, int nblk, int which, uint32_t *vbits, int wpb, hipStream_t st)
    {
        if (nblk <= 0)
            return;
        ProfScope _ps("k_hard_bits", st);
        hipLaunchKernelGGL(k_hard_bits, dim3((unsigned)(((wpb + 255) / 256) * (long long)nblk)), dim3(256), 0, st, hc, soft, nblk, which, vbits, wpb);
    }

    // =============================================================================================
    // Logical bit stream access
    // =============================================================================================
    __device__ __forceinline__ unsigned stream_raw32(const BitStream &bs, long long g)
    {
        unsigned out = 0;
        int got = 0;
        while (got < 32)
        {
            const unsigned *base = nullptr;
            long long off = 0, avail = 32;
            if (g < 0)
            {
                avail = -g;
            }
            else if (g < bs.carry_bits)
            {
                base = bs.carry;
                off = g;
                avail = bs.carry_bits - g;
            }
            else
            {
                const long long g2 = g - bs.carry_bits;
                const long long j = g2 / bs.F;
                if (j < bs.nblk)
                {
                    const long long o = g2 - j * bs.F;
                    base = bs.vbits + (size_t)j * bs.wpb;
                    off = o;
                    avail = bs.F - o;
                }
            }
            int take = 32 - got;
            if (avail < take)
                take = (int)avail;
            const unsigned v = base ? peek_bits(base, off, take) : 0u;
            out |= v << (32 - got - take);
            got += take;
            g += take;
        }
        return out;
    }
    // 32 bits of the stream the deframer sees, starting at logical bit g (NRZ-M: out = b ^ previous b)
    __device__ __forceinline__ unsigned stream32(const BitStream &bs, long long g)
    {
        const unsigned r = stream_raw32(bs, g);
        if (!bs.nrzm)
            return r;
        const unsigned prev = stream_raw32(bs, g - 1);
        return r ^ prev;
    }

    // ---- exact ASM search ---------------------------------------------------------------------
    __global__ __launch_bounds__(256) void k_sync_search(BitStream bs, long long from, long long total, unsigned asm_sync, unsigned *hits, int hits_cap, int *count)
    {
        // each thread owns 32 consecutive END positions p = from + 32*i + k
        const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        const long long p0 = from + 32 * i;
        if (p0 >= total)
            return;
        const unsigned a = stream32(bs, p0 - 31); // bits p0-31 .. p0
        const unsigned b = stream32(bs, p0 + 1);  // bits p0+1 .. p0+32
        const unsigned long long v = ((unsigned long long)a << 32) | b;
        const unsigned inv = ~asm_sync;
#pragma unroll 4
        for (int k2 = 0; k2 < 32; k2++)
        {
            const long long p = p0 + k2;
            if (p >= total)
                break;
            const unsigned w = (unsigned)(v >> (32 - k2));
            if (w == asm_sync || w == inv)
            {
                const int idx = atomicAdd(count, 1);
                if (idx < hits_cap)
                    hits[idx] = (unsigned)((p - from) << 1) | (w == inv ? 1u : 0u);
            }
        }
    }

    void launch_sync_search(const BitStream &bs, int64_t from, uint32_t asm_sync, uint32_t *hits, int hits_cap, int *count, hipStream_t st)
    {
        const int64_t total = bs.carry_bits + bs.nblk * (int64_t)bs.F;
        const int64_t n = (total - from + 31) / 32;
        if (n <= 0)
            return;
        ProfScope _ps("k_sync_search", st);
        hipLaunchKernelGGL(k_sync_search, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bs, (long long)from, (long long)total, asm_sync, hits, hits_cap, count);
    }

    __global__ __launch_bounds__(256) void k_pack_stream(BitStream bs, unsigned char *out, long long total_bits)
    {
        const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i * 32 >= total_bits)
            return;
        const unsigned w = stream32(bs, i * 32);
        out[i * 4 + 0] = (unsigned char)(w >> 24);
        out[i * 4 + 1] = (unsigned char)(w >> 16);
        out[i * 4 + 2] = (unsigned char)(w >> 8);
        out[i * 4 + 3] = (unsigned char)(w);
    }
    // words[k * WIN_OFFS + o] = the 32 bits ending at stream bit p0 + k * step + o of the packed stream, o = 0 .. WIN_OFFS - 1: the
    // window the deframer FSM compares with the ASM at an expected frame position, and the ones it looks at one and two bits
    // further on when that comparison fails while it is still SYNCING (bpsk_ccsds_deframer.cpp:68-87: three failures in a row
    // before it drops to NOSYNC); positions outside [31, total) yield 0
    __global__ __launch_bounds__(256) void k_window_gather(const unsigned char *__restrict__ packed, long long total_bits, long long p0, int step, int K,
                                                            unsigned *words)
    {
        const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (k >= K)
            return;
        const long long pk = p0 + (long long)k * step;
#pragma unroll
        for (int o = 0; o < WIN_OFFS; o++)
        {
            const long long p = pk + o;
            unsigned w = 0;
            if (p >= 31 && p < total_bits)
            {
                const long long s0 = p - 31, byte = s0 >> 3;
                const int sh = (int)(s0 & 7);
                unsigned long long v = 0;
                for (int i = 0; i < 5; i++)
                    v = (v << 8) | packed[byte + i];
                w = (unsigned)((v << (24 + sh)) >> 32);
            }
            words[(size_t)k * WIN_OFFS + o] = w;
        }
    }
    void launch_window_gather(const uint8_t *packed, int64_t total_bits, int64_t p0, int step, int K, uint32_t *words, hipStream_t st)
    {
        if (K <= 0)
            return;
        ProfScope _ps("k_window_gather", st);
        hipLaunchKernelGGL(k_window_gather, dim3((K + 255) / 256), dim3(256), 0, st, packed, (long long)total_bits, (long long)p0, step, K, words);
    }
    void launch_pack_stream(const BitStream &bs, uint8_t *out_bytes, int64_t total_bits, hipStream_t st)
    {
        const int64_t n = (total_bits + 31) / 32;
        if (n <= 0)
            return;
        ProfScope _ps("k_pack_stream", st);
        hipLaunchKernelGGL(k_pack_stream, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bs, out_bytes, (long long)total_bits);
    }

    // =============================================================================================
    // GF(256) tables (libcorrect field.h:32-62, poly 0x187) + CCSDS tables, built once per device
    // =============================================================================================
    struct GfTables
    {
        unsigned char exp[512];
        unsigned char log[256];
        unsigned char to_dual[256];
        unsigned char from_dual[256];
        unsigned char pn[256]; // 255 used
    };
    static GfTables *g_tables_dev[16] = {nullptr};

    static void build_tables_host(GfTables &t)
    {
        unsigned element = 1;
        t.exp[0] = 1;
        t.log[0] = 0;
        for (unsigned i = 1; i < 512; i++)
        {
            element = element * 2;
            element = (element > 255) ? (element ^ 0x187) : element;
            t.exp[i] = (unsigned char)element;
            if (i < 256)
                t.log[element] = (unsigned char)i;
        }
        auto mul = [&](unsigned a, unsigned b) -> unsigned { return (a == 0 || b == 0) ? 0u : t.exp[t.log[a] + t.log[b]]; };
        // CCSDS dual basis: z_j = Tr(x * beta^j), beta = alpha^117, z_0 = MSB (reedsolomon.cpp:6-28 tables)
        for (unsigned x = 0; x < 256; x++)
        {
            unsigned z = 0;
            for (unsigned j = 0; j < 8; j++)
            {
                unsigned y = mul(x, t.exp[(117 * j) % 255]), tr = 0;
                for (int q = 0; q < 8; q++)
                {
                    tr ^= y;
                    y = mul(y, y);
                }
                z |= (tr & 1u) << (7 - j);
            }
            t.to_dual[x] = (unsigned char)z;
        }
        for (unsigned x = 0; x < 256; x++)
            t.from_dual[t.to_dual[x]] = (unsigned char)x;
        // CCSDS pseudo-randomiser h(x) = x^8+x^7+x^5+x^3+1, all-ones seed (randomization.cpp:4-36)
        unsigned char reg[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        for (int i = 0; i < 255; i++)
        {
            unsigned char byte = 0;
            for (int b = 0; b < 8; b++)
            {
                byte = (unsigned char)(byte << 1 | reg[0]);
                unsigned char nb = reg[0] ^ reg[3] ^ reg[5] ^ reg[7];
                memmove(reg, reg + 1, 7);
                reg[7] = nb;
            }
            t.pn[i] = byte;
        }
        t.pn[255] = 0;
    }

    static const GfTables *tables_for_current_device()
    {
        int dev = 0;
        SD_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 16)
            throw HipError("device ordinal out of range");
        if (!g_tables_dev[dev])
        {
            GfTables h;
            build_tables_host(h);
            GfTables *d = nullptr;
            SD_HIP(hipMalloc((void **)&d, sizeof(GfTables)));
            SD_HIP(hipMemcpy(d, &h, sizeof(GfTables), hipMemcpyHostToDevice));
            g_tables_dev[dev] = d;
        }
        return g_tables_dev[dev];
    }

    // =============================================================================================
    // Reed-Solomon, one thread per codeword. cw[] is this thread's 255-byte codeword held in LDS,
    // transposed: byte k of thread `tid` at cw[k * nthr + tid].
    // Mirrors correct_reed_solomon_decode (decode.c:299-379) with num_erasures = 0.
    // =============================================================================================
    struct GfLds
    {
        unsigned char exp[512];
        unsigned char log[256];
    };

    struct RsCtx
    {
        const GfLds *gf;
        int nroots, fcr, gap;
        __device__ __forceinline__ unsigned mul(unsigned l, unsigned r) const { return (l == 0 || r == 0) ? 0u : gf->exp[gf->log[l] + gf->log[r]]; }
        __device__ __forceinline__ unsigned div(unsigned l, unsigned r) const { return (l == 0 || r == 0) ? 0u : gf->exp[255u + gf->log[l] - gf->log[r]]; }
        __device__ __forceinline__ unsigned pw(unsigned e, int p) const
        {
            int m = ((int)gf->log[e] * p) % 255;
            if (m < 0)
                m += 255;
            return gf->exp[m];
        }
        __device__ __forceinline__ static unsigned mul_log(unsigned l, unsigned r)
        {
            const unsigned res = l + r;
            return res > 255 ? res - 255 : res;
        }
    };

    // returns -1 on failure, else msg_length; corrects cw in place (conventional basis, byte 0 = highest order)
    // synd_pre: the codeword's syndromes as k_rs_screen left them (the same Horner evaluation, one lane per root), or nullptr
    __device__ int rs_decode_thread(const RsCtx &R, unsigned char *cw, int stride, const unsigned char *synd_pre = nullptr)
    {
        const int md = R.nroots;
#define CW(k) cw[(k) * stride]
        // syndromes: S_i = r(alpha_i) by Horner over the received polynomial (field arithmetic is exact, so this
        // equals polynomial_eval_lut over generator_root_exp, decode.c:12-28)
        unsigned char synd[32];
        bool all_zero = true;
        if (synd_pre)
        { // 255 x 32 dependent look-ups per thread not spent again: they were ~70 % of this function
            const uint4 a = reinterpret_cast<const uint4 *>(synd_pre)[0], b = reinterpret_cast<const uint4 *>(synd_pre)[1];
            const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            for (int i = 0; i < 32; i++)
                synd[i] = (unsigned char)(w[i >> 2] >> (8 * (i & 3)));
            for (int i = 0; i < md; i++)
                if (synd[i])
                    all_zero = false;
        }
        else
        for (int i = 0; i < md; i++)
        {
            const unsigned lr = (unsigned)((R.gap * (i + R.fcr)) % 255); // log of generator root i (reed-solomon.c:9-11)
            unsigned s = 0;
            for (int k = 0; k < 255; k++)
            {
                if (s)
                    s = R.gf->exp[R.gf->log[s] + lr];
                s ^= CW(k);
            }
            synd[i] = (unsigned char)s;
            if (s)
                all_zero = false;
        }
        if (all_zero)
            return 255 - md;

        // Berlekamp-Massey, decode.c:32-118
        unsigned char loc[40], last[40];
        for (int i = 0; i < 40; i++)
        {
            loc[i] = 0;
            last[i] = 0;
        }
        loc[0] = 1;
        last[0] = 1;
        unsigned loc_order = 0, last_order = 0, numerrors = 0;
        unsigned last_disc = 1, delay = 1;
        for (unsigned i = 0; i < (unsigned)md; i++)
        {
            unsigned disc = synd[i];
            for (unsigned j = 1; j <= numerrors; j++)
                disc ^= R.mul(loc[j], synd[i - j]);
            if (!disc)
            {
                delay++;
                continue;
            }
            if (2 * numerrors <= i)
            {
                for (int j = (int)last_order; j >= 0; j--)
                {
                    const unsigned idx = (unsigned)j + delay;
                    const unsigned v = R.div(R.mul(last[j], disc), last_disc);
                    if (idx < 40)
                        last[idx] = (unsigned char)v;
                }
                for (int j = (int)delay - 1; j >= 0; j--)
                    if (j < 40)
                        last[j] = 0;
                for (int j = 0; j <= (int)(last_order + delay) && j < 40; j++)
                {
                    const unsigned char tmp = loc[j];
                    loc[j] = loc[j] ^ last[j];
                    last[j] = tmp;
                }
                const unsigned tmp_order = loc_order;
                loc_order = last_order + delay;
                last_order = tmp_order;
                numerrors = i + 1 - numerrors;
                last_disc = disc;
                delay = 1;
                continue;
            }
            for (int j = (int)last_order; j >= 0; j--)
            {
                const unsigned idx = (unsigned)j + delay;
                if (idx < 40)
                    loc[idx] ^= (unsigned char)R.div(R.mul(last[j], disc), last_disc);
            }
            loc_order = (last_order + delay > loc_order) ? last_order + delay : loc_order;
            delay++;
        }
        const unsigned order = loc_order;
        if (order >= 40)
            return -1; // outside the reference's own buffers (heap overrun there); cannot be mirrored
        // Chien search over all 256 field elements, decode.c:122-145 (zero coefficients are skipped through the
        // log(0) = 0 sentinel, polynomial.c:136-157)
        unsigned char loc_log[40];
        for (unsigned i = 0; i <= order; i++)
            loc_log[i] = R.gf->log[loc[i]];
        unsigned char roots[40];
        for (unsigned i = 0; i < 40; i++)
            roots[i] = 0;
        unsigned nroot = 0;
        for (unsigned e = 0; e < 256; e++)
        {
            unsigned res;
            if (e == 0)
                res = loc_log[0] == 0 ? 0u : R.gf->exp[loc_log[0]];
            else
            {
                res = 0;
                const unsigned le = R.gf->log[e];
                unsigned ve = 255; // log[1]
                for (unsigned i = 0; i <= order; i++)
                {
                    if (loc_log[i] != 0)
                        res ^= R.gf->exp[(unsigned)loc_log[i] + ve];
                    ve = RsCtx::mul_log(ve, le);
                }
            }
            if (!res)
            {
                if (nroot < 40)
                    roots[nroot] = (unsigned char)e;
                nroot++;
            }
        }
        if (nroot != order)
            return -1;
        // error locations, decode.c:198-222
        unsigned char locs[40];
        for (unsigned i = 0; i < order; i++)
        {
            locs[i] = 0;
            if (roots[i] == 0)
                continue;
            const unsigned l = R.div(1, roots[i]);
            for (unsigned j = 0; j < 256; j++)
                if (R.pw(j, R.gap) == l)
                {
                    locs[i] = R.gf->log[j];
                    break;
                }
        }
        // Forney, decode.c:165-196: evaluator = locator * S mod x^md ; derivative of the locator
        unsigned char ev[32];
        for (int i = 0; i < md; i++)
            ev[i] = 0;
        for (unsigned i = 0; i <= order; i++)
        {
            if (i > (unsigned)md - 1)
                continue;
            const unsigned jl = (unsigned)md - 1 - i;
            for (unsigned j = 0; j <= jl; j++)
                ev[i + j] ^= (unsigned char)R.mul(loc[i], synd[j]);
        }
        for (unsigned i = 0; i < order; i++)
        {
            if (roots[i] == 0)
                continue;
            const unsigned le = R.gf->log[roots[i]];
            // polynomial_eval_lut(evaluator) and (derivative) at the root (val_exp[0] = log[1] = 255 != 0)
            unsigned num = 0, den = 0, ve = 255;
            for (unsigned q = 0; q < (unsigned)md; q++)
            {
                if (ev[q] != 0)
                    num ^= R.gf->exp[(unsigned)R.gf->log[ev[q]] + ve];
                if (q + 1 <= order && q <= order - 1)
                {
                    const unsigned dc = ((q + 1) & 1u) ? loc[q + 1] : 0u; // formal derivative, polynomial.c:82-96
                    if (dc != 0)
                        den ^= R.gf->exp[(unsigned)R.gf->log[dc] + ve];
                }
                ve = RsCtx::mul_log(ve, le);
            }
            const unsigned val = R.mul(R.pw(roots[i], R.fcr - 1), R.div(num, den));
            // received_polynomial.coeff[loc] ^= val ; coeff index i <-> codeword byte 254 - i
            CW(254 - (int)locs[i]) ^= (unsigned char)val;
        }
#undef CW
        return 255 - md;
    }

    // One thread per codeword. Frames live in global memory; codeword b of frame f is bytes data[f*stride + ii*I + b].
    // Reproduces ReedSolomon::decode (reedsolomon.cpp:63-116) incl. fill_bytes handling and the error count.
    constexpr int RS_THREADS = 64;
    // dirty != nullptr: thread i decodes codeword dirty[1 + i] of the dirty[0] the screen found with a non-zero syndrome (a
    // compacted list: with one codeword in five dirty, a thread per codeword leaves four lanes in five of every wave idle through the
    // whole Berlekamp-Massey / Chien / Forney sequence)
    __global__ __launch_bounds__(RS_THREADS) void k_rs(unsigned char *data, int nframes, int frame_stride, int dualbasis, int I, int nroots, int fill_bytes,
                                                        int *errors, const GfTables *tabs, const unsigned char *__restrict__ clean, const int *__restrict__ dirty)
    {
        if (dirty && (long long)blockIdx.x * RS_THREADS >= (long long)dirty[0])
            return;
        __shared__ GfLds gf;
        __shared__ unsigned char dual[512]; // to_dual | from_dual
        __shared__ unsigned char cwbuf[256 * RS_THREADS];
        for (int i = (int)threadIdx.x; i < 512; i += RS_THREADS)
        {
            gf.exp[i] = tabs->exp[i];
            if (i < 256)
            {
                gf.log[i] = tabs->log[i];
                dual[i] = tabs->to_dual[i];
                dual[256 + i] = tabs->from_dual[i];
            }
        }
        __syncthreads();
        const int tid = (int)threadIdx.x;
        long long cwid = (long long)blockIdx.x * RS_THREADS + tid;
        if (dirty)
        {
            if (cwid >= (long long)dirty[0])
                return;
            cwid = dirty[1 + cwid];
        }
        if (cwid >= (long long)nframes * I)
            return;
        if (clean && clean[cwid])
        { // all syndromes zero (k_rs_screen): nothing to correct, nothing to write
            errors[cwid] = 0;
            return;
        }
        const int f = (int)(cwid / I), b = (int)(cwid % I);
        unsigned char *base = data + (size_t)f * frame_stride;
        unsigned char *cw = cwbuf + tid;
        const int fb = fill_bytes < 0 ? 0 : fill_bytes; // fill_bytes == -1: no depuncture (the 256th-byte overrun is not reproduced)
        const int coded = 255 - nroots;
        // deinterleave + depuncture (reedsolomon.cpp:65-69,145-149) + dual->conventional
        for (int k = 0; k < 255; k++)
        {
            unsigned v = (k < fb) ? 0u : base[(size_t)(k - fb) * I + b];
            if (dualbasis && k >= fb)
                v = dual[256 + v];
            else if (dualbasis)
                v = dual[256 + 0];
            cw[k * RS_THREADS] = (unsigned char)v;
        }
        // keep a copy of the received message part for the error count
        RsCtx R{&gf, nroots, nroots == 32 ? 112 : 120, 11};
        // count differences while correcting: snapshot first `coded` bytes is expensive in registers, so
        // recount by re-reading the input bytes after decode
        // scratch layout: [clean flag per codeword | 32 syndrome bytes per codeword] (k_rs_screen)
        const long long ncw_all = (long long)nframes * I;
        const unsigned char *synd_pre = clean ? clean + ((ncw_all + 15) / 16 * 16) + (size_t)cwid * 32 : nullptr;
        const int res = rs_decode_thread(R, cw, RS_THREADS, synd_pre);
        if (res < 0)
        {
            errors[cwid] = -1; // data left untouched (the reference restores it, reedsolomon.cpp:78-90)
            return;
        }
        int err = 0;
        for (int k = 0; k < 255; k++)
        {
            unsigned v = cw[k * RS_THREADS];
            unsigned orig = (k < fb) ? 0u : base[(size_t)(k - fb) * I + b];
            unsigned orig_conv = dualbasis ? dual[256 + orig] : orig;
            if (k < coded)
            {
                if (v != orig_conv)
                    err++;
                // write back corrected message bytes (memcpy(data, odata, coded - fill) then ToDualBasis, :103-109;
                // with fill_bytes > 0 the reference leaves the last fill_bytes message bytes uncorrected)
                if (k >= fb && k < coded - fb)
                    base[(size_t)(k - fb) * I + b] = (unsigned char)(dualbasis ? dual[v] : v);
            }
            // parity bytes: the reference converts the RECEIVED parity back unchanged -> no write needed
        }
        errors[cwid] = err;
    }

    // Syndrome screen: 32 lanes per codeword, one generator root each (the same Horner evaluation as rs_decode_thread). A
    // codeword whose syndromes are all zero is what the decoder leaves untouched with an error count of 0 (decode.c:335-342),
    // so k_rs skips it; on a clean link that is every codeword and the thread-per-codeword kernel (255 x 32 dependent table
    // look-ups per thread before it can tell) has nothing left to do.
    constexpr int RSS_CW = 8;
    __global__ __launch_bounds__(32 * RSS_CW) void k_rs_screen(const unsigned char *__restrict__ data, int nframes, int frame_stride, int dualbasis, int I, int nroots,
                                                                int fill_bytes, unsigned char *clean, const GfTables *tabs, int *errors, int *dirty)
    {
        __shared__ GfLds gf;
        __shared__ unsigned char from_dual[256];
        __shared__ unsigned char cw[RSS_CW][256];
        __shared__ int nz[RSS_CW];
        __shared__ unsigned char sy_sh[RSS_CW][32];
        const int tid = (int)threadIdx.x;
        for (int i = tid; i < 512; i += 32 * RSS_CW)
        {
            gf.exp[i] = tabs->exp[i];
            if (i < 256)
            {
                gf.log[i] = tabs->log[i];
                from_dual[i] = tabs->from_dual[i];
            }
        }
        if (tid < RSS_CW)
            nz[tid] = 0;
        const long long cw0 = (long long)blockIdx.x * RSS_CW, ncw = (long long)nframes * I;
        const int fb = fill_bytes < 0 ? 0 : fill_bytes;
        __syncthreads();
        for (int idx = tid; idx < RSS_CW * 255; idx += 32 * RSS_CW)
        {
            const int c = idx / 255, k = idx - c * 255;
            const long long cwid = cw0 + c;
            unsigned v = 0;
            if (cwid < ncw)
            {
                const int f = (int)(cwid / I), b = (int)(cwid % I);
                v = (k < fb) ? 0u : data[(size_t)f * frame_stride + (size_t)(k - fb) * I + b];
                if (dualbasis)
                    v = from_dual[v];
            }
            cw[c][k] = (unsigned char)v;
        }
        __syncthreads();
        const int c = tid >> 5, r = tid & 31;
        sy_sh[c][r] = 0;
        if (r < nroots && cw0 + c < ncw)
        {
            const int fcr = nroots == 32 ? 112 : 120, gap = 11;
            const unsigned lr = (unsigned)((gap * (r + fcr)) % 255); // log of generator root r (reed-solomon.c:9-11)
            unsigned sy = 0;
            for (int k = 0; k < 255; k++)
            {
                if (sy)
                    sy = gf.exp[gf.log[sy] + lr];
                sy ^= cw[c][k];
            }
            sy_sh[c][r] = (unsigned char)sy;
            if (sy)
                nz[c] = 1;
        }
        __syncthreads();
        if (tid < RSS_CW && cw0 + tid < ncw)
        {
            clean[cw0 + tid] = nz[tid] ? 0 : 1;
            if (dirty)
            { // clean: no errors, nothing more to do; dirty: onto the decoder's list
                if (nz[tid])
                    dirty[1 + atomicAdd(dirty, 1)] = (int)(cw0 + tid);
                else
                    errors[cw0 + tid] = 0;
            }
        }
        // the syndromes themselves, for the decoder behind the screen (32 bytes per codeword, behind the flags)
        if (cw0 + c < ncw && (r & 3) == 0)
        {
            unsigned char *sp = clean + ((ncw + 15) / 16 * 16) + (size_t)(cw0 + c) * 32;
            *reinterpret_cast<unsigned *>(sp + r) = (unsigned)sy_sh[c][r] | ((unsigned)sy_sh[c][r + 1] << 8) | ((unsigned)sy_sh[c][r + 2] << 16) | ((unsigned)sy_sh[c][r + 3] << 24);
        }
    }

    // ---- the screen for I = 4 (every CCSDS 1020- / 1024-byte CADU): the four interleaved codewords of a frame as ONE packed dword per symbol position --------
    // Byte k of codeword b of a frame lies at frame[(k - fb) * 4 + b]: the dword at position k - fb holds symbol k of all four codewords. A lane (one generator
    // root, as above) runs Horner on the four codewords at once: S <- S * alpha^lr ^ symbols, four GF(256) bytes per register. Multiplying a byte by a CONSTANT
    // is linear over GF(2): x * c = Ta[x & 7] ^ Tb[(x >> 3) & 7] ^ Tc[x >> 6], three 8-entry byte tables = two registers each, looked up for four bytes at once by
    // v_perm_b32 (selector byte 0..7 picks that byte of the register pair). 11 VALU instructions per symbol position of FOUR codewords and no table in LDS on the
    // recurrence (k_rs_screen: log and exp look-ups per codeword and symbol, two dependent LDS reads each, bank conflicts above its LDS-active cycles). The dual
    // basis conversion (linear too) is applied the same way while the frame is staged. Same syndromes, flags and lists as k_rs_screen, byte for byte.
    struct Lin8
    {
        unsigned a0, a1, b0, b1, c0; // Ta[0..3], Ta[4..7], Tb[0..3], Tb[4..7], Tc[0..3]
    };
    __device__ __forceinline__ unsigned lin8_apply(const Lin8 &t, unsigned x)
    {
        const unsigned ia = x & 0x07070707u, ib = (x >> 3) & 0x07070707u, ic = (x >> 6) & 0x03030303u;
        return __builtin_amdgcn_perm(t.a1, t.a0, ia) ^ __builtin_amdgcn_perm(t.b1, t.b0, ib) ^ __builtin_amdgcn_perm(0u, t.c0, ic);
    }
    template <class F>
    __device__ __forceinline__ Lin8 lin8_make(F f)
    { // f(x): the linear map's value on byte x
        Lin8 t{0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++)
        {
            t.a0 |= (unsigned)f(i) << (8 * i);
            t.a1 |= (unsigned)f(4 + i) << (8 * i);
            t.b0 |= (unsigned)f(i << 3) << (8 * i);
            t.b1 |= (unsigned)f((4 + i) << 3) << (8 * i);
            t.c0 |= (unsigned)f(i << 6) << (8 * i);
        }
        return t;
    }
    constexpr int RSS4_FR = 8; // frames per block: 32 codewords, 256 threads (frame, root)
    __global__ __launch_bounds__(32 * RSS4_FR) void k_rs_screen4(const unsigned char *__restrict__ data, int nframes, int frame_stride, int dualbasis, int nroots, int fill_bytes,
                                                                 unsigned char *clean, const GfTables *tabs, int *errors, int *dirty)
    {
        __shared__ __attribute__((aligned(16))) unsigned stage[RSS4_FR][256]; // symbol position k of the frame's four codewords (position 255: padding)
        __shared__ __attribute__((aligned(4))) unsigned char sy_sh[RSS4_FR * 4][32];
        const int tid = (int)threadIdx.x;
        const long long f0 = (long long)blockIdx.x * RSS4_FR, ncw = (long long)nframes * 4, cw0 = f0 * 4;
        const int fb = fill_bytes < 0 ? 0 : fill_bytes;
        const Lin8 dual = lin8_make([&](int x) { return tabs->from_dual[x]; });
        for (int idx = tid; idx < RSS4_FR * 256; idx += 32 * RSS4_FR)
        {
            const int fi = idx >> 8, k = idx & 255;
            unsigned v = 0;
            if (f0 + fi < nframes && k >= fb && k < 255)
            {
                v = reinterpret_cast<const unsigned *>(data + (size_t)(f0 + fi) * frame_stride)[k - fb];
                if (dualbasis)
                    v = lin8_apply(dual, v);
            }
            stage[fi][k] = v;
        }
        const int fi = tid >> 5, r = tid & 31;
        unsigned S = 0;
        __syncthreads();
        if (r < nroots && f0 + fi < nframes)
        {
            const int fcr = nroots == 32 ? 112 : 120, gap = 11;
            const unsigned lr = (unsigned)((gap * (r + fcr)) % 255); // log of generator root r (reed-solomon.c:9-11)
            const Lin8 mul = lin8_make([&](int x) { return x ? tabs->exp[tabs->log[x] + lr] : (unsigned char)0; });
            const uint4 *row = reinterpret_cast<const uint4 *>(stage[fi]);
#pragma unroll 4
            for (int k4 = 0; k4 < 64; k4++)
            {
                const uint4 q = row[k4];
                S = lin8_apply(mul, S) ^ q.x;
                S = lin8_apply(mul, S) ^ q.y;
                S = lin8_apply(mul, S) ^ q.z;
                if (k4 < 63)
                    S = lin8_apply(mul, S) ^ q.w;
            }
        }
#pragma unroll
        for (int b = 0; b < 4; b++)
            sy_sh[fi * 4 + b][r] = (unsigned char)(S >> (8 * b));
        __syncthreads();
        { // 32 codewords x 8 dwords of syndromes, behind the flags (the layout k_rs reads)
            const int c = tid >> 3, d = tid & 7;
            if (cw0 + c < ncw)
            {
                unsigned char *sp = clean + ((ncw + 15) / 16 * 16) + (size_t)(cw0 + c) * 32;
                reinterpret_cast<unsigned *>(sp)[d] = reinterpret_cast<const unsigned *>(sy_sh[c])[d];
            }
        }
        if (tid < RSS4_FR * 4 && cw0 + tid < ncw)
        {
            unsigned any = 0;
#pragma unroll
            for (int d = 0; d < 8; d++)
                any |= reinterpret_cast<const unsigned *>(sy_sh[tid])[d];
            clean[cw0 + tid] = any ? 0 : 1;
            if (dirty)
            { // clean: no errors, nothing more to do; dirty: onto the decoder's list
                if (any)
                    dirty[1 + atomicAdd(dirty, 1)] = (int)(cw0 + tid);
                else
                    errors[cw0 + tid] = 0;
            }
        }
    }

    void launch_rs_only(uint8_t *data, int nframes, int frame_stride, int dualbasis, int I, int nroots, int fill_bytes, int *errors, hipStream_t st, uint8_t *clean_scratch)
    {
        const long long n = (long long)nframes * I;
        if (n <= 0)
            return;
        const GfTables *tabs = tables_for_current_device();
        int *dirty = nullptr;
        if (clean_scratch)
        {
            // scratch: [clean flag per codeword | 32 syndrome bytes per codeword | dirty count, dirty codeword ids]
            dirty = reinterpret_cast<int *>(clean_scratch + rs_scratch_list_offset(n));
            SD_HIP(hipMemsetAsync(dirty, 0, sizeof(int), st));
            ProfScope _ps("k_rs_screen", st);
            static const bool packed_ok = !(getenv("SDHIP_RS_SCREEN4") && atoi(getenv("SDHIP_RS_SCREEN4")) == 0);
            if (packed_ok && I == 4 && frame_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(data) & 3u) == 0)
                hipLaunchKernelGGL(k_rs_screen4, dim3((unsigned)((nframes + RSS4_FR - 1) / RSS4_FR)), dim3(32 * RSS4_FR), 0, st, data, nframes, frame_stride, dualbasis, nroots,
                                   fill_bytes, clean_scratch, tabs, errors, dirty);
            else
                hipLaunchKernelGGL(k_rs_screen, dim3((unsigned)((n + RSS_CW - 1) / RSS_CW)), dim3(32 * RSS_CW), 0, st, data, nframes, frame_stride, dualbasis, I, nroots, fill_bytes,
                                   clean_scratch, tabs, errors, dirty);
        }
        ProfScope _ps("k_rs", st);
        hipLaunchKernelGGL(k_rs, dim3((unsigned)((n + RS_THREADS - 1) / RS_THREADS)), dim3(RS_THREADS), 0, st, data, nframes, frame_stride, dualbasis, I, nroots,
                           fill_bytes, errors, tabs, clean_scratch, dirty);
    }

    // ---- frame extraction + derandomiser -----------------------------------------------------------
    __global__ __launch_bounds__(256) void k_extract(BitStream bs, FrameCfg fc, const FrameDesc *frames, int nframes, unsigned char *out, const GfTables *tabs)
    {
        const int f = (int)blockIdx.x;
        if (f >= nframes)
            return;
        const FrameDesc d = frames[f];
        unsigned char *o = out + (size_t)f * fc.cadu_bytes;
        const int nwords = (fc.cadu_bytes + 3) / 4;
        for (int w = (int)threadIdx.x; w < nwords; w += (int)blockDim.x)
        {
            unsigned v;
            if (w == 0)
                v = fc.asm_sync; // reset_frame writes the constant ASM, bpsk_ccsds_deframer.cpp:115-123
            else
            {
                v = stream32(bs, d.pos + (long long)(w - 1) * 32);
                if (d.inv)
                    v = ~v;
            }
            for (int q = 0; q < 4; q++)
            {
                const int k = w * 4 + q;
                if (k >= fc.cadu_bytes)
                    break;
                unsigned byte = (v >> (24 - 8 * q)) & 255u;
                if (fc.derand && !fc.derand_after_rs && k >= fc.derand_start)
                    byte ^= tabs->pn[(k - fc.derand_start) % 255];
                o[k] = (unsigned char)byte;
            }
        }
    }
    __global__ __launch_bounds__(256) void k_derand(unsigned char *frames, int nframes, int cadu_bytes, int derand_start, const GfTables *tabs)
    {
        const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= (long long)nframes * cadu_bytes)
            return;
        const int k = (int)(i % cadu_bytes);
        if (k >= derand_start)
            frames[i] ^= tabs->pn[(k - derand_start) % 255];
    }

    // rs_fill_bytes = -1 (the concatenated decoder's default): ReedSolomon's de/interleave loops run to 255 - (-1) = 256 bytes
    // (reedsolomon.cpp:145-156), one past its 255-byte buffer -- into odata[0], the next member -- and one codeblock byte past the
    // frame: byte b of the NEXT frame in the deframer's output buffer is read into odata[0], a successful decode then leaves the
    // first corrected message byte there (conventional basis), and interleave writes it back. When the deframer returned two
    // frames from one call (CADUs shorter than half a Viterbi buffer: the 2048 / 2072-bit pipelines) that next frame is real:
    // its first rs_i bytes -- sync marker bytes, rs_i <= 4 -- come out as those values. FrameDesc::pad marks a frame whose
    // successor came from the same call.
    __global__ __launch_bounds__(256) void k_rs_overrun(unsigned char *out, const FrameDesc *frames, int nframes, int cadu_bytes, int I, int dualbasis,
                                                         const int *errors, const GfTables *tabs)
    {
        const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        const int f = idx >> 2, b = idx & 3;
        if (f + 1 >= nframes || b >= I || !frames[f].pad || errors[(size_t)f * I + b] < 0)
            return;
        const unsigned v = out[(size_t)f * cadu_bytes + 4 + b];
        out[(size_t)(f + 1) * cadu_bytes + b] = dualbasis ? tabs->from_dual[v] : (unsigned char)v;
    }

    void launch_frames(const BitStream &bs, const FrameCfg &fc, const FrameDesc *frames, int nframes, uint8_t *out, int *errors, hipStream_t st, uint8_t *clean_scratch)
    {
        if (nframes <= 0)
            return;
        const GfTables *tabs = tables_for_current_device();
        {
            ProfScope _ps("k_extract", st);
            hipLaunchKernelGGL(k_extract, dim3(nframes), dim3(256), 0, st, bs, fc, frames, nframes, out, tabs);
        }
        if (fc.rs_i != 0)
            launch_rs_only(out + 4, nframes, fc.cadu_bytes, fc.rs_dualbasis, fc.rs_i, fc.rs_nroots, fc.rs_fill_bytes, errors, st, clean_scratch);
        if (fc.rs_i != 0 && fc.rs_fill_bytes == -1 && nframes > 1)
        {
            ProfScope _ps("k_rs_overrun", st);
            hipLaunchKernelGGL(k_rs_overrun, dim3((unsigned)((nframes * 4 + 255) / 256)), dim3(256), 0, st, out, frames, nframes, fc.cadu_bytes, fc.rs_i, fc.rs_dualbasis,
                               errors, tabs);
        }
        if (fc.derand && fc.derand_after_rs)
        {
            const long long n = (long long)nframes * fc.cadu_bytes;
            ProfScope _ps("k_derand", st);
            hipLaunchKernelGGL(k_derand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, nframes, fc.cadu_bytes, fc.derand_start, tabs);
        }
    }

    __global__ __launch_bounds__(256) void k_compact(const unsigned char *frames, const int *dst_index, int nframes, int cadu_bytes, unsigned char *out)
    {
        const int f = (int)blockIdx.x;
        if (f >= nframes)
            return;
        const int d = dst_index[f];
        if (d < 0)
            return;
        const unsigned char *s = frames + (size_t)f * cadu_bytes;
        unsigned char *o = out + (size_t)d * cadu_bytes;
        for (int k = (int)threadIdx.x; k < cadu_bytes; k += (int)blockDim.x)
            o[k] = s[k];
    }
    // The modules' frame filter (module_ccsds_conv_concat_decoder.cpp:183-196: with rs_usecheck a frame with an uncorrectable codeword is dropped) and the output slot
    // of every kept frame, on the device: the errors of nframes x I codewords used to cross PCIe for a host loop to do this (1.5 MB and ~100 us of host time per
    // 100 k frames, the device idle). Three small launches: a thread per frame counts its wave's kept frames (one ballot), one block scans the wave counts, a
    // thread per frame takes its slot = base + its wave's offset + its rank in the ballot. (A first version -- one block walking contiguous runs of frames --
    // was a chain of uncoalesced loads: 0.37 ms per 100 k frames, slower than the host loop it replaced.) info[0] = frames kept, info[1 + k] = the last frame's
    // errors[k] (the modules' rs_avg statistic).
    __device__ __forceinline__ bool rs_keep(const int *__restrict__ ferr, int f, int I, int rs_i, int usecheck)
    {
        bool valid = true;
        for (int k = 0; k < rs_i; k++)
            valid = valid && ferr[(size_t)f * I + k] != -1;
        return !usecheck || valid;
    }
    __global__ __launch_bounds__(256) void k_rs_keep(const int *__restrict__ ferr, int nframes, int I, int rs_i, int usecheck, int *wcount)
    {
        const int f = (int)(blockIdx.x * 256 + threadIdx.x);
        const bool keep = f < nframes && rs_keep(ferr, f, I, rs_i, usecheck);
        const unsigned long long m = __ballot(keep);
        if (lane_id() == 0)
            wcount[f >> 6] = __popcll(m);
    }
    __global__ __launch_bounds__(1024) void k_rs_filter(const int *__restrict__ ferr, int nframes, int I, int rs_i, int nwaves, int *wcount, int *info)
    { // exclusive scan of the wave counts, in place
        __shared__ int part[1024];
        const int t = (int)threadIdx.x;
        const int per = (nwaves + 1023) / 1024;
        const int a = t * per, b = a + per < nwaves ? a + per : nwaves;
        int c = 0;
        for (int w = a; w < b; w++)
            c += wcount[w];
        part[t] = c;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1)
        {
            const int v = t >= off ? part[t - off] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        int run = part[t] - c;
        for (int w = a; w < b; w++)
        {
            const int n = wcount[w];
            wcount[w] = run;
            run += n;
        }
        if (t == 1023)
            info[0] = part[1023];
        if (t < rs_i && t < 8 && nframes > 0)
            info[1 + t] = ferr[(size_t)(nframes - 1) * I + t];
    }
    __global__ __launch_bounds__(256) void k_rs_slots(const int *__restrict__ ferr, int nframes, int I, int rs_i, int usecheck, int out_base, const int *__restrict__ woff, int *dst)
    {
        const int f = (int)(blockIdx.x * 256 + threadIdx.x);
        const bool keep = f < nframes && rs_keep(ferr, f, I, rs_i, usecheck);
        const unsigned long long m = __ballot(keep);
        const int lane = lane_id();
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (f < nframes)
            dst[f] = keep ? out_base + woff[f >> 6] + rank : -1;
    }
    void launch_rs_filter(const int *ferr, int nframes, int I, int rs_i, int usecheck, int out_base, int *dst, int *info, int *wscratch, hipStream_t st)
    {
        if (nframes <= 0)
            return;
        ProfScope _ps("k_rs_filter", st);
        const int nblk = (nframes + 255) / 256, nwaves = nblk * 4;
        hipLaunchKernelGGL(k_rs_keep, dim3(nblk), dim3(256), 0, st, ferr, nframes, I, rs_i, usecheck, wscratch);
        hipLaunchKernelGGL(k_rs_filter, dim3(1), dim3(1024), 0, st, ferr, nframes, I, rs_i, nwaves, wscratch, info);
        hipLaunchKernelGGL(k_rs_slots, dim3(nblk), dim3(256), 0, st, ferr, nframes, I, rs_i, usecheck, out_base, wscratch, dst);
    }

    void launch_compact(const uint8_t *frames, const int *dst_index, int nframes, int cadu_bytes, uint8_t *out, hipStream_t st)
    {
        if (nframes <= 0)
            return;
        ProfScope _ps("k_compact", st);
        hipLaunchKernelGGL(k_compact, dim3(nframes), dim3(256), 0, st, frames, dst_index, nframes, cadu_bytes, out);
    }
} // namespace sdhip
