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
        const unsigned short *pos;  // [q][CNL][M] data bit of check (layer, check) slot c
        const unsigned char *cnc;   // [q] data bits per check of the layer
        const unsigned char *phase; // [q][M] phase of a check inside its layer (0 unless it shares a bit with a lower-numbered one)
        const unsigned char *nph;   // [q] phases of the layer
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
            const int par0 = g.K + g.M * i + j;
            const int par1 = i ? g.K + g.M * (i - 1) + j : g.K + (g.q - 1) * g.M + j - 1;
#pragma unroll
            for (int w = 0; w < DQ; w++)
                bw[w] = 4 * w < deg ? bnl_f[((size_t)i * DQ + w) * g.M + j] : 0u;
#pragma unroll
            for (int h = 0; h < 2 * DQ; h++)
            {
                unsigned v = 0;
#pragma unroll
                for (int b = 0; b < 2; b++)
                {
                    const int d = 2 * h + b;
                    if (d < deg)
                    {
                        const int n = d < cnt ? (int)g.pos[((size_t)i * g.CNL + d) * g.M + j] : (d == cnt ? par0 : par1);
                        v |= (unsigned)n << (16 * b);
                    }
                }
                nd[h] = v;
            }
        }
        __device__ __forceinline__ int node(int d) const { return (int)((nd[d >> 1] >> (16 * (d & 1))) & 0xFFFFu); }
        __device__ __forceinline__ void update(signed char *llr)
        {
            int min0 = 255, min1 = 255;
            unsigned signs = 0;
#pragma unroll
            for (int w = 0; w < DQ; w++)
#pragma unroll
                for (int b = 0; b < 4; b++)
                {
                    const int d = 4 * w + b;
                    if (d < deg)
                    {
                        const int inp = q8((int)llr[node(d)] - (int)(signed char)(bw[w] >> (8 * b)));
                        int mag = inp < -127 ? 127 : (inp < 0 ? -inp : inp); // vqabs
                        mag = mag > 0 ? mag - 1 : 0;                          // unsigned saturating - beta, beta = nearbyint(0.5 * 2) = 1
                        // mins[1] = min(mins[1], max(mins[0], mag)); mins[0] = min(mins[0], mag) (the first two: min / max of the pair)
                        const int hi = mag > min0 ? mag : min0;
                        min1 = hi < min1 ? hi : min1;
                        min0 = mag < min0 ? mag : min0;
                        signs ^= (unsigned)inp;
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
                        const int inp = q8((int)llr[n] - (int)(signed char)(bw[w] >> (8 * b)));
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
        __device__ __forceinline__ void store(const LdpcDev &g, unsigned *bnl_f, int i, int j) const
        {
#pragma unroll
            for (int w = 0; w < DQ; w++)
                if (4 * w < deg)
                    bnl_f[((size_t)i * DQ + w) * g.M + j] = bw[w];
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
        take(llr[g.K + g.M * i + j]);
        if (i)
            take(llr[g.K + g.M * (i - 1) + j]);
        else if (j)
            take(llr[g.K + (g.q - 1) * g.M + j - 1]);
        for (int c = 0; c < cnt; c++)
            take(llr[g.pos[((size_t)i * g.CNL + c) * g.M + j]]);
        return zero || neg;
    }

    constexpr int LDPC_THREADS = 384; // 360 checks of a layer, six waves
    // trial t of every frame. bad_prev / bad_cur: per frame flags of trial t-1 / t; updates[f]: update passes run so far.
    template <int DQ>
    __global__ __launch_bounds__(LDPC_THREADS) void k_ldpc_trial(LdpcDev g, signed char *frames, unsigned *bnl, int nframes, int batch, int t, const int *bad_prev,
                                                                 int *bad_cur, int *updates, int *any_bad)
    {
        __shared__ signed char llr[64800];
        const int f = (int)blockIdx.x, tid = (int)threadIdx.x;
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
        unsigned *bnl_f = bnl + (size_t)f * g.q * DQ * g.M;
        __syncthreads();
        if (active)
        {
            for (int i = 0; i < g.q; i++)
            {
                const int nph = g.nph[i];
                const int my = tid < g.M ? (int)g.phase[i * g.M + tid] : -1;
                LdpcCheck<DQ> ck;
                if (my >= 0)
                    ck.load(g, bnl_f, i, tid);
                for (int ph = 0; ph < nph; ph++)
                {
                    if (my == ph)
                        ck.update(llr);
                    __syncthreads();
                }
                if (my >= 0)
                    ck.store(g, bnl_f, i, tid);
            }
        }
        bool bad = false;
        if (tid < g.M)
            for (int i = 0; i < g.q && !bad; i++)
                bad = ldpc_check_bad(g, llr, i, tid);
        const int any = __syncthreads_or(bad ? 1 : 0);
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
        DevBuf<unsigned short> d_pos;
        DevBuf<unsigned char> d_cnc, d_phase, d_nph;
        DevBuf<unsigned> d_bnl;
        DevBuf<int> d_flags; // bad[2][nf] | updates[nf] | any
        DevBuf<signed char> d_frames;
        DevBuf<int> d_trials;
        int max_phases = 1, conflict_layers = 0;
        sdhip_ldpc_info info{};

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
            }
            const int DQ = (CNL + 2 + 3) / 4;
            d_pos.reserve(pos.size());
            d_cnc.reserve(q);
            d_nph.reserve(q);
            d_phase.reserve(phase.size());
            SD_HIP(hipMemcpy(d_pos.p, pos.data(), pos.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
            SD_HIP(hipMemcpy(d_cnc.p, cnc.data(), q, hipMemcpyHostToDevice));
            SD_HIP(hipMemcpy(d_nph.p, nph.data(), q, hipMemcpyHostToDevice));
            SD_HIP(hipMemcpy(d_phase.p, phase.data(), phase.size(), hipMemcpyHostToDevice));
            g = LdpcDev{M, N, K, R, q, CNL, DQ, d_pos.p, d_cnc.p, d_phase.p, d_nph.p};
            info.code_len = N;
            info.data_len = K;
            info.layers = q;
            info.links_total = tab->links_total;
            info.max_phases = max_phases;
            info.layers_with_shared_bits = conflict_layers;
            info.msg_bytes_per_frame = (uint64_t)q * DQ * M * 4;
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
            const size_t per = (size_t)g.q * g.DQ * g.M;
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
            return ran;
        }
        void launch_trial(signed char *d_fr, int nframes, int t, const int *prev, int *cur, int *upd, int *any)
        {
            const dim3 grid((unsigned)nframes), block(LDPC_THREADS);
#define SD_LDPC_CASE(D)                                                                                                                      \
    case D:                                                                                                                                  \
        hipLaunchKernelGGL((k_ldpc_trial<D>), grid, block, 0, stream, g, d_fr, d_bnl.p, nframes, cfg.batch, t, prev, cur, upd, any);          \
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
