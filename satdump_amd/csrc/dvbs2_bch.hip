// dvbs2_bch.hip -- DVB-S2 outer code on gfx950: the reference's BCH decoder (plugins/dvb_support/codings/dvb-s2/bbframe_bch.{h,cpp}:
// BBFrameBCH::decode -> CODE::BoseChaudhuriHocquenghemDecoder<NR, 1, MSG, GF>, bch/bose_chaudhuri_hocquenghem_decoder.hh:40-170, on
// CODE::ReedSolomonErrorCorrection, bch/reed_solomon_error_correction.hh: Berlekamp-Massey :218-272, LocationFinder / Artin-Schreier /
// Chien :33-137, Forney :139-216; field arithmetic bch/galois_field.hh) and the hard-decision repack in front of it
// (module_dvbs2_demod.cpp:262-266), bit for bit including the decoder's failure returns.
//
// Mapping. One workgroup per BBFRAME. Syndromes: 256 threads walk the frame's bytes; a set bit at polynomial exponent e adds
// alpha^(k e) to syndrome k -- only the ODD k are summed (NR / 2 table look-ups per set bit), the even ones follow as squares
// (r(alpha^2k) = r(alpha^k)^2 for a binary polynomial: the same field elements the reference's Horner loop arrives at). A frame whose
// syndromes are all zero -- every frame behind a converged LDPC decoder -- is done there. Otherwise lane 0 runs Berlekamp-Massey
// (<= 24 x 24 field operations), the whole workgroup the Chien search (65 535 / 16 383 evaluations) when the locator's degree is
// above 2, lane 0 Forney, the plausibility checks of the reference and the bit flips.
#include "common.h"
#include "../../include/sdhip.h"
#include <memory>
#include <vector>

namespace sdhip
{
    struct BchDev
    {
        int m, N;       // GF(2^m), N = 2^m - 1
        int NR, MSG;    // roots (2t) and message length of the full-length code (the template arguments of the reference's decoder)
        int kbch, nbch; // BBFrameBCH::kbch / nbch of this frame size and rate
        const unsigned short *LOG, *EXP, *IMAP;
    };
    typedef unsigned short u16;

    // ---- GF::Index / GF::Value arithmetic of galois_field.hh on uint16_t, operation for operation (LOG[0] = N, EXP[N] = 0 as in Tables())
    struct Gf
    {
        const u16 *LOG, *EXP;
        u16 N;
        __device__ __forceinline__ u16 imul(u16 a, u16 b) const
        { // Index * Index
            const u16 tmp = (u16)(a + b);
            return (u16)((int)N - (int)a <= (int)b ? (u16)(tmp - N) : tmp);
        }
        __device__ __forceinline__ u16 idiv(u16 a, u16 b) const
        { // Index / Index
            const u16 tmp = (u16)(a - b);
            return a < b ? (u16)(tmp + N) : tmp;
        }
        __device__ __forceinline__ u16 mul(u16 a, u16 b) const { return (!a || !b) ? (u16)0 : EXP[imul(LOG[a], LOG[b])]; } // Value * Value
        __device__ __forceinline__ u16 div(u16 a, u16 b) const { return !a ? (u16)0 : EXP[idiv(LOG[a], LOG[b])]; }         // Value / Value
        __device__ __forceinline__ u16 mul_vi(u16 a, u16 bi) const { return !a ? (u16)0 : EXP[imul(LOG[a], bi)]; }          // Value * Index
    };
    // x mod (2^m - 1) for x < 2^31
    __device__ __forceinline__ unsigned mod_n(unsigned x, int m, unsigned N)
    {
        x = (x & N) + (x >> m);
        x = (x & N) + (x >> m);
        return x >= N ? x - N : x;
    }

    constexpr int BCH_THREADS = 256, BCH_NRMAX = 24;
    __global__ __launch_bounds__(BCH_THREADS) void k_bch_decode(BchDev g, unsigned char *frames, int stride, int nframes, int *corrections)
    {
        __shared__ unsigned part[BCH_NRMAX / 2][BCH_THREADS];
        __shared__ u16 syn[BCH_NRMAX], locator[BCH_NRMAX + 1], locs[BCH_NRMAX];
        __shared__ int sh_deg, sh_cnt, sh_mode;
        const int f = (int)blockIdx.x, t = (int)threadIdx.x;
        unsigned char *fr = frames + (size_t)f * stride;
        const Gf gf{g.LOG, g.EXP, (u16)g.N};
        const int n = g.nbch, nodd = g.NR / 2;
        // ---- syndromes S_k = r(alpha^k), k = 1 .. NR (compute_syndromes, bose_chaudhuri_hocquenghem_decoder.hh:57-79; FCR = 1)
        unsigned acc[BCH_NRMAX / 2];
#pragma unroll
        for (int q = 0; q < BCH_NRMAX / 2; q++)
            acc[q] = 0;
        for (int b = t; b < n / 8; b += BCH_THREADS)
        {
            unsigned v = fr[b];
            while (v)
            {
                const int u = 31 - __clz((int)v); // bit u of the byte (7 = first in time, big endian: bitman.cpp get_be_bit)
                v &= ~(1u << u);
                const unsigned e = (unsigned)(n - 1 - (8 * b + (7 - u)));
                const unsigned e1 = mod_n(e, g.m, (unsigned)g.N);
#pragma unroll
                for (int q = 0; q < BCH_NRMAX / 2; q++)
                    if (q < nodd)
                        acc[q] ^= g.EXP[mod_n((unsigned)(2 * q + 1) * e1, g.m, (unsigned)g.N)];
            }
        }
#pragma unroll
        for (int q = 0; q < BCH_NRMAX / 2; q++)
            part[q][t] = acc[q];
        __syncthreads();
        for (int s = BCH_THREADS / 2; s > 0; s >>= 1)
        {
            if (t < s)
#pragma unroll
                for (int q = 0; q < BCH_NRMAX / 2; q++)
                    part[q][t] ^= part[q][t + s];
            __syncthreads();
        }
        if (t == 0)
        {
            for (int q = 0; q < nodd; q++)
                syn[2 * q] = (u16)part[q][0];          // S_(2q+1)
            for (int k = 2; k <= g.NR; k += 2)          // S_k = S_(k/2)^2
                syn[k - 1] = gf.mul(syn[k / 2 - 1], syn[k / 2 - 1]);
            int nonzero = 0;
            for (int i = 0; i < g.NR; i++)
                nonzero += syn[i] != 0;
            sh_mode = 0; // 0: finished (result written), 1: Chien search wanted
            sh_cnt = 0;
            if (!nonzero)
                corrections[f] = 0;
            else
            {
                // ---- ReedSolomonErrorCorrection::operator() without erasures: locator = 1, Berlekamp-Massey (:218-272)
                const int NR = g.NR;
                u16 C[BCH_NRMAX + 1], B[BCH_NRMAX + 1], T[BCH_NRMAX + 1];
                for (int i = 0; i <= NR; i++)
                    C[i] = B[i] = (u16)(i == 0);
                int L = 0;
                for (int nn = 0, mm = 1; nn < NR; ++nn)
                {
                    u16 d = syn[nn];
                    for (int i = 1; i <= L; ++i)
                        d ^= gf.mul(C[i], syn[nn - i]);
                    if (!d)
                        ++mm;
                    else
                    {
                        for (int i = 0; i < mm; ++i)
                            T[i] = C[i];
                        for (int i = mm; i <= NR; ++i)
                            T[i] = (u16)(gf.mul(d, B[i - mm]) ^ C[i]); // fma(d, B[i - m], C[i])
                        if (2 * L <= nn)
                        {
                            L = nn + 1 - L;
                            for (int i = 0; i <= NR; ++i)
                                B[i] = gf.div(C[i], d);
                            mm = 1;
                        }
                        else
                            ++mm;
                        for (int i = 0; i <= NR; ++i)
                            C[i] = T[i];
                    }
                }
                int deg = L, res = 1;
                while (!C[deg])
                    if (--deg < 0)
                    {
                        res = -1;
                        break;
                    }
                if (res < 0)
                    corrections[f] = -1;
                else
                {
                    for (int i = 0; i <= NR; i++)
                        locator[i] = C[i];
                    sh_deg = deg;
                    // ---- LocationFinder (:101-137)
                    if (deg == 1)
                    {
                        locs[0] = gf.idiv(gf.idiv(g.LOG[C[0]], g.LOG[C[1]]), 1);
                        sh_cnt = 1;
                        sh_mode = 2;
                    }
                    else if (deg == 2)
                    {
                        sh_mode = 2;
                        if (!C[1] || !C[0])
                            sh_cnt = 0;
                        else
                        {
                            const u16 a = C[2], b = C[1], c = C[0];
                            const u16 ba = gf.div(b, a), R = g.IMAP[gf.div(gf.mul(a, c), gf.mul(b, b))];
                            if (!R)
                                sh_cnt = 0;
                            else
                            {
                                const u16 x0 = gf.mul(ba, R);
                                locs[0] = gf.idiv(g.LOG[x0], 1);
                                locs[1] = gf.idiv(g.LOG[(u16)(x0 ^ ba)], 1);
                                sh_cnt = 2;
                            }
                        }
                    }
                    else
                        sh_mode = 1;
                }
            }
        }
        __syncthreads();
        if (sh_mode == 0)
            return;
        if (sh_mode == 1)
        { // ---- Chien::search (:33-57): position i is a root iff sum_j locator[j] alpha^(j (i + 1)) = 0
            const int deg = sh_deg;
            u16 lg[BCH_NRMAX + 1];
            for (int j = 1; j <= deg; j++)
                lg[j] = g.LOG[locator[j]];
            for (int i = t; i < g.N; i += BCH_THREADS)
            {
                unsigned sum = locator[0];
                for (int j = 1; j <= deg; j++)
                    if (locator[j])
                        sum ^= g.EXP[mod_n((unsigned)lg[j] + mod_n((unsigned)j * (unsigned)(i + 1), g.m, (unsigned)g.N), g.m, (unsigned)g.N)];
                if (!sum)
                {
                    const int slot = atomicAdd(&sh_cnt, 1);
                    if (slot < BCH_NRMAX)
                        locs[slot] = (u16)i;
                }
            }
            __syncthreads();
        }
        if (t != 0)
            return;
        int count = sh_cnt < BCH_NRMAX ? sh_cnt : BCH_NRMAX;
        const int deg = sh_deg, NR = g.NR;
        if (sh_mode == 1)
            for (int i = 1; i < count; i++) // ascending positions, the order the reference's sequential search finds them in
                for (int j = i; j > 0 && locs[j - 1] > locs[j]; j--)
                {
                    const u16 x = locs[j];
                    locs[j] = locs[j - 1];
                    locs[j - 1] = x;
                }
        if (count < deg)
        {
            corrections[f] = -1;
            return;
        }
        // ---- Forney (:139-216), FCR = 1
        u16 ev[BCH_NRMAX], mag[BCH_NRMAX];
        const int tmpd = count < NR - 1 ? count : NR - 1;
        int evd = -1;
        for (int i = 0; i <= tmpd; ++i)
        {
            u16 e = gf.mul(syn[i], locator[0]);
            for (int j = 1; j <= i; ++j)
                e ^= gf.mul(syn[i - j], locator[j]);
            ev[i] = e;
            if (e)
                evd = i;
        }
        for (int i = 0; i < count; ++i)
        {
            const u16 root = gf.imul(locs[i], 1);
            u16 tmp = root;
            u16 eval = evd >= 0 ? ev[0] : (u16)0;
            if (evd < 0)
                eval = ev[0]; // evaluator[0] is read whatever the degree (compute_magnitudes :172)
            for (int j = 1; j <= evd; ++j)
            {
                eval ^= gf.mul_vi(ev[j], tmp);
                tmp = gf.imul(tmp, root);
            }
            if (!eval)
            {
                mag[i] = 0;
                continue;
            }
            u16 deriv = locator[1];
            const u16 root2 = gf.imul(root, root);
            u16 tmp2 = root2;
            for (int j = 3; j <= count; j += 2)
            {
                deriv ^= gf.mul_vi(locator[j], tmp2);
                tmp2 = gf.imul(tmp2, root2);
            }
            mag[i] = g.EXP[gf.idiv(g.LOG[eval], g.LOG[deriv])];
        }
        if (count <= 0)
        {
            corrections[f] = count;
            return;
        }
        // ---- BoseChaudhuriHocquenghemDecoder::operator() :128-168 (data_len = kbch, K = MSG)
        for (int i = 0; i < count; ++i)
            if ((int)locs[i] < g.MSG - g.kbch)
            {
                corrections[f] = -1;
                return;
            }
        for (int i = 0; i < count; ++i)
            if (1 < (int)mag[i])
            {
                corrections[f] = -1;
                return;
            }
        int cc = 0;
        for (int i = 0; i < count; ++i)
        {
            const int idx = (int)locs[i] + g.kbch - g.MSG; // bit of the frame (data, then parity: consecutive here)
            if (mag[i])
                fr[idx / 8] ^= (unsigned char)(1u << (7 - idx % 8)); // xor_be_bit
            cc += mag[i] != 0;
        }
        corrections[f] = cc;
    }

    // hard decisions of the first `nbits` soft bits of every frame, MSB first (module_dvbs2_demod.cpp:262-266: bit = soft < 0)
    __global__ __launch_bounds__(256) void k_s2_pack(const signed char *soft, int soft_stride, int nbits, unsigned char *out, int out_stride, int nframes)
    {
        const int f = (int)blockIdx.y, b = (int)(blockIdx.x * 256 + threadIdx.x);
        if (f >= nframes || b >= nbits / 8)
            return;
        const unsigned long long v = *reinterpret_cast<const unsigned long long *>(soft + (size_t)f * soft_stride + 8 * (size_t)b);
        unsigned r = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            r |= (unsigned)((v >> (8 * k + 7)) & 1ull) << (7 - k);
        out[(size_t)f * out_stride + b] = (unsigned char)r;
    }

    // dvbs2::BBFrameDescrambler::work (bbframe_descramble.cpp:117-139): the first kbch / 8 bytes of a frame XOR the PRBS 1 + x^14 + x^15
    // started from 0x4A80 (the sequence is built on the host like init() builds it)
    __global__ __launch_bounds__(256) void k_bb_descramble(unsigned char *frames, int stride, int nbytes, int nframes, const unsigned char *seq)
    {
        const int f = (int)blockIdx.y, b = (int)(blockIdx.x * 256 + threadIdx.x);
        if (f < nframes && b < nbytes)
            frames[(size_t)f * stride + b] ^= seq[b];
    }

    // dvbs2::S2Deinterleaver::deinterleave (codings/dvb-s2/s2_deinterleaver.cpp:24-145): the demapper's soft bits symbol by symbol ->
    // the code word's bit order. QPSK: the two bits of a symbol swapped; 8PSK / 16APSK / 32APSK: bit c of symbol j goes to column c (rows =
    // frame / bits per symbol), 8PSK rate 3/5 with the columns in reverse order. Thread per output byte.
    __global__ __launch_bounds__(256) void k_s2_deinterleave(const signed char *in, signed char *out, int frame_len, int mod_bits, int reversed, int nframes)
    {
        const int f = (int)blockIdx.y, o = (int)(blockIdx.x * 256 + threadIdx.x);
        if (f >= nframes || o >= frame_len)
            return;
        const signed char *fi = in + (size_t)f * frame_len;
        int src;
        if (mod_bits == 2)
            src = o ^ 1;
        else
        {
            const int rows = frame_len / mod_bits, col = o / rows, j = o - col * rows;
            const int c = reversed ? mod_bits - 1 - col : col; // which bit of the symbol this column holds
            src = j * mod_bits + c;
        }
        out[(size_t)f * frame_len + o] = fi[src];
    }

    struct BchEngine
    {
        sdhip_bch_cfg cfg;
        hipStream_t stream = nullptr;
        BchDev g{};
        DevBuf<unsigned short> d_log, d_exp, d_imap;
        DevBuf<unsigned char> d_frames, d_prbs;
        DevBuf<int> d_corr;
        explicit BchEngine(const sdhip_bch_cfg &c) : cfg(c)
        {
            // BBFrameBCH::BBFrameBCH, bbframe_bch.cpp:40-186: kbch / nbch / code per frame size and rate (C7_8 has none)
            static const int kn[12] = {16008, 21408, 25728, 32208, 38688, 43040, 48408, 51648, 53840, 0, 57472, 58192};
            static const int nn[12] = {16200, 21600, 25920, 32400, 38880, 43200, 48600, 51840, 54000, 0, 57600, 58320};
            static const int tn[12] = {12, 12, 12, 12, 12, 10, 12, 12, 10, 0, 8, 8};
            static const int ks[12] = {3072, 5232, 6312, 7032, 9552, 10632, 11712, 12432, 13152, 0, 14232, 0};
            static const int ns[12] = {3240, 5400, 6480, 7200, 9720, 10800, 11880, 12600, 13320, 0, 14400, 0};
            if (c.rate < 0 || c.rate > 11 || (c.framesize != 0 && c.framesize != 1))
                throw HipError("dvbs2 bch: unknown frame size / code rate");
            int poly;
            if (c.framesize == 0)
            {
                g.kbch = kn[c.rate];
                g.nbch = nn[c.rate];
                g.NR = 2 * tn[c.rate];
                g.m = 16;
                poly = 0x1002D; // GF_NORMAL, bbframe_bch.h:47
                g.MSG = 65535 - 16 * tn[c.rate]; // 65343 / 65375 / 65407, bbframe_bch.h:50-52
            }
            else
            {
                g.kbch = ks[c.rate];
                g.nbch = ns[c.rate];
                g.NR = 24;
                g.m = 14;
                poly = 0x402B; // GF_SHORT, bbframe_bch.h:49
                g.MSG = 16215;
            }
            if (g.kbch == 0)
                throw HipError("dvbs2 bch: no code for this frame size / code rate");
            g.N = (1 << g.m) - 1;
            SD_HIP(hipSetDevice(c.device));
            SD_HIP(hipStreamCreate(&stream));
            // GF::Tables (galois_field.hh:104-123) and the Artin-Schreier map (reed_solomon_error_correction.hh:65-86)
            const int Q = 1 << g.m, N = g.N;
            std::vector<unsigned short> lg(Q), ex(Q), im(Q, 0);
            ex[N] = 0;
            lg[0] = (unsigned short)N;
            unsigned a = 1;
            for (int i = 0; i < N; ++i)
            {
                ex[i] = (unsigned short)a;
                lg[a] = (unsigned short)i;
                a = (a & (unsigned)(Q >> 1)) ? (((a << 1) ^ (unsigned)poly) & (unsigned)(Q - 1)) : (a << 1);
            }
            auto mul = [&](unsigned x, unsigned y) -> unsigned {
                if (!x || !y)
                    return 0;
                unsigned s = lg[x] + lg[y];
                return ex[s >= (unsigned)N ? s - N : s];
            };
            for (int i = 2; i < N; i += 2)
            {
                const unsigned xxx = mul(i, i) ^ (unsigned)i;
                if (xxx == (unsigned)N)
                    continue;
                im[xxx] = (unsigned short)i;
            }
            d_log.reserve(Q);
            d_exp.reserve(Q);
            d_imap.reserve(Q);
            SD_HIP(hipMemcpy(d_log.p, lg.data(), Q * 2, hipMemcpyHostToDevice));
            SD_HIP(hipMemcpy(d_exp.p, ex.data(), Q * 2, hipMemcpyHostToDevice));
            SD_HIP(hipMemcpy(d_imap.p, im.data(), Q * 2, hipMemcpyHostToDevice));
            g.LOG = d_log.p;
            g.EXP = d_exp.p;
            g.IMAP = d_imap.p;
            // BBFrameDescrambler::init, bbframe_descramble.cpp:117-131
            std::vector<unsigned char> seq(64800 / 8, 0);
            int sr = 0x4A80;
            for (int i = 0; i < 64800; i++)
            {
                const int b = (sr ^ (sr >> 1)) & 1;
                seq[i / 8] |= (unsigned char)(b << (7 - (i % 8)));
                sr >>= 1;
                if (b)
                    sr |= 0x4000;
            }
            d_prbs.reserve(seq.size());
            SD_HIP(hipMemcpy(d_prbs.p, seq.data(), seq.size(), hipMemcpyHostToDevice));
        }
        int descramble_dev(unsigned char *d_fr, int nframes, int stride)
        {
            SD_HIP(hipSetDevice(cfg.device));
            if (nframes <= 0)
                return 0;
            ProfScope _ps("k_bb_descramble", stream);
            hipLaunchKernelGGL(k_bb_descramble, dim3((unsigned)((g.kbch / 8 + 255) / 256), (unsigned)nframes), dim3(256), 0, stream, d_fr, stride, g.kbch / 8, nframes, d_prbs.p);
            SD_HIP(hipStreamSynchronize(stream));
            return 0;
        }
        ~BchEngine()
        {
            if (stream)
                (void)hipStreamDestroy(stream);
        }
        int decode_dev(unsigned char *d_fr, int nframes, int stride, int *d_corrections)
        {
            SD_HIP(hipSetDevice(cfg.device));
            if (nframes <= 0)
                return 0;
            if (stride < g.nbch / 8)
                throw HipError("dvbs2 bch: frame stride shorter than nbch / 8");
            {
                ProfScope _ps("k_bch_decode", stream);
                hipLaunchKernelGGL(k_bch_decode, dim3((unsigned)nframes), dim3(BCH_THREADS), 0, stream, g, d_fr, stride, nframes, d_corrections);
            }
            SD_HIP(hipStreamSynchronize(stream));
            return 0;
        }
        int decode_host(unsigned char *frames, int nframes, int stride, int *corrections)
        {
            SD_HIP(hipSetDevice(cfg.device));
            if (nframes <= 0)
                return 0;
            d_frames.reserve((size_t)nframes * stride);
            d_corr.reserve(nframes);
            SD_HIP(hipMemcpyAsync(d_frames.p, frames, (size_t)nframes * stride, hipMemcpyHostToDevice, stream));
            decode_dev(d_frames.p, nframes, stride, d_corr.p);
            SD_HIP(hipMemcpyAsync(frames, d_frames.p, (size_t)nframes * stride, hipMemcpyDeviceToHost, stream));
            SD_HIP(hipMemcpyAsync(corrections, d_corr.p, (size_t)nframes * sizeof(int), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            return 0;
        }
        int pack_dev(const signed char *d_soft, int soft_stride, int nframes, unsigned char *d_out, int out_stride)
        {
            SD_HIP(hipSetDevice(cfg.device));
            if (nframes <= 0)
                return 0;
            if ((soft_stride & 7) || (reinterpret_cast<uintptr_t>(d_soft) & 7))
                throw HipError("dvbs2 pack: soft frames must be 8-byte aligned");
            ProfScope _ps("k_s2_pack", stream);
            hipLaunchKernelGGL(k_s2_pack, dim3((unsigned)((g.nbch / 8 + 255) / 256), (unsigned)nframes), dim3(256), 0, stream, d_soft, soft_stride, g.nbch, d_out, out_stride,
                               nframes);
            SD_HIP(hipStreamSynchronize(stream));
            return 0;
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
    void *sdhip_bch_create(const sdhip_bch_cfg *cfg)
    {
        SD_GUARD_BEGIN
        return new BchEngine(*cfg);
        SD_GUARD_END(nullptr)
    }
    void sdhip_bch_destroy(void *h) { delete static_cast<BchEngine *>(h); }
    int sdhip_bch_dims(void *h, int *kbch, int *nbch)
    {
        SD_GUARD_BEGIN
        *kbch = static_cast<BchEngine *>(h)->g.kbch;
        *nbch = static_cast<BchEngine *>(h)->g.nbch;
        return 0;
        SD_GUARD_END(-1)
    }
    int sdhip_bch_decode_dev(void *h, uint8_t *d_frames, int nframes, int stride, int *d_corrections)
    {
        SD_GUARD_BEGIN
        return static_cast<BchEngine *>(h)->decode_dev(d_frames, nframes, stride, d_corrections);
        SD_GUARD_END(-1)
    }
    int sdhip_bch_decode(void *h, uint8_t *frames, int nframes, int stride, int *corrections)
    {
        SD_GUARD_BEGIN
        return static_cast<BchEngine *>(h)->decode_host(frames, nframes, stride, corrections);
        SD_GUARD_END(-1)
    }
    int sdhip_s2_deinterleave_dev(int device, int constellation, int framesize, int rate, const int8_t *d_in, int8_t *d_out, int nframes)
    {
        SD_GUARD_BEGIN
        if (constellation < 0 || constellation > 3 || (framesize != 0 && framesize != 1))
            throw HipError("dvbs2 deinterleaver: unknown constellation / frame size");
        if (nframes <= 0)
            return 0;
        SD_HIP(hipSetDevice(device));
        const int frame_len = framesize == 0 ? 64800 : 16200, mod_bits = constellation + 2;
        const int reversed = (constellation == 1 && rate == 4) ? 1 : 0; // MOD_8PSK with C3_5, s2_deinterleaver.cpp:45-50
        {
            ProfScope _ps("k_s2_deinterleave", nullptr);
            hipLaunchKernelGGL(k_s2_deinterleave, dim3((unsigned)((frame_len + 255) / 256), (unsigned)nframes), dim3(256), 0, nullptr, reinterpret_cast<const signed char *>(d_in),
                               reinterpret_cast<signed char *>(d_out), frame_len, mod_bits, reversed, nframes);
        }
        SD_HIP(hipDeviceSynchronize());
        return 0;
        SD_GUARD_END(-1)
    }
    int sdhip_bb_descramble_dev(void *h, uint8_t *d_frames, int nframes, int stride)
    {
        SD_GUARD_BEGIN
        return static_cast<BchEngine *>(h)->descramble_dev(d_frames, nframes, stride);
        SD_GUARD_END(-1)
    }
    int sdhip_s2_pack_dev(void *h, const int8_t *d_soft, int soft_stride, int nframes, uint8_t *d_out, int out_stride)
    {
        SD_GUARD_BEGIN
        return static_cast<BchEngine *>(h)->pack_dev(reinterpret_cast<const signed char *>(d_soft), soft_stride, nframes, d_out, out_stride);
        SD_GUARD_END(-1)
    }
}
