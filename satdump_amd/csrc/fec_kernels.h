// fec_kernels.h -- launch-side declarations of the CCSDS FEC kernels (gfx950).
#pragma once
#include "common.h"
#include <vector>

namespace sdhip
{
    // How the k=7 r=1/2 decoder's symbol stream is produced from a block of soft bytes.
    //   mode 0 = viterbi::Viterbi1_2  (src-core/common/codings/viterbi/viterbi_1_2.cpp:92-98)
    //   mode 1 = viterbi::Viterbi3_4 MetOp depuncture (viterbi_3_4.cpp:84-105,150-154); with fy = 1 the FengYun one (:58-78)
    struct VitCfg
    {
        int mode;
        int B;        // soft bytes per block (d_buffer_size / BUFFER_SIZE)
        int F;        // decoded bits per block (cc_decoder frame size)
        int nber;     // bits re-encoded for the BER estimate (1024 / 1536)
        int pre_swap; // module-level rotate_soft(.., PHASE_0, true) (module_ccsds_conv_concat_decoder.cpp:149-150)
        int iq_swap;  // d_iq_swap
        int phase;    // d_phase (0,1,2,3 = 0/90/180/270 deg)
        int shift;    // d_shift
        int stride;   // soft bytes from one block to the next; 0 = B. Smaller than B: overlapping blocks -- the windows of a
                      // sliding buffer (generic punctured rates: a decoder block reads its B symbols plus 12 of the next one)
        int nenc;     // bits the BER re-encoder runs per block (its register carries over from there); 0 = nber
        int fy;       // mode 1 only: Viterbi3_4's fymode (viterbi_3_4.cpp:58-78,115) -- the punctured pair keeps its symbol order
                      // (128, in[0], in[1], 128 where MetOp has 128, in[1], in[0], 128) and the lock search tries phase 0 only
    };
    __host__ __device__ inline long long vit_stride(const VitCfg &c) { return c.stride > 0 ? c.stride : c.B; }

    constexpr int VIT_PREPASS = 192;  // steps of the previous block replayed to speculate a start state
    constexpr int VIT_TB_OVERLAP = 96; // extra traceback steps of a speculative traceback segment

    // Per-block control / result words of k_vit_decode.
    struct VitBlockIO
    {
        int start_in;   // >=0: start state to use; -1: speculate from the previous block's tail; -2: unbiased first block
        int start_used; // start state actually used (or -2)
        int ret_state;  // CCDecoder::work's chained start state for the NEXT block
        int end_state;  // find_endstate()
        int tb_fallback;// 1 if the segment-parallel traceback certificate failed and the serial path ran
        int ber_err;    // BER estimate numerator  (filled by k_vit_ber)
        int ber_tot;    // BER estimate denominator (non-erased symbols)
        int pad;
    };

    // Decode `nblk` consecutive blocks: block j reads soft + (first_block + j) * cfg.B.
    //   io[j]       : control/result words
    //   decisions   : scratch, nblk * (F+6) * 8 bytes (lane-order ACS ballots)
    //   vbits       : packed decoded bits, nblk * words_per_block(F) uint32 (MSB-first byte stream)
    //   list        : optional device array of nblk block indices (relative to first_block) to decode instead of 0..nblk);
    //                 io / vbits are indexed by block, the decision scratch by position in the launch
    void launch_vit_decode(const VitCfg &cfg, const int8_t *soft, int64_t first_block, int nblk, VitBlockIO *io, uint64_t *decisions, uint32_t *vbits,
                           hipStream_t st, const int *list = nullptr);


    // ---- packed lane-per-segment decoder (k_vit2_*) ----------------------------------------------------------
    // Same contract as launch_vit_decode (io[] semantics, packed vbits), different mapping: every block is cut into
    // segments of VIT2_SEG trellis steps and ONE LANE runs one segment with all 64 path metrics in 32 VGPRs
    // (two 16-bit metrics per register, v_pk_add_u16 / v_pk_min_u16), see fec_kernels.hip. Segment g > 0 starts
    // from neutral metrics VIT2_WARM steps early; its exactness certificate is "metric vector after the warm-up ==
    // metric vector the previous segment ended with" (the decoder is a deterministic function of that vector).
    // A block whose certificate fails comes back with io[j].tb_fallback == 2 and must be decoded again with
    // launch_vit_decode(start_in = io[j].start_used).
    constexpr int VIT2_SEG = 512;   // trellis steps per lane
#ifndef SDHIP_VIT2_WARM
#define SDHIP_VIT2_WARM 200
#endif
    constexpr int VIT2_WARM = SDHIP_VIT2_WARM;  // warm-up steps (multiple of 8)
    struct Vit2Work
    {
        DevBuf<uint16_t> symu;      // per block: [VIT2_WARM prologue | F+6 steps | pad] unsigned symbol pairs (s0 | s1 << 8)
        DevBuf<uint64_t> dec;       // decisions [step in segment][unit]
        DevBuf<uint32_t> specx, endx; // 32 packed metric registers per unit
        DevBuf<int> entry, exitst;  // traceback hand-off states per unit
    };
    inline bool vit2_supported(const VitCfg &cfg) { return cfg.F >= 2 * VIT2_SEG && cfg.F % VIT2_SEG == 0; }
    void launch_vit_decode2(const VitCfg &cfg, const int8_t *soft, int64_t first_block, int nblk, VitBlockIO *io, uint32_t *vbits, Vit2Work &w,
                            hipStream_t st);

    // BER estimate of every block (viterbi_1_2.cpp:101-102 / viterbi_3_4.cpp:156-157): re-encode the first
    // nber decoded bits (encoder register chained through the previous block, enc_state_in for block 0) and
    // compare with the hard decisions of the input symbols. Also returns the encoder register after the last block.
    void launch_vit_ber(const VitCfg &cfg, const int8_t *soft, int64_t first_block, int nblk, const uint32_t *vbits, unsigned enc_state_in, VitBlockIO *io,
                        hipStream_t st);

    // Lock search on ONE block (Viterbi1_2::work IDLE branch, viterbi_1_2.cpp:54-88; Viterbi3_4: viterbi_3_4.cpp:112-148).
    struct VitSearchState
    {
        // persistent state of cc_decoder_ber / cc_encoder_ber / ber_decoded_buffer
        int ber_first;        // 1 until cc_decoder_ber has decoded once (unbiased metrics)
        int ber_start;        // chained start state of cc_decoder_ber
        unsigned enc_state;   // cc_encoder_ber shift register (low bits)
        uint8_t tail[16];     // first 13 bytes of ber_decoded_buffer (mode 0 overrun source)
        // results, one per candidate in reference order
        int ncand;
        int err[16], tot[16];
    };
    // candidates: mode 0: for s in [0, n_swap) for phase in phases[] for shift in {0,1}; mode 1: phase in {0,1} x shift in {0,1} (fy: phase 0 only)
    void launch_vit_search(const VitCfg &cfg, const int8_t *soft, int64_t block, int n_swap, const int *phases, int nphases, VitSearchState *d_state,
                           hipStream_t st);

    // ---- fengyun_ahrpt_decoder (plugins/fengyun3_support/fengyun3/module_fengyun_ahrpt_decoder.cpp:58-110) ----
    // The module's rail split (:62-68): after rotate_soft(.., PHASE_0, iq_invert = true) -- -128 -> -127, I and Q exchanged -- rail 0 takes byte 0 and
    // rail 1 byte 1 of pair i + shift (rail 1 complemented, ~x, when invert_second is set) for i in [0, 8192) of every 16384-byte block. With shift = 1
    // the module reads pair 8192 of its 8192-pair buffer; that pair is taken as (0, 0) here.
    // mpt (fengyun_mpt_decoder, module_fengyun_mpt_decoder.cpp:62-69): rail 1 always complemented, and both rails through rotate_soft(.., PHASE_0, true) once
    // more: -128 -> -127, then the bytes of every pair of the RAIL exchanged.
    void launch_fy_rails(const int8_t *soft, int64_t first_block, int nblk, int shift, int invert_second, int8_t *rail0, int8_t *rail1, hipStream_t st, int mpt = 0);
    // FengyunDiff::work2 (fengyun3/diff.cpp:49-78) over nblk blocks of bits_per_rail decoded bits of the two rails (packed MSB first, wpb_rail words per
    // block): x = the rail that is in1, y = in2; (x_prev, y_prev) = the pair in front of block 0. Writes 2 * bits_per_rail bits per block into out (wpb_out
    // words per block), the stream the deframer reads.
    void launch_fy_diff(const uint32_t *x, const uint32_t *y, int nblk, int bits_per_rail, int wpb_rail, unsigned x_prev, unsigned y_prev, uint32_t *out, int wpb_out,
                        hipStream_t st);

    // ---- generic punctured rates (depunc.h): per input position of the period, one or two depunctured symbols
    struct PuncPat
    {
        int n;            // period = numstates
        unsigned two;     // bit p: position p emits two symbols (the received one and an erasure)
        unsigned lead128; // bit p: the erasure comes first
        float berscale;   // get_berscale()
    };
    inline PuncPat punc_pattern(int rate)
    { // Depunc23 "a b a", Depunc34 "a b a b", Depunc56 "a b a b c b", Depunc78 "a b b b a b c b" (a: in | b: in,128 | c: 128,in)
        switch (rate)
        {
        case 1:
            return PuncPat{3, 0x02u, 0u, 3.5f};
        case 2:
            return PuncPat{4, 0x0Au, 0u, 5.0f};
        case 3:
            return PuncPat{6, 0x3Au, 0x10u, 8.0f};
        default:
            return PuncPat{8, 0xEEu, 0x40u, 10.0f};
        }
    }
    inline int punc_count(const PuncPat &p, int pos0, int n_in)
    { // symbols n_in inputs expand to, starting at pattern position pos0
        int oo = 0, pos = pos0;
        for (int i = 0; i < n_in; i++)
        {
            oo += ((p.two >> pos) & 1u) ? 2 : 1;
            pos = pos + 1 == p.n ? 0 : pos + 1;
        }
        return oo;
    }
    void launch_punc_static(const VitCfg &c, const int8_t *blk, const PuncPat &pat, int shift, int n_in, unsigned char *out, hipStream_t st);
    void launch_punc_cont(const VitCfg &c, const int8_t *blk, int n_in, const PuncPat &pat, int pos0, int lead, int total, unsigned char *carry,
                          unsigned char *dst, hipStream_t st);
    // a run of calls in one launch: input block first_block + b (n_in symbols each) starts at pattern position desc[b].pos0 and its
    // depunctured symbols go to lin + desc[b].off -- the plain concatenation (no carry / odd-count handling: that only decides WHEN a
    // symbol becomes visible to the sliding buffer, which the host keeps track of)
    struct PuncDesc
    {
        int pos0;
        long long off;
    };
    void launch_punc_batch(const VitCfg &c, const int8_t *soft, long long first_block, int nblk, int n_in, const PuncPat &pat, const PuncDesc *desc, unsigned char *lin,
                           hipStream_t st);

    inline int vit_words_per_block(int F)
    {
        int L = (F + 63) / 64;
        L = (L + 31) / 32 * 32; // bits per traceback lane, whole 32-bit words
        return 64 * L / 32;
    }

    // ---- ccsds_simple_psk_decoder: hard decisions of the .soft stream, packed like the Viterbi output --------------
    // (module_ccsds_simple_psk_decoder.cpp:141-262). One block = cadu_bits soft bytes = cadu_bits bits. which: 0 = the bits the
    // module's main `deframer` sees, 1 = the bits `deframer_qpsk` sees (QPSK without NRZ-M only).
    struct HardCfg
    {
        int qpsk, nrzm, swap_iq, swap_diff, oqpsk_delay, method2, method3;
        int F;                 // bits (= soft bytes) per block
        long long blocks_done; // blocks consumed by earlier calls (QPSKDiff swallows the first two symbols of the stream)
        int tail[4];           // the two symbols (I,Q,I,Q) preceding this call's first soft byte; zeros at the stream start
    };
    void launch_hard_bits(const HardCfg &hc, const int8_t *soft, int nblk, int which, uint32_t *vbits, int wpb, hipStream_t st);

    // ---- logical decoded bit stream ------------------------------------------------------------
    // The deframer consumes [carry (carry_bits, raw Viterbi bits incl. >=33 bits of history)] ++ [blocks 0..nblk)
    // where block j contributes F bits from vbits + j*wpb. NRZ-M (differential/nrzm.cpp:24-33) is applied on the fly.
    struct BitStream
    {
        const uint32_t *carry; // packed MSB-first, carry_bits bits
        int carry_bits;
        const uint32_t *vbits;
        int F, wpb;
        int64_t nblk;
        int nrzm;
    };

    // Exact ASM / ~ASM hits (bpsk_ccsds_deframer.cpp:51-66): appends (pos << 1 | inverted) for every logical bit
    // position pos in [from, total) whose 32-bit window ENDING at pos equals asm / ~asm. count is a device counter.
    void launch_sync_search(const BitStream &bs, int64_t from, uint32_t asm_sync, uint32_t *hits, int hits_cap, int *count, hipStream_t st);

    // Copy the (NRZ-M decoded) logical stream [0,total) to a packed MSB-first byte buffer.
    void launch_pack_stream(const BitStream &bs, uint8_t *out_bytes, int64_t total_bits, hipStream_t st);
    // 32-bit windows of the packed stream at p0 + k * step, k < K (what a locked deframer looks at: one word per frame)
    constexpr int WIN_OFFS = 3; // windows gathered per expected frame position: the position itself and the next two bits
    void launch_window_gather(const uint8_t *packed, int64_t total_bits, int64_t p0, int step, int K, uint32_t *words, hipStream_t st);

    // Frame extraction + derandomiser + Reed-Solomon (module_ccsds_conv_concat_decoder.cpp:173-195).
    struct FrameCfg
    {
        int cadu_bits, cadu_bytes;
        uint32_t asm_sync;
        int derand, derand_after_rs, derand_start;
        int rs_i, rs_fill_bytes, rs_dualbasis, rs_nroots; // rs_nroots 32 (rs223) / 16 (rs239); rs_i == 0 disables RS
    };
    struct FrameDesc
    {
        int64_t pos; // logical bit position of the first payload bit (the bit after the ASM)
        int inv;     // bit_inversion
        int pad;
    };
    // frames: nframes descriptors; out: nframes * cadu_bytes; errors: nframes * max(rs_i,1) ints (-1 = uncorrectable)
    // clean_scratch (optional, rs_scratch_bytes(nframes * rs_i) bytes): enables the syndrome screen in front of the thread-per-codeword
    // decoder (a clean flag per codeword, then the 32 syndromes of every codeword, which the decoder takes over instead of evaluating
    // them again, then the count and the ids of the codewords with a non-zero syndrome: the decoder runs over that compacted list)
    inline size_t rs_scratch_list_offset(long long ncw) { return (size_t)((ncw + 15) / 16 * 16) + (size_t)ncw * 32; }
    inline size_t rs_scratch_bytes(long long ncw) { return rs_scratch_list_offset(ncw) + 4 * (size_t)(ncw + 1) + 64; }
    void launch_frames(const BitStream &bs, const FrameCfg &fc, const FrameDesc *frames, int nframes, uint8_t *out, int *errors, hipStream_t st,
                       uint8_t *clean_scratch = nullptr);
    // Unit entry: RS decode of frames already in memory (sdhip_op_rs_decode).
    void launch_rs_only(uint8_t *data, int nframes, int frame_stride, int dualbasis, int I, int nroots, int fill_bytes, int *errors, hipStream_t st,
                        uint8_t *clean_scratch = nullptr);
    // Compaction: copy frames whose keep[i] != 0 to out in order. Returns nothing; count known to the host.
    void launch_compact(const uint8_t *frames, const int *dst_index, int nframes, int cadu_bytes, uint8_t *out, hipStream_t st);
    // rs_usecheck filter + output slots on the device (fec_kernels.hip: k_rs_filter): dst[f] = out_base + (kept frames in front of f) or -1; info[0] = kept, info[1 + k] = last frame's errors[k]
    // wscratch: (nframes + 255) / 256 * 4 integers (the waves' counts, then their offsets)
    void launch_rs_filter(const int *ferr, int nframes, int I, int rs_i, int usecheck, int out_base, int *dst, int *info, int *wscratch, hipStream_t st);
    // viterbi::Viterbi27::work over consecutive frames of one decoder (fec_engine.hip), device buffers
    // enc_state (may be null): the BER re-encoder's shift register in front of frame 0 in, behind the last frame out (it carries across calls)
    void viterbi27_frames(int frame_bits, int ber_test_size, const int8_t *d_soft, int nframes, int start_in0, uint8_t *d_out, std::vector<int> *ber_err, int *ret_state,
                          unsigned *enc_state = nullptr);
} // namespace sdhip
