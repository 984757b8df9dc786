// dvbs2_stages.h -- what the DVB-S2 demodulator engine (dvbs2_engine.hip) shares with the stage kernels' file (dvbs2_demap.hip): the frame PLL as
// an object that owns its tables and the loop state (dvbs2::S2PLLBlock, plugins/dvb_support/dvbs2/dvbs2_pll.{h,cpp}), the per-device cache of the
// soft demapper stage's tables, and the frame geometry of a MODCOD (get_dvbs2_cfg, codings/dvb-s2/modcod_to_cfg.h:19-151).
#pragma once
#include "common.h"

#include <memory>
#include <vector>

namespace sdhip
{
    struct S2Cfg
    {
        int bits, slots, rate, constellation;
    };
    S2Cfg s2_cfg_of(int modcod, int shortframes);
    // S2PLSyncBlock's constructor (dvbs2_pl_sync.cpp:12-30): symbols per frame as the synchroniser emits them
    int s2_raw_frame_size(int slot_number, int pilots);
    // S2PLLBlock::update (dvbs2_pll.h:33-47): symbols of a frame the loop walks
    int s2_pll_walked(int slots, int pilots);

    // sdhip_s2_pl_sync_dev with the speculation width carried by the caller (dvbs2_demap.hip)
    int64_t s2_pl_sync_run(int device, int slot_number, int pilots, float thresold, const float *d_syms, size_t nsyms, float *d_frames, int frame_stride, size_t max_frames,
                           size_t *consumed, int *best_pos_out, size_t *spec_io);

    struct S2PllState
    {
        float phase, freq;
    };
    struct S2PllStats
    {
        unsigned lanes = 0;   // lanes of the last parallel call
        unsigned rerun = 0;   // lanes re-run from their predecessor's exact end state (certificate missed)
        unsigned forced = 0;  // boundaries let through after the round limit (the loop was not locked there)
        unsigned serial_frames = 0; // frames walked by the serial lane (exact mode, acquisition)
        unsigned branch_tries = 0;  // whole-batch launches spent on the estimates' frequency branch (a quarter of the chain missed)
    };
    struct S2PllImpl;
    // One stream's frame PLL. exact: the serial lane only (bit for bit the reference's loop). Otherwise the frame-parallel schedule (dvbs2_demap.hip):
    // lanes from data-aided header estimates, certified against their predecessors; the first frames of a stream are walked serially until the loop
    // frequency is there to pick the estimates' branch.
    struct S2Pll
    {
        S2Pll(int device, int modcod, int shortframes, int pilots, float loop_bw, const float *lut_phase_error, int lut_resolution, bool exact);
        ~S2Pll();
        int per_frame() const;
        void run(const float *d_in, float *d_out, int stride, int nframes, hipStream_t st = nullptr);
        S2PllState state;  // carried across calls ({0, 0} for a new stream)
        S2PllStats stats;
        bool exact;
        bool have_hint = false; // the carried frequency is a locked loop's
        void set_hint(double freq);   // the loop frequency the estimates pick their branch with (a caller that vouches for a locked state)
        void add_frequency(float df); // the engine's frequency hand-over (freq_prop_factor): the loop's frequency state moves by df
      private:
        std::unique_ptr<S2PllImpl> im;
    };
} // namespace sdhip
