// oracle/ref_wrap_doppler.cpp -- TEST INFRASTRUCTURE ONLY: the reference's dsp::DopplerCorrectBlock (src-core/common/dsp/utils/doppler_correct.{h,cpp}),
// compiled in place together with the libpredict sources it calls (src-core/libs/predict/*.c) and driven buffer by buffer through its own dsp::streams the
// way BaseDemodModule drives it for a baseband FILE (module_demod_base.cpp:161-166: start_time = "start_timestamp"). Its constructor asks SatDump's TLE
// registry for the satellite: ref_shim/init.h stands in for that one call, the registry is defined and filled here. -fno-access-control: work() and
// targ_freq are private. Pins oracle/sd_oracle.c's restatement of the sample loop (sdo_doppler) and is the yardstick of the plugin's target computation.
#include "common/dsp/utils/doppler_correct.h"
#include "init.h"
#include <cstring>

namespace satdump
{
    std::shared_ptr<KeplerDBHandler> db_keplers = std::make_shared<KeplerDBHandler>();
}

extern "C"
{
    // in: n complex samples handed to the block in buffers of `buf` (the last one may be short). out: the corrected samples. targets_out[k] = the block's
    // targ_freq BEHIND buffer k (in force during buffer k + 1). Returns the number of buffers, <0 on error.
    int sdref_doppler(const char *tle1, const char *tle2, int norad, double samplerate, float alpha, double signal_frequency, double qth_lon, double qth_lat, double qth_alt,
                      double start_time, const float *in, long long n, int buf, float *out, float *targets_out, int ntargets_cap)
    {
        try
        {
            satdump::db_keplers->tles.clear();
            satdump::TLE t;
            t.norad = norad;
            t.name = "sat";
            t.line1 = tle1;
            t.line2 = tle2;
            satdump::db_keplers->tles.push_back(t);
            auto src = std::make_shared<dsp::stream<complex_t>>();
            dsp::DopplerCorrectBlock blk(src, samplerate, alpha, signal_frequency, norad, qth_lon, qth_lat, qth_alt);
            blk.start_time = start_time;
            int k = 0;
            for (long long o = 0; o < n; o += buf, k++)
            {
                const int m = (int)std::min<long long>(buf, n - o);
                memcpy(src->writeBuf, in + 2 * o, (size_t)m * sizeof(complex_t));
                src->swap(m);
                blk.work();
                const int got = blk.output_stream->read();
                if (got != m)
                    return -2;
                memcpy(out + 2 * o, blk.output_stream->readBuf, (size_t)m * sizeof(complex_t));
                blk.output_stream->flush();
                if (k < ntargets_cap)
                    targets_out[k] = blk.targ_freq;
            }
            return k;
        }
        catch (std::exception &e)
        {
            fprintf(stderr, "sdref_doppler: %s\n", e.what());
            return -1;
        }
    }
}
