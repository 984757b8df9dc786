// ref_shim/init.h -- stand-in for src-core/init.h (which pulls SatDump's database layer in) for the two places that compile a reference source or the
// plugin OUTSIDE a SatDump tree (oracle/Makefile, tests/minihost, plugin/Makefile's build check): just the TLE registry's one method that
// dsp::DopplerCorrectBlock's constructor calls (src-core/common/dsp/utils/doppler_correct.cpp:17: satdump::db_keplers->get_from_norad(norad).value()).
// Whoever links must define satdump::db_keplers and fill it. In a SatDump tree the real init.h is found first.
#pragma once
#include "common/tracking/tle.h"
#include <memory>
#include <optional>
#include <vector>

namespace satdump
{
    struct KeplerDBHandler
    {
        std::vector<TLE> tles;
        std::optional<TLE> get_from_norad(int norad)
        {
            for (auto &t : tles)
                if (t.norad == norad)
                    return t;
            return std::nullopt;
        }
    };
    extern std::shared_ptr<KeplerDBHandler> db_keplers;
} // namespace satdump
