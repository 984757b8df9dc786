// Stand-in for <volk/volk_alloc.hh> (oracle/_ref build only). The real header drags in
// <cstdint>/<cstdlib>, which the reference's cc_decoder.h relies on.
#pragma once
#include <cstdint>
#include <stdint.h>
#include <cstdlib>
#include <vector>
namespace volk { template <class T> using vector = std::vector<T>; }
