// No-op stand-in for the reference's src-core/logger.h (placed FIRST on the
// include path for the oracle/_ref build). Test infrastructure only.
#pragma once
#include <memory>
#include <string>
namespace slog
{
    struct Logger
    {
        template <class... A> void trace(A...) {}
        template <class... A> void debug(A...) {}
        template <class... A> void info(A...) {}
        template <class... A> void warn(A...) {}
        template <class... A> void error(A...) {}
        template <class... A> void critical(A...) {}
    };
}
extern std::shared_ptr<slog::Logger> logger;
