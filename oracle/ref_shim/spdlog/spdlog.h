// No-op stand-in for the (absent, un-vendored) spdlog submodule so that the reference's
// hot-path sources compile in place.  Test infrastructure only (see oracle/README.md).
#pragma once
#include <string>
namespace spdlog {
template <typename... A> inline void trace(A&&...) {}
template <typename... A> inline void debug(A&&...) {}
template <typename... A> inline void info(A&&...) {}
template <typename... A> inline void warn(A&&...) {}
template <typename... A> inline void error(A&&...) {}
template <typename... A> inline void critical(A&&...) {}
}  // namespace spdlog
