// No-op stand-in for the (absent, un-vendored) spdlog submodule so that the reference's
// hot-path sources compile in place.  Test infrastructure only (see oracle/README.md).
#pragma once
#include <string>
#include <type_traits>
namespace fmt {   // utils/log_utils.h names fmt::format_string in the (default) no-trace build
template <typename... A> struct basic_format_string {
    template <typename S> constexpr basic_format_string(const S &) {}
};
template <typename... A> using format_string = basic_format_string<std::type_identity_t<A>...>;
}  // namespace fmt
namespace spdlog {
template <typename... A> inline void trace(A&&...) {}
template <typename... A> inline void debug(A&&...) {}
template <typename... A> inline void info(A&&...) {}
template <typename... A> inline void warn(A&&...) {}
template <typename... A> inline void error(A&&...) {}
template <typename... A> inline void critical(A&&...) {}
}  // namespace spdlog
