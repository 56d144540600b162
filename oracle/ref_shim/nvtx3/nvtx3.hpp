// No-op stand-in for the (absent) NVTX submodule. Test infrastructure only.
#pragma once
namespace nvtx3 {
struct scoped_range {
    template <typename... A> explicit scoped_range(A&&...) {}
};
}  // namespace nvtx3
#define NVTX3_FUNC_RANGE() do {} while (0)
