// Library-owned scratch (split-KV partials, GEMM split-K slabs).  Grown lazily with hipMalloc
// outside of graph capture; callers that capture graphs pre-size it with omx_set_workspace or by
// running the op once eagerly.
#pragma once
#include "common.hpp"
namespace omx {
int get_workspace(void** ptr, size_t bytes);
}
