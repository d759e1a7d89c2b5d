// Library-owned scratch (split-KV partials, GEMM split-K slabs).  Grown lazily with hipMalloc
// outside of graph capture; callers that capture graphs pre-size it with omx_set_workspace or by
// running the op once eagerly.
#pragma once
#include "common.hpp"
namespace omx {
int get_workspace(void** ptr, size_t bytes);
// independent of the above (callers nested inside a get_workspace user), one buffer per stream: concurrent users on different streams
// never share or reallocate each other's scratch
int get_workspace_aux(void** ptr, size_t bytes, hipStream_t s);
// a graph captured on `s` holds pointers into the stream's buffer: until unpinned, a request that would have to move it is an error
void workspace_aux_pin(hipStream_t s, bool pinned);
void workspace_release_stream(hipStream_t s);       // the stream's owner is about to destroy it
}
