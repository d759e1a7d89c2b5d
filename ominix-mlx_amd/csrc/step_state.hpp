// Step state of the decode engine: lives in device memory so that a captured step never needs host patching (engine.hip).
#pragma once
#include "common.hpp"

namespace omx {

struct StepState {
    int pos;               // tokens in the cache == RoPE offset of the token being processed
    uint32_t cur_token;    // token fed to the embedding this step
    int out_count;         // tokens sampled so far
    int prompt_idx;        // next prompt token to feed during a token-serial prefill
};

}  // namespace omx
