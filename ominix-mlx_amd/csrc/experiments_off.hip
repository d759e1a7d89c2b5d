// The default build's answer to the entry points of the experimental engines (Makefile: EXPERIMENTS=1 builds step_engine.hip and
// aql_step.hip instead of this file): "not built".  engine.hip needs no #ifdef: step_engine_ok() == false keeps the persistent step off,
// aql_build() == nullptr makes the engine mark its AQL replay unavailable and stay on the hipGraph.
#include "aql_step.hpp"
#include "step_engine.hpp"

namespace omx {

bool step_engine_ok(int, int, int, int, int, int, int) { return false; }
size_t step_engine_granules(int, int, int, int, int) { return 0; }
int launch_step_engine(const StepEngineArgs&, int, hipStream_t) {
    return set_error("the persistent decode step is an experiment: build with `make EXPERIMENTS=1` and load it with OMX_LIB_VARIANT=exp");
}

AqlProgram* aql_build(const LaunchRecorder&, int) {
    set_error("the AQL replay of the decode step is an experiment: build with `make EXPERIMENTS=1` and load it with OMX_LIB_VARIANT=exp");
    return nullptr;
}
void aql_destroy(AqlProgram*) {}
int aql_launches(const AqlProgram*) { return 0; }
int aql_replay(AqlProgram*, int, double*, float*) { return set_error("the AQL replay of the decode step is not built (make EXPERIMENTS=1)"); }

}  // namespace omx
