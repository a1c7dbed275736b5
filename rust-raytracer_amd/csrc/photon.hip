// photon.hip — photon emission / bounce tracing, photon index build (placeholder until the
// render path is verified on hardware; replaced by the real kernels next).
#include "runtime.h"
using namespace rrt;
extern "C" {
int rrt_photon_begin(rrt_scene *s, uint64_t seed) { return fail(RRT_ERR_STATE, "photon pass not built yet"); }
int rrt_photon_shoot(rrt_scene *s, int light, int pass, uint64_t first, uint64_t count) { return fail(RRT_ERR_STATE, "photon pass not built yet"); }
int rrt_photon_build(rrt_scene *s) { return fail(RRT_ERR_STATE, "photon pass not built yet"); }
}
