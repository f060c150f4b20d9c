// Process-wide switches of the kernel library.
//
// DPA_DETERMINISTIC=1 (or set_deterministic(true)): bit-reproducible gradients for debugging.  Every place where several
// workgroups add floats into one address with atomics - the pixel-split weight gradient, the BatchNorm statistics that
// conv epilogues accumulate into a few replica rows, the fused bottleneck-tail backward, the embedding gradient - takes
// a single-writer route instead (one split, one statistics row per contributing workgroup folded in a fixed order,
// the unfused kernel sequence, one workgroup per vocabulary row).  Slower; meant for chasing a diverging run, where
// "did the numbers change because of my edit or because of the atomics' order" has to have an answer.
#include <stdlib.h>
#include "kernels.h"

namespace dpa {
namespace {
int g_det = -1;
}
bool deterministic() {
    if (g_det < 0) {
        const char* e = getenv("DPA_DETERMINISTIC");
        g_det = (e && atoi(e) != 0) ? 1 : 0;
    }
    return g_det == 1;
}
void set_deterministic(bool on) { g_det = on ? 1 : 0; }
}  // namespace dpa
