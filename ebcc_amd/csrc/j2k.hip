// j2k.hip - JPEG 2000 base layer on device (placeholder until the kernels land)
#include "engine.hpp"
namespace ebcc {
bool j2k_create(ebcc_hip_ctx *) { return true; }
void j2k_destroy(ebcc_hip_ctx *) {}
}
