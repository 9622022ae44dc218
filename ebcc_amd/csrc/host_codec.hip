// host_codec.hip - placeholder
#include "engine.hpp"
#include "../../include/ebcc_hip.h"
extern "C" {
__attribute__((visibility("default"))) void free_buffer(void *p) { if (p) free(p); }
}
