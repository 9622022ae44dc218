/* placeholder - replaced below */
#include "oracle.h"
size_t orc_j2k_encode(const uint16_t *img, size_t height, size_t width, float base_cr, uint8_t **out) { (void)img;(void)height;(void)width;(void)base_cr;(void)out; return 0; }
size_t orc_j2k_decode(const uint8_t *cs, size_t cs_size, int32_t **samples, size_t *height, size_t *width) { (void)cs;(void)cs_size;(void)samples;(void)height;(void)width; return 0; }
