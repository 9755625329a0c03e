// the device entry points host_scene.cpp's render helper links against; never called by this driver
#include "../../include/rtx.h"
extern "C" {
int rtx_init(int, rtx_ctx **) { return -2; }
const char *rtx_last_error(rtx_ctx *) { return ""; }
int rtx_set_build_threads(rtx_ctx *, int) { return -2; }
int rtx_upload_scene(rtx_ctx *, const rtx_scene_desc *) { return -2; }
int rtx_render(rtx_ctx *, uint32_t, uint32_t, uint64_t, const rtx_partition *, uint32_t, uint8_t *, float *, rtx_stats *) { return -2; }
void rtx_destroy(rtx_ctx *) {}
}
