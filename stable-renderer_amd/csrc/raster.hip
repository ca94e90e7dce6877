// placeholder until the tiled rasterizer lands (same file)
#include "sr_common.h"
extern "C" int sr_gbuffer_clear(const sr_gbuffer*, void*) { SR_FAIL(SR_ERR_UNSUPPORTED, "raster: not built yet"); }
extern "C" int sr_raster_draw(const sr_draw*, const sr_gbuffer*, void*, int64_t, void*) { SR_FAIL(SR_ERR_UNSUPPORTED, "raster: not built yet"); }
extern "C" int64_t sr_raster_scratch_bytes(int32_t, int32_t, int32_t) { return 0; }
