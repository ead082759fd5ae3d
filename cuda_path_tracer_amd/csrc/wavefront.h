// wavefront.h -- wavefront path-tracing driver (implemented in wavefront.hip).
#pragma once
#include <string>
#include "../../include/mipt.h"
#include "device_ctx.h"

namespace mipt {
// Renders cfg's sample set on one device, ADDING sum_s trace_s/spp into d_accum (W*H*3 fp32, device).
int wavefront_render(const HostScene &hs, DeviceCtx &dc, const mipt_render_config &cfg, float *d_accum, mipt_stats *st,
                     std::string &err);
// EnvMap::getPixelColor for n device-resident directions.
int wavefront_env_lookup(DeviceCtx &dc, const float *d_dirs, int64_t n, float *d_rgb, std::string &err);
}  // namespace mipt
