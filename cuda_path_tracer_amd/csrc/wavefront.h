// wavefront.h -- wavefront path-tracing driver (implemented in wavefront.hip).
#pragma once
#include <string>
#include "../../include/mipt.h"
#include "device_ctx.h"

namespace mipt {
// mipt_render_config.flags / mipt_trace_closest flags -> variant of the closest-hit kernel (kernels_extend.h)
inline int extend_mode(int flags) {
    return (flags & MIPT_FLAG_COUNTERS) ? 1 /* kCountRef */ : ((flags & MIPT_FLAG_WORK_COUNTERS) ? 2 /* kCountWork */ : 0 /* kProd */);
}
// Copies the production kernel's work counters (DeviceCtx::counters + kCtrWork) into st->work_*.
inline void fill_work_counters(DeviceCtx &dc, mipt_stats *st) {
    unsigned long long c[6] = {0};
    (void)hipMemcpy(c, dc.counters.as<char>() + kCtrWork, sizeof(c), hipMemcpyDeviceToHost);
    st->work_steps = (int64_t)c[0]; st->work_pops = (int64_t)c[1]; st->work_sphere_pairs = (int64_t)c[2];
    st->work_planes = (int64_t)c[3]; st->work_verts = (int64_t)c[4]; st->work_leaves = (int64_t)c[5];
}
// Renders cfg's sample set on one device, ADDING sum_s trace_s/spp into d_accum (W*H*3 fp32, device).
int wavefront_render(const HostScene &hs, DeviceCtx &dc, const mipt_render_config &cfg, float *d_accum, mipt_stats *st,
                     std::string &err);
// EnvMap::getPixelColor for n device-resident directions.
int wavefront_env_lookup(DeviceCtx &dc, const float *d_dirs, int64_t n, float *d_rgb, std::string &err);
}  // namespace mipt
