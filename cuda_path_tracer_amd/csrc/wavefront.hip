#include "wavefront.h"
namespace mipt {
int wavefront_render(const HostScene &, DeviceCtx &, const mipt_render_config &, float *, mipt_stats *, std::string &err) {
    err = "render not implemented yet";
    return -9;
}
int wavefront_env_lookup(DeviceCtx &, const float *, int64_t, float *, std::string &err) {
    err = "env lookup not implemented yet";
    return -9;
}
}  // namespace mipt
