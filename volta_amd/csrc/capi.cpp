// libvolta_hip.so: error plumbing and library identity.
#include "util.h"
#include "../../include/volta_hip.h"

namespace vk {
static thread_local char g_err[512] = {0};
char* error_buffer() { return g_err; }
int set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error("%s: %s", what, hipGetErrorString(e));
    return 0;
}
}  // namespace vk

extern "C" int vk_version(void) { return 1; }
extern "C" const char* vk_device_arch(void) { return "gfx950"; }
extern "C" const char* vk_last_error(void) { return vk::error_buffer(); }
