// lasgun_amd/csrc/devmem.cpp -- process-wide state of the library (internal.h): the pools, the current device, the last error.
#include "internal.h"

thread_local std::string tl_error;
int g_device = 0;
bool g_device_chosen = false;
std::vector<int> g_devices;

void use_device(int dev) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        throw Error("no HIP device available: liblasgun_hip has no CPU fallback (hipGetDeviceCount: " +
                    std::string(e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + ")");
    if (dev < 0 || dev >= n) throw Error("device index out of range");
    HIP_TRY(hipSetDevice(dev));
}
void use_device() { use_device(g_device); }

// never destroyed: buffers released at interpreter exit, after static destructors have begun, still find them
DevPool &g_pool = *new DevPool();
StreamPool &g_streams = *new StreamPool();
PinnedPool &g_pinned = *new PinnedPool();
ErrWords &g_err_words = *new ErrWords();
