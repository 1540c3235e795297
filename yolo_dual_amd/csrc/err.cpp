#include "common.h"
#include <atomic>
static thread_local std::string g_err;
void ydl_set_error(const std::string& s) { g_err = s; }
extern "C" const char* ydl_last_error(void) { return g_err.c_str(); }
extern "C" int ydl_version(void) { return 2; }

// ---- per-device state (see common.h) -------------------------------------------------------------------
static std::atomic<int> g_attr_sets{0};
bool ydl_dev_once(void* tag_storage) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= YDL_MAX_DEVICES) dev = 0;
    auto* w = reinterpret_cast<std::atomic<unsigned long long>*>(tag_storage);
    const unsigned long long bit = 1ull << dev;
    const unsigned long long old = w->fetch_or(bit, std::memory_order_acq_rel);
    if (old & bit) return false;
    g_attr_sets.fetch_add(1, std::memory_order_relaxed);
    return true;
}
int ydl_device_cus() {
    static std::atomic<int> cus[YDL_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= YDL_MAX_DEVICES) dev = 0;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}
// number of (kernel, device) attribute initialisations so far: lets a test see that a second device id gets its own
extern "C" int ydl_debug_attr_sets(void) { return g_attr_sets.load(); }
// test hook: forget which devices were initialised for the tag words is not possible from here (they are call-site
// statics); instead tests switch devices.  On a one-GPU box ydl_debug_attr_sets() is compared before/after a first launch.

static std::atomic<const char*> g_last_kernel[4];
void ydl_note_kernel(int family, const char* name) {
    if (family >= 0 && family < 4) g_last_kernel[family].store(name, std::memory_order_relaxed);
}
// family: 0 conv_fwd, 1 conv_dgrad, 2 conv_wgrad, 3 bn_finalize.  Returns a static string naming the kernel instantiation the
// last call of that family launched (process-wide, diagnostics for the parity tests only), or "" if none yet.
extern "C" const char* ydl_debug_last_kernel(int family) {
    if (family < 0 || family >= 4) return "";
    const char* s = g_last_kernel[family].load(std::memory_order_relaxed);
    return s ? s : "";
}
