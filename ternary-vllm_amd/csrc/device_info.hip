// Facts about the current device that the launchers need on every call, looked up once per device.
#include "tvl_common.h"
#include "tvl_kernels.h"

#include <atomic>
#include <mutex>
#include <unordered_map>

namespace tvl {

constexpr int MAX_DEVICES = 64;

int device_cu_count() {
    static std::atomic<int> cus[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}

// Kernels that use more than 48 KiB of dynamic LDS must be told so once per device; repeating the runtime call on every
// launch of a 4 .. 100 us kernel is wasted host time.
int set_dynamic_lds_limit(const void* kernel, size_t lds_bytes) {
    if (lds_bytes <= 48 * 1024) return TVL_OK;
    static std::mutex mu;
    static std::unordered_map<uint64_t, size_t> done[MAX_DEVICES];
    int dev = 0;
    TVL_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    auto& seen = done[dev];
    auto it = seen.find((uint64_t)(uintptr_t)kernel);
    if (it != seen.end() && it->second >= lds_bytes) return TVL_OK;
    TVL_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    seen[(uint64_t)(uintptr_t)kernel] = lds_bytes;
    return TVL_OK;
}

}  // namespace tvl
