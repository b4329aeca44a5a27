// Recycling allocator of libhyphy_hip.so (partition.h): device blocks, pinned host blocks and streams of destroyed partitions
// are kept for the next partition of the same shape.  Host-side bookkeeping only.
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "partition.h"

namespace hyhip {
namespace {

struct Pool {
  std::mutex m;
  std::map<std::pair<int, size_t>, std::vector<void *>> free_dev, free_host;  // (device, rounded bytes) -> blocks
  std::unordered_map<void *, std::pair<int, size_t>> live_dev, live_host;      // every block handed out
  std::map<int, std::vector<hipStream_t>> streams;
  size_t cached_dev = 0, cached_host = 0;
};
Pool &pool() {
  static Pool *p = new Pool;  // (never destroyed: the HIP runtime may be gone before static destructors run)
  return *p;
}
size_t cap_bytes() {
  static const size_t cap = [] {
    const char *e = getenv("HYPHY_HIP_POOL_MB");
    const long mb = e ? atol(e) : 1024;
    return (size_t)(mb < 0 ? 0 : mb) << 20;  // (a negative value means "off", not 2^64 bytes)
  }();
  return cap;
}
constexpr size_t kMaxCachedBlock = (size_t)64 << 20;

// Give every cached block of one kind back to the driver (the device's, or the pinned host blocks made under it): cached blocks are
// keyed by exact size, so a new partition of another shape cannot use them — but they still hold the memory it is being refused.
void release_cached(Pool &P, bool host, int dev) {
  std::vector<void *> drop;
  {
    std::lock_guard<std::mutex> lock(P.m);
    auto &lists = host ? P.free_host : P.free_dev;
    size_t &cached = host ? P.cached_host : P.cached_dev;
    for (auto &kv : lists) {
      if (kv.first.first != dev) continue;
      for (void *b : kv.second) {
        drop.push_back(b);
        cached -= kv.first.second;
      }
      kv.second.clear();
    }
  }
  if (drop.empty()) return;
  if (!host) hipDeviceSynchronize();
  for (void *b : drop) {
    if (host) hipHostFree(b);
    else hipFree(b);
  }
}
size_t rounded(size_t bytes) { return bytes == 0 ? 256 : (bytes + 255) & ~(size_t)255; }

}  // namespace

hipError_t pool_malloc(void **out, size_t bytes) {
  Pool &P = pool();
  int dev = 0;
  hipGetDevice(&dev);
  const size_t n = rounded(bytes);
  {
    std::lock_guard<std::mutex> lock(P.m);
    auto it = P.free_dev.find(std::make_pair(dev, n));
    if (it != P.free_dev.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      P.cached_dev -= n;
      P.live_dev[*out] = std::make_pair(dev, n);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, n);
  if (e != hipSuccess) {  // the blocks this pool keeps for partitions of other shapes may be what is missing: release them, once
    (void)hipGetLastError();
    release_cached(P, false, dev);
    e = hipMalloc(out, n);
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(P.m);
    P.live_dev[*out] = std::make_pair(dev, n);
  }
  return e;
}

void pool_free(void *p) {
  if (!p) return;
  Pool &P = pool();
  {
    std::lock_guard<std::mutex> lock(P.m);
    auto it = P.live_dev.find(p);
    if (it != P.live_dev.end()) {
      const std::pair<int, size_t> key = it->second;
      P.live_dev.erase(it);
      if (key.second <= kMaxCachedBlock && P.cached_dev + key.second <= cap_bytes()) {
        P.free_dev[key].push_back(p);
        P.cached_dev += key.second;
        return;
      }
    }
  }
  hipFree(p);
}

void pool_free_sync(void *p) {
  if (!p) return;
  hipDeviceSynchronize();  // (what hipFree does implicitly: nothing in flight may still use the block)
  pool_free(p);
}

hipError_t pool_host_malloc(void **out, size_t bytes) {
  Pool &P = pool();
  int dev = 0;
  hipGetDevice(&dev);
  const size_t n = rounded(bytes);
  {
    std::lock_guard<std::mutex> lock(P.m);
    auto it = P.free_host.find(std::make_pair(dev, n));
    if (it != P.free_host.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      P.cached_host -= n;
      P.live_host[*out] = std::make_pair(dev, n);
      return hipSuccess;
    }
  }
  hipError_t e = hipHostMalloc(out, n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    release_cached(P, true, dev);
    e = hipHostMalloc(out, n);
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(P.m);
    P.live_host[*out] = std::make_pair(dev, n);
  }
  return e;
}

void pool_host_free(void *p) {
  if (!p) return;
  Pool &P = pool();
  {
    std::lock_guard<std::mutex> lock(P.m);
    auto it = P.live_host.find(p);
    if (it != P.live_host.end()) {
      const std::pair<int, size_t> key = it->second;
      P.live_host.erase(it);
      if (key.second <= kMaxCachedBlock && P.cached_host + key.second <= cap_bytes()) {
        P.free_host[key].push_back(p);
        P.cached_host += key.second;
        return;
      }
    }
  }
  hipHostFree(p);
}

hipError_t pool_stream_get(hipStream_t *s) {
  Pool &P = pool();
  int dev = 0;
  hipGetDevice(&dev);
  if (cap_bytes() > 0) {
    std::lock_guard<std::mutex> lock(P.m);
    std::vector<hipStream_t> &v = P.streams[dev];
    if (!v.empty()) {
      *s = v.back();
      v.pop_back();
      return hipSuccess;
    }
  }
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

void pool_stream_put(hipStream_t s) {
  if (!s) return;
  Pool &P = pool();
  int dev = 0;
  hipGetDevice(&dev);
  if (cap_bytes() > 0) {
    std::lock_guard<std::mutex> lock(P.m);
    std::vector<hipStream_t> &v = P.streams[dev];
    if (v.size() < 8) {
      v.push_back(s);
      return;
    }
  }
  hipStreamDestroy(s);
}

}  // namespace hyhip
