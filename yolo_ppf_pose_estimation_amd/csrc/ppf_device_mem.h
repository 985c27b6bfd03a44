/*
 * ppf_device_mem.h — errors (fail / HIPCHK), the process-wide device block cache (DevPool) and the buffers drawn from it (DevBuf).
 * Part of the one translation unit ppf_hip.hip.
 */
#ifndef PPF_DEVICE_MEM_H
#define PPF_DEVICE_MEM_H

/* ============================================================================================ */
/* errors                                                                                         */
/* ============================================================================================ */
namespace {

thread_local std::string g_last_error;

ppf_status fail(ppf_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  /* error paths return while kernels of the failed call may still run; their scratch goes back to the block cache
   * (DevPool) as the locals unwind, so drain the device first.  Errors are rare: the cost does not matter. */
  if (st == PPF_ERR_HIP || st == PPF_ERR_NOMEM || st == PPF_ERR_CAPACITY) (void)hipDeviceSynchronize();
  return st;
}

#define HIPCHK(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t e__ = (expr);                                                                               \
    if (e__ != hipSuccess)                                                                                 \
      return fail(PPF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

/* Device memory for scratch and results comes from a process-wide cache of freed blocks (power-of-two size classes
 * per device): hipMalloc costs tens of microseconds and hipFree synchronises the whole device, which is most of the
 * time of the small stages (cloud stages, ICP set-up).  A block is only released by a DevBuf whose last user has been
 * synchronised with (every entry point waits for its kernels before its scratch goes out of scope), so a reused block
 * is never still in flight.  PPF_NO_POOL=1 turns the cache off. */
class DevPool {
 public:
  static DevPool& get() {
    static DevPool* p = new DevPool(); /* never destroyed: no hipFree after the runtime is gone */
    return *p;
  }
  /* size classes: 8 per octave (1, 1.125, ... 1.875 x 2^k), so a block wastes at most 12.5 % of what was asked for */
  static size_t class_size(int cls) { return ((size_t)8 + (size_t)(cls & 7)) << (cls >> 3); }
  static int class_of(size_t bytes) {
    int cls = 5 * 8; /* 256 B */
    while (class_size(cls) < bytes) cls++;
    return cls;
  }
  hipError_t acquire(size_t bytes, void** out, size_t* granted, int* device) {
    *out = nullptr;
    const size_t want = std::max<size_t>(bytes, 256);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    *device = dev; /* a block goes back to the list of the device it was allocated on, whatever is current then */
    if (off_) { *granted = want; return hipMalloc(out, want); }
    const int cls = class_of(want);
    {
      std::lock_guard<std::mutex> g(mu_);
      auto& lst = free_[key(dev, cls)];
      if (!lst.empty()) { *out = lst.back(); lst.pop_back(); *granted = class_size(cls); return hipSuccess; }
    }
    *granted = class_size(cls);
    e = hipMalloc(out, *granted);
    if (e != hipSuccess) { /* out of memory: drop the cache and retry once */
      trim();
      e = hipMalloc(out, *granted);
    }
    return e;
  }
  void release(void* p, size_t granted, int dev) {
    if (!p) return;
    if (off_) { (void)hipFree(p); return; }
    const int cls = class_of(granted);
    std::lock_guard<std::mutex> g(mu_);
    free_[key(dev, cls)].push_back(p);
  }
  void trim() {
    std::lock_guard<std::mutex> g(mu_);
    for (auto& kv : free_) {
      for (void* p : kv.second) (void)hipFree(p);
      kv.second.clear();
    }
  }

 private:
  DevPool() : off_(getenv("PPF_NO_POOL") != nullptr) {}
  static int key(int dev, int cls) { return dev * 1024 + cls; }
  std::mutex mu_;
  std::map<int, std::vector<void*>> free_;
  bool off_;
};

void sync_device_of_block(int dev); /* = sync_device, defined below */

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;      /* elements usable */
  size_t granted = 0;  /* bytes of the block behind p */
  int device = 0;      /* device the block lives on */
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { DevPool::get().release(p, granted, device); }
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) { /* growing a live buffer (rare): earlier asynchronous work may still use the old block */
      sync_device_of_block(device);
      DevPool::get().release(p, granted, device); p = nullptr; cap = 0; granted = 0;
    }
    void* q = nullptr;
    hipError_t e = DevPool::get().acquire(std::max<size_t>(n, 1) * sizeof(T), &q, &granted, &device);
    if (e == hipSuccess) { p = static_cast<T*>(q); cap = n; }
    return e;
  }
  /* like reserve, but a block more than twice as big as needed (and above 16 MiB) is traded for a fitting one: the hit
   * pools of a workspace shrink again after an unusually dense scene */
  hipError_t fit(size_t n) {
    if (p && granted > ((size_t)16 << 20) && granted > 2 * std::max<size_t>(n, 1) * sizeof(T)) {
      sync_device_of_block(device);
      DevPool::get().release(p, granted, device); p = nullptr; cap = 0; granted = 0;
    }
    return reserve(n);
  }
  size_t bytes() const { return granted; }
};

void sync_device(int dev);
void sync_device_of_block(int dev) { sync_device(dev); }

/* wait for everything enqueued on device `dev` (the device a buffer lives on, which need not be the current one) */
void sync_device(int dev) {
  int cur = -1;
  if (dev < 0 || hipGetDevice(&cur) != hipSuccess || cur == dev) { (void)hipDeviceSynchronize(); return; }
  if (hipSetDevice(dev) == hipSuccess) {
    (void)hipDeviceSynchronize();
    (void)hipSetDevice(cur);
  }
}

constexpr int LDS_BYTES = 160 * 1024;              /* LDS per CU == per k_vote workgroup */
constexpr size_t HIT_BYTES_BUDGET = 4ull << 30;    /* hit scratch per batch of reference points */
constexpr float SPILL_ALPHA_MIN = 3.1415f;         /* entries with alpha_m >= this can reach alpha bin == numAngles */

}  // namespace

#endif /* PPF_DEVICE_MEM_H */
