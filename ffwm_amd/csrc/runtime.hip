// runtime.hip -- error reporting, option switches and the built-in HIP-event launch profiler.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "common.hpp"

namespace ffwm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return FFWM_ERR_LAUNCH;
    }
    return FFWM_OK;
}

void allow_large_lds(const void* kernel) {
    static std::mutex mu;
    static std::set<const void*> done;
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert(kernel).second) {
        // kernels with static __shared__ variables: static + dynamic must stay <= 160 KiB or the call fails
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess) {
            (void)hipGetLastError();   // do not leave a sticky error for the next check_launch()
            if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes - 2048) != hipSuccess)
                (void)hipGetLastError();
        }
    }
}

const char* scope_at(const char* base, int64_t size) {
    static std::mutex mu;
    static std::set<std::string> names;
    char buf[96];
    snprintf(buf, sizeof(buf), "%s@%lld", base, static_cast<long long>(size));
    std::lock_guard<std::mutex> lk(mu);
    return names.insert(buf).first->c_str();
}

Options& options() {
    static Options o;
    return o;
}

void* stream_scratch(hipStream_t st) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, void*> pool;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    auto it = pool.find({dev, st});
    if (it != pool.end()) return it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cs != hipStreamCaptureStatusNone) return nullptr;          // no allocation inside a capture
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    pool[{dev, st}] = p;
    return p;
}

int device_cus() {
    static std::mutex mu;
    static std::vector<int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) { (void)hipGetLastError(); return 256; }
    std::lock_guard<std::mutex> lk(mu);
    if (static_cast<size_t>(dev) >= cache.size()) cache.resize(dev + 1, 0);
    if (cache[dev] <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
        cache[dev] = n;
    }
    return cache[dev];
}

namespace {
__global__ void __launch_bounds__(256) zero_fill_kernel(unsigned* __restrict__ p, size_t n) {          // n dwords
    const size_t mis = (16u - (reinterpret_cast<uintptr_t>(p) & 15u)) & 15u;
    size_t head = mis / 4;
    if (head > n) head = n;
    const size_t body = (n - head) / 4;          // 16-byte stores
    uint4* q = reinterpret_cast<uint4*>(p + head);
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x, stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t j = i; j < body; j += stride) q[j] = make_uint4(0u, 0u, 0u, 0u);
    const size_t done = head + body * 4;
    if (i < head) p[i] = 0u;
    if (i < n - done) p[done + i] = 0u;
}
}  // namespace

int zero_fill(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    if (!p || (bytes & 3u) || (reinterpret_cast<uintptr_t>(p) & 3u)) return 1;
    if (options().zero_fill_memset) return hipMemsetAsync(p, 0, bytes, st) == hipSuccess ? 0 : 1;
    const size_t n = bytes / 4;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_fill_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, static_cast<unsigned*>(p), n);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Channel slab per block: large enough to amortise the per-pixel set-up, small enough that the
// grid has many blocks per CU on all 256 CUs.
Geometry plan(int64_t B, int64_t C, int64_t H, int64_t W, int cs_default) {
    Geometry g;
    g.tiles_x = static_cast<int>((W + kTileX - 1) / kTileX);
    g.tiles_y = static_cast<int>((H + kTileY - 1) / kTileY);
    const int64_t spatial = B * g.tiles_x * g.tiles_y;
    int cs = options().channel_slab > 0 ? options().channel_slab : cs_default;
    if (cs > C) cs = static_cast<int>(C);
    while (cs > 1 && spatial * ((C + cs - 1) / cs) < 4096) cs = (cs + 1) / 2;   // >= 16 blocks per CU
    g.cs = cs;
    g.cslabs = static_cast<int>((C + cs - 1) / cs);
    g.grid = static_cast<unsigned>(spatial * g.cslabs);
    return g;
}

// ------------------------------------------------------------------ profiler
namespace {
struct Pending {
    const char* name;
    hipEvent_t start, stop;
    double bytes, flops;
};
struct Row {
    int64_t launches = 0;
    double ms = 0, bytes = 0, flops = 0;
    double bound_ms = 0;      // sum over launches of max(bytes / HBM peak, flops / fp32 MFMA peak): the time the binding roofline allows
};
std::mutex g_mu;
bool g_prof = false;
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_pool;
std::map<std::string, Row> g_rows;
std::vector<std::pair<std::string, Row>> g_snapshot;

hipEvent_t take_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

LaunchScope::LaunchScope(const char* name, hipStream_t stream, double bytes, double flops)
    : slot_(-1), stream_(stream) {
    if (!g_prof) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Pending p{name, take_event(), take_event(), bytes, flops};
    if (!p.start || !p.stop) return;
    (void)hipEventRecord(p.start, stream);
    g_pending.push_back(p);
    slot_ = static_cast<int>(g_pending.size()) - 1;
}

LaunchScope::~LaunchScope() {
    if (slot_ < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot_ < static_cast<int>(g_pending.size())) (void)hipEventRecord(g_pending[slot_].stop, stream_);
}

}  // namespace ffwm

using namespace ffwm;

extern "C" {

int ffwm_abi_version(void) { return FFWM_ABI_VERSION; }

const char* ffwm_last_error(void) { return g_err; }

int ffwm_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    int prev = g_prof ? 1 : 0;
    g_prof = on != 0;
    return prev;
}

int ffwm_prof_collect(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.stop) == hipSuccess &&
            hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            Row& r = g_rows[p.name];
            r.launches += 1;
            r.ms += ms;
            r.bytes += p.bytes;
            r.flops += p.flops;
            const double tb = p.bytes / 8.0e12, tf = p.flops / 157.3e12;
            r.bound_ms += (tb > tf ? tb : tf) * 1e3;
        }
        g_pool.push_back(p.start);
        g_pool.push_back(p.stop);
    }
    g_pending.clear();
    g_snapshot.assign(g_rows.begin(), g_rows.end());
    return static_cast<int>(g_snapshot.size());
}

int ffwm_prof_get(int row, char* name, int name_len, int64_t* launches, double* total_ms,
                  double* algorithmic_bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (row < 0 || row >= static_cast<int>(g_snapshot.size())) {
        set_error("ffwm_prof_get: row %d out of range", row);
        return FFWM_ERR_ARG;
    }
    const auto& kv = g_snapshot[row];
    if (name && name_len > 0) {
        strncpy(name, kv.first.c_str(), name_len - 1);
        name[name_len - 1] = 0;
    }
    if (launches) *launches = kv.second.launches;
    if (total_ms) *total_ms = kv.second.ms;
    if (algorithmic_bytes) *algorithmic_bytes = kv.second.bytes;
    return FFWM_OK;
}

int ffwm_prof_get_flops(int row, double* algorithmic_flops) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (row < 0 || row >= static_cast<int>(g_snapshot.size())) {
        set_error("ffwm_prof_get_flops: row %d out of range", row);
        return FFWM_ERR_ARG;
    }
    if (algorithmic_flops) *algorithmic_flops = g_snapshot[row].second.flops;
    return FFWM_OK;
}

int ffwm_prof_get_bound(int row, double* roofline_ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (row < 0 || row >= static_cast<int>(g_snapshot.size())) {
        set_error("ffwm_prof_get_bound: row %d out of range", row);
        return FFWM_ERR_ARG;
    }
    if (roofline_ms) *roofline_ms = g_snapshot[row].second.bound_ms;
    return FFWM_OK;
}

int ffwm_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& p : g_pending) {
        (void)hipEventSynchronize(p.stop);
        g_pool.push_back(p.start);
        g_pool.push_back(p.stop);
    }
    g_pending.clear();
    g_rows.clear();
    g_snapshot.clear();
    return FFWM_OK;
}

int ffwm_zero_fill(void* p, int64_t bytes, void* stream) {
    FFWM_REQUIRE(bytes >= 0 && (bytes == 0 || p) && (bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 3u) == 0, FFWM_ERR_ARG,
                 "ffwm_zero_fill: a 4-byte aligned address and a multiple of 4 bytes");
    if (zero_fill(p, static_cast<size_t>(bytes), static_cast<hipStream_t>(stream))) {
        set_error("ffwm_zero_fill: launch failed");
        return FFWM_ERR_LAUNCH;
    }
    return FFWM_OK;
}

int ffwm_set_option(const char* key, int value) {
    if (!key) return FFWM_ERR_ARG;
    Options& o = options();
    int* slot = nullptr;
    if (!strcmp(key, "be_fwd_variant")) slot = &o.be_fwd_variant;
    else if (!strcmp(key, "be_bwd_variant")) slot = &o.be_bwd_variant;
    else if (!strcmp(key, "channel_slab")) slot = &o.channel_slab;
    else if (!strcmp(key, "xcd_remap")) slot = &o.xcd_remap;
    else if (!strcmp(key, "ablate")) slot = &o.ablate;
    else if (!strcmp(key, "warp_multi_order")) slot = &o.warp_multi_order;
    else if (!strcmp(key, "rows_per_thread")) slot = &o.rows_per_thread;
    else if (!strcmp(key, "scatter_variant")) slot = &o.scatter_variant;
    else if (!strcmp(key, "be_bwd_halo")) slot = &o.be_bwd_halo;
    else if (!strcmp(key, "warp_fwd_variant")) slot = &o.warp_fwd_variant;
    else if (!strcmp(key, "warp_nt")) slot = &o.warp_nt;
    else if (!strcmp(key, "warp_pair_loads")) slot = &o.warp_pair_loads;
    else if (!strcmp(key, "warp_multi_lds")) slot = &o.warp_multi_lds;
    else if (!strcmp(key, "warp_multi_planes")) slot = &o.warp_multi_planes;
    else if (!strcmp(key, "conv_wgrad_wino")) slot = &o.conv_wgrad_wino;
    else if (!strcmp(key, "be_bwd_rows")) slot = &o.be_bwd_rows;
    else if (!strcmp(key, "rs_fwd_variant")) slot = &o.rs_fwd_variant;
    else if (!strcmp(key, "rs_bwd1_variant")) slot = &o.rs_bwd1_variant;
    else if (!strcmp(key, "conv_tile_variant")) slot = &o.conv_tile_variant;
    else if (!strcmp(key, "conv_thin_tail")) slot = &o.conv_thin_tail;
    else if (!strcmp(key, "conv_wino_raw")) slot = &o.conv_wino_raw;
    else if (!strcmp(key, "conv_wino_ws")) slot = &o.conv_wino_ws;
    else if (!strcmp(key, "conv_wgrad_slice_target")) slot = &o.conv_wgrad_slice_target;
    else if (!strcmp(key, "conv_fwd_split_target")) slot = &o.conv_fwd_split_target;
    else if (!strcmp(key, "conv_wino_split")) slot = &o.conv_wino_split;
    else if (!strcmp(key, "conv_wgrad_unsliced")) slot = &o.conv_wgrad_unsliced;
    else if (!strcmp(key, "conv_wgrad_prezeroed")) slot = &o.conv_wgrad_prezeroed;
    else if (!strcmp(key, "zero_fill_memset")) slot = &o.zero_fill_memset;
    else if (!strcmp(key, "be_bwd_fixed")) slot = &o.be_bwd_fixed;
    else if (!strcmp(key, "be_bwd_flush")) slot = &o.be_bwd_flush;
    else if (!strcmp(key, "ba_bwd_fused")) slot = &o.ba_bwd_fused;
    else if (!strcmp(key, "ba_bwd_pix")) slot = &o.ba_bwd_pix;
    else if (!strcmp(key, "ba_fwd_pix")) slot = &o.ba_fwd_pix;
    else if (!strcmp(key, "conv_thin_variant")) slot = &o.conv_thin_variant;
    else if (!strcmp(key, "rs_bwd1_owned")) slot = &o.rs_bwd1_owned;
    else if (!strcmp(key, "warp_feat_fixed")) slot = &o.warp_feat_fixed;
    else if (!strcmp(key, "rs_bwd1_owned_blocks")) slot = &o.rs_bwd1_owned_blocks;
    else if (!strcmp(key, "rs_bwd1_owned_min_pixels")) slot = &o.rs_bwd1_owned_min_pixels;
    else if (!strcmp(key, "rs_bwd1_fixed")) slot = &o.rs_bwd1_fixed;
    else if (!strcmp(key, "rs_bwd1_rpt")) slot = &o.rs_bwd1_rpt;
    else if (!strcmp(key, "warp_feat_gps")) slot = &o.warp_feat_gps;
    else if (!strcmp(key, "conv_fwd_kfast")) slot = &o.conv_fwd_kfast;
    if (!slot) {
        set_error("ffwm_set_option: unknown key '%s'", key);
        return FFWM_ERR_ARG;
    }
    int prev = *slot;
    *slot = value;
    return prev;
}

}  // extern "C"
