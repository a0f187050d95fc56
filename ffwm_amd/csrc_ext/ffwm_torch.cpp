// ffwm_torch.cpp -- C++ autograd bindings of the hottest libffwm_hip.so entry points (a pybind11 torch extension).
//
// The Python autograd Functions around the hand-written kernels -- ctypes marshalling, torch.empty calls, Python-level
// bookkeeping -- cost 14-35 us of host time per call (tools/host_profile.py), as much as one of MIOpen's own convolution
// calls, ~750 times per train step; the same Functions in C++ cost a few microseconds.  (Measured: the FFWM step itself
// is bound by GPU-side dispatch gaps, not by the host, so this buys host headroom -- for the RCCL launches of the
// data-parallel step -- rather than step time: DESIGN.md section 6.)  Nothing is computed here: every function allocates
// its outputs, picks the current HIP stream and calls the C ABI (include/ffwm_hip.h), exactly as ffwm_amd/norm.py,
// ffwm_amd/conv.py and ffwm_amd/external_function.py do through ctypes (those stay as the reference implementation of
// the binding and as the path tests/test_gpu_ext.py compares against).
#include <torch/extension.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/ffwm_hip.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed (status ", rc, "): ", ffwm_last_error());
}

// An absent optional tensor travels through Function::apply as a 0-element tensor on the same device (apply() asks every
// tensor argument for its device; an undefined one throws).
bool has(const Tensor& t) { return t.defined() && t.numel() > 0; }
const void* cptr(const Tensor& t) { return has(t) ? t.data_ptr() : nullptr; }
void* mptr(const Tensor& t) { return has(t) ? t.data_ptr() : nullptr; }
Tensor opt(const c10::optional<Tensor>& t, const Tensor& like) { return t.has_value() && t->defined() ? *t : torch::empty({0}, like.options()); }

// One zero-filled scratch per (device, C, stream), filled once: the kernels hand it back zero-filled (include/ffwm_hip.h) and the
// calls that share it are ordered on their stream.  (Round 2 filled a fresh one per call: 60 fill launches per train step.)
std::mutex& scratch_mu() { static std::mutex mu; return mu; }
std::map<std::tuple<int, int64_t, void*>, Tensor>& scratch_cache() {
    static std::map<std::tuple<int, int64_t, void*>, Tensor> cache;
    return cache;
}
// A buffer first made inside a hipGraph capture lives in that graph's private pool and its zero-fill is a node of that graph only:
// capture() drops every cached buffer before it starts (norm.reset_scratch), so each capture makes -- and fills, on every replay --
// its own.
void bn_scratch_reset() {
    std::lock_guard<std::mutex> lock(scratch_mu());
    scratch_cache().clear();
}
Tensor bn_scratch(const Tensor& x) {
    const int64_t C = x.size(1);
    if (!(C < 512 && x.numel() / C >= 32768)) return Tensor();
    std::mutex& mu = scratch_mu();
    auto& cache = scratch_cache();
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_tuple(static_cast<int>(x.device().index()), C, stream_of(x));
    auto it = cache.find(key);
    if (it == cache.end()) it = cache.emplace(key, torch::zeros({2 * C + (C + 1) / 2}, x.options().dtype(torch::kFloat64))).first;
    return it->second;
}

// ---------------------------------------------------------------- BatchNorm2d (training) + LeakyReLU  (norm.py)
struct BnLrelu : public torch::autograd::Function<BnLrelu> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& weight, const Tensor& bias, Tensor running_mean,
                          Tensor running_var, double eps, double momentum, double slope) {
        const c10::DeviceGuard guard(x.device());
        const int64_t B = x.size(0), C = x.size(1), HW = x.size(2) * x.size(3);
        Tensor y = torch::empty_like(x);
        Tensor save_mean = torch::empty({C}, x.options()), save_invstd = torch::empty({C}, x.options());
        Tensor scratch = bn_scratch(x);
        check(ffwm_bn_lrelu_forward(x.data_ptr(), cptr(weight), cptr(bias), mptr(running_mean), mptr(running_var), y.data_ptr(),
                                    save_mean.data_ptr(), save_invstd.data_ptr(), mptr(scratch), B, C, HW, eps, momentum, slope,
                                    FFWM_F32, stream_of(x)), "ffwm_bn_lrelu_forward");
        ctx->save_for_backward({x, weight, bias, save_mean, save_invstd});
        ctx->saved_data["slope"] = slope;
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto saved = ctx->get_saved_variables();
        const Tensor &x = saved[0], &weight = saved[1], &bias = saved[2], &save_mean = saved[3], &save_invstd = saved[4];
        const c10::DeviceGuard guard(x.device());
        const int64_t B = x.size(0), C = x.size(1), HW = x.size(2) * x.size(3);
        Tensor go = grads[0].contiguous();
        Tensor dx = ctx->needs_input_grad(0) ? torch::empty_like(x) : Tensor();
        Tensor dw = (ctx->needs_input_grad(1) && has(weight)) ? torch::empty({C}, x.options()) : Tensor();
        Tensor db = (ctx->needs_input_grad(2) && has(bias)) ? torch::empty({C}, x.options()) : Tensor();
        Tensor scratch = bn_scratch(x);
        check(ffwm_bn_lrelu_backward(x.data_ptr(), go.data_ptr(), cptr(weight), cptr(bias), save_mean.data_ptr(), save_invstd.data_ptr(),
                                     mptr(dx), mptr(dw), mptr(db), mptr(scratch), B, C, HW, ctx->saved_data["slope"].toDouble(),
                                     FFWM_F32, stream_of(x)), "ffwm_bn_lrelu_backward");
        return {dx, dw, db, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

// ---------------------------------------------------------------- relu(h + bias[c])  (external_function.BiasReLUFunction)
struct BiasRelu : public torch::autograd::Function<BiasRelu> {
    static Tensor forward(AutogradContext* ctx, const Tensor& h, const Tensor& bias) {
        const c10::DeviceGuard guard(h.device());
        Tensor hc = h.contiguous();
        const int64_t B = hc.size(0), C = hc.size(1), HW = hc.numel() / (B * C);
        Tensor y = torch::empty_like(hc);
        check(ffwm_bias_relu_forward(hc.data_ptr(), bias.data_ptr(), y.data_ptr(), B, C, HW, FFWM_F32, stream_of(hc)), "ffwm_bias_relu_forward");
        ctx->save_for_backward({y});
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        Tensor y = ctx->get_saved_variables()[0];
        Tensor gh = at::threshold_backward(grads[0], y, 0);
        Tensor gb;
        if (ctx->needs_input_grad(1)) {
            std::vector<int64_t> dims{0};
            for (int64_t d = 2; d < gh.dim(); ++d) dims.push_back(d);
            gb = gh.sum(dims);
        }
        return {gh, gb};
    }
};

// ---------------------------------------------------------------- LightCNN max-feature-map  (external_function.MaxFeatureMapFunction)
struct Mfm : public torch::autograd::Function<Mfm> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& bias) {
        TORCH_CHECK(x.is_contiguous() && x.size(1) % 2 == 0, "mfm: need a contiguous [B, 2C, ...] tensor");
        const c10::DeviceGuard guard(x.device());
        const int64_t B = x.size(0), C = x.size(1) / 2, HW = x.numel() / (B * 2 * C);
        std::vector<int64_t> shape = x.sizes().vec();
        shape[1] = C;
        Tensor y = torch::empty(shape, x.options());
        check(ffwm_mfm_forward(x.data_ptr(), cptr(bias), y.data_ptr(), B, C, HW, FFWM_F32, stream_of(x)), "ffwm_mfm_forward");
        ctx->save_for_backward({x, bias});
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto saved = ctx->get_saved_variables();
        const Tensor &x = saved[0], &bias = saved[1];
        const int64_t B = x.size(0), C = x.size(1) / 2, HW = x.numel() / (B * 2 * C);
        const c10::DeviceGuard guard(x.device());
        Tensor gy = grads[0].contiguous();
        Tensor dx = torch::empty_like(x);
        check(ffwm_mfm_backward(x.data_ptr(), cptr(bias), gy.data_ptr(), dx.data_ptr(), B, C, HW, FFWM_F32, stream_of(x)), "ffwm_mfm_backward");
        Tensor db;
        if (has(bias) && ctx->needs_input_grad(1)) {
            std::vector<int64_t> dims{0};
            for (int64_t d = 2; d < dx.dim(); ++d) dims.push_back(d);
            db = dx.sum(dims);
        }
        return {dx, db};
    }
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "C++ autograd bindings of libffwm_hip.so (see ffwm_amd/csrc_ext/ffwm_torch.cpp)";
    m.def("bn_lrelu", [](const Tensor& x, const c10::optional<Tensor>& w, const c10::optional<Tensor>& b, const c10::optional<Tensor>& rm,
                         const c10::optional<Tensor>& rv, double eps, double momentum, double slope) {
        return BnLrelu::apply(x, opt(w, x), opt(b, x), opt(rm, x), opt(rv, x), eps, momentum, slope);
    });
    m.def("bn_scratch_reset", &bn_scratch_reset);
    m.def("bias_relu", [](const Tensor& h, const Tensor& bias) { return BiasRelu::apply(h, bias); });
    m.def("mfm", [](const Tensor& x, const c10::optional<Tensor>& bias) { return Mfm::apply(x, opt(bias, x)); });
}
