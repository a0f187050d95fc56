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

Tensor bn_scratch(const Tensor& x) {
    const int64_t C = x.size(1);
    if (C < 512 && x.numel() / C >= 32768) return torch::zeros({2 * C}, x.options().dtype(torch::kFloat64));
    return Tensor();
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

// ---------------------------------------------------------------- MFMA convolution forward  (conv.py, flownet_eval.conv_mfma)
// y = conv(x, w) [+ bias]; a layer with few output pixels is cut along its reduction and finished by the bias pass.
// mode: 0 conv, 1 ConvTranspose2d(4, 2, 1), 2 / 3: d(input) of Conv2d(3, 2, 1) / Conv2d(3, 1, 1) (x = grad_output)
Tensor conv_fwd_raw(const Tensor& x, const Tensor& w, const Tensor& bias, int64_t stride, int64_t pad, int mode) {
    const c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
    const int64_t k = w.size(2);
    const bool parity = mode == 1 || mode == 2;
    const int64_t K = mode == 0 ? w.size(0) : w.size(1);
    const int64_t Ho = parity ? 2 * H : (mode == 3 ? H : (H + 2 * pad - k) / stride + 1);
    const int64_t Wo = parity ? 2 * W : (mode == 3 ? W : (W + 2 * pad - k) / stride + 1);
    const bool small = B * Ho * Wo * ((K + 63) / 64) * (parity ? 1 : 4) < 256 * 64 * 4;
    Tensor y = small ? torch::zeros({B, K, Ho, Wo}, x.options()) : torch::empty({B, K, Ho, Wo}, x.options());
    int flag = 0;
    check(ffwm_conv2d_forward(x.data_ptr(), w.data_ptr(), cptr(bias), y.data_ptr(), B, C, H, W, K, static_cast<int>(k),
                              static_cast<int>(stride), static_cast<int>(pad), mode, K * Ho * Wo, 0, 0.0, small ? 1 : 0,
                              &flag, FFWM_F32, stream_of(x)), "ffwm_conv2d_forward");
    if (flag && has(bias))
        check(ffwm_bias_act_forward(y.data_ptr(), bias.data_ptr(), y.data_ptr(), nullptr, B, K, Ho * Wo, K * Ho * Wo, 0, 0, 0.0, FFWM_F32,
                                    stream_of(x)), "ffwm_bias_act_forward");
    return y;
}

// Which parts of the routed convolutions' backward run on the hand-written kernels (the rest goes to ATen / MIOpen):
// FFWM_CONV_DGRAD / FFWM_CONV_WGRAD = 1 | 0, read once.
bool env_flag(const char* name, bool dflt) {
    const char* v = std::getenv(name);
    return v ? (v[0] == '1') : dflt;
}
bool use_dgrad() { static const bool v = env_flag("FFWM_CONV_DGRAD", false); return v; }
bool use_wgrad() { static const bool v = env_flag("FFWM_CONV_WGRAD", false); return v; }

// grad_weight [K, C, k, k] (rows = grad_output, gathered = input) or, for the transposed convolution, [Ci, Co, 4, 4]
Tensor wgrad_raw(const Tensor& rows, const Tensor& gathered, int64_t k, int64_t stride, int64_t pad) {
    const c10::DeviceGuard guard(rows.device());
    Tensor gw = torch::zeros({rows.size(1), gathered.size(1), k, k}, rows.options());
    check(ffwm_conv2d_wgrad(rows.data_ptr(), gathered.data_ptr(), gw.data_ptr(), rows.size(0), rows.size(1), rows.size(2), rows.size(3),
                            gathered.size(1), gathered.size(2), gathered.size(3), static_cast<int>(k), static_cast<int>(stride),
                            static_cast<int>(pad), FFWM_F32, stream_of(rows)), "ffwm_conv2d_wgrad");
    return gw;
}

struct MfmaConv2d : public torch::autograd::Function<MfmaConv2d> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& weight, const Tensor& bias, int64_t stride, int64_t pad) {
        Tensor xc = x.contiguous(), wc = weight.contiguous();
        ctx->save_for_backward({xc, wc});
        ctx->saved_data["stride"] = stride;
        ctx->saved_data["pad"] = pad;
        ctx->saved_data["has_bias"] = has(bias);
        return conv_fwd_raw(xc, wc, bias, stride, pad, 0);
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto saved = ctx->get_saved_variables();
        const Tensor &x = saved[0], &weight = saved[1];
        const int64_t stride = ctx->saved_data["stride"].toInt(), pad = ctx->saved_data["pad"].toInt();
        const bool has_bias = ctx->saved_data["has_bias"].toBool();
        Tensor go = grads[0].contiguous();
        bool need_x = ctx->needs_input_grad(0);
        const bool need_w = ctx->needs_input_grad(1), need_b = has_bias && ctx->needs_input_grad(2);
        Tensor gx, gxa, gw, gb;
        const int64_t k = weight.size(2);
        const bool even = x.size(2) == 2 * go.size(2) && x.size(3) == 2 * go.size(3);
        const Tensor none = torch::empty({0}, go.options());
        if (need_x && use_dgrad() && stride == 2 && pad == 1 && even) {
            // d(input) of Conv2d(C, K, 4, 2, 1) = ConvTranspose2d(K, C, 4, 2, 1) with the SAME weight tensor; of Conv2d(C, K, 3, 2, 1)
            // = the transposed 3x3 with output padding 1 (parity classes with 1 or 2 taps per axis)
            gx = conv_fwd_raw(go, weight, none, 2, 1, k == 4 ? 1 : 2);
            need_x = false;
        } else if (need_x && use_dgrad() && k == 3 && stride == 1 && pad == 1) {
            gx = conv_fwd_raw(go, weight, none, 1, 1, 3);
            need_x = false;
        }
        bool aten_w = false;
        if (need_w && use_wgrad()) gw = wgrad_raw(go, x, k, stride, pad);
        else aten_w = need_w;
        if (need_b) gb = go.sum(std::vector<int64_t>{0, 2, 3});
        if (need_x || aten_w) {
            Tensor gw2, gb2;
            std::tie(gxa, gw2, gb2) = at::convolution_backward(go, x, weight, c10::nullopt, {stride, stride}, {pad, pad}, {1, 1}, false,
                                                               {0, 0}, 1, {need_x, aten_w, false});
            if (aten_w) gw = gw2;
        }
        return {gx.defined() ? gx : gxa, gw, gb, Tensor(), Tensor()};
    }
};

struct MfmaConvTranspose2d : public torch::autograd::Function<MfmaConvTranspose2d> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& weight, const Tensor& bias) {
        Tensor xc = x.contiguous(), wc = weight.contiguous();
        ctx->save_for_backward({xc, wc});
        ctx->saved_data["has_bias"] = has(bias);
        return conv_fwd_raw(xc, wc, bias, 2, 1, 1);
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto saved = ctx->get_saved_variables();
        const Tensor &x = saved[0], &weight = saved[1];
        const bool has_bias = ctx->saved_data["has_bias"].toBool();
        Tensor go = grads[0].contiguous();
        const bool need_w = ctx->needs_input_grad(1), need_b = has_bias && ctx->needs_input_grad(2);
        Tensor gx, gw, gb;
        // d(input) of ConvTranspose2d(C, K, 4, 2, 1) = Conv2d(K, C, 4, 2, 1) with the same weight tensor [C, K, 4, 4]
        bool aten_x = false, aten_w = false;
        if (ctx->needs_input_grad(0) && use_dgrad()) gx = conv_fwd_raw(go, weight, torch::empty({0}, go.options()), 2, 1, 0);
        else aten_x = ctx->needs_input_grad(0);
        // d(weight): a Conv2d weight gradient with the two tensors' roles swapped (rows = input, gathered = grad_output)
        if (need_w && use_wgrad()) gw = wgrad_raw(x, go, 4, 2, 1);
        else aten_w = need_w;
        if (need_b) gb = go.sum(std::vector<int64_t>{0, 2, 3});
        if (aten_x || aten_w) {
            Tensor gx2, gw2, gb2;
            std::tie(gx2, gw2, gb2) = at::convolution_backward(go, x, weight, c10::nullopt, {2, 2}, {1, 1}, {1, 1}, true, {0, 0}, 1,
                                                               {aten_x, aten_w, false});
            if (aten_x) gx = gx2;
            if (aten_w) gw = gw2;
        }
        return {gx, gw, gb};
    }
};

// ---------------------------------------------------------------- conv2d (3x3 / s1 / p1) with the MFMA weight gradient  (conv.py)
struct Conv3x3MfmaWgrad : public torch::autograd::Function<Conv3x3MfmaWgrad> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& weight, const Tensor& bias) {
        ctx->save_for_backward({x, weight});
        ctx->saved_data["has_bias"] = has(bias);
        return at::convolution(x, weight, has(bias) ? c10::optional<Tensor>(bias) : c10::nullopt, {1, 1}, {1, 1}, {1, 1}, false, {0, 0}, 1);
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto saved = ctx->get_saved_variables();
        const Tensor &x = saved[0], &weight = saved[1];
        const bool has_bias = ctx->saved_data["has_bias"].toBool();
        Tensor go = grads[0].contiguous();
        const bool need_x = ctx->needs_input_grad(0), need_w = ctx->needs_input_grad(1), need_b = has_bias && ctx->needs_input_grad(2);
        Tensor gx, gw, gb;
        if (need_w) {
            const c10::DeviceGuard guard(x.device());
            Tensor xc = x.contiguous();
            const int64_t B = xc.size(0), C = xc.size(1), H = xc.size(2), W = xc.size(3), K = go.size(1);
            gw = torch::zeros({K, C, 3, 3}, xc.options());
            if (need_b) gb = torch::zeros({K}, xc.options());
            check(ffwm_conv3x3_wgrad(xc.data_ptr(), go.data_ptr(), gw.data_ptr(), mptr(gb), B, C, K, H, W, FFWM_F32, stream_of(xc)),
                  "ffwm_conv3x3_wgrad");
        }
        if (need_x || (need_b && !gb.defined())) {
            const bool nb = need_b && !gb.defined();
            c10::optional<c10::IntArrayRef> bias_sizes;
            std::vector<int64_t> bs{weight.size(0)};
            if (nb) bias_sizes = c10::IntArrayRef(bs);
            Tensor gw2, gb2;
            std::tie(gx, gw2, gb2) = at::convolution_backward(go, x, weight, bias_sizes, {1, 1}, {1, 1}, {1, 1}, false, {0, 0}, 1,
                                                              {need_x, false, nb});
            if (nb) gb = gb2;
        }
        return {gx, gw, gb};
    }
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "C++ autograd bindings of libffwm_hip.so (see ffwm_amd/csrc_ext/ffwm_torch.cpp)";
    m.def("bn_lrelu", [](const Tensor& x, const c10::optional<Tensor>& w, const c10::optional<Tensor>& b, const c10::optional<Tensor>& rm,
                         const c10::optional<Tensor>& rv, double eps, double momentum, double slope) {
        return BnLrelu::apply(x, opt(w, x), opt(b, x), opt(rm, x), opt(rv, x), eps, momentum, slope);
    });
    m.def("bias_relu", [](const Tensor& h, const Tensor& bias) { return BiasRelu::apply(h, bias); });
    m.def("mfm", [](const Tensor& x, const c10::optional<Tensor>& bias) { return Mfm::apply(x, opt(bias, x)); });
    m.def("conv2d", [](const Tensor& x, const Tensor& w, const c10::optional<Tensor>& bias, int64_t stride, int64_t pad) {
        return MfmaConv2d::apply(x, w, opt(bias, x), stride, pad);
    });
    m.def("conv_transpose2d", [](const Tensor& x, const Tensor& w, const c10::optional<Tensor>& bias) {
        return MfmaConvTranspose2d::apply(x, w, opt(bias, x));
    });
    m.def("conv3x3_mfma_wgrad", [](const Tensor& x, const Tensor& w, const c10::optional<Tensor>& bias) {
        return Conv3x3MfmaWgrad::apply(x, w, opt(bias, x));
    });
}
