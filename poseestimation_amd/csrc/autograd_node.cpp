// frobenius_head's autograd node in C++ (the Python class _FrobeniusHead in rotation_representation.py is its twin and serves
// every case this one declines).  Config #4 (B = 512, bfloat16: 3D-Pose/main.py:60,85,90) is launch-bound: the kernels take 4 us,
// a Python autograd.Function costs 30 us of interpreter and engine bookkeeping before it launches anything.  A C++ node takes the
// interpreter out of forward and backward; what it launches is the same C ABI (include/so3proj.h), reached through function
// addresses the Python side hands over once (no link-time dependency on libso3proj.so, no HIP headers here: the stream is an
// integer from torch's accessor, the device guard is c10's generic one).
//
// Handles: x float32 / bfloat16, contiguous, (B,9) or (B,3,3), B >= 1; R_true float32, contiguous, same device, not requiring
// grad.  Anything else: the caller uses the Python class.
#include <torch/extension.h>
#include <c10/core/DeviceGuard.h>

namespace {

typedef int (*FrobFn)(const void *M, const float *Rtrue, float *R, void *dM, double *loss_sum, float *loss_mean, void *workspace,
                      int64_t B, void *stream);                                      // so3_frob_fwd_bwd_ws_{f32,bf16}
typedef int (*ScaleFn)(const void *src, const float *factor, void *dst, int64_t n, void *stream);   // so3_scale_{f32,bf16}
typedef const char *(*ErrFn)();                                                      // so3_last_error

struct Entry {
    FrobFn frob_f32 = nullptr, frob_bf16 = nullptr;
    ScaleFn scale_f32 = nullptr, scale_bf16 = nullptr;
    ErrFn last_error = nullptr;
    int64_t small_batch = 0;
} g_entry;

void check(int code, const char *what) {
    TORCH_CHECK(code == 0, what, " failed with code ", code, ": ", g_entry.last_error ? g_entry.last_error() : "");
}

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct FrobeniusHeadNode : public torch::autograd::Function<FrobeniusHeadNode> {
    // returns {loss} or {loss, R}; R carries no gradient
    static variable_list forward(AutogradContext *ctx, const at::Tensor &x, const at::Tensor &r_true, bool want_r, int64_t stream,
                                 int64_t workspace) {
        const int64_t b = x.numel() / 9;
        const bool bf16 = x.scalar_type() == at::kBFloat16;
        const bool need_grad = x.requires_grad();
        const auto f32 = x.options().dtype(at::kFloat);
        at::Tensor r = want_r ? at::empty({b, 3, 3}, f32) : at::Tensor();
        at::Tensor dm = need_grad ? at::empty({b, 9}, x.options()) : at::Tensor();
        at::Tensor loss = at::empty({}, f32);                   // the kernel writes the float32 mean itself
        at::Tensor loss_sum = b > g_entry.small_batch ? at::empty({1}, x.options().dtype(at::kDouble)) : at::Tensor();
        {
            c10::DeviceGuard guard(x.device());
            check((bf16 ? g_entry.frob_bf16 : g_entry.frob_f32)(
                      x.data_ptr(), static_cast<const float *>(r_true.data_ptr()), r.defined() ? static_cast<float *>(r.data_ptr()) : nullptr,
                      dm.defined() ? dm.data_ptr() : nullptr, loss_sum.defined() ? static_cast<double *>(loss_sum.data_ptr()) : nullptr,
                      static_cast<float *>(loss.data_ptr()), reinterpret_cast<void *>(workspace), b, reinterpret_cast<void *>(stream)),
                  "so3_frob_fwd_bwd");
        }
        if (need_grad) {
            ctx->saved_data["dm"] = dm;                         // ours, not an input or output: no version-counter bookkeeping needed
            ctx->saved_data["shape"] = x.sizes().vec();
            ctx->saved_data["stream"] = stream;                 // the engine runs backward on the forward's stream
        }
        if (!r.defined()) return {loss};
        ctx->mark_non_differentiable({r});
        return {loss, r};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads) {
        TORCH_CHECK(!at::GradMode::is_enabled(),
                    "trying to differentiate twice a function that was marked with @once_differentiable "
                    "(poseestimation_amd kernels do not support double backward; the reference never uses it)");
        const at::Tensor dm = ctx->saved_data["dm"].toTensor();
        const at::Tensor &g = grads[0];
        at::Tensor gx;
        if (g.defined() && g.scalar_type() == at::kFloat && g.is_cuda() && g.numel() == 1) {
            // out of place: a second backward over the same graph must find the stored gradient unscaled
            gx = at::empty_like(dm);
            const bool bf16 = dm.scalar_type() == at::kBFloat16;
            check((bf16 ? g_entry.scale_bf16 : g_entry.scale_f32)(dm.data_ptr(), static_cast<const float *>(g.data_ptr()), gx.data_ptr(), dm.numel(),
                                                                    reinterpret_cast<void *>(ctx->saved_data["stream"].toInt())),
                  "so3_scale");
        } else if (g.defined()) {
            gx = (dm.to(at::kFloat) * g).to(dm.scalar_type());
        }
        if (gx.defined()) gx = gx.view(ctx->saved_data["shape"].toIntVector());
        return {gx, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

void bind(int64_t frob_f32, int64_t frob_bf16, int64_t scale_f32, int64_t scale_bf16, int64_t last_error, int64_t small_batch) {
    g_entry.frob_f32 = reinterpret_cast<FrobFn>(frob_f32);
    g_entry.frob_bf16 = reinterpret_cast<FrobFn>(frob_bf16);
    g_entry.scale_f32 = reinterpret_cast<ScaleFn>(scale_f32);
    g_entry.scale_bf16 = reinterpret_cast<ScaleFn>(scale_bf16);
    g_entry.last_error = reinterpret_cast<ErrFn>(last_error);
    g_entry.small_batch = small_batch;
}

// None when the arguments are not the node's case (the caller then takes the Python class), else (loss, R or None).
py::object frobenius_head(const at::Tensor &x, const at::Tensor &r_true, bool want_r, int64_t stream, int64_t workspace) {
    const auto dt = x.scalar_type();
    const int64_t d = x.dim();
    const bool shape_ok = (d == 2 && x.size(1) == 9) || (d == 3 && x.size(1) == 3 && x.size(2) == 3);
    if (g_entry.frob_f32 == nullptr || !shape_ok || x.size(0) < 1 || !x.is_cuda() || (dt != at::kFloat && dt != at::kBFloat16) || !x.is_contiguous() ||
        r_true.scalar_type() != at::kFloat || !r_true.is_contiguous() || r_true.numel() != x.numel() || r_true.device() != x.device() ||
        (r_true.requires_grad() && at::GradMode::is_enabled()) || (x.size(0) > g_entry.small_batch && workspace == 0))
        return py::none();
    variable_list out = FrobeniusHeadNode::apply(x, r_true, want_r, stream, workspace);
    return py::make_tuple(out[0], out.size() > 1 ? py::cast(out[1]) : py::none());
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("bind", &bind, "addresses of the C-ABI entry points (so3_frob_fwd_bwd_ws_*, so3_scale_*, so3_last_error) and kSmallBatch");
    m.def("frobenius_head", &frobenius_head);
}
