// The autograd nodes of the launch-bound training step in C++: frobenius_head (the fused spelling), and symmetric_orthogonalization
// and loss_frobenius (the reference's own two-call spelling, 3D-Pose/main.py:60,85).  The Python classes _FrobeniusHead,
// _SymmetricOrthogonalization and _LossFrobenius in rotation_representation.py are their twins and serve every case these decline.  Config #4 (B = 512, bfloat16: 3D-Pose/main.py:60,85,90) is launch-bound: the kernels take 4 us,
// a Python autograd.Function costs 30 us of interpreter and engine bookkeeping before it launches anything.  A C++ node takes the
// interpreter out of forward and backward; what it launches is the same C ABI (include/so3proj.h), reached through function
// addresses the Python side hands over once (no link-time dependency on libso3proj.so, no device code here).
// STREAMS: forward launches on the stream the caller passes (the Python side's torch._C._cuda_getCurrentRawStream); backward
// launches on the stream that is CURRENT when the engine runs the node (c10::hip::getCurrentHIPStream) -- the rule of the Python
// classes, which ask for the current stream in both places.  (Round 3 kept the forward's raw handle as an integer for backward:
// the engine makes the forward's stream current for the node anyway, but a handle must not outlive what it names.)
//
// frobenius_head handles: x float32 / bfloat16, contiguous, (B,9) or (B,3,3), B >= 1; R_true float32, contiguous, same device, not
// requiring grad.  symmetric_orthogonalization: x float32 / bfloat16, contiguous, numel a multiple of 9, requiring grad.
// loss_frobenius: two float32 contiguous tensors of B x 9 elements on one device, at least one requiring grad.  row_head (the 6D
// head and the other row-operation heads): float32, contiguous, requiring grad.  Anything else: the caller uses the Python class.
#include <torch/extension.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <vector>

namespace {

typedef int (*FrobFn)(const void *M, const float *Rtrue, float *R, void *dM, double *loss_sum, float *loss_mean, void *workspace,
                      unsigned flags, int64_t B, void *stream);                      // so3_frob_fwd_bwd_v2_{f32,bf16}
typedef int (*ScaleFn)(const void *src, const float *factor, void *dst, int64_t n, void *stream);   // so3_scale_{f32,bf16}
typedef const char *(*ErrFn)();                                                      // so3_last_error
typedef int (*FwdFn)(const void *M, float *R, uint8_t *flip, int64_t B, void *stream);             // so3_project_fwd_{f32,bf16}
typedef int (*BwdFn)(const void *M, const float *G, void *dM, int64_t B, void *stream);            // so3_project_bwd_{f32,bf16}
typedef int (*LossFn)(const float *Rpred, const float *Rtrue, float *dRpred, double *loss_sum, float *loss_mean, void *workspace,
                      unsigned flags, int64_t B, void *stream);                      // so3_frob_loss_v2_f32

struct Entry {
    FrobFn frob_f32 = nullptr, frob_bf16 = nullptr;
    ScaleFn scale_f32 = nullptr, scale_bf16 = nullptr;
    FwdFn fwd_f32 = nullptr, fwd_bf16 = nullptr;
    BwdFn bwd_f32 = nullptr, bwd_bf16 = nullptr;
    LossFn loss_f32 = nullptr;
    ErrFn last_error = nullptr;
    int64_t small_batch = 0;
} g_entry;

void check(int code, const char *what) {
    TORCH_CHECK(code == 0, what, " failed with code ", code, ": ", g_entry.last_error ? g_entry.last_error() : "");
}

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// once_differentiable's own condition: only a backward whose result would have to be differentiated AGAIN fails -- grad mode on
// (create_graph=True) AND an incoming gradient that requires grad.  create_graph=True alone (e.g. a gradient penalty on another
// branch of the graph) runs the kernels as always: nothing is recorded, the result is a constant.
void no_double_backward(const variable_list &grads) {
    if (!at::GradMode::is_enabled()) return;
    for (const at::Tensor &g : grads)
        TORCH_CHECK(!(g.defined() && g.requires_grad()),
                    "trying to differentiate twice a function that was marked with @once_differentiable "
                    "(poseestimation_amd kernels do not support double backward; the reference never uses it)");
}

// the stream the engine made current for this node, as the C ABI takes it
void *current_stream(const at::Tensor &t) { return static_cast<void *>(c10::hip::getCurrentHIPStream(t.device().index()).stream()); }

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
                      static_cast<float *>(loss.data_ptr()), reinterpret_cast<void *>(workspace), 0u, b, reinterpret_cast<void *>(stream)),
                  "so3_frob_fwd_bwd");
        }
        if (need_grad) {
            ctx->saved_data["dm"] = dm;                         // ours, not an input or output: no version-counter bookkeeping needed
            ctx->saved_data["shape"] = x.sizes().vec();
        }
        if (!r.defined()) return {loss};
        ctx->mark_non_differentiable({r});
        return {loss, r};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads) {
        no_double_backward(grads);
        const at::Tensor dm = ctx->saved_data["dm"].toTensor();
        const at::Tensor &g = grads[0];
        at::Tensor gx;
        if (g.defined() && g.scalar_type() == at::kFloat && g.is_cuda() && g.numel() == 1) {
            // out of place: a second backward over the same graph must find the stored gradient unscaled
            gx = at::empty_like(dm);
            const bool bf16 = dm.scalar_type() == at::kBFloat16;
            check((bf16 ? g_entry.scale_bf16 : g_entry.scale_f32)(dm.data_ptr(), static_cast<const float *>(g.data_ptr()), gx.data_ptr(), dm.numel(),
                                                                    current_stream(dm)),
                  "so3_scale");
        } else if (g.defined()) {
            gx = (dm.to(at::kFloat) * g).to(dm.scalar_type());
        }
        if (gx.defined()) gx = gx.view(ctx->saved_data["shape"].toIntVector());
        return {gx, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// symmetric_orthogonalization(x) with x requiring grad: K1 forward, K2 backward (rotation_representation.py:192-206 and its autograd).
struct ProjectNode : public torch::autograd::Function<ProjectNode> {
    static at::Tensor forward(AutogradContext *ctx, const at::Tensor &x, int64_t stream) {
        const int64_t b = x.numel() / 9;
        at::Tensor r = at::empty({b, 3, 3}, x.options().dtype(at::kFloat));
        {
            c10::DeviceGuard guard(x.device());
            check((x.scalar_type() == at::kBFloat16 ? g_entry.fwd_bf16 : g_entry.fwd_f32)(x.data_ptr(), static_cast<float *>(r.data_ptr()), nullptr, b,
                                                                                         reinterpret_cast<void *>(stream)),
                  "so3_project_fwd");
        }
        ctx->save_for_backward({x});
        return r;
    }
    static variable_list backward(AutogradContext *ctx, variable_list grads) {
        no_double_backward(grads);
        const at::Tensor x = ctx->get_saved_variables()[0];
        at::Tensor g = grads[0];
        if (!g.defined()) return {at::Tensor(), at::Tensor()};
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        g = g.contiguous();
        at::Tensor dm = at::empty_like(x);
        check((x.scalar_type() == at::kBFloat16 ? g_entry.bwd_bf16 : g_entry.bwd_f32)(x.data_ptr(), static_cast<const float *>(g.data_ptr()), dm.data_ptr(),
                                                                                     x.numel() / 9, current_stream(x)),
              "so3_project_bwd");
        return {dm, at::Tensor()};
    }
};

// loss_frobenius(a, b) = mean_b ||b - a||_F, differentiable in both arguments (3D-Pose/loss.py:7-11).  K3' writes the loss and its
// gradient with respect to its FIRST pointer; the loss is symmetric, so the argument that wants a gradient goes first (the
// reference's call sites pass the network's rotation second, 3D-Pose/main.py:85) and the other one's, if asked for, is the negative.
struct FrobLossNode : public torch::autograd::Function<FrobLossNode> {
    static at::Tensor forward(AutogradContext *ctx, const at::Tensor &a, const at::Tensor &b_, int64_t stream, int64_t workspace) {
        const int64_t b = a.numel() / 9;
        const bool first = a.requires_grad();              // gradient is taken with respect to `a` (else `b_`)
        const at::Tensor &p = first ? a : b_, &t = first ? b_ : a;
        const bool need_grad = a.requires_grad() || b_.requires_grad();
        at::Tensor g = need_grad ? at::empty_like(p) : at::Tensor();
        at::Tensor loss_sum = at::empty({1}, a.options().dtype(at::kDouble));
        at::Tensor loss = at::empty({}, a.options());
        {
            c10::DeviceGuard guard(a.device());
            check(g_entry.loss_f32(static_cast<const float *>(p.data_ptr()), static_cast<const float *>(t.data_ptr()),
                                   g.defined() ? static_cast<float *>(g.data_ptr()) : nullptr, static_cast<double *>(loss_sum.data_ptr()),
                                   static_cast<float *>(loss.data_ptr()), reinterpret_cast<void *>(workspace), 0u, b, reinterpret_cast<void *>(stream)),
                  "so3_frob_loss_f32");
        }
        if (need_grad) {
            ctx->saved_data["g"] = g;
            ctx->saved_data["first"] = first;
            ctx->saved_data["other_shape"] = t.sizes().vec();
        }
        return loss;
    }
    static variable_list backward(AutogradContext *ctx, variable_list grads) {
        no_double_backward(grads);
        const at::Tensor g = ctx->saved_data["g"].toTensor();
        const bool first = ctx->saved_data["first"].toBool();
        const at::Tensor &gl = grads[0];
        at::Tensor mine, other;
        if (gl.defined()) {
            if (gl.scalar_type() == at::kFloat && gl.is_cuda() && gl.numel() == 1) {
                mine = at::empty_like(g);
                check(g_entry.scale_f32(g.data_ptr(), static_cast<const float *>(gl.data_ptr()), mine.data_ptr(), g.numel(), current_stream(g)),
                      "so3_scale");
            } else {
                mine = g * gl;
            }
            if (ctx->needs_input_grad(first ? 1 : 0)) other = mine.neg().view(ctx->saved_data["other_shape"].toIntVector());
        }
        return first ? variable_list{mine, other, at::Tensor(), at::Tensor()} : variable_list{other, mine, at::Tensor(), at::Tensor()};
    }
};

// The heads that are plain row operations -- (..., width) float32 -> (..., 3, 3): the 6D Gram-Schmidt head (rotation_representation.py:21-36)
// and the quaternion / Euler / 5D / exponential-map heads of the reference's dispatch tables -- share one node; the entry points
// so3_<head>_fwd_f32(X, R, B, stream) / so3_<head>_bwd_f32(X, G, dX, B, stream) travel as addresses.
typedef int (*RowFwdFn)(const float *X, float *R, int64_t B, void *stream);
typedef int (*RowBwdFn)(const float *X, const float *G, float *dX, int64_t B, void *stream);
struct RowHeadNode : public torch::autograd::Function<RowHeadNode> {
    static at::Tensor forward(AutogradContext *ctx, const at::Tensor &x, int64_t width, int64_t fwd, int64_t bwd, int64_t stream) {
        const int64_t b = x.numel() / width;
        std::vector<int64_t> shape(x.sizes().begin(), x.sizes().end() - 1);
        shape.push_back(3);
        shape.push_back(3);
        at::Tensor r = at::empty(shape, x.options());
        {
            c10::DeviceGuard guard(x.device());
            check(reinterpret_cast<RowFwdFn>(fwd)(static_cast<const float *>(x.data_ptr()), static_cast<float *>(r.data_ptr()), b, reinterpret_cast<void *>(stream)),
                  "row head forward");
        }
        ctx->save_for_backward({x});
        ctx->saved_data["bwd"] = bwd;
        ctx->saved_data["width"] = width;
        return r;
    }
    static variable_list backward(AutogradContext *ctx, variable_list grads) {
        no_double_backward(grads);
        const at::Tensor x = ctx->get_saved_variables()[0];
        at::Tensor g = grads[0];
        if (!g.defined()) return {at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        g = g.contiguous();
        at::Tensor dx = at::empty_like(x);
        check(reinterpret_cast<RowBwdFn>(ctx->saved_data["bwd"].toInt())(static_cast<const float *>(x.data_ptr()), static_cast<const float *>(g.data_ptr()),
                                                                        static_cast<float *>(dx.data_ptr()), x.numel() / ctx->saved_data["width"].toInt(),
                                                                        current_stream(x)),
              "row head backward");
        return {dx, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

void bind(const py::dict &addresses, int64_t small_batch) {
    auto at_ = [&](const char *name) -> int64_t { return addresses[name].cast<int64_t>(); };
    g_entry.frob_f32 = reinterpret_cast<FrobFn>(at_("so3_frob_fwd_bwd_v2_f32"));
    g_entry.frob_bf16 = reinterpret_cast<FrobFn>(at_("so3_frob_fwd_bwd_v2_bf16"));
    g_entry.scale_f32 = reinterpret_cast<ScaleFn>(at_("so3_scale_f32"));
    g_entry.scale_bf16 = reinterpret_cast<ScaleFn>(at_("so3_scale_bf16"));
    g_entry.fwd_f32 = reinterpret_cast<FwdFn>(at_("so3_project_fwd_f32"));
    g_entry.fwd_bf16 = reinterpret_cast<FwdFn>(at_("so3_project_fwd_bf16"));
    g_entry.bwd_f32 = reinterpret_cast<BwdFn>(at_("so3_project_bwd_f32"));
    g_entry.bwd_bf16 = reinterpret_cast<BwdFn>(at_("so3_project_bwd_bf16"));
    g_entry.loss_f32 = reinterpret_cast<LossFn>(at_("so3_frob_loss_v2_f32"));
    g_entry.last_error = reinterpret_cast<ErrFn>(at_("so3_last_error"));
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

// None unless x is the node's case (then the rotation, (B,3,3) float32, with its gradient path).
py::object symmetric_orthogonalization(const at::Tensor &x, int64_t stream) {
    const auto dt = x.scalar_type();
    if (g_entry.fwd_f32 == nullptr || !x.is_cuda() || (dt != at::kFloat && dt != at::kBFloat16) || !x.is_contiguous() || x.numel() < 9 ||
        x.numel() % 9 != 0 || !x.requires_grad() || !at::GradMode::is_enabled())
        return py::none();
    return py::cast(ProjectNode::apply(x, stream));
}

// None unless (a, b) is the node's case (then the 0-dim float32 mean).
py::object loss_frobenius(const at::Tensor &a, const at::Tensor &b, int64_t stream, int64_t workspace) {
    if (g_entry.loss_f32 == nullptr || !a.is_cuda() || a.scalar_type() != at::kFloat || b.scalar_type() != at::kFloat || !a.is_contiguous() ||
        !b.is_contiguous() || a.numel() < 9 || a.numel() % 9 != 0 || b.numel() != a.numel() || b.device() != a.device() ||
        !((a.requires_grad() || b.requires_grad()) && at::GradMode::is_enabled()) || (a.numel() / 9 > g_entry.small_batch && workspace == 0))
        return py::none();
    return py::cast(FrobLossNode::apply(a, b, stream, workspace));
}

// None unless x is the node's case: float32, contiguous, last dimension `width`, at least one row, requiring grad.
py::object row_head(const at::Tensor &x, int64_t width, int64_t fwd, int64_t bwd, int64_t stream) {
    if (fwd == 0 || bwd == 0 || !x.is_cuda() || x.scalar_type() != at::kFloat || !x.is_contiguous() || x.dim() < 1 || x.size(-1) != width ||
        x.numel() < width || !x.requires_grad() || !at::GradMode::is_enabled())
        return py::none();
    return py::cast(RowHeadNode::apply(x, width, fwd, bwd, stream));
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("bind", &bind, "addresses of the C-ABI entry points by name (a dict) and kSmallBatch");
    m.def("frobenius_head", &frobenius_head);
    m.def("symmetric_orthogonalization", &symmetric_orthogonalization);
    m.def("loss_frobenius", &loss_frobenius);
    m.def("row_head", &row_head);
}
