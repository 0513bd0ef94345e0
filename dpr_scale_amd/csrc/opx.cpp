// _opx -- the host side of the autograd operator's single-rank step in C++ (dpr_scale_amd/hotpath.py: InBatchContrastive).
//
// What DenseRetrieverTask.training_step issues per step at world size 1 (dpr_task.py:197-212 and its backward) is ONE library call in
// forward (dprhot_train_step_f32) and ONE in backward (dprhot_rescale_grads).  Around them the Python operator spent ~200 us per step
// on the host at BASELINE cfg2 -- a dozen torch.empty calls, two dozen ctypes conversions, slices and reshapes -- against 9 us of
// kernels.  Here the same sequence is two C++ functions: buffers from the caching allocator (at::empty), the stream from
// c10::hip::getCurrentHIPStream(), the C ABI entry points resolved with dlsym from the libdprhot.so that dpr_scale_amd/_lib.py loaded.
// No device work of its own, no torch types in the ABI it calls; the Python operator keeps every other case (world size > 1, other
// dtypes, ragged context counts, debug modes) and everything about autograd.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <dlfcn.h>

#include <map>
#include <mutex>
#include <tuple>

namespace {

using ws_fn = int (*)(int, int, int, size_t*);
using shape_fn = int (*)(int, int, int, int*);
using err_fn = const char* (*)();
using step_fn = int (*)(const float*, const float*, void*, void*, int, int, int, const int64_t*, int64_t, const uint8_t*, float, float, float,
                        const float*, float*, float*, float*, void*, float*, float*, void*, int, void*, size_t, void*);
using rescale_fn = int (*)(float*, size_t, const float*, int, void*, size_t, int, const float*, const float*, float*, void*);

struct Api {
  ws_fn workspace_bytes = nullptr;
  shape_fn step_wants_g = nullptr, train_dq_slabs = nullptr;
  err_fn last_error = nullptr;
  step_fn train_step_f32 = nullptr;
  rescale_fn rescale_grads = nullptr;
} g_api;

struct Plan { size_t ws_bytes; int wants_g, nslabs; };
std::mutex g_mu;
std::map<std::tuple<int, int, int, int64_t>, Plan> g_plans;  // (B, Nc, d, options epoch)
std::map<int, at::Tensor> g_ws;                                // device index -> workspace

void check(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed with code ", rc, ": ", g_api.last_error ? g_api.last_error() : "");
}

bool init(const std::string& lib_path) {
  void* h = dlopen(lib_path.c_str(), RTLD_NOW | RTLD_GLOBAL);
  TORCH_CHECK(h != nullptr, "dlopen(", lib_path, "): ", dlerror());
  g_api.workspace_bytes = (ws_fn)dlsym(h, "dprhot_workspace_bytes");
  g_api.step_wants_g = (shape_fn)dlsym(h, "dprhot_step_wants_g");
  g_api.train_dq_slabs = (shape_fn)dlsym(h, "dprhot_train_dq_slabs");
  g_api.last_error = (err_fn)dlsym(h, "dprhot_last_error");
  g_api.train_step_f32 = (step_fn)dlsym(h, "dprhot_train_step_f32");
  g_api.rescale_grads = (rescale_fn)dlsym(h, "dprhot_rescale_grads");
  TORCH_CHECK(g_api.workspace_bytes && g_api.step_wants_g && g_api.train_dq_slabs && g_api.last_error && g_api.train_step_f32 && g_api.rescale_grads,
              "libdprhot.so lacks an entry point of include/dprhot.h");
  return true;
}

Plan plan_of(int B, int Nc, int d, int64_t epoch) {
  std::lock_guard<std::mutex> lk(g_mu);
  const auto key = std::make_tuple(B, Nc, d, epoch);
  auto it = g_plans.find(key);
  if (it != g_plans.end()) return it->second;
  Plan p{};
  check(g_api.workspace_bytes(B, Nc, d, &p.ws_bytes), "dprhot_workspace_bytes");
  check(g_api.step_wants_g(B, Nc, d, &p.wants_g), "dprhot_step_wants_g");
  check(g_api.train_dq_slabs(B, Nc, d, &p.nslabs), "dprhot_train_dq_slabs");
  g_plans[key] = p;
  return p;
}

void* workspace(const at::Device& dev, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  at::Tensor& w = g_ws[dev.index()];
  if (!w.defined() || (size_t)w.numel() < bytes)
    w = at::empty({(int64_t)std::max<size_t>(bytes, 1 << 20)}, at::TensorOptions().dtype(at::kByte).device(dev));
  return w.data_ptr();
}

// forward of the operator at world size 1: q [B,d], c [Nc,d] fp32 contiguous, pos_idx [B] int64, mask [Nc] uint8/bool (1 byte each),
// d_scale: DEVICE scalar the gradients are scaled by.  Returns {loss_out [2], row_lse [B], dQ, dC, Qb, Cb, G or empty, slabs or empty}.
std::vector<at::Tensor> train_step(const at::Tensor& q, const at::Tensor& c, const at::Tensor& pos_idx, const at::Tensor& mask, double inv_T,
                                   double grad_scale, double loss_scale, const at::Tensor& d_scale, int64_t options_epoch) {
  TORCH_CHECK(q.is_cuda() && c.is_cuda() && pos_idx.is_cuda() && mask.is_cuda() && d_scale.is_cuda(), "HIP device tensors required (no CPU path)");
  TORCH_CHECK(q.scalar_type() == at::kFloat && c.scalar_type() == at::kFloat && q.is_contiguous() && c.is_contiguous(), "q, c: contiguous fp32");
  TORCH_CHECK(pos_idx.scalar_type() == at::kLong && pos_idx.is_contiguous() && mask.element_size() == 1 && mask.is_contiguous(), "pos_idx int64, mask 1 byte per column");
  const int B = (int)q.size(0), d = (int)q.size(1), Nc = (int)c.size(0);
  TORCH_CHECK(c.size(1) == d && pos_idx.numel() == B && mask.numel() == Nc && Nc % 8 == 0 && d % 8 == 0, "shape");
  const Plan p = plan_of(B, Nc, d, options_epoch);
  const auto dev = q.device();
  const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
  const auto b16 = at::TensorOptions().dtype(at::kBFloat16).device(dev);
  at::Tensor scal = at::empty({2 * (int64_t)B + 2}, f32);  // [row_loss | row_lse | loss_out]
  at::Tensor Qb = at::empty({B, d}, b16), Cb = at::empty({Nc, d}, b16);
  at::Tensor G = p.wants_g ? at::empty({B, Nc}, b16) : at::Tensor();
  at::Tensor dQ = at::empty({B, d}, f32), dC = at::empty({Nc, d}, f32);
  at::Tensor part = p.nslabs > 0 ? at::empty({p.nslabs, B, d}, f32) : at::Tensor();
  void* ws = workspace(dev, p.ws_bytes);
  float* sc = scal.data_ptr<float>();
  void* stream = (void*)c10::hip::getCurrentHIPStream(dev.index()).stream();
  check(g_api.train_step_f32(q.data_ptr<float>(), c.data_ptr<float>(), Qb.data_ptr(), Cb.data_ptr(), B, Nc, d, pos_idx.data_ptr<int64_t>(), 0,
                             (const uint8_t*)mask.data_ptr(), (float)inv_T, (float)grad_scale, (float)loss_scale, d_scale.data_ptr<float>(), sc, sc + B,
                             sc + 2 * B, G.defined() ? G.data_ptr() : nullptr, dQ.data_ptr<float>(), part.defined() ? part.data_ptr<float>() : nullptr,
                             dC.data_ptr(), /*dc_kind fp32*/ 2, ws, p.ws_bytes, stream),
        "dprhot_train_step_f32");
  return {scal.narrow(0, 2 * B, 2), scal.narrow(0, B, B), dQ, dC, Qb, Cb, G, part};
}

// backward: the gradients were computed for grad_output = *used, autograd delivers *go.  Returns {out2 [2], next expected scale [1]}.
std::vector<at::Tensor> rescale(const at::Tensor& dQ, const c10::optional<at::Tensor>& part_opt, const at::Tensor& dC, const at::Tensor& go,
                                const at::Tensor& used, bool need_dq, bool need_dc) {
  const at::Tensor part = part_opt.has_value() ? *part_opt : at::Tensor();
  TORCH_CHECK(go.is_cuda() && go.scalar_type() == at::kFloat && go.numel() == 1 && used.numel() == 1, "go / used: one fp32 value on the device");
  at::Tensor out2 = at::empty({2}, go.options());
  const bool slabs = part.defined() && part.numel() > 0;
  void* stream = (void*)c10::hip::getCurrentHIPStream(go.device().index()).stream();
  check(g_api.rescale_grads(need_dq ? dQ.data_ptr<float>() : nullptr, need_dq ? (size_t)dQ.numel() : 0, (need_dq && slabs) ? part.data_ptr<float>() : nullptr,
                            (need_dq && slabs) ? (int)part.size(0) : 0, need_dc ? dC.data_ptr() : nullptr, need_dc ? (size_t)dC.numel() : 0, 2,
                            go.data_ptr<float>(), used.data_ptr<float>(), out2.data_ptr<float>(), stream),
        "dprhot_rescale_grads");
  return {out2, out2.narrow(0, 1, 1)};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "host side of dpr_scale_amd's single-rank autograd step (C ABI calls + allocations in C++)";
  m.def("init", &init);
  m.def("train_step", &train_step);
  m.def("rescale", &rescale);
}
