// _opx -- the host side of the autograd operator's step in C++ (dpr_scale_amd/hotpath.py: InBatchContrastive): the single-rank step as a
// whole C++ autograd node, the multi-rank packed step as two functions the Python node calls around its collectives.
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
using pstep_fn = int (*)(const float*, const void*, void*, int, int, int, int, int, const int64_t*, float, float, float, const float*, float*, float*,
                         float*, void*, float*, float*, void*, int, void*, size_t, void*);
using rows_fn = int (*)(int, int, int*);
using cast_fn = int (*)(const float*, void*, size_t, void*);
using epoch_fn = long long (*)();

struct Api {
  ws_fn workspace_bytes = nullptr;
  shape_fn step_wants_g = nullptr, train_dq_slabs = nullptr;
  err_fn last_error = nullptr;
  step_fn train_step_f32 = nullptr;
  rescale_fn rescale_grads = nullptr;
  pstep_fn train_step_packed_f32 = nullptr;
  rows_fn packed_rows = nullptr;
  cast_fn cast_bf16 = nullptr;
  epoch_fn options_epoch = nullptr;
} g_api;

struct Plan { size_t ws_bytes; int wants_g, nslabs; };
std::mutex g_mu;
std::map<std::tuple<int, int, int, int64_t>, Plan> g_plans;  // (B, Nc, d, the LIBRARY's options epoch: dprhot_options_epoch)
// one workspace per (device, stream): steps on different streams never share one, and the tensor is allocated on the stream that
// uses it (the caching allocator orders reuse on the allocating stream)
std::map<std::pair<int, void*>, at::Tensor> g_ws;
std::map<std::tuple<int, int, int, int64_t>, int> g_dc_bf16_ok;  // (B, Nc, d, epoch) -> this shape's plan has a bf16 dC epilogue (1) or not (0)

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
  g_api.train_step_packed_f32 = (pstep_fn)dlsym(h, "dprhot_train_step_packed_f32");
  g_api.packed_rows = (rows_fn)dlsym(h, "dprhot_packed_rows");
  g_api.cast_bf16 = (cast_fn)dlsym(h, "dprhot_cast_bf16");
  g_api.options_epoch = (epoch_fn)dlsym(h, "dprhot_options_epoch");
  TORCH_CHECK(g_api.workspace_bytes && g_api.step_wants_g && g_api.train_dq_slabs && g_api.last_error && g_api.train_step_f32 && g_api.rescale_grads &&
                  g_api.train_step_packed_f32 && g_api.packed_rows && g_api.cast_bf16 && g_api.options_epoch,
              "libdprhot.so lacks an entry point of include/dprhot.h");
  return true;
}

Plan plan_of(int B, int Nc, int d, int64_t /*caller's epoch: superseded by the library's own*/) {
  const int64_t epoch = (int64_t)g_api.options_epoch();  // options set through the raw C ABI invalidate the cache too
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

// The tensor is returned BY VALUE: the caller holds it across its launches, so a concurrent call that grows the map's entry cannot
// free memory this call's kernels are about to use.
at::Tensor workspace(const at::Device& dev, void* stream, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  at::Tensor& w = g_ws[std::make_pair((int)dev.index(), stream)];
  if (!w.defined() || (size_t)w.numel() < bytes)
    w = at::empty({(int64_t)std::max<size_t>(bytes, 1 << 20)}, at::TensorOptions().dtype(at::kByte).device(dev));
  return w;
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
  void* stream = (void*)c10::hip::getCurrentHIPStream(dev.index()).stream();
  const at::Tensor wst = workspace(dev, stream, p.ws_bytes);
  void* ws = wst.data_ptr();
  float* sc = scal.data_ptr<float>();
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

// ---- the MULTI-RANK step's host side (round 5): what InBatchContrastive.forward / .backward issue under DDP between the all-gather and
// the reduce-scatter -- allocations, dprhot_train_step_packed_f32 (with the fall-back to fp32 partials where the shape's plan has no
// bf16 dC epilogue), dprhot_rescale_grads, the cast to the wire format, the receive buffer -- as two C++ functions.  The collectives
// themselves stay with dpr_scale_amd.dist (torch.distributed or the C ABI communicator): the Python node calls these two and the
// collectives, nothing else.
// q [B,d] fp32, gathered [W * rows_c, d] bf16 (the all-gathered packed buffer), pos_idx [B] int64, d_scale: device scalar.
// wire_kind: 0 bf16 | 2 fp32 (the reduce-scatter's format).  Returns {loss_out [2], row_lse [B], dQ, dC_part, Qb, G or empty, slabs or empty}.
std::vector<at::Tensor> packed_train_step(const at::Tensor& q, const at::Tensor& gathered, const at::Tensor& pos_idx, int64_t W, int64_t rank, int64_t n_ctx,
                                          double inv_T, double grad_scale, double loss_scale, const at::Tensor& d_scale, int64_t wire_kind) {
  TORCH_CHECK(q.is_cuda() && gathered.is_cuda() && pos_idx.is_cuda() && d_scale.is_cuda(), "HIP device tensors required (no CPU path)");
  TORCH_CHECK(q.scalar_type() == at::kFloat && q.is_contiguous() && gathered.scalar_type() == at::kBFloat16 && gathered.is_contiguous(), "q fp32, gathered bf16, contiguous");
  TORCH_CHECK(pos_idx.scalar_type() == at::kLong && pos_idx.is_contiguous(), "pos_idx int64");
  const int B = (int)q.size(0), d = (int)q.size(1), Nc = (int)gathered.size(0);
  int rows_c = 0;
  check(g_api.packed_rows((int)n_ctx, d, &rows_c), "dprhot_packed_rows");
  TORCH_CHECK(gathered.size(1) == d && Nc == W * rows_c && pos_idx.numel() == B && (wire_kind == 0 || wire_kind == 2), "shape / wire kind");
  const Plan p = plan_of(B, Nc, d, 0);
  const int64_t epoch = (int64_t)g_api.options_epoch();
  const auto dev = q.device();
  const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
  const auto b16 = at::TensorOptions().dtype(at::kBFloat16).device(dev);
  int kind = (int)wire_kind;
  if (kind == 0) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_dc_bf16_ok.find(std::make_tuple(B, Nc, d, epoch));
    if (it != g_dc_bf16_ok.end() && it->second == 0) kind = 2;  // known: this plan writes fp32 partials only
  }
  at::Tensor scal = at::empty({2 * (int64_t)B + 2}, f32);  // [row_loss | row_lse | loss_out]
  at::Tensor Qb = at::empty({B, d}, b16);
  at::Tensor G = p.wants_g ? at::empty({B, Nc}, b16) : at::Tensor();
  at::Tensor dQ = at::empty({B, d}, f32);
  at::Tensor part = p.nslabs > 0 ? at::empty({p.nslabs, B, d}, f32) : at::Tensor();
  void* stream = (void*)c10::hip::getCurrentHIPStream(dev.index()).stream();
  const at::Tensor wst = workspace(dev, stream, p.ws_bytes);
  float* sc = scal.data_ptr<float>();
  at::Tensor dC;
  for (int attempt = 0; attempt < 2; ++attempt) {
    dC = at::empty({Nc, d}, kind == 0 ? b16 : f32);
    const int rc = g_api.train_step_packed_f32(q.data_ptr<float>(), gathered.data_ptr(), Qb.data_ptr(), B, (int)W, (int)rank, (int)n_ctx, d,
                                               pos_idx.data_ptr<int64_t>(), (float)inv_T, (float)grad_scale, (float)loss_scale, d_scale.data_ptr<float>(), sc,
                                               sc + B, sc + 2 * B, G.defined() ? G.data_ptr() : nullptr, dQ.data_ptr<float>(),
                                               part.defined() ? part.data_ptr<float>() : nullptr, dC.data_ptr(), kind, wst.data_ptr(), p.ws_bytes, stream);
    if (rc == 0) {
      if (wire_kind == 0) {
        std::lock_guard<std::mutex> lk(g_mu);
        g_dc_bf16_ok[std::make_tuple(B, Nc, d, epoch)] = kind == 0 ? 1 : 0;
      }
      break;
    }
    const std::string msg = g_api.last_error ? g_api.last_error() : "";
    if (attempt == 0 && kind == 0 && msg.find("dc_kind") != std::string::npos) {
      kind = 2;  // a plan without a bf16 dC epilogue: fp32 partials, rounded to the wire format in backward (packed_backward)
      continue;
    }
    check(rc, "dprhot_train_step_packed_f32");
  }
  return {scal.narrow(0, 2 * B, 2), scal.narrow(0, B, B), dQ, dC, Qb, G, part};
}

// backward: dprhot_rescale_grads on what packed_train_step left (dQ, its slabs, dC_part in its storage kind), then dC_part in the wire
// format (one cast launch where the step wrote fp32 for a bf16 wire) and the receive buffer of the reduce-scatter.
// Returns {next expected scale [1], dC_part in the wire format, mine [rows_c, d] in the wire format (uninitialised)}.
std::vector<at::Tensor> packed_backward(const at::Tensor& dQ, const c10::optional<at::Tensor>& part_opt, const at::Tensor& dC, const at::Tensor& go,
                                        const at::Tensor& used, int64_t wire_kind, int64_t rows_c, bool need_dq, bool need_dc) {
  const at::Tensor part = part_opt.has_value() ? *part_opt : at::Tensor();
  TORCH_CHECK(go.is_cuda() && go.scalar_type() == at::kFloat && go.numel() == 1 && used.numel() == 1, "go / used: one fp32 value on the device");
  const int kind = dC.scalar_type() == at::kBFloat16 ? 0 : 2;
  at::Tensor out2 = at::empty({2}, go.options());
  const bool slabs = part.defined() && part.numel() > 0;
  void* stream = (void*)c10::hip::getCurrentHIPStream(go.device().index()).stream();
  check(g_api.rescale_grads(need_dq ? dQ.data_ptr<float>() : nullptr, need_dq ? (size_t)dQ.numel() : 0, (need_dq && slabs) ? part.data_ptr<float>() : nullptr,
                            (need_dq && slabs) ? (int)part.size(0) : 0, need_dc ? dC.data_ptr() : nullptr, need_dc ? (size_t)dC.numel() : 0, kind,
                            go.data_ptr<float>(), used.data_ptr<float>(), out2.data_ptr<float>(), stream),
        "dprhot_rescale_grads");
  at::Tensor wire = dC, mine;
  if (need_dc) {
    const auto wopt = at::TensorOptions().dtype(wire_kind == 0 ? at::kBFloat16 : at::kFloat).device(dC.device());
    if (wire_kind == 0 && kind == 2) {
      wire = at::empty(dC.sizes(), wopt);
      check(g_api.cast_bf16(dC.data_ptr<float>(), wire.data_ptr(), (size_t)dC.numel(), stream), "dprhot_cast_bf16");
    }
    TORCH_CHECK(!(wire_kind == 2 && kind == 0), "bf16 partials cannot feed an fp32 wire");
    mine = at::empty({rows_c, dC.size(1)}, wopt);
  }
  return {out2.narrow(0, 1, 1), wire, mine};
}

// ---- the operator itself as a C++ autograd node (forward + backward without a Python frame) ----------------------------------
// InBatchContrastive (hotpath.py) keeps every other case; this node is what a single-GPU training step runs: the gradients are
// computed by the forward call for the grad_output the previous backward saw (a device scalar per (device, B, Nc, d)), backward
// compares and rescales in one launch (dprhot_rescale_grads) and hands the tensors to autograd without keeping a reference --
// AccumulateGrad takes a gradient nobody else holds as .grad itself instead of cloning it.
std::map<std::tuple<int, int, int, int>, at::Tensor> g_scale;  // (device, B, Nc, d) -> [1] fp32: expected grad_output
py::object* g_regen = nullptr;  // Python: second backward through a retained graph (hotpath._opx_regen); leaked on purpose (interpreter exit)

at::Tensor scale_get(const at::Device& dev, int B, int Nc, int d) {
  std::lock_guard<std::mutex> lk(g_mu);
  at::Tensor& t = g_scale[std::make_tuple((int)dev.index(), B, Nc, d)];
  if (!t.defined()) t = at::ones({1}, at::TensorOptions().dtype(at::kFloat).device(dev));
  return t;
}
void scale_put(const at::Device& dev, int B, int Nc, int d, const at::Tensor& t) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_scale[std::make_tuple((int)dev.index(), B, Nc, d)] = t;
}

struct InBatchFn : public torch::autograd::Function<InBatchFn> {
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& q, const at::Tensor& c, const at::Tensor& pos_idx,
                            const at::Tensor& mask, double inv_T, int64_t epoch) {
    const int B = (int)q.size(0), d = (int)q.size(1), Nc = (int)c.size(0);
    const at::Tensor used = scale_get(q.device(), B, Nc, d);
    std::vector<at::Tensor> out = train_step(q, c, pos_idx, mask, inv_T, inv_T / B, 1.0 / B, used, epoch);
    auto& sd = ctx->saved_data;
    sd["dQ"] = out[2];
    sd["dC"] = out[3];
    if (out[7].defined()) sd["part"] = out[7];
    sd["used"] = used;
    sd["Qb"] = out[4];  // (what a second backward through a retained graph recomputes from)
    sd["Cb"] = out[5];
    if (out[6].defined()) sd["G"] = out[6];
    sd["pos"] = pos_idx;
    sd["mask"] = mask;
    sd["inv_T"] = inv_T;
    return out[0].select(0, 0);
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list gos) {
    auto& sd = ctx->saved_data;
    at::Tensor go = gos[0];
    if (go.scalar_type() != at::kFloat || !go.is_contiguous()) go = go.detach().to(at::kFloat).contiguous();
    const bool need_dq = ctx->needs_input_grad(0), need_dc = ctx->needs_input_grad(1);
    at::Tensor dQ, dC;
    const at::Tensor Qb = sd["Qb"].toTensor(), Cb = sd["Cb"].toTensor();
    const int B = (int)Qb.size(0), d = (int)Qb.size(1), Nc = (int)Cb.size(0);
    if (sd.find("dQ") != sd.end()) {
      dQ = sd["dQ"].toTensor();
      dC = sd["dC"].toTensor();
      c10::optional<at::Tensor> part;
      if (sd.find("part") != sd.end()) part = sd["part"].toTensor();
      const at::Tensor used = sd["used"].toTensor();
      sd.erase("dQ");
      sd.erase("dC");
      sd.erase("part");
      sd.erase("used");
      std::vector<at::Tensor> r = rescale(dQ, part, dC, go, used, need_dq, need_dc);
      scale_put(go.device(), B, Nc, d, r[1]);
    } else {
      // a second backward through a retained graph: the first one gave its gradient tensors away; the backward GEMMs run again on
      // the operands the step left behind (exact, rare) -- in Python, where the general kernels' wrappers live
      TORCH_CHECK(g_regen != nullptr, "opx: no regeneration callback registered (hotpath sets it at import)");
      py::gil_scoped_acquire gil;
      const double inv_T = sd["inv_T"].toDouble();
      py::object G = sd.find("G") != sd.end() ? py::cast(sd["G"].toTensor()) : py::none();
      py::tuple res = (*g_regen)(Qb, Cb, G, sd["pos"].toTensor(), sd["mask"].toTensor(), inv_T, inv_T / B, go, need_dq, need_dc);
      if (!res[0].is_none()) dQ = res[0].cast<at::Tensor>();
      if (!res[1].is_none()) dC = res[1].cast<at::Tensor>();
      sd["G"] = res[2].cast<at::Tensor>();
    }
    return {need_dq ? dQ : at::Tensor(), need_dc ? dC : at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

// None when the step is not the plain single-rank fp32 one (the caller then takes InBatchContrastive).
c10::optional<at::Tensor> inbatch_loss(const at::Tensor& q, const at::Tensor& c, const at::Tensor& pos_idx, const at::Tensor& mask, double inv_T,
                                       int64_t options_epoch) {
  const bool ok = q.is_cuda() && c.is_cuda() && pos_idx.is_cuda() && mask.is_cuda() && q.dim() == 2 && c.dim() == 2 && q.scalar_type() == at::kFloat &&
                  c.scalar_type() == at::kFloat && q.is_contiguous() && c.is_contiguous() && c.size(1) == q.size(1) && c.size(0) % 8 == 0 &&
                  q.size(1) % 8 == 0 && mask.is_contiguous() && mask.element_size() == 1 && mask.numel() == c.size(0) &&
                  pos_idx.scalar_type() == at::kLong && pos_idx.is_contiguous() && pos_idx.numel() == q.size(0) &&
                  (q.requires_grad() || c.requires_grad()) && at::GradMode::is_enabled();
  if (!ok) return c10::nullopt;
  return InBatchFn::apply(q, c, pos_idx, mask, inv_T, options_epoch);
}

void set_regen(py::object fn) { g_regen = new py::object(std::move(fn)); }

// Measurement probe (bench.py, operator block): the same node shape -- two differentiable inputs, a scalar output, fresh gradient
// tensors from the caching allocator -- that launches NOTHING.  What remains is torch's autograd machinery for a C++ node.
struct FloorFn : public torch::autograd::Function<FloorFn> {
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& q, const at::Tensor& c) {
    ctx->saved_data["qs"] = q.sizes().vec();
    ctx->saved_data["cs"] = c.sizes().vec();
    return at::empty({}, q.options());
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list gos) {
    const auto opt = gos[0].options();
    return {at::empty(ctx->saved_data["qs"].toIntVector(), opt), at::empty(ctx->saved_data["cs"].toIntVector(), opt)};
  }
};
at::Tensor floor_loss(const at::Tensor& q, const at::Tensor& c) { return FloorFn::apply(q, c); }

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "host side of dpr_scale_amd's single-rank autograd step (C ABI calls + allocations in C++)";
  m.def("init", &init);
  m.def("train_step", &train_step);
  m.def("rescale", &rescale);
  m.def("packed_train_step", &packed_train_step);
  m.def("packed_backward", &packed_backward);
  m.def("inbatch_loss", &inbatch_loss);
  m.def("set_regen", &set_regen);
  m.def("floor_loss", &floor_loss);
}
