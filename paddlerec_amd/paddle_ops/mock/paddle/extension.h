// Stand-in for PaddlePaddle's custom-operator header `paddle/extension.h` (SURVEY.md:342, §8(b)).
//
// PaddlePaddle is not installable in this environment, so the custom-op shim (../../rec_paddle_ops.cc) is compiled
// against this file instead: the SUBSET of Paddle's public custom-op API the shim uses, with the same spellings —
//   paddle::Tensor {shape, numel, dtype, place, data<T>, stream}, paddle::empty / full, paddle::DataType, paddle::Grad /
//   Vec, PD_BUILD_OP / PD_BUILD_GRAD_OP (.Inputs .Outputs .Attrs .SetKernelFn .SetInferShapeFn .SetInferDtypeFn),
//   PD_KERNEL / PD_INFER_SHAPE / PD_INFER_DTYPE, PD_CHECK / PD_THROW —
// so that the same .cc builds unchanged with the real header (paddle.utils.cpp_extension.load).  Unlike a compile-only
// mock this one EXECUTES: every PD_BUILD_OP lands in a registry that the `pd_mock_*` C entry points at the bottom expose,
// and `paddle::empty` asks the host (a callback the loader installs) for device memory.  The compat namespace's
// paddle.utils.cpp_extension.load() drives registered kernels through those entry points, so the reference's trainer
// reaches the HIP kernels through the very functions a real Paddle build would call.
// Nothing of Paddle's implementation is reproduced here; only the call surface.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#define PD_MOCK_EXTENSION_H 1

extern "C" {
typedef struct pd_mock_tensor {
  void* data;
  int64_t shape[8];
  int32_t ndim;
  int32_t dtype;   /* paddle::DataType */
  int32_t device;  /* GPU ordinal, -1 = host memory */
  int32_t reserved;
  void* stream;    /* the framework's current stream for this tensor's device */
  int64_t handle;  /* host-side owner of memory that paddle::empty obtained; 0 for tensors the caller passed in */
} pd_mock_tensor;
typedef struct pd_mock_attr {
  int32_t kind; /* 0 int64, 1 double, 2 string, 3 bool, 4 list of float (s = const float*, i = count) */
  int32_t reserved;
  int64_t i;
  double f;
  const char* s;
} pd_mock_attr;
/* host allocator: returns a handle (> 0) and the data pointer; fill == NULL leaves the memory uninitialised */
typedef int64_t (*pd_mock_alloc_fn)(const int64_t* shape, int32_t ndim, int32_t dtype, int32_t device, const double* fill,
                                    void** data);
}

namespace paddle {

enum class DataType : int32_t {
  UNDEFINED = 0, BOOL, UINT8, INT8, INT16, INT32, INT64, FLOAT16, BFLOAT16, FLOAT32, FLOAT64
};

class Place {
 public:
  Place() = default;
  explicit Place(int device) : device_(device) {}
  bool is_gpu() const { return device_ >= 0; }
  int GetDeviceId() const { return device_ < 0 ? 0 : device_; }
  int raw() const { return device_; }
  bool operator==(const Place& o) const { return device_ == o.device_; }

 private:
  int device_ = -1;
};
struct CPUPlace : Place { CPUPlace() : Place(-1) {} };
struct GPUPlace : Place { explicit GPUPlace(int id = 0) : Place(id) {} };

namespace mock {
struct Runtime {
  pd_mock_alloc_fn alloc = nullptr;
  void* stream = nullptr;          // ambient stream of the running kernel (outputs inherit it)
  std::string last_error;
};
inline Runtime& runtime() {
  static thread_local Runtime rt;
  return rt;
}
inline pd_mock_alloc_fn& allocator() {
  static pd_mock_alloc_fn fn = nullptr;
  return fn;
}
template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr DataType v = DataType::FLOAT32; };
template <> struct dtype_of<double> { static constexpr DataType v = DataType::FLOAT64; };
template <> struct dtype_of<int64_t> { static constexpr DataType v = DataType::INT64; };
template <> struct dtype_of<int32_t> { static constexpr DataType v = DataType::INT32; };
template <> struct dtype_of<uint8_t> { static constexpr DataType v = DataType::UINT8; };
template <> struct dtype_of<int8_t> { static constexpr DataType v = DataType::INT8; };
template <> struct dtype_of<bool> { static constexpr DataType v = DataType::BOOL; };
}  // namespace mock

#define PD_THROW(...)                                          \
  do {                                                         \
    std::ostringstream pd_os__;                                \
    ::paddle::mock_stream_all(pd_os__, __VA_ARGS__);           \
    throw std::runtime_error(pd_os__.str());                   \
  } while (0)
#define PD_CHECK(cond, ...)                                    \
  do {                                                         \
    if (!(cond)) {                                             \
      std::ostringstream pd_os__;                              \
      pd_os__ << "PD_CHECK(" #cond ") failed: ";              \
      ::paddle::mock_stream_all(pd_os__, ##__VA_ARGS__);       \
      throw std::runtime_error(pd_os__.str());                 \
    }                                                          \
  } while (0)
inline void mock_stream_all(std::ostringstream&) {}
template <typename T, typename... R>
inline void mock_stream_all(std::ostringstream& os, const T& t, const R&... r) {
  os << t;
  mock_stream_all(os, r...);
}

class Tensor {
 public:
  Tensor() = default;
  explicit Tensor(const pd_mock_tensor& t) : t_(std::make_shared<pd_mock_tensor>(t)) {}
  bool defined() const { return static_cast<bool>(t_); }
  bool initialized() const { return defined() && (t_->data != nullptr || numel() == 0); }
  std::vector<int64_t> shape() const { return std::vector<int64_t>(t_->shape, t_->shape + t_->ndim); }
  int64_t numel() const {
    int64_t n = 1;
    for (int i = 0; i < t_->ndim; ++i) n *= t_->shape[i];
    return n;
  }
  int64_t size() const { return numel(); }
  DataType dtype() const { return static_cast<DataType>(t_->dtype); }
  DataType type() const { return dtype(); }
  Place place() const { return Place(t_->device); }
  bool is_gpu() const { return t_->device >= 0; }
  bool is_cpu() const { return t_->device < 0; }
  void* stream() const { return t_->stream; }   // gpuStream_t in Paddle
  template <typename T> const T* data() const {
    PD_CHECK(mock::dtype_of<T>::v == dtype(), "Tensor::data<T>(): dtype mismatch (tensor holds ", t_->dtype, ")");
    return static_cast<const T*>(t_->data);
  }
  template <typename T> T* data() {
    PD_CHECK(mock::dtype_of<T>::v == dtype(), "Tensor::data<T>(): dtype mismatch (tensor holds ", t_->dtype, ")");
    return static_cast<T*>(t_->data);
  }
  const void* data() const { return t_->data; }
  const pd_mock_tensor& raw() const { return *t_; }

 private:
  std::shared_ptr<pd_mock_tensor> t_;
};

inline Tensor mock_new_tensor(const std::vector<int64_t>& shape, DataType dtype, const Place& place, const double* fill) {
  PD_CHECK(mock::allocator() != nullptr, "no host allocator installed (pd_mock_set_allocator)");
  PD_CHECK(shape.size() <= 8, "rank > 8");
  pd_mock_tensor t;
  std::memset(&t, 0, sizeof(t));
  t.ndim = static_cast<int32_t>(shape.size());
  for (size_t i = 0; i < shape.size(); ++i) t.shape[i] = shape[i];
  t.dtype = static_cast<int32_t>(dtype);
  t.device = place.raw();
  t.stream = mock::runtime().stream;
  t.handle = mock::allocator()(t.shape, t.ndim, t.dtype, t.device, fill, &t.data);
  PD_CHECK(t.handle > 0, "host allocator failed");
  return Tensor(t);
}
inline Tensor empty(const std::vector<int64_t>& shape, DataType dtype = DataType::FLOAT32, const Place& place = CPUPlace()) {
  return mock_new_tensor(shape, dtype, place, nullptr);
}
inline Tensor full(const std::vector<int64_t>& shape, double value, DataType dtype = DataType::FLOAT32,
                   const Place& place = CPUPlace()) {
  return mock_new_tensor(shape, dtype, place, &value);
}
inline Tensor zeros(const std::vector<int64_t>& shape, DataType dtype = DataType::FLOAT32, const Place& place = CPUPlace()) {
  return full(shape, 0.0, dtype, place);
}

inline std::string Grad(const std::string& name) { return name + "@GRAD"; }
inline std::string Vec(const std::string& name) { return name + "@VECTOR"; }

// ---------------------------------------------------------------------------------------------------------------------
// registry
namespace mock {
struct CallArgs {
  const std::vector<std::vector<Tensor>>* inputs;   // one entry per declared input (a Vec input holds several)
  const pd_mock_attr* attrs;
  int n_attrs;
};
using KernelFn = std::vector<Tensor> (*)(const CallArgs&);
using ShapeVec = std::vector<std::vector<int64_t>>;
struct InferArgs {
  const std::vector<ShapeVec>* shapes;              // per declared input
  const std::vector<std::vector<DataType>>* dtypes;
  const pd_mock_attr* attrs;
  int n_attrs;
};
using InferShapeFn = ShapeVec (*)(const InferArgs&);
using InferDtypeFn = std::vector<DataType> (*)(const InferArgs&);

struct OpDef {
  std::string name;
  bool is_grad = false;
  std::vector<std::string> inputs, outputs, attrs;
  KernelFn kernel = nullptr;
  InferShapeFn infer_shape = nullptr;
  InferDtypeFn infer_dtype = nullptr;
  std::vector<std::string> notes;   // free-form "key=value" metadata a binder attaches (REC_OP_NOTE)
};
inline std::vector<OpDef>& registry() {
  static std::vector<OpDef> r;
  return r;
}

// ---- argument unpacking: Tensor / vector<Tensor> arguments consume declared inputs in order, everything else attrs
template <typename T> struct ArgGet;
template <> struct ArgGet<Tensor> {
  static const Tensor& get(const CallArgs& a, size_t& ti, size_t&) {
    PD_CHECK(ti < a.inputs->size() && (*a.inputs)[ti].size() == 1, "kernel wants a Tensor for input #", ti);
    return (*a.inputs)[ti++][0];
  }
};
template <> struct ArgGet<std::vector<Tensor>> {
  static const std::vector<Tensor>& get(const CallArgs& a, size_t& ti, size_t&) {
    PD_CHECK(ti < a.inputs->size(), "kernel wants a tensor list for input #", ti);
    return (*a.inputs)[ti++];
  }
};
inline const pd_mock_attr& next_attr(const pd_mock_attr* attrs, int n, size_t& ai) {
  PD_CHECK(static_cast<int>(ai) < n, "kernel wants attribute #", ai, " but only ", n, " were passed");
  return attrs[ai++];
}
#define PD_MOCK_ATTR_GET(T, EXPR)                                                          \
  template <> struct ArgGet<T> {                                                           \
    static T get(const CallArgs& a, size_t&, size_t& ai) {                                 \
      const pd_mock_attr& v = next_attr(a.attrs, a.n_attrs, ai);                           \
      (void)v;                                                                             \
      return EXPR;                                                                         \
    }                                                                                      \
    static T from(const pd_mock_attr& v) { return EXPR; }                                  \
  };
PD_MOCK_ATTR_GET(int64_t, (v.kind == 1 ? static_cast<int64_t>(v.f) : v.i))
PD_MOCK_ATTR_GET(int, static_cast<int>(v.kind == 1 ? static_cast<int64_t>(v.f) : v.i))
PD_MOCK_ATTR_GET(bool, (v.i != 0))
PD_MOCK_ATTR_GET(float, static_cast<float>(v.kind == 1 ? v.f : static_cast<double>(v.i)))
PD_MOCK_ATTR_GET(double, (v.kind == 1 ? v.f : static_cast<double>(v.i)))
PD_MOCK_ATTR_GET(std::string, std::string(v.s ? v.s : ""))
PD_MOCK_ATTR_GET(std::vector<float>, (v.kind == 4 && v.s ? std::vector<float>(reinterpret_cast<const float*>(v.s),
                                                                              reinterpret_cast<const float*>(v.s) + v.i)
                                                          : std::vector<float>()))
#undef PD_MOCK_ATTR_GET

template <typename F, F fn> struct KernelImpl;
template <typename... Args, std::vector<Tensor> (*fn)(Args...)>
struct KernelImpl<std::vector<Tensor> (*)(Args...), fn> {
  static std::vector<Tensor> Run(const CallArgs& a) {
    size_t ti = 0, ai = 0;
    return Call(a, ti, ai, std::index_sequence_for<Args...>{});
  }
  template <size_t... I> static std::vector<Tensor> Call(const CallArgs& a, size_t& ti, size_t& ai, std::index_sequence<I...>) {
    // braced initialisation: the arguments are unpacked left to right
    std::tuple<Holder<std::decay_t<Args>>...> held{Holder<std::decay_t<Args>>(a, ti, ai)...};
    PD_CHECK(ti == a.inputs->size(), "kernel consumed ", ti, " inputs, op declares ", a.inputs->size());
    return fn(std::get<I>(held).ref()...);
  }
  template <typename T, typename = void> struct Holder {     // attributes: by value
    T v;
    Holder(const CallArgs& a, size_t& ti, size_t& ai) : v(ArgGet<T>::get(a, ti, ai)) {}
    const T& ref() const { return v; }
  };
  template <typename T>
  struct Holder<T, std::enable_if_t<std::is_same<T, Tensor>::value || std::is_same<T, std::vector<Tensor>>::value>> {
    const T* p;                                                // tensors: by reference into the call's input table
    Holder(const CallArgs& a, size_t& ti, size_t& ai) : p(&ArgGet<T>::get(a, ti, ai)) {}
    const T& ref() const { return *p; }
  };
};

// ---- InferShape: vector<int64_t> / vector<vector<int64_t>> consume inputs, everything else attrs
template <typename T> struct ShapeGet {
  static T get(const InferArgs& a, size_t&, size_t& ai) { return ArgGet<T>::from(next_attr(a.attrs, a.n_attrs, ai)); }
};
template <> struct ShapeGet<std::vector<int64_t>> {
  static std::vector<int64_t> get(const InferArgs& a, size_t& ti, size_t&) {
    PD_CHECK(ti < a.shapes->size() && (*a.shapes)[ti].size() == 1, "infer-shape wants one shape for input #", ti);
    return (*a.shapes)[ti++][0];
  }
};
template <> struct ShapeGet<ShapeVec> {
  static ShapeVec get(const InferArgs& a, size_t& ti, size_t&) { return (*a.shapes)[ti++]; }
};
template <typename F, F fn> struct InferShapeImpl;
template <typename... Args, ShapeVec (*fn)(Args...)> struct InferShapeImpl<ShapeVec (*)(Args...), fn> {
  static ShapeVec Run(const InferArgs& a) {
    size_t ti = 0, ai = 0;
    std::tuple<std::decay_t<Args>...> held{ShapeGet<std::decay_t<Args>>::get(a, ti, ai)...};
    return std::apply(fn, held);
  }
};
template <typename T> struct DtypeGet;
template <> struct DtypeGet<DataType> {
  static DataType get(const InferArgs& a, size_t& ti) { return (*a.dtypes)[ti++][0]; }
};
template <> struct DtypeGet<std::vector<DataType>> {
  static std::vector<DataType> get(const InferArgs& a, size_t& ti) { return (*a.dtypes)[ti++]; }
};
template <typename F, F fn> struct InferDtypeImpl;
template <typename... Args, std::vector<DataType> (*fn)(Args...)>
struct InferDtypeImpl<std::vector<DataType> (*)(Args...), fn> {
  static std::vector<DataType> Run(const InferArgs& a) {
    size_t ti = 0;
    std::tuple<std::decay_t<Args>...> held{DtypeGet<std::decay_t<Args>>::get(a, ti)...};
    return std::apply(fn, held);
  }
};

class OpBuilder {
 public:
  OpBuilder(const char* name, bool is_grad) {
    registry().emplace_back();
    idx_ = registry().size() - 1;
    registry()[idx_].name = name;
    registry()[idx_].is_grad = is_grad;
  }
  OpBuilder& Inputs(std::vector<std::string> v) { registry()[idx_].inputs = std::move(v); return *this; }
  OpBuilder& Outputs(std::vector<std::string> v) { registry()[idx_].outputs = std::move(v); return *this; }
  OpBuilder& Attrs(std::vector<std::string> v) { registry()[idx_].attrs = std::move(v); return *this; }
  OpBuilder& SetKernelFn(KernelFn f) { registry()[idx_].kernel = f; return *this; }
  OpBuilder& SetInferShapeFn(InferShapeFn f) { registry()[idx_].infer_shape = f; return *this; }
  OpBuilder& SetInferDtypeFn(InferDtypeFn f) { registry()[idx_].infer_dtype = f; return *this; }
  OpBuilder& Note(std::string kv) { registry()[idx_].notes.push_back(std::move(kv)); return *this; }

 private:
  size_t idx_;
};
}  // namespace mock
}  // namespace paddle

#define PD_KERNEL(...) (&::paddle::mock::KernelImpl<decltype(&__VA_ARGS__), &__VA_ARGS__>::Run)
#define PD_INFER_SHAPE(...) (&::paddle::mock::InferShapeImpl<decltype(&__VA_ARGS__), &__VA_ARGS__>::Run)
#define PD_INFER_DTYPE(...) (&::paddle::mock::InferDtypeImpl<decltype(&__VA_ARGS__), &__VA_ARGS__>::Run)
#define PD_MOCK_CAT_(a, b) a##b
#define PD_MOCK_CAT(a, b) PD_MOCK_CAT_(a, b)
#define PD_BUILD_OP(op_name) \
  static ::paddle::mock::OpBuilder PD_MOCK_CAT(pd_mock_op_##op_name##_, __LINE__) = ::paddle::mock::OpBuilder(#op_name, false)
#define PD_BUILD_GRAD_OP(op_name) \
  static ::paddle::mock::OpBuilder PD_MOCK_CAT(pd_mock_grad_##op_name##_, __LINE__) = ::paddle::mock::OpBuilder(#op_name, true)

// ---------------------------------------------------------------------------------------------------------------------
// C entry points of the built shim (what the loader binds with ctypes).  Defined here so that a one-file shim exports them.
#define PD_MOCK_API extern "C" __attribute__((visibility("default")))

PD_MOCK_API void pd_mock_set_allocator(pd_mock_alloc_fn fn) { ::paddle::mock::allocator() = fn; }
PD_MOCK_API const char* pd_mock_last_error(void) { return ::paddle::mock::runtime().last_error.c_str(); }
PD_MOCK_API int32_t pd_mock_op_count(void) { return static_cast<int32_t>(::paddle::mock::registry().size()); }

// describe(i): "name|fwd or grad|in1,in2|out1,out2|attr: type,...|has_kernel has_shape has_dtype|note;note"
PD_MOCK_API const char* pd_mock_op_describe(int32_t i) {
  static thread_local std::string s;
  auto& r = ::paddle::mock::registry();
  if (i < 0 || i >= static_cast<int32_t>(r.size())) return nullptr;
  auto join = [](const std::vector<std::string>& v, const char* sep) {
    std::string o;
    for (size_t k = 0; k < v.size(); ++k) o += (k ? sep : "") + v[k];
    return o;
  };
  const auto& d = r[i];
  s = d.name + "|" + (d.is_grad ? "grad" : "fwd") + "|" + join(d.inputs, ",") + "|" + join(d.outputs, ",") + "|" +
      join(d.attrs, ",") + "|" + (d.kernel ? "1" : "0") + (d.infer_shape ? "1" : "0") + (d.infer_dtype ? "1" : "0") + "|" +
      join(d.notes, ";");
  return s.c_str();
}

namespace paddle {
namespace mock {
inline int guarded(const std::function<void()>& body) {
  try {
    body();
    return 0;
  } catch (const std::exception& e) {
    runtime().last_error = e.what();
  } catch (...) {
    runtime().last_error = "unknown C++ exception";
  }
  return -1;
}
inline std::vector<std::vector<Tensor>> group_inputs(const OpDef& d, const pd_mock_tensor* in, const int32_t* counts,
                                                     int32_t n_decl) {
  PD_CHECK(n_decl == static_cast<int32_t>(d.inputs.size()), d.name, ": ", n_decl, " inputs passed, ", d.inputs.size(),
           " declared");
  std::vector<std::vector<Tensor>> g(n_decl);
  size_t k = 0;
  for (int32_t i = 0; i < n_decl; ++i)
    for (int32_t j = 0; j < counts[i]; ++j) g[i].emplace_back(in[k++]);
  return g;
}
}  // namespace mock
}  // namespace paddle

// run(op): inputs flat + counts per declared input; outputs written to out[0..*n_out) (handles name host-owned memory)
PD_MOCK_API int32_t pd_mock_op_run(int32_t op, const pd_mock_tensor* in, const int32_t* counts, int32_t n_decl,
                                   const pd_mock_attr* attrs, int32_t n_attrs, void* stream, pd_mock_tensor* out,
                                   int32_t max_out, int32_t* n_out) {
  using namespace ::paddle::mock;
  return guarded([&] {
    auto& r = registry();
    PD_CHECK(op >= 0 && op < static_cast<int32_t>(r.size()), "no such op index ", op);
    const OpDef& d = r[op];
    PD_CHECK(d.kernel != nullptr, d.name, ": no kernel registered");
    auto g = group_inputs(d, in, counts, n_decl);
    runtime().stream = stream;
    CallArgs a{&g, attrs, n_attrs};
    std::vector<::paddle::Tensor> res = d.kernel(a);
    PD_CHECK(res.size() == d.outputs.size(), d.name, ": kernel returned ", res.size(), " tensors, op declares ",
             d.outputs.size(), " outputs");
    PD_CHECK(static_cast<int32_t>(res.size()) <= max_out, "output array too small");
    for (size_t i = 0; i < res.size(); ++i) out[i] = res[i].raw();
    *n_out = static_cast<int32_t>(res.size());
  });
}

// infer(op): shapes / dtypes of `in` (data ignored) -> shapes / dtypes of the outputs.  -2: the op registered neither.
PD_MOCK_API int32_t pd_mock_op_infer(int32_t op, const pd_mock_tensor* in, const int32_t* counts, int32_t n_decl,
                                     const pd_mock_attr* attrs, int32_t n_attrs, pd_mock_tensor* out, int32_t max_out,
                                     int32_t* n_out) {
  using namespace ::paddle::mock;
  auto& r = registry();
  if (op < 0 || op >= static_cast<int32_t>(r.size())) return -1;
  if (!r[op].infer_shape && !r[op].infer_dtype) return -2;
  return guarded([&] {
    const OpDef& d = r[op];
    std::vector<ShapeVec> shapes(n_decl);
    std::vector<std::vector<::paddle::DataType>> dtypes(n_decl);
    size_t k = 0;
    for (int32_t i = 0; i < n_decl; ++i)
      for (int32_t j = 0; j < counts[i]; ++j, ++k) {
        shapes[i].emplace_back(in[k].shape, in[k].shape + in[k].ndim);
        dtypes[i].push_back(static_cast<::paddle::DataType>(in[k].dtype));
      }
    InferArgs a{&shapes, &dtypes, attrs, n_attrs};
    ShapeVec os;
    std::vector<::paddle::DataType> od;
    if (d.infer_shape) os = d.infer_shape(a);
    if (d.infer_dtype) od = d.infer_dtype(a);
    size_t n = d.infer_shape ? os.size() : od.size();
    PD_CHECK(n == d.outputs.size(), d.name, ": infer functions describe ", n, " outputs, op declares ", d.outputs.size());
    PD_CHECK(static_cast<int32_t>(n) <= max_out, "output array too small");
    for (size_t i = 0; i < n; ++i) {
      std::memset(&out[i], 0, sizeof(out[i]));
      out[i].ndim = -1;
      if (d.infer_shape) {
        out[i].ndim = static_cast<int32_t>(os[i].size());
        for (size_t j = 0; j < os[i].size(); ++j) out[i].shape[j] = os[i][j];
      }
      out[i].dtype = d.infer_dtype && i < od.size() ? static_cast<int32_t>(od[i]) : 0;
    }
    *n_out = static_cast<int32_t>(n);
  });
}
