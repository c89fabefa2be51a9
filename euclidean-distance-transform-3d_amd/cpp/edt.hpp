// edt.hpp -- drop-in C++ header for the MI355X-native EDT.
//
// Same namespaces, function names, argument order and defaults as the reference header
// (reference: src/edt.hpp:32-803 `namespace pyedt`, :805-954 `namespace edt`;
// src/edt_voxel_graph.hpp:54-236), so existing callers -- including the reference's Cython
// binding, which does `cdef extern from "edt.hpp" namespace "pyedt"` (src/edt.pyx:62-87) and
// `cdef extern from "edt_voxel_graph.hpp"` (:89-113; the sibling header of this directory) --
// compile unchanged against the pair (tests/test_cython_dropin.py builds the unmodified edt.pyx
// against them).  Every template is a thin inline forwarder into the
// C ABI of include/edt_hip.h (link with -ledt_hip); the label type becomes a dtype code.
//
// Behaviour kept from the reference:
//   * `workspace/output == NULL` -> the result is allocated with new float[] (freed again if the call throws) and owned by
//     the caller (delete[]), otherwise the return value aliases the caller's buffer
//     (src/edt.hpp:424-426);
//   * `parallel` is accepted and ignored (the GPU grid replaces the thread pool);
//   * 1-D edt::edt ignores black_border exactly like the reference (src/edt.hpp:807-821).
// Difference: failures (no GPU, out of memory, unsupported dtype) throw std::runtime_error --
// the reference has no failure modes here, and silently computing on the CPU is not an option.
// Under the reference's Cython binding that would be std::terminate: src/edt.pyx:62-113 declares these
// functions without `except +` and calls them `nogil`.  Two ways out: add `except +` to
// the extern block (the exception becomes a Python RuntimeError), or -- for the UNMODIFIED binding --
// compile it with -DEDT_HIP_PYTHON_ERRORS: a failing call then sets a Python RuntimeError through the
// interpreter already in the process (symbols looked up at run time: no Python.h here), fills the result
// with NaN and returns; CPython turns "returned a result with an exception set" into a SystemError
// chained to it.  The interpreter survives and the caller sees why.
#ifndef EDT_AMD_EDT_HPP
#define EDT_AMD_EDT_HPP

#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "edt_hip.h"

namespace pyedt {

template <typename T>
constexpr int dtype_code() {
  static_assert(std::is_arithmetic<T>::value, "labels must be an arithmetic type");
  if (std::is_same<T, bool>::value) return EDT_BOOL;
  if (std::is_floating_point<T>::value) return sizeof(T) == 4 ? EDT_F32 : EDT_F64;
  // signed integers are reinterpreted as unsigned, like src/edt.pyx:670-705
  return sizeof(T) == 1 ? EDT_U8 : sizeof(T) == 2 ? EDT_U16 : sizeof(T) == 4 ? EDT_U32 : EDT_U64;
}

// The result buffer of a call: the caller's, or a fresh `new float[]` the caller will own -- released again if the call
// throws, and NOT zero-filled first (the reference's `new float[n]()` is: every entry point here writes every element).
struct Result {
  float* p;
  size_t n;
  std::unique_ptr<float[]> own;
  Result(float* given, size_t count) : p(given), n(count) {
    if (p == NULL) {
      own.reset(new float[count]);
      p = own.get();
    }
  }
  float* release() {
    own.release();
    return p;
  }
};

#ifdef EDT_HIP_PYTHON_ERRORS
}  // namespace pyedt
#include <dlfcn.h>
namespace pyedt {
// a Python RuntimeError through the interpreter of this process; false: there is none (a plain C++ host)
inline bool raise_in_python(const std::string& msg) {
  typedef int (*int_fn)();
  typedef void (*release_fn)(int);
  typedef void (*setstr_fn)(void*, const char*);
  int_fn is_init = (int_fn)dlsym(RTLD_DEFAULT, "Py_IsInitialized");
  int_fn ensure = (int_fn)dlsym(RTLD_DEFAULT, "PyGILState_Ensure");
  release_fn release = (release_fn)dlsym(RTLD_DEFAULT, "PyGILState_Release");
  setstr_fn setstr = (setstr_fn)dlsym(RTLD_DEFAULT, "PyErr_SetString");
  void** exc = (void**)dlsym(RTLD_DEFAULT, "PyExc_RuntimeError");
  if (!is_init || !ensure || !release || !setstr || !exc || !is_init()) return false;
  const int state = ensure();  // (the binding calls us `nogil`)
  setstr(*exc, msg.c_str());
  release(state);
  return true;
}
#endif

// out / count: the result buffer of the failing call (filled with NaN where the call returns instead of throwing)
inline void check(int rc, float* out = NULL, size_t count = 0) {
  if (rc == EDT_OK) return;
  const std::string msg = std::string("edt_hip: ") + edt_hip_last_error();
#ifdef EDT_HIP_PYTHON_ERRORS
  if (raise_in_python(msg)) {
    for (size_t i = 0; out != NULL && i < count; ++i) out[i] = std::numeric_limits<float>::quiet_NaN();
    return;
  }
#endif
  throw std::runtime_error(msg);
}

// src/edt.hpp:70-119
template <typename T>
void squared_edt_1d_multi_seg(T* segids, float* d, const int64_t n, const int64_t stride,
                              const float anisotropy, const bool black_border = false) {
  if (n == 0) return;
  check(edt_hip_squared_edt_1d_multi_seg(segids, dtype_code<T>(), d, n, stride, anisotropy,
                                         black_border));
}

// src/edt.hpp:411-484 (bool: :580-587)
template <typename T>
float* _edt3dsq(T* labels, const int64_t sx, const int64_t sy, const int64_t sz, const float wx,
                const float wy, const float wz, const bool black_border = false,
                const int parallel = 1, float* workspace = NULL) {
  Result out(workspace, (size_t)(sx * sy * sz));
  check(edt_hip_edt3dsq(labels, dtype_code<T>(), sx, sy, sz, wx, wy, wz, black_border, parallel,
                        out.p), out.p, out.n);
  return out.release();
}

// src/edt.hpp:591-604
template <typename T>
float* _edt3d(T* labels, const int64_t sx, const int64_t sy, const int64_t sz, const float wx,
              const float wy, const float wz, const bool black_border = false,
              const int parallel = 1, float* workspace = NULL) {
  Result out(workspace, (size_t)(sx * sy * sz));
  check(edt_hip_edt3d(labels, dtype_code<T>(), sx, sy, sz, wx, wy, wz, black_border, parallel,
                      out.p), out.p, out.n);
  return out.release();
}

// src/edt.hpp:632-678 (bool: :758-772)
template <typename T>
float* _edt2dsq(T* labels, const int64_t sx, const int64_t sy, const float wx, const float wy,
                const bool black_border = false, const int parallel = 1,
                float* workspace = NULL) {
  Result out(workspace, (size_t)(sx * sy));
  check(edt_hip_edt2dsq(labels, dtype_code<T>(), sx, sy, wx, wy, black_border, parallel, out.p), out.p, out.n);
  return out.release();
}

// src/edt.hpp:776-797
template <typename T>
float* _edt2d(T* labels, const int64_t sx, const int64_t sy, const float wx, const float wy,
              const bool black_border = false, const int parallel = 1, float* output = NULL) {
  Result out(output, (size_t)(sx * sy));
  check(edt_hip_edt2d(labels, dtype_code<T>(), sx, sy, wx, wy, black_border, parallel, out.p), out.p, out.n);
  return out.release();
}

// The binary route (src/edt.hpp:487-576, :607-629, :681-755): pass X splits runs at label changes, passes Y and Z
// scan every column as ONE envelope from its first non-zero value (background voxels are height-0 sites).  On 0/1
// (and bool) input that equals the multi-label transform; on multi-valued T it does not, and the C ABI reproduces
// the reference's values (edt_hip_binary_edtsq / EDT_FLAG_BINARY_YZ).
template <typename T>
float* _binary_edt3dsq(T* img, const int64_t sx, const int64_t sy, const int64_t sz, const float wx,
                       const float wy, const float wz, const bool black_border = false,
                       const int parallel = 1, float* workspace = NULL) {
  (void)parallel;
  Result out(workspace, (size_t)(sx * sy * sz));
  check(edt_hip_binary_edtsq(img, dtype_code<T>(), 3, sx, sy, sz, wx, wy, wz, black_border, 0, out.p), out.p, out.n);
  return out.release();
}
template <typename T>
float* _binary_edt3d(T* img, const int64_t sx, const int64_t sy, const int64_t sz, const float wx,
                     const float wy, const float wz, const bool black_border = false,
                     const int parallel = 1, float* workspace = NULL) {
  (void)parallel;
  Result out(workspace, (size_t)(sx * sy * sz));
  check(edt_hip_binary_edtsq(img, dtype_code<T>(), 3, sx, sy, sz, wx, wy, wz, black_border, 1, out.p), out.p, out.n);
  return out.release();
}
template <typename T>
float* _binary_edt2dsq(T* img, const int64_t sx, const int64_t sy, const float wx, const float wy,
                       const bool black_border = false, const int parallel = 1,
                       float* workspace = NULL) {
  (void)parallel;
  Result out(workspace, (size_t)(sx * sy));
  check(edt_hip_binary_edtsq(img, dtype_code<T>(), 2, sx, sy, 1, wx, wy, 1.0f, black_border, 0, out.p), out.p, out.n);
  return out.release();
}
template <typename T>
float* _binary_edt2d(T* img, const int64_t sx, const int64_t sy, const float wx, const float wy,
                     const bool black_border = false, const int parallel = 1, float* output = NULL) {
  (void)parallel;
  Result out(output, (size_t)(sx * sy));
  check(edt_hip_binary_edtsq(img, dtype_code<T>(), 2, sx, sy, 1, wx, wy, 1.0f, black_border, 1, out.p), out.p, out.n);
  return out.release();
}

// src/edt_voxel_graph.hpp:54-117, :120-214, :216-236 (GRAPH_TYPE is always uint8_t upstream)
template <typename T, typename GRAPH_TYPE = uint8_t>
float* _edt2dsq_voxel_graph(T* labels, GRAPH_TYPE* graph, const int64_t sx, const int64_t sy,
                            const float wx, const float wy, const bool black_border = false,
                            float* workspace = NULL) {
  static_assert(sizeof(GRAPH_TYPE) == 1, "voxel graph must be one byte per voxel");
  Result out(workspace, (size_t)(sx * sy));
  check(edt_hip_edt2dsq_voxel_graph(labels, dtype_code<T>(), reinterpret_cast<const uint8_t*>(graph),
                                    sx, sy, wx, wy, black_border, out.p), out.p, out.n);
  return out.release();
}
template <typename T, typename GRAPH_TYPE = uint8_t>
float* _edt3dsq_voxel_graph(T* labels, GRAPH_TYPE* graph, const int64_t sx, const int64_t sy,
                            const int64_t sz, const float wx, const float wy, const float wz,
                            const bool black_border = false, float* workspace = NULL) {
  static_assert(sizeof(GRAPH_TYPE) == 1, "voxel graph must be one byte per voxel");
  Result out(workspace, (size_t)(sx * sy * sz));
  check(edt_hip_edt3dsq_voxel_graph(labels, dtype_code<T>(), reinterpret_cast<const uint8_t*>(graph),
                                    sx, sy, sz, wx, wy, wz, black_border, out.p), out.p, out.n);
  return out.release();
}
template <typename T, typename GRAPH_TYPE = uint8_t>
float* _edt3d_voxel_graph(T* labels, GRAPH_TYPE* graph, const int64_t sx, const int64_t sy,
                          const int64_t sz, const float wx, const float wy, const float wz,
                          const bool black_border = false, float* workspace = NULL) {
  float* out = _edt3dsq_voxel_graph<T, GRAPH_TYPE>(labels, graph, sx, sy, sz, wx, wy, wz,
                                                   black_border, workspace);
  for (int64_t i = 0; i < sx * sy * sz; i++) out[i] = std::sqrt(out[i]);
  return out;
}

}  // namespace pyedt

namespace edt {

// 1-D (src/edt.hpp:807-821, :884-893)
template <typename T>
float* edt(T* labels, const int sx, const float wx, const bool black_border = false) {
  pyedt::Result out(NULL, (size_t)sx);
  float* d = out.p;
  pyedt::squared_edt_1d_multi_seg(labels, d, sx, 1, wx);  // sic: black_border not forwarded upstream
  for (int i = 0; i < sx; i++) d[i] = std::sqrt(d[i]);
  (void)black_border;
  return out.release();
}
template <typename T>
float* edtsq(T* labels, const int sx, const float wx, const bool black_border = false) {
  pyedt::Result out(NULL, (size_t)sx);
  pyedt::squared_edt_1d_multi_seg(labels, out.p, sx, 1, wx, black_border);
  return out.release();
}

// 2-D (src/edt.hpp:823-833, :895-905)
template <typename T>
float* edt(T* labels, const int sx, const int sy, const float wx, const float wy,
           const bool black_border = false, const int parallel = 1, float* output = NULL) {
  return pyedt::_edt2d(labels, sx, sy, wx, wy, black_border, parallel, output);
}
template <typename T>
float* edtsq(T* labels, const int sx, const int sy, const float wx, const float wy,
             const bool black_border = false, const int parallel = 1, float* output = NULL) {
  return pyedt::_edt2dsq(labels, sx, sy, wx, wy, black_border, parallel, output);
}

// 3-D (src/edt.hpp:836-844, :907-922)
template <typename T>
float* edt(T* labels, const int sx, const int sy, const int sz, const float wx, const float wy,
           const float wz, const bool black_border = false, const int parallel = 1,
           float* output = NULL) {
  return pyedt::_edt3d(labels, sx, sy, sz, wx, wy, wz, black_border, parallel, output);
}
template <typename T>
float* edtsq(T* labels, const int sx, const int sy, const int sz, const float wx, const float wy,
             const float wz, const bool black_border = false, const int parallel = 1,
             float* output = NULL) {
  return pyedt::_edt3dsq(labels, sx, sy, sz, wx, wy, wz, black_border, parallel, output);
}

// binary_* (src/edt.hpp:846-882, :924-951)
template <typename T>
float* binary_edt(T* labels, const int sx, const float wx, const bool black_border = false) {
  return edt::edt(labels, sx, wx, black_border);
}
template <typename T>
float* binary_edt(T* labels, const int sx, const int sy, const float wx, const float wy,
                  const bool black_border = false, const int parallel = 1, float* output = NULL) {
  return pyedt::_binary_edt2d(labels, sx, sy, wx, wy, black_border, parallel, output);
}
template <typename T>
float* binary_edt(T* labels, const int sx, const int sy, const int sz, const float wx,
                  const float wy, const float wz, const bool black_border = false,
                  const int parallel = 1, float* output = NULL) {
  return pyedt::_binary_edt3d(labels, sx, sy, sz, wx, wy, wz, black_border, parallel, output);
}
template <typename T>
float* binary_edtsq(T* labels, const int sx, const float wx, const bool black_border = false,
                    const int parallel = 1) {
  (void)parallel;
  return edt::edtsq(labels, sx, wx, black_border);
}
template <typename T>
float* binary_edtsq(T* labels, const int sx, const int sy, const float wx, const float wy,
                    const bool black_border = false, const int parallel = 1) {
  return pyedt::_binary_edt2dsq(labels, sx, sy, wx, wy, black_border, parallel);
}
template <typename T>
float* binary_edtsq(T* labels, const int sx, const int sy, const int sz, const float wx,
                    const float wy, const float wz, const bool black_border = false,
                    const int parallel = 1, float* output = NULL) {
  // upstream passes (parallel, output) in the (black_border, parallel) slots (src/edt.hpp:950);
  // this header forwards the arguments by name instead.
  return pyedt::_binary_edt3dsq(labels, sx, sy, sz, wx, wy, wz, black_border, parallel, output);
}

}  // namespace edt

#endif  // EDT_AMD_EDT_HPP
