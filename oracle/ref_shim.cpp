// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" shim around the *unmodified* reference headers, compiled from
// where they lie under /root/reference/src (see oracle/Makefile, target `ref`).
// Output goes to oracle/_ref/libedt_ref.so (git-ignored, travels with gpurun).
// Nothing from the reference is copied into this repository: this file only
// #includes the headers and forwards a dtype code to the reference templates
//   pyedt::squared_edt_1d_multi_seg<T>   (src/edt.hpp:70-119)
//   pyedt::_edt2dsq<T>                   (src/edt.hpp:632-678, bool: :758-772)
//   pyedt::_edt3dsq<T>                   (src/edt.hpp:411-484, bool: :580-587)
//   pyedt::_binary_edt{2,3}dsq<T>        (src/edt.hpp:681-732, :487-576)
//   pyedt::_edt2dsq_voxel_graph<T,u8>    (src/edt_voxel_graph.hpp:54-117)
//   pyedt::_edt3dsq_voxel_graph<T,u8>    (src/edt_voxel_graph.hpp:120-214)
// exactly as the reference Cython binding does (src/edt.pyx:62-113).
#include <cstdint>
#include <map>
#include <stdexcept>
#include <vector>
#include "edt.hpp"
#include "edt_voxel_graph.hpp"

// dtype codes shared with include/edt_hip.h
enum { DT_U8 = 0, DT_U16 = 1, DT_U32 = 2, DT_U64 = 3, DT_F32 = 4, DT_F64 = 5, DT_BOOL = 6 };

#define DISPATCH(CALL)                                             \
  switch (dtype) {                                                 \
    case DT_U8:  { typedef uint8_t  T; CALL; return 0; }          \
    case DT_U16: { typedef uint16_t T; CALL; return 0; }          \
    case DT_U32: { typedef uint32_t T; CALL; return 0; }          \
    case DT_U64: { typedef uint64_t T; CALL; return 0; }          \
    case DT_F32: { typedef float    T; CALL; return 0; }          \
    case DT_F64: { typedef double   T; CALL; return 0; }          \
    case DT_BOOL:{ typedef bool     T; CALL; return 0; }          \
    default: return -1;                                            \
  }

extern "C" {

int ref_edt1dsq(void* labels, int dtype, int64_t n, float w, int bb, float* out) {
  DISPATCH(pyedt::squared_edt_1d_multi_seg<T>((T*)labels, out, n, 1, w, bb != 0))
}

int ref_edt2dsq(void* labels, int dtype, int64_t sx, int64_t sy, float wx, float wy,
                int bb, int parallel, float* out) {
  DISPATCH(pyedt::_edt2dsq<T>((T*)labels, sx, sy, wx, wy, bb != 0, parallel, out))
}

int ref_edt3dsq(void* labels, int dtype, int64_t sx, int64_t sy, int64_t sz,
                float wx, float wy, float wz, int bb, int parallel, float* out) {
  DISPATCH(pyedt::_edt3dsq<T>((T*)labels, sx, sy, sz, wx, wy, wz, bb != 0, parallel, out))
}

// the binary route as the C++ facade reaches it for ANY label type (edt::binary_edt* -> pyedt::_binary_edt{2,3}dsq<T>,
// src/edt.hpp:487-576, :681-732): labels split runs in pass 1 only
int ref_binary_edt2dsq(void* labels, int dtype, int64_t sx, int64_t sy, float wx, float wy,
                       int bb, int parallel, float* out) {
  DISPATCH(pyedt::_binary_edt2dsq<T>((T*)labels, sx, sy, wx, wy, bb != 0, parallel, out))
}

int ref_binary_edt3dsq(void* labels, int dtype, int64_t sx, int64_t sy, int64_t sz,
                       float wx, float wy, float wz, int bb, int parallel, float* out) {
  DISPATCH(pyedt::_binary_edt3dsq<T>((T*)labels, sx, sy, sz, wx, wy, wz, bb != 0, parallel, out))
}

int ref_edt2dsq_voxel_graph(void* labels, int dtype, uint8_t* graph, int64_t sx, int64_t sy,
                            float wx, float wy, int bb, float* out) {
  DISPATCH((pyedt::_edt2dsq_voxel_graph<T, uint8_t>((T*)labels, graph, sx, sy, wx, wy, bb != 0, out)))
}

int ref_edt3dsq_voxel_graph(void* labels, int dtype, uint8_t* graph, int64_t sx, int64_t sy,
                            int64_t sz, float wx, float wy, float wz, int bb, float* out) {
  DISPATCH((pyedt::_edt3dsq_voxel_graph<T, uint8_t>((T*)labels, graph, sx, sy, sz, wx, wy, wz, bb != 0, out)))
}

}  // extern "C"
