// Compiles the drop-in C++ header pair against the C ABI and exercises the reference's public signatures
// (edt::edt / edtsq / binary_edt, pyedt::_edt3dsq / _edt2dsq / _edt3dsq_voxel_graph, extract_runs ...).
//
//   cpp_dropin                 fixed 3x3x3 known answers
//   cpp_dropin <cases.bin>     randomized parity: the file (written by tests/test_gpu_parity.py from the CPU
//                              oracle) holds cases { dtype code, ndim, sx, sy, sz, wx, wy, wz, black_border,
//                              voxel_graph flag, labels, [graph], expected edtsq }; every case is run through the
//                              template of its label type and compared bit for bit (edt: sqrt of it).
// Exit code: 0 = everything matches, 1 = wrong values, 3 = the library reported "no device" (CPU-only host).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "edt_voxel_graph.hpp"

namespace {

template <typename T>
bool run_case(const std::vector<char>& lab, const std::vector<uint8_t>& graph, int mode, int ndim, int64_t sx,
              int64_t sy, int64_t sz, float wx, float wy, float wz, bool bb, const std::vector<float>& want) {
  T* l = reinterpret_cast<T*>(const_cast<char*>(lab.data()));
  const int64_t vox = sx * sy * sz;
  std::vector<float> got(vox, -1.0f);
  float* owned = nullptr;
  const bool has_graph = mode == 1;
  if (mode == 2) {
    // the binary route through the facade (edt::binary_edtsq / binary_edt -> pyedt::_binary_edt{2,3}d[sq]<T>):
    // `want` holds the reference's values for a MULTI-VALUED image (labels split runs along x only)
    if (ndim == 2) {
      owned = edt::binary_edtsq<T>(l, (int)sx, (int)sy, wx, wy, bb, 1);
      std::memcpy(got.data(), owned, vox * sizeof(float));
      delete[] owned;
      float* d = edt::binary_edt<T>(l, (int)sx, (int)sy, wx, wy, bb, 1);
      bool ok = true;
      for (int64_t i = 0; i < vox && ok; i++) ok = d[i] == std::sqrt(want[i]);
      delete[] d;
      if (!ok) return false;
    } else {
      edt::binary_edtsq<T>(l, (int)sx, (int)sy, (int)sz, wx, wy, wz, bb, 1, got.data());
      float* d = edt::binary_edt<T>(l, (int)sx, (int)sy, (int)sz, wx, wy, wz, bb, 1);
      bool ok = true;
      for (int64_t i = 0; i < vox && ok; i++) ok = d[i] == std::sqrt(want[i]);
      delete[] d;
      if (!ok) return false;
    }
    return std::memcmp(got.data(), want.data(), vox * sizeof(float)) == 0;
  }
  if (has_graph) {
    uint8_t* g = const_cast<uint8_t*>(graph.data());
    if (ndim == 2) pyedt::_edt2dsq_voxel_graph<T, uint8_t>(l, g, sx, sy, wx, wy, bb, got.data());
    else pyedt::_edt3dsq_voxel_graph<T, uint8_t>(l, g, sx, sy, sz, wx, wy, wz, bb, got.data());
  } else if (ndim == 1) {
    pyedt::squared_edt_1d_multi_seg<T>(l, got.data(), sx, 1, wx, bb);
  } else if (ndim == 2) {
    pyedt::_edt2dsq<T>(l, sx, sy, wx, wy, bb, 1, got.data());
  } else {
    // alternate between the caller-owned and the library-allocated output (ownership: src/edt.hpp:424-426)
    if (sx & 1) pyedt::_edt3dsq<T>(l, sx, sy, sz, wx, wy, wz, bb, 2, got.data());
    else {
      owned = edt::edtsq<T>(l, (int)sx, (int)sy, (int)sz, wx, wy, wz, bb, 1);
      std::memcpy(got.data(), owned, vox * sizeof(float));
      delete[] owned;
    }
  }
  if (std::memcmp(got.data(), want.data(), vox * sizeof(float)) != 0) return false;
  if (!has_graph && ndim == 3) {  // the sqrt entry point: correctly rounded root of the same values
    float* d = edt::edt<T>(l, (int)sx, (int)sy, (int)sz, wx, wy, wz, bb, 1);
    bool ok = true;
    for (int64_t i = 0; i < vox && ok; i++) ok = d[i] == std::sqrt(want[i]);
    delete[] d;
    if (!ok) return false;
  }
  return true;
}

int run_file(const char* path) {
  std::FILE* f = std::fopen(path, "rb");
  if (!f) { std::printf("cannot open %s\n", path); return 2; }
  int32_t ncases = 0;
  if (std::fread(&ncases, 4, 1, f) != 1) return 2;
  int bad = 0;
  for (int c = 0; c < ncases; c++) {
    int32_t head[7];   // dtype, ndim, sx, sy, sz, bb, mode (0: edtsq, 1: voxel graph follows, 2: binary route)
    float w[3];
    if (std::fread(head, 4, 7, f) != 7 || std::fread(w, 4, 3, f) != 3) return 2;
    const int dtype = head[0], ndim = head[1];
    const int64_t sx = head[2], sy = head[3], sz = head[4], vox = sx * sy * sz;
    static const int size_of[] = {1, 2, 4, 8, 4, 8, 1};
    std::vector<char> lab((size_t)vox * size_of[dtype]);
    std::vector<uint8_t> graph(head[6] == 1 ? vox : 0);
    std::vector<float> want(vox);
    if (std::fread(lab.data(), 1, lab.size(), f) != lab.size()) return 2;
    if (head[6] == 1 && std::fread(graph.data(), 1, graph.size(), f) != graph.size()) return 2;
    if (std::fread(want.data(), 4, vox, f) != (size_t)vox) return 2;
    bool ok = false;
    switch (dtype) {
      case EDT_U8: ok = run_case<uint8_t>(lab, graph, head[6], ndim, sx, sy, sz, w[0], w[1], w[2], head[5], want); break;
      case EDT_U16: ok = run_case<uint16_t>(lab, graph, head[6], ndim, sx, sy, sz, w[0], w[1], w[2], head[5], want); break;
      case EDT_U32: ok = run_case<uint32_t>(lab, graph, head[6], ndim, sx, sy, sz, w[0], w[1], w[2], head[5], want); break;
      case EDT_U64: ok = run_case<uint64_t>(lab, graph, head[6], ndim, sx, sy, sz, w[0], w[1], w[2], head[5], want); break;
      case EDT_F32: ok = run_case<float>(lab, graph, head[6], ndim, sx, sy, sz, w[0], w[1], w[2], head[5], want); break;
      case EDT_F64: ok = run_case<double>(lab, graph, head[6], ndim, sx, sy, sz, w[0], w[1], w[2], head[5], want); break;
      case EDT_BOOL: ok = run_case<bool>(lab, graph, head[6], ndim, sx, sy, sz, w[0], w[1], w[2], head[5], want); break;
    }
    if (!ok) { std::printf("case %d (dtype %d, %dD %lldx%lldx%lld) differs\n", c, dtype, ndim, (long long)sx, (long long)sy, (long long)sz); bad++; }
  }
  std::fclose(f);
  std::printf("%d of %d cases match\n", ncases - bad, ncases);
  return bad ? 1 : 0;
}

}  // namespace

int main(int argc, char** argv) {
  const int n = 3;
  std::vector<uint32_t> lab(n * n * n, 1u);
  try {
    if (argc > 1) return run_file(argv[1]);
    float* sq = edt::edtsq<uint32_t>(lab.data(), n, n, n, 4.f, 4.f, 4.f, true);
    float* d = edt::edt<uint32_t>(lab.data(), n, n, n, 6.f, 6.f, 5.f, true, 2);
    std::vector<float> out(n * n * n);
    pyedt::_edt3dsq<uint32_t>(lab.data(), n, n, n, 1.f, 1.f, 1.f, false, 1, out.data());
    bool ok = sq[13] == 64.f && sq[0] == 16.f && d[13] == 10.f && std::isinf(out[13]);
    std::vector<uint8_t> img(n * n, 1);
    float* b2 = edt::binary_edtsq<uint8_t>(img.data(), n, n, 1.f, 1.f, true);
    ok = ok && b2[4] == 4.f;
    float* one = edt::edtsq<uint16_t>(reinterpret_cast<uint16_t*>(lab.data()), 4, 2.f, true);
    // the host-side run utilities of edt_voxel_graph.hpp
    std::vector<uint16_t> r = {7, 7, 0, 7, 3, 3};
    auto runs = pyedt::extract_runs<uint16_t>(r.data(), (int64_t)r.size());
    ok = ok && runs.size() == 3 && runs[7].size() == 2 && runs[7][1] == std::make_pair<int64_t, int64_t>(3, 4);
    pyedt::set_run_voxels<uint16_t>(9, runs[7], r.data(), (int64_t)r.size());
    ok = ok && r[0] == 9 && r[1] == 9 && r[2] == 0 && r[3] == 9;
    bool threw = false;
    try { pyedt::set_run_voxels<uint16_t>(1, {{2, 2}}, r.data(), (int64_t)r.size()); }
    catch (const std::runtime_error& e) { threw = std::strcmp(e.what(), "Invalid run.") == 0; }
    ok = ok && threw;
    delete[] sq; delete[] d; delete[] b2; delete[] one;
    std::printf(ok ? "cpp drop-in ok\n" : "cpp drop-in WRONG VALUES\n");
    return ok ? 0 : 1;
  } catch (const std::runtime_error& e) {
    std::printf("caught: %s\n", e.what());
    return std::strstr(e.what(), "no HIP device") ? 3 : 2;
  }
}
