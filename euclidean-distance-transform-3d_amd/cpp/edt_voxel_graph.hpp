// edt_voxel_graph.hpp -- second half of the drop-in header pair.
//
// The reference's Cython binding takes its symbols from TWO headers: "edt.hpp" and
// "edt_voxel_graph.hpp" (src/edt.pyx:62-87 and :89-113).  The voxel-graph transforms themselves
// (pyedt::_edt2dsq_voxel_graph / _edt3dsq_voxel_graph / _edt3d_voxel_graph, reference:
// src/edt_voxel_graph.hpp:54-236) are forwarders into the C ABI and live in edt.hpp next to the other
// transforms; this file re-exports them and adds the three host-side run utilities the binding also
// expects under pyedt:: (reference: src/edt_voxel_graph.hpp:238-310, bound at src/edt.pyx:101-113 and used by
// edt.runs / draw / erase / transfer / each, src/edt.pyx:847-994):
//
//   extract_runs(labels, voxels)              -> map label -> list of half-open [start, end) runs of that
//                                                label in the flattened array, in order of appearance
//   set_run_voxels(val, runs, labels, voxels)  -> paint every run with val
//   transfer_run_voxels(runs, src, dest, vox)  -> copy the voxels of every run from src to dest
//
// Both writers reject a run that is empty, reversed or reaches outside [0, voxels] with
// std::runtime_error("Invalid run.") -- the message the reference throws (src/edt_voxel_graph.hpp:277-283,
// :299-305) and the Cython layer turns into a Python RuntimeError (`except +`).
// They are consumers of the distance transform that walk host memory run by run; there is nothing for a
// GPU to do here, so they are plain inline C++ (the device-resident counterpart is edt_hip_select_label_device).
#ifndef EDT_AMD_EDT_VOXEL_GRAPH_HPP
#define EDT_AMD_EDT_VOXEL_GRAPH_HPP

#include <algorithm>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <utility>
#include <vector>

#include "edt.hpp"

namespace pyedt {

typedef std::pair<int64_t, int64_t> voxel_run;  // [first, second)

template <typename T>
std::map<T, std::vector<voxel_run>> extract_runs(T* labels, const int64_t voxels) {
  std::map<T, std::vector<voxel_run>> by_label;
  int64_t begin = 0;
  while (begin < voxels) {
    const T value = labels[begin];
    int64_t end = begin + 1;
    while (end < voxels && labels[end] == value) end++;
    by_label[value].emplace_back(begin, end);
    begin = end;
  }
  return by_label;
}

namespace detail {
inline void require_valid_run(const voxel_run& r, const int64_t voxels) {
  const bool inside = r.first >= 0 && r.second >= 0 && r.second <= voxels;
  if (!inside || r.first >= r.second) throw std::runtime_error("Invalid run.");
}
}  // namespace detail

template <typename T>
void set_run_voxels(const T val, const std::vector<voxel_run> runs, T* labels, const int64_t voxels) {
  // validated one run at a time, like upstream: the runs before a bad one are already painted when it throws
  for (const voxel_run& r : runs) {
    detail::require_valid_run(r, voxels);
    std::fill(labels + r.first, labels + r.second, val);
  }
}

template <typename T = float>
void transfer_run_voxels(const std::vector<voxel_run> runs, T* src, T* dest, const int64_t voxels) {
  for (const voxel_run& r : runs) {
    detail::require_valid_run(r, voxels);
    std::copy(src + r.first, src + r.second, dest + r.first);
  }
}

}  // namespace pyedt

#endif  // EDT_AMD_EDT_VOXEL_GRAPH_HPP
