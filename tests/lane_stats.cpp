// tests/lane_stats.cpp -- TEST/TUNING FIXTURE: the lane emulation of tests/lane_emul.cpp with event
// counters switched on.  For every slow path of the column kernel it reports how often a LANE takes it
// and how often a WAVE has to execute it (a wave pays for a path as soon as one of its 64 lanes needs
// it; loops cost the maximum trip count over the lanes).  Used by tools/lane_stats.py.
#define EDT_LANE_STATS 1
#include <cstdint>
#include <map>
#include <vector>
static int g_lane = 0;                                   // global lane id set by the emulation loop
static std::vector<std::map<std::pair<int, int>, int>> g_stat;  // [kind] -> {(lane,row) -> count}
void edt_lane_stat(int kind, int row, int count);
#include "lane_emul.cpp"
void edt_lane_stat(int kind, int row, int count) {
  if ((int)g_stat.size() <= kind) g_stat.resize(kind + 1);
  g_stat[kind][{g_lane, row}] += count;
}
extern "C" void lane_stats_reset() { g_stat.clear(); }
// per kind: out[2*kind] = sum over lanes (lane events), out[2*kind+1] = sum over (wave,row) of the max over
// the wave's lanes (wave executions)
// grouping: 0 = the kernel's waves (CW columns x all bands); 1 = "row layout" (32 columns x 2 bands),
// cw = columns per wave of the emulated kernel shape (64 / bands per column, power of two)
static int g_grouping = 0, g_cw = 4;
extern "C" void lane_stats_grouping(int mode, int cw) { g_grouping = mode; g_cw = cw; }
static int wave_of(int id) {
  if (g_grouping == 0) return id / 64;
  const int per_tile = (g_cw == 2 ? 16 : 32) * (64 / g_cw);  // lanes per tile (TileGeom)
  const int tile = id / per_tile, in = id % per_tile;
  const int wave = in / 64, lane = in % 64;
  const int band = lane / g_cw;                       // lane = c + cw*b inside the kernel's wave
  (void)wave;
  return tile * 64 + band / 2;                        // 32 columns x 2 bands
}
extern "C" void lane_stats_get(double *out, int nkinds) {
  for (int k = 0; k < nkinds; ++k) {
    double lanes = 0, waves = 0;
    if (k < (int)g_stat.size()) {
      std::map<std::pair<int, int>, int> wmax;
      for (auto &e : g_stat[k]) {
        lanes += e.second;
        auto key = std::make_pair(wave_of(e.first.first), e.first.second);
        if (wmax[key] < e.second) wmax[key] = e.second;
      }
      for (auto &e : wmax) waves += e.second;
    }
    out[2 * k] = lanes;
    out[2 * k + 1] = waves;
  }
}
extern "C" int lane_stats_lanes() { return g_lane; }
