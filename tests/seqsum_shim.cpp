// TEST FIXTURE: edt_seq_sum_at (csrc/edt_seqsum.h) compiled for the host, next to the plain loop it must equal.
#include "edt_seqsum.h"
extern "C" float seqsum_jump(float w, long long k) { return edt_seq_sum_at(w, k); }
extern "C" float seqsum_loop(float w, long long k) {
  volatile float t = 0.0f;
  for (long long i = 0; i < k; ++i) t = t + w;
  return t;
}
// every checkpoint of one walk: returns the number of mismatches between the loop and the jump at the listed k (sorted)
extern "C" long long seqsum_check(float w, const long long *ks, long long nk) {
  volatile float t = 0.0f;
  long long at = 0, bad = 0;
  for (long long q = 0; q < nk; ++q) {
    for (; at < ks[q]; ++at) t = t + w;
    const float a = t, b = edt_seq_sum_at(w, ks[q]);
    if (!(a == b) && !(a != a && b != b)) ++bad;
  }
  return bad;
}
