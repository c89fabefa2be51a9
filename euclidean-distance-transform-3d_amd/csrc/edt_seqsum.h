// edt_seqsum.h -- T[k] of the reference's pass 1 without walking k steps.
//
// Pass 1 (src/edt.hpp:92-114) accumulates distances as SEQUENTIAL fp32 sums of the voxel size: T[0] = 0,
// T[k] = fl32(T[k-1] + w).  Where k*w is not exactly representable these sums have no closed form in k -- but inside
// one binade [2^(e-1), 2^e) they are an arithmetic progression: every value is a multiple of the binade's ulp u, and
// t + w rounds to t + q*u or t + (q+1)*u with q = floor(w/u), by a rule that depends only on the remainder of w (a tie,
// remainder exactly u/2, rounds to even: after the first step inside the binade the value is even and the rule repeats
// itself).  So the sum can JUMP through a binade: take real fp32 steps until two consecutive increments agree, then
// advance by as many steps as stay strictly inside the binade, all at once (exact in fp64), and cross the boundary
// with real steps again.  O(number of binades) instead of O(k); bit-identical to the loop (tests/test_seqsum.py holds
// it against the loop for thousands of voxel sizes, ties and stagnating sums included).
//
// Used to build the table of pass 1 in parallel (edt_line.hip: k_line_ttab): a 2^31-voxel line at a voxel size like
// 0.1 needed one thread to perform 2^31 dependent additions.
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef EDT_HD
#if defined(__HIPCC__)
#define EDT_HD __host__ __device__ inline
#else
#define EDT_HD inline
#endif
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

EDT_HD float edt_seq_sum_at(float w, int64_t k) {
  if (w < 0.0f) return -edt_seq_sum_at(-w, k);  // (round-to-nearest is symmetric: the sums of -w are the negated sums of w)
  float t = 0.0f;
  int64_t done = 0;
  while (done < k) {
    // one real step, then look at the next two increments
    t = t + w;
    ++done;
    if (done >= k) break;
    // voxel sizes the jump conditions below can never serve -- NaN, +-inf, zero, negative, subnormal sums -- would walk the
    // whole line one addition at a time (a thread per chunk, each k0 steps: 2^31-voxel lines would take hours).  Their
    // sums are fixed points or settle at once: NaN and inf stay, a zero step adds nothing; negative and subnormal steps
    // are walked only while the sum still moves in whole steps (the sequential loop below is exact for them).
    if (!(t == t) || t - t != 0.0f) return t;        // NaN or +-inf: every further sum is the same value
    if (w == 0.0f) return t;
    if (t < 2.3509887e-38f && w < 2.3509887e-38f) {
      // below 2^-125 every value is a multiple of the smallest subnormal and the additions are exact: jump to just
      // below that bound in one go
      const double room = (2.350988701644575e-38 - (double)t) / (double)w;
      int64_t j = room > 4.0e18 ? (int64_t)4000000000000000000ll : (int64_t)room - 1;
      if (j > k - done) j = k - done;
      if (j > 0) {
        t = (float)((double)t + (double)j * (double)w);  // exact
        done += j;
        continue;
      }
    }
    const float t1 = t + w;
    const float t2 = t1 + w;
    const float d1 = t1 - t, d2 = t2 - t1;
    if (!(d1 == d1) || !(d2 == d2)) continue;        // (inf - inf: the sum has overflowed; real steps finish the job)
    if (d1 == 0.0f && d2 == 0.0f) return t;          // w is below half an ulp of t: the sum no longer moves
    int e0, e2;
    (void)frexpf(t, &e0);
    (void)frexpf(t2, &e2);
    // steady only if the two increments agree, all three values share a binade, and that binade has normal ulps
    if (d1 != d2 || e0 != e2 || e0 < -125 || !(t > 0.0f)) continue;  // (e0 >= -125: ulps of 2^-149 and up, exact in fp64)
    // values are multiples of u = 2^(e0-24); steps left strictly inside the binade [2^(e0-1), 2^e0):
    const double u = ldexp(1.0, e0 - 24);
    const int64_t A = (int64_t)((ldexp(1.0, e0) - (double)t) / u);   // exact: both are multiples of u, quotient < 2^24
    const int64_t D = (int64_t)((double)d1 / u);                      // exact, >= 1 here (d1 == d2 != 0 or handled above)
    if (D <= 0) continue;
    int64_t j = (A - 1) / D;                                          // largest j with t + j*d1 < 2^e0
    if (j > k - done) j = k - done;
    if (j > 0) {
      t = (float)((double)t + (double)j * (double)d1);               // exact: a multiple of u inside the binade
      done += j;
    }
  }
  return t;
}
