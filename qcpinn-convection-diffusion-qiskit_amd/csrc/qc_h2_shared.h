// Definitions shared by the two forms of the HBM-family stage kernels: the run-time plan interpreter
// (qc_circuit_hbm2.hip) and the compile-time stage programs (qc_circuit_h2s_kernels.h, generated plans).
#pragma once
#include "qc_internal.h"
#include "qc_gates.h"
#include "qc_hbm2_plan.h"

// Diagnostic, timing-only builds (results are wrong): -DH2_ABLATE_GATES skips the gate arithmetic of the rounds,
// -DH2_ABLATE_SYNC drops the block barriers of the stage kernel.  Never defined for the shipped library.
#ifdef H2_ABLATE_SYNC
#define H2_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define H2_SYNC() __syncthreads()
#endif

namespace {

struct Cplx {
  float re, im;
};

constexpr int H2_XW = 24;   // floats per <Z> partial record (n <= 20 used)

// LDS swizzle of a local index (8-byte slots).  The lanes of a round vary the local positions outside its register
// group; a GF(2)-linear map of the low 5 bits chosen so that every group pattern of the tile (and the linear load /
// store pattern) is conflict-free under BOTH banking rules of the 64-bit accesses: ds_read_b64 serves a wave as two
// halves of 32 lanes over 64 dword banks (32 distinct slots mod 32), ds_write_b64 as four quarters of 16 lanes over 32
// dword banks (16 distinct slots mod 16).  Low-5 image of position p (unit vectors for p < 4):
//   RB = 4 (groups {0-3}, {4-7}, {8-11}):          4 -> 17, 5 -> 2, 6 -> 4, 7 -> 8, 8 -> 16   = low5 ^ idx[4..8]
//   RB = 3 (groups {0-2}, {3-5}, {6-8}, {9-11}):   4 -> 17, 5 -> 18, 6 -> 12, 7 -> 16
// (tools/lds_bank_check.py replays every round pattern against both rules.)
template <int RB>
__host__ __device__ __forceinline__ constexpr int h2_swz(int l) {
  if constexpr (RB == 3)
    return l ^ ((l >> 4) & 1) ^ (((l >> 5) & 1) * 18) ^ (((l >> 6) & 1) * 12) ^ (((l >> 7) & 1) * 16);
  else
    return l ^ ((l >> 4) & 31);
}
__device__ __forceinline__ Cplx cmul(Cplx a, Cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Cplx cmulc(Cplx a, Cplx b) { return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; }

// ---- kernel arguments of one stage launch (both kernel forms)
struct H2Args {
  Cplx* store;           // tile slots
  int64_t slot_elems;    // complex64 elements per slot = 2 * nch * 64 * 2^n
  int64_t B;             // points of the call ([feature][B] arrays)
  int64_t p_first;       // point index of (tile 0, t = 0) of this launch
  int64_t pt_stride;     // points covered by this launch rounded up to 64 (row stride of the partial buffers)
  int n;
  int first, last;       // this stage is the first / last of the plan
  int keep_final;        // forward, last stage: leave the final states in the slot
  H2Stage sd;
  const H2Round* rounds;
  const H2Gate* gates;
  const QcTrig* trig;
  const float* umat;
  const Cplx* tabs;      // [n_tables][2^n]
  const Cplx* rph;       // compile-time stage programs: phase records of the fused RZ runs, [n_runs][2^RB]
  const float* wd;       // [pt_stride][n][8]
  float* xpart;          // forward, last stage: [8][pt_stride][ntau][H2_XW]
  const float* qbar;     // backward, last stage: [nch][n][B]
  float* gpart;          // backward: [np][pt_stride * ntau] partials of the in-round parametric gates
  float* dpart;          // backward: [ntab][pt_stride * ntau][nc] Walsh-Hadamard coefficients of t
  Cplx* xi;              // backward, first stage: [nch][pt_stride][ntau][nx] un-embedded cotangents, weight <= 3
  const int* sparse_idx;
  const int* wht_idx;
  int nx, nc;
  int amp;               // amplitude encoding: the initial state is the feature jets themselves (plan interpreter only)
  const float* ajets;    // amp: [nch][n][B] initial-amplitude jets (forward) ;  float* abar below receives their cotangents
  float* abar;           // amp, backward first stage: [nch][n][B]
};

// Plan records are read with wave-uniform addresses; readfirstlane tells the compiler so (SGPRs instead of VGPRs for
// every index, offset and coefficient derived from them).
__device__ __forceinline__ int h2_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float h2_unif(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// Read-only plan / coefficient records through the constant address space with a wave-uniform pointer: scalar loads
// (s_load_dwordxN into SGPRs) instead of per-lane global loads followed by readfirstlane.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* h2_const(const T* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return (const __attribute__((address_space(4))) T*)(((unsigned long long)hi << 32) | lo);
}
}  // namespace
