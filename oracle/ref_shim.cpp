// oracle/ref_shim.cpp -- C-linkage doorway onto the REFERENCE's own naive attention.
//
// TEST INFRASTRUCTURE ONLY.  Holds no attention code of its own: it #includes the
// reference's files where they lie (the Makefile passes -I$(REF)/src) and re-exports their
// functions with C linkage so ctypes can call them.  Built ONLY in the dev container
// (where /root/reference exists) into oracle/_ref/libref_naive.so, which is git-ignored
// and is used (a) to prove oracle/naive_attention.c bit-identical to the reference and
// (b) to generate tests/golden/ (oracle/gen_golden.py).
#include <cmath>     // the reference header relies on its includer for sqrtf/expf/fmax
#include <cfloat>
#include <cstring>
#include <cstdio>
#include "util/naive_attention.h"

// src/00_naive_attention/main.cpp defines naive_attention() and a main(); rename the
// latter so the translation unit can live inside a shared object.
#define main ref00_self_test_main
#include "00_naive_attention/main.cpp"
#undef main

extern "C" {

void ref_naive_attention(const float* Q, const float* K, const float* V, float* O, int N, int d)
{ naive_attention(Q, K, V, O, N, d); }

void ref_naive_forward_pass(const float* Q, const float* K, const float* V, float* O, float* L,
                            int N, int d, float scale)
{ naive_forward_pass(Q, K, V, O, L, N, d, scale); }

void ref_naive_attention_forward(const float* Q, const float* K, const float* V, float* O,
                                 int N, int d, float scale)
{ naive_attention_forward(Q, K, V, O, N, d, scale); }

void ref_naive_attention_backward(const float* Q, const float* K, const float* V, const float* O,
                                  const float* L, const float* dO, float* dQ, float* dK, float* dV,
                                  int N, int d, float scale)
{ naive_attention_backward(Q, K, V, O, L, dO, dQ, dK, dV, N, d, scale); }

// The reference's own 2x2 known-answer self test (main.cpp:40-85); returns its exit code.
int ref_naive00_self_test(void) { return ref00_self_test_main(); }

}  // extern "C"
