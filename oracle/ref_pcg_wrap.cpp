// ref_pcg_wrap.cpp -- C entry points around the REFERENCE's own Pcg class.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes
// the reference's Sampler.h where it lies (the Makefile passes
// -I$(REFERENCE_DIR)) and is compiled only when that directory exists.  The
// result, oracle/_ref/libref_pcg.so, is used to pin oracle/rt_oracle.cpp's
// restatement of the generator and to produce tests/golden/pcg_kat.json.
// Sampler.h and globals.h are the only files of the reference's hot path that compile
// without headers this image lacks (every other one includes SFML through Vec3.h or
// Embree 3 through Obj.h/Scene.h).
#include <cstddef>
#include <cstdint>

#include "Sampler.h"  // reference: Sampler.h:11-42
#include "globals.h"  // reference: globals.h:3-8 (macros only; it includes nothing)

extern "C" {
uint32_t ref_pcg_hash(uint32_t input) {
  Pcg g;
  return g.pcg_hash(input);
}
void ref_pcg_rand(uint32_t* state, uint32_t* out, size_t n) {
  Pcg g(*state);
  for (size_t i = 0; i < n; i++) out[i] = g.rand_pcg();
  *state = g.rng_state;
}
// The reference's compile-time constants, as its own header defines them.
float ref_eps(void) { return EPS; }
float ref_nee_emission_threshold(void) { return NEE_EMISSION_THRESHOLD; }
int ref_width(void) { return WIDTH; }
int ref_height(void) { return HEIGHT; }
void ref_pcg_uniform(uint32_t* state, float* out, size_t n) {
  Pcg g(*state);
  for (size_t i = 0; i < n; i++) out[i] = g.Uniform();
  *state = g.rng_state;
}
}
