// Entry points between the translation units of the library (not part of the C ABI).
#pragma once
#include <stdint.h>

struct bild_model;
struct bild_trajset;

namespace bild {

// One batch of the sampler's own (s, theta) rows that are ALREADY in HBM (ss: n x K1 float64, thetas: n x K1 uint8), on the
// model's own stream, results to d_out (device, n doubles); nothing is waited for.  `status` (2 ints, device-visible,
// zeroed by the caller) reports rows that are no points on the simplex.  Serialised per model like the host-buffer calls;
// the model's stream is returned in *stream (a hipStream_t).  K1 <= 16.
int internal_logl_st_resident(const bild_model *m, const bild_trajset *ts, int64_t n, int K1, const double *d_ss, const uint8_t *d_thetas,
                              unsigned flags, double *d_out, int32_t *status, void **stream);

// the model's own stream (a hipStream_t): device copies of the model are made on first use; null on failure
void *internal_model_stream(const bild_model *m);

} // namespace bild
