// Host-only build for the sanitizers (make asan): the kernel launchers of the .hip files, which this build does not
// contain.  The sanitizer run happens on a machine without a GPU, where no evaluation gets as far as a launch
// (bild_trajset_create fails with BILD_ERR_NO_DEVICE first); the geometry tables are host code and are duplicated here
// only as far as packing needs them (padded_rows), with the same values as kernels.hip.
#include "common.h"
#include "amis_math.h"
#include "internal.h"

namespace bild {
int padded_rows(int n)
{
    static const int rows[] = {4, 8, 10, 12, 16, 20, 24, 28, 32};
    for (int r : rows)
        if (r >= n) return r;
    return 0;
}
bool geometry_for(int, int, int64_t, int, Geometry *) { return false; }
const char *kernel_name(const Geometry &, int) { return "none"; }
int launch_logl(const Geometry &, int, const KParams &, int, size_t, void *, void *, void *) { return 1; }
int launch_reduce_partials(const double *, double *, int64_t, int, void *) { return 1; }
int launch_prefix_L(const TrajDesc *, int, int, int, int, int, double *, double *, void *) { return 1; }
int launch_validate(const int32_t *, const int32_t *, const int32_t *, const int32_t *, int64_t, int, int, int, int *, void *) { return 1; }
bool builder_geometry(int, Geometry *) { return false; }
bool listed_geometry(const Geometry &, Geometry *) { return false; }
bool dense_mfma_supported(int NP) { return NP % 4 == 0 && NP >= 4 && NP <= 24; }
int launch_logl_dense_mfma(int, const KParams &, void *) { return 1; }
bool modal_mfma_supported(int NP) { return NP == 36 || NP == 40; }
int launch_logl_modal_mfma(int, const KParams &, void *) { return 1; }
size_t wide_lds_bytes(int) { return 0; }
int launch_pair_tasks(const int64_t *, int, const TrajDesc *, int, int, int64_t, int32_t *, int32_t *, int32_t *, void *) { return 1; }
size_t device_schedule_bytes(int64_t) { return 0; }
int device_schedule(const int32_t *, const int32_t *, const TrajDesc *, int, int64_t, int, int, int, int, int64_t, void *, size_t, const int32_t **, void *) { return 1; }
int launch_logl_wide(int, const KParams &, int, void *) { return 1; }
int launch_walk(const WalkParams &, void *, void *, void *) { return 1; }
int launch_tail(const TrajDesc *, int, int, int, int, int, const double *, const double *, const int64_t *, int64_t, double *, double *, void *) { return 1; }
int launch_mark_refused_rows(const int32_t *, int, int64_t, double *, void *) { return 1; }
int amis_dev_pass_a_rows(int64_t, int64_t) { return 0; }
int amis_dev_draw(int, int, int64_t, uint64_t, uint64_t, const double *, const double *, const uint8_t *, double *, uint8_t *, void *) { return 1; }
int amis_dev_pass_a(const AmisView &, int64_t, int64_t, int64_t, double, double *, double *, double *, double *, double *, int *, void *, const uint8_t *, uint8_t *, int32_t *, int32_t *, int32_t *, double *) { return 1; }
int amis_dev_pass_b(const AmisView &, int64_t, double, int, const double *, double *, double *, int, void *) { return 1; }
int amis_dev_pass_c(const AmisView &, int64_t, const double *, double, const double *, const double *, double *, int, void *) { return 1; }
} // namespace bild
#include "exchange.h"
namespace bild {
int launch_exchange(const ExParams &, void *) { return 1; }
} // namespace bild
