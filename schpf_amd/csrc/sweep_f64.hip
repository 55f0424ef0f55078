// float64 instantiation of the sweep kernels (see sweep_impl.h)
#include "sweep_impl.h"
namespace schpf {
template hipError_t launch_sweep<double>(const SweepArgs<double> &, int, int, int, int64_t, hipStream_t);
template hipError_t launch_random_phi<double>(const SweepArgs<double> &, int, int, uint64_t, int, int64_t, hipStream_t);
template hipError_t launch_tile_sweep<double>(const TileArgs<double> &, int, int, int, int, int64_t, int, size_t, hipStream_t);
template hipError_t launch_tile_sweep_dual<double>(const TileArgs<double> &, const TileArgs<double> &, const int *, int, int, int, int64_t, int, size_t, int *, int, hipStream_t);
}  // namespace schpf
