// float32 instantiation of the sweep kernels (see sweep_impl.h)
#include "sweep_impl.h"
namespace schpf {
template hipError_t launch_sweep<float>(const SweepArgs<float> &, int, int, int, int64_t, hipStream_t);
template hipError_t launch_random_phi<float>(const SweepArgs<float> &, int, int, uint64_t, int, int64_t, hipStream_t);
template hipError_t launch_tile_sweep<float>(const TileArgs<float> &, int, int, int, int, int64_t, int, size_t, hipStream_t);
template hipError_t launch_tile_sweep_dual<float>(const TileArgs<float> &, const TileArgs<float> &, const int *, int, int, int, int64_t, int, size_t, int *, int, hipStream_t);
}  // namespace schpf
