// Calibration of rocprofv3's MeanOccupancyPerCU: kernels whose occupancy is known by construction.
//   spin<THREADS>: every wave spins for about `cycles` clock ticks; grid large enough to keep every slot busy.
// build: hipcc --offload-arch=gfx950 -O2 tools/occupancy_probe.hip -o tools/occupancy_probe
// run:   rocprofv3 --pmc MeanOccupancyPerCU --output-format csv -d out -- tools/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int THREADS, int WAVES_PER_SIMD>
__global__ __launch_bounds__(THREADS, WAVES_PER_SIMD) void spin(long long cycles, int* out) {
  const long long t0 = wall_clock64();
  int acc = 0;
  while (wall_clock64() - t0 < cycles) acc += threadIdx.x;
  if (acc == -1) out[0] = acc;
}

int main() {
  int* out;
  if (hipMalloc(&out, 4) != hipSuccess) return 1;
  // wall_clock64 runs at 100 MHz.  Long-lived waves first (20 us, 8 rounds of the chip), then short-lived ones
  // (5 us and 2 us, 64 rounds) with and without dynamic LDS: what is lost between a wave's end and its successor's start.
  for (long long ticks : {2000ll, 500ll, 200ll}) {
    const int rounds = ticks == 2000 ? 8 : 64;
    spin<64, 8><<<256 * 32 * rounds, 64>>>(ticks, out);
    spin<128, 8><<<256 * 16 * rounds, 128>>>(ticks, out);
    spin<256, 8><<<256 * 8 * rounds, 256>>>(ticks, out);
    spin<128, 4><<<256 * 16 * rounds, 128, 4352>>>(ticks, out);
  }
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  printf("done\n");
  return 0;
}
