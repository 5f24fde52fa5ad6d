// tools/queue_probe.hip -- developer tool: which HIP streams share a hardware queue?
// Two one-workgroup kernels that spin for a fixed number of clock ticks, one per stream: on different queues
// they run side by side (elapsed ~ 1x), on one queue one after the other (~ 2x).
//   hipcc --offload-arch=gfx950 -O2 -o tools/queue_probe tools/queue_probe.hip && gpurun -- ./tools/queue_probe 12
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(long long ticks, int *sink)
{
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) {
	}
	if (sink && ticks < 0) {
		*sink = 1;
	}
}

static double pair_ms(hipStream_t a, hipStream_t b, long long ticks)
{
	hipDeviceSynchronize();
	const auto t0 = std::chrono::steady_clock::now();
	hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, ticks, nullptr);
	hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, ticks, nullptr);
	hipStreamSynchronize(a);
	hipStreamSynchronize(b);
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv)
{
	const int n = argc > 1 ? atoi(argv[1]) : 8;
	const int prio = argc > 2 ? atoi(argv[2]) : 0;
	const long long ticks = 100000 * 50;        // wall_clock64 runs at 100 MHz: 50 ms
	std::vector<hipStream_t> s(n);
	int lo = 0, hi = 0;
	hipDeviceGetStreamPriorityRange(&lo, &hi);
	printf("priority range: least %d greatest %d\n", lo, hi);
	for (int i = 0; i < n; ++i) {
		if (hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, prio) != hipSuccess) {
			printf("create failed\n");
			return 1;
		}
	}
	// first use in creation order
	for (int i = 0; i < n; ++i) {
		hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], 1000, nullptr);
	}
	hipDeviceSynchronize();
	const double one = pair_ms(s[0], s[0], ticks / 2);
	printf("same stream, two halves: %.1f ms\n", one);
	printf("     ");
	for (int j = 0; j < n; ++j) {
		printf("%5d", j);
	}
	printf("   (S = serialised)\n");
	for (int i = 0; i < n; ++i) {
		printf("%3d: ", i);
		for (int j = 0; j < n; ++j) {
			if (j <= i) {
				printf("    .");
				continue;
			}
			const double ms = pair_ms(s[i], s[j], ticks);
			printf("%5s", ms > 75.0 ? "S" : "-");
		}
		printf("\n");
	}
	// the null stream against each
	printf("null:");
	for (int j = 0; j < n; ++j) {
		const double ms = pair_ms(nullptr, s[j], ticks);
		printf("%5s", ms > 75.0 ? "S" : "-");
	}
	printf("\n");
	return 0;
}
