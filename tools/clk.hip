// developer probe: shader clock vs 100 MHz wall clock, and cycles per dependent VALU op for a lone wave
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(long long *out, int n)
{
	long long c0 = clock64(), w0 = wall_clock64();
	unsigned x = threadIdx.x;
	for (int i = 0; i < n; ++i) {
		x = x * 3u + 1u;      // dependent chain: v_mad / v_mul+v_add
		x ^= x >> 3;
	}
	long long c1 = clock64(), w1 = wall_clock64();
	if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = x; }
}
int main()
{
	long long *d, h[3];
	hipMalloc(&d, 24);
	for (int rep = 0; rep < 3; ++rep) {
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1000000);
		hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
		printf("clock64 %lld wall(100MHz) %lld -> clock64 rate %.1f MHz; %.2f clock64 ticks / iteration (3 dependent ops)\n", h[0], h[1], h[0] / (h[1] / 100.0), h[0] / 1e6);
	}
	return 0;
}
