// tools/valu_rate.hip -- developer microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the
// vector instructions the pixel kernels are made of.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float seed, int n)
{
	float a[8];
	double d[8];
	f2 p[8];
	int iv[8];
	for (int i = 0; i < 8; ++i) {
		a[i] = seed + i + threadIdx.x;
		d[i] = a[i];
		p[i] = f2{a[i], a[i] + 1.f};
		iv[i] = (int)a[i];
	}
	const float c = seed * 0.999f;
	const double cd = c;
	const f2 cp = {c, c * 1.01f};
	for (int it = 0; it < n; ++it) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
			if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
			if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
			if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
			if (OP == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
			if (OP == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
			if (OP == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
			if (OP == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(cp));
			if (OP == 8) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
			if (OP == 9) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
			if (OP == 10) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(iv[i]) : "v"(iv[(i + 1) & 7]));
			if (OP == 11) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(iv[i]) : "v"(iv[(i + 1) & 7]));
			if (OP == 12) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(iv[i]) : "v"(a[i]));
			if (OP == 13) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(iv[i]));
			if (OP == 14) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(cd));
			if (OP == 15) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(iv[i]) : "v"(iv[(i + 1) & 7]));
			if (OP == 16) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(iv[i]));
			if (OP == 17) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(iv[i]) : "v"(iv[(i + 1) & 7]));
		}
	}
	float s = 0;
	for (int i = 0; i < 8; ++i) {
		s += a[i] + (float)d[i] + p[i].x + p[i].y + (float)iv[i];
	}
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, float *d_out)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int wgs = 256 * 8;             // 8 workgroups of 4 waves per CU: 8 waves per SIMD
	hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, d_out, 1.0f, 16);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, d_out, 1.0f, ITER);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double insts_per_simd = (double)wgs * 4 / (256 * 4) * ITER * 8;      // wave instructions per SIMD
	printf("%-16s %8.3f ms  %6.2f ns per wave-instruction per SIMD (= %5.2f cycles at 2.4 GHz)\n", name, ms,
	       ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}

int main()
{
	float *d_out;
	hipMalloc(&d_out, 256 * 8 * 256 * 4);
	run<0>("v_mul_f32", d_out);
	run<1>("v_add_f32", d_out);
	run<2>("v_pk_mul_f32", d_out);
	run<3>("v_pk_add_f32", d_out);
	run<6>("v_fma_f32", d_out);
	run<7>("v_pk_fma_f32", d_out);
	run<4>("v_mul_f64", d_out);
	run<5>("v_add_f64", d_out);
	run<14>("v_fma_f64", d_out);
	run<8>("v_cvt_f64_f32", d_out);
	run<9>("v_cvt_f32_f64", d_out);
	run<10>("v_mul_lo_u32", d_out);
	run<11>("v_mad_u32_u24", d_out);
	run<12>("v_cvt_i32_f32", d_out);
	run<13>("v_cvt_f32_i32", d_out);
	run<15>("v_lshl_add_u32", d_out);
	run<16>("v_cvt_f32_ubyte0", d_out);
	run<17>("v_bfi_b32", d_out);
	return 0;
}
