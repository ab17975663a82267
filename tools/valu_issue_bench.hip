// tools/valu_issue_bench.hip -- VALU issue-rate micro-benchmark for gfx950 (MI355X).
//
// Why: the two DP kernels (k_scan, k_align_fwd) are VALU-issue bound, so every "fraction of peak" quoted for them
// needs the issue cost of the instruction class they are made of (packed 16-bit integer VOP3P ops).  This program
// measures, per instruction, the shader clocks one SIMD spends per wave64 instruction:
//     cycles/instruction/SIMD = s_memtime ticks of one wave / (instructions of that wave x waves per SIMD)
// for 1 .. 8 waves per SIMD, every CU busy, 8 independent dependency chains per wave.
// It also prints the effective clock (s_memtime ticks / wall time), which is below the 2.4 GHz peak under load.
//
// Build + run on the GPU box (output kept under profiles/):
//     hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_issue_bench tools/valu_issue_bench.hip && /tmp/valu_issue_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITERS = 2000;        // loop iterations
constexpr int PER_ITER = 64;       // instructions per iteration (8 chains x 8 repeats of the body)

// 2-operand form: op dst, dst, src     3-operand form: op dst, dst, src, src2
#define BODY2(OP) \
	asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" \
	             OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" \
		: "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x));
#define BODY2S(OP, SUF) \
	asm volatile(OP " %0, %0, %8 " SUF "\n" OP " %1, %1, %8 " SUF "\n" OP " %2, %2, %8 " SUF "\n" OP " %3, %3, %8 " SUF "\n" \
	             OP " %4, %4, %8 " SUF "\n" OP " %5, %5, %8 " SUF "\n" OP " %6, %6, %8 " SUF "\n" OP " %7, %7, %8 " SUF "\n" \
		: "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x));
#define BODY3(OP) \
	asm volatile(OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" \
	             OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n" \
		: "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x), "v"(y));
#define BODYDPP() \
	asm volatile("v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
	             "v_mov_b32_dpp %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
	             "v_mov_b32_dpp %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
	             "v_mov_b32_dpp %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
		: "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x));

// the instruction mix of one DP row pair of k_scan: perm, add clamp, 3 x max, 3 x sat-sub, 2 x max (10 ops; here 8 of them per body)
#define BODYMIX() \
	asm volatile("v_perm_b32 %0, %0, %8, %9\nv_pk_add_i16 %1, %1, %8 clamp\nv_pk_max_i16 %2, %2, %8\nv_pk_max_i16 %3, %3, %8\n" \
	             "v_pk_sub_u16 %4, %4, %8 clamp\nv_pk_sub_u16 %5, %5, %8 clamp\nv_pk_max_u16 %6, %6, %8\nv_pk_max_u16 %7, %7, %8\n" \
		: "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x), "v"(y));
// half packed, half plain: does a full-rate op fill the slots a packed op leaves?
#define BODYHALF() \
	asm volatile("v_pk_max_i16 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_pk_max_i16 %2, %2, %8\nv_add_u32 %3, %3, %8\n" \
	             "v_pk_max_i16 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_pk_max_i16 %6, %6, %8\nv_add_u32 %7, %7, %8\n" \
		: "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x));

#define KERNEL(NAME, BODY) \
__global__ void __launch_bounds__(1024) NAME(uint64_t* ticks, uint32_t* sink, uint32_t seed) \
{ \
	uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, r6 = r0 * 17u, r7 = r0 * 19u; \
	uint32_t x = seed ^ 0x00010001u, y = 0x05040100u; (void)y; \
	__syncthreads(); \
	const uint64_t t0 = __builtin_amdgcn_s_memtime(); \
	for (int it = 0; it < ITERS; it++) { BODY BODY BODY BODY BODY BODY BODY BODY } \
	const uint64_t t1 = __builtin_amdgcn_s_memtime(); \
	if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
	if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 0x12345u) sink[0] = r0; \
}

KERNEL(k_add_u32, BODY2("v_add_u32"))
KERNEL(k_sub_u32, BODY2("v_sub_u32"))
KERNEL(k_and_b32, BODY2("v_and_b32"))
KERNEL(k_max_i32, BODY2("v_max_i32"))
KERNEL(k_max_u32, BODY2("v_max_u32"))
KERNEL(k_max3_i32, BODY3("v_max3_i32"))
KERNEL(k_perm_b32, BODY3("v_perm_b32"))
KERNEL(k_alignbit, BODY3("v_alignbit_b32"))
KERNEL(k_mov_dpp, BODYDPP())
KERNEL(k_max_f32, BODY2("v_max_f32"))
KERNEL(k_add_f32, BODY2("v_add_f32"))
KERNEL(k_max3_f32, BODY3("v_max3_f32"))
KERNEL(k_max_f16, BODY2("v_max_f16"))
KERNEL(k_max_i16, BODY2("v_max_i16"))
KERNEL(k_add_u16, BODY2("v_add_u16"))
KERNEL(k_pk_add_i16, BODY2("v_pk_add_i16"))
KERNEL(k_pk_add_i16_clamp, BODY2S("v_pk_add_i16", "clamp"))
KERNEL(k_pk_sub_u16_clamp, BODY2S("v_pk_sub_u16", "clamp"))
KERNEL(k_pk_max_i16, BODY2("v_pk_max_i16"))
KERNEL(k_pk_max_u16, BODY2("v_pk_max_u16"))
KERNEL(k_pk_min_u16, BODY2("v_pk_min_u16"))
KERNEL(k_pk_max_f16, BODY2("v_pk_max_f16"))
KERNEL(k_pk_add_f16, BODY2("v_pk_add_f16"))
KERNEL(k_mix_row, BODYMIX())
KERNEL(k_mix_half, BODYHALF())

typedef void (*kern_t)(uint64_t*, uint32_t*, uint32_t);
struct Op { const char* name; kern_t k; };

int main()
{
	const Op ops[] = {
		{ "v_add_u32", k_add_u32 }, { "v_sub_u32", k_sub_u32 }, { "v_and_b32", k_and_b32 }, { "v_max_i32", k_max_i32 }, { "v_max_u32", k_max_u32 },
		{ "v_max3_i32", k_max3_i32 }, { "v_perm_b32", k_perm_b32 }, { "v_alignbit_b32", k_alignbit }, { "v_mov_b32 dpp wave_shr:1", k_mov_dpp },
		{ "v_max_f32", k_max_f32 }, { "v_add_f32", k_add_f32 }, { "v_max3_f32", k_max3_f32 }, { "v_max_f16", k_max_f16 }, { "v_max_i16", k_max_i16 },
		{ "v_add_u16", k_add_u16 },
		{ "v_pk_add_i16", k_pk_add_i16 }, { "v_pk_add_i16 clamp", k_pk_add_i16_clamp }, { "v_pk_sub_u16 clamp", k_pk_sub_u16_clamp },
		{ "v_pk_max_i16", k_pk_max_i16 }, { "v_pk_max_u16", k_pk_max_u16 }, { "v_pk_min_u16", k_pk_min_u16 }, { "v_pk_max_f16", k_pk_max_f16 },
		{ "v_pk_add_f16", k_pk_add_f16 }, { "DP row mix (perm,add,max,sub)", k_mix_row }, { "v_pk_max_i16 / v_add_u32 1:1", k_mix_half },
	};
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("# device %s (%s), %d CUs, clockRate %d kHz\n", prop.name, prop.gcnArchName, cus, prop.clockRate);
	printf("# %d x %d instructions per wave, 8 independent chains, W waves per SIMD = W workgroups of 256 threads per CU (grid = W x CUs)\n",
		ITERS, PER_ITER);
	printf("# cyc = median over waves of s_memtime ticks / (instructions per wave x W); GHz = ticks / wall time of the kernel\n");
	printf("%-30s %8s %8s %8s %8s %8s %8s %8s %8s\n", "instruction", "cyc W=1", "cyc W=2", "cyc W=3", "cyc W=4", "cyc W=5", "cyc W=6", "cyc W=8", "GHz W=4");
	uint64_t* ticks = nullptr; uint32_t* sink = nullptr;
	CHECK(hipMalloc(&ticks, sizeof(uint64_t) * cus * 64));
	CHECK(hipMalloc(&sink, 64));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	const double instr = (double)ITERS * PER_ITER;
	for (const Op& op : ops) {
		double cyc[7] = { 0, 0, 0, 0, 0, 0, 0 }, wall[7] = { 0, 0, 0, 0, 0, 0, 0 }, ghz = 0;
		const int Ws[7] = { 1, 2, 3, 4, 5, 6, 8 };
		for (int wi = 0; wi < 7; wi++) {
			const int W = Ws[wi];
			// 256-thread workgroups (4 waves: one per SIMD, the shape of k_scan), W of them per CU
			const int threads = 256, blocks = W * cus;
			hipLaunchKernelGGL(op.k, dim3(blocks), dim3(threads), 0, 0, ticks, sink, 1u);     // warm-up
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(e0, 0));
			hipLaunchKernelGGL(op.k, dim3(blocks), dim3(threads), 0, 0, ticks, sink, 2u);
			CHECK(hipEventRecord(e1, 0));
			CHECK(hipDeviceSynchronize());
			float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
			std::vector<uint64_t> h((size_t)blocks * (threads / 64));
			CHECK(hipMemcpy(h.data(), ticks, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost));
			std::sort(h.begin(), h.end());
			const double med = (double)h[h.size() / 2];
			cyc[wi] = med / (instr * W);
			// the same from the wall clock: ns of one SIMD per wave64 instruction (independent of the clock estimate)
			wall[wi] = (double)ms * 1e6 / (instr * W);
			if (W == 4) ghz = med / (ms * 1e-3) / 1e9;
		}
		printf("%-30s %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f\n", op.name, cyc[0], cyc[1], cyc[2], cyc[3], cyc[4], cyc[5], cyc[6], ghz);
		printf("%-30s %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f\n", "   ns per instruction (wall)", wall[0], wall[1], wall[2], wall[3], wall[4], wall[5], wall[6]);
	}
	printf("# reading: ~2 cyc = full rate (one wave64 instruction per 2 clocks and SIMD, MI355X_MICROARCH.md).\n");
	return 0;
}
