// jpeg_amd/csrc/huffsync.hip -- finding the true parse of a Huffman stream in parallel.
//
// The unstuffed stream of a frame is cut into sub-sequences of JH_SUBSEQ_BITS bits.  "Chain t" is
// the decoder started at the first bit of sub-sequence t with the guess (DC code of block 0 next).
// Chain 0 is the true parse.  Huffman streams self-synchronise, so chain t soon coincides with the
// true parse -- but a JPEG decoder's state is (bit position, block-in-MCU, zig-zag index) and the
// block phase only locks by chance, so a few chains stay wrong for a long time.  Waiting for EVERY
// chain (the obvious protocol, profiles/r01_a) costs ~255 sparse steps per workgroup.  Instead:
//
//   k_huff_chains   (one workgroup = JH_SEQ_OWN own + JH_CHAIN_MAX ghost sub-sequences)
//     step 0   every chain decodes its own sub-sequence;
//     step d   every chain still alive decodes sub-sequence t + d and looks its exit state up in a
//              small per-sub-sequence directory (LDS) of the exit states YOUNGER chains had there.
//              Equal state = identical future: the chain records a link to the younger chain and
//              stops.  Chains are compacted every step, so sparse steps cost few wave-instructions;
//              after JH_CHAIN_MAX steps the survivors (wrong, irrelevant parses) are cut.
//     The links form a forest.  The true parse is a PATH in it: on chain r until r merges, then on
//     the chain r merged into, ...  The path from the workgroup's first chain (the "canonical"
//     path) is marked by pointer doubling (8 rounds) and every sub-sequence j gets the record of the
//     path chain that covers it: its exit state, blocks completed and DC sums, decoded from the
//     TRUE entry state.  For the first workgroup of a frame the canonical path is the true parse.
//   k_huff_fix      (one thread per workgroup boundary)
//     walks the true path from the previous workgroup's exit chain until it joins this workgroup's
//     canonical path (a few hops), rewriting the records of those first sub-sequences.  If the true
//     path reaches a cut chain it decodes on by itself until it meets a stored chain again, which
//     guarantees progress for ANY stream.  A walk that does not join changes that workgroup's exit;
//     rounds repeat (full walks) until nothing changes -- in practice the first round is final.
//
// Restart markers need nothing extra: crossing a restart boundary puts every chain into the same
// state, so all chains merge there at the latest.
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "kernels.hpp"
#include "huffcore.hpp"

#define KMAX JH_CHAIN_MAX
#define OWN JH_SEQ_OWN
#define DIRN 4                          // distinct exit states remembered per sub-sequence
#define NONE16 0xffffu
#define NONE32 0xffffffffu

// parse state + restart segment as one comparable word (segment modulo 2^16: two chains at the same
// bit position can only differ by the boundaries processed AT that position)
__device__ __forceinline__ unsigned long long chain_key(unsigned long long state, uint32_t seg)
{
	return (state & STATE_SYNC_MASK) | ((unsigned long long)(seg & 0xffffu) << 48);
}

__device__ __forceinline__ void state_from(DecState &st, unsigned long long state, uint32_t seg)
{
	st.pos = (uint32_t)(state & 0xffffffffu);
	st.c = (int)((state >> 32) & 0xff);
	st.z = (int)((state >> 40) & 0xff);
	st.seg = seg;
	st.reset = 0;
	st.nb = 0;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		st.dc[k] = 0;
	}
}

// append `value` of the threads with keep == true to list[], densely (order irrelevant)
__device__ __forceinline__ void compact_append(bool keep, uint16_t value, uint16_t *list, int *count)
{
	const unsigned long long m = __ballot(keep);
	const int lane = threadIdx.x & 63;
	int base = 0;
	if (lane == 0 && m) {
		base = atomicAdd(count, __popcll(m));
	}
	base = __shfl(base, 0, 64);
	if (keep) {
		list[base + __popcll(m & ((1ull << lane) - 1ull))] = value;
	}
}

__global__ __launch_bounds__(256) void k_huff_chains(const HuffArgs a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	unsigned char *luts = smem;
	uint32_t *sw = reinterpret_cast<uint32_t *>(smem + sizeof(JhHuffLut) * a.w.max_luts);
	__shared__ DecTables tb;
	__shared__ int dcs[JH_MAX_COMP][256];
	__shared__ unsigned long long dkey[256][DIRN];
	__shared__ uint16_t downer[256][DIRN];
	__shared__ uint8_t dcnt[256];
	__shared__ unsigned long long cstate[256];
	__shared__ uint32_t cseg[256];
	__shared__ uint16_t alive[2][256];
	__shared__ int nalive[2];
	__shared__ uint16_t succ[256];
	__shared__ uint8_t nst[256];
	__shared__ uint16_t reach[2][256];
	__shared__ uint8_t mark[256];
	__shared__ uint16_t pred[256];
	__shared__ int path_last_cov;
	__shared__ unsigned exit_chain;

	const JhDecodeWork &w = a.w;
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	const JhDecState fs = w.state[f];
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t nsub = (total_bits + JH_SUBSEQ_BITS - 1) / JH_SUBSEQ_BITS;
	const uint32_t j0 = blockIdx.x * (uint32_t)OWN;
	if (j0 >= nsub) {
		return;
	}
	load_tables(w, fr, luts, &tb, a.g);

	const uint8_t *cs = w.cs + fr.cs_off;
	const uint32_t *seg_start = w.seg_start + fr.seg_off;
	const uint32_t nseg = min(fs.nseg, fr.seg_cap - 1);
	const int tid = threadIdx.x;
	const int range_len = (int)min(256u, nsub - j0);      // own + ghost sub-sequences of this workgroup
	JhSubInfo *rec = w.rec + (size_t)(fr.sub_off + j0) * (KMAX + 1);
	LdsStream S;
	S.sw = sw;
	S.wbase = j0 * (JH_SUBSEQ_BITS / 32);
	stage_stream(sw, cs, S.wbase, (uint32_t)range_len * (JH_SUBSEQ_BITS / 32) + 4);
	int *mydcs = &dcs[0][tid];

	// ---- step 0: every chain decodes its own sub-sequence -------------------------------------
	if (tid < 2) {
		nalive[tid] = 0;
	}
	dcnt[tid] = 0;
	nst[tid] = 0;
	succ[tid] = NONE16;
	__syncthreads();
	{
		bool keep = false;
		if (tid < range_len) {
			DecState st;
			const uint32_t t = j0 + tid;
			state_from(st, (unsigned long long)t * JH_SUBSEQ_BITS, 0);
			st.seg = nseg > 1 ? find_segment(seg_start, nseg, st.pos >> 3) : 0u;
			decode_span<false>(st, (t + 1) * JH_SUBSEQ_BITS, S, total_bits, seg_start, nseg, luts, tb, a.g.B, mydcs, 256, nullptr, 0, 0);
			if (tid < OWN) {
				store_info(&rec[(size_t)tid * (KMAX + 1)], st);
			}
			const unsigned long long ps = pack_state(st);
			dkey[tid][0] = chain_key(ps, st.seg);
			downer[tid][0] = (uint16_t)tid;
			dcnt[tid] = 1;
			cstate[tid] = ps;
			cseg[tid] = st.seg;
			keep = tid + 1 < range_len;
		}
		compact_append(keep, (uint16_t)tid, alive[0], &nalive[0]);
	}

	// ---- steps 1..KMAX: chains walk on until they coincide with a younger chain -----------------
	int cur = 0;
	for (int d = 1; d <= KMAX; ++d) {
		__syncthreads();
		const int n = nalive[cur];
		if (n == 0) {
			break;
		}
		__syncthreads();
		if (tid == 0) {
			nalive[cur ^ 1] = 0;
		}
		__syncthreads();
		bool keep = false;
		int i = 0;
		if (tid < n) {
			i = alive[cur][tid];
			const int jl = i + d;
			DecState st;
			state_from(st, cstate[i], cseg[i]);
			decode_span<false>(st, (j0 + jl + 1) * JH_SUBSEQ_BITS, S, total_bits, seg_start, nseg, luts, tb, a.g.B, mydcs, 256, nullptr, 0, 0);
			if (i < OWN) {
				store_info(&rec[(size_t)i * (KMAX + 1) + d], st);
			}
			const unsigned long long ps = pack_state(st);
			const unsigned long long key = chain_key(ps, st.seg);
			const int cnt = dcnt[jl];
			int hit = -1;
			for (int e = 0; e < cnt; ++e) {
				if (dkey[jl][e] == key) {
					hit = downer[jl][e];
				}
			}
			if (hit >= 0) {
				succ[i] = (uint16_t)hit;           // same state as a younger chain: identical from here on
				nst[i] = (uint8_t)d;
			} else {
				if (cnt < DIRN) {
					dkey[jl][cnt] = key;
					downer[jl][cnt] = (uint16_t)i;
					dcnt[jl] = (uint8_t)(cnt + 1);
				}
				if (d == KMAX || jl + 1 >= range_len) {
					nst[i] = (uint8_t)d;           // cut: still an unmerged parse at the horizon
				} else {
					keep = true;
					cstate[i] = ps;
					cseg[i] = st.seg;
				}
			}
		}
		compact_append(keep, (uint16_t)i, alive[cur ^ 1], &nalive[cur ^ 1]);
		cur ^= 1;
	}
	__syncthreads();

	// ---- canonical path: chains reachable from this workgroup's first chain (pointer doubling) ----
	reach[0][tid] = succ[tid];
	mark[tid] = tid == 0 ? 1 : 0;
	pred[tid] = NONE16;
	if (tid == 0) {
		path_last_cov = -1;
		exit_chain = NONE32;
	}
	__syncthreads();
	int rb = 0;
	for (int k = 0; k < 8; ++k) {
		const unsigned r = reach[rb][tid];
		if (mark[tid] && r != NONE16) {
			mark[r] = 1;
		}
		reach[rb ^ 1][tid] = r != NONE16 ? reach[rb][r] : (uint16_t)NONE16;
		__syncthreads();
		rb ^= 1;
	}
	if (mark[tid] && succ[tid] != NONE16) {
		pred[succ[tid]] = (uint16_t)tid;
	}
	__syncthreads();
	// chain s covers local sub-sequences [from, to]: from the step after its predecessor merged into it
	const bool onp = mark[tid] && tid < range_len;
	int from = 0, to = -1;
	if (onp) {
		from = tid == 0 ? 0 : (int)pred[tid] + (int)nst[pred[tid]] + 1;
		to = tid + (int)nst[tid];
		if (succ[tid] == NONE16) {
			path_last_cov = to;
		}
	}
	__syncthreads();
	const int need = min(range_len, OWN + 1);               // own sub-sequences + the hand-over one
	const bool broken = path_last_cov < need - 1;
	JhSubInfo *truth = w.sub + fr.sub_off + j0;
	if (onp && !broken) {
		const int hi = min(min(to, OWN - 1), range_len - 1);
		for (int jl = from; jl <= hi; ++jl) {
			truth[jl] = rec[(size_t)tid * (KMAX + 1) + (jl - tid)];
		}
		if (from <= OWN && OWN <= to) {
			exit_chain = j0 + (uint32_t)tid;               // carries the true parse into the next workgroup
		}
	}
	__syncthreads();
	if (tid < OWN && tid < range_len) {
		const uint32_t link = succ[tid] != NONE16 ? (uint32_t)(succ[tid] - tid) : 0u;
		w.meta[fr.sub_off + j0 + tid] = (uint32_t)nst[tid] | (link << 8);
		w.onfrom[fr.sub_off + j0 + tid] = (onp && !broken) ? (int)(j0 + (uint32_t)from) : -1;
	}
	if (tid == 0) {
		w.wg_exit[fr.wg_off + blockIdx.x] = broken ? NONE32 : exit_chain;
	}
}

// ===================================================================================== fix-up walk

__global__ __launch_bounds__(64) void k_huff_fix(const HuffArgs a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	unsigned char *luts = smem;
	__shared__ DecTables tb;
	__shared__ int dcs[JH_MAX_COMP][64];

	const JhDecodeWork &w = a.w;
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	JhDecState &fs = w.state[f];
	if (a.round > 0 && fs.changed[a.round - 1] == 0) {
		return;                            // fixed point reached in an earlier round (changed[] stays 0)
	}
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t nsub = (total_bits + JH_SUBSEQ_BITS - 1) / JH_SUBSEQ_BITS;
	const uint32_t nwg = (nsub + OWN - 1) / OWN;
	if (blockIdx.x * 64u >= nwg) {
		return;
	}
	load_tables(w, fr, luts, &tb, a.g);
	const uint32_t b = blockIdx.x * 64u + threadIdx.x;
	if (b >= nwg) {
		return;
	}
	const uint8_t *cs = w.cs + fr.cs_off;
	const uint32_t *seg_start = w.seg_start + fr.seg_off;
	const uint32_t nseg = min(fs.nseg, fr.seg_cap - 1);
	JhSubInfo *truth = w.sub + fr.sub_off;
	const JhSubInfo *rec = w.rec + (size_t)fr.sub_off * (KMAX + 1);
	const uint32_t *meta = w.meta + fr.sub_off;
	const int *onfrom = w.onfrom + fr.sub_off;
	uint32_t *wg_exit = w.wg_exit + fr.wg_off;
	const bool full = a.round > 0;

	const uint32_t j0 = b * (uint32_t)OWN;
	const uint32_t own_end = min(nsub, j0 + (uint32_t)OWN);
	uint32_t r = b == 0 ? 0u : wg_exit[b - 1];
	if (r == NONE32) {
		atomicAdd(&fs.changed[a.round], 1u);   // the previous workgroup has no exit yet: another round
		return;
	}
	for (uint32_t jn = j0; jn < own_end; ++jn) {
		// the true parse is on chain r for sub-sequence jn
		if (!full && r >= j0 && onfrom[r] >= 0 && (uint32_t)onfrom[r] <= jn) {
			return;                        // joined this workgroup's canonical path: the rest is already right
		}
		const uint32_t m = meta[r];
		const uint32_t steps = m & 0xffu, link = m >> 8;
		const uint32_t d = jn - r;
		if (r <= jn && d <= steps) {
			truth[jn] = rec[(size_t)r * (KMAX + 1) + d];
			if (d == steps && link != 0) {
				r += link;
			}
			continue;
		}
		// chain r was cut before jn: decode on from the true state, then look for a stored chain again
		DecState st;
		if (jn == 0) {
			state_from(st, 0ull, 0);
		} else {
			const JhSubInfo p = truth[jn - 1];
			state_from(st, p.state, p.seg);
		}
		if (st.c >= a.g.B || st.z > 63 || st.seg > nseg) {
			atomicAdd(&fs.changed[a.round], 1u);   // raced with a neighbour's rewrite: redo next round
			return;
		}
		decode_span<false>(st, (jn + 1) * JH_SUBSEQ_BITS, GlobalStream{cs}, total_bits, seg_start, nseg, luts, tb, a.g.B, &dcs[0][threadIdx.x], 64, nullptr, 0, 0);
		store_info(&truth[jn], st);
		const unsigned long long key = chain_key(pack_state(st), st.seg);
		const uint32_t lo = jn > (uint32_t)KMAX ? jn - KMAX : 0u;
		for (uint32_t c = jn + 1; c-- > lo;) {
			const uint32_t mc = meta[c];
			const uint32_t dc = jn - c;
			if (dc <= (mc & 0xffu)) {
				const JhSubInfo x = rec[(size_t)c * (KMAX + 1) + dc];
				if (chain_key(x.state, x.seg) == key) {
					r = (dc == (mc & 0xffu) && (mc >> 8) != 0) ? c + (mc >> 8) : c;
					break;
				}
			}
		}
	}
	if (own_end < nsub && wg_exit[b] != r) {
		wg_exit[b] = r;
		atomicAdd(&fs.changed[a.round], 1u);
	}
}

int jh_launch_huff_sync(const JhDecodeWork &w, const JhGeom &g, hipStream_t s, bool resync_only, const JhTimerHook *th)
{
	HuffArgs a;
	a.w = w;
	a.g = g;
	a.round = 0;
	const size_t lds = sizeof(JhHuffLut) * (size_t)w.max_luts;
	const size_t lds_chains = lds + JH_STREAM_LDS_WORDS(256) * sizeof(uint32_t);
	const unsigned nwg = (w.max_subs + OWN - 1) / OWN;
	if (!resync_only) {
		// static + dynamic LDS of this kernel exceeds the 64 KB default limit
		if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_huff_chains), hipFuncAttributeMaxDynamicSharedMemorySize,
		                        (int)lds_chains) != hipSuccess) {
			return -1;
		}
		JH_T_BEGIN(th, "k_huff_chains", s);
		hipLaunchKernelGGL(k_huff_chains, dim3(nwg, w.nframes), dim3(256), lds_chains, s, a);
		JH_T_END(th, s);
	}
	JH_T_BEGIN(th, "k_huff_fix", s);
	for (int r = 0; r < JH_SYNC_ITERS; ++r) {
		a.round = resync_only ? r + 1 : r;                // a resync continues with full walks
		if (a.round > JH_SYNC_ITERS) {
			break;
		}
		hipLaunchKernelGGL(k_huff_fix, dim3((nwg + 63) / 64, w.nframes), dim3(64), lds, s, a);
	}
	JH_T_END(th, s);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}
