// jpeg_amd/csrc/huffsync.hip -- finding the true parse of a Huffman stream in parallel.
//
// The unstuffed stream of a frame is cut into sub-sequences of 2^sub_shift bits (2048 as a rule).  "Chain t" is
// the decoder started at the first bit of sub-sequence t with the guess (DC code of block 0 next).
// Chain 0 is the true parse.  Huffman streams self-synchronise, so chain t soon coincides with the
// true parse -- but a JPEG decoder's state is (bit position, block-in-MCU, zig-zag index) and the
// block phase only locks by chance, so a few chains stay wrong for a long time.  Therefore:
//
//   k_huff_chains   (one workgroup = JH_SEQ_OWN own + JH_CHAIN_MAX ghost sub-sequences, stream in LDS)
//     step 0   every chain decodes its own sub-sequence;
//     step d   every chain still alive decodes sub-sequence t + d and looks its exit state up in a
//              small per-sub-sequence directory (LDS) of the exit states YOUNGER chains had there.
//              Equal state = identical future: the chain records a link to the younger chain and
//              stops.  A chain that runs into garbage (no code matches, no restart segment left) is
//              linked to the chain that starts at the next sub-sequence ("restart link"): fine for a
//              guess, flagged so that it is never trusted on the true parse.  Alive chains are
//              compacted every step; after JH_CHAIN_MAX steps the survivors are cut.
//     The links form a forest; the true parse is a PATH in it.  The workgroup marks the path from
//     its first chain (pointer doubling) and emits, for every sub-sequence on it from the
//     JH_CHAIN_MAX-th on, the record decoded from the path's state, the exclusive prefix of blocks /
//     DC sums inside the workgroup and the workgroup's totals -- plus the state it ASSUMED at its first
//     emitted sub-sequence.  The first JH_CHAIN_MAX sub-sequences are emitted by the PREVIOUS workgroup,
//     whose path has long joined the true parse there.
//   k_huff_walk     (one wave per workgroup, a few rounds; normally returns at once)  If the state a
//     workgroup assumed is not what its predecessor emitted, or its path had a hole (a chain cut
//     before it merged: dense high-quality data synchronises slowly), the wave walks the true parse
//     from the last trustworthy record -- following stored chains where one carries it, decoding by
//     itself where none does -- until it falls in with the records already there.
//   k_huff_finish   checks, workgroup by workgroup from the frame's first one (whose first chain IS the
//     true parse), that every assumed entry state equals the predecessor's last record: that proves
//     every emitted record is on the true parse.  Then the exclusive scan of the workgroup totals (one
//     wave per frame).  Frames that fail the check, or whose walks gave up, are walked serially by one
//     thread instead: slow, but right for ANY stream (corrupt ones included: the reference skips to
//     the end of the segment at undecodable data, decoder.c:376-381).
//
// Restart markers need nothing extra: crossing a restart boundary puts every chain into the same
// state, so all chains merge there at the latest.
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "kernels.hpp"
#include "huffcore.hpp"

#define KMAX JH_CHAIN_MAX
#define OWN JH_SEQ_OWN
#define DIRN 2                          // distinct exit states remembered per sub-sequence (4 buys nothing measurable
                                        // and costs the LDS of a fifth workgroup per CU; 1 would make room for a sixth,
                                        // but the chains that then miss each other cost more: 1.46 against 1.41 ms)
#define NONE16 0xffffu
#define META_RESTART 0x80000000u
#define JH_WALK_ROUNDS 12                // a walk that changes a workgroup's last record sends its successor round again

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

// ---- the segmented sum of (blocks, DC sums): a restart boundary cuts the DC sums -----------------------

struct Agg {
	int nb, rs;
	int dc[JH_MAX_COMP];
};

__device__ __forceinline__ Agg agg_identity()
{
	Agg a;
	a.nb = 0;
	a.rs = 0;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		a.dc[k] = 0;
	}
	return a;
}

__device__ __forceinline__ Agg agg_combine(const Agg &a, const Agg &b)      // a comes first
{
	Agg r;
	r.nb = a.nb + b.nb;
	r.rs = a.rs | b.rs;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		r.dc[k] = b.rs ? b.dc[k] : a.dc[k] + b.dc[k];
	}
	return r;
}

__device__ __forceinline__ Agg agg_shfl_up(const Agg &a, int o)
{
	Agg r;
	r.nb = __shfl_up(a.nb, o, 64);
	r.rs = __shfl_up(a.rs, o, 64);
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		r.dc[k] = __shfl_up(a.dc[k], o, 64);
	}
	return r;
}

// inclusive / exclusive scan over the lanes of a wave, lane order
__device__ __forceinline__ void wave_seg_scan(const Agg &v, Agg &incl, Agg &excl)
{
	const int lane = threadIdx.x & 63;
	Agg x = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const Agg y = agg_shfl_up(x, o);
		if (lane >= o) {
			x = agg_combine(y, x);
		}
	}
	incl = x;
	const Agg e = agg_shfl_up(x, 1);
	excl = lane == 0 ? agg_identity() : e;
}

// over the threads of a 256-thread workgroup, thread order; scratch = 4 entries of LDS
__device__ __forceinline__ void block_seg_scan(const Agg &v, Agg *scratch, Agg &incl, Agg &excl)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	Agg wi, we;
	wave_seg_scan(v, wi, we);
	if (lane == 63) {
		scratch[wv] = wi;
	}
	__syncthreads();
	Agg pre = agg_identity();
	for (int i = 0; i < wv; ++i) {
		pre = agg_combine(pre, scratch[i]);
	}
	incl = agg_combine(pre, wi);
	excl = agg_combine(pre, we);
	__syncthreads();
}

__device__ __forceinline__ Agg agg_of(const JhSubInfo &r)
{
	Agg a;
	a.nb = r.nb;
	a.rs = (int)((r.state >> 48) & 1);
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		a.dc[k] = r.dc[k];
	}
	return a;
}

// ---- path selection -------------------------------------------------------------------------------------

#define REC_INVALID (1ull << 63)        // JhSubInfo.state: no record (the workgroup's path does not reach here)

struct PathLds {                      // written while the chains walk, read by the emission
	uint16_t succ[256];               // chain this chain merged into (local index), NONE16 = none
	uint8_t nst[256];                 // steps walked: the chain covers local sub-sequences [i, i + nst]
	uint8_t dflag[256];               // its link is a restart link (garbage before it)
	Agg scratch[4];
	int gap;
};

struct PathWork {                     // emission only: lives where the arrays of the chain steps were
	JhSubInfo trec[256];              // records of the marked path, by local sub-sequence
	uint16_t reach[2][256];
	uint8_t mark[256];
	uint16_t pred[256];
	uint8_t cov[256];
};

// marks the chains on the path that starts with chain `seed`; pred[] = predecessor on the path.
// A restart link that ends at or after local sub-sequence `trust` is not followed: from there on the
// path claims to be the true parse, and a fresh guess is not.
__device__ __forceinline__ void mark_path(const PathLds &L, PathWork &W, int seed, int trust)
{
	const int tid = threadIdx.x;
	const bool cut = L.dflag[tid] && tid + (int)L.nst[tid] >= trust;
	W.reach[0][tid] = cut ? (uint16_t)NONE16 : L.succ[tid];
	W.mark[tid] = tid == seed ? 1 : 0;
	W.pred[tid] = NONE16;
	__syncthreads();
	int rb = 0;
	for (int k = 0; k < 8; ++k) {
		const unsigned r = W.reach[rb][tid];
		if (W.mark[tid] && r != NONE16) {
			W.mark[r] = 1;
		}
		W.reach[rb ^ 1][tid] = r != NONE16 ? W.reach[rb][r] : (uint16_t)NONE16;
		__syncthreads();
		rb ^= 1;
	}
	if (W.mark[tid] && L.succ[tid] != NONE16 && !cut) {
		W.pred[L.succ[tid]] = (uint16_t)tid;
	}
	__syncthreads();
}

// entry record of local sub-sequence jl of an emitted range from the exclusive prefix over the range
__device__ __forceinline__ JhSubStart make_start(const Agg &excl, unsigned long long prev_state, uint32_t prev_seg)
{
	JhSubStart o;
	o.state = (prev_state & STATE_SYNC_MASK) | ((unsigned long long)(excl.rs & 1) << 48);
	o.seg = prev_seg;
	o.blk = excl.nb;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		o.pred[k] = excl.dc[k];
	}
	return o;
}

__device__ __forceinline__ void store_agg(JhWgAgg *dst, const Agg &t, uint32_t extra_flags)
{
	JhWgAgg o;
	o.nb = t.nb;
	o.flags = (t.rs ? (uint32_t)JH_AGG_RESET : 0u) | extra_flags;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		o.dc[k] = t.dc[k];
	}
	o.next_seed = JH_SEED_NONE;
	o.pad = 0;
	*dst = o;
}

// =================================================================================== chains
//
// Four kernels.  k_huff_chains: steps 0 and 1 of every chain, one thread per chain, no exchange between threads
// (dense waves, tables and DC sums are all the LDS it needs).  k_huff_links: which chains coincide with their
// younger neighbour after step 1 (nine in ten), which ran into garbage, which go on -- those are listed.
// k_huff_survivors: one thread per listed chain walks on, up to JH_CHAIN_MAX steps, comparing its exit state of
// every sub-sequence jl with the two states steps 0 and 1 left there (chain jl's own, chain jl-1's); dense
// waves again.  k_huff_emit: the path of every workgroup range and its records.
// (One kernel did all of this before, with the survivors' steps as sparse rounds between workgroup barriers:
// those rounds took 29 % of its time for an eighth of the work, while the LDS of the whole workgroup stayed taken.)

// The chain met bits no code matches before the stream was over: it ran off the end of the data (no restart
// segment left), or skipped to the start of the next restart segment as read_ecs() does (decoder.c:376-381) and
// that lies behind the whole of the next sub-sequence.  For a guess that is the end of its use -- the next chain's
// guess takes over (restart link) -- and cheap to say: left alone, such a chain idles through its sixteen steps
// towards that segment, links to nothing, and a workgroup range whose path starts with it has to be walked again
// (with optimised tables, where many codes do not exist, one chain in a hundred is of this kind).
__device__ __forceinline__ bool key_garbage(unsigned long long key, uint32_t total_bits, uint32_t next_sub, uint32_t nsub,
                                            uint32_t sub_shift)
{
	const uint32_t pos = (uint32_t)(key & 0xffffffffu);    // chain_key() and JhSubInfo.state alike
	return next_sub < nsub && (pos >= total_bits || pos >= ((next_sub + 1u) << sub_shift));
}

template <bool SEGMENTED>
__global__ __launch_bounds__(256) void k_huff_chains(const HuffArgs a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	unsigned char *luts = smem;
	__shared__ DecTables tb;
	__shared__ int dcs[JH_MAX_COMP][256];

	const JhDecodeWork &w = a.w;
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	const JhDecState fs = w.state[f];
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t sub_shift = w.sub_shift;
	const uint32_t nsub = (total_bits + (1u << sub_shift) - 1u) >> sub_shift;
	if (blockIdx.x * 256u >= nsub) {
		return;
	}
	load_tables(w, fr, luts, &tb, a.g, 256 * 4);              // ends with a barrier, the only one
	const int tid = threadIdx.x;
	const uint32_t t = blockIdx.x * 256u + tid;
	if (t >= nsub) {
		return;
	}
	const uint8_t *cs = w.cs + fr.cs_off;
	const uint32_t *seg_start = w.seg_start + fr.seg_off;
	const uint32_t nseg = min(fs.nseg, fr.seg_cap - 1);
	JhSubInfo *rec = w.rec + (size_t)(fr.sub_off + t) * (KMAX + 1);
	const RegStream S{cs};
	int *mydcs = &dcs[0][tid];

	// step 0: the chain's own sub-sequence, from the guess
	DecState st;
	state_from(st, (unsigned long long)t << sub_shift, 0);
	st.seg = nseg > 1 ? find_segment(seg_start, nseg, st.pos >> 3) : 0u;
	decode_span<SEGMENTED>(st, ((t + 1) << sub_shift), S, total_bits, seg_start, nseg, luts, tb, a.g.B, mydcs, 256, nullptr, 0, 0);
	store_info(&rec[0], st);
	// the exit states once more, packed: what k_huff_links and k_huff_survivors compare (8 bytes instead of a line)
	unsigned long long *key01 = w.key01 + (size_t)(fr.sub_off + t) * 2;
	key01[0] = chain_key(pack_state(st), st.seg);
	if (st.pos >= total_bits || st.pos >= ((t + 2) << sub_shift) || t + 1 >= nsub) {
		return;                            // garbage (key_garbage: k_huff_links restarts with the next chain's guess), or the last one
	}
	// step 1: the next sub-sequence from where step 0 ended (sums counted anew)
	const unsigned long long ps = pack_state(st);
	const uint32_t sg = st.seg;
	state_from(st, ps, sg);
	decode_span<SEGMENTED>(st, ((t + 2) << sub_shift), S, total_bits, seg_start, nseg, luts, tb, a.g.B, mydcs, 256, nullptr, 0, 0);
	store_info(&rec[1], st);
	key01[1] = chain_key(pack_state(st), st.seg);
}

__global__ __launch_bounds__(256) void k_huff_links(const HuffArgs a)
{
	const JhDecodeWork &w = a.w;
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	const JhDecState fs = w.state[f];
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t sub_shift = w.sub_shift;
	const uint32_t nsub = (total_bits + (1u << sub_shift) - 1u) >> sub_shift;
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (blockIdx.x * 256u >= nsub) {
		return;
	}
	bool survive = false;
	if (t < nsub) {
		const unsigned long long *key01 = w.key01 + (size_t)(fr.sub_off + t) * 2;
		const unsigned long long k0 = key01[0];
		uint32_t meta;                     // steps walked | link distance << 8 | restart link
		if (key_garbage(k0, total_bits, t + 1, nsub, sub_shift)) {
			meta = 0u | (1u << 8) | META_RESTART;
		} else if (t + 1 >= nsub) {
			meta = 0u;
		} else {
			const unsigned long long k1 = key01[1];
			const unsigned long long n0 = key01[2];            // the younger neighbour after its step 0
			if (key_garbage(k1, total_bits, t + 2, nsub, sub_shift)) {
				meta = 1u | (2u << 8) | META_RESTART;
			} else if (!key_garbage(n0, total_bits, t + 2, nsub, sub_shift) && k1 == n0) {
				meta = 1u | (1u << 8);                     // same state as the younger chain: identical from here on
			} else {
				meta = 1u;
				survive = KMAX > 1 && t + 2 < nsub;
			}
		}
		w.meta[fr.sub_off + t] = meta;     // a survivor's is written again when it stops
	}
	const unsigned long long m = __ballot(survive);
	if (m) {
		const int lane = threadIdx.x & 63;
		uint32_t base = 0;
		if (lane == (int)__ffsll((long long)m) - 1) {
			base = atomicAdd(&w.state[f].nsurv, (uint32_t)__popcll(m));
		}
		base = __shfl(base, (int)__ffsll((long long)m) - 1, 64);
		if (survive) {
			w.surv[fr.sub_off + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = t;
		}
	}
}

template <bool SEGMENTED>
__global__ __launch_bounds__(256) void k_huff_survivors(const HuffArgs a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	unsigned char *luts = smem;
	__shared__ DecTables tb;
	__shared__ int dcs[JH_MAX_COMP][256];

	const JhDecodeWork &w = a.w;
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	const JhDecState fs = w.state[f];
	const uint32_t nsurv = fs.nsurv;
	if (blockIdx.x * 256u >= nsurv) {
		return;
	}
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t sub_shift = w.sub_shift;
	const uint32_t nsub = (total_bits + (1u << sub_shift) - 1u) >> sub_shift;
	load_tables(w, fr, luts, &tb, a.g, 256 * 4);
	const int tid = threadIdx.x;
	if (blockIdx.x * 256u + tid >= nsurv) {
		return;
	}
	const uint32_t i = w.surv[fr.sub_off + blockIdx.x * 256u + tid];
	const uint8_t *cs = w.cs + fr.cs_off;
	const uint32_t *seg_start = w.seg_start + fr.seg_off;
	const uint32_t nseg = min(fs.nseg, fr.seg_cap - 1);
	JhSubInfo *rec = w.rec + (size_t)(fr.sub_off + i) * (KMAX + 1);
	const RegStream S{cs};
	int *mydcs = &dcs[0][tid];

	unsigned long long ps = rec[1].state;
	uint32_t sg = rec[1].seg;
	uint32_t meta = 1u;
	for (uint32_t d = 2; d <= KMAX; ++d) {
		const uint32_t jl = i + d;         // < nsub
		DecState st;
		state_from(st, ps, sg);
		decode_span<SEGMENTED>(st, ((jl + 1) << sub_shift), S, total_bits, seg_start, nseg, luts, tb, a.g.B, mydcs, 256, nullptr, 0, 0);
		store_info(&rec[d], st);
		meta = d;
		if ((st.pos >= total_bits || st.pos >= ((jl + 2u) << sub_shift)) && jl + 1 < nsub) {
			meta |= ((d + 1u) << 8) | META_RESTART;        // garbage (key_garbage): restart with the next chain's guess
			break;
		}
		ps = pack_state(st);
		sg = st.seg;
		const unsigned long long key = chain_key(ps, sg);
		// Younger chains that were here in steps 0 and 1: chain jl (its own sub-sequence) and chain jl - 1.  Equal
		// state = identical future.  Either of them has the record of sub-sequence jl + 1 if there is one: chain
		// jl from its step 1; chain jl - 1, which did not coincide with chain jl or this chain would too, as a survivor.
		const unsigned long long *kj = w.key01 + (size_t)(fr.sub_off + jl) * 2;
		const unsigned long long o0 = kj[0], p0 = kj[-2];
		uint32_t hit = 0;
		if (!key_garbage(p0, total_bits, jl, nsub, sub_shift)) {
			const unsigned long long p1 = kj[-1];
			if (!key_garbage(p1, total_bits, jl + 1, nsub, sub_shift) && p1 == key) {
				hit = d - 1u;
			}
		}
		if (!key_garbage(o0, total_bits, jl + 1, nsub, sub_shift) && o0 == key) {
			hit = d;
		}
		if (hit) {
			meta |= hit << 8;
			break;
		}
		if (jl + 1 >= nsub) {
			break;
		}
	}
	w.meta[fr.sub_off + i] = meta;
}

// One workgroup per range of JH_SEQ_OWN own + JH_CHAIN_MAX ghost sub-sequences: the path from the range's first
// chain through the links, and the records along it.
__global__ __launch_bounds__(256) void k_huff_emit(const HuffArgs a)
{
	__shared__ __align__(16) PathWork W;
	__shared__ PathLds P;

	const JhDecodeWork &w = a.w;
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	const JhDecState fs = w.state[f];
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t sub_shift = w.sub_shift;
	const uint32_t nsub = (total_bits + (1u << sub_shift) - 1u) >> sub_shift;
	const uint32_t b = blockIdx.x;
	if (nsub == 0 || b >= jh_num_chain_wgs(nsub)) {
		return;
	}
	const uint32_t j0 = b * (uint32_t)OWN;
	const int tid = threadIdx.x;
	const int range_len = (int)min(256u, nsub - j0);      // own + ghost sub-sequences of this workgroup
	const JhSubInfo *rec = w.rec + (size_t)(fr.sub_off + j0) * (KMAX + 1);

	// links inside the range (a chain that leaves it has none here: the next range sees it whole)
	{
		uint32_t nst = 0, link = 0, flag = 0;
		if (tid < range_len) {
			const uint32_t m = w.meta[fr.sub_off + j0 + tid];
			nst = m & 0xffu;
			link = (m >> 8) & 0xffu;
			flag = m & META_RESTART;
		}
		P.nst[tid] = (uint8_t)nst;
		P.succ[tid] = (link != 0 && tid + (int)link < range_len) ? (uint16_t)(tid + link) : (uint16_t)NONE16;
		P.dflag[tid] = flag ? 1 : 0;
	}

	// ---- emit from the path of the first chain ---------------------------------------------------
	const int lo = b == 0 ? 0 : KMAX, hi = range_len - 1;
	if (tid == 0) {
		P.gap = 0;
	}
	W.cov[tid] = 0;
	__syncthreads();
	mark_path(P, W, 0, lo - 1);
	if (W.mark[tid] && tid < range_len) {
		const int from = tid == 0 ? 0 : (int)W.pred[tid] + (int)P.nst[W.pred[tid]] + 1;
		const int to = tid + (int)P.nst[tid];
		for (int jl = from; jl <= min(to, hi); ++jl) {
			W.trec[jl] = rec[(size_t)tid * (KMAX + 1) + (jl - tid)];
			W.cov[jl] = 1;
		}
	}
	__syncthreads();
	const bool mine = tid >= lo && tid <= hi;
	if (mine && !W.cov[tid]) {
		P.gap = 1;                         // a chain on the path was cut, or the true parse ran into garbage
	}
	__syncthreads();
	const bool gap = P.gap != 0;
	JhSubInfo *prec = w.prec + fr.sub_off + j0;
	if (mine) {
		JhSubInfo r = W.trec[tid];
		if (!W.cov[tid]) {
			r.state = REC_INVALID;
		}
		prec[tid] = r;
	}
	Agg v = agg_identity();
	if (mine && !gap) {
		v = agg_of(W.trec[tid]);
	}
	Agg incl, excl;
	block_seg_scan(v, P.scratch, incl, excl);
	if (mine && !gap) {
		const bool first = tid == 0;       // only in the frame's first workgroup: a DC code of block 0 starts at bit 0
		w.start[fr.sub_off + j0 + tid] = make_start(excl, first ? 0ull : W.trec[tid - 1].state, first ? 0u : W.trec[tid - 1].seg);
	}
	if (tid == 255) {
		store_agg(&w.wg_agg[fr.wg_off + b], incl, gap ? (uint32_t)JH_AGG_GAP : 0u);
	}
	if (tid == 0) {
		// the state this workgroup ASSUMED at its first emitted sub-sequence; the walk kernel compares it
		// with what the predecessor says
		unsigned long long key = REC_INVALID;
		if (b > 0 && W.cov[lo - 1]) {
			key = chain_key(W.trec[lo - 1].state, W.trec[lo - 1].seg);
		}
		if ((w.debug & 1) && (b & 1)) {
			key = REC_INVALID;                 // test hook: make every other workgroup take the walk
		}
		w.wg_key[fr.wg_off + b] = key;
		if (w.debug & 2) {
			atomicOr(&w.state[f].status, (uint32_t)JH_ST_SERIAL);
			atomicOr(&w.state[f].serial_cause, 16u);
		}
	}
}

// =================================================================================== walk

struct WalkCtx {
	const JhDecodeWork *w;
	const JhDecFrame *fr;
	const uint8_t *cs;
	const uint32_t *seg_start;
	const JhSubInfo *rec;                 // of the frame's chain 0
	const uint32_t *meta;
	uint32_t total_bits, nseg, nsub, sub_shift;
	int B;
};

// Record of sub-sequence q decoded from the true entry state (T, Tseg).  r = stored chain known to
// carry the true parse into q (0xffffffff: unknown), updated.  Executed by a whole wave: the lanes
// check the candidate chains in parallel; lane 0 decodes by itself where no stored chain helps
// (*selfdec counts those).  The result is returned in every lane.
__device__ __forceinline__ JhSubInfo walk_step(const WalkCtx &c, unsigned long long T, uint32_t Tseg, uint32_t q, uint32_t &r,
                                               const unsigned char *luts, const DecTables &tb, int *dcs, JhSubInfo *bcast,
                                               int *selfdec)
{
	const int lane = threadIdx.x & 63;
	JhSubInfo R;
	if ((uint32_t)(T & 0xffffffffu) >= c.total_bits) {
		// the parse has ended (end of data, or garbage with no restart segment left)
		R.state = T & STATE_SYNC_MASK;
		R.seg = Tseg;
		R.nb = 0;
#pragma unroll
		for (int k = 0; k < JH_MAX_COMP; ++k) {
			R.dc[k] = 0;
		}
		r = 0xffffffffu;
		return R;
	}
	const unsigned long long keyT = chain_key(T, Tseg);
	if (r != 0xffffffffu && q - r > (c.meta[r] & 0xffu)) {
		r = 0xffffffffu;                   // the chain that was being followed has no record of this sub-sequence
	}
	if (r == 0xffffffffu) {
		// a stored chain that ENTERS q in the true state?  chain q itself (lane 0) or an older one (lane d)
		bool hit = false;
		if (lane <= KMAX && (uint32_t)lane <= q) {
			const uint32_t ch = q - (uint32_t)lane;
			const uint32_t steps = c.meta[ch] & 0xffu;
			if ((uint32_t)lane <= steps) {
				if (lane == 0) {
					const uint32_t p0 = (q << c.sub_shift);
					const uint32_t s0 = c.nseg > 1 ? find_segment(c.seg_start, c.nseg, p0 >> 3) : 0u;
					hit = chain_key((unsigned long long)p0, s0) == keyT;
				} else {
					const JhSubInfo x = c.rec[(size_t)ch * (KMAX + 1) + (lane - 1)];
					hit = chain_key(x.state, x.seg) == keyT;
				}
			}
		}
		const unsigned long long m = __ballot(hit);
		if (m) {
			r = q - (uint32_t)(__ffsll((long long)m) - 1);
		}
	}
	if (r != 0xffffffffu) {
		const uint32_t m = c.meta[r];
		const uint32_t steps = m & 0xffu, link = (m >> 8) & 0xffu, d = q - r;
		R = c.rec[(size_t)r * (KMAX + 1) + d];
		if (d == steps) {
			// the chain ends here: follow a true merge, never a restart link
			r = (link != 0 && !(m & META_RESTART)) ? r + link : 0xffffffffu;
		}
		return R;
	}
	if (lane == 0) {
		DecState st;
		state_from(st, T, Tseg);
		if (st.c >= c.B || st.z > 63 || st.seg >= c.nseg) {
			st.pos = c.total_bits;             // cannot happen for states produced by decode_span
		}
		decode_span<true>(st, ((q + 1) << c.sub_shift), RegStream{c.cs}, c.total_bits, c.seg_start, c.nseg, luts, tb, c.B, dcs, 64,
		                   nullptr, 0, 0);
		store_info(bcast, st);
		*selfdec += 1;
	}
	__builtin_amdgcn_wave_barrier();
	__threadfence_block();
	R = *bcast;
	return R;
}

#define WALK_SELF_LIMIT 96              // sub-sequences one wave may decode by itself per round before giving up

// One wave per chains workgroup; does nothing unless the workgroup's assumed entry state differs from
// what its predecessor emitted, or its own path had a hole.  Then it walks the true parse from the
// last trustworthy record -- following stored chains where one carries it, decoding where none does --
// until it falls in with the records already emitted, and redoes the workgroup's prefix sums.
__global__ __launch_bounds__(64) void k_huff_walk(const HuffArgs a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	unsigned char *luts = smem;
	__shared__ DecTables tb;
	__shared__ int dcs[JH_MAX_COMP][64];
	__shared__ JhSubInfo prl[256];
	__shared__ JhSubInfo bcast;
	__shared__ int selfdec;

	const JhDecodeWork &w = a.w;
	const int f = blockIdx.y;
	const uint32_t b = blockIdx.x;
	const JhDecFrame fr = w.frames[f];
	const JhDecState fs = w.state[f];
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t sub_shift = w.sub_shift;
	const uint32_t nsub = (total_bits + (1u << sub_shift) - 1u) >> sub_shift;
	if (nsub == 0 || b >= jh_num_chain_wgs(nsub) || (fs.status & JH_ST_SERIAL)) {
		return;
	}
	const int lane = threadIdx.x;
	const uint32_t j0 = b * (uint32_t)OWN;
	const int range_len = (int)min(256u, nsub - j0);
	const int lo = b == 0 ? 0 : KMAX, hi = range_len - 1;
	JhSubInfo *prec = w.prec + fr.sub_off + j0;
	JhWgAgg *agg = &w.wg_agg[fr.wg_off + b];
	const bool gap = (agg->flags & JH_AGG_GAP) != 0;
	JhSubInfo pe;
	pe.state = 0;
	pe.seg = 0;
	bool mismatch = false;
	if (b > 0) {
		pe = prec[lo - 1];                    // the predecessor's last record = the state this range starts in
		if (pe.state & REC_INVALID) {
			return;                           // the predecessor still has a hole: a later round
		}
		mismatch = chain_key(pe.state, pe.seg) != w.wg_key[fr.wg_off + b];
	}
	if (!mismatch && !gap) {
		return;
	}
	load_tables(w, fr, luts, &tb, a.g, 64 * 4);
	for (int jl = lo + lane; jl <= hi; jl += 64) {
		prl[jl] = prec[jl];
	}
	if (lane == 0) {
		selfdec = 0;
	}
	__syncthreads();

	WalkCtx c;
	c.w = &w;
	c.fr = &fr;
	c.cs = w.cs + fr.cs_off;
	c.seg_start = w.seg_start + fr.seg_off;
	c.rec = w.rec + (size_t)fr.sub_off * (KMAX + 1);
	c.meta = w.meta + fr.sub_off;
	c.sub_shift = w.sub_shift;
	c.total_bits = total_bits;
	c.nseg = min(fs.nseg, fr.seg_cap - 1);
	c.nsub = nsub;
	c.B = a.g.B;

	unsigned long long T = pe.state & STATE_SYNC_MASK;
	uint32_t Tseg = pe.seg;
	bool in_sync = !mismatch;
	uint32_t r = 0xffffffffu;
	const unsigned long long old_exit = prl[hi].state & REC_INVALID ? REC_INVALID : chain_key(prl[hi].state, prl[hi].seg);
	bool gave_up = false;
	for (int q = lo; q <= hi; ++q) {
		const JhSubInfo old = prl[q];
		const bool oldvalid = !(old.state & REC_INVALID);
		if (in_sync && oldvalid) {
			T = old.state & STATE_SYNC_MASK;
			Tseg = old.seg;
			r = 0xffffffffu;
			continue;
		}
		if (selfdec > WALK_SELF_LIMIT) {
			gave_up = true;
			break;
		}
		const JhSubInfo R = walk_step(c, T, Tseg, j0 + (uint32_t)q, r, luts, tb, &dcs[0][lane], &bcast, &selfdec);
		in_sync = oldvalid && chain_key(old.state, old.seg) == chain_key(R.state, R.seg);
		__builtin_amdgcn_wave_barrier();
		if (lane == 0) {
			prl[q] = R;
		}
		__builtin_amdgcn_wave_barrier();
		__threadfence_block();
		T = R.state & STATE_SYNC_MASK;
		Tseg = R.seg;
	}
	if (gave_up) {
		if (lane == 0) {
			atomicOr(&w.state[f].status, (uint32_t)JH_ST_SERIAL);
			atomicOr(&w.state[f].serial_cause, 4u);
		}
		return;
	}
	__syncthreads();
	// records back, prefix sums and entry records of the whole range again
	Agg carry = agg_identity();
	for (int t0 = lo; t0 <= hi; t0 += 64) {
		const int jl = t0 + lane;
		const bool have = jl <= hi;
		Agg v = agg_identity();
		if (have) {
			prec[jl] = prl[jl];
			v = agg_of(prl[jl]);
		}
		Agg incl, excl;
		wave_seg_scan(v, incl, excl);
		if (have) {
			const Agg e = agg_combine(carry, excl);
			const bool first = b == 0 && jl == 0;
			const unsigned long long ps = first ? 0ull : (jl == lo ? pe.state : prl[jl - 1].state);
			const uint32_t pg = first ? 0u : (jl == lo ? pe.seg : prl[jl - 1].seg);
			w.start[fr.sub_off + j0 + jl] = make_start(e, ps, pg);
		}
		Agg last;
		last.nb = __shfl(incl.nb, 63, 64);
		last.rs = __shfl(incl.rs, 63, 64);
#pragma unroll
		for (int k = 0; k < JH_MAX_COMP; ++k) {
			last.dc[k] = __shfl(incl.dc[k], 63, 64);
		}
		carry = agg_combine(carry, last);
	}
	if (lane == 0) {
		store_agg(agg, carry, 0u);
		if (b > 0) {
			w.wg_key[fr.wg_off + b] = chain_key(pe.state, pe.seg);
		}
		atomicAdd(&w.state[f].nrepaired, 1u);
		(void)old_exit;
	}
}

// =================================================================================== finish

// serial walk of one frame by one thread: every sub-sequence gets its entry record in ABSOLUTE form
// (the workgroup bases are set to the identity).  Follows stored chains where they carry the true
// parse, decodes by itself where none does.
__device__ void serial_walk(const HuffArgs &a, const JhDecFrame &fr, JhDecState &fs, uint32_t nsub, uint32_t nwg,
                            const unsigned char *luts, const DecTables &tb, int *dcs)
{
	const JhDecodeWork &w = a.w;
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t sub_shift = w.sub_shift;
	const uint8_t *cs = w.cs + fr.cs_off;
	const uint32_t *seg_start = w.seg_start + fr.seg_off;
	const uint32_t nseg = min(fs.nseg, fr.seg_cap - 1);
	const JhSubInfo *rec = w.rec + (size_t)fr.sub_off * (KMAX + 1);
	const uint32_t *meta = w.meta + fr.sub_off;
	JhSubStart *start = w.start + fr.sub_off;

	Agg acc = agg_identity();
	unsigned long long T = 0;              // true entry state of sub-sequence jn
	uint32_t Tseg = 0;
	uint32_t r = 0xffffffffu;              // stored chain known to carry the true parse INTO sub-sequence jn
	for (uint32_t jn = 0; jn < nsub; ++jn) {
		JhSubStart o;
		o.state = T & STATE_SYNC_MASK;
		o.seg = Tseg;
		o.blk = acc.nb;
#pragma unroll
		for (int k = 0; k < JH_MAX_COMP; ++k) {
			o.pred[k] = acc.dc[k];
		}
		start[jn] = o;

		JhSubInfo R;
		const unsigned long long keyT = chain_key(T, Tseg);
		if ((uint32_t)(T & 0xffffffffu) >= total_bits) {
			// the parse has ended (end of data, or garbage with no restart segment left)
			R.state = T & STATE_SYNC_MASK;
			R.seg = Tseg;
			R.nb = 0;
#pragma unroll
			for (int k = 0; k < JH_MAX_COMP; ++k) {
				R.dc[k] = 0;
			}
			r = 0xffffffffu;
		} else {
			if (r == 0xffffffffu) {
				// a stored chain that ENTERS jn in the true state?  chain jn itself, or an older one
				const uint32_t lo = jn > (uint32_t)KMAX ? jn - KMAX : 0u;
				for (uint32_t c = jn + 1; c-- > lo && r == 0xffffffffu;) {
					const uint32_t mc = meta[c];
					const uint32_t steps = mc & 0xffu;
					if (jn - c > steps) {
						continue;
					}
					if (c == jn) {
						const uint32_t p0 = (jn << sub_shift);
						const uint32_t s0 = nseg > 1 ? find_segment(seg_start, nseg, p0 >> 3) : 0u;
						if (chain_key((unsigned long long)p0, s0) == keyT) {
							r = c;
						}
					} else {
						const JhSubInfo x = rec[(size_t)c * (KMAX + 1) + (jn - 1 - c)];
						if (chain_key(x.state, x.seg) == keyT) {
							r = c;
						}
					}
				}
			}
			if (r != 0xffffffffu) {
				const uint32_t m = meta[r];
				const uint32_t steps = m & 0xffu, link = (m >> 8) & 0xffu, d = jn - r;
				R = rec[(size_t)r * (KMAX + 1) + d];
				if (d == steps) {
					// the chain ends here: follow a true merge, never a restart link
					r = (link != 0 && !(m & META_RESTART)) ? r + link : 0xffffffffu;
				}
			} else {
				DecState st;
				state_from(st, T, Tseg);
				if (st.c >= a.g.B || st.z > 63 || st.seg >= nseg) {
					st.pos = total_bits;       // cannot happen for states produced by decode_span
				}
				decode_span<true>(st, ((jn + 1) << sub_shift), RegStream{cs}, total_bits, seg_start, nseg, luts, tb, a.g.B, dcs, 64, nullptr, 0, 0);
				store_info(&R, st);
			}
		}
		acc = agg_combine(acc, agg_of(R));
		T = R.state;
		Tseg = R.seg;
	}
	fs.nblocks = (uint32_t)acc.nb;
	for (uint32_t b = 0; b < nwg; ++b) {
		JhWgAgg z;
		z.nb = 0;
		z.flags = 0;
#pragma unroll
		for (int k = 0; k < JH_MAX_COMP; ++k) {
			z.dc[k] = 0;
		}
		z.next_seed = JH_SEED_NONE;
		z.pad = 0;
		w.wg_base[fr.wg_off + b] = z;
	}
}

// one wave per frame: exclusive scan of the workgroup totals, or the serial walk
__global__ __launch_bounds__(64) void k_huff_finish(const HuffArgs a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	unsigned char *luts = smem;
	__shared__ DecTables tb;
	__shared__ int dcs[JH_MAX_COMP][64];

	const JhDecodeWork &w = a.w;
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	JhDecState &fs = w.state[f];
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t sub_shift = w.sub_shift;
	const uint32_t nsub = (total_bits + (1u << sub_shift) - 1u) >> sub_shift;
	if (nsub == 0) {
		return;
	}
	const uint32_t nwg = jh_num_chain_wgs(nsub);
	if (fs.status & JH_ST_SERIAL) {
		load_tables(w, fr, luts, &tb, a.g, 64 * 4);
		if (threadIdx.x == 0) {
			serial_walk(a, fr, fs, nsub, nwg, luts, tb, &dcs[0][0]);
		}
		return;
	}
	// every workgroup's assumed entry state must be its predecessor's last record, no holes left
	bool bad = false;
	for (uint32_t b0 = 0; b0 < nwg; b0 += 64) {
		const uint32_t b = b0 + threadIdx.x;
		if (b < nwg) {
			bad |= (w.wg_agg[fr.wg_off + b].flags & JH_AGG_GAP) != 0;
			if (b > 0) {
				const JhSubInfo pe = w.prec[fr.sub_off + b * (uint32_t)OWN + KMAX - 1];
				bad |= (pe.state & REC_INVALID) != 0 || chain_key(pe.state, pe.seg) != w.wg_key[fr.wg_off + b];
			}
		}
	}
	if (__any(bad)) {
		if (threadIdx.x == 0) {
			fs.status |= JH_ST_SERIAL;
			fs.serial_cause |= 2u;
		}
		load_tables(w, fr, luts, &tb, a.g, 64 * 4);
		if (threadIdx.x == 0) {
			serial_walk(a, fr, fs, nsub, nwg, luts, tb, &dcs[0][0]);
		}
		return;
	}
	Agg carry = agg_identity();
	for (uint32_t b0 = 0; b0 < nwg; b0 += 64) {
		const uint32_t b = b0 + threadIdx.x;
		Agg v = agg_identity();
		if (b < nwg) {
			const JhWgAgg g = w.wg_agg[fr.wg_off + b];
			v.nb = g.nb;
			v.rs = (g.flags & JH_AGG_RESET) ? 1 : 0;
#pragma unroll
			for (int k = 0; k < JH_MAX_COMP; ++k) {
				v.dc[k] = g.dc[k];
			}
		}
		Agg incl, excl;
		wave_seg_scan(v, incl, excl);
		const Agg e = agg_combine(carry, excl);
		if (b < nwg) {
			JhWgAgg o;
			o.nb = e.nb;
			o.flags = e.rs ? (uint32_t)JH_AGG_RESET : 0u;
#pragma unroll
			for (int k = 0; k < JH_MAX_COMP; ++k) {
				o.dc[k] = e.dc[k];
			}
			o.next_seed = JH_SEED_NONE;
			o.pad = 0;
			w.wg_base[fr.wg_off + b] = o;
		}
		Agg last;
		last.nb = __shfl(incl.nb, 63, 64);
		last.rs = __shfl(incl.rs, 63, 64);
#pragma unroll
		for (int k = 0; k < JH_MAX_COMP; ++k) {
			last.dc[k] = __shfl(incl.dc[k], 63, 64);
		}
		carry = agg_combine(carry, last);
	}
	if (threadIdx.x == 0) {
		fs.nblocks = (uint32_t)carry.nb;
	}
}

int jh_launch_huff_sync(const JhDecodeWork &w, const JhGeom &g, hipStream_t s, const JhTimerHook *th)
{
	HuffArgs a;
	a.w = w;
	a.g = g;
	a.round = 0;
	const size_t lds = (size_t)w.lut_bytes;
	const unsigned nwg = jh_num_chain_wgs(w.max_subs);
	const unsigned nblk = (w.max_subs + 255u) / 256u;
	JH_T_BEGIN(th, "k_huff_chains", s);
	if (w.segmented) {
		hipLaunchKernelGGL(k_huff_chains<true>, dim3(nblk, w.nframes), dim3(256), lds, s, a);
	} else {
		hipLaunchKernelGGL(k_huff_chains<false>, dim3(nblk, w.nframes), dim3(256), lds, s, a);
	}
	JH_T_END(th, s);
	JH_T_BEGIN(th, "k_huff_links", s);
	hipLaunchKernelGGL(k_huff_links, dim3(nblk, w.nframes), dim3(256), 0, s, a);
	JH_T_END(th, s);
	JH_T_BEGIN(th, "k_huff_survivors", s);
	if (w.segmented) {
		hipLaunchKernelGGL(k_huff_survivors<true>, dim3(nblk, w.nframes), dim3(256), lds, s, a);
	} else {
		hipLaunchKernelGGL(k_huff_survivors<false>, dim3(nblk, w.nframes), dim3(256), lds, s, a);
	}
	JH_T_END(th, s);
	JH_T_BEGIN(th, "k_huff_emit", s);
	hipLaunchKernelGGL(k_huff_emit, dim3(nwg, w.nframes), dim3(256), 0, s, a);
	JH_T_END(th, s);
	JH_T_BEGIN(th, "k_huff_walk", s);
	for (int round = 0; round < JH_WALK_ROUNDS; ++round) {
		a.round = round;
		hipLaunchKernelGGL(k_huff_walk, dim3(nwg, w.nframes), dim3(64), lds, s, a);
	}
	JH_T_END(th, s);
	JH_T_BEGIN(th, "k_huff_finish", s);
	hipLaunchKernelGGL(k_huff_finish, dim3(1, w.nframes), dim3(64), lds, s, a);
	JH_T_END(th, s);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}
