// next_batch_pairwise (util/utils.py:123-157) ON THE GPU, bit-identical to the reference's NumPy stream.
//
// The host sampler (sampler.hip) walks the MT19937 stream draw by draw on one core: ~3.4 ms per MovieLens epoch, more
// than the 3.0 ms the GPU needs to train it, plus an upload of the triples.  Everything in that loop is either a
// data-parallel pass or a scan with a tiny carried state, so it is restated here as a chain of kernels on a side
// stream; the triples are born in HBM and the host only exchanges the 625-word generator state with NumPy.
//
//   1. ds_mt_kernel         the raw stream: successive MT19937 twists (3 dependent phases of <= 227 independent
//                           words each, in LDS), written tempered (W) and raw (for the state hand-back)
//   2. ds_shuffle_scan      np.random.shuffle's masked-rejection draws: which raw words are accepted depends on the
//                           running index i only, and i moves by at most 64 per wave step, so one wave classifies 64
//                           draws at a time (surely accepted: v <= i - lane; surely rejected: v > i) and falls back to
//                           a serial walk of the group when a draw is ambiguous (~lane / mask, i.e. rare until the
//                           last few thousand positions).  Output: J[i], the swap partner of position i.
//   3. ds_fy_*              Fisher-Yates from the top with KNOWN partners, resolved in parallel: position i is final
//                           after step i and receives what position J[i] held just before; what a position p holds
//                           before step t is decided by the most recent earlier step that targeted p (the smallest
//                           s > t with J[s] = p), which in turn moved the content of position s before step s, ...
//                           -- a chain of "previous occurrence" look-ups (mean length 1, max ~20) over the steps
//                           bucketed by target (counting sort with atomics; buckets are scanned, not sorted).
//   4. ds_gather            user / positive of every slot of the epoch from the new permutation
//   5. ds_filter_*          np.random.choice(item_list, m) is masked rejection over the SAME stream whatever the batch:
//                           the accepted item draws V[0..] are a compaction of the remaining raw words
//   6. ds_batches_kernel    per batch (serial: each batch starts where the previous one stopped drawing): slot t takes
//                           V[o + t]; the slots whose draw is one of the user's training items are compacted in slot
//                           order and redrawn from the following V entries, until none is left (utils.py:141-153)
//
// Host entry: crh_dsampler_epoch (asynchronous on `stream`, nothing allocated, no host round trip).
#include <math.h>

#include "crh_common.h"

namespace {

constexpr int MTN = 624, MTM = 397;

__host__ __device__ inline uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t far) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

struct Ctrl {            // device-side control block (ints): results that later kernels of the same epoch consume
    int q_shuffle_end;   // raw words the shuffle consumed
    int n_accepted;      // accepted item draws available in V
    int consumed;        // raw words the whole epoch consumed (shuffle + item draws)
    int status;          // 0 ok; 1 = the generated stream was too short; 2 = V exhausted
};

// ---------------------------------------------------------------- 1. the stream
// state: key[624] + pos (numpy's get_state()).  Stream word q (q = 0 is the next word numpy would draw) lives at flat
// index pos + q of the block array, block 0 being the current key.  pos == 624 simply starts in block 1.
__global__ __launch_bounds__(256) void ds_mt_kernel(const uint32_t* __restrict__ state, int n_blocks,
                                                    uint32_t* __restrict__ W, uint32_t* __restrict__ Kraw) {
    __shared__ uint32_t key[2][MTN];
    const int tid = threadIdx.x;
    for (int e = tid; e < MTN; e += 256) {
        const uint32_t k = state[e];
        key[0][e] = k;
        Kraw[e] = k;
        W[e] = mt_temper(k);
    }
    __syncthreads();
    int cur = 0;
    for (int b = 1; b < n_blocks; ++b) {
        const uint32_t* o = key[cur];
        uint32_t* nw = key[cur ^ 1];
        uint32_t* kr = Kraw + (size_t)b * MTN;
        uint32_t* wo = W + (size_t)b * MTN;
        if (tid < MTN - MTM) {                                   // i in [0, 227): old words only
            const uint32_t x = mt_mix(o[tid], o[tid + 1], o[tid + MTM]);
            nw[tid] = x;
            kr[tid] = x;
            wo[tid] = mt_temper(x);
        }
        __syncthreads();
        if (tid < MTN - MTM) {                                   // i in [227, 454): new[i - 227] from phase 1
            const int i = tid + (MTN - MTM);
            const uint32_t x = mt_mix(o[i], o[i + 1], nw[i - (MTN - MTM)]);
            nw[i] = x;
            kr[i] = x;
            wo[i] = mt_temper(x);
        }
        __syncthreads();
        {                                                        // i in [454, 624): new[i - 227] from phase 2
            const int i = tid + 2 * (MTN - MTM);
            if (i < MTN - 1) {
                const uint32_t x = mt_mix(o[i], o[i + 1], nw[i - (MTN - MTM)]);
                nw[i] = x;
                kr[i] = x;
                wo[i] = mt_temper(x);
            } else if (i == MTN - 1) {                           // key[623] = key[396] ^ f(key[623], NEW key[0])
                const uint32_t x = mt_mix(o[MTN - 1], nw[0], nw[MTM - 1]);
                nw[i] = x;
                kr[i] = x;
                wo[i] = mt_temper(x);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ---------------------------------------------------------------- 2. shuffle draws
__device__ __forceinline__ uint32_t pow2_mask(uint32_t x) { return 0xffffffffu >> __builtin_clz(x); }   // x >= 1

constexpr int CHUNK = 4096, RING_CHUNKS = 4, RING = CHUNK * RING_CHUNKS;   // LDS window over the stream (64 KiB)

// One wave.  W: the generated words, flat (block-major); the stream starts at flat index state[624] (= pos).
// Writes J[i] for i = n-1 .. 1 and ctrl->q_shuffle_end.  The window is filled by LDS-DMA (global_load_lds) one
// aligned 4096-word chunk at a time, two chunks ahead of the reader, so the loop itself never waits for memory.
__global__ __launch_bounds__(64) void ds_shuffle_scan_kernel(const uint32_t* __restrict__ W, const uint32_t* __restrict__ state,
                                                             int64_t n_words_total, int n, int32_t* __restrict__ J,
                                                             Ctrl* __restrict__ ctrl) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[RING];
    const int lane = threadIdx.x;
    const int pos = (int)state[MTN];
    const int64_t fend = n_words_total;                          // flat bound
    const int64_t last16 = fend - 4;                             // last 16-byte unit inside W
    // chunk c = flat words [c * CHUNK, (c + 1) * CHUNK) -> ring slot c % 4; 16 DMA pieces of 1 KiB each
    auto issue_chunk = [&](int64_t c) {
        uint32_t* dst = ring + (c & (RING_CHUNKS - 1)) * CHUNK;
#pragma unroll
        for (int k = 0; k < CHUNK / 256; ++k) {
            int64_t w = c * CHUNK + k * 256 + lane * 4;
            if (w > last16) w = last16;                          // never read outside W (such words are never consumed)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + w),
                                             (__attribute__((address_space(3))) void*)(dst + k * 256), 16, 0, 0);
        }
    };
    int64_t f = pos;
    int64_t loaded = f / CHUNK;                                  // chunks [first, loaded) have been issued
    for (int k = 0; k < 3; ++k) issue_chunk(loaded++);
    int64_t ready = 0;                                           // chunks below `ready` are known to have landed
    int i = n - 1;
    int status = 0;
    if (lane == 0) J[0] = 0;
    while (i >= 1) {
        if (f + 64 > fend) { status = 1; break; }
        const int64_t cf = f / CHUNK;
        if (cf + 2 >= loaded) issue_chunk(loaded++);             // entering a chunk: fetch the one two ahead (its slot
                                                                 // held chunk loaded - 4 <= cf - 2, no longer needed)
        if ((f + 63) / CHUNK >= ready) {                         // first touch of a chunk: its DMA must have landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ready = loaded;
        }
        const uint32_t mask = pow2_mask((uint32_t)i);
        const int stop = (int)(mask >> 1);
        const int cap = i - stop;                                // accepts this mask still allows
        const int v = (int)(ring[(f + lane) & (RING - 1)] & mask);
        const bool acc = v <= i - lane;                          // accepted whatever the earlier lanes did
        const bool rej = v > i;
        const unsigned long long amb = __ballot(!(acc || rej));
        if (amb == 0ull) {
            const unsigned long long bal = __ballot(acc);
            int A = __popcll(bal);
            const int pref = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            bool valid = acc;
            int c = 64;
            if (A >= cap) {                                      // the mask changes inside this group: stop behind the
                // cap-th accepted draw (also when it is the group's last accept: the lanes behind it were judged
                // with the old mask and must be read again with the new one)
                const unsigned long long last = __ballot(acc && pref == cap - 1);
                const int L = __builtin_ctzll(last);
                valid = acc && lane <= L;
                A = cap;
                c = L + 1;
            }
            if (valid) J[i - pref] = v;
            i -= A;
            f += c;
        } else {                                                 // rare: walk the group draw by draw
            int a = 0, c = 0;
            for (int l = 0; l < 64; ++l) {
                const int vl = __builtin_amdgcn_readlane(v, l);
                ++c;
                if (vl <= i - a) {
                    if (lane == 0) J[i - a] = vl;
                    ++a;
                    if (a == cap) break;
                }
            }
            i -= a;
            f += c;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // no DMA may outlive the kernel's LDS
    if (lane == 0) {
        ctrl->q_shuffle_end = (int)(f - pos);
        ctrl->status = status;
    }
}

// ---------------------------------------------------------------- scans (shared by 3 and 5)
// exclusive scan of n ints in three launches: block sums (1024 elements per block) -> scan of the sums (one block) ->
// local scans + offsets.  out[n] = total when out has n + 1 entries (total_out != NULL also receives it).
constexpr int SCAN_PER_BLOCK = 1024;

__device__ __forceinline__ int block_scan_256(int v, int* red, int& total) {
    // exclusive scan of one int per thread over a 256-thread block
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    __syncthreads();
    if (lane == 63) red[wv] = x;
    __syncthreads();
    const int t0 = red[0], t1 = red[1], t2 = red[2], t3 = red[3];
    total = t0 + t1 + t2 + t3;
    const int base = wv == 0 ? 0 : (wv == 1 ? t0 : (wv == 2 ? t0 + t1 : t0 + t1 + t2));
    return base + x - v;
}

__global__ __launch_bounds__(256) void ds_scan_sums_kernel(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ bsum) {
    __shared__ int red[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * 4;
    int s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += base + c < n ? in[base + c] : 0;
    int total;
    block_scan_256(s, red, total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void ds_scan_top_kernel(int32_t* __restrict__ bsum, int nb, int32_t* __restrict__ total_out) {
    __shared__ int red[4];
    int carry = 0;
    for (int base = 0; base < nb; base += 256) {
        const int e = base + threadIdx.x;
        const int v = e < nb ? bsum[e] : 0;
        int total;
        const int ex = block_scan_256(v, red, total);
        if (e < nb) bsum[e] = carry + ex;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(256) void ds_scan_apply_kernel(const int32_t* __restrict__ in, int64_t n, const int32_t* __restrict__ bsum,
                                                            int32_t* __restrict__ out, int32_t* __restrict__ out2) {
    __shared__ int red[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        v[c] = base + c < n ? in[base + c] : 0;
        s += v[c];
    }
    int total;
    int ex = block_scan_256(s, red, total) + bsum[blockIdx.x];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (base + c < n) {
            out[base + c] = ex;
            if (out2) out2[base + c] = ex;
        }
        ex += v[c];
    }
}

// ---------------------------------------------------------------- 3. Fisher-Yates with known partners
// (every kernel behind the scan leaves at once when the stream turned out too short: J is then incomplete)
__global__ __launch_bounds__(256) void ds_fy_count_kernel(const int32_t* __restrict__ J, int n, int32_t* __restrict__ cnt,
                                                          const Ctrl* __restrict__ ctrl) {
    if (ctrl->status) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 1 && i < n) {
        const int j = J[i];
        if (j != i) atomicAdd(cnt + j, 1);
    }
}

__global__ __launch_bounds__(256) void ds_fy_fill_kernel(const int32_t* __restrict__ J, int n, int32_t* __restrict__ cursor,
                                                         int32_t* __restrict__ bins, const Ctrl* __restrict__ ctrl) {
    if (ctrl->status) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 1 && i < n) {
        const int j = J[i];
        if (j != i) bins[atomicAdd(cursor + j, 1)] = i;
    }
}

// off[p] .. off[p] + cnt[p]: the steps that targeted position p (unordered).  new_order[i] = what position i holds at the end.
__global__ __launch_bounds__(256) void ds_fy_chase_kernel(const int32_t* __restrict__ J, int n, const int32_t* __restrict__ off,
                                                          const int32_t* __restrict__ cnt, const int32_t* __restrict__ bins,
                                                          const int32_t* __restrict__ order, int32_t* __restrict__ new_order,
                                                          const Ctrl* __restrict__ ctrl) {
    if (ctrl->status) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int p = i >= 1 ? J[i] : 0, t = i;
    for (;;) {
        const int b0 = off[p], b1 = b0 + cnt[p];
        int m = 0x7fffffff;
        for (int e = b0; e < b1; ++e) {
            const int s = bins[e];
            if (s > t && s < m) m = s;
        }
        if (m == 0x7fffffff) break;
        p = t = m;
    }
    new_order[i] = order[p];
}

// ---------------------------------------------------------------- 4. users / positives of the epoch
__global__ __launch_bounds__(256) void ds_gather_kernel(const int32_t* __restrict__ new_order, int n, const int32_t* __restrict__ rec_u,
                                                        const int32_t* __restrict__ rec_i, int32_t* __restrict__ order,
                                                        int32_t* __restrict__ user_out, int32_t* __restrict__ pos_out,
                                                        const Ctrl* __restrict__ ctrl) {
    if (ctrl->status) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int r = new_order[t];
    order[t] = r;
    user_out[t] = rec_u[r];
    pos_out[t] = rec_i[r];
}

// ---------------------------------------------------------------- 5. accepted item draws
__global__ __launch_bounds__(256) void ds_filter_flag_kernel(const uint32_t* __restrict__ W, const uint32_t* __restrict__ state,
                                                             int64_t n_words_total, const Ctrl* __restrict__ ctrl, uint32_t imask,
                                                             uint32_t imax, int32_t* __restrict__ flag) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;      // stream word index
    const int pos = (int)state[MTN];
    if (q >= n_words_total - pos) return;
    flag[q] = (q >= ctrl->q_shuffle_end && (W[pos + q] & imask) <= imax) ? 1 : 0;
}

__global__ __launch_bounds__(256) void ds_filter_scatter_kernel(const uint32_t* __restrict__ W, const uint32_t* __restrict__ state,
                                                                int64_t n_words_total, const int32_t* __restrict__ flag,
                                                                const int32_t* __restrict__ off, uint32_t imask,
                                                                int32_t* __restrict__ V, int32_t* __restrict__ Vraw) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int pos = (int)state[MTN];
    if (q >= n_words_total - pos) return;
    if (flag[q]) {
        const int o = off[q];
        V[o] = (int32_t)(W[pos + q] & imask);
        Vraw[o] = (int32_t)q;
    }
}

// ---------------------------------------------------------------- 6. batches
struct RatedTest {
    const uint32_t* bits;     // users x wpu words, bit (item & 31) of word item >> 5
    int64_t wpu;
    const int64_t* rowptr;    // or CSR (ascending per user)
    const int32_t* col;
};

__device__ __forceinline__ bool is_rated(const RatedTest& r, int u, int item) {
    if (r.bits) return (r.bits[(int64_t)u * r.wpu + (item >> 5)] >> (item & 31)) & 1u;
    int64_t lo = r.rowptr[u], hi = r.rowptr[u + 1];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int c = r.col[mid];
        if (c == item) return true;
        if (c < item) lo = mid + 1; else hi = mid;
    }
    return false;
}

constexpr int DS_THREADS = 512, DS_WAVES = DS_THREADS / 64, DS_MAX_BATCH = 8192, DS_MAX_R = DS_MAX_BATCH / DS_THREADS;

// exclusive position of every set flag among Rn x DS_THREADS slots in slot order (slot = r * DS_THREADS + tid); returns
// the total.  part: [R][DS_WAVES] wave counts in LDS.  Rn (<= R, block-uniform) = passes actually in use.
template <int R>
__device__ __forceinline__ int block_compact_offsets(const bool (&f)[R], int (&rank)[R], int* part, int Rn) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r < Rn) {
            const unsigned long long b = __ballot(f[r]);
            rank[r] = __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
            if (lane == 0) part[r * DS_WAVES + wv] = __popcll(b);
        }
    }
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r < Rn) {
            int base = total;
#pragma unroll
            for (int w = 0; w < DS_WAVES; ++w) {
                const int c = part[r * DS_WAVES + w];
                if (w < wv) base += c;
                total += c;
            }
            rank[r] += base;
        }
    }
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(DS_THREADS) void ds_batches_kernel(const int32_t* __restrict__ V, const int32_t* __restrict__ Vraw,
                                                                 Ctrl* __restrict__ ctrl, RatedTest rt, int n, int B,
                                                                 const int32_t* __restrict__ user_out, int32_t* __restrict__ neg_out,
                                                                 const uint32_t* __restrict__ Kraw, uint32_t* __restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* chk0 = reinterpret_cast<int*>(smem);
    int* chk1 = chk0 + DS_MAX_BATCH;
    int* part = chk1 + DS_MAX_BATCH;                               // [DS_MAX_R][DS_WAVES]
    const int tid = threadIdx.x;
    const int nV = ctrl->n_accepted;
    int status = ctrl->status;
    int64_t o = 0;
    for (int lo = 0; lo < n && status == 0; lo += B) {
        const int S = min(B, n - lo);
        if (o + S > nV) { status = 2; break; }
        bool f[DS_MAX_R];
        int rank[DS_MAX_R], item[DS_MAX_R], usr[DS_MAX_R];
        const int Rn = (S + DS_THREADS - 1) / DS_THREADS;
#pragma unroll
        for (int r = 0; r < DS_MAX_R; ++r) {
            const int t = r * DS_THREADS + tid;
            item[r] = usr[r] = 0;
            f[r] = false;
            if (r < Rn && t < S) {
                item[r] = V[o + t];
                usr[r] = user_out[lo + t];
            }
        }
#pragma unroll
        for (int r = 0; r < DS_MAX_R; ++r) {
            const int t = r * DS_THREADS + tid;
            if (r < Rn && t < S) {
                neg_out[lo + t] = item[r];
                f[r] = is_rated(rt, usr[r], item[r]);
            }
        }
        int nc = block_compact_offsets<DS_MAX_R>(f, rank, part, Rn);
#pragma unroll
        for (int r = 0; r < DS_MAX_R; ++r)
            if (f[r]) chk0[rank[r]] = r * DS_THREADS + tid;
        __syncthreads();
        o += S;
        int* cur = chk0;
        int* nxt = chk1;
        while (nc > 0) {                                           // redraw the rejected slots, in slot order
            if (o + nc > nV) { status = 2; break; }
            int slot[DS_MAX_R];
            const int Rc = (nc + DS_THREADS - 1) / DS_THREADS;
#pragma unroll
            for (int r = 0; r < DS_MAX_R; ++r) {
                const int q = r * DS_THREADS + tid;
                f[r] = false;
                slot[r] = 0;
                if (r < Rc && q < nc) {
                    slot[r] = cur[q];
                    const int it = V[o + q];
                    neg_out[lo + slot[r]] = it;
                    f[r] = is_rated(rt, user_out[lo + slot[r]], it);
                }
            }
            const int nn = block_compact_offsets<DS_MAX_R>(f, rank, part, Rc);
#pragma unroll
            for (int r = 0; r < DS_MAX_R; ++r)
                if (f[r]) nxt[rank[r]] = slot[r];
            __syncthreads();
            o += nc;
            nc = nn;
            int* sw = cur; cur = nxt; nxt = sw;
        }
    }
    // hand the generator back: position behind the last raw word consumed
    const int q_end = ctrl->q_shuffle_end;
    const int64_t consumed = o > 0 ? (int64_t)Vraw[o - 1] + 1 : q_end;
    const int pos0 = (int)state[MTN];
    const int64_t flat = pos0 + consumed;                          // flat index of the next unread word
    int64_t blk = flat / MTN;
    int npos = (int)(flat - blk * MTN);
    if (npos == 0 && blk > 0) { blk -= 1; npos = MTN; }            // numpy keeps pos = 624 until the next draw twists
    __syncthreads();                                               // every thread has read state[MTN]
    if (status == 0) {
        for (int e = tid; e < MTN; e += DS_THREADS) state[e] = Kraw[blk * MTN + e];
        if (tid == 0) state[MTN] = (uint32_t)npos;
    }
    if (tid == 0) {
        ctrl->consumed = (int)consumed;
        ctrl->status = status;
        state[MTN + 1] = (uint32_t)status;
    }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct DsLayout {
    size_t W, Kraw, J, cnt, off, cursor, bins, new_order, flag, foff, bsum, V, Vraw, ctrl, total;
};

DsLayout ds_layout(int64_t n, int64_t n_blocks) {
    const size_t words = (size_t)n_blocks * MTN;
    DsLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += align256(bytes); return at; };
    L.W = take(words * 4);
    L.Kraw = take(words * 4);
    L.J = take((size_t)n * 4);
    L.cnt = take((size_t)(n + 1) * 4);
    L.off = take((size_t)(n + 1) * 4);
    L.cursor = take((size_t)(n + 1) * 4);
    L.bins = take((size_t)n * 4);
    L.new_order = take((size_t)n * 4);
    L.flag = take(words * 4);
    L.foff = take((words + 1) * 4);
    L.bsum = take((std::max(words, (size_t)n + 1) / SCAN_PER_BLOCK + 2) * 4);
    L.V = take(words * 4);
    L.Vraw = take(words * 4);
    L.ctrl = take(sizeof(Ctrl));
    L.total = o;
    return L;
}

int ds_scan(const int32_t* in, int64_t n, int32_t* bsum, int32_t* out, int32_t* out2, int32_t* total_out, hipStream_t st) {
    const int nb = (int)((n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
    hipLaunchKernelGGL(ds_scan_sums_kernel, dim3(nb), dim3(256), 0, st, in, n, bsum);
    hipLaunchKernelGGL(ds_scan_top_kernel, dim3(1), dim3(256), 0, st, bsum, nb, total_out);
    hipLaunchKernelGGL(ds_scan_apply_kernel, dim3(nb), dim3(256), 0, st, in, n, bsum, out, out2);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

}  // namespace

extern "C" size_t crh_dsampler_workspace_bytes(int64_t n_records, int64_t n_blocks) {
    if (n_records <= 0 || n_blocks < 2) return 0;
    return ds_layout(n_records, n_blocks).total;
}

// Key blocks to generate for one epoch: expected raw draws of the shuffle (sum over i of (mask_i + 1) / (i + 1)) and of
// the item draws ((1 + reject_rate) * n_records accepted at (imask + 1) / n_items raw words each), + 3 % + a margin.
extern "C" int64_t crh_dsampler_blocks_hint(int64_t n_records, int32_t n_items, double reject_rate) {
    if (n_records <= 0 || n_items <= 1) return 0;
    double raws = 0.0;
    int64_t i = n_records - 1;
    while (i >= 1) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz((uint32_t)i);
        const int64_t stop = (int64_t)(mask >> 1);
        // sum_{k = stop+1 .. i} (mask + 1) / (k + 1)
        raws += ((double)mask + 1.0) * (log((double)i + 1.5) - log((double)stop + 1.5));
        i = stop;
    }
    const uint32_t imax = (uint32_t)(n_items - 1);
    const uint32_t imask = 0xffffffffu >> __builtin_clz(imax);
    if (reject_rate < 0.0) reject_rate = 0.0;
    if (reject_rate > 0.95) reject_rate = 0.95;
    raws += (double)n_records / (1.0 - reject_rate) * ((double)imask + 1.0) / (double)n_items;
    return (int64_t)(raws * 1.03 / MTN) + 24;
}

extern "C" int crh_dsampler_max_batch(void) { return DS_MAX_BATCH; }

// One epoch of next_batch_pairwise triples, produced on the device (see the file header).  Asynchronous on `stream`.
extern "C" int crh_dsampler_epoch(const crh_dsampler_io* io, int64_t batch_size, int64_t n_blocks, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    CRH_CHECK_ARG(io && io->rec_user && io->rec_item && io->order && io->state && io->user_out && io->pos_out && io->neg_out,
                  "crh_dsampler_epoch: NULL pointer");
    const int64_t n = io->n_records;
    CRH_CHECK_ARG(n >= 1 && n < ((int64_t)1 << 30), "crh_dsampler_epoch: n_records=%lld outside 1..2^30", (long long)n);
    CRH_CHECK_ARG(io->n_items >= 2 && io->n_users >= 1, "crh_dsampler_epoch: needs at least two items (numpy draws nothing for one)");
    CRH_CHECK_ARG(batch_size >= 1 && batch_size <= DS_MAX_BATCH, "crh_dsampler_epoch: batch_size=%lld outside 1..%d",
                  (long long)batch_size, DS_MAX_BATCH);
    CRH_CHECK_ARG(io->rated_bits || (io->rated_rowptr && io->rated_col), "crh_dsampler_epoch: no rated-item table");
    CRH_CHECK_ARG(n_blocks >= 2 && n_blocks * MTN < ((int64_t)1 << 31), "crh_dsampler_epoch: n_blocks=%lld", (long long)n_blocks);
    const DsLayout L = ds_layout(n, n_blocks);
    if (!workspace || workspace_bytes < L.total) {
        crh_set_error("crh_dsampler_epoch: workspace %zu < %zu bytes", workspace_bytes, L.total);
        return CRH_ERR_WS;
    }
    char* ws = reinterpret_cast<char*>(workspace);
    uint32_t* W = reinterpret_cast<uint32_t*>(ws + L.W);
    uint32_t* Kraw = reinterpret_cast<uint32_t*>(ws + L.Kraw);
    int32_t* J = reinterpret_cast<int32_t*>(ws + L.J);
    int32_t* cnt = reinterpret_cast<int32_t*>(ws + L.cnt);
    int32_t* off = reinterpret_cast<int32_t*>(ws + L.off);
    int32_t* cursor = reinterpret_cast<int32_t*>(ws + L.cursor);
    int32_t* bins = reinterpret_cast<int32_t*>(ws + L.bins);
    int32_t* new_order = reinterpret_cast<int32_t*>(ws + L.new_order);
    int32_t* flag = reinterpret_cast<int32_t*>(ws + L.flag);
    int32_t* foff = reinterpret_cast<int32_t*>(ws + L.foff);
    int32_t* bsum = reinterpret_cast<int32_t*>(ws + L.bsum);
    int32_t* V = reinterpret_cast<int32_t*>(ws + L.V);
    int32_t* Vraw = reinterpret_cast<int32_t*>(ws + L.Vraw);
    Ctrl* ctrl = reinterpret_cast<Ctrl*>(ws + L.ctrl);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t words = n_blocks * MTN;
    const int nn = (int)n;
    const unsigned gb = (unsigned)((n + 255) / 256);

    CRH_HIP(hipMemsetAsync(ctrl, 0, sizeof(Ctrl), st));
    CRH_HIP(hipMemsetAsync(cnt, 0, (size_t)(n + 1) * 4, st));
    hipLaunchKernelGGL(ds_mt_kernel, dim3(1), dim3(256), 0, st, io->state, (int)n_blocks, W, Kraw);
    hipLaunchKernelGGL(ds_shuffle_scan_kernel, dim3(1), dim3(64), 0, st, W, io->state, words, nn, J, ctrl);
    hipLaunchKernelGGL(ds_fy_count_kernel, dim3(gb), dim3(256), 0, st, J, nn, cnt, ctrl);
    int rc = ds_scan(cnt, n + 1, bsum, off, cursor, nullptr, st);
    if (rc != CRH_OK) return rc;
    hipLaunchKernelGGL(ds_fy_fill_kernel, dim3(gb), dim3(256), 0, st, J, nn, cursor, bins, ctrl);
    hipLaunchKernelGGL(ds_fy_chase_kernel, dim3(gb), dim3(256), 0, st, J, nn, off, cnt, bins, io->order, new_order, ctrl);
    hipLaunchKernelGGL(ds_gather_kernel, dim3(gb), dim3(256), 0, st, new_order, nn, io->rec_user, io->rec_item, io->order,
                       io->user_out, io->pos_out, ctrl);
    const uint32_t imax = (uint32_t)(io->n_items - 1);
    const uint32_t imask = 0xffffffffu >> __builtin_clz(imax);
    const unsigned wb = (unsigned)((words + 255) / 256);
    CRH_HIP(hipMemsetAsync(flag, 0, (size_t)words * 4, st));
    hipLaunchKernelGGL(ds_filter_flag_kernel, dim3(wb), dim3(256), 0, st, W, io->state, words, ctrl, imask, imax, flag);
    rc = ds_scan(flag, words, bsum, foff, nullptr, &ctrl->n_accepted, st);
    if (rc != CRH_OK) return rc;
    hipLaunchKernelGGL(ds_filter_scatter_kernel, dim3(wb), dim3(256), 0, st, W, io->state, words, flag, foff, imask, V, Vraw);
    RatedTest rt{io->rated_bits, io->bits_words_per_user, io->rated_rowptr, io->rated_col};
    const size_t lds = (size_t)(2 * DS_MAX_BATCH + DS_MAX_R * DS_WAVES) * sizeof(int);
    CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ds_batches_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    hipLaunchKernelGGL(ds_batches_kernel, dim3(1), dim3(DS_THREADS), lds, st, V, Vraw, ctrl, rt, nn, (int)batch_size,
                       io->user_out, io->neg_out, Kraw, io->state);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}
