// Host-side negative sampler that reproduces the reference's NumPy stream bit for bit.
//
// Replaces util/utils.py:123-157 (next_batch_pairwise): np.random.shuffle(training_data) once per
// epoch (in place, cumulative), then per batch np.random.choice(item_list, n) with rejection of
// the user's training items, redrawing only the rejected slots until none is left.
// NumPy's legacy global RNG is MT19937 (np.random.seed(s) == init_genrand(s)); randint/choice use
// masked rejection over 32-bit draws and shuffle is Fisher-Yates from the top with the same
// bounded draw -- all restated here so a whole epoch of triples is produced
// in a few milliseconds on one host thread, off the GPU's critical path.  Pure host code: no HIP call.
//
// The other samplers of util/utils.py are restated below the same way:
// next_batch_pairwise_LARA (:160-188), _CLCRec (:191-233), _CCFCRec (:237-300) draw from CPython's `random`
// module (also MT19937: getrandbits(k) = top k bits of one 32-bit output, _randbelow(n) = rejection over
// getrandbits(n.bit_length()), shuffle = Fisher-Yates from the top, sample = pool / selected-set), CCFCRec
// additionally from NumPy's stream for the positives; next_batch_cgrc (:303-336) draws from NumPy only and
// returns list(set(...)), whose order is CPython's set-table slot order (restated in IntSet).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <type_traits>
#include <string>
#include <vector>

#include <sys/mman.h>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

#include "crh_common.h"

// host-only function multiversioning (the device pass of hipcc parses this file too and knows no x86 targets)
#if defined(__HIP_DEVICE_COMPILE__)
#define CRH_HOST_CLONES
#else
#define CRH_HOST_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#endif

namespace {

// The randomly accessed tables of the sampler (the permutation, the membership bitmap: 3 - 5 MB each) live in 2 MB pages when the
// kernel grants them (transparent huge pages, madvise): with 4 KB pages every swap of the shuffle and every membership test
// also misses the first-level TLB.
template <typename T>
struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <typename U> HugeAlloc(const HugeAlloc<U>&) {}
    static constexpr size_t HUGE = (size_t)2 << 20;
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < HUGE) {
            void* q = std::malloc(bytes ? bytes : 1);
            if (!q) throw std::bad_alloc();
            return (T*)q;
        }
        void* q = std::aligned_alloc(HUGE, (bytes + HUGE - 1) / HUGE * HUGE);
        if (!q) throw std::bad_alloc();
        madvise(q, (bytes + HUGE - 1) / HUGE * HUGE, MADV_HUGEPAGE);     // advisory: the result is not checked
        return (T*)q;
    }
    void deallocate(T* q, size_t) { std::free(q); }
    template <typename U> bool operator==(const HugeAlloc<U>&) const { return true; }
    template <typename U> bool operator!=(const HugeAlloc<U>&) const { return false; }
};
template <typename T> using HugeVec = std::vector<T, HugeAlloc<T>>;

// The two stream compactions of an epoch -- which raw draws of the shuffle are accepted, which masked words are item ids --
// 16 draws per step on hosts with AVX-512 (mask, two compares, vpcompressd in its register form, one store): the scalar forms
// cost 2.5 and 1.5 cycles per raw draw, these a quarter of a cycle (tools/probes/sampler_host_probe.cpp).  Both return the
// number of raw draws they consumed and stop where the scalar walk has to take over (a group with a draw in (ii - 16, ii], the
// end of the window, the end of the mask's range).
#if !defined(__HIP_DEVICE_COMPILE__)
#define CRH_HAVE_AVX512_PATH 1
__attribute__((target("avx512f"))) int shuffle_accept_avx512(const uint32_t* __restrict__ w, int avail, uint32_t mask, int64_t stop,
                                                             int64_t& ii, uint32_t* __restrict__ jl, int64_t& m) {
    int used = 0;
    const __m512i vmask = _mm512_set1_epi32((int)mask);
    while (used + 16 <= avail && ii - stop >= 16) {
        const __m512i v = _mm512_and_si512(_mm512_loadu_si512((const void*)(w + used)), vmask);
        const __mmask16 acc = _mm512_cmple_epu32_mask(v, _mm512_set1_epi32((int)(uint32_t)(ii - 16)));
        const __mmask16 rej = _mm512_cmpgt_epu32_mask(v, _mm512_set1_epi32((int)(uint32_t)ii));
        if (__builtin_expect((__mmask16)(acc | rej) != (__mmask16)0xffff, 0)) break;
        _mm512_storeu_si512((void*)(jl + m), _mm512_maskz_compress_epi32(acc, v));   // (16 lanes stored: m + 16 <= used + 16 <= 624)
        const int c = __builtin_popcount((unsigned)acc);
        m += c;
        ii -= c;
        used += 16;
    }
    return used;
}
__attribute__((target("avx512f"))) int draw_items_avx512(const uint32_t* __restrict__ w, int avail, uint32_t imask, uint32_t imax,
                                                         int32_t* __restrict__ dst, int64_t cnt, int64_t& w_) {
    int used = 0;
    const __m512i vm = _mm512_set1_epi32((int)imask), vx = _mm512_set1_epi32((int)imax);
    while (used + 16 <= avail && w_ + 16 <= cnt) {           // (a store of 16 lanes stays inside dst[0, cnt))
        const __m512i v = _mm512_and_si512(_mm512_loadu_si512((const void*)(w + used)), vm);
        const __mmask16 acc = _mm512_cmple_epu32_mask(v, vx);
        _mm512_storeu_si512((void*)(dst + w_), _mm512_maskz_compress_epi32(acc, v));
        w_ += __builtin_popcount((unsigned)acc);
        used += 16;
    }
    return used;
}
bool host_has_avx512() {
    static const bool have = __builtin_cpu_supports("avx512f") && getenv("CRH_SAMPLER_NO_AVX512") == nullptr;
    return have;
}
#else
int shuffle_accept_avx512(const uint32_t*, int, uint32_t, int64_t, int64_t&, uint32_t*, int64_t&) { return 0; }
int draw_items_avx512(const uint32_t*, int, uint32_t, uint32_t, int32_t*, int64_t, int64_t&) { return 0; }
bool host_has_avx512() { return false; }
#endif

struct MT19937 {
    uint32_t key[624];
    int pos;
    void seed(uint32_t s) {
        key[0] = s;
        for (int i = 1; i < 624; ++i) key[i] = 1812433253u * (key[i - 1] ^ (key[i - 1] >> 30)) + (uint32_t)i;
        pos = 624;
    }
    // the twist has a dependency distance of 227 words and the tempering none: both loops vectorise; the clones let the
    // loader pick the widest unit of the machine the library runs on (the build container and the GPU box differ)
    CRH_HOST_CLONES void gen() {
        const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MAT = 0x9908b0dfu;
        int i;
        for (i = 0; i < 624 - 397; ++i) {
            uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);
        }
        for (; i < 623; ++i) {
            uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);
        }
        uint32_t y = (key[623] & UP) | (key[0] & LO);
        key[623] = key[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);
        pos = 0;
    }
    // outputs of the current key block, tempered all at once (the loop vectorises); next() is then one load
    uint32_t out[624];
    CRH_HOST_CLONES void temper_block() {
        for (int q = 0; q < 624; ++q) {
            uint32_t y = key[q];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            out[q] = y;
        }
    }
    inline uint32_t next() {
        if (pos >= 624) {
            gen();
            temper_block();
        }
        return out[pos++];
    }
    // the unread words of the current block (at least one): callers walk them with local variables and report back
    // how many they took, so the hot loops carry no generator state through memory
    inline const uint32_t* window(int& avail) {
        if (pos >= 624) {
            gen();
            temper_block();
        }
        avail = 624 - pos;
        return out + pos;
    }
    // uniform integer in [0, max] by masked rejection over 32-bit draws (numpy legacy, max < 2^32)
    inline uint32_t bounded(uint32_t max) {
        if (max == 0) return 0;
        uint32_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        while ((v = (next() & mask)) > max) {}
        return v;
    }
    // CPython random._randbelow_with_getrandbits(n), n < 2^32: k = n.bit_length(); r = getrandbits(k) until r < n
    inline uint32_t py_randbelow(uint32_t n) {
        if (n == 0) return 0;
        const int shift = __builtin_clz(n);          // 32 - bit_length
        uint32_t r;
        while ((r = (next() >> shift)) >= n) {}
        return r;
    }
};

// Order of list(set_of_small_ints) in CPython 3.7 - 3.12 (Objects/setobject.c): open addressing, hash(i) = i,
// 9 linear probes then the perturbed jump; the table is rebuilt (slot order) at 4x (2x beyond 50000) the used
// count once fill * 5 >= mask * 3.  No deletions happen on this path, so there are no dummy entries.
struct IntSet {
    std::vector<int32_t> tab;      // -1 = unused
    size_t mask = 7, used = 0;
    IntSet() : tab(8, -1) {}
    void clear() { tab.assign(8, -1); mask = 7; used = 0; }
    static void insert_clean(std::vector<int32_t>& t, size_t m, int32_t key) {
        size_t perturb = (size_t)key, i = (size_t)key & m;
        for (;;) {
            if (t[i] < 0) { t[i] = key; return; }
            if (i + 9 <= m)
                for (size_t j = 1; j <= 9; ++j)
                    if (t[i + j] < 0) { t[i + j] = key; return; }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & m;
        }
    }
    void add(int32_t key) {
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        for (;;) {
            const size_t probes = (i + 9 <= mask) ? 9 : 0;
            for (size_t j = 0; j <= probes; ++j) {
                if (tab[i + j] == key) return;
                if (tab[i + j] < 0) {
                    tab[i + j] = key;
                    ++used;
                    if (used * 5 >= mask * 3) {
                        const size_t want = used > 50000 ? used * 2 : used * 4;
                        size_t size = 8;
                        while (size <= want) size <<= 1;
                        std::vector<int32_t> nt(size, -1);
                        for (int32_t k : tab)
                            if (k >= 0) insert_clean(nt, size - 1, k);
                        tab.swap(nt);
                        mask = size - 1;
                    }
                    return;
                }
            }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
};

}  // namespace

struct crh_sampler {
    MT19937 rng;
    std::vector<int32_t> rec_u, rec_i;    // training records, internal ids, file order
    std::vector<int64_t> rec_ui;          // the same, packed (user | item << 32): one cache line per gathered record
    HugeVec<int32_t> order;               // cumulative permutation of the records
    // next_batch_pairwise on record sets whose ids fit 16 bits (MovieLens, CiteULike): the permutation carries each record WITH it
    // -- index | (user << 16 | item) << 32 -- so that the shuffle needs no third random access for the gather (ord64_live: this
    // array is the authoritative permutation and `order` is stale until sync_order)
    HugeVec<int64_t> ord64, snap_ord64;
    bool ord64_live = false, snap_live = false;
    bool pack16_ok() const { return n_users <= 65536 && n_items <= 65536; }
    void sync_order() {
        if (!ord64_live) return;
        keep_snapshot_copy();                                   // (an undo log speaks of the array it was written against)
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int32_t)(uint32_t)ord64[i];
        ord64_live = false;
    }
    // crh_sampler_snapshot / _restore.  A snapshot copies nothing: the pairwise epoch that follows it logs the targets of its
    // swaps (one uint32 per record, written in stream order by the loop that computes them anyway), and restoring replays them
    // backwards -- a copy of the permutation per epoch (5 MB read + 5 MB written on the worker's core, and the same again
    // evicted from its cache) cost a sixth of the epoch for a roll-back that happens once per training run.  Whatever else
    // touches the permutation under a snapshot (a second epoch, the other samplers' shuffles) first turns the snapshot into a
    // plain copy.
    bool snap_armed = false, snap_logged = false, snap_copied = false;
    HugeVec<uint32_t> swap_log;                                 // swap_log[t] = target of the swap of slot n - 1 - t
    template <typename E> static void undo_swaps(E* ord, const uint32_t* log, int64_t n) {
        for (int64_t pi = 1; pi < n; ++pi) std::swap(ord[pi], ord[log[n - 1 - pi]]);
    }
    void keep_snapshot_copy() {
        if (!snap_armed || snap_copied) return;
        const int64_t n = (int64_t)order.size();
        snap_live = ord64_live;
        if (ord64_live) {
            snap_ord64 = ord64;
            if (snap_logged) undo_swaps(snap_ord64.data(), swap_log.data(), n);
        } else {
            snap_order = order;
            if (snap_logged) undo_swaps(snap_order.data(), swap_log.data(), n);
        }
        snap_copied = true;
        snap_logged = false;
    }
    std::vector<int64_t> rowptr;          // per user: sorted training items (rejection test)
    std::vector<int32_t> items;
    int32_t n_users, n_items;
    std::vector<int32_t> check, next_check, redraw;
    MT19937 snap_rng, snap_pyrng;         // crh_sampler_snapshot / _restore (speculative sampling of the next epoch)
    HugeVec<int32_t> snap_order;
    HugeVec<uint64_t> bits;               // users x items membership bitmap when it is small enough to stay cached
    int64_t words_per_user = 0;
    // --- the other samplers (set by crh_sampler_set_catalogue) ---
    MT19937 pyrng;                        // CPython `random` module stream
    int32_t n_users_seen = 0;             // len(data.user): the negative-user pool of LARA / CCFCRec
    std::vector<int64_t> first_ptr;       // per user: distinct training items in first-appearance (dict) order
    std::vector<int32_t> first_items;
    std::vector<int32_t> warm_items;      // non-cold item ids ascending = the CLCRec / CCFCRec candidate pool order
    std::vector<int64_t> rw_ptr;          // per user: ranks (in warm_items) of its distinct training items, ascending
    std::vector<int32_t> rw_rank;
    std::vector<int32_t> picked;          // random.sample scratch (copy-the-pool branch)
    std::vector<int32_t> cand_list;       // one user's candidate list, materialised when a record draws many times
    std::vector<uint32_t> stamp;          // random.sample's "selected" set as per-position stamps
    uint32_t stamp_now = 0;
    std::vector<int32_t> item_users;      // per item: distinct training users (a full item has no negative user)
    IntSet bset;
    bool catalogue = false;
    // --- background epoch (crh_sampler_epoch_async / _wait): ONE persistent worker thread per sampler.  A thread that is
    // created per epoch starts on a sleeping core at its lowest clock and runs the 3 ms of an epoch at half speed
    // (measured: 3.1 ms on the calling thread, 5.9 - 7.6 ms on a fresh one); the worker instead spins for a while
    // after each job (the next one normally arrives within a few milliseconds), then sleeps on a condition variable.
    std::thread worker;
    std::mutex wm;
    std::condition_variable wcv;
    std::atomic<int> job{0};              // 0 idle, 1 queued, 2 running, 3 done
    std::atomic<bool> shutdown{false};    // set by crh_sampler_destroy only; never overwritten by a finishing job
    int64_t job_batch = 0;
    int32_t *job_u = nullptr, *job_p = nullptr, *job_n = nullptr;
    int job_rc = 0, job_snapshot = 0;
    std::string job_err;
    int64_t n_candidates(int32_t u) const { return (int64_t)warm_items.size() - (rw_ptr[u + 1] - rw_ptr[u]); }
    // j-th (0-based) entry of [k for k in warm item order if k not in training_set_u[user]]
    int32_t candidate(int32_t u, int64_t j) const {
        const int32_t* r = rw_rank.data() + rw_ptr[u];
        int64_t lo = 0, hi = rw_ptr[u + 1] - rw_ptr[u];          // count of i with r[i] - i <= j (non-decreasing in i)
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)r[mid] - mid <= j) lo = mid + 1; else hi = mid;
        }
        return warm_items[(size_t)(j + lo)];
    }
    // the whole list (merge of the pool with the user's sorted ranks): pays off from about pool/16 draws per record
    const int32_t* materialise(int32_t u) {
        const int32_t* r = rw_rank.data() + rw_ptr[u];
        const int64_t nr = rw_ptr[u + 1] - rw_ptr[u], nw = (int64_t)warm_items.size();
        cand_list.resize((size_t)(nw - nr));
        int64_t w = 0, k = 0;
        for (int64_t i = 0; i <= nr; ++i) {
            const int64_t stop = i < nr ? r[i] : nw;
            for (; k < stop; ++k) cand_list[(size_t)w++] = warm_items[(size_t)k];
            ++k;
        }
        return cand_list.data();
    }
    bool rated(int32_t u, int32_t it) const {
        if (u >= n_users) return false;
        if (words_per_user) return (bits[(size_t)u * words_per_user + (it >> 6)] >> (it & 63)) & 1u;
        const int32_t* lo = items.data() + rowptr[u];
        const int32_t* hi = items.data() + rowptr[u + 1];
        return std::binary_search(lo, hi, it);
    }
};

extern "C" crh_sampler* crh_sampler_create(const int32_t* rec_user_host, const int32_t* rec_item_host,
                                           int64_t n_records, int32_t n_users, int32_t n_items_seen) {
    if (!rec_user_host || !rec_item_host || n_records <= 0 || n_users <= 0 || n_items_seen <= 0) {
        crh_set_error("crh_sampler_create: bad arguments");
        return nullptr;
    }
    crh_sampler* s = new (std::nothrow) crh_sampler();
    if (!s) return nullptr;
    s->n_users = n_users;
    s->n_items = n_items_seen;
    s->rec_u.assign(rec_user_host, rec_user_host + n_records);
    s->rec_i.assign(rec_item_host, rec_item_host + n_records);
    s->order.resize(n_records);
    for (int64_t i = 0; i < n_records; ++i) s->order[i] = (int32_t)i;
    s->rec_ui.resize(n_records);
    for (int64_t i = 0; i < n_records; ++i)
        s->rec_ui[i] = (int64_t)(uint32_t)rec_user_host[i] | ((int64_t)(uint32_t)rec_item_host[i] << 32);
    s->rowptr.assign((size_t)n_users + 1, 0);
    for (int64_t r = 0; r < n_records; ++r) {
        if (s->rec_u[r] < 0 || s->rec_u[r] >= n_users || s->rec_i[r] < 0 || s->rec_i[r] >= n_items_seen) {
            crh_set_error("crh_sampler_create: record %lld out of range", (long long)r);
            delete s;
            return nullptr;
        }
        s->rowptr[s->rec_u[r] + 1]++;
    }
    for (int32_t u = 0; u < n_users; ++u) s->rowptr[u + 1] += s->rowptr[u];
    s->items.resize(n_records);
    std::vector<int64_t> fill(s->rowptr.begin(), s->rowptr.end() - 1);
    for (int64_t r = 0; r < n_records; ++r) s->items[fill[s->rec_u[r]]++] = s->rec_i[r];
    for (int32_t u = 0; u < n_users; ++u) std::sort(s->items.begin() + s->rowptr[u], s->items.begin() + s->rowptr[u + 1]);
    // O(1) rejection test for catalogues whose users x items bitmap fits a few tens of MB (MovieLens: 2.8 MB,
    // CiteULike: 11.8 MB); larger ones keep the binary search in the user's sorted list
    const int64_t wpu = ((int64_t)n_items_seen + 63) / 64;
    if (wpu * n_users * 8 <= (int64_t)48 << 20) {
        s->words_per_user = wpu;
        s->bits.assign((size_t)(wpu * n_users), 0);
        for (int64_t r = 0; r < n_records; ++r)
            s->bits[(size_t)s->rec_u[r] * wpu + (s->rec_i[r] >> 6)] |= (uint64_t)1 << (s->rec_i[r] & 63);
    }
    s->rng.seed(5489u);
    return s;
}

extern "C" void crh_sampler_destroy(crh_sampler* s) {
    if (!s) return;
    if (s->worker.joinable()) {
        {
            // an epoch that is queued or running is allowed to finish first: the worker writes the caller's output
            // arrays until then, so returning earlier would let them be freed under it
            std::unique_lock<std::mutex> lk(s->wm);
            s->wcv.wait(lk, [&] { const int j = s->job.load(); return j == 0 || j == 3; });
            s->shutdown.store(true);
        }
        s->wcv.notify_all();
        s->worker.join();
    }
    delete s;
}

extern "C" int crh_sampler_seed(crh_sampler* s, uint32_t seed) {
    CRH_CHECK_ARG(s, "crh_sampler_seed: NULL sampler");
    s->rng.seed(seed);
    return CRH_OK;
}

extern "C" int crh_sampler_set_state(crh_sampler* s, const uint32_t* key624_host, int pos) {
    CRH_CHECK_ARG(s && key624_host && pos >= 0 && pos <= 624, "crh_sampler_set_state: bad arguments");
    memcpy(s->rng.key, key624_host, sizeof(s->rng.key));
    s->rng.pos = pos;
    s->rng.temper_block();
    return CRH_OK;
}

extern "C" int crh_sampler_get_state(const crh_sampler* s, uint32_t* key624_host, int* pos_host) {
    CRH_CHECK_ARG(s && key624_host && pos_host, "crh_sampler_get_state: bad arguments");
    memcpy(key624_host, s->rng.key, sizeof(s->rng.key));
    *pos_host = s->rng.pos;
    return CRH_OK;
}

// Snapshot / restore of everything an epoch call advances (both generators, the cumulative permutation): lets the
// host sample the NEXT epoch speculatively on a worker thread and take it back if training stops first.
extern "C" int crh_sampler_snapshot(crh_sampler* s) {
    CRH_CHECK_ARG(s, "crh_sampler_snapshot: NULL sampler");
    s->snap_rng = s->rng;
    s->snap_pyrng = s->pyrng;
    s->snap_armed = true;
    s->snap_logged = s->snap_copied = false;
    return CRH_OK;
}

extern "C" int crh_sampler_restore(crh_sampler* s) {
    CRH_CHECK_ARG(s && s->snap_armed, "crh_sampler_restore: no snapshot");
    s->rng = s->snap_rng;
    s->pyrng = s->snap_pyrng;
    if (s->snap_copied) {
        s->ord64_live = s->snap_live;
        if (s->snap_live) s->ord64 = s->snap_ord64;
        else s->order = s->snap_order;
    } else if (s->snap_logged) {
        const int64_t n = (int64_t)s->order.size();
        if (s->ord64_live) crh_sampler::undo_swaps(s->ord64.data(), s->swap_log.data(), n);
        else crh_sampler::undo_swaps(s->order.data(), s->swap_log.data(), n);
        s->snap_logged = false;                                 // (the permutation IS the snapshot again)
    }
    return CRH_OK;
}

namespace {
// The membership test of a batch's negatives against the users x items bitmap, as loops of their own: inside the epoch function
// (a dozen live pointers, the generic `rated` fallback inlined beside it) the same loop reloaded four spilled pointers per test and
// ran at 1.4 ns per test where this one runs at 0.95 (tools/probes/sampler_host_probe.cpp).  Branch-free compaction: the slot is
// written, the count advances by the bit.  One random 64-byte line per test; the line of the test 32 slots ahead is prefetched.
__attribute__((noinline)) int64_t rated_slots_bits(const uint64_t* __restrict__ bits, int64_t wpu, const int32_t* __restrict__ u,
                                                   const int32_t* __restrict__ neg, int64_t lo, int64_t hi, int32_t* __restrict__ chk) {
    constexpr int64_t PF = 32;
    int64_t nc = 0;
    for (int64_t t = lo; t < hi; ++t) {
        if (t + PF < hi) __builtin_prefetch(&bits[(size_t)u[t + PF] * wpu + (neg[t + PF] >> 6)], 0, 3);
        chk[nc] = (int32_t)t;
        nc += (bits[(size_t)u[t] * wpu + (neg[t] >> 6)] >> (neg[t] & 63)) & 1u;
    }
    return nc;
}
// one redraw round: the slots of `chk` take the new draws; those that are rated again are compacted into `nxt`
__attribute__((noinline)) int64_t redraw_round_bits(const uint64_t* __restrict__ bits, int64_t wpu, const int32_t* __restrict__ u,
                                                    int32_t* __restrict__ neg, const int32_t* __restrict__ chk, int64_t nc,
                                                    const int32_t* __restrict__ drawn, int32_t* __restrict__ nxt) {
    int64_t nn = 0;
    for (int64_t q = 0; q < nc; ++q) {
        const int32_t t = chk[q], it = drawn[q];
        neg[t] = it;
        nxt[nn] = t;
        nn += (bits[(size_t)u[t] * wpu + (it >> 6)] >> (it & 63)) & 1u;
    }
    return nn;
}
}  // namespace

// One epoch: all batches concatenated (the last one is short).  Output arrays hold n_records int32.
extern "C" int crh_sampler_epoch(crh_sampler* s, int64_t batch_size, int32_t* user_out_host,
                                 int32_t* pos_out_host, int32_t* neg_out_host) {
    CRH_CHECK_ARG(s && user_out_host && pos_out_host && neg_out_host, "crh_sampler_epoch: NULL pointer");
    CRH_CHECK_ARG(batch_size > 0, "crh_sampler_epoch: batch_size=%lld", (long long)batch_size);
    const int64_t n = (int64_t)s->order.size();
    CRH_CHECK_ARG(n < ((int64_t)1 << 31), "crh_sampler_epoch: more than 2^31-1 records");
    // np.random.shuffle(training_data): for i = n-1 .. 1: j = bounded(i); swap   (utils.py:125).
    // The draws of one key block are consumed through a local window (no generator state in the loops).
    MT19937& g = s->rng;
#ifdef CRH_PROFILE
    static const bool timing = getenv("CRH_SAMPLER_TIMING") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    double t_gather = 0, t_draw = 0, t_reject = 0, t_first = 0;
#endif
    // The gather of the shuffled records rides in the shuffle's shadow: position ii is final once its (accepted) swap is done
    // -- Fisher-Yates from the top never touches it again -- so user / positive item of slot ii are fetched and stored right
    // there (a rejected draw writes the not-yet-final record: the accepted one overwrites it).  The loop is bound by its
    // chain through ii and the swapped words; the extra load is independent of that chain and costs next to nothing, where
    // the separate pass over all n slots was a third of an epoch (random 8-byte reads from a 5 MB table).
    // When user and item ids fit 16 bits the permutation's elements carry the record itself (ord64): two random accesses per draw
    // instead of three.
    const int64_t* __restrict__ recs = s->rec_ui.data();
    const bool wide = s->pack16_ok();
    if (s->snap_armed && (s->snap_logged || wide != s->ord64_live)) s->keep_snapshot_copy();   // second epoch under one snapshot / form changes
    const bool logging = s->snap_armed && !s->snap_copied;
    if (logging) s->swap_log.resize((size_t)n);
    uint32_t* __restrict__ slog = logging ? s->swap_log.data() : nullptr;
    if (wide && !s->ord64_live) {
        s->ord64.resize((size_t)n);
        for (int64_t t = 0; t < n; ++t) {
            const int64_t ui = recs[s->order[t]];
            s->ord64[t] = (int64_t)(uint32_t)s->order[t] | ((int64_t)(((uint32_t)ui << 16) | (uint32_t)(ui >> 32)) << 32);
        }
        s->ord64_live = true;
    }
    const bool avx512 = host_has_avx512();
    auto shuffle = [&](auto WideC) {
        constexpr bool WIDE = decltype(WideC)::value;
        using E = std::conditional_t<WIDE, int64_t, int32_t>;
        E* __restrict__ ord;
        if constexpr (WIDE) ord = s->ord64.data();
        else ord = s->order.data();
        auto emit = [&](int64_t slot, E e) {
            if constexpr (WIDE) {
                const uint32_t pk = (uint32_t)((uint64_t)e >> 32);
                user_out_host[slot] = (int32_t)(pk >> 16);
                pos_out_host[slot] = (int32_t)(pk & 0xffffu);
            } else {
                const int64_t ui = recs[e];
                user_out_host[slot] = (int32_t)(uint32_t)ui;
                pos_out_host[slot] = (int32_t)(ui >> 32);
            }
        };
        // Two phases per window of raw draws.  Phase A settles which draws are accepted and compacts their targets, without
        // touching the permutation; a group of 8 draws does not even ride on the chain through ii: a draw of at most ii - 8 is
        // accepted wherever in the group it stands, one above ii is rejected, and only a draw in between (8 values out of ii)
        // sends the group through the exact walk.  Phase B swaps the accepted draws only: no self swaps, so no store is reloaded
        // by the next iteration, and the targets are known early enough to be prefetched.
        // (tools/probes/sampler_host_probe.cpp: 1.20 -> 0.83 ms per MovieLens epoch against the one-pass walk per raw draw.)
        uint32_t jl[624 + 8];
        constexpr int PF = 16;
        int64_t i = n - 1;
        while (i >= 1) {
            const uint32_t mask = 0xffffffffu >> __builtin_clz((uint32_t)i);
            const int64_t stop = (int64_t)(mask >> 1);           // the mask holds while i > mask / 2
            while (i > stop) {
                int avail;
                const uint32_t* __restrict__ w = g.window(avail);
                int used = 0;
                int64_t ii = i, m = 0;
                while (used < avail && ii > stop) {
                    if (avx512) {
                        used += shuffle_accept_avx512(w + used, avail - used, mask, stop, ii, jl, m);
                        if (!(used < avail && ii > stop)) break;
                    } else if (used + 8 <= avail && ii - stop >= 8) {
                        const uint32_t lo = (uint32_t)(ii - 8), hi = (uint32_t)ii;
                        uint32_t v[8];
                        uint32_t amb = 0;
                        for (int t = 0; t < 8; ++t) {
                            v[t] = w[used + t] & mask;
                            amb |= (uint32_t)(v[t] > lo) & (uint32_t)(v[t] <= hi);
                        }
                        if (__builtin_expect(amb == 0, 1)) {
                            int64_t mm = m;
                            for (int t = 0; t < 8; ++t) {
                                jl[mm] = v[t];
                                mm += v[t] <= lo;
                            }
                            ii -= mm - m;
                            m = mm;
                            used += 8;
                            continue;
                        }
                    }
                    const uint32_t v = w[used++] & mask;
                    const bool ok = v <= (uint32_t)ii;
                    jl[m] = v;
                    m += ok;
                    ii -= ok;
                }
                g.pos += used;
                if (slog) memcpy(slog + (n - 1 - i), jl, (size_t)m * sizeof(uint32_t));
                for (int64_t t = 0; t < m; ++t) {
                    if (t + PF < m) __builtin_prefetch(&ord[jl[t + PF]], 1, 3);
                    const int64_t pi = i - t, j = (int64_t)jl[t];
                    const E a = ord[pi], b = ord[j];
                    ord[pi] = b;
                    ord[j] = a;
                    emit(pi, b);
                }
                i = ii;
            }
        }
        emit(0, ord[0]);                                          // slot 0 is never the top of a swap
    };
    if (wide) shuffle(std::true_type{});
    else shuffle(std::false_type{});
    if (logging) s->snap_logged = true;
#ifdef CRH_PROFILE
    const auto tp1 = std::chrono::steady_clock::now();
#endif
    const uint32_t imax = (uint32_t)(s->n_items - 1);
    const uint32_t imask = imax ? 0xffffffffu >> __builtin_clz(imax) : 0u;
    // masked-rejection draws of `cnt` item ids, again walked per raw draw: write, then advance only if accepted
    auto draw_items = [&](int32_t* __restrict__ dst, int64_t cnt) {
        if (imax == 0) {                                     // one item: numpy returns 0 without drawing
            for (int64_t q = 0; q < cnt; ++q) dst[q] = 0;
            return;
        }
        int64_t w_ = 0;
        while (w_ < cnt) {
            int avail;
            const uint32_t* __restrict__ w = g.window(avail);
            int used = avx512 ? draw_items_avx512(w, avail, imask, imax, dst, cnt, w_) : 0;
            while (used < avail && w_ < cnt) {
                const uint32_t v = w[used++] & imask;
                dst[w_] = (int32_t)v;
                w_ += v <= imax;
            }
            g.pos += used;
        }
    };
    s->check.resize((size_t)std::min(batch_size, n) + 1);
    s->next_check.resize(s->check.size());
    s->redraw.resize(s->check.size());
    const bool use_bits = s->words_per_user != 0;
    const uint64_t* __restrict__ bits = s->bits.data();
    const int64_t wpu = s->words_per_user;
    auto is_rated = [&](int32_t u, int32_t it) -> bool { return s->rated(u, it); };     // (record sets too large for a bitmap)
    for (int64_t lo = 0; lo < n; lo += batch_size) {
        const int64_t hi = std::min(lo + batch_size, n);
#ifdef CRH_PROFILE
        const auto tb0 = std::chrono::steady_clock::now();
#endif
#ifdef CRH_PROFILE
        const auto tb1 = std::chrono::steady_clock::now();
#endif
        // first round over the whole batch without materialising the slot list (utils.py:141-153)
        draw_items(neg_out_host + lo, hi - lo);
#ifdef CRH_PROFILE
        const auto tb2 = std::chrono::steady_clock::now();
        t_gather += std::chrono::duration<double>(tb1 - tb0).count();
        t_draw += std::chrono::duration<double>(tb2 - tb1).count();
#endif
        int32_t* chk = s->check.data();
        int64_t nc = 0;
        if (use_bits) nc = rated_slots_bits(bits, wpu, user_out_host, neg_out_host, lo, hi, chk);
        else
            for (int64_t t = lo; t < hi; ++t) {              // compaction without a branch on the outcome
                chk[nc] = (int32_t)t;
                nc += is_rated(user_out_host[t], neg_out_host[t]);
            }
#ifdef CRH_PROFILE
        const auto tb3 = std::chrono::steady_clock::now();
        t_first += std::chrono::duration<double>(tb3 - tb2).count();
#endif
        while (nc > 0) {                                     // redraw only the rejected slots, in slot order
            draw_items(s->redraw.data(), nc);
            int32_t* nxt = s->next_check.data();
            int64_t nn = 0;
            if (use_bits) nn = redraw_round_bits(bits, wpu, user_out_host, neg_out_host, chk, nc, s->redraw.data(), nxt);
            else
                for (int64_t q = 0; q < nc; ++q) {
                    const int32_t t = chk[q];
                    neg_out_host[t] = s->redraw[q];
                    nxt[nn] = t;
                    nn += is_rated(user_out_host[t], neg_out_host[t]);
                }
            s->check.swap(s->next_check);
            chk = s->check.data();
            nc = nn;
        }
#ifdef CRH_PROFILE
        t_reject += std::chrono::duration<double>(std::chrono::steady_clock::now() - tb2).count();
#endif
    }
#ifdef CRH_PROFILE
    if (timing)
        fprintf(stderr, "[crh sampler] shuffle %.3f ms, gather %.3f ms, first draws %.3f ms, membership + redraws %.3f ms (first pass %.3f)\n",
                std::chrono::duration<double>(tp1 - tp0).count() * 1e3, t_gather * 1e3, t_draw * 1e3, t_reject * 1e3, t_first * 1e3);
#endif
    return CRH_OK;
}

// One epoch in the background (the trainers sample epoch e+1 while the GPU trains epoch e): _async queues the job for
// the sampler's worker thread and returns at once; _wait blocks until it is finished and returns crh_sampler_epoch's
// code (its message through crh_last_error).  The output arrays (pinned host memory in coldrec_amd: the upload that
// follows is then asynchronous) belong to the worker between the two calls, and so does the sampler: no other
// crh_sampler_* call on it in between.
namespace {
void sampler_worker(crh_sampler* s) {
    for (;;) {
        int st = s->job.load(std::memory_order_acquire);
        if (st != 1 && !s->shutdown.load(std::memory_order_acquire)) {
            // stay hot for ~30 ms, then sleep until a job (or the shutdown) arrives
            const auto t0 = std::chrono::steady_clock::now();
            while ((st = s->job.load(std::memory_order_acquire)) != 1 && !s->shutdown.load(std::memory_order_acquire)) {
                __builtin_ia32_pause();
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(30)) {
                    std::unique_lock<std::mutex> lk(s->wm);
                    s->wcv.wait(lk, [&] { return s->job.load() == 1 || s->shutdown.load(); });
                    st = s->job.load();
                    break;
                }
            }
        }
        if (st != 1) return;            // shut down (destroy never sets the flag while a job is queued or running)
        s->job.store(2, std::memory_order_release);
        // the snapshot is taken HERE, on the core that is about to shuffle: copied by the calling thread, the 2.6 MB
        // permutation would sit in that core's cache and every line of it would have to migrate back (measured at
        // MovieLens size: 4.4 ms per epoch instead of 2.7)
        s->job_rc = s->job_snapshot ? crh_sampler_snapshot(s) : CRH_OK;
        if (s->job_rc == CRH_OK) s->job_rc = crh_sampler_epoch(s, s->job_batch, s->job_u, s->job_p, s->job_n);
        s->job_err = s->job_rc != CRH_OK ? crh_last_error() : "";
        {
            std::lock_guard<std::mutex> lk(s->wm);
            s->job.store(3, std::memory_order_release);
        }
        s->wcv.notify_all();
    }
}
}  // namespace

extern "C" int crh_sampler_epoch_async(crh_sampler* s, int64_t batch_size, int32_t* user_out_host, int32_t* pos_out_host,
                                       int32_t* neg_out_host, int snapshot_first) {
    CRH_CHECK_ARG(s && user_out_host && pos_out_host && neg_out_host, "crh_sampler_epoch_async: NULL pointer");
    const int st = s->job.load();
    CRH_CHECK_ARG(st == 0, "crh_sampler_epoch_async: an epoch is already in flight (call crh_sampler_epoch_wait first)");
    s->job_batch = batch_size;
    s->job_snapshot = snapshot_first;
    s->job_u = user_out_host;
    s->job_p = pos_out_host;
    s->job_n = neg_out_host;
    if (!s->worker.joinable()) s->worker = std::thread(sampler_worker, s);
    {
        std::lock_guard<std::mutex> lk(s->wm);
        s->job.store(1, std::memory_order_release);
    }
    s->wcv.notify_all();
    return CRH_OK;
}

extern "C" int crh_sampler_epoch_wait(crh_sampler* s) {
    CRH_CHECK_ARG(s, "crh_sampler_epoch_wait: NULL sampler");
    int st = s->job.load(std::memory_order_acquire);
    CRH_CHECK_ARG(st != 0, "crh_sampler_epoch_wait: no epoch in flight");
    // SLEEP until the worker is done: a waiter that spins (or yields in a loop) on the hyperthread next to the worker
    // takes a third of its speed (measured: 4.2 ms per epoch with a spinning waiter, 2.6 ms with a sleeping one)
    if ((st = s->job.load(std::memory_order_acquire)) != 3) {
        std::unique_lock<std::mutex> lk(s->wm);
        s->wcv.wait(lk, [&] { return s->job.load() == 3; });
    }
    const int rc = s->job_rc;
    if (rc != CRH_OK) crh_set_error("%s", s->job_err.c_str());
    s->job.store(0, std::memory_order_release);
    return rc;
}

// ------------------------------------------------------------------------------------------------------------
// The other samplers of util/utils.py.  All of them walk the shuffled records one by one, so
// the batch size only cuts the output (except next_batch_cgrc, whose item set is per batch).
// ------------------------------------------------------------------------------------------------------------

extern "C" int crh_sampler_set_catalogue(crh_sampler* s, int32_t n_users_seen, const uint8_t* item_is_cold_host) {
    CRH_CHECK_ARG(s && n_users_seen > 0, "crh_sampler_set_catalogue: bad arguments");
    const int64_t n = (int64_t)s->rec_u.size();
    s->n_users_seen = n_users_seen;
    // dict order of training_set_u[user]: distinct items by first appearance in the (unshuffled) records
    std::vector<int64_t> cnt((size_t)s->n_users + 1, 0);
    std::vector<uint8_t> first(n, 0);
    {
        std::vector<int64_t> idx(n);
        for (int64_t r = 0; r < n; ++r) idx[r] = r;
        std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) {
            return s->rec_u[a] != s->rec_u[b] ? s->rec_u[a] < s->rec_u[b] : s->rec_i[a] < s->rec_i[b];
        });
        for (int64_t t = 0; t < n; ++t) {
            const int64_t r = idx[t];
            if (t == 0 || s->rec_u[idx[t - 1]] != s->rec_u[r] || s->rec_i[idx[t - 1]] != s->rec_i[r]) {
                first[r] = 1;                               // stable sort: the earliest record of each (user, item)
                cnt[s->rec_u[r] + 1]++;
            }
        }
    }
    s->item_users.assign((size_t)s->n_items, 0);
    for (int64_t r = 0; r < n; ++r)
        if (first[r]) s->item_users[s->rec_i[r]]++;
    s->first_ptr.assign(cnt.begin(), cnt.end());
    for (int32_t u = 0; u < s->n_users; ++u) s->first_ptr[u + 1] += s->first_ptr[u];
    s->first_items.resize((size_t)s->first_ptr[s->n_users]);
    std::vector<int64_t> fill(s->first_ptr.begin(), s->first_ptr.end() - 1);
    for (int64_t r = 0; r < n; ++r)
        if (first[r]) s->first_items[(size_t)fill[s->rec_u[r]]++] = s->rec_i[r];
    // candidate pool: items in id order that are not cold; per user the pool ranks of its training items
    std::vector<int32_t> rank((size_t)s->n_items, -1);
    s->warm_items.clear();
    for (int32_t it = 0; it < s->n_items; ++it)
        if (!item_is_cold_host || !item_is_cold_host[it]) {
            rank[it] = (int32_t)s->warm_items.size();
            s->warm_items.push_back(it);
        }
    s->rw_ptr.assign((size_t)s->n_users + 1, 0);
    s->rw_rank.clear();
    for (int32_t u = 0; u < s->n_users; ++u) {
        int32_t prev = -1;
        for (int64_t e = s->rowptr[u]; e < s->rowptr[u + 1]; ++e) {      // ascending ids, duplicates adjacent
            const int32_t it = s->items[e];
            if (it != prev && rank[it] >= 0) s->rw_rank.push_back(rank[it]);
            prev = it;
        }
        s->rw_ptr[u + 1] = (int64_t)s->rw_rank.size();
    }
    s->pyrng.seed(5489u);
    s->catalogue = true;
    return CRH_OK;
}

extern "C" int crh_sampler_set_py_state(crh_sampler* s, const uint32_t* key624_host, int pos) {
    CRH_CHECK_ARG(s && key624_host && pos >= 0 && pos <= 624, "crh_sampler_set_py_state: bad arguments");
    memcpy(s->pyrng.key, key624_host, sizeof(s->pyrng.key));
    s->pyrng.pos = pos;
    s->pyrng.temper_block();
    return CRH_OK;
}

extern "C" int crh_sampler_get_py_state(const crh_sampler* s, uint32_t* key624_host, int* pos_host) {
    CRH_CHECK_ARG(s && key624_host && pos_host, "crh_sampler_get_py_state: bad arguments");
    memcpy(key624_host, s->pyrng.key, sizeof(s->pyrng.key));
    *pos_host = s->pyrng.pos;
    return CRH_OK;
}

extern "C" int64_t crh_sampler_min_candidates(const crh_sampler* s) {
    if (!s || !s->catalogue) return -1;
    int64_t m = (int64_t)s->warm_items.size();
    for (int32_t u = 0; u < s->n_users; ++u)
        if (s->rowptr[u + 1] > s->rowptr[u]) m = std::min(m, s->n_candidates(u));
    return m;
}

namespace {
// random.shuffle(training_data): for i = n-1 .. 1: j = _randbelow(i + 1); swap
void py_shuffle(crh_sampler* s) {
    s->keep_snapshot_copy();
    for (int64_t i = (int64_t)s->order.size() - 1; i >= 1; --i)
        std::swap(s->order[i], s->order[s->pyrng.py_randbelow((uint32_t)(i + 1))]);
}
// neg_user = choice(user_list) until it is not in training_set_i[item]      (utils.py:183-185, 263-265)
inline int32_t draw_neg_user(crh_sampler* s, int32_t item) {
    int32_t v;
    do { v = (int32_t)s->pyrng.py_randbelow((uint32_t)s->n_users_seen); } while (s->rated(v, item));
    return v;
}
}  // namespace

// util/utils.py:160-188.  neg_* hold n_records * n_negs entries (the reference appends n_negs per record).
extern "C" int crh_sampler_epoch_lara(crh_sampler* s, int32_t n_negs, int32_t* user_out_host, int32_t* item_out_host,
                                      int32_t* neg_user_out_host, int32_t* neg_item_out_host) {
    if (s) s->sync_order();           // (the pairwise sampler may hold the permutation in its wide form)
    CRH_CHECK_ARG(s && s->catalogue, "crh_sampler_epoch_lara: call crh_sampler_set_catalogue first");
    CRH_CHECK_ARG(user_out_host && item_out_host && neg_user_out_host && neg_item_out_host && n_negs >= 0,
                  "crh_sampler_epoch_lara: bad arguments");
    py_shuffle(s);
    const int64_t n = (int64_t)s->order.size();
    for (int64_t t = 0; t < n; ++t) {
        const int32_t u = s->rec_u[s->order[t]], it = s->rec_i[s->order[t]];
        user_out_host[t] = u;
        item_out_host[t] = it;
        // the reference's rejection loops never end for these records; fail instead of hanging the host
        CRH_CHECK_ARG(n_negs == 0 || (s->first_ptr[u + 1] - s->first_ptr[u] < s->n_items && s->item_users[it] < s->n_users_seen),
                      "crh_sampler_epoch_lara: user %d rated every item or item %d is rated by every user", (int)u, (int)it);
        for (int32_t m = 0; m < n_negs; ++m) {
            int32_t v;
            do { v = (int32_t)s->pyrng.py_randbelow((uint32_t)s->n_items); } while (s->rated(u, v));
            neg_item_out_host[t * n_negs + m] = v;
            neg_user_out_host[t * n_negs + m] = draw_neg_user(s, it);
        }
    }
    return CRH_OK;
}

// util/utils.py:191-233.  item_out (n_records, 1 + n_negs): the positive, then random.sample(candidates, n_negs).
// sample_setsize = 21 (+ 4 ** ceil(log(3 * n_negs, 4)) when n_negs > 5), computed by the caller with Python's
// own math so that the pool / selected-set switch of random.sample falls where CPython puts it.
extern "C" int crh_sampler_epoch_clcrec(crh_sampler* s, int32_t n_negs, int64_t sample_setsize, int32_t* user_out_host,
                                        int32_t* item_out_host) {
    if (s) s->sync_order();           // (the pairwise sampler may hold the permutation in its wide form)
    CRH_CHECK_ARG(s && s->catalogue, "crh_sampler_epoch_clcrec: call crh_sampler_set_catalogue first");
    CRH_CHECK_ARG(user_out_host && item_out_host && n_negs >= 0 && sample_setsize >= 21,
                  "crh_sampler_epoch_clcrec: bad arguments");
    CRH_CHECK_ARG(!s->warm_items.empty(), "next_batch_pairwise_CLCRec: warm-item negative pool is empty; check cold_item split.");
    py_shuffle(s);
    const int64_t n = (int64_t)s->order.size(), w = 1 + n_negs;
    for (int64_t t = 0; t < n; ++t) {
        const int32_t u = s->rec_u[s->order[t]];
        user_out_host[t] = u;
        int32_t* row = item_out_host + t * w;
        row[0] = s->rec_i[s->order[t]];
        const int64_t nc = s->n_candidates(u);
        CRH_CHECK_ARG(nc >= n_negs, "next_batch_pairwise_CLCRec: user has only %lld warm negatives available but n_negs=%d.",
                      (long long)nc, (int)n_negs);
        if (nc <= sample_setsize) {                       // pool = list(population); result[i] = pool[j]; pool[j] = pool[n-i-1]
            s->picked.resize((size_t)nc);
            for (int64_t c = 0; c < nc; ++c) s->picked[c] = s->candidate(u, c);
            for (int32_t i = 0; i < n_negs; ++i) {
                const uint32_t j = s->pyrng.py_randbelow((uint32_t)(nc - i));
                row[1 + i] = s->picked[j];
                s->picked[j] = s->picked[nc - i - 1];
            }
        } else {                                          // selected-set variant: redraw positions already taken
            if (s->stamp.size() < (size_t)nc) s->stamp.resize(s->warm_items.size(), 0);
            if (++s->stamp_now == 0) { std::fill(s->stamp.begin(), s->stamp.end(), 0u); s->stamp_now = 1; }
            const int32_t* list = (int64_t)n_negs * 16 > nc ? s->materialise(u) : nullptr;
            for (int32_t i = 0; i < n_negs; ++i) {
                uint32_t j;
                do { j = s->pyrng.py_randbelow((uint32_t)nc); } while (s->stamp[j] == s->stamp_now);
                s->stamp[j] = s->stamp_now;
                row[1 + i] = list ? list[j] : s->candidate(u, j);
            }
        }
    }
    return CRH_OK;
}

// util/utils.py:237-300.  pos_items_out (n, P) from NumPy's stream; neg_items_out (n, P, N) and self_neg_out (n, S)
// from CPython's.
extern "C" int crh_sampler_epoch_ccfcrec(crh_sampler* s, int32_t positive_number, int32_t negative_number,
                                         int32_t self_neg_number, int32_t* user_out_host, int32_t* item_out_host,
                                         int32_t* neg_user_out_host, int32_t* pos_items_out_host,
                                         int32_t* neg_items_out_host, int32_t* self_neg_out_host) {
    if (s) s->sync_order();           // (the pairwise sampler may hold the permutation in its wide form)
    CRH_CHECK_ARG(s && s->catalogue, "crh_sampler_epoch_ccfcrec: call crh_sampler_set_catalogue first");
    CRH_CHECK_ARG(user_out_host && item_out_host && neg_user_out_host && pos_items_out_host && neg_items_out_host &&
                  self_neg_out_host && positive_number >= 0 && negative_number >= 0 && self_neg_number >= 0,
                  "crh_sampler_epoch_ccfcrec: bad arguments");
    CRH_CHECK_ARG(!s->warm_items.empty(),
                  "next_batch_pairwise_CCFCRec: warm-item candidate pool is empty; check the cold item split.");
    py_shuffle(s);
    const int64_t n = (int64_t)s->order.size(), P = positive_number, PN = P * negative_number, S = self_neg_number;
    for (int64_t t = 0; t < n; ++t) {
        const int32_t u = s->rec_u[s->order[t]], it = s->rec_i[s->order[t]];
        user_out_host[t] = u;
        item_out_host[t] = it;
        CRH_CHECK_ARG(s->item_users[it] < s->n_users_seen,
                      "crh_sampler_epoch_ccfcrec: item %d is rated by every user (no negative user exists)", (int)it);
        neg_user_out_host[t] = draw_neg_user(s, it);
        // np.random.choice(list(training_set_u[user]), P, replace=True) = list[randint(0, len, P)]
        const int32_t* pos = s->first_items.data() + s->first_ptr[u];
        const uint32_t np1 = (uint32_t)(s->first_ptr[u + 1] - s->first_ptr[u]) - 1u;
        for (int64_t m = 0; m < P; ++m) pos_items_out_host[t * P + m] = pos[s->rng.bounded(np1)];
        const int64_t nc = s->n_candidates(u);
        CRH_CHECK_ARG(nc > 0, "next_batch_pairwise_CCFCRec: user %d has no warm negative items available after "
                              "excluding cold items and training positives.", (int)u);
        if ((PN + S) * 16 > nc) {
            const int32_t* list = s->materialise(u);
            for (int64_t m = 0; m < PN; ++m) neg_items_out_host[t * PN + m] = list[s->pyrng.py_randbelow((uint32_t)nc)];
            for (int64_t m = 0; m < S; ++m) self_neg_out_host[t * S + m] = list[s->pyrng.py_randbelow((uint32_t)nc)];
        } else {
            for (int64_t m = 0; m < PN; ++m) neg_items_out_host[t * PN + m] = s->candidate(u, s->pyrng.py_randbelow((uint32_t)nc));
            for (int64_t m = 0; m < S; ++m) self_neg_out_host[t * S + m] = s->candidate(u, s->pyrng.py_randbelow((uint32_t)nc));
        }
    }
    return CRH_OK;
}

// util/utils.py:303-336.  Per batch the shared item set B = positives U up to ranking_neg_per_user non-rated draws
// per record (at most 50x that many tries), returned in CPython's list(set) order: bset_out[bset_ptr[b] ..
// bset_ptr[b+1]).  bset_ptr_out holds n_batches + 1 offsets; capacity = entries bset_out can take.
extern "C" int crh_sampler_epoch_cgrc(crh_sampler* s, int64_t batch_size, int32_t ranking_neg_per_user,
                                      int32_t* user_out_host, int32_t* item_out_host, int64_t* bset_ptr_out_host,
                                      int32_t* bset_out_host, int64_t capacity) {
    if (s) s->sync_order();           // (the pairwise sampler may hold the permutation in its wide form)
    CRH_CHECK_ARG(s && user_out_host && item_out_host && bset_ptr_out_host && bset_out_host && batch_size > 0 &&
                  ranking_neg_per_user >= 0, "crh_sampler_epoch_cgrc: bad arguments");
    const int64_t n = (int64_t)s->order.size();
    s->keep_snapshot_copy();
    for (int64_t i = n - 1; i >= 1; --i) std::swap(s->order[i], s->order[s->rng.bounded((uint32_t)i)]);   // np.random.shuffle
    const uint32_t imax = (uint32_t)(s->n_items - 1);
    const int64_t max_tries = (int64_t)ranking_neg_per_user * 50;
    int64_t w = 0, b = 0;
    bset_ptr_out_host[0] = 0;
    for (int64_t lo = 0; lo < n; lo += batch_size, ++b) {
        const int64_t hi = std::min(lo + batch_size, n);
        s->bset.clear();
        for (int64_t t = lo; t < hi; ++t) {
            user_out_host[t] = s->rec_u[s->order[t]];
            item_out_host[t] = s->rec_i[s->order[t]];
            s->bset.add(item_out_host[t]);
        }
        for (int64_t t = lo; t < hi; ++t) {
            int64_t added = 0, tries = 0;
            while (added < ranking_neg_per_user && tries < max_tries) {
                ++tries;
                const int32_t j = (int32_t)s->rng.bounded(imax);
                if (!s->rated(user_out_host[t], j)) { s->bset.add(j); ++added; }
            }
        }
        CRH_CHECK_ARG(w + (int64_t)s->bset.used <= capacity, "crh_sampler_epoch_cgrc: item-set buffer too small");
        for (int32_t k : s->bset.tab)
            if (k >= 0) bset_out_host[w++] = k;
        bset_ptr_out_host[b + 1] = w;
    }
    return CRH_OK;
}
