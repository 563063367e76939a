// Host-side negative sampler that reproduces the reference's NumPy stream bit for bit.
//
// Replaces util/utils.py:123-157 (next_batch_pairwise): np.random.shuffle(training_data) once per
// epoch (in place, cumulative), then per batch np.random.choice(item_list, n) with rejection of
// the user's training items, redrawing only the rejected slots until none is left.
// NumPy's legacy global RNG is MT19937 (np.random.seed(s) == init_genrand(s)); randint/choice use
// masked rejection over 32-bit draws and shuffle is Fisher-Yates from the top with the same
// bounded draw -- all restated here so a whole epoch of triples is produced
// in a few milliseconds on one host thread, off the GPU's critical path.  Pure host code: no HIP call.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "crh_common.h"

namespace {

struct MT19937 {
    uint32_t key[624];
    int pos;
    void seed(uint32_t s) {
        key[0] = s;
        for (int i = 1; i < 624; ++i) key[i] = 1812433253u * (key[i - 1] ^ (key[i - 1] >> 30)) + (uint32_t)i;
        pos = 624;
    }
    void gen() {
        const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MAT = 0x9908b0dfu;
        int i;
        for (i = 0; i < 624 - 397; ++i) {
            uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
        }
        for (; i < 623; ++i) {
            uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
        }
        uint32_t y = (key[623] & UP) | (key[0] & LO);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
        pos = 0;
    }
    inline uint32_t next() {
        if (pos >= 624) gen();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
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
};

}  // namespace

struct crh_sampler {
    MT19937 rng;
    std::vector<int32_t> rec_u, rec_i;    // training records, internal ids, file order
    std::vector<int32_t> order;           // cumulative permutation of the records
    std::vector<int64_t> rowptr;          // per user: sorted training items (rejection test)
    std::vector<int32_t> items;
    int32_t n_users, n_items;
    std::vector<int32_t> check, next_check;
    std::vector<uint64_t> bits;           // users x items membership bitmap when it is small enough to stay cached
    int64_t words_per_user = 0;
    bool rated(int32_t u, int32_t it) const {
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

extern "C" void crh_sampler_destroy(crh_sampler* s) { delete s; }

extern "C" int crh_sampler_seed(crh_sampler* s, uint32_t seed) {
    CRH_CHECK_ARG(s, "crh_sampler_seed: NULL sampler");
    s->rng.seed(seed);
    return CRH_OK;
}

extern "C" int crh_sampler_set_state(crh_sampler* s, const uint32_t* key624_host, int pos) {
    CRH_CHECK_ARG(s && key624_host && pos >= 0 && pos <= 624, "crh_sampler_set_state: bad arguments");
    memcpy(s->rng.key, key624_host, sizeof(s->rng.key));
    s->rng.pos = pos;
    return CRH_OK;
}

extern "C" int crh_sampler_get_state(const crh_sampler* s, uint32_t* key624_host, int* pos_host) {
    CRH_CHECK_ARG(s && key624_host && pos_host, "crh_sampler_get_state: bad arguments");
    memcpy(key624_host, s->rng.key, sizeof(s->rng.key));
    *pos_host = s->rng.pos;
    return CRH_OK;
}

extern "C" int64_t crh_sampler_num_records(const crh_sampler* s) { return s ? (int64_t)s->order.size() : -1; }

// One epoch: all batches concatenated (the last one is short).  Output arrays hold n_records int32.
extern "C" int crh_sampler_epoch(crh_sampler* s, int64_t batch_size, int32_t* user_out_host,
                                 int32_t* pos_out_host, int32_t* neg_out_host) {
    CRH_CHECK_ARG(s && user_out_host && pos_out_host && neg_out_host, "crh_sampler_epoch: NULL pointer");
    CRH_CHECK_ARG(batch_size > 0, "crh_sampler_epoch: batch_size=%lld", (long long)batch_size);
    const int64_t n = (int64_t)s->order.size();
    CRH_CHECK_ARG(n < ((int64_t)1 << 31), "crh_sampler_epoch: more than 2^31-1 records");
    // np.random.shuffle(training_data): for i = n-1 .. 1: j = bounded(i); swap   (utils.py:125)
    for (int64_t i = n - 1; i >= 1; --i) {
        const int64_t j = (int64_t)s->rng.bounded((uint32_t)i);
        std::swap(s->order[i], s->order[j]);
    }
    const uint32_t imax = (uint32_t)(s->n_items - 1);
    for (int64_t lo = 0; lo < n; lo += batch_size) {
        const int64_t hi = std::min(lo + batch_size, n);
        for (int64_t t = lo; t < hi; ++t) {
            const int64_t r = s->order[t];
            user_out_host[t] = s->rec_u[r];
            pos_out_host[t] = s->rec_i[r];
        }
        // first round over the whole batch without materialising the slot list (utils.py:141-153)
        for (int64_t t = lo; t < hi; ++t) neg_out_host[t] = (int32_t)s->rng.bounded(imax);
        s->check.clear();
        for (int64_t t = lo; t < hi; ++t)
            if (s->rated(user_out_host[t], neg_out_host[t])) s->check.push_back((int32_t)t);
        while (!s->check.empty()) {                      // redraw only the rejected slots, in slot order
            for (int32_t t : s->check) neg_out_host[t] = (int32_t)s->rng.bounded(imax);
            s->next_check.clear();
            for (int32_t t : s->check)
                if (s->rated(user_out_host[t], neg_out_host[t])) s->next_check.push_back(t);
            s->check.swap(s->next_check);
        }
    }
    return CRH_OK;
}
