/* Host helper of the batched samplers: draws from MANY independent numpy legacy streams in one call.
 *
 * The reference samplers draw from numpy's legacy global generator, one stream per MPI rank
 * (pyhmc/hmc.py:43,61: np.random.seed(seed + rank); :146 randn(n); :193 rand()).  The batched samplers keep one
 * numpy RandomState per chain to reproduce those streams; with thousands of chains finishing a trajectory in the same
 * step, thousands of Python-level rs.randn(n) / rs.rand() calls (microseconds each) become the bottleneck of the
 * sampler.  Here the same draws are made by one C loop over the chains that advances each stream's Mersenne-Twister
 * state IN PLACE inside numpy (BitGenerator.ctypes.state_address -> struct { uint32 key[624]; int pos; }), so that the
 * values are bit-identical to rs.rand() / rs.randn() and the streams stay usable from Python (rs.randint, get_state):
 *   rand   = ((a >> 5) * 2^26 + (b >> 6)) / 2^53 from two 32-bit outputs    (numpy mt19937_next_double)
 *   randn  = legacy_gauss: Marsaglia polar method with the second deviate cached (numpy legacy-distributions.c);
 *            the cache (has_gauss, gauss) of every stream is kept by the caller in two arrays.
 * Build: gcc -O2 -fPIC -shared rngbatch.c -o librngbatch.so -lm   (rfsurfhmc_amd/build.py) */
#include <math.h>
#include <stdint.h>

#define MT_N 624
#define MT_M 397
typedef struct { uint32_t key[MT_N]; int pos; } mt_state;     /* numpy/random/src/mt19937/mt19937.h */

static void mt_gen(mt_state* s) {
    uint32_t y;
    int i;
    for (i = 0; i < MT_N - MT_M; i++) {
        y = (s->key[i] & 0x80000000u) | (s->key[i + 1] & 0x7fffffffu);
        s->key[i] = s->key[i + MT_M] ^ (y >> 1) ^ (-(y & 1) & 0x9908b0dfu);
    }
    for (; i < MT_N - 1; i++) {
        y = (s->key[i] & 0x80000000u) | (s->key[i + 1] & 0x7fffffffu);
        s->key[i] = s->key[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(y & 1) & 0x9908b0dfu);
    }
    y = (s->key[MT_N - 1] & 0x80000000u) | (s->key[0] & 0x7fffffffu);
    s->key[MT_N - 1] = s->key[MT_M - 1] ^ (y >> 1) ^ (-(y & 1) & 0x9908b0dfu);
    s->pos = 0;
}

static inline uint32_t mt_u32(mt_state* s) {
    uint32_t y;
    if (s->pos == MT_N) mt_gen(s);
    y = s->key[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static inline double mt_double(mt_state* s) {
    const int32_t a = mt_u32(s) >> 5, b = mt_u32(s) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

static inline double gauss_one(mt_state* st, int32_t* has, double* cache) {
    if (*has) {
        const double t = *cache;
        *has = 0; *cache = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * mt_double(st) - 1.0;
        x2 = 2.0 * mt_double(st) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    *cache = f * x1; *has = 1;
    return f * x2;
}

/* out[i][0..n) = n standard normals from stream idx[i]; states[c] / has_gauss[c] / gauss[c] belong to stream c */
void rngbatch_randn(const uint64_t* states, int32_t* has_gauss, double* gauss, const int64_t* idx, int64_t nidx,
                    int64_t n, double* out) {
    for (int64_t i = 0; i < nidx; i++) {
        const int64_t c = idx[i];
        mt_state* st = (mt_state*)(uintptr_t)states[c];
        double* o = out + i * n;
        if (i + 1 < nidx) {     /* the streams are separate heap objects: fetch the next one's 2.5 KB while this one works */
            const char* nx = (const char*)(uintptr_t)states[idx[i + 1]];
            for (int b = 0; b < (int)sizeof(mt_state); b += 64) __builtin_prefetch(nx + b, 1, 1);
        }
        for (int64_t k = 0; k < n; k++) o[k] = gauss_one(st, &has_gauss[c], &gauss[c]);
    }
}

/* out[i] = one uniform deviate in [0, 1) from stream idx[i] */
void rngbatch_rand(const uint64_t* states, const int64_t* idx, int64_t nidx, double* out) {
    for (int64_t i = 0; i < nidx; i++) out[i] = mt_double((mt_state*)(uintptr_t)states[idx[i]]);
}

/* Snapshot / restore of whole streams (624 key words + position, then the caller-held Gaussian cache): a sampler that
 * draws AHEAD of time for a trajectory still running takes a snapshot first and restores it should the trajectory fail
 * (the reference skips the acceptance draw on its failure paths, pyhmc/hmc.py:156,173,177,179).  buf: nidx * 625 uint32. */
void rngbatch_save(const uint64_t* states, const int64_t* idx, int64_t nidx, uint32_t* buf) {
    for (int64_t k = 0; k < nidx; k++) {
        const mt_state* st = (const mt_state*)(uintptr_t)states[idx[k]];
        uint32_t* b = buf + k * (MT_N + 1);
        for (int i = 0; i < MT_N; i++) b[i] = st->key[i];
        b[MT_N] = (uint32_t)st->pos;
    }
}
void rngbatch_load(const uint64_t* states, const int64_t* idx, int64_t nidx, const uint32_t* buf) {
    for (int64_t k = 0; k < nidx; k++) {
        mt_state* st = (mt_state*)(uintptr_t)states[idx[k]];
        const uint32_t* b = buf + k * (MT_N + 1);
        for (int i = 0; i < MT_N; i++) st->key[i] = b[i];
        st->pos = (int)b[MT_N];
    }
}
