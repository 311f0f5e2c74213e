/*
 * gs_synth.c — deterministic synthetic-scene generator for benches and tests (SURVEY.md §8(d),
 * DESIGN.md §6).  Input generator only: it is neither the product nor the oracle.
 *
 * Counter based: Gaussian i depends only on (seed, i), so any sub-range can be generated
 * independently and in parallel.  h(i,k) = splitmix64(seed ^ (i*64 + k)), u = (h >> 40) * 2^-24.
 *   k 0..2   position   (lerp(-14,14,u0), lerp(-8,8,u1), -lerp(2,26,u2))
 *   k 3..10  four Box-Muller pairs -> z0..z7; rot = normalize(z0..z3) (xyzw),
 *            scale_axis = exp(-3.6 + 0.5 * z4..z6)
 *   k 11..14 colour: rgb = floor(256 u), alpha = 32 + floor(224 u)
 *   k 16..60 SH rest coefficients = lerp(-0.25, 0.25, u)
 * Transcendentals are evaluated in double and rounded once to float.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    float rot[4];
    float pos[3];
    uint8_t color[4];
    float sh[45];
    float scale[3];
} synth_gaussian;

static inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline double uni(uint64_t seed, uint64_t i, uint64_t k) {
    return (double)(splitmix64(seed ^ (i * 64u + k)) >> 40) * (1.0 / 16777216.0);
}

/* OpenMP threads of gs_synth_scene (a launcher may have exported OMP_NUM_THREADS=1) */
void gs_synth_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

void gs_synth_scene(uint64_t seed, uint64_t first, uint64_t count, synth_gaussian *out) {
#pragma omp parallel for schedule(static)
    for (long long j = 0; j < (long long)count; j++) {
        uint64_t i = first + (uint64_t)j;
        synth_gaussian *g = &out[j];
        g->pos[0] = (float)(-14.0 + 28.0 * uni(seed, i, 0));
        g->pos[1] = (float)(-8.0 + 16.0 * uni(seed, i, 1));
        g->pos[2] = (float)(-(2.0 + 24.0 * uni(seed, i, 2)));
        double z[8];
        for (int p = 0; p < 4; p++) {
            double u1 = 1.0 - uni(seed, i, 3 + 2 * p); /* (0,1] */
            double u2 = uni(seed, i, 4 + 2 * p);
            double r = sqrt(-2.0 * log(u1));
            z[2 * p] = r * cos(6.283185307179586 * u2);
            z[2 * p + 1] = r * sin(6.283185307179586 * u2);
        }
        double len = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2] + z[3] * z[3]);
        if (!(len > 1e-12)) { z[0] = z[1] = z[2] = 0.0; z[3] = 1.0; len = 1.0; }
        for (int c = 0; c < 4; c++) g->rot[c] = (float)(z[c] / len);
        for (int c = 0; c < 3; c++) g->scale[c] = (float)exp(-3.6 + 0.5 * z[4 + c]);
        for (int c = 0; c < 3; c++) g->color[c] = (uint8_t)(256.0 * uni(seed, i, 11 + c));
        g->color[3] = (uint8_t)(32.0 + floor(224.0 * uni(seed, i, 14)));
        for (int c = 0; c < 45; c++) g->sh[c] = (float)(-0.25 + 0.5 * uni(seed, i, 16 + c));
    }
}
