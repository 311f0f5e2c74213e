// VERDICT r03 #5, measured: is a short row scan cheaper as its own kernel or folded into the tail of the kernel that
// writes the rows?  The product's shape at 1 M: NB <= 256 workgroups each write one column of a 512-row table
// (k_sort_hist), then k_sort_scan_rows_small scans every row (one wave per row) and writes the row totals.
//   A  two kernels        : k_write, k_scan_small              (what the product does)
//   A0 one kernel         : k_write alone                      (A - A0 = what the second launch costs)
//   B  last block scans   : k_write + __threadfence + ticket; the workgroup that draws the last ticket scans all rows
//   C  every block scans  : k_write + release + arrive; all workgroups wait for the last arrival (bounded spin: a wave
//                           that does not see it within 2^22 polls sets an error word and leaves), acquire, then each
//                           scans its share of the rows
// Launched back to back on one stream, `iters` times; microseconds per iteration.  Results are compared with A's.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mb/mb_fold tools/mb/mb_fold.hip && tools/mb/mb_fold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr uint32_t R = 512, WAVE = 64, PER = 4;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < (int)WAVE; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, WAVE);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}

__device__ __forceinline__ uint32_t cell(uint32_t d, uint32_t b, uint32_t salt) { return (d * 7u + b * 13u + salt) & 15u; }

__device__ __forceinline__ void write_column(uint32_t *tab, uint32_t nb, uint32_t salt) {
    for (uint32_t d = threadIdx.x; d < R; d += blockDim.x) tab[(size_t)d * nb + blockIdx.x] = cell(d, blockIdx.x, salt);
}

// one wave scans row r in place (nb <= 256: one step), total out.  COHERENT: the row was written by other XCDs in
// this very kernel -> loads that do not trust this XCD's L2
template <bool COHERENT>
__device__ __forceinline__ void scan_row(uint32_t *tab, uint32_t nb, uint32_t r, uint32_t lane, uint32_t *totals) {
    uint32_t *row = tab + (size_t)r * nb;
    const uint32_t i0 = lane * PER;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        v[k] = 0;
        if (i0 + k < nb) v[k] = COHERENT ? __hip_atomic_load(&row[i0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : row[i0 + k];
        sum += v[k];
    }
    const uint32_t inc = wave_inclusive_scan(sum, lane);
    uint32_t run = inc - sum;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        if (i0 + k < nb) row[i0 + k] = run;
        run += v[k];
    }
    if (lane == 63u) totals[r] = inc;
}

__global__ __launch_bounds__(256) void k_write(uint32_t *tab, uint32_t nb, uint32_t salt) { write_column(tab, nb, salt); }

__global__ __launch_bounds__(256) void k_scan_small(uint32_t *tab, uint32_t nb, uint32_t *totals) {
    const uint32_t lane = threadIdx.x & 63u, r = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r < R) scan_row<false>(tab, nb, r, lane, totals);
}

// A with the row length read from DEVICE memory first (the product's kernels learn V and D that way: the counts never
// return to the host inside a frame) — one more dependent round trip in front of the row loads
__global__ __launch_bounds__(256) void k_write_dev(uint32_t *tab, const uint32_t *nb_dev, uint32_t salt) {
    const uint32_t nb = *nb_dev;
    if (blockIdx.x < nb) write_column(tab, nb, salt);
}
__global__ __launch_bounds__(256) void k_scan_small_dev(uint32_t *tab, const uint32_t *nb_dev, uint32_t *totals) {
    const uint32_t nb = *nb_dev;
    const uint32_t lane = threadIdx.x & 63u, r = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r < R) scan_row<false>(tab, nb, r, lane, totals);
}

// B: the workgroup that draws the last ticket scans everything
__global__ __launch_bounds__(256) void k_write_last_scans(uint32_t *tab, uint32_t nb, uint32_t salt, uint32_t *totals,
                                                          uint32_t *ticket) {
    __shared__ uint32_t s_last;
    write_column(tab, nb, salt);
    __threadfence();                     // release: this workgroup's column is visible device-wide
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == nb - 1u) ? 1u : 0u;
        if (t == nb - 1u) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    for (uint32_t r = wid; r < R; r += 4u) scan_row<true>(tab, nb, r, lane, totals);
}

// C: everybody waits for the last arrival (bounded), then scans a share
__global__ __launch_bounds__(256) void k_write_all_scan(uint32_t *tab, uint32_t nb, uint32_t salt, uint32_t *totals,
                                                        uint32_t *arrive, uint32_t target, uint32_t *error) {
    __shared__ uint32_t s_ok;
    write_column(tab, nb, salt);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t polls = 0, ok = 1;
        while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++polls > (1u << 22)) { ok = 0; atomicOr(error, 1u); break; }      // exit condition every wave reaches
            __builtin_amdgcn_s_sleep(1);
        }
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    for (uint32_t r = blockIdx.x * 4u + wid; r < R; r += nb * 4u) scan_row<true>(tab, nb, r, lane, totals);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    uint32_t *tab, *totals, *word, *error;
    CK(hipMalloc(&tab, (size_t)R * 256 * 4));
    CK(hipMalloc(&totals, R * 4));
    CK(hipMalloc(&word, 4));
    CK(hipMalloc(&error, 4));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t sizes[] = {64, 173, 245};
    uint32_t *nb_dev;
    CK(hipMalloc(&nb_dev, 4));
    for (uint32_t nb : sizes) {
        // the same pair with the row length in device memory, written by a kernel-ordered copy like V and D are
        {
            CK(hipMemcpyAsync(nb_dev, &nb, 4, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; i++) {
                    hipLaunchKernelGGL(k_write_dev, dim3(nb), dim3(256), 0, st, tab, nb_dev, (uint32_t)i);
                    hipLaunchKernelGGL(k_scan_small_dev, dim3(R / 4), dim3(256), 0, st, tab, nb_dev, totals);
                }
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("NB %3u: two kernels, row length from device memory %6.2f us\n", nb, ms * 1e3 / iters);
            }
        }
        std::vector<uint32_t> ref((size_t)R * nb), got((size_t)R * nb), rt(R), gt(R);
        double us[4] = {0, 0, 0, 0};
        bool same[4] = {true, true, true, true};
        for (int variant = 0; variant < 4; variant++) {
            CK(hipMemsetAsync(word, 0, 4, st));
            CK(hipMemsetAsync(error, 0, 4, st));
            uint32_t launches = 0;
            auto run = [&](uint32_t salt) {
                switch (variant) {
                case 0:
                    hipLaunchKernelGGL(k_write, dim3(nb), dim3(256), 0, st, tab, nb, salt);
                    hipLaunchKernelGGL(k_scan_small, dim3(R / 4), dim3(256), 0, st, tab, nb, totals);
                    break;
                case 1:
                    hipLaunchKernelGGL(k_write, dim3(nb), dim3(256), 0, st, tab, nb, salt);
                    break;
                case 2:
                    hipLaunchKernelGGL(k_write_last_scans, dim3(nb), dim3(256), 0, st, tab, nb, salt, totals, word);
                    break;
                default:
                    launches++;
                    hipLaunchKernelGGL(k_write_all_scan, dim3(nb), dim3(256), 0, st, tab, nb, salt, totals, word, launches * nb, error);
                    break;
                }
            };
            for (int i = 0; i < 20; i++) run(i);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; i++) run(100 + i);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            us[variant] = ms * 1e3 / iters;
            run(7);                                     // one more with a fixed salt for the comparison
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(variant == 0 ? ref.data() : got.data(), tab, (size_t)R * nb * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(variant == 0 ? rt.data() : gt.data(), totals, R * 4, hipMemcpyDeviceToHost));
            if (variant >= 2) same[variant] = got == ref && gt == rt;
            uint32_t err = 0;
            CK(hipMemcpy(&err, error, 4, hipMemcpyDeviceToHost));
            if (err) printf("  variant %d: a wave gave up waiting (error word %u)\n", variant, err);
        }
        printf("NB %3u: A two kernels %6.2f us | A0 write alone %6.2f us (second launch = %5.2f us) | B last block scans %6.2f us%s | "
               "C all wait + share %6.2f us%s\n", nb, us[0], us[1], us[0] - us[1], us[2], same[2] ? "" : " (WRONG RESULT)",
               us[3], same[3] ? "" : " (WRONG RESULT)");
    }
    return 0;
}
