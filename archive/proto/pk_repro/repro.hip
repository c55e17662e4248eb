// Stand-alone launch of the weight-gradient workgroup body (mpg_amd/csrc/mlp_wgrad.h) on synthetic stashes: does the packed-FMA form
// of its thin block (-DMPG_AB_PKFMA) lose products outside the library?  Every launch repeats the same work on the same inputs; any
// launch whose slabs differ bit for bit from the first one's is a failure (the shipped form is bit-identical over 20 000 launches).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I ../../../include -I ../../../mpg_amd/csrc -DMPG_AB_PKFMA [-DV_...] \
//         -o repro repro.hip && ./repro [launches]
// Shrinking switches (see run.sh for the table of outcomes):
//   -DV_JOBS=n        jobs in the launch (3: two critics <8,1> + the policy <6,2>, the bench step's launch; 1: one critic)
//   -DV_ALL_A         all jobs are of the critic type <8,1> (no second code path co-resident)
//   -DV_ROWS=r        rows per job (4096)
//   -DMPG_AB_WG_NOMFMA  (mlp_wgrad.h) no matrix loop
//   -DV_ROLES=1|2     (with -DV_ALL_A) two workgroups per (chunk, slice): one does the thin pieces only, one the matrix loop only;
//                     1: the thin ones are dispatched first, 2: the matrix ones
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "mlp_wgrad.h"

#ifndef V_JOBS
#define V_JOBS 3
#endif
#ifndef V_NEIGHBOR_ITERS
#define V_NEIGHBOR_ITERS 3000
#endif
#ifndef V_ROWS
#define V_ROWS 4096
#endif

using namespace mlp;
#ifdef V_ROLES
#define GRID_MULT 2
#else
#define GRID_MULT 1
#endif

struct Multi {
    int n_jobs;
    WgradArgs a[3];
    int type[3];
    int chunk_off[4];
};

__global__ void __launch_bounds__(NTHREAD, V_BOUNDS) k_repro(const Multi m) {
    constexpr int NQA = wgrad_nq<8, 1>(), NQB = wgrad_nq<6, 2>();
    __shared__ __attribute__((aligned(16))) float sRed[NWAVE * (NQA > NQB ? NQA : NQB) * 64];
    int gchunk, sl;
#ifdef V_ROLES      // first half of the grid: the thin pieces only (the packed FMAs); second half: the matrix loops only
    const int nb = gridDim.x >> 1;
    const bool thin_role = V_ROLES == 1 ? (int)blockIdx.x < nb : (int)blockIdx.x >= nb;
    wgrad_map((int)blockIdx.x % nb, nb >> 3, gchunk, sl);
#else
    wgrad_map(blockIdx.x, gridDim.x >> 3, gchunk, sl);
#endif
    int j = 0;
    while (j + 1 < m.n_jobs && gchunk >= m.chunk_off[j + 1]) ++j;
    const int chunk = gchunk - m.chunk_off[j];
#ifdef V_ROLES
    if (thin_role) wgrad_body<8, 1, 2>(m.a[j], sl, chunk, sRed);
    else {
#if !defined(V_NEIGHBOR)
        wgrad_body<8, 1, 1>(m.a[j], sl, chunk, sRed);
#else
        // a synthetic neighbour instead of the matrix loop: which kind of activity beside the packed FMAs loses their products?
        const int lane = threadIdx.x & 63;
        float keep = 0.f;
#if V_NEIGHBOR == 1          // matrix instructions on registers only
        typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
        f16x8_ fa, fb;
        for (int q = 0; q < 8; ++q) { fa[q] = (_Float16)(0.01f * (lane + q)); fb[q] = (_Float16)(0.02f * (lane - q)); }
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < V_NEIGHBOR_ITERS; ++it)
#pragma unroll
            for (int u = 0; u < 12; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[u & 3], 0, 0, 0);
        keep = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
#elif V_NEIGHBOR == 2        // LDS traffic and workgroup barriers only
        f32x4* l4 = reinterpret_cast<f32x4*>(sRed);
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < V_NEIGHBOR_ITERS; ++it) {
            __syncthreads();
            l4[threadIdx.x] = t + f32x4{1.f, 2.f, 3.f, 4.f};
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) t += l4[(q * 64 + lane) & 511];
        }
        keep = t[0] + t[1] + t[2] + t[3];
#elif V_NEIGHBOR == 3        // global loads only (the job's own h1 stash, the matrix loop's A operand)
        const f32x4* H1 = reinterpret_cast<const f32x4*>(m.a[j].h1);
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < V_NEIGHBOR_ITERS; ++it) t += H1[((size_t)(blockIdx.x * 37 + it) * 512 + threadIdx.x) % ((size_t)V_ROWS * H / 4)];
        keep = t[0] + t[1] + t[2] + t[3];
#elif V_NEIGHBOR == 4        // plain vector arithmetic only
        float v[4] = {1.f, 2.f, 3.f, 4.f};
        for (int it = 0; it < V_NEIGHBOR_ITERS * 12; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_fmaf(v[u], 0.999f, 0.001f);
        keep = v[0] + v[1] + v[2] + v[3];
#endif
        if (keep == 12345.678f) m.a[j].slabs[0] = keep;          // (never true: keeps the loop alive)
#endif
    }
#else
    if (m.type[j] == 0) wgrad_body<8, 1>(m.a[j], sl, chunk, sRed);
    else wgrad_body<6, 2>(m.a[j], sl, chunk, sRed);
#endif
}

static float* dev_random(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n);
    srand(seed);
    for (size_t i = 0; i < n; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f * scale;
    float* d;
    hipMalloc(&d, n * sizeof(float));
    hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 1000;
    const int rows = V_ROWS;
    const long ngroups = (rows + GROUP - 1) / GROUP;
    Multi m;
    memset(&m, 0, sizeof(m));
    m.n_jobs = V_JOBS;
    int off = 0;
    size_t slab_floats[3], slab_off[3], total = 0;
    for (int j = 0; j < V_JOBS; ++j) {
#ifdef V_ALL_A
        const bool critic = true;
#else
        const bool critic = j < 2;
#endif
        WgradArgs& a = m.a[j];
        a.no_thin = 0;
        a.in_dim = critic ? 8 : 6; a.out_dim = critic ? 1 : 4; a.rows = rows;
        const int ou = critic ? 1 : 2;
        a.x.x0 = dev_random((size_t)rows * 6, 1.f, 10 + j); a.x.d0 = 6; a.x.ld0 = 6;
        a.x.x1 = critic ? dev_random((size_t)rows * 2, 1.f, 20 + j) : nullptr; a.x.d1 = critic ? 2 : 0; a.x.ld1 = 2;
        for (int i = 0; i < 16; ++i) a.x.scale[i] = i < 6 ? 0.5f + 0.1f * i : 1.f;
        a.x.n_scaled = 6;
        a.h1 = dev_random((size_t)ngroups * GROUP * H, 1.f, 30 + j);
        a.h2 = dev_random((size_t)ngroups * GROUP * H, 1.f, 40 + j);
        a.dz1 = dev_random((size_t)ngroups * GROUP * H, 1e-3f, 50 + j);
        a.dz2 = dev_random((size_t)ngroups * GROUP * H, 1e-3f, 60 + j);
        a.dz3 = dev_random((size_t)rows * ou, 1e-3f, 70 + j);
        a.groups_per_chunk = wgrad_groups_per_chunk(ngroups);
        const int nch = (int)((ngroups + a.groups_per_chunk - 1) / a.groups_per_chunk);
        m.type[j] = critic ? 0 : 1;
        m.chunk_off[j] = off;
        off += nch;
        slab_floats[j] = (size_t)nch * net_size(a.in_dim, a.out_dim);
        slab_off[j] = total;
        total += slab_floats[j];
    }
    for (int j = V_JOBS; j < 4; ++j) m.chunk_off[j] = off;
    float* slabs;
    hipMalloc(&slabs, total * sizeof(float));
    for (int j = 0; j < V_JOBS; ++j) m.a[j].slabs = slabs + slab_off[j];
    std::vector<float> first(total), cur(total);
    int bad = 0, bad_job[3] = {0, 0, 0};
    size_t min_diff = total, max_diff = 0;
    for (int l = 0; l < launches; ++l) {
        hipMemset(slabs, 0xff, total * sizeof(float));
        hipLaunchKernelGGL(k_repro, dim3(GRID_MULT * 8 * off), dim3(NTHREAD), 0, 0, m);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        hipMemcpy(cur.data(), slabs, total * sizeof(float), hipMemcpyDeviceToHost);
        if (l == 0) { first = cur; continue; }
        if (memcmp(first.data(), cur.data(), total * sizeof(float)) != 0) {
            ++bad;
            size_t nd = 0;
            for (int j = 0; j < V_JOBS; ++j) {
                size_t ndj = 0;
                for (size_t i = 0; i < slab_floats[j]; ++i) ndj += memcmp(&first[slab_off[j] + i], &cur[slab_off[j] + i], 4) != 0;
                if (ndj) ++bad_job[j];
                nd += ndj;
            }
            if (bad <= 3) {            // where, and by how much: (job, chunk, array, index)
                int shown = 0;
                for (int j = 0; j < V_JOBS && shown < 24; ++j) {
                    const size_t ns = net_size(m.a[j].in_dim, m.a[j].out_dim);
                    for (size_t i = 0; i < slab_floats[j] && shown < 24; ++i)
                        if (memcmp(&first[slab_off[j] + i], &cur[slab_off[j] + i], 4) != 0) {
                            const size_t e = i % ns;
                            const int in = m.a[j].in_dim;
                            if (e < (size_t)in * H) printf("   launch %d job %d chunk %zu dW1[i=%zu][col=%zu] first %.9g now %.9g diff %.3g\n", l, j, i / ns, e / H, e % H, first[slab_off[j] + i], cur[slab_off[j] + i], cur[slab_off[j] + i] - first[slab_off[j] + i]);
                            else printf("   launch %d job %d chunk %zu other[%zu] first %.9g now %.9g\n", l, j, i / ns, e, first[slab_off[j] + i], cur[slab_off[j] + i]);
                            ++shown;
                        }
                }
            }
            if (nd < min_diff) min_diff = nd;
            if (nd > max_diff) max_diff = nd;
        }
    }
    printf("jobs %d rows %d grid %d: %d of %d launches differ from the first (%.2f %%); per job %d %d %d; differing floats per bad launch %zu .. %zu\n",
           V_JOBS, rows, 8 * off, bad, launches - 1, 100.0 * bad / (launches - 1), bad_job[0], bad_job[1], bad_job[2], bad ? min_diff : 0, max_diff);
    return 0;
}
