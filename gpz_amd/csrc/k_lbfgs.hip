// Device-resident L-BFGS memory for the caller of the objective (minFunc's lbfgsAdd.m / lbfgsProd.m).
//
// theta, the gradient and the search direction never leave the GPU; per iteration the host sees a handful of scalars.
// The two-loop recursion of lbfgsProd.m is a chain of 2k dependent dot products over p-vectors (p = 113 001 at c4,
// k = 100 corrections, 180 MB of S and Y): as kernels that is 2k launches, as one workgroup it is bound by a single
// CU's bandwidth.  It is evaluated here in its "vector-free" form: every vector of the recursion is a combination of
// the basis  b = [s_0..s_{k-1}, y_0..y_{k-1}, g],  so the recursion runs on the (2k+1) coefficients with the Gram matrix
// B = b'b — O(k^2) flops on the host — and the device does three bandwidth-bound passes per iteration:
//     add        [S Y]' [s y]      (new Gram rows/columns; the old entries are kept)
//     direction  [S Y]' g          then   d = [S Y g] * delta
// Same arithmetic as the two-loop up to summation order.  Skipped updates (y's <= 1e-10), the ring order and
// Hdiag = y's / y'y follow lbfgsAdd.m and minFunc.m:553-578.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <vector>

#include "../../include/gpz_hip.h"
#include "gpz_dev.h"

#define LB_ROWS 4096          // rows per workgroup of the dot kernel
#define LB_MAXC 258           // 2 * (corrections + 1)

#define LB_CG 4               // columns of [S | Y] per workgroup of the dot kernel
// part[chunk][c][q] = sum over the chunk's rows of V_c[r] * x_q[r];  V = [S | Y] (each p x cap, column-major), q < nx <= 3.
// Grid: (row chunks, groups of LB_CG columns).  (Until round 6 a workgroup walked ALL 2 cap columns of its rows, one after the other with
// two barriers each: at p = 4 601 - BASELINE config 2 - that was 2 workgroups x 202 columns = 298 us per call, twice per iteration,
// more than the 0.55 ms evaluation the optimiser drives; now 102 workgroups x 4 columns.  Same sums in the same order.)
__global__ __launch_bounds__(256) void k_lb_dots(const double *__restrict__ S, const double *__restrict__ Y, long p, int cap,
                                                  const double *__restrict__ x0, const double *__restrict__ x1,
                                                  const double *__restrict__ x2, int nx, double *__restrict__ part) {
    __shared__ double red[4][LB_CG][3];
    const long r0 = (long)blockIdx.x * LB_ROWS;
    const int c0 = (int)blockIdx.y * LB_CG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PT = LB_ROWS / 256;
    double xa[PT], xb[PT], xc[PT];
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const long r = r0 + tid + 256L * u;
        const bool ok = r < p;
        xa[u] = ok ? x0[r] : 0.0;
        xb[u] = (ok && nx > 1) ? x1[r] : 0.0;
        xc[u] = (ok && nx > 2) ? x2[r] : 0.0;
    }
#pragma unroll
    for (int cc = 0; cc < LB_CG; ++cc) {
        const int c = c0 + cc < 2 * cap ? c0 + cc : 2 * cap - 1;      // (a group past the last column repeats it: not written)
        const double *v = (c < cap) ? S + (size_t)c * p : Y + (size_t)(c - cap) * p;
        double a = 0.0, b = 0.0, d = 0.0;
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const long r = r0 + tid + 256L * u;
            const double vv = (r < p) ? v[r] : 0.0;
            a = fma(vv, xa[u], a); b = fma(vv, xb[u], b); d = fma(vv, xc[u], d);
        }
        a = wave_sum(a); b = wave_sum(b); d = wave_sum(d);
        if (lane == 0) { red[wave][cc][0] = a; red[wave][cc][1] = b; red[wave][cc][2] = d; }
    }
    __syncthreads();
    if (tid < 3 * LB_CG) {
        const int cc = tid / 3, q = tid % 3, c = c0 + cc;
        if (c < 2 * cap) part[((size_t)blockIdx.x * 2 * cap + c) * 3 + q] = red[0][cc][q] + red[1][cc][q] + red[2][cc][q] + red[3][cc][q];
    }
}
// out[e] = sum over chunks (fixed order)
__global__ void k_lb_sum(const double *__restrict__ part, int nchunk, int count, double *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    double s = 0.0;
    for (int q = 0; q < nchunk; ++q) s += part[(size_t)q * count + e];
    out[e] = s;
}
// plain dots of up to 3 pairs: out[q] = a_q . b_q   (one workgroup per 4096 rows -> part, then k_lb_sum)
__global__ __launch_bounds__(256) void k_lb_pairdots(const double *__restrict__ a0, const double *__restrict__ b0,
                                                      const double *__restrict__ a1, const double *__restrict__ b1,
                                                      const double *__restrict__ a2, const double *__restrict__ b2, long p,
                                                      double *__restrict__ part) {
    __shared__ double sh4[4];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (long r = (long)blockIdx.x * LB_ROWS + threadIdx.x; r < min(p, ((long)blockIdx.x + 1) * LB_ROWS); r += 256) {
        s0 = fma(a0[r], b0[r], s0);
        if (a1) s1 = fma(a1[r], b1[r], s1);
        if (a2) s2 = fma(a2[r], b2[r], s2);
    }
    s0 = block_sum_256(s0, sh4); __syncthreads();
    s1 = block_sum_256(s1, sh4); __syncthreads();
    s2 = block_sum_256(s2, sh4);
    if (threadIdx.x == 0) { part[blockIdx.x * 3] = s0; part[blockIdx.x * 3 + 1] = s1; part[blockIdx.x * 3 + 2] = s2; }
}
// s = t*d, y = g - g_old into column `col` of S, Y
__global__ void k_lb_store(double *__restrict__ S, double *__restrict__ Y, long p, int col, const double *__restrict__ g,
                           const double *__restrict__ g_old, double t, const double *__restrict__ d) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p) return;
    S[(size_t)col * p + r] = t * d[r];
    Y[(size_t)col * p + r] = g[r] - g_old[r];
}
// d = dg*g + sum_c ds[c]*S_c + dy[c]*Y_c     (coefficients in coef[0..2cap] = [ds | dy | dg])
// A workgroup takes 64 rows; its four waves a quarter of the columns each (every load is issued whatever its coefficient - a zero
// coefficient multiplies a zero, not the column's value, which may be a rejected pair's - so eight loads per lane are in flight), the
// four partial sums are added in a fixed order.  (One thread per row walking all 2 cap columns behind a branch each: 41 us at p = 4 601.)
__global__ __launch_bounds__(256) void k_lb_combine(const double *__restrict__ S, const double *__restrict__ Y, long p, int cap,
                                                     const double *__restrict__ coef, const double *__restrict__ g,
                                                     double *__restrict__ d) {
    __shared__ double cf[2 * LB_MAXC + 1];
    __shared__ double ps[4][64];
    for (int e = threadIdx.x; e < 2 * cap + 1; e += 256) cf[e] = coef[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * 64 + lane, rr = r < p ? r : p - 1;
    const int per = (2 * cap + 3) / 4, c0 = wave * per, c1 = min(2 * cap, c0 + per);
    double acc = 0.0;
#pragma unroll 8
    for (int c = c0; c < c1; ++c) {
        const double v = (c < cap) ? S[(size_t)c * p + rr] : Y[(size_t)(c - cap) * p + rr];
        const double k = cf[c];
        acc = fma(k, k != 0.0 ? v : 0.0, acc);
    }
    ps[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && r < p) d[r] = fma(cf[2 * cap], g[r], ((ps[0][lane] + ps[1][lane]) + ps[2][lane]) + ps[3][lane]);
}
// out4 = [g.d, max|g|, sum|g|, max|d|]  (single workgroup partials -> host reduces nblk records)
__global__ __launch_bounds__(256) void k_vec_stats(const double *__restrict__ g, const double *__restrict__ d, long p,
                                                    double *__restrict__ part) {
    __shared__ double sh4[4];
    __shared__ double shm[2][4];
    double gd = 0.0, sg = 0.0, mg = 0.0, md = 0.0;
    for (long r = (long)blockIdx.x * LB_ROWS + threadIdx.x; r < min(p, ((long)blockIdx.x + 1) * LB_ROWS); r += 256) {
        const double gv = g[r], dv = d ? d[r] : 0.0;
        gd = fma(gv, dv, gd);
        sg += fabs(gv);
        // NaN-propagating maxima (isLegal.m must see them)
        mg = (gv != gv || mg != mg) ? NAN : fmax(mg, fabs(gv));
        md = (dv != dv || md != md) ? NAN : fmax(md, fabs(dv));
    }
    gd = block_sum_256(gd, sh4); __syncthreads();
    sg = block_sum_256(sg, sh4);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double a = __shfl_xor(mg, off, 64), b = __shfl_xor(md, off, 64);
        mg = (a != a || mg != mg) ? NAN : fmax(mg, a);
        md = (b != b || md != md) ? NAN : fmax(md, b);
    }
    if (lane == 0) { shm[0][wave] = mg; shm[1][wave] = md; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mg = (shm[0][w] != shm[0][w] || mg != mg) ? NAN : fmax(mg, shm[0][w]);
            md = (shm[1][w] != shm[1][w] || md != md) ? NAN : fmax(md, shm[1][w]);
        }
        double *o = part + (size_t)blockIdx.x * 4;
        o[0] = gd; o[1] = mg; o[2] = sg; o[3] = md;
    }
}
__global__ void k_vec_axpy(double *__restrict__ out, const double *__restrict__ x, double t, const double *__restrict__ d, long p) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < p) out[r] = fma(t, d[r], x[r]);
}

static thread_local char lb_err[256];
struct gpz_lbfgs {
    long p = 0;
    int cap = 0, nc = 0, spare = 0, device = 0, nchunk = 0;   // nc = cap + 1 physical columns: `spare` takes the candidate pair
    std::vector<int> order;                                       // physical columns of the stored pairs, oldest first
    hipStream_t st = nullptr;
    double *S = nullptr, *Y = nullptr, *part = nullptr, *red = nullptr, *coef = nullptr;
    double hdiag = 1.0;
    std::vector<double> SS, SY, YY;     // Gram blocks over physical columns: SS[a*nc+b] = s_a.s_b, SY[a*nc+b] = s_a.y_b, YY likewise
    std::vector<double> hbuf;
    std::vector<double> gdots;          // [S Y]' g of the g gpz_lbfgs_add was called with (the direction call that follows it takes them)
    const double *g_of_gdots = nullptr;
};
#define LBCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(lb_err, sizeof lb_err, "%s: %s", #x, hipGetErrorString(e_)); return GPZ_ERR_HIP; } } while (0)

extern "C" const char *gpz_lbfgs_last_error(void) { return lb_err; }

extern "C" int gpz_lbfgs_create(int64_t p, int32_t corrections, int32_t device, void *stream, gpz_lbfgs **out) {
    if (p < 1 || corrections < 1 || 2 * (corrections + 1) > LB_MAXC || !out) { snprintf(lb_err, sizeof lb_err, "gpz_lbfgs_create: bad argument (corrections <= %d)", LB_MAXC / 2 - 1); return GPZ_ERR_ARG; }
    LBCHK(hipSetDevice(device));
    gpz_lbfgs *h = new gpz_lbfgs();
    h->p = p; h->cap = corrections; h->nc = corrections + 1; h->device = device; h->st = (hipStream_t)stream;
    h->nchunk = (int)((p + LB_ROWS - 1) / LB_ROWS);
    const int nc = h->nc;
    const size_t cnt = (size_t)2 * nc * 3;
    if (hipMalloc((void **)&h->S, (size_t)p * nc * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&h->Y, (size_t)p * nc * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&h->part, (size_t)h->nchunk * cnt * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&h->red, cnt * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&h->coef, (size_t)(2 * nc + 1) * sizeof(double)) != hipSuccess) {
        snprintf(lb_err, sizeof lb_err, "gpz_lbfgs_create: hipMalloc failed");
        gpz_lbfgs_destroy(h);
        return GPZ_ERR_ALLOC;
    }
    (void)hipMemsetAsync(h->S, 0, (size_t)p * nc * sizeof(double), h->st);
    (void)hipMemsetAsync(h->Y, 0, (size_t)p * nc * sizeof(double), h->st);
    h->SS.assign((size_t)nc * nc, 0.0);
    h->SY = h->SS; h->YY = h->SS;
    h->hbuf.assign(cnt, 0.0);
    *out = h;
    return GPZ_OK;
}
extern "C" void gpz_lbfgs_destroy(gpz_lbfgs *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->st);
    for (double *q : {h->S, h->Y, h->part, h->red, h->coef}) if (q) (void)hipFree(q);
    delete h;
}

// lbfgsAdd.m: y = g - g_old, s = t*d; the pair is skipped when y's <= 1e-10.
extern "C" int gpz_lbfgs_add(gpz_lbfgs *h, const double *g_dev, const double *g_old_dev, double t, const double *d_dev,
                             int32_t *added) {
    if (!h || !g_dev || !g_old_dev || !d_dev) { snprintf(lb_err, sizeof lb_err, "gpz_lbfgs_add: null argument"); return GPZ_ERR_ARG; }
    LBCHK(hipSetDevice(h->device));
    const int nc = h->nc, slot = h->spare;   // the candidate goes to the spare column: a skipped pair must not destroy a stored one
    const long p = h->p;
    hipLaunchKernelGGL(k_lb_store, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, h->st, h->S, h->Y, p, slot, g_dev, g_old_dev, t, d_dev);
    const double *s = h->S + (size_t)slot * p, *y = h->Y + (size_t)slot * p;
    // the third vector is g itself: minFunc asks for the direction at this g next (gpz_lbfgs_direction), and [S Y]' g costs nothing
    // beside [S Y]' [s y] - one pass over the memory, one read-back and one synchronisation less per iteration
    hipLaunchKernelGGL(k_lb_dots, dim3(h->nchunk, (2 * nc + LB_CG - 1) / LB_CG), dim3(256), 0, h->st, (const double *)h->S, (const double *)h->Y, p, nc, s, y,
                       g_dev, 3, h->part);
    const int cnt = 2 * nc * 3;
    hipLaunchKernelGGL(k_lb_sum, dim3((cnt + 255) / 256), dim3(256), 0, h->st, (const double *)h->part, h->nchunk, cnt, h->red);
    LBCHK(hipMemcpyAsync(h->hbuf.data(), h->red, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, h->st));
    LBCHK(hipStreamSynchronize(h->st));
    // hbuf[c*3 + q]: column c of [S|Y] dotted with s (q = 0) / y (q = 1) / g (q = 2)
    h->gdots.resize((size_t)2 * nc);
    for (int c = 0; c < 2 * nc; ++c) h->gdots[c] = h->hbuf[(size_t)c * 3 + 2];
    h->g_of_gdots = g_dev;
    const double ys = h->hbuf[(size_t)(nc + slot) * 3 + 0];        // y_slot . s
    const double yy = h->hbuf[(size_t)(nc + slot) * 3 + 1];        // y_slot . y
    if (!(ys > 1e-10)) {                                           // lbfgsAdd.m:3
        if (added) *added = 0;
        return GPZ_OK;
    }
    for (int c = 0; c < nc; ++c) {
        const double ss = h->hbuf[(size_t)c * 3 + 0], ycs = h->hbuf[(size_t)(nc + c) * 3 + 0];    // s_c.s , y_c.s
        const double scy = h->hbuf[(size_t)c * 3 + 1], yyc = h->hbuf[(size_t)(nc + c) * 3 + 1];   // s_c.y , y_c.y
        h->SS[(size_t)slot * nc + c] = ss; h->SS[(size_t)c * nc + slot] = ss;
        h->SY[(size_t)slot * nc + c] = ycs;             // s_slot . y_c
        h->SY[(size_t)c * nc + slot] = scy;             // s_c . y_slot
        h->YY[(size_t)slot * nc + c] = yyc; h->YY[(size_t)c * nc + slot] = yyc;
    }
    if ((int)h->order.size() < h->cap) {
        h->order.push_back(slot);
        h->spare = (int)h->order.size();               // columns are handed out in order while the memory fills up
    } else {
        h->spare = h->order.front();                   // the oldest pair's column becomes the next scratch column
        h->order.erase(h->order.begin());
        h->order.push_back(slot);
    }
    h->hdiag = ys / yy;
    if (added) *added = 1;
    return GPZ_OK;
}

// lbfgsProd.m: d = -H*g through the two-loop recursion in coefficient space.
extern "C" int gpz_lbfgs_direction(gpz_lbfgs *h, const double *g_dev, double *d_dev) {
    if (!h || !g_dev || !d_dev) { snprintf(lb_err, sizeof lb_err, "gpz_lbfgs_direction: null argument"); return GPZ_ERR_ARG; }
    LBCHK(hipSetDevice(h->device));
    const int nc = h->nc, k = (int)h->order.size();
    const long p = h->p;
    if (h->g_of_gdots == g_dev && (int)h->gdots.size() == 2 * nc) {   // the g of the gpz_lbfgs_add just before: its dots are here already
        for (int c = 0; c < 2 * nc; ++c) h->hbuf[(size_t)c * 3] = h->gdots[c];
    } else {
        hipLaunchKernelGGL(k_lb_dots, dim3(h->nchunk, (2 * nc + LB_CG - 1) / LB_CG), dim3(256), 0, h->st, (const double *)h->S, (const double *)h->Y, p, nc, g_dev,
                           (const double *)nullptr, (const double *)nullptr, 1, h->part);
        const int cnt = 2 * nc * 3;
        hipLaunchKernelGGL(k_lb_sum, dim3((cnt + 255) / 256), dim3(256), 0, h->st, (const double *)h->part, h->nchunk, cnt, h->red);
        LBCHK(hipMemcpyAsync(h->hbuf.data(), h->red, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, h->st));
        LBCHK(hipStreamSynchronize(h->st));
    }
    h->g_of_gdots = nullptr;   // (one use: the same address may hold another vector at the next call)
    std::vector<double> ds(nc, 0.0), dy(nc, 0.0), al(nc, 0.0);
    double dg = -1.0;                                                          // q = -g
    auto sg = [&](int c) { return h->hbuf[(size_t)c * 3]; };                   // s_c . g
    auto yg = [&](int c) { return h->hbuf[(size_t)(nc + c) * 3]; };            // y_c . g
    auto dot_s = [&](int i) {                                                  // s_i . q
        double v = dg * sg(i);
        for (int c : h->order) v += ds[c] * h->SS[(size_t)i * nc + c] + dy[c] * h->SY[(size_t)i * nc + c];
        return v;
    };
    auto dot_y = [&](int i) {                                                  // y_i . q
        double v = dg * yg(i);
        for (int c : h->order) v += ds[c] * h->SY[(size_t)c * nc + i] + dy[c] * h->YY[(size_t)i * nc + c];
        return v;
    };
    for (int q = k - 1; q >= 0; --q) {                                         // lbfgsProd.m:17-20
        const int i = h->order[q];
        al[i] = dot_s(i) / h->SY[(size_t)i * nc + i];
        dy[i] -= al[i];
    }
    dg *= h->hdiag;                                                            // :23
    for (int c : h->order) { ds[c] *= h->hdiag; dy[c] *= h->hdiag; }
    for (int q = 0; q < k; ++q) {                                              // :25-28
        const int i = h->order[q];
        const double be = dot_y(i) / h->SY[(size_t)i * nc + i];
        ds[i] += al[i] - be;
    }
    std::vector<double> coef(2 * nc + 1, 0.0);
    for (int c : h->order) { coef[c] = ds[c]; coef[nc + c] = dy[c]; }
    coef[2 * nc] = dg;
    LBCHK(hipMemcpyAsync(h->coef, coef.data(), coef.size() * sizeof(double), hipMemcpyHostToDevice, h->st));
    hipLaunchKernelGGL(k_lb_combine, dim3((unsigned)((p + 63) / 64)), dim3(256), 0, h->st, (const double *)h->S, (const double *)h->Y, p,
                       nc, (const double *)h->coef, g_dev, d_dev);
    LBCHK(hipStreamSynchronize(h->st));   // coef is a stack vector
    return GPZ_OK;
}

// out[4] = [g.d, max|g|, sum|g|, max|d|] (NaN-propagating maxima); d may be NULL.
extern "C" int gpz_vec_stats(const double *g_dev, const double *d_dev, int64_t p, int32_t device, void *stream, double out[4]) {
    if (!g_dev || p < 1 || !out) return GPZ_ERR_ARG;
    LBCHK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)((p + LB_ROWS - 1) / LB_ROWS);
    // per-thread scratch for the block partials, grown on demand and kept (a hipMalloc per call would cost more than the
    // reduction); one device per calling thread is the supported use
    static thread_local double *part = nullptr;
    static thread_local int part_cap = 0, part_dev = -1;
    if (nb > part_cap || part_dev != device) {
        if (part) (void)hipFree(part);
        part = nullptr; part_cap = 0;
        LBCHK(hipMalloc((void **)&part, (size_t)nb * 4 * sizeof(double)));
        part_cap = nb; part_dev = device;
    }
    hipLaunchKernelGGL(k_vec_stats, dim3(nb), dim3(256), 0, st, g_dev, d_dev, (long)p, part);
    std::vector<double> hb((size_t)nb * 4);
    hipError_t e = hipMemcpyAsync(hb.data(), part, hb.size() * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { snprintf(lb_err, sizeof lb_err, "gpz_vec_stats: %s", hipGetErrorString(e)); return GPZ_ERR_HIP; }
    double gd = 0.0, mg = 0.0, sg = 0.0, md = 0.0;
    for (int b = 0; b < nb; ++b) {
        gd += hb[b * 4]; sg += hb[b * 4 + 2];
        mg = (hb[b * 4 + 1] != hb[b * 4 + 1] || mg != mg) ? NAN : fmax(mg, hb[b * 4 + 1]);
        md = (hb[b * 4 + 3] != hb[b * 4 + 3] || md != md) ? NAN : fmax(md, hb[b * 4 + 3]);
    }
    out[0] = gd; out[1] = mg; out[2] = sg; out[3] = md;
    return GPZ_OK;
}
// out = x + t*d  (out may alias x)
extern "C" int gpz_vec_axpy(double *out_dev, const double *x_dev, double t, const double *d_dev, int64_t p, int32_t device,
                            void *stream) {
    if (!out_dev || !x_dev || !d_dev || p < 1) return GPZ_ERR_ARG;
    LBCHK(hipSetDevice(device));
    hipLaunchKernelGGL(k_vec_axpy, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out_dev, x_dev, t, d_dev, (long)p);
    return GPZ_OK;
}
