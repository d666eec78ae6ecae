// image_amd/csrc/knn.hip -- exact k-nearest-neighbour search between two descriptor sets.
//
// The step the reference's users run right after image_surf(): image.dlib/README.md:19-37 matches the `surf` matrices of
// two images with FNN::get.knnx(sp1$surf, sp2$surf, k = 1).  FNN (CRAN, not part of /root/reference) answers with the
// exact k nearest rows under the Euclidean distance; this is the same query by brute force: every query row against
// every data row, squared differences added in ascending dimension order in f64 (no contraction), sqrt at the end;
// equal distances are ranked by ascending data index.
//
// One workgroup = 4 waves x 2 or 4 queries; a tile of 64 data rows is staged TRANSPOSED in LDS (dimension-major, so lane j
// reads row j conflict-free) and shared by the workgroup's queries.  Lane j of a wave sees the data rows j, j+64, ... and keeps a
// sorted list of its k best per query in LDS; at the end the 64 lists of a query are merged by k rounds of a wave-wide
// lexicographic (distance, index) minimum.
#include "common.h"

#include <math.h>

#include <algorithm>

#define KNN_KMAX 8
#define KNN_DMAX 64
#define KNN_WAVES 4

struct KnnView {
    const double *p;
    long long rs, cs;  // element (i, j) = p[i*rs + j*cs]
    long long n;
};

// KNN_QPW queries per wave (independent accumulators per lane): 4 for large query sets, 2 when the grid would otherwise
// not fill the device
template <int KNN_QPW>
__global__ void __launch_bounds__(64 * KNN_WAVES) knn_kernel(KnnView data, KnnView query, int dim, int k,
                                                              int *__restrict__ nn_index, double *__restrict__ nn_dist)
{
    constexpr int KNN_QPB = KNN_QPW * KNN_WAVES;
    __shared__ double tile[KNN_DMAX][65];                 // [dimension][data row of the tile]
    __shared__ double qv[KNN_QPB][KNN_DMAX];
    // per query, per slot, per lane: squared distance and index -- sized by the k of this launch (dynamic LDS), so that
    // the usual k = 1, 2 leave room for three workgroups per CU
    HIP_DYNAMIC_SHARED(double, knn_dyn)
    double *ld = knn_dyn;                                                   // [KNN_QPB][k][64]
    int *li = reinterpret_cast<int *>(knn_dyn + (size_t)KNN_QPB * k * 64);  // [KNN_QPB][k][64]
#define LD(q, s_, l) ld[((q) * k + (s_)) * 64 + (l)]
#define LI(q, s_, l) li[((q) * k + (s_)) * 64 + (l)]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long q0 = (long long)blockIdx.x * KNN_QPB;
    for (int e = tid; e < KNN_QPB * dim; e += blockDim.x) {
        const int ql = e / dim, t = e % dim;
        const long long q = q0 + ql;
        qv[ql][t] = q < query.n ? query.p[q * query.rs + t * query.cs] : 0.0;
    }
    for (int e = tid; e < KNN_QPB * k * 64; e += blockDim.x) {
        ld[e] = INFINITY;
        li[e] = 0x7fffffff;
    }
    __syncthreads();
    for (long long j0 = 0; j0 < data.n; j0 += 64) {
        // stage 64 data rows; consecutive threads walk the faster-varying direction of the source
        for (int e = tid; e < 64 * dim; e += blockDim.x) {
            int r, t;
            if (data.cs == 1) { r = e / dim; t = e % dim; } else { t = e / 64; r = e % 64; }
            const long long j = j0 + r;
            tile[t][r] = j < data.n ? data.p[j * data.rs + t * data.cs] : 0.0;
        }
        __syncthreads();
        const long long j = j0 + lane;
        if (j < data.n) {
            double acc[KNN_QPW];
#pragma unroll
            for (int u = 0; u < KNN_QPW; u++) acc[u] = 0.0;
            for (int t = 0; t < dim; t++) {
                const double v = tile[t][lane];
#pragma unroll
                for (int u = 0; u < KNN_QPW; u++) {
                    const double df = qv[wave * KNN_QPW + u][t] - v;
                    acc[u] += df * df;
                }
            }
#pragma unroll
            for (int u = 0; u < KNN_QPW; u++) {
                const int ql = wave * KNN_QPW + u;
                if (acc[u] < LD(ql, k - 1, lane)) {  // beats this lane's k-th best: insert (equal distances keep the earlier row)
                    int s = k - 1;
                    while (s > 0 && acc[u] < LD(ql, s - 1, lane)) {
                        LD(ql, s, lane) = LD(ql, s - 1, lane);
                        LI(ql, s, lane) = LI(ql, s - 1, lane);
                        s--;
                    }
                    LD(ql, s, lane) = acc[u];
                    LI(ql, s, lane) = (int)j;
                }
            }
        }
        __syncthreads();
    }
    // merge: k rounds of "smallest head over the 64 lanes"
    for (int u = 0; u < KNN_QPW; u++) {
        const int ql = wave * KNN_QPW + u;
        const long long q = q0 + ql;
        int head = 0;
        for (int round = 0; round < k; round++) {
            double d = head < k ? LD(ql, head, lane) : INFINITY;
            int ix = head < k ? LI(ql, head, lane) : 0x7fffffff;
            double bd = d;
            int bi = ix;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const double od = __shfl_xor(bd, m);
                const int oi = __shfl_xor(bi, m);
                if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
            }
            if (bi == ix && bd == d && bi != 0x7fffffff) head++;  // this lane's head won (indices are unique)
            if (lane == 0 && q < query.n) {
                const bool have = bi != 0x7fffffff;
                nn_index[q * k + round] = have ? bi : -1;
                nn_dist[q * k + round] = have ? sqrt(bd) : INFINITY;
            }
        }
    }
}

#undef LD
#undef LI

static imgfd_status knn_launch(imgfd_ctx *ctx, const KnnView &data, const KnnView &query, int dim, int k, int *d_index,
                               double *d_dist)
{
    if (query.n < 1) return IMGFD_OK;
    const bool wide = query.n >= 16 * 1024 / 2;  // >= 512 workgroups of 16 queries: two per CU
    const int qpb = (wide ? 4 : 2) * KNN_WAVES;
    const long long blocks = (query.n + qpb - 1) / qpb;
    const size_t dyn = (size_t)qpb * k * 64 * (sizeof(double) + sizeof(int));
    if (wide) {
        IMGFD_HIP(ctx, hipFuncSetAttribute((const void *)knn_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(knn_kernel<4>, dim3((unsigned)blocks), dim3(64 * KNN_WAVES), dyn, ctx->stream, data, query, dim, k, d_index, d_dist);
    } else {
        IMGFD_HIP(ctx, hipFuncSetAttribute((const void *)knn_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(knn_kernel<2>, dim3((unsigned)blocks), dim3(64 * KNN_WAVES), dyn, ctx->stream, data, query, dim, k, d_index, d_dist);
    }
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

static bool knn_args_ok(int64_t n_data, int64_t n_query, int dim, int k)
{
    return n_data >= 0 && n_query >= 0 && n_data < 0x7fffffff && n_query < 0x7fffffff && dim >= 1 && dim <= KNN_DMAX && k >= 1 &&
           k <= KNN_KMAX;
}

extern "C" {

imgfd_status imgfd_knn_dev(imgfd_ctx *ctx, const double *d_data, int64_t n_data, int64_t data_row_stride,
                           int64_t data_col_stride, const double *d_query, int64_t n_query, int64_t query_row_stride,
                           int64_t query_col_stride, int dim, int k, int32_t *d_nn_index, double *d_nn_dist)
{
    if (!ctx) return IMGFD_ERR_INVALID;
    if (!knn_args_ok(n_data, n_query, dim, k) || (n_data && !d_data) || (n_query && (!d_query || !d_nn_index || !d_nn_dist)))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_knn_dev: bad argument (1 <= dim <= 64, 1 <= k <= 8)");
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const KnnView dv{d_data, data_row_stride, data_col_stride, n_data}, qv{d_query, query_row_stride, query_col_stride, n_query};
    return knn_launch(ctx, dv, qv, dim, k, d_nn_index, d_nn_dist);
}

imgfd_status imgfd_knn(imgfd_ctx *ctx, const double *data, int64_t n_data, const double *query, int64_t n_query, int dim,
                       int k, int column_major, int32_t *nn_index, double *nn_dist)
{
    if (!ctx) return IMGFD_ERR_INVALID;
    if (!knn_args_ok(n_data, n_query, dim, k) || (n_data && !data) || (n_query && (!query || !nn_index || !nn_dist)))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_knn: bad argument (1 <= dim <= 64, 1 <= k <= 8)");
    if (!n_query) return IMGFD_OK;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bd = sizeof(double) * (size_t)n_data * dim, bq = sizeof(double) * (size_t)n_query * dim;
    const size_t bi = sizeof(int32_t) * (size_t)n_query * k, bo = sizeof(double) * (size_t)n_query * k;
    IMGFD_TRY(ws_reserve(ctx, align_up(bd, 256) + align_up(bq, 256) + align_up(bi, 256) + align_up(bo, 256) + 1024));
    double *d_data = (double *)ws_alloc(ctx, std::max<size_t>(bd, 8)), *d_query = (double *)ws_alloc(ctx, bq);
    int *d_index = (int *)ws_alloc(ctx, bi);
    double *d_dist = (double *)ws_alloc(ctx, bo);
    if (!d_data || !d_query || !d_index || !d_dist) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    if (bd) IMGFD_HIP(ctx, hipMemcpyAsync(d_data, data, bd, hipMemcpyHostToDevice, ctx->stream));
    IMGFD_HIP(ctx, hipMemcpyAsync(d_query, query, bq, hipMemcpyHostToDevice, ctx->stream));
    const KnnView dv{d_data, column_major ? 1 : dim, column_major ? n_data : 1, n_data};
    const KnnView qv{d_query, column_major ? 1 : dim, column_major ? n_query : 1, n_query};
    IMGFD_TRY(knn_launch(ctx, dv, qv, dim, k, d_index, d_dist));
    IMGFD_HIP(ctx, hipMemcpyAsync(nn_index, d_index, bi, hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipMemcpyAsync(nn_dist, d_dist, bo, hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IMGFD_OK;
}

}  // extern "C"
