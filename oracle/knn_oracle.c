/*
 * oracle/knn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Exact k-nearest-neighbour search by exhaustive comparison: the query the reference's README runs on two images'
 * SURF descriptors, FNN::get.knnx(sp1$surf, sp2$surf, k = 1) (image.dlib/README.md:19-37).
 *
 * PARITY UNPINNED against FNN itself: FNN (CRAN, version unpinned by the reference -- it is only named in the README,
 * not in DESCRIPTION) is not part of /root/reference and is not installed here.  Its documented contract -- the k
 * nearest rows of `data` for every row of `query` under the Euclidean distance, nn.dist ascending -- is restated; the
 * restatement is checked against an independent exact search (scipy.spatial.cKDTree) in tests/test_knn.py.  Squared
 * differences are added in ascending dimension order, sqrt once; ties rank by ascending data index.
 */
#include <math.h>
#include <stdlib.h>

#define ORC_API __attribute__((visibility("default")))

/* data: nd x dim, query: nq x dim, both row-major; idx (0-based, -1 = none) and dist: nq x k */
ORC_API void orc_knn(const double *data, long nd, const double *query, long nq, int dim, int k, int *idx, double *dist)
{
    double *bd = (double *)malloc(sizeof(double) * (size_t)k);
    for (long q = 0; q < nq; q++) {
        int have = 0;
        int *bi = idx + q * k;
        for (long j = 0; j < nd; j++) {
            double acc = 0.0;
            for (int t = 0; t < dim; t++) {
                const double df = query[q * dim + t] - data[j * dim + t];
                acc += df * df;
            }
            /* insert after every entry that is <= acc: earlier rows win ties */
            int s = have;
            if (have == k) {
                if (!(acc < bd[k - 1])) continue;
                s = k - 1;
            } else {
                have++;
            }
            while (s > 0 && acc < bd[s - 1]) { bd[s] = bd[s - 1]; bi[s] = bi[s - 1]; s--; }
            bd[s] = acc;
            bi[s] = (int)j;
        }
        for (int s = 0; s < k; s++) {
            if (s < have) dist[q * k + s] = sqrt(bd[s]);
            else { dist[q * k + s] = INFINITY; bi[s] = -1; }
        }
    }
    free(bd);
}
